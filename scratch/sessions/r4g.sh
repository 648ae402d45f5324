#!/usr/bin/env bash
out=gpurun_out/r4g; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -30 $out/smoke.log; }
tail -1 $out/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -8 $out/pytest.log
for cfg in "-" "GCC_BN_BWD_GRID_EX=0" "GCC_BN_BWD_GRID_EX=2"; do
  [ "$cfg" = "-" ] && envs="" || envs="$cfg"
  echo "== $cfg" | tee -a $out/unet_ab.txt
  env $envs timeout 300 python scratch/unet_ab.py 2>/dev/null | grep "U-Net" | tee -a $out/unet_ab.txt
done
bash scratch/ab_quick.sh r4g "-" "GCC_BN_BWD_GRID_EX=0" "GCC_BN_BWD_GRID_EX=2"
bash scratch/unet_chain.sh $out student
