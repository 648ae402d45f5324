#!/usr/bin/env bash
out=gpurun_out/r4d; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -30 $out/smoke.log; }
tail -1 $out/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -12 $out/pytest.log
timeout 600 python scratch/ab_shapes.py fd "D.L" hc0:HALO_HC=0 hc1:HALO_HC=1 hc128:HALO_HC=128 2>&1 | tee $out/halo_hc_shapes.txt
timeout 600 python scratch/ab_shapes.py fd "tG.d" hc0:HALO_HC=0 hc128:HALO_HC=128 nohalo:IGEMM_HALO=0 2>&1 | tee -a $out/halo_hc_shapes.txt
for cfg in "-" "GCC_IN_CONV_FINALIZE=1" "GCC_FUSE_BN=2" "GCC_FUSE_BN=2 GCC_IN_CONV_FINALIZE=1" "GCC_BN_BWD_TAIL=1"; do
  [ "$cfg" = "-" ] && envs="" || envs="$cfg"
  echo "== $cfg" | tee -a $out/unet_ab.txt
  env $envs timeout 300 python scratch/unet_ab.py 2>/dev/null | grep "U-Net" | tee -a $out/unet_ab.txt
done
bash scratch/ab_quick.sh r4d "-" "GCC_IN_CONV_FINALIZE=1" "GCC_BN_BWD_TAIL=1" "GCC_TEACHER_EARLY_DREAL=2" "GCC_HALO_HC=1" "GCC_HALO_HC=128"
