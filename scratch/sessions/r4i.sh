#!/usr/bin/env bash
out=gpurun_out/r4i; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -30 $out/smoke.log; }
tail -1 $out/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -8 $out/pytest.log
for cfg in "-" "GCC_BN_BWD_TAIL=1"; do
  [ "$cfg" = "-" ] && envs="" || envs="$cfg"
  echo "== $cfg" | tee -a $out/unet_ab.txt
  env $envs timeout 300 python scratch/unet_ab.py 2>/dev/null | grep "U-Net" | tee -a $out/unet_ab.txt
done
bash scratch/ab_quick.sh r4i "-" "GCC_BN_BWD_TAIL=1"
timeout 900 python bench.py --no-other-configs --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4i/bench.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], d['launches_per_step'], 'frac', r['frac'], 'alone', r.get('frac_alone_plan'))
for k,v in r['per_kernel'].items(): print(' ', k, v)
print(' alone', r['alone_plan']['per_kernel'])
print(r['generator'])
print(r.get('conv_roofline'))
PY
