#!/usr/bin/env bash
# production step with fewer workgroups in the BatchNorm / activation streaming kernels (do they leave CUs to the other streams' convs?)
run() { env GCC_BENCH_ALLOW_OPTIONS=1 "$@" python bench.py --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.1f images/s' % (d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
echo "default                     $(run)"
echo "BN_MAXBLK=512               $(run GCC_BN_MAXBLK=512)"
echo "BN_MAXBLK=256               $(run GCC_BN_MAXBLK=256)"
echo "BN_REDUCE_CAP=256           $(run GCC_BN_REDUCE_CAP=256)"
echo "MAXBLK=256 REDUCE_CAP=256   $(run GCC_BN_MAXBLK=256 GCC_BN_REDUCE_CAP=256)"
echo "MAXBLK=128 REDUCE_CAP=128   $(run GCC_BN_MAXBLK=128 GCC_BN_REDUCE_CAP=128)"
done
