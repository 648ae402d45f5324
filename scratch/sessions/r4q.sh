#!/usr/bin/env bash
out=gpurun_out/r4q; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "halo_conv" -p no:cacheprovider 2>&1 | tail -5
for v in 0 1; do
  echo "== GCC_HALO_XCD_COLS=$v"
  GCC_HALO_XCD_COLS=$v python scratch/bench_igemm.py 30 "D.L4" fd
done
# HBM fetch bytes of the L4 forward / data gradient under both orders (PMC pass of its own)
for v in 0 1; do
  (cd /tmp && GCC_HALO_XCD_COLS=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc$v -- python3 $GRAFT_REPO_ROOT/scratch/bench_igemm.py 5 "D.L4" fd > $GRAFT_REPO_ROOT/$out/pmc$v.log 2>&1)
  f=$(find $out/pmc$v -name '*counter_collection.csv' | head -1)
  python - "$f" $v <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'halo' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        acc[r['Kernel_Name'][:60] + ' grid ' + r.get('Grid_Size', '?')].append(float(r['Counter_Value']))
for k, v in acc.items():
    print('XCD_COLS=%s  %-80s n=%d  FETCH_SIZE median %.0f (x64 B = %.1f MB uncorrected)' % (sys.argv[2], k, len(v), sorted(v)[len(v)//2], sorted(v)[len(v)//2] * 64 / 1e6))
PY
  rm -rf $out/pmc$v
done
bash scratch/ab_quick.sh r4q "-" "GCC_HALO_XCD_COLS=1" "GCC_HALO_XCD_COLS=2"
