#!/usr/bin/env bash
# kernel-time sum per iteration of the launch-bound configs: scratch/other_prof.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for w in cyclegan sagan srgan; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 20 > $GRAFT_REPO_ROOT/$out/$w.log 2>&1)
  tail -1 $out/$w.log
  f=$(find $out/$w -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('   kernels: %.3f ms per iteration (25 iterations), %d launches per iteration' % (tot / 25 / 1e6, calls / 25))
for r in rows[:8]:
    print('   %-60s %6d calls %8.3f ms/iter' % (r['Name'][:60], int(r['Calls']) / 25, float(r['TotalDurationNs']) / 25 / 1e6))
PY
  find $out/$w -name '*kernel_trace.csv' -delete
done
