#!/usr/bin/env bash
out=gpurun_out/r4p; mkdir -p $out; export TMPDIR=/tmp
w=srgan_96_to_384
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 10 > $GRAFT_REPO_ROOT/$out/$w.log 2>&1)
tail -1 $out/$w.log
f=$(find $out/$w -name '*kernel_stats.csv' | head -1)
cp $f $out/${w}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('   kernels: %.3f ms per iteration (15 iterations), %d launches per iteration' % (tot / 15 / 1e6, calls / 15))
for r in rows[:30]:
    print('   %-90s %6.1f calls %8.3f ms/iter avg %7.1f us' % (r['Name'][:90], int(r['Calls']) / 15, float(r['TotalDurationNs']) / 15 / 1e6, float(r['AverageNs'])/1e3))
PY
find $out/$w -name '*kernel_trace.csv' -delete
