#!/usr/bin/env bash
out=gpurun_out/r4ar; mkdir -p $out; export TMPDIR=/tmp
w=srgan_96_to_384
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 6 > $GRAFT_REPO_ROOT/$out/$w.log 2>&1)
tail -1 $out/$w.log
f=$(find $out/$w -name '*kernel_stats.csv' | head -1)
cp $f $out/${w}_kernel_stats_serialized.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = 11.0
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('   kernels: %.3f ms per iteration (%d iterations), %d launches per iteration' % (tot / it / 1e6, it, calls / it))
for r in rows[:34]:
    print('   %-90s %6.1f calls %8.3f ms/iter avg %7.1f us' % (r['Name'][:90], int(r['Calls']) / it, float(r['TotalDurationNs']) / it / 1e6, float(r['AverageNs'])/1e3))
PY
find $out/$w -name '*kernel_trace.csv' -delete
