#!/usr/bin/env bash
# SRGAN 96 -> 384, serialized: the longest dispatches of the streaming (non-MFMA) kernels with their grid sizes
out=gpurun_out/r5_srgan5; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py srgan_96_to_384 2 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
f=$(find $out/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $out/slow_streaming.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'igemm' in name or 'wgrad_kernel' in name or 'ring3' in name or 'thinout' in name: continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[name[:70]].append((d, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))
tot = sorted(((sum(x[0] for x in v), k) for k, v in agg.items()), reverse=True)[:14]
for t, k in tot:
    v = sorted(agg[k], reverse=True)
    print('%-72s n=%4d total %8.1f us  top: %s' % (k, len(v), t, ', '.join('%.0f us grid %s/%s' % x for x in v[:4])))
PY
rm -rf $out/prof
cat $out/slow_streaming.txt
