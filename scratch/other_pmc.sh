#!/usr/bin/env bash
# HBM traffic per launch of the CycleGAN iteration's kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
out=gpurun_out/r3z_cpmc; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py cyclegan 4 > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1)
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1); cp $f $out/c$i.csv; rm -rf $out/p$i
done
python - <<'PY'
import csv, collections
def load(p, name):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] != name: continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
        d[k][0] += float(r['Counter_Value']); d[k][1] += 1
    return d
f, w = load('gpurun_out/r3z_cpmc/c1.csv', 'FETCH_SIZE'), load('gpurun_out/r3z_cpmc/c2.csv', 'WRITE_SIZE')
rows = sorted(f.items(), key=lambda kv: -kv[1][0])[:14]
print('kernel                                            launches   fetch KB/launch   write KB/launch  (raw counter units: KB)')
for k, (v, n) in rows:
    wv, wn = w.get(k, [0.0, 1])
    print('%-50s %7d %16.1f %17.1f' % (k, n, v / n, wv / max(wn, 1)))
PY
rm -f $out/c1.csv $out/c2.csv
