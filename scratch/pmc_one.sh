#!/usr/bin/env bash
# usage: scratch/pmc_one.sh <outdir under gpurun_out> "<shape>" <f|d|w> [OPT=val ...]   -- L2 / fabric counters of one conv op
out=gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -- python3 $GRAFT_REPO_ROOT/scratch/run_one.py "$@" > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1)
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'igemm' in k or 'wgrad_kernel' in k or 'halo' in k:
        acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
  rm -rf $out/p$i
done
