#!/usr/bin/env bash
# one GPU-box session: parity tests, bench line, rocprof kernel stats.  usage: scratch/gpu_round.sh <tag> [what...]
tag=$1; shift
what=${*:-tests bench prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# fail fast: a Python error in the model path must not burn GPU minutes in every later stage
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -30 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
for w in $what; do
case $w in
tests)
  export GCC_TEST_REPORT=$PWD/$out/test_report.txt; rm -f $GCC_TEST_REPORT
  timeout 3000 python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -5 $out/pytest.log;;
bench)
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; cat $out/bench.json | head -c 3000;;
prof)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_serialized -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --serialize-streams > $GRAFT_REPO_ROOT/$out/prof_serialized.log 2>&1)
  find $out/prof_serialized -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_serialized.csv
  find $out/prof_serialized -name '*kernel_trace.csv' -delete
  head -25 $out/kernel_stats_serialized.csv;;
profdefault)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline > $GRAFT_REPO_ROOT/$out/prof_default.log 2>&1)
  find $out/prof_default -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_default.csv
  f=$(find $out/prof_default -name '*kernel_trace.csv' | head -1)
  python scratch/trace_waits.py $f > $out/trace_waits_default.txt 2>&1
  python scratch/trace_summary.py $f > $out/trace_summary_default.txt 2>&1
  python scratch/trace_fill.py $f > $out/trace_fill_default.txt 2>&1
  python scratch/trace_timeline.py $f 22 100 > $out/trace_timeline_default.txt 2>&1
  rm -f $f;;
shapes)
  GCC_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-configs > $out/shapes.json 2> $out/shapes.txt; grep -c . $out/shapes.txt;;
graph)
  timeout 600 python scratch/probe_graph_pix2pix.py > $out/graph_probe.txt 2>&1; tail -8 $out/graph_probe.txt;;
dp)
  timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu -x 2>&1 | tail -15;;
pmc)
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline --serialize-streams > $GRAFT_REPO_ROOT/$out/pmc$i.log 2>&1)
    f=$(find $out/pmc$i -name '*counter_collection.csv' | head -1); cp $f $out/pmc${i}_counters.csv 2>/dev/null
    rm -rf $out/pmc$i
  done
  python scratch/pmc_mfma.py $out/pmc_mfma_lds.json $out/pmc1_counters.csv $out/pmc2_counters.csv
  python scratch/pmc_traffic.py $out/pmc3_counters.csv $out/pmc4_counters.csv $out/igemm_hbm_traffic.json "${GCC_GIT_HEAD:-unknown}"
  rm -f $out/pmc*_counters.csv;;
counters)
  rocprofv3 -L > $out/counters_list.txt 2>&1; grep -c . $out/counters_list.txt;;
esac
done
