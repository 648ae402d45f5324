#!/usr/bin/env bash
out=gpurun_out/r4aa; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "forks" -p no:cacheprovider 2>&1 | tail -5
GCC_ARCH_EARLY=1 timeout 900 python -m pytest tests/test_pix2pix_gpu.py tests/test_replay_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
bash scratch/ab_quick.sh r4aa "-" "GCC_ARCH_EARLY=1"
i=2
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline --serialize-streams > $GRAFT_REPO_ROOT/$out/pmc$i.log 2>&1)
  f=$(find $out/pmc$i -name '*counter_collection.csv' | head -1); cp $f $out/pmc${i}_counters.csv 2>/dev/null
  rm -rf $out/pmc$i
done
python scratch/pmc_traffic.py $out/pmc3_counters.csv $out/pmc4_counters.csv $out/igemm_hbm_traffic.json "${GCC_GIT_HEAD:-unknown}" | cut -c1-600
rm -f $out/pmc*_counters.csv
