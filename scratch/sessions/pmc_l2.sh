out=gpurun_out/${1:-r02r}; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $GRAFT_REPO_ROOT/$out/l2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --serialize-streams > $GRAFT_REPO_ROOT/$out/l2.log 2>&1)
c=$(find $out/l2 -name '*counter_collection.csv' | head -1); t=$(find $out/l2 -name '*kernel_trace.csv' | head -1)
python scratch/pmc_l2.py $c $t | tee $out/pmc_l2.txt
head -2 $c
rm -rf $out/l2
