#!/usr/bin/env bash
# round 6, GPU session i: collector in the PatchGAN engine (batch-1 discriminators): full GPU suite, A/B of cyclegan, headline bench
out=gpurun_out/r6i; mkdir -p $out
export GCC_TEST_REPORT=$PWD/$out/test_report.txt; rm -f $GCC_TEST_REPORT
timeout 3000 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "not 384-16" 2>&1 | tail -4
bash scratch/ab_other.sh cyclegan "GCC_WGRAD_GROUP=0" "-" 2>&1 | tee $out/ab_other.txt
bash scratch/ab_quick.sh r6i_ab "GCC_WGRAD_GROUP=0" "-" 2>&1 | tee $out/ab.txt
