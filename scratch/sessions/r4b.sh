#!/usr/bin/env bash
out=gpurun_out/r4b; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -30 $out/smoke.log; }
tail -1 $out/smoke.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "finalize or conv_bn_act or instance_norm or batchnorm" -p no:cacheprovider > $out/pytest_new.log 2>&1; tail -15 $out/pytest_new.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -30 $out/pytest.log
timeout 900 python bench.py --no-other-configs --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; head -c 700 $out/bench.json; echo
GCC_IN_CONV_FINALIZE=0 timeout 900 python bench.py --no-other-configs --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | head -c 300; echo
timeout 900 python bench.py --no-other-configs --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | head -c 300; echo
bash scratch/unet_chain.sh $out student
bash scratch/unet_chain.sh $out teacher
