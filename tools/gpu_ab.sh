out=gpurun_out/r3x; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_pool_cyclosynch.py tests/test_gpu_cyclosynch.py -m gpu -q --durations=5 > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $out/tests.log | cut -c1-300
