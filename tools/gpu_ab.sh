out=gpurun_out/r3tab; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hot_table.py tests/test_gpu_pool.py -m gpu -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $out/tests.log | cut -c1-300
