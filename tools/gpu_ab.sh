out=gpurun_out/r3aa; rm -rf $out; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_output.py tests/test_gpu_pool.py -m gpu -q -k "converts_the_resident or profile_totals or outbox" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $out/tests.log | cut -c1-300
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.txt
