out=gpurun_out/r3full; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $out/tests.log | cut -c1-300
