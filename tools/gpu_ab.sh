out=gpurun_out/r3r; rm -rf $out; mkdir -p $out
timeout -k 10 800 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $out/tests.log | cut -c1-250
[ $rc -eq 124 ] && exit 1
MODE=fast WINDOWS=0 timeout -k 10 200 python tools/lundman_run.py > $out/lundman_fast_auto.txt 2>&1; echo "lundman rc=$?"; tail -25 $out/lundman_fast_auto.txt
