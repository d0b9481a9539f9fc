# scratch script for GPU calls during development (the round's measurements are tools/profile_round.sh)
out=gpurun_out/final; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log | cut -c1-300
