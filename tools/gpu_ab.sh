# scratch script for GPU calls during development (the round's measurements are tools/profile_round.sh)
out=gpurun_out/lund; rm -rf $out; mkdir -p $out
for w in 0 8 32 128; do MODE=fast WINDOWS=$w timeout -k 10 200 python tools/lundman_run.py > $out/fast_$w.txt 2>&1; echo "fast $w: $(tail -1 $out/fast_$w.txt)"; done
MODE=exact timeout -k 10 300 python tools/lundman_run.py > $out/exact.txt 2>&1; echo "exact: $(tail -1 $out/exact.txt)"
