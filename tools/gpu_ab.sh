out=gpurun_out/r3c; mkdir -p $out
python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -8 $out/tests.log
B="--pools 1 --no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0"
python bench.py $B > $out/bench_p1.json 2> $out/bench_p1.err; echo "bench rc=$?"
python tools/dbg_3dsph.py > $out/dbg.txt 2>&1; echo "dbg rc=$?"; cat $out/dbg.txt | cut -c1-400
hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gb 2> $out/gb_build.log && /tmp/gb > $out/gather.txt 2>&1; echo "gather rc=$?"; cat $out/gather.txt
