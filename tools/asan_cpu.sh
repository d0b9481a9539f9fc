#!/bin/bash
# CPU-side sanitizer run (GPU AddressSanitizer is not available on the pool): the oracle and the host C mirror rebuilt with
# -fsanitize=address,undefined, the CPU test files that exercise them run under the preloaded runtimes, the regular builds
# restored afterwards.   tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
python -c "from mcrat_amd import build; build.build(); from mcrat_amd.host import build_host; build_host.build(); from oracle import oracle_py; oracle_py.build()"
cp oracle/liboracle.so /tmp/liboracle.keep; cp mcrat_amd/host/libmcrat_hip_host.so /tmp/libhost.keep
restore() { cp /tmp/liboracle.keep oracle/liboracle.so; cp /tmp/libhost.keep mcrat_amd/host/libmcrat_hip_host.so; touch oracle/liboracle.so mcrat_amd/host/libmcrat_hip_host.so; }
trap restore EXIT
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o oracle/liboracle.so \
    oracle/mcrat_oracle.c oracle/oracle_ingest.c oracle/oracle_cyclosynch.c oracle/oracle_rng.c -lm
gcc -std=gnu99 -O1 -g -Wall -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I include mcrat_amd/host/mcrat_hip_host.c \
    -o mcrat_amd/host/libmcrat_hip_host.so -L mcrat_amd -lmcrat_hip -Wl,-rpath,'$ORIGIN/..'
touch oracle/liboracle.so mcrat_amd/host/libmcrat_hip_host.so
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$ASAN:$UBSAN python -m pytest tests/test_oracle_kat.py tests/test_oracle_cyclosynch.py tests/test_golden.py tests/test_ingest_cpu.py \
    tests/test_h5_readers.py tests/test_host_c.py tests/test_output_cpu.py -q -m "not gpu" -p no:cacheprovider
