# scratch script for GPU calls during development (the round's measurements are tools/profile_round.sh)
timeout -k 10 400 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_rank_pool_host.py -m gpu -q 2>&1 | tail -4
