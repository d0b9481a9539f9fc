"""bench.py's multi-GPU path itself -- the environment torch.distributed.run sets up, the process group, the max-over-ranks timing, the
`strong` and `shared_clock` objects -- started the way the driver starts it, as two ranks that share the one GPU of the test box
(MCRAT_BENCH_ONE_DEVICE=1: both ranks on cuda:0, gloo for the scalar exchanges because RCCL refuses two ranks on one device) with a
small mesh and few photons.  No scaling figure is asked of it: what is tested is that the line comes out and describes two ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_runs_as_two_ranks_under_torch_distributed_run():
    env = dict(os.environ, MCRAT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--photons", "20000", "--nzc", "8", "--rank-photons", "500", "--no-cpu-baseline", "--host-driver", "0",
           "--shared-clock-rounds", "40"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["scatter_events"] > 0
    assert out["config"]["photons_per_gpu"] == 20000 and "x2" in out["config"]["parallelism"]
    # weak scaling: both ranks' events are in the line (each rank holds its own 20 000 photons)
    assert out["photon_steps_per_s"] > 0 and out["loop_passes"] > 0
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and 0 < roof["frac_headline"] < 1 and roof["peak"] == 8000.0
    strong = out["strong"]
    assert "error" not in strong, strong
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["photons_total"] == 20000 and strong["value"] > 0
    assert strong["lists_per_gpu"] == 20                          # 40 lists of 500 photons dealt out to two ranks
    sc = out["shared_clock"]
    assert "error" not in sc, sc
    assert sc["photons_total"] == 40000 and sc["rounds"] == 40 and sc["loop_passes"] > 0 and sc["scatter_events_per_s"] > 0
    fast = out["fast_mode"]
    assert fast is None or "error" not in fast, fast
