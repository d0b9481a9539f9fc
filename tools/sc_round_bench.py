"""Round time of the shared-clock mode on one GPU (10^6 photons, cfg2): the Python host loop, the C loop, the C loop with its rounds in a hipGraph."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mcrat_amd import engine, synth  # noqa: E402
from mcrat_amd.host import binding as B  # noqa: E402

n = int(os.environ.get("N", "1000000"))
frame, ph, cfg = synth.config2(n_photons=n)
host, rccl = B.host(), B.host_rccl()
rem = float(os.environ.get("REM", "0.02"))           # a tenth of the frame: ~800 passes of the one list
for variant in ("c-loop", "c-graph", "c-graph+rccl", "c-loop+device", "c-graph+device"):
    stream = torch.cuda.Stream()
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream.cuda_stream)
    e.set_hydro(frame)
    e.set_photons(ph)
    t, st = C.c_double(0.0), engine.FrameStats()
    comm = C.c_void_p()
    if variant.endswith("+device"):                      # device-initiated exchange (one rank: with itself): a push and a wait kernel per round
        recv, _, flags, _ = e.shared_clock_attach_device(1, 0, 0)
        e.shared_clock_set_peers([recv], [flags])
    if variant == "c-graph+rccl":
        rccl.mcrat_host_rccl_comm_single(C.byref(comm))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if variant == "c-loop+device":
        cb = C.cast(host.mcrat_host_exchange_device, B.ALLGATHER)
        rc = host.mcrat_host_shared_clock_frame(e.ctx, 1, 0, 0, cb, e.ctx, C.c_void_p(stream.cuda_stream), C.byref(t), rem, 7, 64, C.byref(st))
    elif variant == "c-loop":
        rc = host.mcrat_host_shared_clock_frame(e.ctx, 1, 0, 0, None, None, C.c_void_p(stream.cuda_stream), C.byref(t), rem, 7, 64, C.byref(st))
    else:
        rc = rccl.mcrat_host_shared_clock_frame_graph(e.ctx, 1, 0, 0, comm, C.c_void_p(stream.cuda_stream), C.byref(t), rem, 7, 64, C.byref(st))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-13s rc %d  passes %d  events %d  %.1f us per pass" % (variant, rc, st.iterations, st.frame_scatt_cnt, dt / max(1, st.iterations) * 1e6), flush=True)
    e.close()
