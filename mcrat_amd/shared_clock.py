"""One photon list over several GPUs with ONE clock (SURVEY.md 8e "exact mode"; include/mcrat_hip.h,
mcrat_hip_shared_clock_*).

The reference never couples its ranks (every MPI rank runs Src/mcrat.c:761-851 on its own photons with its own clock);
`sharding.py` is that mode.  Here the event order of ONE list is kept while its slots are spread over the GPUs: per
round every GPU proposes its earliest candidates, the proposals are all-gathered -- the only collective of the data
path, `bytes_per_rank` (736 B) per GPU per round, latency-bound over xGMI -- and every GPU runs the same photonEvent
walk on the merged candidates.  The kernels are in mcrat_amd/csrc/kernels.hip (sc_*); this file is the host loop:
torch supplies the device buffers, the stream and the process group ("nccl" = RCCL on the GPU box).

Every rank must call with the same seed, time_now and remaining_time, and create its Engine with the same rng_stream.
"""
import torch
import torch.distributed as dist

from .engine import Engine


def make_engine(dimensions, geometry, stokes=0, device=0, rng_stream=0, **kw):
    """An Engine whose kernels run on a torch stream, so that torch collectives and copies order with them."""
    dev = torch.device("cuda", device)
    stream = torch.cuda.Stream(device=dev)
    eng = Engine(dimensions, geometry, stokes=stokes, device=device, stream=stream.cuda_stream, rng_stream=rng_stream, **kw)
    eng.torch_stream = stream
    return eng


class SharedClock:
    """Attach an engine (made by make_engine, photons already set) to a group of `world` GPUs as rank `rank` owning
    the global slots [slot_base, slot_base + engine.n)."""

    def __init__(self, engine, world, rank, slot_base, group=None, host_staged=None):
        self.engine = engine
        self.world, self.rank, self.group = int(world), int(rank), group
        self.stream = engine.torch_stream
        dev = self.stream.device
        self.bytes = engine.shared_clock_bytes_per_rank()
        self.send = torch.zeros(self.bytes, dtype=torch.uint8, device=dev)
        self.recv = self.send if self.world == 1 else torch.zeros(self.world * self.bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        engine.shared_clock_attach(self.world, self.rank, slot_base, self.send.data_ptr(), self.recv.data_ptr())
        if host_staged is None:
            host_staged = self.world > 1 and dist.is_initialized() and dist.get_backend(group) != "nccl"
        self.host_staged = bool(host_staged)      # gloo rehearsal: the proposals go through host memory
        self.rounds = 0

    def exchange(self):
        """the all-gather of one round, on the engine's stream"""
        if self.world == 1:
            return
        if self.host_staged:
            mine = self.send.cpu()                                  # synchronises the current (= engine) stream
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            self.recv.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)

    def round(self):
        self.engine.shared_clock_propose()
        self.exchange()
        self.engine.shared_clock_resolve()
        self.rounds += 1

    def propagate_frame(self, time_now, remaining_time, seed, rounds_per_poll=32, max_iterations=0):
        """The loop of mcrat.c:761-851 for the whole list.  Returns (time_now, FrameStats).  With max_iterations the
        state is polled every round and the loop stops once that many iterations are decided."""
        eng = self.engine
        eng.begin_frame(seed, time_now, remaining_time)
        per_poll = 1 if max_iterations else int(rounds_per_poll)
        with torch.cuda.stream(self.stream):
            while True:
                for _ in range(per_poll):
                    self.round()
                done, st = eng.shared_clock_poll()                  # identical on every rank: the loop stays in step
                if done or (max_iterations and st.iterations >= max_iterations):
                    break
        st = eng.shared_clock_finish()
        return st.time_now, st


class LocalGroup:
    """`world` contexts on ONE GPU in ONE process, exchanging by device copies: the whole device side of the
    shared-clock protocol without a process group (tests, and a rehearsal of multi-GPU runs on a one-GPU box)."""

    def __init__(self, dimensions, geometry, stokes, frame, shards, device=0, rng_stream=0, device_exchange=False):
        dev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=dev)
        self.members = []
        self.device_exchange = bool(device_exchange)      # the GPUs exchange by peer writes + a wait kernel (mcrat_hip_shared_clock_exchange)
        if self.device_exchange:
            self._init_device_exchange(dimensions, geometry, stokes, frame, shards, device, rng_stream)
            return
        base = 0
        for r, ph in enumerate(shards):
            eng = Engine(dimensions, geometry, stokes=stokes, device=device, stream=self.stream.cuda_stream, rng_stream=rng_stream)
            eng.torch_stream = self.stream
            eng.set_hydro(frame)
            eng.set_photons(ph)
            if base % 2:
                raise ValueError("every shard but the last must hold an even number of slots")
            self.members.append(SharedClock(eng, len(shards), r, base, host_staged=False))
            base += eng.n
        self.n_total = base

    def _init_device_exchange(self, dimensions, geometry, stokes, frame, shards, device, rng_stream):
        class _Member:                                    # what propagate_frame / get_photons use of a SharedClock
            def __init__(self, engine, world):
                self.engine, self.world = engine, world
        base, bufs = 0, []
        for r, ph in enumerate(shards):
            eng = Engine(dimensions, geometry, stokes=stokes, device=device, stream=self.stream.cuda_stream, rng_stream=rng_stream)
            eng.torch_stream = self.stream
            eng.set_hydro(frame)
            eng.set_photons(ph)
            if base % 2:
                raise ValueError("every shard but the last must hold an even number of slots")
            bufs.append(eng.shared_clock_attach_device(len(shards), r, base))
            self.members.append(_Member(eng, len(shards)))
            base += eng.n
        for m in self.members:                            # one process: the peers' buffers are addressed as they are
            m.engine.shared_clock_set_peers([b[0] for b in bufs], [b[2] for b in bufs])
        self.n_total = base

    def exchange(self):
        if self.device_exchange:                          # every push is queued before any wait: one stream serves all members here
            for m in self.members:
                m.engine.shared_clock_exchange_push()
            for m in self.members:
                m.engine.shared_clock_exchange_wait()
            return
        b = self.members[0].bytes
        for dst in self.members:
            if dst.world == 1:
                continue
            for r, src in enumerate(self.members):
                dst.recv[r * b:(r + 1) * b].copy_(src.send, non_blocking=True)

    def propagate_frame(self, time_now, remaining_time, seed, rounds_per_poll=32, max_iterations=0):
        for m in self.members:
            m.engine.begin_frame(seed, time_now, remaining_time)
        per_poll = 1 if max_iterations else int(rounds_per_poll)
        with torch.cuda.stream(self.stream):
            while True:
                for _ in range(per_poll):
                    for m in self.members:
                        m.engine.shared_clock_propose()
                    self.exchange()
                    for m in self.members:
                        m.engine.shared_clock_resolve()
                polls = [m.engine.shared_clock_poll() for m in self.members]
                done, st = polls[0]
                if done or (max_iterations and st.iterations >= max_iterations):
                    break
        stats = [m.engine.shared_clock_finish() for m in self.members]
        return stats[0].time_now, stats

    def get_photons(self):
        """the shards' photons, concatenated in global slot order"""
        import numpy as np
        parts = [m.engine.get_photons() for m in self.members]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def close(self):
        for m in self.members:
            m.engine.close()
