"""Host logic of the multi-GPU path (one process per GPU, torch.distributed; backend "nccl" = RCCL on the
GPU box, "gloo" in CPU tests).

Photons shard by index; every rank owns a disjoint contiguous range, a replica of the hydro frame and its own
clock and RNG stream (rng_stream = rank), exactly like the reference's MPI ranks, which never communicate
inside the photon loop.  There is therefore NO data-path collective: the only exchanges
are the per-frame scalars main() needs -- the photons' r/theta extent that selects the next hydro slab
(phMinMax, Src/mcrat.c:704,721) and the counters it logs (Src/mcrat.c:883-890).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_photons, world_size, rank):
    """contiguous [lo, hi) of photon indices owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(n_photons), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_bounds_even(n_photons, world_size, rank):
    """like shard_bounds, but every boundary is an even slot (the shared-clock mode pairs slots for its free-path
    random numbers, rng.hpp); only the last shard may hold an odd count."""
    pairs = (int(n_photons) + 1) // 2
    lo, hi = shard_bounds(pairs, world_size, rank)
    return min(2 * lo, int(n_photons)), min(2 * hi, int(n_photons))


def shard_photons(ph, world_size, rank, even=False):
    n = len(ph["p0"])
    lo, hi = (shard_bounds_even if even else shard_bounds)(n, world_size, rank)
    return {k: (v[lo:hi].copy() if hasattr(v, "__len__") and len(v) == n else v) for k, v in ph.items()}


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def reduce_minmax(min_r, max_r, min_theta, max_theta):
    """phMinMax over all ranks: the slab of the next frame must hold every rank's photons (mcrat.c:704-721)."""
    lo = torch.tensor([min_r, min_theta], dtype=torch.float64, device=_dev())
    hi = torch.tensor([max_r, max_theta], dtype=torch.float64, device=_dev())
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1])


def reduce_counters(scatterings, photon_steps, relocations, seconds):
    """whole-job totals: counts add up, the wall time of a frame is the slowest rank's."""
    c = torch.tensor([float(scatterings), float(photon_steps), float(relocations)], dtype=torch.float64, device=_dev())
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=_dev())
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(c[0]), float(c[1]), float(c[2]), float(t[0])
