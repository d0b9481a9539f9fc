"""Build libmcrat_hip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m mcrat_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps every a*b+c as two IEEE
roundings so that double results track the reference's plain C ("Numerics").
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcrat_hip.so")
KERNEL_TUS = ["kernels%s_d%d.hip" % (m, d) for m in ("", "_table") for d in (0, 1, 2)]   # kernels.hip per TAU_CALCULATION x DIMENSIONS
SOURCES = KERNEL_TUS + ["launchers.hip", "grid_build.hip", "staging.hip", "inject.hip", "ingest.hip", "hot_table.hip", "functions.hip", "engine.hip"]
HEADERS = ["kernels.hip", "device_types.hpp", "launch.hpp", "physics.hpp", "rng.hpp", "cs_device.hpp", os.path.join("..", "..", "include", "mcrat_hip.h")]
# -amdgpu-prealloc-sgpr-spill-vgprs: the loop kernels sit at 256 VGPRs with hundreds of scalar registers spilled to lanes of vector registers; with the
#   compiler's default (those vector registers chosen after everything else is allocated) single instantiations wrote a wrong Stokes V -- another
#   instantiation after every larger edit (round 3: 3-D spherical; round 4: 3-D polar; without the shadow draws: 3-D spherical again), every time cured
#   by keeping the spilled scalars elsewhere (profiles/r04_s3_corruption_probe.txt).  Reserving the lane-spill registers up front costs nothing
#   measurable (headline 0.493 vs 0.493 ms per frame, cfg5 the same, cfg3 3 %) and tests/test_gpu_instantiations.py is green in all 36 tuples x forms.
# --offload-compress: the device code of ~420 instantiations of the loop kernel is 53 MB; compressed in the bundle 8 MB (the HIP runtime unpacks it
#   when the library is loaded).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-fvisibility=hidden",
         "-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs=1", "--offload-compress"]
OBJDIR = os.path.join(HERE, "_obj")
_KERNEL_DEPS = ["kernels.hip", "device_types.hpp", "launch.hpp", "physics.hpp", "rng.hpp", "cs_device.hpp"]
DEPS = {"launchers.hip": ["launchers.hip", "device_types.hpp", "launch.hpp"],
        "grid_build.hip": ["grid_build.hip", "device_types.hpp", "launch.hpp"],
        "staging.hip": ["staging.hip", "device_types.hpp", "launch.hpp", os.path.join("..", "..", "include", "mcrat_hip.h")],
        "inject.hip": ["inject.hip", "device_types.hpp", "launch.hpp", "physics.hpp", "rng.hpp", "cs_device.hpp"],
        "ingest.hip": ["ingest.hip", "device_types.hpp", "launch.hpp"],
        "hot_table.hip": ["hot_table.hip", "device_types.hpp", "launch.hpp", "physics.hpp", "rng.hpp"],
        "functions.hip": ["functions.hip", "device_types.hpp", "launch.hpp", "physics.hpp", "rng.hpp", os.path.join("..", "..", "include", "mcrat_hip.h")],
        "engine.hip": ["engine.hip", "device_types.hpp", "launch.hpp", "rng.hpp", os.path.join("..", "..", "include", "mcrat_hip.h")]}
for _tu in KERNEL_TUS:
    DEPS[_tu] = _KERNEL_DEPS + [_tu]

def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm is required to build libmcrat_hip.so)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, resource_log=None, extra_flags=(), lib=None, objdir=None):
    """extra_flags / lib / objdir: another build of the same sources beside the product one (tools/: -DMCRAT_DIAG=1)"""
    variant = lib is not None
    lib = lib or LIB
    objdir = objdir or OBJDIR
    if not variant and not force and not stale():
        return lib
    # one hipcc per translation unit, side by side (kernels.hip and kernels_table.hip take minutes each), then the link
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags) + ["-c"]
    if resource_log:
        cflags.append("-Rpass-analysis=kernel-resource-usage")
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        deps = [os.path.join(CSRC, d) for d in DEPS[src]]
        fresh = (not force and not resource_log and os.path.exists(obj) and
                 all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps))
        # stderr to a file, not a pipe: the resource remarks of one kernels TU are megabytes, and a full pipe would stall that TU until its turn below
        errf = None if fresh else open(obj + ".log", "w+")
        procs.append((src, obj, None if fresh else subprocess.Popen([hipcc()] + cflags + [os.path.join(CSRC, src), "-o", obj],
                                                                    stdout=subprocess.DEVNULL, stderr=errf, text=True), errf))
    log, failed = [], False
    for src, obj, p, errf in procs:
        if p is None:
            continue
        p.wait()
        errf.seek(0)
        err = errf.read()
        errf.close()
        os.remove(obj + ".log")
        log.append(err)
        if p.returncode != 0:
            sys.stderr.write(err)
            failed = True
    if resource_log:
        with open(resource_log, "w") as f:
            f.write("".join(log))
    if failed:
        raise RuntimeError("hipcc failed building libmcrat_hip.so")
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + [obj for _, obj, _, _ in procs] + ["-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise RuntimeError("hipcc failed linking libmcrat_hip.so")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, resource_log=os.environ.get("MCRAT_RESOURCE_LOG")))
