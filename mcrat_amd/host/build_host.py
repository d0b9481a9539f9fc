"""gcc build of the host-side C mirror (libmcrat_hip_host.so), linked against libmcrat_hip.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libmcrat_hip_host.so")
SRC = os.path.join(HERE, "mcrat_hip_host.c")


def build(force=False):
    hip_lib_dir = os.path.dirname(HERE)
    deps = [SRC, os.path.join(HERE, "mcrat_hip_host.h"), os.path.join(ROOT, "include", "mcrat_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-pthread", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", LIB, "-L", hip_lib_dir, "-lmcrat_hip", "-Wl,-rpath,$ORIGIN/.."]
    subprocess.run(cmd, check=True)
    return LIB


H5_LIB = os.path.join(HERE, "libmcrat_hip_host_h5.so")
H5_SRC = os.path.join(HERE, "mcrat_hip_host_h5.c")


def hdf5_prefix():
    """where an HDF5 C library lives (MCRaT itself needs one); None if there is none"""
    for prefix in (os.environ.get("HDF5_DIR"), "/opt/conda", "/usr", "/usr/local"):
        if prefix and os.path.exists(os.path.join(prefix, "include", "hdf5.h")) and any(
                os.path.exists(os.path.join(prefix, d, "libhdf5.so")) for d in ("lib", "lib64", "lib/x86_64-linux-gnu")):
            return prefix
    return None


def build_h5(force=False):
    """printPhotons' HDF5 writer (mcrat_hip_host_h5.c); returns None when no HDF5 C library is installed"""
    prefix = hdf5_prefix()
    if prefix is None:
        return None
    deps = [H5_SRC, os.path.join(HERE, "mcrat_hip_host.h"), os.path.join(ROOT, "include", "mcrat_hip.h")]
    if not force and os.path.exists(H5_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(H5_LIB) for d in deps):
        return H5_LIB
    libdir = next(os.path.join(prefix, d) for d in ("lib", "lib64", "lib/x86_64-linux-gnu") if os.path.exists(os.path.join(prefix, d, "libhdf5.so")))
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(prefix, "include"),
           H5_SRC, "-o", H5_LIB, "-L", os.path.dirname(HERE), "-lmcrat_hip", "-L", libdir, "-lhdf5", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True)
    return H5_LIB


RCCL_LIB = os.path.join(HERE, "libmcrat_hip_host_rccl.so")
RCCL_SRC = os.path.join(HERE, "mcrat_hip_host_rccl.c")


def build_rccl(force=False):
    """the shared-clock exchange over RCCL and its hipGraph (mcrat_hip_host_rccl.c); None where ROCm's RCCL headers are not installed"""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not (os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")) and os.path.exists(os.path.join(rocm, "lib", "librccl.so"))):
        return None
    deps = [RCCL_SRC, os.path.join(HERE, "mcrat_hip_host.h"), os.path.join(ROOT, "include", "mcrat_hip.h")]
    if not force and os.path.exists(RCCL_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(RCCL_LIB) for d in deps):
        return RCCL_LIB
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           RCCL_SRC, "-o", RCCL_LIB, "-L", os.path.dirname(HERE), "-lmcrat_hip", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.run(cmd, check=True)
    return RCCL_LIB


if __name__ == "__main__":
    print(build(force=True))
    print(build_h5(force=True))
    print(build_rccl(force=True))
