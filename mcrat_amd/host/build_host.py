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
    cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", LIB, "-L", hip_lib_dir, "-lmcrat_hip", "-Wl,-rpath,$ORIGIN/.."]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
