"""A/B builds of the loop kernels: libmcrat_hip_<name>.so = the product objects with kernels_d0.hip (DIRECT, 2-D) recompiled under extra flags.

    python tools/variant_build.py name [-DFLAG=...]...      (here, before a GPU call: the libraries travel with the snapshot)
    MCRAT_HIP_LIB=mcrat_amd/libmcrat_hip_<name>.so python bench.py ...        (on the GPU box)

-DMCRAT_DEV_GEOM=2 (cylindrical only: the benchmark's cfg2) is added unless a -DMCRAT_DEV_GEOM is given: a third of the compile time.
Every other translation unit is the product build's object, so the product library must be built first."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcrat_amd import build  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    tus = ["kernels_d0.hip"]
    if "--all-dims" in flags:
        flags.remove("--all-dims")
        tus = ["kernels_d0.hip", "kernels_d1.hip", "kernels_d2.hip"]
    if not any(f.startswith("-DMCRAT_DEV_GEOM") for f in flags):
        flags.append("-DMCRAT_DEV_GEOM=2")
    flags = [f for f in flags if f != "-DMCRAT_DEV_GEOM=all"]
    objdir = os.path.join(build.HERE, "_obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in build.FLAGS if f != "-shared"] + flags + ["-c"]
    log = os.path.join(objdir, "resources.log")
    procs = []
    for tu in tus:
        obj = os.path.join(objdir, os.path.splitext(tu)[0] + ".o")
        errf = open(obj + ".log", "w+")                 # (a file, not a pipe: the remarks of one TU fill a pipe and the TUs would compile one after the other)
        procs.append((obj, subprocess.Popen([build.hipcc()] + cflags + ["-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, tu), "-o", obj],
                                            stdout=subprocess.DEVNULL, stderr=errf, text=True), errf))
    errs = []
    for obj, p, errf in procs:
        p.wait()
        errf.seek(0)
        err = errf.read()
        errf.close()
        errs.append(err)
        if p.returncode != 0:
            sys.stderr.write(err)
            raise SystemExit("hipcc failed")
    open(log, "w").write("".join(errs))
    built = {os.path.basename(o) for o, _, _ in procs}
    objs = [o for o, _, _ in procs]
    for src in build.SOURCES:
        o = os.path.splitext(src)[0] + ".o"
        if o not in built:
            objs.append(os.path.join(build.OBJDIR, o))
    lib = os.path.join(build.HERE, "libmcrat_hip_%s.so" % name)
    r = subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise SystemExit("link failed")
    print(lib)


if __name__ == "__main__":
    main()
