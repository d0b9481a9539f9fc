"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/mcrat_hip.h declares, the header compiles as plain C with the struct layouts the Python binding
assumes, and the product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcrat_hip.h")


@pytest.fixture(scope="module")
def engine():
    from mcrat_amd import build, engine
    build.build()                       # hipcc cross-compiles gfx950 without a GPU
    return engine


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcrat_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(engine):
    lib = engine.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(engine.SYMBOLS), set(names) ^ set(engine.SYMBOLS)
    assert b"gfx950" in lib.mcrat_hip_version()
    assert lib.mcrat_hip_strerror(-2) == b"no usable HIP device"


def test_library_exports_nothing_the_header_does_not_declare(engine):
    """-fvisibility=hidden + the header's visibility push: the product library's dynamic symbols are the C ABI and nothing else
    (the diagnostic entry points exist in the -DMCRAT_DIAG build only)"""
    import shutil
    import subprocess
    from mcrat_amd import build
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("no nm in this image")
    out = subprocess.run([nm, "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    extra = {n for n in exported if not n.startswith(("_init", "_fini"))} - set(declared_symbols())
    assert not extra, sorted(extra)


def test_header_is_plain_c_and_layouts_match_binding(engine, tmp_path):
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "mcrat_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mcrat_hip_photon), sizeof(mcrat_hip_photon_list), sizeof(mcrat_hip_photon_soa),
           sizeof(mcrat_hip_hydro), sizeof(mcrat_hip_frame_stats), sizeof(mcrat_hip_config));
    printf("%zu %zu %zu %zu %zu %zu\n", offsetof(mcrat_hip_photon, p0), offsetof(mcrat_hip_photon, num_scatt),
           offsetof(mcrat_hip_photon, recalc_properties), offsetof(mcrat_hip_photon, weight),
           offsetof(mcrat_hip_photon, nearest_block_index), offsetof(mcrat_hip_photon, total_optical_depth));
    return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = list(map(int, out[:6]))
    assert sizes == [176, C.sizeof(engine.PhotonList), C.sizeof(engine.PhotonSoA), C.sizeof(engine.Hydro),
                     C.sizeof(engine.FrameStats), C.sizeof(engine.Config)]
    assert list(map(int, out[6:])) == [8, 128, 136, 144, 152, 168]      # struct photon offsets, SURVEY.md section 5
    assert engine.PHOTON_DTYPE.itemsize == 176
    for name, off in (("p0", 8), ("num_scatt", 128), ("recalc_properties", 136), ("weight", 144),
                      ("nearest_block_index", 152), ("time_to_scatter", 160), ("total_optical_depth", 168)):
        assert engine.PHOTON_DTYPE.fields[name][1] == off


def test_no_cpu_fallback(engine):
    """without a GPU the context cannot be created and nothing is computed; with one, bad switches are refused."""
    import torch
    lib = engine.load_library()
    ctx = C.c_void_p()
    cfg = engine.Config(engine.ABI_VERSION, engine.TWO, engine.CYLINDRICAL, 0, engine.TAU_DIRECT, 0, 0, None, 0, 0, 0, 0, 0)
    rc = lib.mcrat_hip_init(C.byref(ctx), C.byref(cfg))
    if not torch.cuda.is_available():
        assert rc == -2 and not ctx.value                  # MCRAT_HIP_ENODEV
        with pytest.raises(engine.McratHipError):
            engine.Engine(engine.TWO, engine.CYLINDRICAL)
    else:
        assert rc == 0
        lib.mcrat_hip_destroy(ctx)
    bad = engine.Config(engine.ABI_VERSION + 1, engine.TWO, engine.CYLINDRICAL, 0, engine.TAU_DIRECT, 0, 0, None, 0, 0, 0, 0, 0)
    assert lib.mcrat_hip_init(C.byref(ctx), C.byref(bad)) == -1
    unknown_tau = engine.Config(engine.ABI_VERSION, engine.TWO, engine.CYLINDRICAL, 0, 7, 0, 0, None, 0, 0, 0, 0, 0)   # no such TAU_CALCULATION
    assert lib.mcrat_hip_init(C.byref(ctx), C.byref(unknown_tau)) == -1
    # CYCLOSYNCHROTRON_SWITCH ON is one list per context: refused together with virtual ranks
    cyclo = engine.Config(engine.ABI_VERSION, engine.TWO, engine.CYLINDRICAL, 0, engine.TAU_DIRECT, 1, 0, None, 0, 0, 0, 0, 1000)
    assert lib.mcrat_hip_init(C.byref(ctx), C.byref(cyclo)) == -1
    cyclo2 = engine.Config(engine.ABI_VERSION, engine.TWO, engine.CYLINDRICAL, 0, engine.TAU_DIRECT, 2, 0, None, 0, 0, 0, 0, 0)      # no such value
    assert lib.mcrat_hip_init(C.byref(ctx), C.byref(cyclo2)) == -1
    assert lib.mcrat_hip_init(None, C.byref(cfg)) == -1


def test_product_package_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under mcrat_amd/ or include/ may reference it."""
    for base in ("mcrat_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "oracle_py" not in text and "liboracle" not in text and "mcrat_oracle" not in text, os.path.join(dirpath, f)
    code = "import sys; import mcrat_amd, mcrat_amd.engine, mcrat_amd.synth; assert not any(m.startswith('oracle') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_synthetic_configurations_are_well_formed():
    from mcrat_amd import synth
    for frame, ph, cfg in (synth.config1(n_photons=500, n0=16, n1=16), synth.config2(n_photons=500, nzc=4),
                           synth.config3(n_photons=500, nr=128, nth=64), synth.config_3d_cartesian(n_photons=300, n=(8, 8, 8))):
        m = frame["num_elements"]
        for k in ("r0", "r1", "r0_size", "r1_size", "v0", "v1", "dens_lab", "temp", "gamma"):
            assert frame[k].shape == (m,) and np.isfinite(frame[k]).all(), k
        assert (frame["gamma"] >= 1).all() and (frame["dens_lab"] > 0).all() and (frame["temp"] > 0).all()
        speed = np.sqrt(frame["v0"] ** 2 + frame["v1"] ** 2 + (frame["v2"] ** 2 if "v2" in frame else 0))
        assert (speed < 1).all() and np.allclose(speed, np.sqrt(1 - frame["gamma"] ** -2.0), rtol=1e-9, atol=1e-12)
        n = ph["p0"].size
        assert n == (500 if cfg["dimensions"] != synth.THREE else 300)
        norm = np.sqrt(ph["p1"] ** 2 + ph["p2"] ** 2 + ph["p3"] ** 2)
        assert np.allclose(norm, ph["p0"], rtol=1e-13)
        assert (ph["weight"] == 1).all() and (ph["recalc_properties"] == 1).all() and (ph["type"] == b"i").all()
    # the full-size cfg2 mesh is the 16 384-block, 1 048 576-cell frame BASELINE.json names
    mesh = synth.flash_like_mesh(2.5e8, 64, 128, 64, 1e12 - 1.6e10, (0.0, 5e12), (0.0, 2.5e13), 5.0)
    assert mesh["num_elements"] == 16384 * 64 == 1048576
