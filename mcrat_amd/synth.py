"""Synthetic hydro frames and photon sets for the BASELINE.json configurations.

No FLASH/PLUTO data ships with the reference, so every input is generated from a
seed.  A *frame* is a dict holding the fields of the
reference's ``struct hydro_dataframe`` (Src/mcrat.h:194-244) in the form its
readers leave them (cgs units, velocities in units of c); a *photon set* is a dict
of SoA columns named after ``struct photon`` (Src/mcrat.h:142-171).

The fluid fields follow the reference's analytic outflows
(Src/analytic_outflows.c:70-145 spherical wind, :147-236 Lundman+14 structured
jet) and the photons follow the rules of photonInjection (Src/mclib.c:9-300:
slab selection, n_i ~ V_i Gamma_i T_i^3, Bjorkman-Wood black-body frequencies,
isotropic comoving directions boosted to the lab, uniform in-cell positions) with
a fixed total and equal weights.  These are input generators, not part of the
accelerated path; they are plain numpy.
"""
import numpy as np

# Src/mclib.c:4-5 (verbatim values)
A_RAD = 7.56e-15
C_LIGHT = 2.99792458e10
PL_CONST = 6.6260755e-27
K_B = 1.380658e-16
M_P = 1.6726231e-24
THOM_X_SECT = 6.65246e-25
M_EL = 9.1093879e-28

CARTESIAN, SPHERICAL, CYLINDRICAL, POLAR = 0, 1, 2, 3
TWO, TWO_POINT_FIVE, THREE = 0, 1, 2

PHOTON_F8_FIELDS = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3",
                    "r0", "r1", "r2", "s0", "s1", "s2", "s3", "num_scatt", "weight",
                    "time_to_scatter", "total_optical_depth")


# --------------------------------------------------------------------------- meshes
def _finish_frame(frame):
    """fill r, theta as fillHydroCoordinateToSpherical does (Src/geometry.c:66-106,156-174)."""
    dims, geom = frame["dimensions"], frame["geometry"]
    r0, r1 = frame["r0"], frame["r1"]
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            frame["r"] = np.sqrt(r0 * r0 + r1 * r1)
            frame["theta"] = np.arctan2(r0, r1)
        else:
            frame["r"] = r0.copy()
            frame["theta"] = r1.copy()
    else:
        r2 = frame["r2"]
        if geom == CARTESIAN:
            frame["r"] = np.sqrt(r0 * r0 + r1 * r1 + r2 * r2)
            frame["theta"] = np.arccos(r2 / frame["r"])
        elif geom == SPHERICAL:
            frame["r"] = r0.copy()
            frame["theta"] = r1.copy()
        else:
            frame["r"] = np.sqrt(r0 * r0 + r2 * r2)
            frame["theta"] = np.arccos(r2 / frame["r"])
    frame["num_elements"] = int(r0.size)
    return frame


def uniform_mesh_2d(r0_lo, r0_hi, n0, r1_lo, r1_hi, n1, geometry, domain0, domain1, fps):
    """uniform (r0, r1) cells, r0 fastest."""
    d0 = (r0_hi - r0_lo) / n0
    d1 = (r1_hi - r1_lo) / n1
    c0 = r0_lo + d0 * (np.arange(n0) + 0.5)
    c1 = r1_lo + d1 * (np.arange(n1) + 0.5)
    R0, R1 = np.meshgrid(c0, c1, indexing="xy")
    frame = dict(dimensions=TWO, geometry=geometry,
                 r0=R0.ravel().copy(), r1=R1.ravel().copy(),
                 r0_size=np.full(n0 * n1, d0), r1_size=np.full(n0 * n1, d1),
                 r0_domain=tuple(domain0), r1_domain=tuple(domain1), r2_domain=(0.0, 0.0), fps=float(fps))
    return _finish_frame(frame)


def flash_like_mesh(block_side, nxf, nxc, nzc, z_lo, domain0, domain1, fps, geometry=CYLINDRICAL):
    """Two-level FLASH-like block mesh in (r, z): `nxf` x 2*nzc fine leaf blocks of side
    `block_side` next to the axis and `nxc` x nzc coarse blocks of twice the side beyond;
    every leaf block expands to 8x8 cells with x fastest and cell centres at
    (+-1,3,5,7)/16 of the block size (Src/mclib_flash.c:69,253-264)."""
    off = (np.arange(8) - 3.5) / 8.0
    pieces = []
    # fine blocks, then coarse blocks; row-major in z within each level
    for (nx, nz, side, x_lo) in ((nxf, 2 * nzc, block_side, 0.0),
                                 (nxc, nzc, 2.0 * block_side, nxf * block_side)):
        bx = x_lo + side * (np.arange(nx) + 0.5)
        bz = z_lo + side * (np.arange(nz) + 0.5)
        BX, BZ = np.meshgrid(bx, bz, indexing="xy")          # (nz, nx)
        BX, BZ = BX.ravel(), BZ.ravel()
        cx = BX[:, None, None] + side * off[None, None, :]     # x fastest
        cz = BZ[:, None, None] + side * off[None, :, None]
        cx = np.broadcast_to(cx, (BX.size, 8, 8)).reshape(-1)
        cz = np.broadcast_to(cz, (BX.size, 8, 8)).reshape(-1)
        pieces.append((cx, cz, np.full(cx.size, side / 8.0)))
    r0 = np.concatenate([p[0] for p in pieces])
    r1 = np.concatenate([p[1] for p in pieces])
    sz = np.concatenate([p[2] for p in pieces])
    frame = dict(dimensions=TWO, geometry=geometry, r0=r0, r1=r1, r0_size=sz.copy(), r1_size=sz.copy(),
                 r0_domain=tuple(domain0), r1_domain=tuple(domain1), r2_domain=(0.0, 0.0), fps=float(fps))
    return _finish_frame(frame)


def pluto_spherical_mesh(r_lo, r_hi, nr, th_lo, th_hi, nth, fps):
    """PLUTO-like 2-D spherical (r, theta) grid, log-spaced in r, uniform in theta, r fastest;
    centre = (left+right)/2, size = right-left (Src/mclib_pluto.c:951-971)."""
    edges_r = np.exp(np.linspace(np.log(r_lo), np.log(r_hi), nr + 1))
    edges_t = np.linspace(th_lo, th_hi, nth + 1)
    cr, sr = 0.5 * (edges_r[1:] + edges_r[:-1]), edges_r[1:] - edges_r[:-1]
    ct, st = 0.5 * (edges_t[1:] + edges_t[:-1]), edges_t[1:] - edges_t[:-1]
    R, T = np.meshgrid(cr, ct, indexing="xy")
    SR, ST = np.meshgrid(sr, st, indexing="xy")
    frame = dict(dimensions=TWO, geometry=SPHERICAL, r0=R.ravel().copy(), r1=T.ravel().copy(),
                 r0_size=SR.ravel().copy(), r1_size=ST.ravel().copy(),
                 r0_domain=(r_lo, r_hi), r1_domain=(th_lo, th_hi), r2_domain=(0.0, 0.0), fps=float(fps))
    return _finish_frame(frame)


def uniform_mesh_3d_cartesian(lo, hi, n, fps):
    """uniform 3-D Cartesian cells (x fastest) for the THREE/CARTESIAN code path."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    d = (hi - lo) / np.asarray(n)
    cs = [lo[a] + d[a] * (np.arange(n[a]) + 0.5) for a in range(3)]
    Z, Y, X = np.meshgrid(cs[2], cs[1], cs[0], indexing="ij")
    m = X.size
    frame = dict(dimensions=THREE, geometry=CARTESIAN, r0=X.ravel().copy(), r1=Y.ravel().copy(), r2=Z.ravel().copy(),
                 r0_size=np.full(m, d[0]), r1_size=np.full(m, d[1]), r2_size=np.full(m, d[2]),
                 r0_domain=(lo[0], hi[0]), r1_domain=(lo[1], hi[1]), r2_domain=(lo[2], hi[2]), fps=float(fps))
    return _finish_frame(frame)


def uniform_mesh_3d(geometry, lo, hi, n, fps, log_axis0=False):
    """uniform 3-D cells in the hydro coordinates of `geometry` (axis 0 fastest): SPHERICAL (r, theta, phi),
    POLAR (r, phi, z) or CARTESIAN (x, y, z); axis 0 may be log-spaced.  The domain equals the mesh extent."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    cs, ss = [], []
    for a in range(3):
        if a == 0 and log_axis0:
            e = np.exp(np.linspace(np.log(lo[0]), np.log(hi[0]), n[0] + 1))
        else:
            e = np.linspace(lo[a], hi[a], n[a] + 1)
        cs.append(0.5 * (e[1:] + e[:-1]))
        ss.append(e[1:] - e[:-1])
    C2, C1, C0 = np.meshgrid(cs[2], cs[1], cs[0], indexing="ij")
    S2, S1, S0 = np.meshgrid(ss[2], ss[1], ss[0], indexing="ij")
    frame = dict(dimensions=THREE, geometry=geometry,
                 r0=C0.ravel().copy(), r1=C1.ravel().copy(), r2=C2.ravel().copy(),
                 r0_size=S0.ravel().copy(), r1_size=S1.ravel().copy(), r2_size=S2.ravel().copy(),
                 r0_domain=(lo[0], hi[0]), r1_domain=(lo[1], hi[1]), r2_domain=(lo[2], hi[2]), fps=float(fps))
    return _finish_frame(frame)


# --------------------------------------------------------------------------- reader inputs
def _raw_fluid(x, y, rng, v3=False):
    """smooth, physical code-unit fluid for reader tests: |v| < 1, positive density and pressure"""
    rr = np.sqrt(x * x + y * y) + 1e-300
    speed = 0.2 + 0.75 / (1.0 + (np.arctan2(x, y) / 0.3) ** 2)
    jitter = 1.0 + 0.01 * rng.standard_normal(x.shape)
    vx, vy = speed * x / rr * jitter, speed * y / rr
    out = dict(vx1=vx, vx2=vy, rho=1e-9 * (1.0 + rng.random(x.shape)) / (1.0 + (rr / rr.mean()) ** 2),
               prs=3e-7 * (1.0 + 0.1 * rng.random(x.shape)) / (1.0 + (rr / rr.mean()) ** 2.5))
    if v3:
        out["vx3"] = 0.1 * rng.standard_normal(x.shape) * np.sqrt(np.maximum(1 - vx * vx - vy * vy, 0)) * 0.5
    return out


def flash_raw_blocks(block_side, nxf, nxc, nzc, z_lo, l_scale=1e9, seed=0, parent_every=5):
    """The datasets of a FLASH checkpoint as readAndDecimate reads them (Src/mclib_flash.c:143-193), in code units
    (cm / l_scale): the leaf blocks are flash_like_mesh's, in its order; after every `parent_every` leaves comes a
    parent block (node type 2, overlapping its neighbours, garbage-free but different data) that the reader must
    skip.  'coordinates' has three doubles per block, 'block size' three (the reader copies it with a stride of two:
    mclib_flash.c:112,135 -- pass bsize as [n,2] accordingly), the variables are [n_blocks, 1, 8, 8] with x fastest."""
    rng = np.random.default_rng(seed)
    rows = []
    for (nx, nz, side, x_lo) in ((nxf, 2 * nzc, block_side, 0.0), (nxc, nzc, 2.0 * block_side, nxf * block_side)):
        bx = x_lo + side * (np.arange(nx) + 0.5)
        bz = z_lo + side * (np.arange(nz) + 0.5)
        BX, BZ = np.meshgrid(bx, bz, indexing="xy")
        for cx, cz in zip(BX.ravel(), BZ.ravel()):
            rows.append((cx, cz, side, 1))
            if parent_every and len([r for r in rows if r[3] == 1]) % parent_every == 0:
                rows.append((cx + side, cz + side, 2.0 * side, 2))
    rows = np.array(rows)
    n = len(rows)
    coords = np.zeros((n, 3)); coords[:, 0], coords[:, 1] = rows[:, 0] / l_scale, rows[:, 1] / l_scale
    bsize = np.zeros((n, 2)); bsize[:, 0] = bsize[:, 1] = rows[:, 2] / l_scale
    off = (np.arange(8) - 3.5) / 8.0
    x = (coords[:, 0, None, None] + bsize[:, 0, None, None] * off[None, None, :]) * l_scale + 0 * off[None, :, None]
    y = (coords[:, 1, None, None] + bsize[:, 1, None, None] * off[None, :, None]) * l_scale + 0 * off[None, None, :]
    f = _raw_fluid(x, y, rng)
    shape = (n, 1, 8, 8)
    return dict(kind="flash", coordinates=coords, block_size=bsize, node_type=rows[:, 3].astype(np.int32),
                velx=f["vx1"].reshape(shape), vely=f["vx2"].reshape(shape), dens=f["rho"].reshape(shape), pres=f["prs"].reshape(shape),
                l_scale=float(l_scale), d_scale=1.0, p_scale=C_LIGHT ** 2)


def pluto_raw_grid(dimensions, geometry, lo, hi, n, l_scale=1e9, seed=0, log_axis0=False):
    """A PLUTO .dbl frame as readPluto holds it after the file reads (Src/mclib_pluto.c:1085-1128): readGridFile's
    centre / width arrays per axis (in code units; angles in radians) and the variable blocks [nz][ny][nx].
    lo / hi / n: per-axis bounds in physical units (cm or rad) and cell counts; axes that are lengths in `geometry` are
    stored divided by l_scale."""
    rng = np.random.default_rng(seed)
    three = dimensions == THREE
    naxes = 3 if three else 2
    length = {CARTESIAN: (1, 1, 1), CYLINDRICAL: (1, 1, 1), SPHERICAL: (1, 0, 0), POLAR: (1, 0, 1)}[geometry]
    cs, ws = [], []
    for a in range(naxes):
        edges = np.exp(np.linspace(np.log(lo[a]), np.log(hi[a]), n[a] + 1)) if (a == 0 and log_axis0) else np.linspace(lo[a], hi[a], n[a] + 1)
        if length[a]:
            edges = edges / l_scale
        cs.append(0.5 * (edges[:-1] + edges[1:]))
        ws.append(edges[1:] - edges[:-1])
    nx, ny, nz = n[0], n[1], (n[2] if three else 1)
    phys = [cs[a] * (l_scale if length[a] else 1.0) for a in range(naxes)]
    if three:
        X3, X2, X1 = np.meshgrid(phys[2], phys[1], phys[0], indexing="ij")
    else:
        X2, X1 = np.meshgrid(phys[1], phys[0], indexing="ij")
        X2, X1 = X2[None], X1[None]
    # an (x, y) pair in the meridional plane for the smooth test fluid
    if geometry == SPHERICAL:
        px, py = X1 * np.sin(X2), X1 * np.cos(X2)
    elif geometry == POLAR:
        px, py = X1, X3
    else:
        px, py = X1, (X3 if (three and geometry == CARTESIAN) else X2)
    f = _raw_fluid(px, py, rng, v3=dimensions != TWO)
    raw = dict(kind="pluto", nx=nx, ny=ny, nz=nz, x1=cs[0], dx1=ws[0], x2=cs[1], dx2=ws[1],
               x3=cs[2] if three else None, dx3=ws[2] if three else None,
               rho=f["rho"], vx1=f["vx1"], vx2=f["vx2"], vx3=f.get("vx3"), prs=f["prs"],
               l_scale=float(l_scale), d_scale=1.0, p_scale=C_LIGHT ** 2)
    return raw


def chombo_raw(dimensions, geometry, lo, hi, n0, levels=3, box=8, refine_below=(0.5, 0.25), l_scale=1e9, seed=0, logr=False,
               var_names=("rho", "vx1", "vx2", "vx3", "prs", "tr1")):
    """A PLUTO-Chombo AMR frame as readPlutoChombo holds it after its HDF5 reads (Src/mclib_pluto.c:60-430): per level
    the "boxes" (lo_i, lo_j, [lo_k], hi_i, hi_j, [hi_k]), "data:offsets=0" and "data:datatype=0" (per box, variable-major,
    x fastest) datasets and the attributes prob_domain, ref_ratio, dx, logr, domBeg1-3, g_x2stretch, g_x3stretch.
    Level 0 tiles n0 cells over [lo, hi] (physical units; lengths are stored / l_scale) with boxes of `box` cells per side;
    level l+1 refines (ratio 2) every level-l box whose centre lies in the lowest refine_below[l] fraction of axis 1's
    range (the jet axis region), so the levels are properly nested.  Axis 0 may be logarithmic (logr, :454-457)."""
    rng = np.random.default_rng(seed)
    three = dimensions == THREE
    nd = 3 if three else 2
    length = {CARTESIAN: (1, 1, 1), CYLINDRICAL: (1, 1, 1), SPHERICAL: (1, 0, 0), POLAR: (1, 0, 1)}[geometry]
    lo_c = [lo[a] / l_scale if length[a] else lo[a] for a in range(nd)]
    hi_c = [hi[a] / l_scale if length[a] else hi[a] for a in range(nd)]
    if logr:
        dx0 = np.log(hi_c[0] / lo_c[0]) / n0[0]
    else:
        dx0 = (hi_c[0] - lo_c[0]) / n0[0]
    stretch = [1.0] + [((hi_c[a] - lo_c[a]) / n0[a]) / dx0 for a in range(1, nd)]
    names = [v for v in var_names if three or dimensions == TWO_POINT_FIVE or v != "vx3"]
    nv = len(names)

    def centres(level, a, idx):
        dx = dx0 / 2 ** level
        if a == 0 and logr:
            return lo_c[0] * 0.5 * (np.exp(dx * (idx + 1)) + np.exp(dx * idx))
        return lo_c[a] + dx * stretch[a] * (idx + 0.5)

    out_levels = []
    # level 0 boxes
    counts = [n0[a] // box for a in range(nd)]
    grids = np.meshgrid(*[np.arange(counts[a]) * box for a in reversed(range(nd))], indexing="ij")
    los = np.stack([g.ravel() for g in reversed(grids)], axis=1)            # (n_boxes, nd), axis 0 fastest
    for level in range(levels):
        n_boxes = len(los)
        boxes = np.concatenate([los, los + box - 1], axis=1).astype(np.int32)
        cells_per_box = box ** nd
        offsets = (np.arange(n_boxes) * cells_per_box * nv).astype(np.int32)
        # cell positions of every box: index arrays with axis 0 fastest
        loc = np.meshgrid(*[np.arange(box) for _ in range(nd)], indexing="ij")           # [k][j][i] order for nd=3
        rel = [loc[nd - 1 - a].ravel() for a in range(nd)]                                # rel[a]: offset along axis a, x fastest
        idx = [los[:, a, None] + rel[a][None, :] for a in range(nd)]                      # (n_boxes, cells_per_box)
        phys = [centres(level, a, idx[a]) * (l_scale if length[a] else 1.0) for a in range(nd)]
        if geometry == SPHERICAL:
            px, py = phys[0] * np.sin(phys[1]), phys[0] * np.cos(phys[1])
        elif geometry == POLAR:
            px, py = phys[0], phys[2]
        else:
            px, py = phys[0], (phys[2] if (three and geometry == CARTESIAN) else phys[1])
        f = _raw_fluid(px, py, rng, v3=True)
        f["tr1"] = np.full_like(px, 1.0 + level)
        data = np.stack([f[v] for v in names], axis=1).reshape(-1)                        # per box: variable-major
        ext = [n0[a] * 2 ** level for a in range(nd)]
        out_levels.append(dict(boxes=boxes, box_offsets=offsets, data=data, prob_domain=[0] * nd + [e - 1 for e in ext],
                               ref_ratio=2, logr=int(bool(logr)), dx=dx0 / 2 ** level, dombeg1=lo_c[0], dombeg2=lo_c[1],
                               dombeg3=lo_c[2] if three else 0.0, g_x2stretch=stretch[1], g_x3stretch=stretch[2] if three else 1.0))
        if level + 1 < levels:
            frac = refine_below[min(level, len(refine_below) - 1)]
            mid1 = (los[:, 1] + 0.5 * box) / ext[1]
            chosen = los[mid1 < frac]
            kids = np.meshgrid(*[np.array([0, box]) for _ in range(nd)], indexing="ij")
            kid = np.stack([kids[nd - 1 - a].ravel() for a in range(nd)], axis=1)         # (2^nd, nd)
            los = (2 * chosen[:, None, :] + kid[None, :, :]).reshape(-1, nd)
    return dict(kind="chombo", levels=out_levels, var_names=names, l_scale=float(l_scale), d_scale=1.0, p_scale=C_LIGHT ** 2)


# --------------------------------------------------------------------------- fluids
def _radial_velocity(frame, vel):
    dims, geom = frame["dimensions"], frame["geometry"]
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            rr = np.sqrt(frame["r0"] ** 2 + frame["r1"] ** 2)
            frame["v0"] = vel * frame["r0"] / rr
            frame["v1"] = vel * frame["r1"] / rr
        else:
            frame["v0"] = vel.copy()
            frame["v1"] = np.zeros_like(vel)
        if dims == TWO_POINT_FIVE:
            frame["v2"] = np.zeros_like(vel)
    else:
        if geom == CARTESIAN:
            rr = np.sqrt(frame["r0"] ** 2 + frame["r1"] ** 2 + frame["r2"] ** 2)
            frame["v0"] = vel * frame["r0"] / rr
            frame["v1"] = vel * frame["r1"] / rr
            frame["v2"] = vel * frame["r2"] / rr
        elif geom == SPHERICAL:
            frame["v0"] = vel.copy()
            frame["v1"] = np.zeros_like(vel)
            frame["v2"] = np.zeros_like(vel)
        else:
            rr = np.sqrt(frame["r0"] ** 2 + frame["r2"] ** 2)
            frame["v0"] = vel * frame["r0"] / rr
            frame["v1"] = np.zeros_like(vel)
            frame["v2"] = vel * frame["r2"] / rr


def spherical_outflow(frame, gamma_infinity=100.0, lumi=1e54, r00=1e8):
    """formulas of sphericalPrep, Src/analytic_outflows.c:70-145."""
    r = frame["r"]
    coast = r >= r00 * gamma_infinity
    gamma = np.where(coast, gamma_infinity, r / r00)
    pres = np.where(coast,
                    (lumi * r00 ** (2.0 / 3.0) * r ** (-8.0 / 3.0)) / (12.0 * np.pi * C_LIGHT * gamma_infinity ** (4.0 / 3.0)),
                    (lumi * r00 ** 2.0) / (12.0 * np.pi * C_LIGHT * r ** 4.0))
    dens = lumi / (4 * np.pi * r ** 2.0 * C_LIGHT ** 3.0 * gamma_infinity * gamma)
    frame.update(gamma=gamma, pres=pres, dens=dens, dens_lab=dens * gamma, temp=(3 * pres / A_RAD) ** 0.25)
    _radial_velocity(frame, np.sqrt(1 - gamma ** -2.0))
    return frame


def structured_fireball(frame, gamma_0=100.0, lumi=3e50, r00=1e8, theta_j=0.1, p=4.0):
    """formulas of structuredFireballPrep (Lundman, Pe'er & Ryde 2014), Src/analytic_outflows.c:147-236,
    with the parameters the manual quotes for its validation run (Doc/mcrat_doc.tex:553)."""
    r, theta = frame["r"], frame["theta"]
    T_0 = (lumi / (4 * np.pi * r00 * r00 * A_RAD * C_LIGHT)) ** 0.25
    eta = gamma_0 / np.sqrt(1 + (theta / theta_j) ** (2 * p))
    eta = np.where(theta >= theta_j * (gamma_0 / 2) ** (1.0 / p), 2.0, eta)
    r_sat = eta * r00
    coast = r >= r_sat
    # below saturation the reference sets gamma = r / r_sat, which is < 1 (NaN velocity) inside r00; no
    # photon is ever injected there ("it shouldn't matter", analytic_outflows.c:181), so keep the frame finite
    gamma = np.where(coast, eta, np.maximum(r / r_sat, 1.0 + 1e-6))
    temp = np.where(coast, T_0 * (r_sat / r) ** (2.0 / 3.0) / eta, T_0)
    vel = np.sqrt(1 - gamma ** -2.0)
    dens = M_P * lumi / (4 * np.pi * M_P * C_LIGHT ** 3 * eta * vel * gamma * r * r)
    frame.update(gamma=gamma, temp=temp, dens=dens, dens_lab=dens * gamma, pres=A_RAD * temp ** 4.0 / 3)
    _radial_velocity(frame, vel)
    return frame


# --------------------------------------------------------------------------- geometry helpers (vectorised)
def hydro_vector_to_cartesian(frame, idx, phi):
    """Src/geometry.c:189-253 for arrays of cell indices and photon azimuths."""
    dims, geom = frame["dimensions"], frame["geometry"]
    v0, v1 = frame["v0"][idx], frame["v1"][idx]
    v2 = frame["v2"][idx] if (dims != TWO and "v2" in frame) else np.zeros_like(v0)
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            return np.stack([v0 * np.cos(phi) - v2 * np.sin(phi), v0 * np.sin(phi) + v2 * np.cos(phi), v1], axis=-1)
        th = frame["r1"][idx]
        return np.stack([v0 * np.sin(th) * np.cos(phi) + v1 * np.cos(th) * np.cos(phi) - v2 * np.sin(phi),
                         v0 * np.sin(th) * np.sin(phi) + v1 * np.cos(th) * np.sin(phi) + v2 * np.cos(phi),
                         v0 * np.cos(th) - v1 * np.sin(th)], axis=-1)
    if geom == CARTESIAN:
        return np.stack([v0, v1, v2], axis=-1)
    if geom == SPHERICAL:
        th, ph = frame["r1"][idx], frame["r2"][idx]
        return np.stack([v0 * np.sin(th) * np.cos(ph) + v1 * np.cos(th) * np.cos(ph) - v2 * np.sin(ph),
                         v0 * np.sin(th) * np.sin(ph) + v1 * np.cos(th) * np.sin(ph) + v2 * np.cos(ph),
                         v0 * np.cos(th) - v1 * np.sin(th)], axis=-1)
    ph = frame["r1"][idx]
    return np.stack([v0 * np.cos(ph) - v1 * np.sin(ph), v0 * np.sin(ph) + v1 * np.cos(ph), v2], axis=-1)


def lorentz_boost(beta_vec, p4):
    """textbook boost of 4-vectors p4[...,4] into the frames moving with beta_vec[...,3]."""
    b2 = np.sum(beta_vec * beta_vec, axis=-1)
    gamma = 1.0 / np.sqrt(1.0 - b2)
    bp = np.sum(beta_vec * p4[..., 1:], axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        coef = np.where(b2 > 0, (gamma - 1.0) * bp / b2, 0.0)
    out = np.empty_like(p4)
    out[..., 0] = gamma * (p4[..., 0] - bp)
    out[..., 1:] = p4[..., 1:] + (coef - gamma * p4[..., 0])[..., None] * beta_vec
    return out


def element_volume(frame, idx):
    """Src/geometry.c:255-296."""
    dims, geom = frame["dimensions"], frame["geometry"]
    r0, s0, r1, s1 = frame["r0"][idx], frame["r0_size"][idx], frame["r1"][idx], frame["r1_size"][idx]
    a, b = r0 + 0.5 * s0, r0 - 0.5 * s0
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            return np.pi * (a * a - b * b) * s1
        return (2.0 * np.pi / 3.0) * (a ** 3 - b ** 3) * (np.cos(r1 - 0.5 * s1) - np.cos(r1 + 0.5 * s1))
    s2 = frame["r2_size"][idx]
    if geom == CARTESIAN:
        return s0 * s1 * s2
    if geom == SPHERICAL:
        return (1.0 / 3.0) * (a ** 3 - b ** 3) * (np.cos(r1 - 0.5 * s1) - np.cos(r1 + 0.5 * s1)) * s2
    return 0.5 * (a * a - b * b) * s1 * s2


def _corner_spherical(frame, sign):
    dims, geom = frame["dimensions"], frame["geometry"]
    c0 = frame["r0"] + sign * 0.5 * frame["r0_size"]
    c1 = frame["r1"] + sign * 0.5 * frame["r1_size"]
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            return np.sqrt(c0 * c0 + c1 * c1), np.arctan2(c0, c1)
        return c0, c1
    c0 = np.abs(frame["r0"]) + sign * 0.5 * frame["r0_size"]
    c1 = np.abs(frame["r1"]) + sign * 0.5 * frame["r1_size"]
    c2 = np.abs(frame["r2"]) + sign * 0.5 * frame["r2_size"]
    if geom == CARTESIAN:
        rr = np.sqrt(c0 * c0 + c1 * c1 + c2 * c2)
        return rr, np.arccos(c2 / rr)
    if geom == SPHERICAL:
        return c0, c1
    rr = np.sqrt(c0 * c0 + c2 * c2)
    return rr, np.arccos(c2 / rr)


# --------------------------------------------------------------------------- photons
def inject_photons(frame, n_photons, r_inj, theta_min, theta_max, seed, weight=1.0):
    """Fixed-total, equal-weight restatement of photonInjection's rules (Src/mclib.c:9-300)."""
    rng = np.random.default_rng(seed)
    dims = frame["dimensions"]
    rmin = r_inj - 0.5 * C_LIGHT / frame["fps"]
    rmax = r_inj + 0.5 * C_LIGHT / frame["fps"]
    r_in, th_in = _corner_spherical(frame, -1.0)
    r_out, th_out = _corner_spherical(frame, +1.0)
    sel = np.nonzero((rmin <= r_out) & (r_in <= rmax) & (th_out >= theta_min) & (th_in <= theta_max))[0]
    if sel.size == 0:
        raise ValueError("no hydro cell intersects the injection slab")
    expect = (4.0 / 3.0) * element_volume(frame, sel) * frame["gamma"][sel] * 20.29 * frame["temp"][sel] ** 3
    counts = rng.multinomial(n_photons, expect / expect.sum())
    cell = np.repeat(sel, counts)                       # photons ordered by cell, like the reference's loops
    n = cell.size
    T = frame["temp"][cell]

    # black-body frequency, Bjorkman & Wood 2001 as in mclib.c:199-213
    u = 1.0 - rng.random((n, 5))                        # (0,1]
    target = (np.pi ** 4 / 90.0) * u[:, 0]
    csum = np.cumsum(1.0 / np.arange(1, 2001, dtype=np.float64) ** 4)
    m = np.minimum(np.searchsorted(csum, target, side="left"), csum.size - 1) + 1.0
    freq = -np.log(u[:, 1] * u[:, 2] * u[:, 3] * u[:, 4]) / m * K_B * T / PL_CONST

    position_phi = rng.random(n) * 2 * np.pi if dims != THREE else np.zeros(n)
    com_phi = rng.random(n) * 2 * np.pi
    com_theta = np.arccos(rng.random(n) * 2 - 1)
    e = PL_CONST * freq / C_LIGHT
    p_comv = np.stack([e, e * np.sin(com_theta) * np.cos(com_phi), e * np.sin(com_theta) * np.sin(com_phi),
                       e * np.cos(com_theta)], axis=-1)
    beta = hydro_vector_to_cartesian(frame, cell, position_phi)
    p_lab = lorentz_boost(-beta, p_comv)
    nrm = np.sqrt(np.sum(p_lab[:, 1:] ** 2, axis=-1))
    p_lab[:, 1:] *= (p_lab[:, 0] / nrm)[:, None]        # zeroNorm, mclib.c:409

    h0 = frame["r0"][cell] + (rng.random(n) - 0.5) * frame["r0_size"][cell]
    h1 = frame["r1"][cell] + (rng.random(n) - 0.5) * frame["r1_size"][cell]
    geom = frame["geometry"]
    if dims in (TWO, TWO_POINT_FIVE):
        if geom in (CARTESIAN, CYLINDRICAL):
            x, y, z = h0 * np.cos(position_phi), h0 * np.sin(position_phi), h1
        else:
            x, y, z = h0 * np.sin(h1) * np.cos(position_phi), h0 * np.sin(h1) * np.sin(position_phi), h0 * np.cos(h1)
    else:
        h2 = frame["r2"][cell] + (rng.random(n) - 0.5) * frame["r2_size"][cell]
        if geom == CARTESIAN:
            x, y, z = h0, h1, h2
        elif geom == SPHERICAL:
            x, y, z = h0 * np.sin(h1) * np.cos(h2), h0 * np.sin(h1) * np.sin(h2), h0 * np.cos(h1)
        else:
            x, y, z = h0 * np.cos(h1), h0 * np.sin(h1), h2

    ph = dict(
        type=np.full(n, b"i", dtype="S1"),
        p0=p_lab[:, 0].copy(), p1=p_lab[:, 1].copy(), p2=p_lab[:, 2].copy(), p3=p_lab[:, 3].copy(),
        comv_p0=p_comv[:, 0].copy(), comv_p1=p_comv[:, 1].copy(), comv_p2=p_comv[:, 2].copy(), comv_p3=p_comv[:, 3].copy(),
        r0=np.ascontiguousarray(x), r1=np.ascontiguousarray(y), r2=np.ascontiguousarray(z),
        s0=np.ones(n), s1=np.zeros(n), s2=np.zeros(n), s3=np.zeros(n),
        num_scatt=np.zeros(n), weight=np.full(n, float(weight)),
        nearest_block_index=np.zeros(n, dtype=np.int32), recalc_properties=np.ones(n, dtype=np.int32),
        time_to_scatter=np.zeros(n), total_optical_depth=np.zeros(n),
    )
    return ph


def photons_to_aos(ph, dtype):
    """pack SoA columns into the 176-byte AoS records of struct photon."""
    n = ph["p0"].size
    aos = np.zeros(n, dtype=dtype)
    for k in dtype.names:
        aos[k] = ph[k]
    return aos


def photons_from_aos(aos):
    return {k: np.ascontiguousarray(aos[k]) for k in aos.dtype.names}


def select_slab(frame, keep):
    """keep a subset of cells (what the readers' slab selection leaves, Src/mclib_flash.c:279-326)."""
    out = dict(frame)
    for k, v in frame.items():
        if isinstance(v, np.ndarray) and v.shape == (frame["num_elements"],):
            out[k] = np.ascontiguousarray(v[keep])
    out["num_elements"] = int(np.count_nonzero(keep)) if keep.dtype == bool else int(len(keep))
    return out


# --------------------------------------------------------------------------- BASELINE.json configurations
def config1(n_photons=10_000, seed=0, n0=64, n1=64, fps=5.0, r_inj=1e12, stokes=0):
    """cfg1: analytic spherical wind on a 2-D CARTESIAN uniform mesh, Compton-only, STOKES off.
    The mesh covers the photons' slab r in [r_inj-3c/fps, r_inj+3c/fps], theta in [0,5deg]."""
    th = 5.0 * np.pi / 180
    dr = 3 * C_LIGHT / fps
    x_hi = (r_inj + dr) * np.sin(th)
    z_lo, z_hi = (r_inj - dr) * np.cos(th), r_inj + dr
    frame = uniform_mesh_2d(0.0, x_hi, n0, z_lo, z_hi, n1, CARTESIAN, (0.0, 2.5e13), (0.0, 2.5e13), fps)
    spherical_outflow(frame)
    ph = inject_photons(frame, n_photons, r_inj, 0.0, 3.0 * np.pi / 180, seed)
    cfg = dict(dimensions=TWO, geometry=CARTESIAN, stokes=int(stokes), name="cfg1-spherical-wind-2d-cartesian")
    return frame, ph, cfg


def config2(n_photons=1_000_000, seed=0x4D435261, nzc=64, fps=5.0, r_inj=1e12, block_side=2.5e8, stokes=0, lumi=3e50):
    """cfg2: FLASH-like two-level block mesh, CYLINDRICAL (r,z), Lundman structured jet, Compton+KN.
    nzc=64 gives 16 384 leaf blocks = 1 048 576 cells; smaller nzc scales the mesh down for tests
    (cells grow so that the mesh always covers the same slab)."""
    scale = 64 // nzc
    side = block_side * scale
    nxf, nxc = nzc, 2 * nzc
    H = 2 * nzc * side
    z_lo = r_inj - 0.5 * H
    frame = flash_like_mesh(side, nxf, nxc, nzc, z_lo, (0.0, 5e12), (0.0, 2.5e13), fps)
    structured_fireball(frame, lumi=lumi)
    ph = inject_photons(frame, n_photons, r_inj, 0.0, 3.0 * np.pi / 180, seed)
    cfg = dict(dimensions=TWO, geometry=CYLINDRICAL, stokes=int(stokes), name="cfg2-flash-2d-cylindrical-jet")
    return frame, ph, cfg


def config3(n_photons=10_000_000, seed=0x4D435262, nr=2048, nth=512, fps=5.0, r_inj=1e12, stokes=1, lumi=3e50):
    """cfg3: PLUTO-like 2-D SPHERICAL log-r grid, same jet, STOKES on."""
    frame = pluto_spherical_mesh(1e9, 2.5e13, nr, 0.0, np.pi / 2, nth, fps)
    structured_fireball(frame, lumi=lumi)
    ph = inject_photons(frame, n_photons, r_inj, 0.0, 6.0 * np.pi / 180, seed)
    cfg = dict(dimensions=TWO, geometry=SPHERICAL, stokes=int(stokes), name="cfg3-pluto-2d-spherical-jet")
    return frame, ph, cfg


def add_toroidal_flow(frame, fraction=0.3):
    """give the flow a phi component (v2) at unchanged speed, so that gamma stays consistent: exercises the v2 terms
    of hydroVectorToCartesian (Src/geometry.c:212-225) in 2.5-D"""
    k = np.sqrt(1.0 - fraction * fraction)
    speed = np.sqrt(frame["v0"] ** 2 + frame["v1"] ** 2)
    frame["v0"] = frame["v0"] * k
    frame["v1"] = frame["v1"] * k
    frame["v2"] = fraction * speed
    return frame


def config_25d(geometry=CYLINDRICAL, n_photons=2000, seed=21, stokes=1, lumi=1e54):
    """TWO_POINT_FIVE: the cfg2 / cfg3 meshes with a three-component velocity (parity tests only)."""
    if geometry == SPHERICAL:
        frame, ph, cfg = config3(n_photons=n_photons, seed=seed, nr=256, nth=128, stokes=stokes, lumi=lumi)
    else:
        frame, ph, cfg = config2(n_photons=n_photons, seed=seed, nzc=8, stokes=stokes, lumi=lumi)
    frame["dimensions"] = TWO_POINT_FIVE
    add_toroidal_flow(frame)
    ph = inject_photons(frame, n_photons, 1e12, 0.0, 3.0 * np.pi / 180, seed)
    cfg = dict(cfg, dimensions=TWO_POINT_FIVE, name="2.5d-" + cfg["name"])
    return frame, ph, cfg


def config_3d(geometry, n_photons=1500, seed=9, fps=5.0, r_inj=1e12, stokes=1):
    """small THREE/SPHERICAL or THREE/POLAR wedge around the jet axis with the spherical wind (parity tests only)."""
    if geometry == SPHERICAL:
        frame = uniform_mesh_3d(SPHERICAL, (0.97e12, 0.002, 0.0), (1.03e12, 0.08, 2 * np.pi), (24, 16, 12), fps, log_axis0=True)
    elif geometry == POLAR:
        frame = uniform_mesh_3d(POLAR, (1e9, 0.0, 0.97e12), (8e10, 2 * np.pi, 1.03e12), (20, 12, 24), fps)
    else:
        raise ValueError("use config_3d_cartesian")
    spherical_outflow(frame)
    ph = inject_photons(frame, n_photons, r_inj, 0.0, 3.0 * np.pi / 180, seed)
    cfg = dict(dimensions=THREE, geometry=geometry, stokes=int(stokes), name="3d-%s-wind" % ("spherical" if geometry == SPHERICAL else "polar"))
    return frame, ph, cfg


def config_3d_cartesian(n_photons=2000, seed=7, n=(24, 24, 24), fps=5.0, r_inj=1e12):
    """small THREE/CARTESIAN case for the 3-D code path (parity tests only)."""
    half = 8e10
    frame = uniform_mesh_3d_cartesian((-half, -half, r_inj - 2e10), (half, half, r_inj + 2e10), n, fps)
    spherical_outflow(frame)
    ph = inject_photons(frame, n_photons, r_inj, 0.0, 3.0 * np.pi / 180, seed)
    cfg = dict(dimensions=THREE, geometry=CARTESIAN, stokes=1, name="3d-cartesian-wind")
    return frame, ph, cfg
