"""Synthetic particle clouds standing in for the reference scenes (none of whose assets is loadable:
SURVEY.md §0, §8d).  Conventions follow the reference's sampling helpers: mass = rho*dx^3/ppc,
volume = dx^3/ppc, F = I, C = 0 (Lib/MPM/MpmInitializationHelper.h:116-135), Lame parameters from (E, nu)
(Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:69-73), scenes near world (5,5,5) with dx = 0.01
(Projects/multigrid/MultigridInit3D.h:2483-2524)."""
import numpy as np

_FACT = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2), 12: (2, 2, 3), 16: (2, 2, 4), 20: (2, 2, 5), 27: (3, 3, 3), 64: (4, 4, 4), 80: (4, 4, 5),
         343: (7, 7, 7)}


def lame(E, nu):
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    return mu, lam


def cube_cloud(n, ppc=8, dx=0.01, corner=(5.0, 5.0, 5.0), E=5e4, nu=0.3, rho=2000.0, dtype=np.float64,
               seed=123, omega=(0.0, 0.0, 2.0), noise=0.1, cells=None):
    """Solid axis-aligned block of `cells` (default n^3) grid cells, `ppc` particles per cell on a jittered
    stratified lattice.  Velocity = rigid spin about the centre + Gaussian noise so that grad v != 0."""
    cells = (n, n, n) if cells is None else tuple(cells)
    fx, fy, fz = _FACT[ppc]
    rng = np.random.default_rng(seed)
    ci, cj, ck = np.meshgrid(np.arange(cells[0]), np.arange(cells[1]), np.arange(cells[2]), indexing="ij")
    cell = np.stack([ci.ravel(), cj.ravel(), ck.ravel()], 1).astype(np.float64)  # (Nc,3)
    si, sj, sk = np.meshgrid(np.arange(fx), np.arange(fy), np.arange(fz), indexing="ij")
    sub = np.stack([si.ravel() / fx, sj.ravel() / fy, sk.ravel() / fz], 1)  # (ppc,3) stratum origin
    ext = np.array([1.0 / fx, 1.0 / fy, 1.0 / fz])
    jit = 0.1 + 0.8 * rng.random((cell.shape[0], ppc, 3))
    X = (cell[:, None, :] + sub[None, :, :] + jit * ext[None, None, :]).reshape(-1, 3) * dx + np.asarray(corner)
    Np = X.shape[0]
    centre = np.asarray(corner) + 0.5 * dx * np.asarray(cells)
    rng2 = np.random.default_rng(seed + 1)
    V = np.cross(np.asarray(omega)[None, :], X - centre[None, :]) + noise * rng2.standard_normal((Np, 3))
    # shuffle so that the caller's particle order is NOT the sorted order (exercises particle_order)
    perm = np.random.default_rng(seed + 2).permutation(Np)
    X, V = X[perm], V[perm]
    mu, lam = lame(E, nu)
    T = dtype
    return dict(X=X.astype(T), V=V.astype(T), mass=np.full(Np, rho * dx ** 3 / ppc, T), vol=np.full(Np, dx ** 3 / ppc, T),
                mu=np.full(Np, mu, T), lam=np.full(Np, lam, T), dx=dx)


def cube_slab(n, x0, x1, ppc=8, dx=0.01, corner=(5.0, 5.0, 5.0), E=5e4, nu=0.3, rho=2000.0, dtype=np.float64, seed=123, omega=(0.0, 0.0, 2.0), noise=0.1):
    """The cell planes x0 <= i < x1 of an n^3-cell block like cube_cloud's, generated plane by plane from per-plane random streams: any
    partition of the planes among processes yields the same body, and nobody has to hold all of it (bench.py --scaling strong).  "index" =
    the particles' positions in the whole body (plane-major): their global ids."""
    fx, fy, fz = _FACT[ppc]
    cj, ck = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    si, sj, sk = np.meshgrid(np.arange(fx), np.arange(fy), np.arange(fz), indexing="ij")
    sub = np.stack([si.ravel() / fx, sj.ravel() / fy, sk.ravel() / fz], 1)
    ext = np.array([1.0 / fx, 1.0 / fy, 1.0 / fz])
    centre = np.asarray(corner) + 0.5 * dx * n
    per_plane = n * n * ppc
    Xs, Vs, ids = [], [], []
    for i in range(x0, x1):
        rng = np.random.default_rng([seed, i])
        cell = np.stack([np.full(n * n, float(i)), cj.ravel().astype(np.float64), ck.ravel().astype(np.float64)], 1)
        jit = 0.1 + 0.8 * rng.random((n * n, ppc, 3))
        X = (cell[:, None, :] + sub[None, :, :] + jit * ext[None, None, :]).reshape(-1, 3) * dx + np.asarray(corner)
        V = np.cross(np.asarray(omega)[None, :], X - centre[None, :]) + noise * rng.standard_normal((per_plane, 3))
        perm = rng.permutation(per_plane)  # the caller's particle order is not the sorted order
        Xs.append(X[perm]), Vs.append(V[perm]), ids.append((i * per_plane + perm).astype(np.int64))
    X, V = (np.concatenate(a) if a else np.zeros((0, 3)) for a in (Xs, Vs))
    Np = X.shape[0]
    mu, lam = lame(E, nu)
    T = dtype
    return dict(X=X.astype(T), V=V.astype(T), mass=np.full(Np, rho * dx ** 3 / ppc, T), vol=np.full(Np, dx ** 3 / ppc, T), mu=np.full(Np, mu, T), lam=np.full(Np, lam, T), dx=dx,
                index=(np.concatenate(ids) if ids else np.zeros(0, np.int64)))


def sticky_floor(corner_y, dx, layers=2):
    """Half space {y <= corner_y + (layers-0.5)*dx}: the bottom `layers` node layers of a cloud whose lowest
    cell starts at corner_y (node layer k sits at corner_y + k*dx ... the kernel reaches one layer below)."""
    origin = np.array([[0.0, corner_y + (layers - 1.5) * dx, 0.0]])
    normal = np.array([[0.0, 1.0, 0.0]])
    return origin, normal


# BASELINE.json configs
CONFIGS = {
    "C1": dict(n=22, ppc=20, E=5e4, nu=0.3, rho=2000.0, dtype=np.float64, levelCnt=1, dt=1.0 / 24),
    "C2": dict(n=63, ppc=8, E=5e4, nu=0.3, rho=2000.0, dtype=np.float64, levelCnt=3, dt=1.0 / 24),
    "C3": dict(n=100, ppc=8, E=1e9, nu=0.3, rho=2000.0, dtype=np.float32, levelCnt=3, dt=1.0 / 24),
    # wheel 777019: VonMisesFixedCorotated(240e6) (Projects/multigrid/MultigridInit3D.h:3313-3331)
    "C4": dict(n=126, ppc=8, E=69e9, nu=0.33, rho=2700.0, dtype=np.float64, levelCnt=4, dt=1.0 / 24, plasticity=1, yield_stress=240e6),
    # "flow / goo": SnowPlasticity(psi 0, theta_c 0.01, theta_s 0.001, min_Jp -2, max_Jp 5) (MultigridInit3D.h:3056-3061)
    "C5": dict(n=200, ppc=8, E=1e5, nu=0.35, rho=2000.0, dtype=np.float32, levelCnt=3, dt=1.0 / 24, plasticity=2, snow=(0.0, 0.01, 0.001, -2.0, 5.0)),
}


def plasticity_kwargs(cfg):
    """hot_config fields of a CONFIGS entry's plastic return mapping (empty for the elastic configurations)."""
    return {k: cfg[k] for k in ("plasticity", "yield_stress", "snow") if k in cfg}
