"""ctypes binding of the C ABI declared in include/hot_mi355x.h.

`HotLib(path, prefix)` binds one shared library exporting that ABI under the symbol prefix `prefix`
("hot_" for the HIP product library).  `Context` is the host-side mirror of the reference objects that own
this path — `MultigridSimulation<T,3>` (Projects/multigrid/MultigridSimulation.h), its
`ImplicitSolverObjective` (Projects/multigrid/ImplicitSolver.h) and the static `MultigridOperator`
(Projects/multigrid/MultigridPreconditioner.h): method names follow the reference members they replace
(particlesToGrid -> p2g, gridToParticles -> g2p, computeResidual -> residual, ...).

Array arguments are numpy arrays (host) or anything exposing `data_ptr()` (torch tensors, host or HIP
device memory); the library resolves the pointer kind itself.
"""
import ctypes as C
import numpy as np

_DT = {0: np.float32, 1: np.float64}


class hot_config(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("device", C.c_int32), ("dx", C.c_double), ("gravity", C.c_double * 3),
        ("apic_rpic_ratio", C.c_double), ("cfl", C.c_double), ("lsolver", C.c_int32), ("Ainv", C.c_int32),
        ("smoother", C.c_int32), ("coarseSolver", C.c_int32), ("levelCnt", C.c_int32), ("times", C.c_int32),
        ("levelscale", C.c_int32), ("omega", C.c_double), ("topomega", C.c_double), ("cneps", C.c_double),
        ("useCN", C.c_int32), ("project", C.c_int32), ("systemBCProject", C.c_int32), ("linesearch", C.c_int32),
        ("matrixFree", C.c_int32), ("boundaryType", C.c_int32), ("useAdaptiveHessian", C.c_int32),
        ("topDownMGS", C.c_int32), ("max_iterations", C.c_int32), ("plasticity", C.c_int32),
        ("yield_stress", C.c_double), ("snow", C.c_double * 5), ("profile", C.c_int32), ("debug_store", C.c_int32), ("useBaselineMultigrid", C.c_int32), ("gs_chain", C.c_int32), ("gs_sub_block", C.c_int32), ("shard_gs", C.c_int32), ("shard_replicated", C.c_int32), ("ls_energy_only", C.c_int32), ("linear_iteration_cap", C.c_int32), ("shard_owner", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


class hot_collision_object(C.Structure):
    _fields_ = [("shape", C.c_int32), ("type", C.c_int32), ("p0", C.c_double * 3), ("p1", C.c_double * 3), ("friction", C.c_double),
                ("b", C.c_double * 3), ("dbdt", C.c_double * 3), ("R", C.c_double * 9), ("omega", C.c_double * 3), ("s", C.c_double), ("dsdt", C.c_double), ("lsq", C.c_double * 4)]


STICKY, SLIP, SEPARATE = 1, 2, 3
HALFSPACE, SPHERE, BOX, CAPPED_CYLINDER, TORUS, ROTATED_BOX, UNION, DIFFERENCE = 0, 1, 2, 3, 4, 5, 6, 7


class hot_stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("converged", C.c_int32), ("linesearch_trials", C.c_int32),
        ("linear_iterations", C.c_int32), ("vcycles", C.c_int32), ("dropped_pairs", C.c_int32),
        ("num_nodes", C.c_int32), ("num_levels", C.c_int32), ("final_scaled_residual", C.c_double),
        ("energy", C.c_double), ("ms_sort", C.c_double), ("ms_p2g", C.c_double), ("ms_begin", C.c_double),
        ("ms_hessian", C.c_double), ("ms_mg_build", C.c_double), ("ms_solve", C.c_double), ("ms_g2p", C.c_double),
        ("ms_total", C.c_double), ("comm_calls", C.c_int64), ("comm_bytes_index", C.c_int64), ("comm_bytes_data", C.c_int64), ("comm_calls_index", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every entry point include/hot_mi355x.h declares (tests check each is exported)
ABI_SYMBOLS = [
    "default_config", "create", "destroy", "last_error", "sync", "set_particles", "get_particles", "sort",
    "get_counts", "get_indexing", "p2g", "get_grid", "set_bc", "set_sticky_halfspaces", "set_collision_objects", "begin_step", "get_dv",
    "set_dv", "update_state", "get_particle_state", "residual", "project", "cn_tolerance", "build_hessian",
    "matfree_multiply", "build_mg", "get_level", "get_matrix", "get_level_nnzb", "get_prolongation", "spmv", "restrict", "prolong",
    "smooth", "vcycle", "solve", "g2p", "line_search", "should_exit", "recover_solution", "transform_residual", "compute_step", "write_partio", "write_restart", "read_restart", "set_particle_ids", "get_particle_ids", "get_stream", "set_comm", "constitutive_eval", "plasticity_eval", "advance", "calculate_dt", "advance_frame", "profile_reset", "profile_count", "profile_get", "version", "abi_version",
]
ABI_VERSION = 6  # include/hot_mi355x.h HOT_ABI_VERSION: the layout of hot_config / hot_stats this module mirrors


# declared by the header for the HIP product only (device-runtime services a host-memory implementation of the ABI has no use for)
PRODUCT_ONLY_SYMBOLS = ["rccl_unique_id", "rccl_attach", "rccl_selftest", "get_level_inblock_nnzb", "copy_bandwidth"]


class HotError(RuntimeError):
    pass


def _ptr(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


class HotLib:
    def __init__(self, path, prefix="hot_"):
        self.path = str(path)
        self.prefix = prefix
        self.lib = C.CDLL(self.path)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        P = C.POINTER
        sig = {
            "default_config": (None, [P(hot_config)]),
            "create": (C.c_int, [P(hot_config), P(vp)]),
            "destroy": (None, [vp]),
            "last_error": (C.c_char_p, [vp]),
            "sync": (C.c_int, [vp]),
            "set_particles": (C.c_int, [vp, i64] + [vp] * 9),
            "get_particles": (C.c_int, [vp] + [vp] * 7),
            "sort": (C.c_int, [vp]),
            "get_counts": (C.c_int, [vp, P(i64), P(i32), P(i32), P(i32)]),
            "get_indexing": (C.c_int, [vp] + [vp] * 5),
            "p2g": (C.c_int, [vp]),
            "get_grid": (C.c_int, [vp, vp, vp, vp]),
            "set_bc": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp]),
            "set_sticky_halfspaces": (C.c_int, [vp, i32, vp, vp]),
            "set_collision_objects": (C.c_int, [vp, i32, vp]),
            "begin_step": (C.c_int, [vp, dbl]),
            "get_dv": (C.c_int, [vp, vp]),
            "set_dv": (C.c_int, [vp, vp]),
            "update_state": (C.c_int, [vp, vp, P(dbl)]),
            "get_particle_state": (C.c_int, [vp, vp, vp, vp]),
            "residual": (C.c_int, [vp, vp]),
            "project": (C.c_int, [vp, vp]),
            "cn_tolerance": (C.c_int, [vp, vp]),
            "build_hessian": (C.c_int, [vp]),
            "matfree_multiply": (C.c_int, [vp, vp, vp]),
            "build_mg": (C.c_int, [vp]),
            "get_level": (C.c_int, [vp, i32, P(i32), P(i32), vp]),
            "get_matrix": (C.c_int, [vp, i32, vp, vp]),
            "get_level_nnzb": (C.c_int, [vp, i32, P(i64)]),
            "get_prolongation": (C.c_int, [vp, i32, vp, vp]),
            "spmv": (C.c_int, [vp, i32, vp, vp]),
            "restrict": (C.c_int, [vp, i32, vp, vp]),
            "prolong": (C.c_int, [vp, i32, vp, vp]),
            "smooth": (C.c_int, [vp, i32, i32, i32, dbl, vp, vp, vp]),
            "vcycle": (C.c_int, [vp, vp, vp]),
            "solve": (C.c_int, [vp, P(hot_stats)]),
            "g2p": (C.c_int, [vp, dbl, P(i32)]),
            "line_search": (C.c_int, [vp, vp, vp, dbl, P(dbl)]),
            "should_exit": (C.c_int, [vp, vp, P(i32), P(dbl)]),
            "recover_solution": (C.c_int, [vp, vp]),
            "transform_residual": (C.c_int, [vp, vp]),
            "compute_step": (C.c_int, [vp, vp, vp]),
            "write_partio": (C.c_int, [vp, C.c_char_p]),
            "write_restart": (C.c_int, [vp, C.c_char_p]),
            "read_restart": (C.c_int, [vp, C.c_char_p]),
            "set_particle_ids": (C.c_int, [vp, vp]),
            "get_particle_ids": (C.c_int, [vp, vp]),
            "get_stream": (C.c_int, [vp, P(vp)]),
            "set_comm": (C.c_int, [vp, vp]),
            "constitutive_eval": (C.c_int, [vp, i32, vp, vp, vp, i32, vp, vp, vp]),
            "plasticity_eval": (C.c_int, [vp, i32, i32, vp, vp, vp, vp]),
            "advance": (C.c_int, [vp, dbl, P(hot_stats)]),
            "calculate_dt": (C.c_int, [vp, dbl, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double)]),
            "advance_frame": (C.c_int, [vp, dbl, dbl, dbl, P(C.c_int32), P(C.c_int32), P(hot_stats)]),
            "profile_reset": (C.c_int, [vp]),
            "profile_count": (C.c_int, [vp, P(i32)]),
            "profile_get": (C.c_int, [vp, i32, C.c_char_p, P(i64), P(dbl)]),
            "version": (C.c_char_p, []),
            "abi_version": (C.c_int, []),
        }
        self.fn = {}
        missing = []
        for name in ABI_SYMBOLS:
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                missing.append(prefix + name)
                continue
            f.restype, f.argtypes = sig[name]
            self.fn[name] = f
        if missing:
            raise HotError(f"{self.path} does not export: {missing}")
        if self.fn["abi_version"]() != ABI_VERSION:  # structures are passed by pointer and written whole: a mismatch is memory corruption, not a warning
            raise HotError(f"{self.path}: ABI version {self.fn['abi_version']()} where this module mirrors {ABI_VERSION} (include/hot_mi355x.h HOT_ABI_VERSION)")

    def default_config(self, **kw):
        cfg = hot_config()
        self.fn["default_config"](C.byref(cfg))
        for k, v in kw.items():
            if k == "gravity":
                for d in range(3):
                    cfg.gravity[d] = float(v[d])
            elif k == "snow":
                for d in range(5):
                    cfg.snow[d] = float(v[d])
            else:
                if not hasattr(cfg, k):
                    raise KeyError(k)
                setattr(cfg, k, v)
        return cfg

    def version(self):
        return self.fn["version"]().decode()

    def context(self, cfg=None, **kw):
        return Context(self, cfg if cfg is not None else self.default_config(**kw))


class Context:
    """One simulation context (== one MultigridSimulation<T,3> + its objective + MG operator)."""

    def __init__(self, lib, cfg):
        self.lib = lib
        self.cfg = cfg
        self.T = _DT[cfg.dtype]
        h = C.c_void_p()
        rc = lib.fn["create"](C.byref(cfg), C.byref(h))
        if rc != 0 or not h:
            raise HotError(f"create failed rc={rc}")
        self.h = h
        self.Np = 0

    def close(self):
        if self.h:
            self.lib.fn["destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        rc = self.lib.fn[name](self.h, *args)
        if rc != 0:
            msg = self.lib.fn["last_error"](self.h)
            raise HotError(f"{self.lib.prefix}{name} -> {rc}: {msg.decode() if msg else ''}")

    def _real(self, a):
        if a is None or hasattr(a, "data_ptr"):
            return a
        return np.ascontiguousarray(a, dtype=self.T)

    # ---- particles
    def set_particles(self, X, V, mass, vol, mu, lam, C_=None, F=None, Jp=None):
        arrs = [self._real(a) for a in (X, V, mass, C_, F, vol, mu, lam, Jp)]
        self.Np = int(X.shape[0])
        self._keep = arrs
        self._call("set_particles", C.c_int64(self.Np), *[_ptr(a) for a in arrs])

    def set_particle_ids(self, ids):
        """Global particle ids of a sharded run (tie break of the sort key, identity of a migrating particle)."""
        ids = np.ascontiguousarray(ids, np.int32)
        assert ids.shape[0] == self.Np
        self._call("set_particle_ids", _ptr(ids))

    def particle_ids(self):
        self.Np = self.counts()["Np"]
        ids = np.empty(self.Np, np.int32)
        self._call("get_particle_ids", _ptr(ids))
        return ids

    def get_particles(self):
        self.Np = self.counts()["Np"]  # a sharded context's particle set changes as particles migrate between ranks
        n, T = self.Np, self.T
        out = dict(X=np.empty((n, 3), T), V=np.empty((n, 3), T), C=np.empty((n, 9), T), F=np.empty((n, 9), T),
                   mu=np.empty(n, T), lam=np.empty(n, T), Jp=np.empty(n, T))
        self._call("get_particles", *[_ptr(out[k]) for k in ("X", "V", "C", "F", "mu", "lam", "Jp")])
        return out

    # ---- sort / indexing
    def sort(self):
        self._call("sort")

    def counts(self):
        np_, ng, nb, nn = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        self._call("get_counts", C.byref(np_), C.byref(ng), C.byref(nb), C.byref(nn))
        return dict(Np=np_.value, Ng=ng.value, Nb=nb.value, Nn=nn.value)

    def indexing(self):
        c = self.counts()
        out = dict(particle_order=np.empty(c["Np"], np.int32), particle_base_offset=np.empty(c["Np"], np.uint64),
                   particle_group=np.empty((c["Ng"], 2), np.int32), block_offset=np.empty(c["Ng"], np.uint64),
                   blocks=np.empty(c["Nb"], np.uint64))
        self._call("get_indexing", *[_ptr(out[k]) for k in ("particle_order", "particle_base_offset", "particle_group", "block_offset", "blocks")])
        return out

    # ---- grid
    def p2g(self):
        self._call("p2g")

    @property
    def Nn(self):
        return self.counts()["Nn"]

    def grid(self):
        n = self.Nn
        out = dict(id2coord=np.empty((n, 3), np.int32), mass=np.empty(n, self.T), v=np.empty((n, 3), self.T))
        self._call("get_grid", _ptr(out["id2coord"]), _ptr(out["mass"]), _ptr(out["v"]))
        return out

    def set_bc(self, node_id, P, R=None, Rinv=None, slip=None, dv_collide=None):
        node_id = np.ascontiguousarray(node_id, np.int32)
        slip = None if slip is None else np.ascontiguousarray(slip, np.uint8)
        arrs = [self._real(a) for a in (P, R, Rinv)]
        dvc = self._real(dv_collide)
        self._call("set_bc", C.c_int32(len(node_id)), _ptr(node_id), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(slip), _ptr(dvc))

    def set_sticky_halfspaces(self, origin, normal):
        o = np.ascontiguousarray(origin, np.float64).reshape(-1, 3)
        n = np.ascontiguousarray(normal, np.float64).reshape(-1, 3)
        self._call("set_sticky_halfspaces", C.c_int32(len(o)), _ptr(o), _ptr(n))

    def set_collision_objects(self, objects):
        """objects: list of dicts(shape, type, p0, p1, friction=0, b=(0,0,0), dbdt=(0,0,0), R=I (3x3), omega=(0,0,0), s=1, dsdt=0, lsq=(1,0,0,0))
        evaluated per node at begin_step; p0 / p1 are in the object's material space (world x = R s X + b)."""
        flat = []  # a composite (shape UNION / DIFFERENCE with a "members" list of primitive dicts) is followed by its members in the array
        for d in objects:
            if "members" in d:
                flat.append(dict(d, p0=(0, 0, 0), p1=(float(len(d["members"])), 0.0, 0.0)))
                flat += [dict(m, type=d["type"]) for m in d["members"]]
            else:
                flat.append(d)
        objects = flat
        arr = (hot_collision_object * max(len(objects), 1))()
        for o, d in zip(arr, objects):
            o.shape, o.type, o.friction = d["shape"], d["type"], d.get("friction", 0.0)
            p1 = d["p1"] if np.ndim(d["p1"]) else (d["p1"], 0.0, 0.0)
            for k in range(3):
                o.p0[k], o.p1[k], o.b[k], o.dbdt[k] = d["p0"][k], p1[k], d.get("b", (0, 0, 0))[k], d.get("dbdt", (0, 0, 0))[k]
                o.omega[k] = d.get("omega", (0, 0, 0))[k]
            R = np.asarray(d.get("R", np.eye(3)), np.float64).reshape(3, 3)
            for k in range(9):
                o.R[k] = R[k % 3, k // 3]  # column-major
            o.s, o.dsdt = d.get("s", 1.0), d.get("dsdt", 0.0)
            for k in range(4):
                o.lsq[k] = d.get("lsq", (1.0, 0.0, 0.0, 0.0))[k]
        self._call("set_collision_objects", C.c_int32(len(objects)), C.cast(arr, C.c_void_p))

    def begin_step(self, dt):
        self._call("begin_step", C.c_double(dt))

    def get_dv(self):
        out = np.empty((self.Nn, 3), self.T)
        self._call("get_dv", _ptr(out))
        return out

    def set_dv(self, dv):
        d = self._real(dv)
        self._call("set_dv", _ptr(d))

    # ---- objective
    def update_state(self, dv=None):
        e = C.c_double()
        d = self._real(dv)
        self._call("update_state", _ptr(d), C.byref(e))
        return e.value

    def particle_state(self):
        n = self.Np
        out = dict(F=np.empty((n, 9), self.T), stress=np.empty((n, 9), self.T), gradV=np.empty((n, 9), self.T))
        self._call("get_particle_state", _ptr(out["F"]), _ptr(out["stress"]), _ptr(out["gradV"]))
        return out

    def residual(self):
        out = np.empty((self.Nn, 3), self.T)
        self._call("residual", _ptr(out))
        return out

    def project(self, v):
        v = np.array(v, dtype=self.T, order="C")
        self._call("project", _ptr(v))
        return v

    def cn_tolerance(self):
        out = np.empty(self.Nn, self.T)
        self._call("cn_tolerance", _ptr(out))
        return out

    def build_hessian(self):
        self._call("build_hessian")

    def matfree_multiply(self, x):
        y = np.empty((self.Nn, 3), self.T)
        xr = self._real(x)
        self._call("matfree_multiply", _ptr(xr), _ptr(y))
        return y

    def build_mg(self):
        self._call("build_mg")

    def level(self, level, coords=True):
        nr, cs = C.c_int32(), C.c_int32()
        self._call("get_level", C.c_int32(level), C.byref(nr), C.byref(cs), None)
        out = dict(nrows=nr.value, colsize=cs.value)
        if coords:
            ic = np.empty((nr.value, 3), np.int32)
            self._call("get_level", C.c_int32(level), C.byref(nr), C.byref(cs), _ptr(ic))
            out["id2coord"] = ic
        return out

    def matrix(self, level):
        info = self.level(level, coords=False)
        col = np.empty((info["nrows"], info["colsize"]), np.int32)
        val = np.empty((info["nrows"], info["colsize"], 9), self.T)
        self._call("get_matrix", C.c_int32(level), _ptr(col), _ptr(val))
        return col, val

    def level_nnzb(self, level):
        v = C.c_int64()
        self._call("get_level_nnzb", C.c_int32(level), C.byref(v))
        return v.value

    def level_inblock_nnzb(self, level):
        """off-diagonal blocks whose column lies in the row's own colour block (HIP product only)"""
        f = getattr(self.lib.lib, self.lib.prefix + "get_level_inblock_nnzb")
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        v = C.c_int64()
        rc = f(self.h, C.c_int32(level), C.byref(v))
        if rc != 0:
            raise HotError(f"get_level_inblock_nnzb -> {rc}")
        return v.value

    def copy_bandwidth(self, nbytes=1 << 30, reps=20):
        """GB/s (read + written) of the library's own 16-byte-per-lane copy kernel on the context's stream (HIP product only)"""
        f = getattr(self.lib.lib, self.lib.prefix + "copy_bandwidth")
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
        v = C.c_double()
        rc = f(self.h, C.c_int64(nbytes), C.c_int32(reps), C.byref(v))
        if rc != 0:
            raise HotError(f"copy_bandwidth -> {rc}")
        return v.value

    def prolongation(self, level):
        n = self.level(level, coords=False)["nrows"]
        col = np.empty((n, 8), np.int32)
        w = np.empty((n, 8), self.T)
        self._call("get_prolongation", C.c_int32(level), _ptr(col), _ptr(w))
        return col, w

    def spmv(self, level, x):
        n = self.level(level, coords=False)["nrows"]
        y = np.empty((n, 3), self.T)
        xr = self._real(x)
        self._call("spmv", C.c_int32(level), _ptr(xr), _ptr(y))
        return y

    def restrict(self, level, fine):
        n = self.level(level + 1, coords=False)["nrows"]
        y = np.empty((n, 3), self.T)
        xr = self._real(fine)
        self._call("restrict", C.c_int32(level), _ptr(xr), _ptr(y))
        return y

    def prolong(self, level, coarse):
        n = self.level(level, coords=False)["nrows"]
        y = np.empty((n, 3), self.T)
        xr = self._real(coarse)
        self._call("prolong", C.c_int32(level), _ptr(xr), _ptr(y))
        return y

    def smooth(self, level, kind, iterations, u, r, tolerance=0.0, initial_residual=None):
        u = np.array(u, dtype=self.T, order="C")
        r = np.array(r, dtype=self.T, order="C")
        r0 = self._real(initial_residual)
        self._call("smooth", C.c_int32(level), C.c_int32(kind), C.c_int32(iterations), C.c_double(tolerance), _ptr(u), _ptr(r), _ptr(r0))
        return u, r

    def vcycle(self, x):
        y = np.empty((self.Nn, 3), self.T)
        xr = self._real(x)
        self._call("vcycle", _ptr(xr), _ptr(y))
        return y

    def solve(self):
        st = hot_stats()
        self._call("solve", C.byref(st))
        return st.as_dict()

    def g2p(self, dt):
        f = C.c_int32()
        self._call("g2p", C.c_double(dt), C.byref(f))
        return f.value

    # ---- the objective concept, member by member (what LBFGS::solve / ExtendedNewtonsMethod::solve call)
    def line_search(self, ddv, alpha=1.0):
        """lineSearch: returns (scaled + transformed ddv, residual at the accepted point, accepted alpha)."""
        d = np.array(ddv, dtype=self.T, order="C")
        r = np.empty((self.Nn, 3), self.T)
        a = C.c_double()
        self._call("line_search", _ptr(d), _ptr(r), C.c_double(alpha), C.byref(a))
        return d, r, a.value

    def should_exit(self, residual):
        e, s = C.c_int32(), C.c_double()
        r = self._real(residual)
        self._call("should_exit", _ptr(r), C.byref(e), C.byref(s))
        return bool(e.value), s.value

    def recover_solution(self, v):
        v = np.array(v, dtype=self.T, order="C")
        self._call("recover_solution", _ptr(v))
        return v

    def transform_residual(self, v):
        v = np.array(v, dtype=self.T, order="C")
        self._call("transform_residual", _ptr(v))
        return v

    def compute_step(self, residual):
        r = self._real(residual)
        st = np.empty((self.Nn, 3), self.T)
        self._call("compute_step", _ptr(r), _ptr(st))
        return st

    # ---- frame output
    def write_partio(self, path):
        self._call("write_partio", str(path).encode())

    def write_restart(self, path):
        self._call("write_restart", str(path).encode())

    def read_restart(self, path):
        self._call("read_restart", str(path).encode())
        self.Np = self.counts()["Np"]

    def rccl_unique_id(self):
        """128-byte ncclUniqueId (rank 0 creates it, every rank passes it to rccl_attach).  HIP library only."""
        buf = C.create_string_buffer(128)
        f = getattr(self.lib.lib, self.lib.prefix + "rccl_unique_id")
        f.restype, f.argtypes = C.c_int, [C.c_void_p]
        rc = f(buf)
        if rc != 0:
            raise HotError(f"rccl_unique_id -> {rc} (RCCL not available)")
        return buf.raw

    def rccl_attach(self, unique_id, rank, size, partition_min_rows=0):
        f = getattr(self.lib.lib, self.lib.prefix + "rccl_attach")
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
        rc = f(self.h, unique_id, rank, size, partition_min_rows)
        if rc != 0:
            msg = self.lib.fn["last_error"](self.h)
            raise HotError(f"rccl_attach -> {rc}: {msg.decode() if msg else ''}")

    def rccl_selftest(self):
        f = getattr(self.lib.lib, self.lib.prefix + "rccl_selftest")
        f.restype, f.argtypes = C.c_int, [C.c_void_p]
        rc = f(self.h)
        if rc != 0:
            raise HotError(f"rccl_selftest -> {rc}")

    def set_comm(self, comm):
        """Install a hot_amd.dist.TorchComm (one connected body over several ranks) or remove it (None).  Before set_particles."""
        self._comm = comm  # keeps the ctypes callbacks alive
        self._call("set_comm", C.byref(comm.struct) if comm is not None else None)

    def constitutive_eval(self, F, mu, lam, project=True, derivative=True):
        """psi (n), P (n,9), dPdF (n,81 or None) of the fixed-corotated model for deformation gradients F (n,9 column-major)."""
        F = np.ascontiguousarray(F, self.T).reshape(-1, 9)
        n = F.shape[0]
        mu = np.ascontiguousarray(np.broadcast_to(mu, (n,)), self.T)
        lam = np.ascontiguousarray(np.broadcast_to(lam, (n,)), self.T)
        psi, P = np.empty(n, self.T), np.empty((n, 9), self.T)
        D = np.empty((n, 81), self.T) if derivative else None
        self._call("constitutive_eval", C.c_int32(n), _ptr(F), _ptr(mu), _ptr(lam), C.c_int32(int(project)), _ptr(psi), _ptr(P), _ptr(D))
        return psi, P, D

    def trial_energy(self, F, mu, lam):
        """psi (n) as the line search's energy-only trials evaluate it (hot_constitutive_eval with project = 2)."""
        F = np.ascontiguousarray(F, self.T).reshape(-1, 9)
        n = F.shape[0]
        mu = np.ascontiguousarray(np.broadcast_to(mu, (n,)), self.T)
        lam = np.ascontiguousarray(np.broadcast_to(lam, (n,)), self.T)
        psi = np.empty(n, self.T)
        self._call("constitutive_eval", C.c_int32(n), _ptr(F), _ptr(mu), _ptr(lam), C.c_int32(2), _ptr(psi), None, None)
        return psi

    def plasticity_eval(self, kind, F, mu, lam, Jp=None):
        """In-place return mapping (1 von Mises with cfg.yield_stress, 2 snow with cfg.snow) on copies: (F, mu, lam, Jp)."""
        F = np.array(F, self.T, order="C").reshape(-1, 9)
        n = F.shape[0]
        mu = np.array(np.broadcast_to(mu, (n,)), self.T)
        lam = np.array(np.broadcast_to(lam, (n,)), self.T)
        Jp = np.ones(n, self.T) if Jp is None else np.array(Jp, self.T)
        self._call("plasticity_eval", C.c_int32(kind), C.c_int32(n), _ptr(F), _ptr(mu), _ptr(lam), _ptr(Jp))
        return F, mu, lam, Jp

    def advance(self, dt):
        st = hot_stats()
        self._call("advance", C.c_double(dt), C.byref(st))
        return st.as_dict()

    def calculate_dt(self, max_dt=1.0 / 24):
        """CFL step (MpmSimulationBase::calculateDt): dict(dt, max_speed, min_corner, max_corner)."""
        dt, ms = C.c_double(), C.c_double()
        lo, hi = (C.c_double * 3)(), (C.c_double * 3)()
        self._call("calculate_dt", C.c_double(max_dt), C.byref(dt), C.byref(ms), lo, hi)
        return dict(dt=dt.value, max_speed=ms.value, min_corner=np.array(lo[:]), max_corner=np.array(hi[:]))

    def advance_frame(self, frame_dt=1.0 / 24, min_dt=1e-6, max_dt=None):
        """One frame of CFL-limited substeps (SimulationBase::advanceOneFrame): (substeps, total iterations, last stats)."""
        n, its, st = C.c_int32(), C.c_int32(), hot_stats()
        self._call("advance_frame", C.c_double(frame_dt), C.c_double(min_dt), C.c_double(frame_dt if max_dt is None else max_dt), C.byref(n), C.byref(its), C.byref(st))
        return n.value, its.value, st.as_dict()

    def sync(self):
        self._call("sync")

    # ---- profiling
    def profile_reset(self):
        self._call("profile_reset")

    def profile(self):
        n = C.c_int32()
        self._call("profile_count", C.byref(n))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            calls, ms = C.c_int64(), C.c_double()
            self._call("profile_get", C.c_int32(i), name, C.byref(calls), C.byref(ms))
            out[name.value.decode()] = dict(calls=calls.value, total_ms=ms.value)
        return out
