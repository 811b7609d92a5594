"""Test-side loader of the CPU oracle (oracle/liboracle.so).  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may touch oracle/ — the product package never does."""
import ctypes as C
import os
import subprocess

import numpy as np

# The oracle is OpenMP code full of short parallel loops: on a many-core GPU host (possibly with a smaller
# cgroup quota than nproc reports) an unbounded team makes it orders of magnitude slower.  Bound the team
# before libgomp is loaded; bench.py's cpu_baseline leg sets its own value and reports it.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, len(os.sched_getaffinity(0))))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_cached = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def load_oracle():
    global _cached
    if _cached is None:
        from hot_amd.binding import HotLib
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build_oracle()
        _cached = HotLib(path, prefix="hoto_")
    return _cached


def raw():
    return load_oracle().lib


import contextlib  # noqa: E402


@contextlib.contextmanager
def wide_sums(on=True):
    """Oracle contexts CREATED AND USED inside this block run the "wide sums" variant (oracle/sim_core.hpp wide_flag): fp32 node sums
    of the particle scatters and all dot products are accumulated in double and rounded once — what the HIP library's fp32 build does
    by construction — instead of in float like the reference.  The flag is process-wide in the oracle: it is read when a context is
    created and consulted while it runs, so do not interleave wide and narrow contexts."""
    old = os.environ.pop("HOT_ORACLE_WIDE", None)
    if on:
        os.environ["HOT_ORACLE_WIDE"] = "1"
    try:
        yield
    finally:
        os.environ.pop("HOT_ORACLE_WIDE", None)
        if old is not None:
            os.environ["HOT_ORACLE_WIDE"] = old
        load_oracle().lib.hoto_set_wide(1 if old is not None else 0)


@contextlib.contextmanager
def psi_invariants(on=True):
    """Oracle contexts created and used inside this block evaluate EVERY strain energy in the product's line-search form (psi from the invariants of
    F^T F, oracle/corotated.hpp corotated_psi_product_form) instead of the reference's mu |F - R|^2 + lambda / 2 (J - 1)^2: the two forms side by side
    on the accept / reject decisions of the line search.  Process-wide like wide_sums()."""
    old = os.environ.pop("HOT_ORACLE_PSI_INVARIANTS", None)
    if on:
        os.environ["HOT_ORACLE_PSI_INVARIANTS"] = "1"
    try:
        yield
    finally:
        os.environ.pop("HOT_ORACLE_PSI_INVARIANTS", None)
        if old is not None:
            os.environ["HOT_ORACLE_PSI_INVARIANTS"] = old
        load_oracle().lib.hoto_set_psi_invariants(1 if old is not None else 0)


def linear_offset(dtype, ijk):
    ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
    out = np.empty(len(ijk), np.uint64)
    raw().hoto_linear_offset(C.c_int(dtype), C.c_int(len(ijk)), C.c_void_p(ijk.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def linear_to_coord(dtype, off):
    off = np.ascontiguousarray(off, np.uint64)
    out = np.empty((len(off), 3), np.int32)
    raw().hoto_linear_to_coord(C.c_int(dtype), C.c_int(len(off)), C.c_void_p(off.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def packed_add(dtype, a, b):
    a = np.ascontiguousarray(a, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    out = np.empty(len(a), np.uint64)
    raw().hoto_packed_add(C.c_int(dtype), C.c_int(len(a)), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def svd3(A):
    A = np.ascontiguousarray(A)
    dtype = 0 if A.dtype == np.float32 else 1
    n = A.shape[0]
    U, S, V = np.empty_like(A), np.empty((n, 3), A.dtype), np.empty_like(A)
    raw().hoto_svd3(C.c_int(dtype), C.c_int(n), C.c_void_p(A.ctypes.data), C.c_void_p(U.ctypes.data), C.c_void_p(S.ctypes.data), C.c_void_p(V.ctypes.data))
    return U, S, V


def corotated(F, mu, lam, project):
    F = np.ascontiguousarray(F, np.float64)
    n = F.shape[0]
    psi, P, dPdF = np.empty(n), np.empty((n, 9)), np.empty((n, 81))
    raw().hoto_corotated(C.c_int(n), C.c_void_p(F.ctypes.data), C.c_double(mu), C.c_double(lam), C.c_int(int(project)),
                         C.c_void_p(psi.ctypes.data), C.c_void_p(P.ctypes.data), C.c_void_p(dPdF.ctypes.data))
    return psi, P, dPdF


def corotated_differential(F, dF, mu, lam, project):
    F = np.ascontiguousarray(F, np.float64)
    dF = np.ascontiguousarray(dF, np.float64)
    n = F.shape[0]
    dP = np.empty((n, 9))
    raw().hoto_corotated_differential(C.c_int(n), C.c_void_p(F.ctypes.data), C.c_void_p(dF.ctypes.data), C.c_double(mu), C.c_double(lam),
                                      C.c_int(int(project)), C.c_void_p(dP.ctypes.data))
    return dP


def make_pd3(S):
    S = np.array(S, np.float64, order="C")
    raw().hoto_make_pd3(C.c_int(S.shape[0]), C.c_void_p(S.ctypes.data))
    return S


def make_pd2(abd):
    abd = np.array(abd, np.float64, order="C")
    raw().hoto_make_pd2(C.c_int(abd.shape[0]), C.c_void_p(abd.ctypes.data))
    return abd


def plasticity(kind, F, mu, lam, Jp, yield_stress=0.0, snow=(10, 2e-2, 7.5e-3, 0.6, 20)):
    F = np.array(F, np.float64, order="C")
    mu = np.array(mu, np.float64)
    lam = np.array(lam, np.float64)
    Jp = np.array(Jp, np.float64)
    snow = np.array(snow, np.float64)
    raw().hoto_plasticity(C.c_int(kind), C.c_int(F.shape[0]), C.c_void_p(F.ctypes.data), C.c_void_p(mu.ctypes.data), C.c_void_p(lam.ctypes.data),
                          C.c_void_p(Jp.ctypes.data), C.c_double(yield_stress), C.c_void_p(snow.ctypes.data))
    return F, mu, lam, Jp
