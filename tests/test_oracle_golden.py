"""The CPU oracle against the numpy-generated golden vectors (tests/golden/fp_golden.npz): an independent pin of the
oracle's constitutive model, return mappings and APIC P2G (the same vectors pin the HIP library in tests/test_gpu_golden.py)."""
import os

import pytest

from tests import golden_checks as gc
from tests.oracle_lib import load_oracle


@pytest.fixture(scope="module")
def oracle_path():
    return load_oracle().path


@pytest.mark.parametrize("dtype", [1, 0])
def test_oracle_constitutive_against_numpy_golden(oracle_path, dtype):
    print(gc.check_constitutive(oracle_path, "hoto_", dtype))


@pytest.mark.parametrize("dtype", [1, 0])
def test_oracle_plasticity_against_numpy_golden(oracle_path, dtype):
    gc.check_plasticity(oracle_path, "hoto_", dtype)


@pytest.mark.parametrize("dtype", [1, 0])
def test_oracle_p2g_against_numpy_golden(oracle_path, dtype):
    gc.check_p2g(oracle_path, "hoto_", dtype)


def test_generator_is_reproducible(tmp_path):
    """The committed vectors are what the committed generator writes."""
    import subprocess, sys, shutil, numpy as np
    src = os.path.join(gc.ROOT, "tests", "golden", "make_fp_golden.py")
    dst = tmp_path / "make_fp_golden.py"
    shutil.copy(src, dst)
    subprocess.check_call([sys.executable, str(dst)], stdout=subprocess.DEVNULL)
    a, b = np.load(gc.GOLDEN), np.load(tmp_path / "fp_golden.npz")
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=0), k


STEP_TOL = dict(mass=1e-12, v=1e-11, dv0=1e-11, energy=1e-12, residual=1e-11, hessian=1e-12, prolongation=0.0, coarse_matrix=1e-12, gs_u=1e-10, gs_r=1e-10, vcycle=1e-9,
                lbfgs_dv=1e-9, stored_mass=1e-12, stored_v=1e-11, stored_dv0=1e-11, stored_energy=1e-12, stored_residual=1e-11, stored_hessian=1e-12, stored_prolongation=0.0,
                stored_coarse_matrix=1e-12)


def check_step_result(out, scale=1.0):
    t0, t1 = out.pop("linesearch_trials")
    assert t0 == t1, (t0, t1)
    for k, v in out.items():
        assert v <= STEP_TOL[k] * scale, (k, v)


def test_oracle_whole_tiny_step_against_numpy(oracle_path):
    """rows a15 - a23 of SURVEY section 8: assembled Hessian + boundary projection, prolongation, Galerkin coarse matrix, one symmetric coloured
    GS sweep in the reference's order, the two-level V-cycle, two L-BFGS iterations with their line searches — against tests/golden/np_step.py
    (numpy only) in the library's own numbering, and against the stored coordinate-keyed vectors"""
    check_step_result(gc.check_step(oracle_path, "hoto_", 1))


def test_numpy_step_restatement_reproduces_its_stored_vectors():
    gc.check_step_numpy_regression()


def test_step_generator_is_reproducible(tmp_path):
    import subprocess, sys, shutil, numpy as np
    for f in ("make_step_golden.py", "np_step.py", "make_fp_golden.py"):
        shutil.copy(os.path.join(gc.ROOT, "tests", "golden", f), tmp_path / f)
    subprocess.check_call([sys.executable, str(tmp_path / "make_step_golden.py")], stdout=subprocess.DEVNULL)
    a, b = np.load(os.path.join(gc.ROOT, "tests", "golden", "step_golden.npz")), np.load(tmp_path / "step_golden.npz")
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.allclose(a[k], b[k], rtol=1e-11, atol=1e-300), k
