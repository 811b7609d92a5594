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
