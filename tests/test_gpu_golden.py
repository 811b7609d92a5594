"""The HIP library against the numpy-generated golden vectors (tests/golden/fp_golden.npz) — no oracle, no shared binding:
the device SVD / fixed-corotated model / PSD-projected dP/dF, the von Mises and snow return mappings and the APIC P2G are
compared with numpy.linalg-based restatements directly through the C ABI."""
import pytest

import hot_amd
from tests import golden_checks as gc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [1, 0])
def test_hip_constitutive_against_numpy_golden(dtype):
    print(gc.check_constitutive(hot_amd.LIB_PATH, "hot_", dtype))


@pytest.mark.parametrize("dtype", [1, 0])
def test_hip_plasticity_against_numpy_golden(dtype):
    gc.check_plasticity(hot_amd.LIB_PATH, "hot_", dtype)


@pytest.mark.parametrize("dtype", [1, 0])
def test_hip_p2g_against_numpy_golden(dtype):
    gc.check_p2g(hot_amd.LIB_PATH, "hot_", dtype)
