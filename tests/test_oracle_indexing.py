"""Pins the oracle's integer path against golden vectors produced by the REAL reference SPGrid code
(tests/golden/spgrid_index_*.json, generator: tests/golden/make_spgrid_golden.py + oracle/spgrid_ref_driver.cpp).
Everything here is bit-exact."""
import json
import os

import numpy as np
import pytest

from tests import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


def load(kind):
    with open(os.path.join(HERE, "golden", f"spgrid_index_{kind}.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("kind,dtype", [("float", 0), ("double", 1)])
def test_mask_against_reference(kind, dtype):
    g = load(kind)
    assert g["struct_bytes"] == (64 if dtype == 0 else 128)
    assert (g["block_xbits"], g["block_ybits"], g["block_zbits"]) == ((2, 2, 2) if dtype == 0 else (1, 2, 2))
    coords = np.array(g["coords"], np.int32)
    off = oracle_lib.linear_offset(dtype, coords)
    assert off.tolist() == g["offsets"]
    assert oracle_lib.linear_to_coord(dtype, off).tolist() == g["roundtrip"]
    rc = np.array(g["rand_coords"], np.int32)
    a = oracle_lib.linear_offset(dtype, rc[:, :3])
    assert a.tolist() == g["rand_offsets"]
    b = oracle_lib.linear_offset(dtype, rc[:, 3:])
    s = oracle_lib.packed_add(dtype, a, b)
    assert s.tolist() == g["rand_packed_add"]
    # Packed_Add is coordinate addition
    assert oracle_lib.linear_to_coord(dtype, s).tolist() == (rc[:, :3] + rc[:, 3:]).tolist()


def test_survey_appendix_a_known_answers():
    # SURVEY.md Appendix A (verified there by compiling Lib/SPGrid/Core)
    assert oracle_lib.linear_offset(0, [[1, 0, 0], [0, 1, 0], [0, 0, 1], [4, 4, 4]]).tolist() == [0x400, 0x100, 0x40, 0x7000]
    assert oracle_lib.linear_offset(1, [[1, 0, 0], [0, 1, 0], [0, 0, 1], [4, 4, 4]]).tolist() == [0x800, 0x200, 0x80, 0xe000]
    assert oracle_lib.linear_offset(0, [[501, 502, 503]])[0] == 0x00000001fffc76c0
    assert oracle_lib.linear_offset(1, [[501, 502, 503]])[0] == 0x00000003fff8ed80


@pytest.mark.parametrize("kind,dtype", [("float", 0), ("double", 1)])
def test_sort_and_numbering_against_reference(oracle, kind, dtype):
    g = load(kind)
    T = np.float32 if dtype == 0 else np.float64
    X = np.array(g["X"], T)
    n = len(X)
    ctx = oracle.context(dtype=dtype, dx=g["dx"])
    one = np.ones(n, T)
    ctx.set_particles(X, np.zeros((n, 3), T), one, one, one, one)
    ctx.sort()
    idx = ctx.indexing()
    assert idx["particle_order"].tolist() == g["particle_order"]
    assert idx["particle_base_offset"].tolist() == g["particle_base_offset"]
    assert idx["particle_group"].tolist() == g["particle_group"]
    assert idx["block_offset"].tolist() == g["block_offset"]
    assert idx["blocks"].tolist() == g["blocks"]
    ctx.p2g()
    grid = ctx.grid()
    assert ctx.Nn == g["num_nodes"]
    assert grid["id2coord"].tolist() == g["id2coord"]
    # mass conservation / partition of unity on the way
    assert abs(grid["mass"].sum() - n) < (1e-3 if dtype == 0 else 1e-9) * n


@pytest.mark.parametrize("kind,dtype", [("float", 0), ("double", 1)])
def test_tie_points_take_the_fma_rounding(oracle, kind, dtype):
    """Cell-face tie points: the oracle's base nodes are the single-rounding (fma) ones, which is what the reference's statement yields under the
    reference's own Release flags (shown by the second build of the golden driver); see tests/golden_checks.check_tie_points."""
    from tests import golden_checks
    assert golden_checks.check_tie_points(oracle, kind, dtype) >= 8
