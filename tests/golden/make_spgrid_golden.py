#!/usr/bin/env python3
"""Regenerates tests/golden/spgrid_index_{float,double}.json by running the REAL reference SPGrid code
(oracle/_ref/spgrid_ref, built by `make -C oracle ref` from /root/reference/Lib/SPGrid/Core).
Only runs where /root/reference exists; the JSON vectors (data, not source) are what is committed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
exe = os.path.join(ROOT, "oracle", "_ref", "spgrid_ref")
for kind in ("float", "double"):
    out = subprocess.check_output([exe, kind, "1200", "1"])
    path = os.path.join(ROOT, "tests", "golden", f"spgrid_index_{kind}.json")
    with open(path, "wb") as f:
        f.write(out)
    print("wrote", path, len(out), "bytes")
    # tie points (base node within a few ulp of a cell face): both builds of the driver — x86-64 baseline (-O2: no FMA instruction exists) and the
    # reference's Release flags (-O3 -march=native -fno-math-errno) on this FMA-capable host
    for build, binary in (("O2", "spgrid_ref"), ("native", "spgrid_ref_native")):
        out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", binary), kind, "tie", "0"])
        path = os.path.join(ROOT, "tests", "golden", f"spgrid_tie_{kind}_{build}.json")
        with open(path, "wb") as f:
            f.write(out)
        print("wrote", path, len(out), "bytes")
