"""Kernels with exactly known HBM bytes, to be run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (profiles/run_calibration.sh): the library's
copy kernel (16 bytes per lane, 1 GiB read + 1 GiB written per launch) and k_spmv on level 0 of the C2 hierarchy (8 bytes per lane at a 72-byte stride —
the access pattern of the GS kernels: nnzb x 76 bytes of matrix + column ids, 24 N bytes written, x from the caches).  Prints the known bytes as JSON."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
cfg = dict(synth.CONFIGS["C2"])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
lib = hot_amd.load()
ctx = bench.make_ctx(lib, cloud, cfg)
ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
ctx.update_state(ctx.get_dv())
ctx.build_hessian(), ctx.build_mg()
N, nnzb = ctx.level(0, coords=False)["nrows"], ctx.level_nnzb(0)
x = np.random.default_rng(0).standard_normal((N, 3))
for _ in range(10):
    ctx.spmv(0, x)
gbs = ctx.copy_bandwidth(1 << 30, 10)
print(json.dumps({"k_spmv_L0": {"rows": N, "nnzb": nnzb, "read_bytes": nnzb * 76 + N * 24, "read_bytes_stored_rows": N * 125 * 76, "write_bytes": N * 24, "launches": 10},
                  "k_copy16": {"read_bytes": 1 << 30, "write_bytes": 1 << 30, "launches": 13, "gbytes_per_s": gbs}}))
