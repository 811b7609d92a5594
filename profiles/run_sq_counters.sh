#!/bin/bash
# SQ counter evidence for the kernels furthest below their roofline (Hessian assembly, state update, coloured GS):
#   bash profiles/run_sq_counters.sh <tag> [config]      -> gpurun_out/<tag>/sq_<config>_summary.json (copy into profiles/)
# Two passes of 8 SQ counters each (one rocprofv3 --pmc run per pass, never combined with trace domains other than the kernel trace).
set -u
TAG=${1:-r03}
CFG=${2:-C2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
    i=$((i+1))
    (cd /tmp && timeout 900 rocprofv3 --pmc $P --output-format csv -d "$OUT/sq_${CFG}_$i" -o pmc -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu > "$OUT/sq_${CFG}_$i.json" 2> "$OUT/sq_${CFG}_$i.err")
done
python profiles/summarize_pmc.py "$OUT/sq_${CFG}_summary.json" "$OUT/sq_${CFG}_1" "$OUT/sq_${CFG}_2"
find "$OUT" -name "*counter_collection.csv" -delete
python - <<PY
import json
d = json.load(open("$OUT/sq_${CFG}_summary.json"))
keep = ("k_hessian_rows", "k_dpdf_rec", "k_state", "k_gs_block", "k_gs_offblock", "k_gs_subst", "k_gs_colour", "k_gs_sweep", "k_gs_residual", "k_force_cells", "k_p2g_cells2", "k_p2g_stream", "k_g2p", "k_dpdf45", "k_spmv", "k_apmv_sub", "k_cg_persist")
for k, v in sorted(d.items()):
    if any(s in k for s in keep):
        print(k[:70], {c: round(r["avg"], 1) for c, r in v.items() if isinstance(r, dict)})
PY
