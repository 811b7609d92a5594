for gs in 1 0; do for so in 1 2; do
python bench.py --gpus 8 --share-gpu --cells 32 --steps 2 --warmup 1 --no-cpu --shard-gs $gs --shard-owner $so 2>/dev/null | grep '^{' | tail -1 > gpurun_out/own_${gs}_${so}.json
python - <<PY
import json
d=json.load(open("gpurun_out/own_${gs}_${so}.json"))
b=[r["data_bytes"]/1e6 for r in d["last_step_by_rank"]]
print("shard_gs ${gs} shard_owner ${so}: it/step %.1f ms/step %.0f data MB per rank min %.0f max %.0f ratio %.2f calls %d %s" % (d["iterations_per_step"], d["ms_per_step"], min(b), max(b), max(b)/min(b), d["last_step_by_rank"][0]["collective_calls"], [round(x) for x in b]))
PY
done; done
