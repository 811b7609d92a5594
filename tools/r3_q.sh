for i in 1 2; do
timeout 600 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/iter', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'it/step', d['iterations_per_step'])
t=d['transfers']; print(' xfer', round(t['p2g_ms'],3), round(t['g2p_ms'],3), round(t['mparticles_per_s']), round(t['frac_of_hbm_peak'],3))
"
done
timeout 600 python tools/p2g_time.py C2 2>&1 | grep -v amdgpu.ids | tail -1
