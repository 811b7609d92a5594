#!/bin/bash
# usage: ab.sh "ENV1=.. ENV2=.." ...   -> one bench line summary per setting
for setting in "$@"; do
  echo "== $setting"
  env $setting timeout 300 python bench.py --no-cpu --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/iter', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'it/step', d['iterations_per_step'], 'build', round(d['hessian_mg_build_ms_per_step'],1))
print(' top', d['kernel_ms_per_step_top'][:10])
r=d['roofline']; print(' roof', r['kernel'], round(r['achieved']), round(r['frac'],3), {k:(v['calls'],round(v['avg_ms'],4)) for k,v in r['per_level'].items()})
t=d['transfers']; print(' xfer', round(t['p2g_ms'],3), round(t['g2p_ms'],3), round(t['mparticles_per_s']), round(t['frac_of_hbm_peak'],3))
"
done
