mkdir -p gpurun_out/r03
for c in C1 C3 C4 C5; do
  timeout 900 python bench.py --config $c --no-cpu --steps 2 --warmup 1 > gpurun_out/r03/bench_$c.json 2> gpurun_out/r03/bench_$c.err
  echo "$c rc=$?"; tail -c 600 gpurun_out/r03/bench_$c.err; head -c 1500 gpurun_out/r03/bench_$c.json; echo
done
rocm-smi --showmeminfo vram | tail -4
nproc; free -g | head -2
