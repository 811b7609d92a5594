set -x
timeout 2400 python bench.py --config C2 --cells 32 --gpus 8 --share-gpu --steps 2 --warmup 1 --no-cpu > gpurun_out/r04_weak_8ranks_one_gpu.json 2> gpurun_out/r04_weak_8ranks_one_gpu.err; tail -3 gpurun_out/r04_weak_8ranks_one_gpu.err; grep "^{" gpurun_out/r04_weak_8ranks_one_gpu.json | head -c 3000
