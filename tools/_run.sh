bash tools/profile_gpu.sh g > gpurun_out/prof_g.log 2>&1
timeout 300 python bench.py > gpurun_out/r01g_bench.json 2> gpurun_out/r01g_bench.err
cut -c1-300 gpurun_out/r01g_bench.json
timeout 300 python bench.py --workload tracks --steps 2 --warmup 1 > gpurun_out/r01g_bench_tracks.json 2>/dev/null
timeout 300 python bench.py --bf16-weights --batch 1024 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r01g_bench_bf16_b1024.json 2>/dev/null
timeout 300 python bench.py --batch 1024 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r01g_bench_b1024.json 2>/dev/null
