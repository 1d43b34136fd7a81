cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -1 gpurun_out/final/bench.json | cut -c1-200
bash tools/profile_gpu.sh k > gpurun_out/final/profile.log 2>&1
tail -1 gpurun_out/final/profile.log
timeout 300 python bench.py --no-cpu-baseline --workload tracks --tracks 1000 > gpurun_out/final/bench_tracks.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --no-cpu-baseline --batch 1024 --steps 10 > gpurun_out/final/bench_b1024.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --no-cpu-baseline --batch 1024 --steps 10 --bf16-weights > gpurun_out/final/bench_bf16_b1024.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --no-cpu-baseline --batch 512 --steps 10 --ext-cqt-44k > gpurun_out/final/bench_ext44k_b512.json 2>> gpurun_out/final/bench.err
for f in tracks b1024 bf16_b1024 ext44k_b512; do tail -1 gpurun_out/final/bench_$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']), d['ms_per_step'])"; done
