cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for pl in 1 0; do
echo "PIPELINE=$pl"
BP_PIPELINE=$pl timeout 300 python bench.py --no-cpu-baseline --workload tracks --tracks 1000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' tracks', round(d['value']))"
BP_PIPELINE=$pl timeout 300 python bench.py --no-cpu-baseline --batch 1024 --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' b1024', round(d['value']))"
done
done
