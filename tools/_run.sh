set -x
cd $GRAFT_REPO_ROOT
BP_CONV1=regw timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "parity or golden or forward" 2>&1 | tail -5
for i in 1 2; do
BP_CONV1=regw timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('REGW', d['value'], d.get('stage_ms'))"
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BASE', d['value'], d.get('stage_ms'))"
done
