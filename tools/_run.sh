cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -v "^  File\|Extension modules" | tail -3
for i in 1 2; do
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('NEW ', round(d['value']), {k: round(v,4) for k,v in s.items() if v>0})"
BASIC_PITCH_AMD_LIB=$GRAFT_REPO_ROOT/tools/ubench/bin/libbase.so timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('BASE', round(d['value']), {k: round(v,4) for k,v in s.items() if v>0})"
done
