timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print(round(d['value']), {k: round(v, 4) for k, v in d['stage_ms'].items() if v}, round(d['roofline']['frac'],4))
    else: print(line)
"
BP_BRANCH_PROF=1 timeout 120 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep brprof | grep "wave 1"
