cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py > gpurun_out/bench_new.json 2> gpurun_out/bench_new.err; tail -3 gpurun_out/bench_new.err
tail -1 gpurun_out/bench_new.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_note'][-40:], {k: round(v,4) for k,v in d['stage_ms'].items() if v>0}, d.get('cqt_stage'))"
