mkdir -p gpurun_out
for d in 0 -30 -60 -200 -1000; do
  echo "=== variant 2 dephase $d"
  BP_CONTOUR_DEPHASE=$d BP_CONTOUR_VARIANT=2 timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['stage_ms']['contour'])
"
done
for d in 0 -30 1000; do
  echo "=== variant 2 dbg 1 dephase $d"
  BP_CONTOUR_DBG=1 BP_CONTOUR_DEPHASE=$d BP_CONTOUR_VARIANT=2 timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['stage_ms']['contour'])
"
done
