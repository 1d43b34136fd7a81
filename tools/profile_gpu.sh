#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_$1/{trace,fetch,write,sq}/ as CSV; summarise with tools/pmc_report.py.
# (--pmc passes are separate runs with no other tracing, as gpurun requires.)
set -u
TAG=${1:-x}
STEPS=${2:-10}     # 10 = the burst regime of the default bench; ~300 = steady state (clocks settled)
WARM=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --sustained-s 0 --no-exact-f32 --no-config-extras --no-fp8-extra"
# the trace pass keeps bench.py's own order — two seconds of sustained steps, then warm-up + timed steps — so that the
# kernels' average durations are those of the regime `value` is measured in
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- ${CMD/--sustained-s 0/--sustained-s 1} > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o bench -- $CMD > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -20
# keep the merge small: drop everything except the per-kernel CSVs
find $OUT -type f ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" -delete
du -sh $OUT
tail -2 $OUT/trace.log
