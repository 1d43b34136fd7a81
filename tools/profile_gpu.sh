#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + two PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_$1/{trace,fetch,write}/ ; summarise with tools/rocprof_summary.py.
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -d $OUT/sq -o bench -- $CMD > $OUT/sq.log 2>&1
ls -R $OUT | head -40
tail -2 $OUT/trace.log
