#!/bin/bash
# rocprofv3 passes for the dominant kernels; run on the GPU box via gpurun:
#   gpurun -- 'bash scripts/profile_gpu.sh contact'
# Writes raw output under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand afterwards.
set -u
TAG=${1:-contact}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
# pass 1: kernel trace + stats (no counters)
rocprofv3 -f csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
# pass 2/3: HBM traffic counters, one pass each (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
rocprofv3 -f csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 -f csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
# pass 4: issue mix
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 -d $OUT/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python $REPO/scripts/summarize_pmc.py $OUT
