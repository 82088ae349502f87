#!/bin/bash
# rocprofv3 passes over bench.py; run on the GPU box via gpurun:
#   gpurun -- 'bash scripts/profile_gpu.sh <tag> [bench.py args]'
# Raw output goes under gpurun_out/prof_<tag>/ ; only the small summaries are kept there (kernel_stats.csv,
# pmc_summary.txt) and are then copied into profiles/ with scripts/save_profile.py.
set -u
TAG=${1:-contact}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline $*"
# pass 1: kernel trace + stats (no counters)
# (the adaptive sections run in child processes since r6, which would write their own trace files over this one: left out here;
#  the segmentation plan has its own pass below)
rocprofv3 -f csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD --skip adaptive > $OUT/trace.log 2>&1
grep -E "^\{\"metric" $OUT/trace.log | tail -1 > $OUT/bench_line_under_profiler.json
rm -f $OUT/trace/*kernel_trace.csv
# passes 2-4: counters, one group per pass (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2); never mixed with trace domains
# (counter passes run the launch list eagerly with 2 DDIM steps: rocprofv3 --pmc segfaults under hipGraph replay here)
PCMD="$CMD --eager --skip adaptive --ddim-steps 2 --steps 1 --warmup 0 --contact-steps 3"
rocprofv3 -f csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $PCMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 -f csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $PCMD > $OUT/pmc_write.log 2>&1
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o sq -- $PCMD > $OUT/pmc_sq.log 2>&1
python $REPO/scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
for f in $OUT/pmc_*.log; do tail -3 $f | cut -c1-200; done
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/*.log
# segmentation plan (coma_amd/seg): kernel trace of the captured forward + eager timing, then the counters on the eager launch list
SEG=$REPO/gpurun_out/prof_seg
rm -rf $SEG; mkdir -p $SEG
rocprofv3 -f csv --kernel-trace --stats -d $SEG/trace -o trace -- python $REPO/scripts/time_seg.py 8 4 > $SEG/time_seg.log 2>&1
rm -f $SEG/trace/*kernel_trace.csv
SCMD="python $REPO/scripts/time_seg.py 8 4 --eager-only"
rocprofv3 -f csv --pmc FETCH_SIZE -d $SEG/pmc_fetch -o fetch -- $SCMD > $SEG/pmc_fetch.log 2>&1
rocprofv3 -f csv --pmc WRITE_SIZE -d $SEG/pmc_write -o write -- $SCMD > $SEG/pmc_write.log 2>&1
rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $SEG/pmc_sq -o sq -- $SCMD > $SEG/pmc_sq.log 2>&1
python $REPO/scripts/summarize_pmc.py $SEG > $SEG/pmc_summary.txt
rm -rf $SEG/pmc_fetch $SEG/pmc_write $SEG/pmc_sq $SEG/pmc_*.log
grep -E "captured|eager sum|seg gemm$" $SEG/time_seg.log
head -8 $SEG/trace/trace_kernel_stats.csv | cut -c1-160
grep -E "seg::" $SEG/pmc_summary.txt | head -30
head -12 $OUT/trace/trace_kernel_stats.csv | cut -c1-160
grep -E "sd::|coma::|_ZN2sd" $OUT/pmc_summary.txt | head -60
