#!/bin/bash
# rocprofv3 counter passes over an arbitrary command (tuning aid):
#   gpurun -- 'bash scripts/pmc_cmd.sh <tag> "<counters group 1>" "<counters group 2>" -- <command>'
set -u
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 -f csv --pmc $g -d $OUT/g$i -o g$i -- "$@" > $OUT/g$i.log 2>&1
  i=$((i+1))
done
cd $REPO
python scripts/summarize_pmc.py $OUT > $OUT/summary.txt
for f in $OUT/g*.log; do grep -iE "error|invalid|not found|unsupported" $f | head -3; done
rm -rf $OUT/g[0-9]
cat $OUT/summary.txt | grep -v "^==" | grep -E "${PMC_FILTER:-attention|conv_gemm}" | head -80
