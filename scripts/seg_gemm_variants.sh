#!/bin/bash
# A/B of the fp32 GEMM's tile order / epilogue-prefetch switches (and, when coma_amd/_ab/seg_head.so exists, of another build of the library)
# on the layer shapes of the plan.
cd ${GRAFT_REPO_ROOT:-.}
S="python scripts/time_seg_gemm.py $*"
echo "#### current build, defaults";            $S 2>&1 | grep -v amdgpu.ids
echo "#### SEG_XCD_BAND=0";                     SEG_XCD_BAND=0 $S 2>&1 | grep "tile=0"
echo "#### SEG_EPI_PREFETCH=0";                 SEG_EPI_PREFETCH=0 $S 2>&1 | grep "tile=0"
echo "#### both off";                           SEG_XCD_BAND=0 SEG_EPI_PREFETCH=0 $S 2>&1 | grep "tile=0"
if [ -f coma_amd/_ab/seg_head.so ]; then
echo "#### round-start seg_gemm.hip";           COMA_HIP_LIB=coma_amd/_ab/seg_head.so $S 2>&1 | grep "tile=0"
fi
