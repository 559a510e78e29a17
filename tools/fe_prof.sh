#!/bin/bash
# kernel trace of one multi-view pass shape with a given build of libdvsraster.so (GPU box):
#   tools/fe_prof.sh LIB [n W H V iters] -> per-(kernel, grid) rows of the k_seg_* / preprocess kernels (rocprofv3 --kernel-trace)
LIB=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/fe_prof_$$
DVS_RASTER_LIB=$PWD/$LIB rocprofv3 --kernel-trace -d $OUT -o p -- python tools/r5_ab.py ${@:-1000000 1920 1080 8 6} > $OUT.log 2>&1
tail -2 $OUT.log | cut -c1-300
python tools/rocpd_summary.py $(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1) --by-grid | grep -v "at::\|rocclr\|render" | cut -c1-140
rm -rf $OUT $OUT.log
