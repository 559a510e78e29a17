#!/bin/bash
# Build an EXPERIMENT copy of libdvsraster.so: the composite kernels recompiled with -DDVS_EXPERIMENT (which compiles the timing-only
# ablation knobs DVS_TR_DEBUG / DVS_MM_DEBUG / DVS_A9V_NOHOIST / DVS_BWD_EXTRA_LDS in) plus any extra -D flags:
#   tools/xbuild.sh NAME [-DFOO ...]  -> tools/xlib/lib_NAME.so      (select it with DVS_RASTER_LIB=tools/xlib/lib_NAME.so)
# The release library (make -C divshot_amd/csrc) never carries those knobs.
set -e
cd "$(dirname "$0")/../divshot_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/xlib _obj/x_$NAME
HIPCC=/opt/rocm/bin/hipcc
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -DDVS_EXPERIMENT"
FAST="-ffp-contract=fast -munsafe-fp-atomics -fno-slp-vectorize"
for f in render render_tr render_blocks; do
  $HIPCC $COMMON $FAST "$@" -c $f.hip -o _obj/x_$NAME/$f.o 2>/dev/null &
done
$HIPCC $COMMON -ffp-contract=off "$@" -c preprocess.hip -o _obj/x_$NAME/preprocess.o 2>/dev/null &
wait
OBJS=$(ls _obj/*.o | grep -v -e "/render.o" -e "/render_tr.o" -e "/render_blocks.o" -e "/preprocess.o" | tr '\n' ' ')
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../../tools/xlib/lib_$NAME.so $OBJS _obj/x_$NAME/*.o -ldl
rm -rf _obj/x_$NAME
echo built tools/xlib/lib_$NAME.so
