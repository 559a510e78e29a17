#!/bin/bash
# build an experiment copy of libdvsraster.so with extra -D flags on render_tr.hip: tools/xbuild.sh NAME -DFOO [-DBAR ...]  -> tools/xlib/lib_NAME.so
set -e
cd "$(dirname "$0")/../divshot_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/xlib _obj
HIPCC=/opt/rocm/bin/hipcc
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics -fno-slp-vectorize "$@" -c render_tr.hip -o _obj/render_tr_$NAME.o 2>/dev/null
OBJS=$(ls _obj/*.o | grep -v "render_tr" | tr '\n' ' ')
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../../tools/xlib/lib_$NAME.so $OBJS _obj/render_tr_$NAME.o -ldl
rm -f _obj/render_tr_$NAME.o
echo built tools/xlib/lib_$NAME.so
