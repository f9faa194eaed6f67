#!/bin/bash
# A variant build of the library for a same-box A/B: tools/build_variant.sh NAME file.hip "-DMACRO=..." -> dfnet_amd/libvar_NAME.so
# (the named source recompiled with the extra flags, every other object as shipped)
set -e
cd "$(dirname "$0")/../dfnet_amd/csrc"
NAME=$1; SRC=$2; shift 2
mkdir -p build_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c $SRC -o build_abl/${NAME}_${SRC%.hip}.o
OBJS=$(ls build/*.o | grep -v "build/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_$NAME.so $OBJS build_abl/${NAME}_${SRC%.hip}.o
echo "dfnet_amd/libvar_$NAME.so"
