#!/bin/bash
# A variant of the library for A/B runs on the GPU box: bash tools/build_variant.sh <tag> <file.hip> <extra hipcc flags...>
# -> sourmash_amd/libsourmash_amd_<tag>.so (the other objects are the default build's); select with SMG_LIBRARY=<path>.
set -e
cd "$(dirname "$0")/../sourmash_amd/csrc"
TAG=$1; SRC=$2; shift 2
make -s -j8
mkdir -p build/var_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -c $SRC -o build/var_$TAG/${SRC%.hip}.o
OBJS=$(ls build/*.o | grep -v "/${SRC%.hip}.o" | grep -v asan)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsourmash_amd_$TAG.so $OBJS build/var_$TAG/${SRC%.hip}.o -lz
echo built ../libsourmash_amd_$TAG.so
