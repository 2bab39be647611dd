#!/bin/bash
# usage: scratch/build_variant.sh <name> <file.hip> [-DFLAG ...] : bow_amd/libbowgpu_<name>.so = the library with ONE translation unit
# recompiled with extra flags (A/B of kernel variants: run with BOWGPU_LIB=bow_amd/libbowgpu_<name>.so)
set -e
cd "$(dirname "$0")/../bow_amd/csrc"
NAME=$1; SRC=$2; shift; shift
make -j6 all > /dev/null
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c $SRC -o build/var_${NAME}_$BASE.o
OBJS=$(ls build/*.o | grep -v -e asan_ -e stamps_ -e var_ -e "build/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbowgpu_$NAME.so $OBJS build/var_${NAME}_$BASE.o
echo built bow_amd/libbowgpu_$NAME.so
