#!/bin/bash
# A/B builds of libdlwp_hip.so with extra hipcc flags for ONE source: tools/ab_build2.sh <tag> <file.hip> [-D...]
# -> dlwp_benchmark_amd/ab/lib_<tag>.so (git-ignored, travels to the GPU box); select with DLWP_HIP_LIB=<path>.
set -e
tag=$1; src=$2; shift; shift
base=${src%.hip}
cd "$(dirname "$0")/../dlwp_benchmark_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../ab build/ab_$tag
extra=""
case $base in token_mlp|linear) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off $extra "$@" -c $src -o build/ab_$tag/$base.o
objs=$(ls build/*.o | grep -v "build/$base.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab/lib_$tag.so build/ab_$tag/$base.o $objs -L/opt/rocm/lib -lhipfft
echo built dlwp_benchmark_amd/ab/lib_$tag.so
