#!/bin/bash
# A/B builds of libdlwp_hip.so for timing experiments: tools/ab_build.sh <tag> [extra hipcc flags for fno2d.hip ...]
# -> dlwp_benchmark_amd/ab/lib_<tag>.so (git-ignored, travels to the GPU box); select with DLWP_HIP_LIB=<path>.
set -e
tag=$1; shift
cd "$(dirname "$0")/../dlwp_benchmark_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../ab build/ab_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off "$@" -c fno2d.hip -o build/ab_$tag/fno2d.o
objs=$(ls build/*.o | grep -v fno2d.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab/lib_$tag.so build/ab_$tag/fno2d.o $objs -L/opt/rocm/lib -lhipfft
echo built dlwp_benchmark_amd/ab/lib_$tag.so
