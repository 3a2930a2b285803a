#!/bin/bash
# tools/build_variant.sh <tag> [extra hipcc flags]  ->  tools/scratch/lib_<tag>.so  (a library build for tools/gpu_ab.sh)
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/scratch
python -c "import __graft_entry__ as g; g._embed_kernel_sources()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=on -Wno-unused-function "$@" \
    -shared -o tools/scratch/lib_$tag.so vkresample_amd/csrc/fftup.hip && echo tools/scratch/lib_$tag.so
