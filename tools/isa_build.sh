#!/bin/bash
# Device-only compile of the launch unit (every frame kernel) to assembly and a summary of one kernel:
#   tools/isa_build.sh <tag> <kernel-substring> [extra hipcc flags]   ->  /tmp/isa/<tag>.s
tag=$1; key=$2; shift 2
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Wno-unused-function --cuda-device-only -S "$@" \
    -o /tmp/isa/$tag.s "$(dirname "$0")/../vkresample_amd/csrc/fftup_launch.hip" || exit 1
awk -v k="$key" '/\.name:/{f = index($0, k) > 0} f && /\.name:|vgpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed/{print}' /tmp/isa/$tag.s
python3 "$(dirname "$0")/isa_hist.py" /tmp/isa/$tag.s "$key" --blocks | grep -v "^  "
