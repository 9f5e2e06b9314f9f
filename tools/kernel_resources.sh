#!/bin/bash
# usage: kernel_resources.sh [grep-pattern]  -> name scratch sgpr vgpr spill, for every kernel of the library
# (reads the objects of the last build, optable_amd/csrc/build/*.o: one code object per translation unit)
LLVM=/opt/rocm/lib/llvm/bin
HERE=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for o in "$HERE"/optable_amd/csrc/build/*.o; do
    $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$o" $T/fat.bin
    $LLVM/clang-offload-bundler --type=o --input=$T/fat.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
    $LLVM/llvm-readelf --notes $T/dev.co | grep -E "^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.sgpr_count|vgpr_spill|sgpr_spill" | paste - - - - - - | awk '{print $2, $3"="$4, $5"="$6, $7"="$8, $9"="$10, $11"="$12}'
done | sort | grep -E "${1:-.}"
rm -rf $T
