#!/bin/bash
# usage: kres.sh lib.so [grep-pattern]  -> name scratch sgpr vgpr spill
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$LLVM/clang-offload-bundler --type=o --input=$T/fat.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$LLVM/llvm-readelf --notes $T/dev.co | grep -E "^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.sgpr_count|vgpr_spill" | paste - - - - - | awk '{print $2, "scratch="$4, "sgpr="$6, "vgpr="$8, "spill="$10}' | sort | grep -E "${2:-.}"
rm -rf $T
