#!/bin/bash
# Kernel resource usage of a compiled object of genpc_amd/lib/obj (VGPRs, spills, scratch, LDS):
#   tools/kmeta.sh nn_f16 [name-filter]
set -e
O=/root/repo/genpc_amd/lib/obj/$1.o
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin "$O" "$T/fat.bin"
tgt=$($L/clang-offload-bundler --list --type=o --input="$T/fat.bin" | grep gfx950)
$L/clang-offload-bundler --type=o --targets="$tgt" --input="$T/fat.bin" --output="$T/dev.co" --unbundle
$L/llvm-readelf --notes "$T/dev.co" | grep -E "^ *\.name:|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size|\.sgpr_count" \
  | awk '/\.name:/{if(n)print n, r; n=$2; r=""} !/\.name:/{r=r" "$1$2} END{print n, r}' | c++filt | grep -i "${2:-.}"
rm -rf "$T"
