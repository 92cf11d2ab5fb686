#!/bin/bash
# VGPR / SGPR / spill / scratch figures of every kernel, from the code-object metadata of the
# per-translation-unit objects (secp256k1_voi_amd/build/*.o) that make up the library.
D=${1:-$(dirname $0)/../secp256k1_voi_amd/build}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for o in $D/*.o; do
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/fat.bin 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co 2>/dev/null || continue
  $B/llvm-readelf --notes $T/dev.co | awk -v unit=$(basename $o) '
    /\.name:/ {name=$2}
    /\.private_segment_fixed_size:/ {scr=$2}
    /\.sgpr_count:/ {sg=$2}
    /\.vgpr_count:/ {vg=$2}
    /\.vgpr_spill_count:/ {sp=$2; printf "%-10s %-62s vgpr=%-4s sgpr=%-4s spill=%-4s scratch=%s\n", unit, name, vg, sg, sp, scr}'
done
rm -rf $T
