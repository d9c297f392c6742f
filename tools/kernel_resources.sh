#!/bin/bash
# Per-kernel register / scratch / LDS usage of the built library (from the code objects' metadata notes).
# Usage: bash tools/kernel_resources.sh [lib] [name-filter]
LIB=${1:-ray-tracing_amd/librt_hip.so}
FILT=${2:-render_kernel}
T=$(mktemp -d); cp "$LIB" "$T/lib.so"; cd "$T"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for f in lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | awk -v filt="$FILT" '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2}
    /\.sgpr_count:/ {s=$2}
    /\.private_segment_fixed_size:/ {p=$2}
    /\.group_segment_fixed_size:/ {g=$2}
    /\.vgpr_spill_count:/ {sp=$2}
    /\.wavefront_size:/ { if (name ~ filt) printf "%-110s vgpr %3s sgpr %3s scratch %4s B spill %3s lds %6s\n", name, v, s, p, sp, g }'
done
rm -rf "$T"
