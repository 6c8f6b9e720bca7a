#!/bin/bash
# Register and scratch use of the kernels in a built library: tools/kernel_regs.sh [library] [name pattern]
LIB=${1:-ray-tracer_amd/librt_mi355.so}
PAT=${2:-k_trace_primary}
T=$(mktemp -d)
cp "$LIB" "$T/lib.so"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in "$T"/lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | awk -v pat="$PAT" '
    /\.group_segment_fixed_size:/ {lds=$2} /\.symbol:/ {name=$2} /\.private_segment_fixed_size:/ {priv=$2}
    /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2}
    /\.wavefront_size:/ { if (name ~ pat) printf "%-110s vgpr %3d (spill %d) sgpr %3d (spill %d) scratch %d lds %d\n", name, vg, vs, sg, ss, priv, lds }'
done | sort -u
rm -rf "$T"
