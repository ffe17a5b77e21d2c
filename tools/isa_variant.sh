#!/bin/bash
# Device ISA of a subset of the ADMM kernel variants (development aid):
#   tools/isa_variant.sh <TG_VARIANTS mask> [out.s]   then   python tools/loop_spills.py out.s
cd "$(dirname "$0")/../trajectory_generation_amd/csrc" || exit 1
out=${2:-/tmp/admm_v.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTG_VARIANTS=$1 -S --cuda-device-only -o "$out" admm_kernels.hip 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" "$out" | grep -A4 admm_kernel | paste - - - - - 2>/dev/null | sed 's/_ZN2tg11admm_kernel//' | head -20
