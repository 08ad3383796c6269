#!/bin/bash
# VGPR / spill / LDS metadata of every kernel of one translation unit (cross-compiles, no GPU needed):
#   bash profiles/tools/kernel_regs.sh cherryml_amd/csrc/cb_likelihood.hip [name filter]
set -e
src=$(realpath "$1"); filt=${2:-.}
tmp=$(mktemp -d); cd "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o tu.o -save-temps 2>/dev/null
grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|\.group_segment_fixed_size" *gfx950*.s \
  | paste - - - - - | sed 's/\s\+/ /g' | grep -E "$filt" | while read -r line; do
    name=$(echo "$line" | sed 's/.*\.name: \([^ ]*\).*/\1/' | c++filt | cut -c1-70)
    echo "$name | $(echo "$line" | sed 's/\.name: [^ ]* //')"
  done
rm -rf "$tmp"
