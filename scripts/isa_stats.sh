#!/bin/bash
# Developer script: disassembles the device code of dftpav_amd/csrc/solver.o into /tmp/isa/solver.s and prints, per
# function, its instruction count and the most frequent opcodes (default: two_loop_lane<5>).
#   scripts/isa_stats.sh [symbol-substring] [rows]
sym=${1:-two_loop_laneILi5E}
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $R/dftpav_amd/csrc/solver.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=solver.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d solver.elf > solver.s
start=$(grep -n "^[0-9a-f]* <.*$sym" solver.s | head -1 | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /^[0-9a-f]+ </ {print NR; exit}' solver.s)
echo "lines $start..$end: $((end-start-1)) instructions"
sed -n "$((start+1)),$((end-1))p" solver.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-16}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes solver.elf | grep -E "\.name:|\.vgpr_count|\.vgpr_spill|\.sgpr_spill|private_segment_fixed" | sed 's/  */ /g' | paste - - - - - | grep "Li5E"
