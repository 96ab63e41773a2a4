#!/bin/bash
# Register / LDS / scratch use of the kernels in one translation unit's object (build container, no GPU needed):
#   tools/kernel_regs.sh api_mlp [pattern]
# unbundles the gfx950 code object from cliora_amd/csrc/build/<unit>.hip.o and prints the AMDGPU metadata notes per kernel.
U=${1:-api_mlp}; PAT=${2:-.}
O=/root/repo/cliora_amd/csrc/build/$U.hip.o; [ -f "$U" ] && O=$U
T=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
$LL/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O || exit 1
$LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle || exit 1
$LL/llvm-readelf --notes $T/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if not re.search(r'$PAT', name): continue
    import subprocess
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:110]
    ag = blk.split()[0]
    print('vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill_v %3s  %s' % (g('vgpr_count'), ag, g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count'), dem))
"
rm -rf $T
