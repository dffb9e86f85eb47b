#!/bin/bash
# Per-kernel register / scratch / code-size summary of the built objects (development aid).
#   tools/kernel_stats.sh [object, library or directory of objects]   (default: ilqr_amd/csrc/obj)
set -e
SRC=${1:-ilqr_amd/csrc/obj}
if [ -d "$SRC" ]; then FILES=$(ls "$SRC"/*.o); else FILES=$SRC; fi
for F in $FILES; do
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$F" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co > $T/k.notes
/opt/rocm/lib/llvm/bin/llvm-readelf -s $T/k.co > $T/k.syms
python3 - $T <<'PY'
import re, sys
T = sys.argv[1]
size = {}
for line in open(T + '/k.syms'):
    t = line.split()
    if len(t) >= 8 and t[3] == 'FUNC':
        size[t[7]] = int(t[2])
t = open(T + '/k.notes').read()
for blk in t.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    print(f"{name[:64]:64s} vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} code {size.get(name, 0):>7}")
PY
rm -rf $T
done
