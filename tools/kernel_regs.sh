#!/bin/bash
# usage: tools/kernel_regs.sh <object.o> [name-filter]   -- VGPR/AGPR/SGPR/scratch/LDS per gfx950 kernel of a hipcc object
set -e
o=$1; f=${2:-.}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$o" $tmp/fb
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/co | python3 -c "
import sys,re
cur={}
rows=[]
for l in sys.stdin:
    m=re.match(r'\s*-?\s*\.(\w+):\s*(.*)',l)
    if not m: continue
    k,v=m.groups()
    if k=='name' and v.startswith('_Z') or k=='name' and v.startswith('k_'):
        pass
    if k in('agpr_count','group_segment_fixed_size','private_segment_fixed_size','sgpr_count','vgpr_count','symbol','name'):
        cur[k]=v.strip()
    if k=='wavefront_size':
        rows.append(cur); cur={}
for r in rows:
    n=r.get('name','?')
    if re.search('$f',n): print('%-70s vgpr %4s agpr %4s sgpr %4s scratch %6s lds %6s'%(n[:70],r.get('vgpr_count'),r.get('agpr_count'),r.get('sgpr_count'),r.get('private_segment_fixed_size'),r.get('group_segment_fixed_size')))
"
rm -rf $tmp
