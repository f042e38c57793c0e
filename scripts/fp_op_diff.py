"""Development aid: do two builds of one translation unit do the same floating-point arithmetic?
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -c pogs_amd/csrc/<unit>.hip -o a.s     (once per build)
    python scripts/fp_op_diff.py a.s b.s [label]
Per kernel (by mangled name) the multiplies, FMAs and adds are counted (a packed instruction counts twice) and kernels whose
counts differ are listed.  With -ffp-contract=fast which products get fused follows instruction selection, so an edit that
only moves code around a dot product can change its roundings; this shows where (round 5: profiles/NOTES_r05.md)."""
import re, sys
from collections import Counter
def kernels(path):
    out={}; name=None; buf=[]
    for line in open(path):
        m=re.match(r'^(_Z\w+):', line)
        if m:
            name=m.group(1); buf=[]; out[name]=buf; continue
        if name is None: continue
        if line.startswith('.Lfunc_end'): name=None; continue
        t=line.strip().split()
        if not t or t[0].startswith(';') or t[0].startswith('.'): continue
        buf.append(t[0])
    return out
def flops(ops):
    c=Counter()
    for o in ops:
        o=o.replace('_dpp','').replace('_e32','').replace('_e64','')
        m=re.match(r'v_(pk_)?(fma|fmac|mul|add|sub|mad)_(f32|f64)$',o)
        if not m: continue
        k={'fmac':'fma','mad':'fma','sub':'add'}.get(m.group(2),m.group(2))+'_'+m.group(3)
        c[k]+=2 if m.group(1) else 1
    return c
b=kernels(sys.argv[1]); n=kernels(sys.argv[2]); bad=0
for k in b:
    if k not in n: continue
    cb,cn=flops(b[k]),flops(n[k])
    if cb!=cn:
        bad+=1
        print(k[:100], {x:(cb.get(x,0),cn.get(x,0)) for x in set(cb)|set(cn) if cb.get(x,0)!=cn.get(x,0)})
print(sys.argv[3] if len(sys.argv)>3 else '', "kernels", len(b), "differing", bad)
