"""diagnostic: per-phase instruction mix of a kernel built with s_memtime stamps (-DSAIP_STAMP_OCT -save-temps):
   python tools/asm_phase_counts.py file.s mangled_kernel_name"""
import re, sys
f, name = sys.argv[1], sys.argv[2]
lines = open(f).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.size') and name in lines[i])
def newseg(): return dict(valu=0, f64=0, dpp=0, trans=0, cnd=0, mov=0, lds=0, vmem=0, smem=0, salu=0, wait=0, branch=0)
seg, cur = [], newseg()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
    op = l.split()[0]
    if op == 's_memtime':
        seg.append(cur); cur = newseg(); continue
    if op.startswith('v_'):
        cur['valu'] += 1
        if 'dpp' in l: cur['dpp'] += 1
        if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log)', op): cur['trans'] += 1
        if re.match(r'v_(fma|mul|add|max|min|fmac)_f64', op): cur['f64'] += 1
        if op.startswith('v_cndmask'): cur['cnd'] += 1
        if op.startswith(('v_mov', 'v_accvgpr', 'v_readlane', 'v_writelane', 'v_readfirstlane')): cur['mov'] += 1
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
    elif op.startswith(('s_load', 's_buffer')): cur['smem'] += 1
    elif op.startswith(('s_waitcnt', 's_barrier', 's_nop')): cur['wait'] += 1
    elif op.startswith(('s_cbranch', 's_branch')): cur['branch'] += 1
    elif op.startswith('s_'): cur['salu'] += 1
seg.append(cur)
tot = newseg()
print("seg " + " ".join(f"{k:>6s}" for k in tot))
for i, c in enumerate(seg):
    print(f"{i:3d} " + " ".join(f"{c[k]:6d}" for k in c))
    for k in c: tot[k] += c[k]
print("tot " + " ".join(f"{tot[k]:6d}" for k in tot))
