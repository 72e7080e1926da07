#!/usr/bin/env python3
"""where a kernel's scratch (spill) instructions sit, by source line -- from the compiler's own assembly, no GPU:
   python tools/spill_map.py csrc/saip_kernel_wave.hip 'saip_cycle_waveILb0E' [extra hipcc flags...]
Compiles the file with -save-temps -gline-tables-only into a temporary directory, then prints, for the first kernel whose mangled name contains
the pattern: instructions, MFMAs, scratch loads / stores in total and per source line (file:line of the innermost inlined frame)."""
import collections, os, re, subprocess, sys, tempfile
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sai-primitives_amd")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", "-save-temps", "-gline-tables-only", "-I", os.path.join(root, "csrc")]
                   + extra + ["-o", os.path.join(td, "x.o"), os.path.join(root, src) if not os.path.isabs(src) else src], cwd=td, check=True, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(td, asm)).read()
m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M)
start, end = m.start(), s.index(".Lfunc_end", m.start())
ft = {int(a): (c or b).split("/")[-1] for a, b, c in re.findall(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
cur, tot, st, ld, mf = None, 0, collections.Counter(), collections.Counter(), 0
for l in s[start:end].split("\n"):
    mm = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if mm:
        cur = (ft.get(int(mm.group(1)), mm.group(1)), int(mm.group(2)))
        continue
    t = l.strip()
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    tot += 1
    mf += t.startswith("v_mfma")
    if t.startswith("scratch_store"):
        st[cur] += 1
    if t.startswith("scratch_load"):
        ld[cur] += 1
print(m.group(1), "instructions", tot, "mfma", mf, "scratch stores", sum(st.values()), "loads", sum(ld.values()))
for k in sorted(set(st) | set(ld), key=lambda k: (k[0] != os.path.basename(src), k[0], k[1])):
    print(f"  {k[0]}:{k[1]:<5d} st {st[k]:3d} ld {ld[k]:3d}")
