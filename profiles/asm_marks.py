"""Instructions between the named markers of a -DPT_ASM_MARKS assembly listing, in layout order.
usage: python profiles/asm_marks.py <asm file> <kernel name fragment>
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize
       -DPT_ASM_MARKS -Iinclude -Ipathtracer_amd/csrc -S --cuda-device-only pathtracer_amd/csrc/pt_kernels.hip -o <asm file>"""
import sys

txt = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and frag in l.split(":")[0])
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
mark = "(entry)"
n = {"v": 0, "s": 0, "m": 0, "t": 0}
rows = []
for l in txt[start:end]:
    t = l.strip()
    if t.startswith("; PTMARK"):
        rows.append((mark, dict(n)))
        mark = t.split()[2]
        n = {"v": 0, "s": 0, "m": 0, "t": 0}
    elif l.startswith("\t") and not t.startswith((".", ";")):
        op = t.split()[0]
        if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
            n["t"] += 1
        elif op.startswith("v_"):
            n["v"] += 1
        elif op.startswith("s_"):
            n["s"] += 1
        else:
            n["m"] += 1
rows.append((mark, n))
for m, c in rows:
    print(f"{m:20s} vector {c['v']:5d}  transcendental {c['t']:3d}  scalar {c['s']:5d}  memory {c['m']:4d}  all {sum(c.values()):5d}")
