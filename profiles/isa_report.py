"""Static report on a kernel's ISA: instruction mix, scratch (spill) traffic and where it sits.
usage: python profiles/isa_report.py <asm file from hipcc -S --cuda-device-only> <kernel name fragment>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and frag in l and l.rstrip().endswith(frag and l.split(":")[0]) or (l.startswith("_Z") and frag in l.split(":")[0]))
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
body = txt[start:end]
ins = [(i, l) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
print("instructions", len(ins))
mix = collections.Counter(l.split()[0] for _, l in ins)
groups = collections.Counter()
for op, n in mix.items():
    g = "scratch" if op.startswith("scratch") else "global" if op.startswith(("global", "flat", "buffer")) else "ds" if op.startswith("ds_") else "smem" if op.startswith("s_load") or op.startswith("s_buffer") else "valu" if op.startswith("v_") else "salu/ctl"
    groups[g] += n
print(dict(groups))
for op in ("v_div_scale_f32", "v_div_fmas_f32", "v_sqrt_f32", "v_rcp_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "s_waitcnt", "v_readlane_b32", "v_writelane_b32"):
    print(f"  {op:24s} {mix.get(op, 0)}")
label = None
depth = []
for i, l in enumerate(body):
    if re.match(r"^\.LBB\d+_\d+:", l):
        label = l.split(":")[0]
    if "scratch_" in l:
        print(f"  line {i:6d} after {label}: {l.strip()}")
