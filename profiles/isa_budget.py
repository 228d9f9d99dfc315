#!/usr/bin/env python3
"""Static instruction budget of the render kernel, per source section.

  hipcc ... -DGRT_MARKS -S --cuda-device-only -o stream_marks.s csrc/grt_render_stream.hip   (see profiles/collect_isa.sh)
  python3 profiles/isa_budget.py stream_marks.s <mangled-name substring> [trip-counts.json]

-DGRT_MARKS makes every GRT_W(section) of the kernel source leave a `; GRT_MARK section` comment in the assembly.
The script cuts the kernel's instruction stream at the markers (file order = source order at -O3 for this kernel:
the sections are separated by wave-uniform branches) and counts instructions per class in each piece.  Multiplying
by the WAVE-level trip counts of a -DGRT_WPROF run (the same markers count executions there) gives the dynamic
budget per section that DESIGN.md §6 quotes.  Pieces: the text between marker k and marker k+1 belongs to k.
"""
import json, re, sys, collections

path, want = sys.argv[1], sys.argv[2]
trips = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and want in l.split(":")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))

def klass(op):
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "xlane"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "valu_cmp"
    if "dpp" in op: return "valu_dpp"
    if op.startswith("v_mov_b64"): return "valu_mov64"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"

pieces = []  # (marker, Counter)
cur = ("prologue", collections.Counter())
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r"; GRT_MARK (\w+)", t)
    if m:
        pieces.append(cur)
        cur = (m.group(1), collections.Counter())
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    # an inline-asm block holds several instructions on separate lines already (\n\t in the template)
    cur[1][klass(op)] += 1
pieces.append(cur)

cols = ["valu", "valu_cmp", "valu_mov64", "valu_dpp", "xlane", "salu", "branch", "smem", "lds", "vmem", "waitcnt", "nop"]
print(f"{'piece':16s}" + "".join(f"{c:>11s}" for c in cols) + f"{'VALU all':>10s}{'total':>8s}")
out = []
for name, c in pieces:
    v = c["valu"] + c["valu_cmp"] + c["valu_mov64"] + c["valu_dpp"] + c["xlane"]
    tot = sum(c.values())
    print(f"{name:16s}" + "".join(f"{c[k]:11d}" for k in cols) + f"{v:10d}{tot:8d}")
    out.append({"piece": name, **{k: c[k] for k in cols}, "valu_all": v, "total": tot})
json.dump(out, open(path.replace(".s", "_budget.json"), "w"), indent=1)
