#!/usr/bin/env python3
"""isa_budget_current.py [--marks file.s] [--out file.json] — the register / LDS / spill budget of every kernel the library ships, from the
device assembly the build keeps (gaussian-ray-tracing_amd/csrc/build_asm/*.s, written by hipcc_via_asm.py) -> build_asm/isa_budget.json
(an untracked build artefact: tests/test_isa_lint.py reads it, and regenerates it when the assembly is newer; `--out profiles/isa_budget_rNN.json`
writes the reviewed snapshot a round commits).

Why: the tile kernel's frame time follows its resident waves (128 VGPRs and <= 9984 B of LDS = 16 waves per CU; 13 waves:
+22 %) and the register allocation of its hot loop re-draws with every edit (an atomicOr cost 13-16 %, a 30-line cold block
took the spill instructions 21 -> 44: DESIGN.md 5.2).  __graft_entry__.build() runs this after compiling, and
tests/test_isa_lint.py asserts the limits, so that such a change fails at build time instead of at the bench.

Per kernel: VGPRs, SGPRs, spilled VGPRs / SGPRs (registers), scratch bytes, LDS bytes, instructions, VALU / SALU instructions,
spill instructions (scratch_*), v_readlane / v_writelane (SGPR spill traffic).  With --marks (a -DGRT_MARKS -S build of the
tile kernel): the same static counts of the camera-ray kernel per marked section of its source."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "gaussian-ray-tracing_amd", "csrc", "build_asm")
OUT = os.path.join(ASM, "isa_budget.json")
C3_KERNEL = "_ZN3grt12_GLOBAL__N_113k_render_tileILb0ELb0ELb0ELi0ELb0EEEvNS_10RenderArgsE"


def demangle(names):
    try:
        r = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True, check=True)
        return [re.sub(r"^void ", "", x.replace("(anonymous namespace)::", "").replace("(grt::RenderArgs)", "")) for x in r.stdout.strip().split("\n")]
    except Exception:
        return names


def instr_stats(lines):
    """(the assembly printer says for every machine basic block which loop it is in: `.LBBx_y: ; in Loop: Header=… Depth=N`)"""
    c = {"instructions": 0, "valu": 0, "salu": 0, "spill_instructions": 0, "spill_instructions_in_loops": 0, "lane_moves": 0,
         "lane_moves_in_loops": 0, "lds": 0, "vmem": 0, "smem": 0}
    depth = 0
    for l in lines:
        m = re.match(r"^\.LBB\d+_\d+:(.*)$", l)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(1))
            depth = int(d.group(1)) if d else 0
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        c["instructions"] += 1
        if op.startswith(("v_readlane", "v_writelane")):
            c["lane_moves"] += 1
            c["lane_moves_in_loops"] += 1 if depth else 0
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith(("s_load", "s_buffer_load")):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("scratch_"):
            c["spill_instructions"] += 1
            c["spill_instructions_in_loops"] += 1 if depth else 0
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
    return c


def kernels_of(path):
    L = open(path).read().split("\n")
    meta = {}
    cur = None
    for l in L:  # the code-object metadata at the end of the file
        m = re.match(r"^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)", l)
        if not m:
            if l.strip().startswith("- .agpr_count") or l.strip().startswith("- .args"):
                cur = {}
            continue
        if cur is None:
            cur = {}
        k, v = m.group(1), m.group(2)
        if k == "name":
            meta[v] = cur
            cur["name"] = v
        else:
            cur[k] = int(v)
    out = {}
    i = 0
    while i < len(L):
        m = re.match(r"^(_Z\w+):", L[i])
        if m and m.group(1) in meta:
            # (to the end of the FUNCTION, not to its first s_endpgm: a kernel with an early exit — the bundle kernel's skip of a chunk
            #  on the early list, round 6 — has several)
            e = next(k for k in range(i, len(L)) if L[k].startswith(".Lfunc_end")) - 1
            out[m.group(1)] = (meta[m.group(1)], L[i + 1:e + 1])
            i = e
        i += 1
    return out


def sections(lines):
    pieces, cur, name = [], [], "prologue"
    seen = {}
    for l in lines:
        m = re.match(r"\s*; GRT_MARK (\w+)", l)
        if m:
            pieces.append((name, cur))
            name = m.group(1)
            seen[name] = seen.get(name, 0) + 1
            if seen[name] > 1:
                name += str(seen[name])
            cur = []
        else:
            cur.append(l)
    pieces.append((name, cur))
    return [{"section": n, **instr_stats(c)} for n, c in pieces]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    marks, out = None, OUT
    while argv:
        if argv[0] == "--marks":
            marks = argv[1]
        elif argv[0] == "--out":
            out = argv[1]
        else:
            raise SystemExit(__doc__)
        argv = argv[2:]
    res = {"source": "gaussian-ray-tracing_amd/csrc/build_asm/*.s (the assembly the shipped objects were assembled from)", "kernels": []}
    for path in sorted(glob.glob(os.path.join(ASM, "*.s"))):
        if path.endswith("_marks.s"):
            continue
        ks = kernels_of(path)
        names = list(ks)
        for mangled, nice in zip(names, demangle(names) if names else []):
            md, body = ks[mangled]
            st = instr_stats(body)
            res["kernels"].append({"kernel": nice, "mangled": mangled, "file": os.path.basename(path),
                                   "vgprs": md.get("vgpr_count"), "sgprs": md.get("sgpr_count"),
                                   "spilled_vgprs": md.get("vgpr_spill_count"), "spilled_sgprs": md.get("sgpr_spill_count"),
                                   "scratch_bytes": md.get("private_segment_fixed_size"), "lds_bytes": md.get("group_segment_fixed_size"), **st})
        rep = path[:-2] + ".repairs.txt"
        if os.path.exists(rep):
            res.setdefault("exec_prologue_repairs", {})[os.path.basename(path)] = open(rep).read().strip().split("\n")
    if marks and os.path.exists(marks):
        ks = kernels_of(marks)
        if C3_KERNEL in ks:
            res["camera_ray_kernel_sections"] = {"kernel": "grt::k_render_tile<false, false, false, 0, false>",
                                                 "build": "-DGRT_MARKS (the markers are empty asm statements: the allocation may differ slightly from the shipped object's)",
                                                 "vgprs": ks[C3_KERNEL][0].get("vgpr_count"), "scratch_bytes": ks[C3_KERNEL][0].get("private_segment_fixed_size"),
                                                 "sections": sections(ks[C3_KERNEL][1])}
    json.dump(res, open(out, "w"), indent=1)
    print(f"isa budget: {len(res['kernels'])} kernels -> {os.path.relpath(out, ROOT)}")


if __name__ == "__main__":
    main()
