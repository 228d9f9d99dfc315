"""Build-time guards for the hand-tuned kernels (CPU; they read what build() left behind).

1. Inline assembly: every asm statement of the kernel sources — the generated window macros of grt_slots_gen.inc and the
   hand-written ones — must name, as an output or a clobber, every register it writes IMPLICITLY: SCC (s_and* / s_or* /
   s_andn2* / s_cmp* / s_add* ...), VCC (v_cmp* with a vcc destination); a block that writes EXEC must save it first and
   restore it last.  (Round 3 shipped generated s_and_saveexec_b64 blocks without the SCC clobber for a whole round.)
2. Device assembly: the toolchain's register allocator sometimes puts spill code in FRONT of a join block's EXEC restore
   (gaussian-ray-tracing_amd/csrc/hipcc_via_asm.py; the cause of round 3's "not understood" watchdog failure).  The
   build repairs it; here the assembly the shipped objects were assembled from must be clean, and the repair itself is
   checked on the recorded failing snippet.
3. ISA budget: csrc/build_asm/isa_budget.json (written by build(); regenerated here when the kept assembly is newer) — the camera-ray kernel's frame time follows its
   resident waves (<= 128 VGPRs, <= 9984 B of LDS per wave: 16 waves per CU; 13 waves cost 22 %) and its spill count
   re-draws with every edit: the limits are asserted so that such a change fails here, not at the bench."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-ray-tracing_amd", "csrc")
sys.path.insert(0, CSRC)
import hipcc_via_asm as V  # noqa: E402

SCC_WRITERS = re.compile(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|not|add|sub|addc|subb|min|max|lshl|lshr|ashr|bfe|abs|absdiff|cmp|cmpk|"
                         r"bitcmp|wqm|quadmask|bcnt|ff0|ff1|flbit|lshl[1-4]_add)\w*")
SCC_FREE = re.compile(r"^s_(mov|cmov|cselect|bitset|brev|sext|load|buffer_load|waitcnt|nop|branch|cbranch|getreg|setreg|setprio|mul_i32)\w*")


def asm_statements(text):
    """(position, template instructions, outputs, inputs, clobbers) of every asm statement in a C++ source text"""
    out = []
    for m in re.finditer(r"\basm\s*(volatile)?\s*\(", text):
        i = m.end()
        depth, j, in_str = 1, i, False
        while depth and j < len(text):
            ch = text[j]
            if in_str:
                if ch == "\\":
                    j += 1
                elif ch == '"':
                    in_str = False
            elif ch == '"':
                in_str = True
            elif ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            j += 1
        body = text[i:j - 1].replace("\\\n", " ")
        parts, cur, in_str, depth, k = [], "", False, 0, 0
        while k < len(body):  # split at top-level ':'
            ch = body[k]
            if in_str:
                cur += ch
                if ch == "\\":
                    cur += body[k + 1]
                    k += 1
                elif ch == '"':
                    in_str = False
            elif ch == '"':
                in_str = True
                cur += ch
            elif ch in "([":
                depth += 1
                cur += ch
            elif ch in ")]":
                depth -= 1
                cur += ch
            elif ch == ":" and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
            k += 1
        parts.append(cur)
        tmpl = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
        ins = [x.strip() for x in tmpl.replace("\\t", " ").split("\\n") if x.strip()]
        clob = re.findall(r'"([^"]+)"', parts[3]) if len(parts) > 3 else []
        out.append((text.count("\n", 0, m.start()) + 1, ins, parts[1] if len(parts) > 1 else "", parts[2] if len(parts) > 2 else "", clob))
    return out


def implicit_writes(ins):
    w = set()
    for t in ins:
        if t.startswith(";") or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        ops = [x.strip() for x in t[len(op):].split(",")]
        if SCC_WRITERS.match(op) and not SCC_FREE.match(op):
            w.add("scc")
        if op.startswith("v_cmp") and ops and ops[0] == "vcc":
            w.add("vcc")
        if ops and ops[0] == "exec" or op.endswith("_saveexec_b64"):
            w.add("exec")
    return w


def test_every_asm_statement_declares_its_implicit_register_writes():
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")))
    n = 0
    for f in files:
        for line, ins, outs, inputs, clob in asm_statements(open(f).read()):
            n += 1
            w = implicit_writes(ins)
            where = f"{os.path.basename(f)}:{line}"
            for reg in ("scc", "vcc"):
                assert reg not in w or reg in clob, f"{where}: the asm block writes {reg.upper()} and does not clobber it: {ins[:3]}..."
            if "exec" in w:
                first = ins[0]
                assert (first.startswith("s_mov_b64 %[sv], exec") or first.startswith("s_and_saveexec_b64 %[sv]")), f"{where}: EXEC is written but not saved first"
                assert ins[-1].startswith("s_mov_b64 exec, %[sv]"), f"{where}: EXEC is written but not restored by the last instruction"
                assert re.search(r'\[sv\]\s*"=&s"', outs), f"{where}: the EXEC save register must be an early-clobber output"
    assert n >= 10  # the generated macros (3 window sizes) + the hand-written statements were found at all


def test_the_statement_parser_sees_a_missing_clobber():
    bad = 'asm volatile("s_and_saveexec_b64 %[sv], %[m]\\n\\t" "v_mov_b64 %[k0], -1\\n\\t" "s_mov_b64 exec, %[sv]" : [k0] "+v"(k0), [sv] "=&s"(sv_) : [m] "s"(m_));'
    (line, ins, outs, inputs, clob), = asm_statements(bad)
    assert implicit_writes(ins) == {"scc", "exec"} and clob == []
    good = bad.replace('(m_));', '(m_) : "scc");')
    assert asm_statements(good)[0][4] == ["scc"]


# the join block of the failing build, as the compiler made it (k_render_tile<true,false,false,0,false> of the
# GRT_FIT_APPROX=7 variant, profiles/r04_experiments_log.md): spill store and rematerialised constant under the partial mask
FAILING = """\
.LBB10_271:                             ;   in Loop: Header=BB10_64 Depth=2
	global_load_dwordx4 v[2:5], v[6:7], off
	s_nop 0
	global_load_dwordx4 v[6:9], v[6:7], off offset:16
.LBB10_272:                             ;   in Loop: Header=BB10_64 Depth=2
	v_writelane_b32 v127, s99, 25
	s_waitcnt vmcnt(1)
	v_mov_b32_e32 v17, 0x7f800000
	scratch_store_dword off, v115, off      ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[12:13]
	s_mov_b32 s99, s87
"""


def test_exec_prologue_repair_on_the_recorded_failure():
    assert [t for _, t in V.lint(FAILING)] == ["v_mov_b32_e32 v17, 0x7f800000", "scratch_store_dword off, v115, off      ; 4-byte Folded Spill"]
    fixed, moved = V.repair(FAILING)
    assert len(moved) == 2 and V.lint(fixed) == []
    L = [l.split(";")[0].strip() for l in fixed.split("\n")]
    r = L.index("s_or_b64 exec, exec, s[12:13]")
    assert L[r - 1] == "s_waitcnt vmcnt(1)" and L[r + 1].startswith("v_mov_b32_e32 v17") and L[r + 2].startswith("scratch_store_dword off, v115")
    # a then-branch that block placement merged with its join block is ordinary work under the partial mask: left alone
    merged = "; %bb.264:\n\tv_lshlrev_b32_e32 v3, 2, v3\n\tds_write_b32 v3, v35 offset:9856\n\ts_or_b64 exec, exec, s[6:7]\n"
    assert V.lint(merged) == [] and V.repair(merged)[1] == []


def _built_asm():
    files = sorted(f for f in glob.glob(os.path.join(CSRC, "build_asm", "*.s")) if not f.endswith("_marks.s"))
    assert files, "no device assembly under csrc/build_asm: run __graft_entry__.build() (make -C gaussian-ray-tracing_amd/csrc)"
    return files


def test_shipped_assembly_has_no_spill_code_in_front_of_an_exec_restore():
    names = {os.path.basename(f) for f in _built_asm()}
    assert {"grt_render_tile.s", "grt_render_tile_single.s", "grt_render_stream.s", "grt_render.s", "grt_render_wave.s", "grt_bvh.s", "grt_api.s"} <= names
    for f in _built_asm():
        assert V.lint(open(f).read()) == [], os.path.basename(f)


def _budget():
    """the budget of the assembly that is there NOW: the JSON is a build artefact beside it, made again when any kept .s is newer"""
    files = _built_asm()
    p = os.path.join(CSRC, "build_asm", "isa_budget.json")
    if not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(f) for f in files):
        import subprocess
        marks = os.path.join(CSRC, "build_asm", "grt_render_tile_marks.s")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "profiles", "isa_budget_current.py")] + (["--marks", marks] if os.path.exists(marks) else []))
    return {k["kernel"]: k for k in json.load(open(p))["kernels"]}


def test_isa_budget_of_the_render_kernels():
    b = _budget()
    tile = {k: v for k, v in b.items() if "k_render_tile<" in k}
    assert len(tile) == 32
    for name, k in tile.items():
        mode = int(re.search(r"k_render_tile<\w+, \w+, \w+, (\d)", name).group(1))
        if mode == 2:  # one ray per wave: 3 waves per SIMD, 11 per CU by LDS
            assert k["vgprs"] <= 168 and k["lds_bytes"] <= 14336, (name, k)
        else:          # 4 waves per SIMD, 16 per CU: <= 128 VGPRs and <= 9984 B of LDS (64 KB / 16 x 2.5: 160 KB per CU)
            assert k["vgprs"] <= 128 and k["lds_bytes"] <= 9984, (name, k)
    c3 = b["grt::k_render_tile<false, false, false, 0, false>"]
    # the headline kernel: NO spill instruction inside any loop (what it spills is saved before the passes and reloaded for the
    # pixel write), a bounded number outside, bounded SGPR spill traffic (v_readlane / v_writelane) — more than this means the
    # allocator gave up somewhere new: look before shipping (profiles/tools/movcount.sh shows where)
    assert c3["spill_instructions_in_loops"] <= 1 and c3["spill_instructions"] <= 12 and c3["scratch_bytes"] <= 32, c3
    # (SGPR-spill traffic inside loops predicts the frame: 146 / 171 / 210 v_readlane + v_writelane in loops = -2.7 % / 0 / +3.6 % on C3,
    #  profiles/r04_experiments_log.md 10)
    assert c3["lane_moves"] <= 165 and c3["lane_moves_in_loops"] <= 145 and c3["spilled_sgprs"] <= 8 and c3["instructions"] <= 5200, c3
    c5 = b["grt::k_render_tile<false, false, false, 0, true>"]  # the same with pieces (needle / sheet scenes)
    assert c5["spill_instructions_in_loops"] <= 2 and c5["spill_instructions"] <= 16, c5  # (two in its piece-ownership block)
