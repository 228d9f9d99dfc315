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
import subprocess
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


def test_exec_prologue_lint_and_narrow_repair_on_the_recorded_failure():
    # the lint names both instructions; the build does NOT move a rematerialised constant on its own (a then-branch's phi copy
    # looks the same: ADVICE r04) — it fails with the snippet
    assert [t for _, t in V.lint(FAILING)] == ["v_mov_b32_e32 v17, 0x7f800000", "scratch_store_dword off, v115, off      ; 4-byte Folded Spill"]
    with pytest.raises(V.ExecPrologueError, match="not a VGPR spill store: v_mov_b32_e32 v17"):
        V.repair(FAILING)
    # ... unless the block is the TARGET of an s_cbranch_execz (rule A): no then-branch work can stand there, the constant is the
    # allocator's and moves with the store (the instrumented builds of round 5 drew exactly this)
    at_join = "\ts_cbranch_execz .LBB10_272\n" + FAILING
    fixed, moved = V.repair(at_join)
    assert len(moved) == 2 and V.lint(fixed) == []
    L = [l.split(";")[0].strip() for l in fixed.split("\n")]
    r = L.index("s_or_b64 exec, exec, s[12:13]")
    assert L[r - 1] == "s_waitcnt vmcnt(1)" and L[r + 1].startswith("v_mov_b32_e32 v17") and L[r + 2].startswith("scratch_store_dword off, v115")
    # the spill store alone is what the repair is for: it moves behind the restore, the waitcnt IN FRONT of it stays where it is
    store_only = FAILING.replace("\tv_mov_b32_e32 v17, 0x7f800000\n", "")
    fixed, moved = V.repair(store_only)
    assert len(moved) == 1 and V.lint(fixed) == []
    L = [l.split(";")[0].strip() for l in fixed.split("\n")]
    r = L.index("s_or_b64 exec, exec, s[12:13]")
    assert L[r - 1] == "s_waitcnt vmcnt(1)" and L[r + 1].startswith("scratch_store_dword off, v115")
    # a then-branch that block placement merged with its join block is ordinary work under the partial mask: left alone
    merged = "; %bb.264:\n\tv_lshlrev_b32_e32 v3, 2, v3\n\tds_write_b32 v3, v35 offset:9856\n\ts_or_b64 exec, exec, s[6:7]\n"
    assert V.lint(merged) == [] and V.repair(merged)[1] == []


def test_exec_prologue_repair_refuses_what_depends_on_its_place():
    # a RELOAD in front of the restore, its s_waitcnt and the v_readlane that consumes it: not moved, the build fails
    reload = """\
\ts_cbranch_execz .LBB3_9
; %bb.8:
\tv_add_f32_e32 v2, v2, v3
.LBB3_9:
\tscratch_load_dword v44, off, off offset:12 ; 4-byte Folded Reload
\ts_waitcnt vmcnt(0)
\tv_readlane_b32 s6, v44, 3
\tv_readlane_b32 s7, v44, 4
\ts_or_b64 exec, exec, s[6:7]
"""
    assert len(V.lint(reload)) == 1
    with pytest.raises(V.ExecPrologueError, match="not a VGPR spill store"):
        V.repair(reload)
    # a spill store whose register a lane move BEHIND it (staying in front of the restore) writes, or with a vmcnt wait behind it
    # that would stay behind together with something else
    for tail, why in (("\tv_writelane_b32 v115, s4, 2\n", "names a moved register"),
                      ("\ts_waitcnt vmcnt(0)\n\tv_readlane_b32 s4, v9, 1\n", "vector-memory counter")):
        t = ".LBB1_2:\n\tscratch_store_dword off, v115, off      ; 4-byte Folded Spill\n" + tail + "\ts_or_b64 exec, exec, s[12:13]\n"
        assert len(V.lint(t)) == 1
        with pytest.raises(V.ExecPrologueError, match=why):
            V.repair(t)


def test_exec_prologue_repair_moves_a_tail_of_stores_and_waits_as_one_piece():
    # round 5 (the bags' chunks): spill stores interleaved with the waits for the loads whose results they spill.  From the first
    # store to the restore there is nothing but stores, s_waitcnt and s_nop: the tail goes behind the restore in its own order —
    # the same as hoisting the restore over it
    t = """\
\ts_cbranch_execz .LBB1_284
.LBB1_283:
\tglobal_load_dwordx4 v[6:9], v[2:3], off offset:16
\ts_nop 0
\tglobal_load_dwordx4 v[2:5], v[2:3], off
.LBB1_284:
\ts_nop 0
\tscratch_store_dword off, v102, off offset:24 ; 4-byte Folded Spill
\ts_waitcnt vmcnt(4)
\tscratch_store_dword off, v93, off offset:20 ; 4-byte Folded Spill
\ts_waitcnt vmcnt(2)
\tscratch_store_dword off, v101, off offset:16 ; 4-byte Folded Spill
\ts_or_b64 exec, exec, s[12:13]
\ts_mov_b32 s30, s59
"""
    assert len(V.lint(t)) == 3
    fixed, moved = V.repair(t)
    assert V.lint(fixed) == [] and len(moved) == 5
    L = [l.split(";")[0].strip() for l in fixed.split("\n")]
    r = L.index("s_or_b64 exec, exec, s[12:13]")
    assert L[r - 1] == "s_nop 0" and L[r - 2] == ".LBB1_284:"
    assert L[r + 1:r + 7] == ["scratch_store_dword off, v102, off offset:24", "s_waitcnt vmcnt(4)", "scratch_store_dword off, v93, off offset:20",
                              "s_waitcnt vmcnt(2)", "scratch_store_dword off, v101, off offset:16", "s_mov_b32 s30, s59"]


def test_exec_prologue_lint_is_structural_at_join_labels():
    # rule (A): the target of an s_cbranch_execz is entered with EXEC = 0 — ANYTHING EXEC-dependent between its label and the
    # restore is wrong there, marked by the allocator or not (a constant move, an LDS write, a v_readfirstlane)
    for ins in ("v_mov_b32_e32 v3, 0", "ds_write_b32 v3, v35 offset:16", "v_readfirstlane_b32 s4, v9", "global_load_dword v4, v[4:5], off"):
        t = f"\ts_and_saveexec_b64 s[6:7], vcc\n\ts_cbranch_execz .LBB2_4\n; %bb.3:\n\tv_add_f32_e32 v1, v1, v2\n.LBB2_4:\n\t{ins}\n\ts_or_b64 exec, exec, s[6:7]\n"
        assert [x for _, x in V.lint(t)] == [ins], ins
        if ins.startswith("v_mov_b32"):  # a constant at a rule-(A) label is the allocator's rematerialisation: it moves behind the restore
            fixed, moved = V.repair(t)
            assert moved == [ins] and V.lint(fixed) == []
        else:
            with pytest.raises(V.ExecPrologueError):
                V.repair(t)
    # ... while an OUT-OF-LINE then-block (a label reached by s_cbranch_execnz) holds ordinary work in front of its copy of the restore
    outl = "\ts_and_saveexec_b64 s[18:19], vcc\n\ts_cbranch_execnz .LBB13_28\n.LBB13_20:\n\ts_or_b64 exec, exec, s[18:19]\n\ts_endpgm\n" \
           ".LBB13_28:\n\tv_lshl_add_u64 v[4:5], v[2:3], 2, s[16:17]\n\tglobal_load_dword v4, v[4:5], off\n\ts_or_b64 exec, exec, s[18:19]\n"
    assert V.lint(outl) == []
    # the scan ends at an earlier EXEC write of the block (what stands behind it runs under the mask that write made) ...
    behind = ".LBB4_7:\n\ts_andn2_b64 exec, exec, s[0:1]\n\tscratch_store_dword off, v9, off ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[2:3]\n"
    assert V.lint(behind) == []
    # ... restores inside hand-written asm regions are the region's own business, and labels in there do not open blocks ...
    region = ".LBB5_1:\n\t;;#ASMSTART\n\ts_mov_b64 s[18:19], exec\n\t.Lgrt_e2_0:\n\tv_cmp_lt_u64 s[24:25], v[80:81], v[44:45]\n\ts_mov_b64 exec, s[24:25]\n" \
             "\tv_mov_b64 v[44:45], v[80:81]\n\ts_mov_b64 exec, s[18:19]\n\t;;#ASMEND\n"
    assert V.lint(region) == []
    # ... and `s_mov_b64 exec, sN` with sN formed in the block itself is the expanded s_and_saveexec_b64 of an if: it narrows EXEC
    narrow = "\ts_cbranch_execz .LBB6_280\n.LBB6_280:\n\tv_mov_b32_e32 v82, 0\n\ts_mov_b64 s[0:1], exec\n\ts_and_b64 s[0:1], s[0:1], s[4:5]\n\ts_mov_b64 exec, s[0:1]\n"
    assert V.lint(narrow) == []


def test_the_recorded_reproducer_still_shows_the_compiler_defect(tmp_path):
    """profiles/r05_exec_prologue_repro: one kernel's optimised IR that the image's `llc -O3` turns into spill stores in front of
    a join block's EXEC restore.  While this passes the build's post-pass is needed; when it fails the tool chain was fixed."""
    import gzip
    import shutil
    llc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llc")
    if not os.path.exists(llc):
        pytest.skip("no llc")
    rep = os.path.join(ROOT, "profiles", "r05_exec_prologue_repro")
    ll = tmp_path / "r.ll"
    with gzip.open(os.path.join(rep, "k_render_tile_counted_sh_pieces.ll.gz"), "rb") as f, open(ll, "wb") as g:
        shutil.copyfileobj(f, g)
    out = tmp_path / "r.s"
    subprocess.run([llc, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", str(ll), "-o", str(out)], check=True)
    text = open(out).read()
    bad = V.lint(text)
    assert bad and all("Folded Spill" in t for _, t in bad), bad
    fixed, moved = V.repair(text)  # ... and the build's repair takes this very case (stores and the waits between them, as one piece)
    assert V.lint(fixed) == [] and len(moved) >= len(bad)
    r = subprocess.run([sys.executable, os.path.join(rep, "check.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "Folded Spill" in r.stdout


def _built_asm():
    files = sorted(f for f in glob.glob(os.path.join(CSRC, "build_asm", "*.s")) if not f.endswith("_marks.s"))
    assert files, "no device assembly under csrc/build_asm: run __graft_entry__.build() (make -C gaussian-ray-tracing_amd/csrc)"
    return files


def test_shipped_assembly_has_no_spill_code_in_front_of_an_exec_restore():
    names = {os.path.basename(f) for f in _built_asm()}
    assert {"grt_render_tile.s", "grt_render_tile_single.s", "grt_render_tile_quad.s", "grt_render_stream.s", "grt_render.s", "grt_render_wave.s", "grt_bvh.s", "grt_api.s"} <= names
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
    assert len(tile) == 36
    for name, k in tile.items():
        mode = int(re.search(r"k_render_tile<\w+, \w+, \w+, (\d)", name).group(1))
        if mode == 2:  # one ray per wave: 3 waves per SIMD, 11 per CU by LDS
            assert k["vgprs"] <= 168 and k["lds_bytes"] <= 14336, (name, k)
        elif mode == 3:  # quad kernel (lanes = rays x slots): 3 waves per SIMD; the uninstrumented instantiations spill nothing
            assert k["vgprs"] <= 168 and k["lds_bytes"] <= 14336, (name, k)
            if name.startswith("grt::k_render_tile<false"):
                assert k["spill_instructions"] == 0 and k["spilled_sgprs"] <= 8 and k["lane_moves_in_loops"] <= 130, (name, k)
        else:          # 4 waves per SIMD, 16 per CU: <= 128 VGPRs and <= 9984 B of LDS (64 KB / 16 x 2.5: 160 KB per CU)
            assert k["vgprs"] <= 128 and k["lds_bytes"] <= 9984, (name, k)
    c3 = b["grt::k_render_tile<false, false, false, 0, false>"]
    # the headline kernel: NO spill instruction inside any loop (what it spills is saved before the passes and reloaded for the
    # pixel write), a bounded number outside, bounded SGPR spill traffic (v_readlane / v_writelane) — more than this means the
    # allocator gave up somewhere new: look before shipping (profiles/tools/movcount.sh shows where)
    assert c3["spill_instructions_in_loops"] <= 1 and c3["spill_instructions"] <= 12 and c3["scratch_bytes"] <= 32, c3
    # (SGPR-spill traffic inside loops predicts the frame: 146 / 171 / 210 v_readlane + v_writelane in loops = -2.7 % / 0 / +3.6 % on C3,
    #  profiles/r04_experiments_log.md 10)
    # (spilled SGPRs: 6 until the bags' size classes, 12 with them — two more values live per tile; the frame did not move: r05 log 7)
    assert c3["lane_moves"] <= 165 and c3["lane_moves_in_loops"] <= 145 and c3["spilled_sgprs"] <= 12 and c3["instructions"] <= 5200, c3
    # the mesh frame's stages (C4): guarded where they stand (VERDICT r04 asked for <= 200 / <= 20 / <= 300; not reached this round —
    # what is asserted is that they do not get WORSE unseen, the uninstrumented instantiations that frames run)
    for name, k in tile.items():
        if not name.startswith("grt::k_render_tile<false"):
            continue
        mode = int(re.search(r"k_render_tile<\w+, \w+, \w+, (\d)", name).group(1))
        if mode == 1:
            assert k["lane_moves_in_loops"] <= 260 and k["spill_instructions_in_loops"] <= 60, (name, k)
        if mode == 2:
            assert k["lane_moves_in_loops"] <= 400 and k["spill_instructions_in_loops"] <= 24, (name, k)
    # round 6: C4's critical path is the primary stage (with stage 1 — the tile's rays walking the mesh tree together — fused into its
    # head) and then the one-ray-per-wave kernel; the bundle kernel runs in that kernel's shadow.  The fused instantiation spills more
    # OUTSIDE its loops (85 VGPRs, 34 instructions) and must keep its loops clean; mode 2 stands at 372 lane moves in loops, mode 1 at
    # 250 / 23 (VERDICT r05 asked for <= 250 / <= 8 there: not done — see profiles/r06_experiments_log.md 4)
    c4 = b["grt::k_render_tile<false, false, true, 0, false>"]
    assert c4["spill_instructions_in_loops"] == 0 and c4["lane_moves_in_loops"] <= 150 and c4["spill_instructions"] <= 40, c4
    m1 = b["grt::k_render_tile<false, false, true, 1, false>"]
    m2 = b["grt::k_render_tile<false, false, true, 2, false>"]
    assert m1["instructions"] > 4000, m1  # (the budget reads a kernel to the END of its function: the bundle kernel has an early exit)
    assert m1["spill_instructions_in_loops"] <= 30 and m2["lane_moves_in_loops"] <= 390 and m2["spill_instructions"] == 0, (m1, m2)
    c5 = b["grt::k_render_tile<false, false, false, 0, true>"]  # the same with pieces (needle / sheet scenes)
    assert c5["spill_instructions_in_loops"] <= 2 and c5["spill_instructions"] <= 16, c5  # (two in its piece-ownership block)


def test_non_default_configurations_of_the_tile_kernel_compile_and_pass_the_lints(tmp_path):
    """The compile-time variants the tile kernel still has (GRT_TILE_DIAG: trip counters; GRT_TILE_CHECK: invariant checks;
    GRT_MARKS: section marks; GRT_TILE_KS = 8 is the single-ray translation unit of every build) go through the same
    hipcc -> assembly -> lint -> assembler path as the shipped objects, so that none of them rots unseen."""
    import subprocess
    flags = subprocess.check_output(["make", "-s", "-C", CSRC, "print-flags"], text=True).split()
    hipcc = subprocess.check_output(["make", "-s", "-C", CSRC, "print-hipcc"], text=True).strip()
    env = dict(os.environ, HIPCC=hipcc)
    procs = []
    for name, extra in (("diag", ["-DGRT_TILE_DIAG"]), ("check", ["-DGRT_TILE_CHECK"]), ("marks", ["-DGRT_MARKS"])):
        out = tmp_path / name
        out.mkdir()
        cmd = [sys.executable, os.path.join(CSRC, "hipcc_via_asm.py"), "--keep-asm", str(out), str(out / "tile.o"),
               os.path.join(CSRC, "grt_render_tile.hip")] + flags + extra
        procs.append((name, out, subprocess.Popen(cmd, cwd=CSRC, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, out, p in procs:
        log = p.communicate()[0]
        assert p.returncode == 0, f"-D variant '{name}' does not build:\n{log[-3000:]}"
        text = open(out / "tile.s").read()
        assert V.lint(text) == [], name
        assert (out / "tile.o").stat().st_size > 100000
