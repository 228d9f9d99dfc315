#!/usr/bin/env python3
"""hipcc_via_asm.py <out.o> <src.hip> [hipcc flags...] — compile one HIP translation unit THROUGH its device assembly, so
that the assembly can be checked (and repaired) before it becomes the code object that ships.

Why (profiles/r04_experiments_log.md, "the watchdog failure of round 3"): this toolchain's register allocator sometimes
puts what it inserts at the top of a control-flow join block — a VGPR spill store, a rematerialised constant — in FRONT
of the block's `s_or_b64 exec, exec, sN`, i.e. under the partial EXEC mask of the branch that just ended.  The lanes that
skipped the branch then keep a stale scratch slot / register, and the value reloaded later under the full mask is
garbage for them.  It happens where an SGPR spill (v_writelane) and a VGPR spill meet at the same block top, so only in
the instantiations under the highest register pressure, and any edit of the kernel re-draws where.  Round 3's
`GRT_FIT_APPROX=7` build hit it in k_render_tile<true,false,false,0,false> (a loop-carried per-lane counter: "watchdog
expired", "absurd counters"); the SHIPPED round-3 library had the same pattern in <true,true,false,0,true>.

Steps (what `hipcc -c` does, with one stop in the middle):
  1. hipcc --cuda-device-only -S           -> device assembly
  2. exec-prologue lint + repair            a labelled block whose first EXEC write widens EXEC must have nothing EXEC-dependent
                                            in front of that write (structural: see "the rule" below); VGPR spill stores found
                                            there move BEHIND the restore when nothing depends on their place, anything else
                                            fails the build with the snippet; then the lint must find nothing
                                            (profiles/tools/exec_prologue_lint.py applies it to any assembly file)
  3. clang -x assembler, lld, clang-offload-bundler -> fat binary
  4. hipcc --cuda-host-only -fcuda-include-gpubinary -> the object

`--keep-asm DIR` keeps the (repaired) assembly there (the ISA budget of build(), tests/test_isa_lint.py)."""
import os
import re
import subprocess
import sys
import tempfile

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
LLVM = os.path.join(ROCM, "lib", "llvm", "bin")
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
ARCH = os.environ.get("GRT_ARCH", "gfx950")

# ---- the rule ------------------------------------------------------------------------------------------------------------
# (A) structural.  A block that is the target of an `s_cbranch_execz` is a JOIN block: on that edge it is entered with
#     EXEC = 0, so the compiler never schedules EXEC-dependent work between its label and the instruction that widens EXEC
#     again (s_or_b64 exec, exec, s[..] / s_mov_b64 exec, s[..] / s_or_saveexec_b64) — what stands there was put there by the
#     register allocator at the "top of the block" (spill stores / reloads, rematerialised constants) and runs under the
#     partial mask of the branch that just ended.  ANY EXEC-dependent instruction there (vector ALU, vector / scratch / LDS
#     memory, v_readfirstlane) is an offender, whatever it looks like.
# (B) by the allocator's marks.  Elsewhere (a join block whose skip branch was removed is reached by falling through only,
#     `; %bb.N:`; an out-of-line then-block carries a label too) ordinary vector work in front of a restore is a then-branch
#     that block placement merged with its join: it belongs under the partial mask.  There a block is taken for a join only
#     when EVERYTHING EXEC-dependent between its top and the restore is of the allocator's own making (`Folded Spill` /
#     `Folded Reload`, constant moves) and at least one spill / reload is among it.
# Either scan ends at the block's top, at a branch, at any earlier EXEC write of the block and at a hand-written asm region
# (those save and restore EXEC themselves: tests/test_isa_lint.py); restores INSIDE asm regions are not looked at.
#
# The repair is deliberately narrow: offenders move behind the restore only when ALL of them are VGPR spill STORES
# (`scratch_store_dword* ... ; N-byte Folded Spill`: the value of the lanes that skipped the branch must reach the slot too)
# — at a rule-(A) label also rematerialised constants (`v_mov_b32 vN, const`: nothing there can be a then-branch's phi copy) —
# and nothing that stays between a moved store and the restore waits on the vector-memory counter or names a moved register
# (s_waitcnt vmcnt / v_readlane / v_writelane: their order against the store would change).  One exception, because it changes
# no order at all: when EVERYTHING from the first moved store to the restore is a moved instruction, an s_waitcnt or an s_nop
# (the allocator interleaves the stores with the waits for the loads whose results it spills), that whole tail goes behind the
# restore as it stands — which is the same as hoisting the restore (a scalar instruction that reads an SGPR pair none of them
# writes) in front of it.  Anything else — a reload, a rematerialised constant (a then-branch's phi copy looks the same), a
# dependence — FAILS the build with the snippet.
EXEC_WRITE = re.compile(r"^(s_\w+\s+exec\b|s_\w+_saveexec_b64\b|v_cmpx_)")
WIDEN = re.compile(r"^(s_or_b64 exec, exec, s\[\d+:\d+\]|s_mov_b64 exec, s\[\d+:\d+\]|s_or_saveexec_b64 s\[\d+:\d+\], s\[\d+:\d+\])")
EXEC_FREE = re.compile(r"^(s_|v_readlane_b32|v_writelane_b32|;|\.)")
BLOCK_END = re.compile(r"^s_(cbranch|branch|endpgm|setpc)")
LABEL = re.compile(r"^([.\w$]+):")
SPILL_STORE = re.compile(r"^scratch_store_dword(x[234])?\b.*;\s*\d+-byte Folded Spill")
RA_MADE = re.compile(r"^(scratch_(store|load)_dword(x[234])?\b.*;\s*\d+-byte Folded (Spill|Reload)|"
                     r"v_mov_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|v_mov_b64_e32 v\[\d+:\d+\], (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|"
                     r"v_bfrev_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9]+)\s*$)")
CONST_MOVE = re.compile(r"^(v_mov_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|v_mov_b64_e32 v\[\d+:\d+\], (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|"
                        r"v_bfrev_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9]+)\s*$)")
# offenders of rule (A) accepted as they stand, by opcode, each with its reason (none today)
WHITELIST = {}


def vgprs_of(t):
    """VGPR numbers an instruction names (v7, v[4:7])"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", t.split(";")[0]):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Asm:
    """the lines of a device assembly file, which of them lie inside hand-written asm regions, and the join labels"""

    def __init__(self, text):
        self.L = text.split("\n")
        self.in_asm = []
        inside = False
        self.execz_targets = set()
        for l in self.L:
            u = l.strip()
            if u.startswith(";;#ASMSTART"):
                inside = True
            self.in_asm.append(inside)
            if u.startswith(";;#ASMEND"):
                inside = False
            m = re.match(r"^s_cbranch_execz\s+([.\w$]+)", u)
            if m:
                self.execz_targets.add(m.group(1))

    def offenders(self, i):
        """Line i widens EXEC (outside asm regions): the indices of the instructions in front of it that run under the
        wrong mask by rule (A) or (B), else []."""
        L = self.L
        dep = []
        k = i - 1
        top = None
        # (`s_mov_b64 exec, s[a:b]` with s[a:b] formed by a scalar instruction of this very block — the expanded form of
        #  s_and_saveexec_b64 when the saved mask is spilled — narrows EXEC: the start of an if, not a join)
        msrc = re.match(r"^s_mov_b64 exec, (s\[\d+:\d+\])", L[i].strip())
        while k >= 0:
            u = L[k].strip()
            if msrc and re.match(r"^s_\w+\s+" + re.escape(msrc.group(1)) + r"\s*,", u):
                return []
            m = LABEL.match(u)
            if m and not self.in_asm[k]:
                top = m.group(1)
                break
            if u.startswith("; %bb."):
                top = ""
                break
            if BLOCK_END.match(u) or u.startswith(";;#ASMEND") or EXEC_WRITE.match(u):
                return []  # behind a branch / behind hand-written asm / not the block's first EXEC write
            if u and not EXEC_FREE.match(u):
                dep.append(k)
            k -= 1
        if top is None or not dep:
            return []
        if top in self.execz_targets:  # rule (A)
            return sorted(k for k in dep if L[k].strip().split()[0] not in WHITELIST)
        if all(RA_MADE.match(L[k].strip()) for k in dep) and any("Folded" in L[k] for k in dep):  # rule (B)
            return sorted(dep)
        return []

    def block_top(self, i):
        k = i - 1
        while k >= 0 and not ((LABEL.match(self.L[k].strip()) and not self.in_asm[k]) or self.L[k].strip().startswith("; %bb.")):
            k -= 1
        return k

    def why_not_movable(self, i, off):
        """None when the offenders in front of the restore at line i may simply move behind it, else the reason why not."""
        L = self.L
        moved = set()
        # (at the target of an s_cbranch_execz — rule (A) — nothing in front of the restore can be a then-branch's own work, so a
        #  rematerialised constant there is the allocator's too and moves with the stores; elsewhere it could be a phi copy: refused)
        m = LABEL.match(L[self.block_top(i)].strip())
        at_join = bool(m) and m.group(1) in self.execz_targets
        for k in off:
            u = L[k].strip()
            if not SPILL_STORE.match(u) and not (at_join and CONST_MOVE.match(u)):
                return f"not a VGPR spill store: {u}"
            moved |= vgprs_of(u)
        stores = [k for k in off if SPILL_STORE.match(L[k].strip())]
        tail = self.tail_with_waits(i, off)
        for k in range(off[0] + 1, i):  # what stays between the first moved instruction and the restore
            if k in off or k in tail:
                continue
            u = L[k].strip()
            if not u or u.startswith(";") or u.startswith("."):
                continue
            if stores and k > stores[0] and re.match(r"^s_waitcnt\b.*vmcnt", u):
                return f"an s_waitcnt on the vector-memory counter stays between a moved store and the restore: {u}"
            if vgprs_of(u) & moved:
                return f"an instruction that stays between a moved store and the restore names a moved register: {u}"
        return None

    def tail_with_waits(self, i, off):
        """The s_waitcnt / s_nop lines that move WITH the offenders: all of them behind the first moved store, when nothing else
        stands between that store and the restore (the tail then moves as one piece, its order unchanged); else none."""
        L = self.L
        stores = [k for k in off if SPILL_STORE.match(L[k].strip())]
        if not stores:
            return []
        extra = []
        for k in range(stores[0] + 1, i):
            if k in off:
                continue
            u = L[k].strip()
            if not u or u.startswith(";") or u.startswith("."):
                continue
            if re.match(r"^(s_waitcnt|s_nop)\b", u):
                extra.append(k)
            else:
                return []
        return extra

    def snippet(self, i):
        return "\n".join("    " + self.L[k] for k in range(max(self.block_top(i), i - 24), i + 1))


class ExecPrologueError(Exception):
    pass


def repair(text):
    A = Asm(text)
    L = A.L
    moved = []
    i = 0
    while i < len(L):
        if WIDEN.match(L[i].strip()) and not A.in_asm[i]:
            off = A.offenders(i)
            if off:
                why = A.why_not_movable(i, off)
                if why:
                    raise ExecPrologueError(f"EXEC-dependent instructions in front of a join block's EXEC restore (line {i + 1}) that the "
                                            f"build will not move on its own — {why}\n{A.snippet(i)}")
                off = sorted(set(off) | set(A.tail_with_waits(i, off)))
                ins = [L[k] for k in off]
                for k in reversed(off):
                    del L[k]
                    del A.in_asm[k]
                i -= len(off)
                for n, l in enumerate(ins):
                    L.insert(i + 1 + n, l + "  ; moved behind the EXEC restore (hipcc_via_asm.py)")
                    A.in_asm.insert(i + 1 + n, False)
                moved += [l.strip() for l in ins]
                i += len(ins)
        i += 1
    return "\n".join(L), moved


def lint(text):
    A = Asm(text)
    bad = []
    for i, l in enumerate(A.L):
        if WIDEN.match(l.strip()) and not A.in_asm[i]:
            bad += [(k + 1, A.L[k].strip()) for k in A.offenders(i)]
    return bad


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        sys.exit(r.returncode)
    out = "\n".join(l for l in r.stdout.split("\n") if l and "argument unused during compilation" not in l)
    if out:
        sys.stderr.write(out + "\n")


def main():
    a = sys.argv[1:]
    keep = None
    norepair = False
    while a and a[0].startswith("--"):
        if a[0] == "--keep-asm":
            keep = a[1]
            a = a[2:]
        elif a[0] == "--no-repair":  # experiments only: the assembly as the compiler made it
            norepair = True
            a = a[1:]
        else:
            break
    out, src, flags = a[0], a[1], a[2:]
    base = os.path.splitext(os.path.basename(out))[0]
    with tempfile.TemporaryDirectory() as td:
        s0 = os.path.join(td, base + ".s")
        run([HIPCC] + flags + ["--cuda-device-only", "-S", "-o", s0, src])
        text = open(s0).read()
        moved = []
        if not norepair:
            try:
                text, moved = repair(text)
            except ExecPrologueError as e:
                sys.stderr.write(f"hipcc_via_asm: {src}: {e}\n")
                sys.exit(1)
            left = lint(text)
            if left:
                sys.stderr.write(f"hipcc_via_asm: {src}: EXEC-dependent instructions still in front of an EXEC restore:\n")
                for ln, t in left:
                    sys.stderr.write(f"  line {ln}: {t}\n")
                sys.exit(1)
        s1 = os.path.join(td, base + ".fixed.s")
        open(s1, "w").write(text)
        if keep:
            os.makedirs(keep, exist_ok=True)
            open(os.path.join(keep, base + ".s"), "w").write(text)
            open(os.path.join(keep, base + ".repairs.txt"), "w").write(
                f"{len(moved)} instruction(s) moved behind an EXEC restore\n" + "\n".join(moved) + ("\n" if moved else ""))
        if moved:
            sys.stderr.write(f"hipcc_via_asm: {src}: {len(moved)} instruction(s) moved behind an EXEC restore: {'; '.join(moved)}\n")
        o = os.path.join(td, base + ".dev.o")
        run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", s1, "-o", o])
        hs = os.path.join(td, base + ".hsaco")
        run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hs, o])
        fb = os.path.join(td, base + ".hipfb")
        run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={hs}", f"-output={fb}"])
        run([HIPCC] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", out])


if __name__ == "__main__":
    main()
