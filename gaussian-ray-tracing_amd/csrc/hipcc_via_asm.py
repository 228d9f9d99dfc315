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
  2. exec-prologue repair                   where everything EXEC-dependent between the start of a machine basic block and
                                            the `s_or_b64 exec, exec, s[..]` in it is of the register allocator's own
                                            making (spill store / reload, rematerialised constant), those instructions
                                            move BEHIND the restore (they were meant for the join block: all lanes);
                                            then the same rule, as a lint, must find nothing
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

EXEC_FREE = re.compile(r"^(s_|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32|;|\.)")
BLOCK_END = re.compile(r"^s_(cbranch|branch|endpgm|setpc)")
RESTORE = re.compile(r"^s_or_b64 exec, exec, s\[\d+:\d+\]")
LABEL = re.compile(r"^[.\w$]+:")
# what the register allocator inserts on its own: spill stores / reloads (the assembly printer marks them) and
# rematerialised constants
RA_MADE = re.compile(r"^(scratch_(store|load)_dword(x[234])?\b.*;\s*\d+-byte Folded (Spill|Reload)|"
                     r"v_mov_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|v_mov_b64_e32 v\[\d+:\d+\], (0x[0-9a-fA-F]+|-?[0-9.]+)\s*$|"
                     r"v_bfrev_b32_e32 v\d+, (0x[0-9a-fA-F]+|-?[0-9]+)\s*$)")


def block_prologue_offenders(L, i):
    """Line i is an EXEC restore.  When everything EXEC-dependent between the start of its machine basic block and i is of
    the register allocator's own making (spill store / reload, rematerialised constant), the block is a join block and
    those instructions were meant for ALL its lanes: their indices.  (A block whose front holds ordinary vector work is a
    then-branch that block placement merged with its join block: that work belongs under the partial mask.)"""
    out = []
    k = i - 1
    while k >= 0:
        u = L[k].strip()
        if LABEL.match(u) or u.startswith("; %bb.") or BLOCK_END.match(u):
            break
        if u.startswith(";;#ASMEND"):
            break  # hand-written asm manages EXEC itself
        if u and not EXEC_FREE.match(u):
            if not RA_MADE.match(u):
                return []
            out.append(k)
        k -= 1
    return sorted(out) if any("Folded" in L[k] for k in out) else []


def repair(text):
    L = text.split("\n")
    moved = []
    i = 0
    while i < len(L):
        if RESTORE.match(L[i].strip()):
            off = block_prologue_offenders(L, i)
            if off:
                ins = [L[k] for k in off]
                for k in reversed(off):
                    del L[k]
                i -= len(off)
                for n, l in enumerate(ins):
                    L.insert(i + 1 + n, l + "  ; moved behind the EXEC restore (hipcc_via_asm.py)")
                moved += [l.strip() for l in ins]
                i += len(ins)
        i += 1
    return "\n".join(L), moved


def lint(text):
    L = text.split("\n")
    bad = []
    for i, l in enumerate(L):
        if RESTORE.match(l.strip()):
            bad += [(k + 1, L[k].strip()) for k in block_prologue_offenders(L, i)]
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
            text, moved = repair(text)
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
