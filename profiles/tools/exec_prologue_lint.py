#!/usr/bin/env python3
"""exec_prologue_lint.py <file.s> [...] — spill code that the register allocator put in FRONT of a join block's EXEC restore.

A divergent `if` is `s_and_saveexec_b64 sN, cond ... s_or_b64 exec, exec, sN`; the restore opens the join block, and
everything the join block does is meant for ALL the lanes that entered the `if`.  The register allocator of this
toolchain (ROCm 7.2 LLVM) sometimes places what it inserts at the top of such a block — a VGPR spill store, a reload, a
rematerialised constant — BEFORE that restore, i.e. under the then-branch's partial mask: the lanes that skipped the
branch keep a stale scratch slot / register.  That is what made round 3's `GRT_FIT_APPROX=7` variant of the tile kernel
report an expired watchdog and absurd counters in its instrumented instantiation (profiles/r04_experiments_log.md): a
loop-carried per-lane counter was spilled by `scratch_store_dword` one instruction before `s_or_b64 exec, exec, s[12:13]`
and reloaded under the full mask.  Nothing in the source can express or prevent it, so the build is checked instead:
the rule lives in gaussian-ray-tracing_amd/csrc/hipcc_via_asm.py (which also repairs what it finds on the way to the
object file); this script applies it to any device assembly (`hipcc -S --cuda-device-only`).
Exit code 1 when something is found."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gaussian-ray-tracing_amd", "csrc"))
import hipcc_via_asm as V  # noqa: E402


def main():
    bad = 0
    for p in sys.argv[1:]:
        text = open(p).read()
        f = V.lint(text)
        n = sum(1 for l in text.split("\n") if V.WIDEN.match(l.strip()))
        print(f"{p}: {n} EXEC restores, {len(f)} register-allocator-made instruction(s) in front of one inside its join block")
        L = text.split("\n")
        for ln, t in f:
            k = ln - 1
            while k > 0 and not L[k].startswith("_Z"):
                k -= 1
            print(f"  line {ln}: {t}   [{L[k][:72]}]")
        bad += len(f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
