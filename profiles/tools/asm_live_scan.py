#!/usr/bin/env python3
"""asm_live_scan.py <file.s> [...] — does a value the compiler keeps in SCC / VCC survive an inline-asm block that
overwrites it without saying so?

Reads the device assembly of a translation unit (`hipcc -S --cuda-device-only`), where every inline-asm statement sits
between `;;#ASMSTART` and `;;#ASMEND`.  For every kernel and every asm block it collects the implicit registers the block WRITES
(SCC: s_and* / s_or* / s_andn2* / s_cmp* / s_add* / s_bitset…; VCC: v_cmp* without an SGPR destination, or a `vcc`
destination; EXEC is always saved and restored by the generated blocks and is checked by tests/test_slots_gen.py) and then
walks forward from the block's end, along the fall-through path and through unconditional branches, until the register
is written again: a READ on the way (s_cbranch_scc*, s_cselect*, s_addc*, s_subb*, s_cmov*; s_cbranch_vcc*, v_cndmask
with an implicit vcc, …) means the compiler's value crossed the block — legal only if the block does not clobber it.

Used in round 4 to settle round 3's open failure (profiles/r04_experiments_log.md): the tile kernel built WITHOUT the
"scc" clobber on the generated s_and_saveexec_b64 blocks keeps compare results in SCC across them.
Prints one line per finding and a summary; exit code 1 when something crossed."""
import re
import sys

SCC_WRITE = re.compile(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|not|add|sub|addc|subb|min|max|lshl|lshr|ashr|bfe|bfm|abs|absdiff|"
                       r"cmp|cmpk|bitcmp|wqm|quadmask|bcnt|ff|flbit|mul_hi|lshl[1-4]_add|and_saveexec|or_saveexec|xor_saveexec|"
                       r"andn2_saveexec|orn2_saveexec|nand_saveexec|nor_saveexec|xnor_saveexec|andn1_saveexec|orn1_saveexec|"
                       r"andn1_wrexec|andn2_wrexec)")
SCC_NOWRITE = re.compile(r"^s_(mul_i32|mulk|movk|mov|cmov|cmovk|cselect|bitset|bitreplicate|brev|sext|getpc|setpc|swappc|"
                         r"load|store|buffer|scratch|dcache|waitcnt|nop|sleep|branch|cbranch|barrier|endpgm|setprio|sendmsg|"
                         r"movrel|getreg|setreg|memtime|memrealtime|atc|ff[01]_i32|flbit_i32)")
SCC_READ = re.compile(r"^s_(cbranch_scc[01]|cselect_b(32|64)|addc_u32|subb_u32|cmov_b(32|64)|cmovk_i32)")
VCC_READ = re.compile(r"^(s_cbranch_vcc(n?z)|v_(cndmask_b32(_e32)?|addc_co_u32(_e32)?|subb_co_u32(_e32)?|subbrev_co_u32(_e32)?|div_fmas_f32|div_fmas_f64))\b")


def writes_scc(op):
    if SCC_NOWRITE.match(op):
        return False
    return bool(SCC_WRITE.match(op))


def writes_vcc(t):
    op = t.split()[0]
    ops = t[len(op):]
    if "vcc" in ops.split(",")[0]:
        return True
    return op.startswith("v_cmp") and op.endswith("_e32")


def reads_vcc(t):
    op = t.split()[0]
    if VCC_READ.match(op):
        return True
    ops = [x.strip() for x in t[len(op):].split(",")]
    return any(o.startswith("vcc") for o in ops[1:])


def kernels(lines):
    """(name, first, last) of every function body"""
    out = []
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if m:
            cur = (m.group(1), i)
        elif cur and l.strip().startswith("s_endpgm"):
            out.append((cur[0], cur[1], i))
            cur = None
    return out


def scan(path):
    L = open(path).read().split("\n")
    findings = []
    n_blocks = 0
    for name, s, e in kernels(L):
        labels = {L[i].strip()[:-1]: i for i in range(s, e) if re.match(r"^[.\w$]+:", L[i].strip())}
        i = s
        while i < e:
            if L[i].strip() != ";;#ASMSTART":
                i += 1
                continue
            j = i + 1
            w_scc = w_vcc = False
            while L[j].strip() != ";;#ASMEND":
                t = L[j].strip()
                if t and not t.startswith(";") and not t.endswith(":") and not t.startswith("."):
                    op = t.split()[0]
                    w_scc |= writes_scc(op)
                    w_vcc |= writes_vcc(t)
                j += 1
            n_blocks += 1
            for reg, wrote in (("scc", w_scc), ("vcc", w_vcc)):
                if not wrote:
                    continue
                k, hops = j + 1, 0
                while k < e and hops < 400:
                    t = L[k].strip()
                    k += 1
                    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                        continue
                    hops += 1
                    op = t.split()[0]
                    if reg == "scc":
                        if SCC_READ.match(op):
                            findings.append((name, i + 1, k, reg, t))
                            break
                        if writes_scc(op):
                            break
                    else:
                        if reads_vcc(t):
                            findings.append((name, i + 1, k, reg, t))
                            break
                        if writes_vcc(t):
                            break
                    if op == "s_branch":
                        tgt = t.split()[1]
                        if tgt in labels:
                            k = labels[tgt]
                        else:
                            break
                    if op in ("s_endpgm", "s_setpc_b64"):
                        break
            i = j + 1
    return n_blocks, findings


def main():
    bad = 0
    for p in sys.argv[1:]:
        n, f = scan(p)
        print(f"{p}: {n} inline-asm blocks, {len(f)} implicit register(s) live across a block that overwrites them")
        for name, a, b, reg, t in f:
            print(f"  {reg} written by the asm block at line {a}, read at line {b}: {t}   [{name[:60]}]")
        bad += len(f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
