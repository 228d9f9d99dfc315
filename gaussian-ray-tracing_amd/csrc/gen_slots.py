#!/usr/bin/env python3
"""Generates grt_slots_gen.inc: the per-lane sorted key window of the streaming kernel as inline gfx950 assembly.

A slot move under a per-lane condition costs two v_cndmask_b32 in compiler output (64-bit keys).  Here the
condition goes into EXEC instead and every move is ONE v_mov_b64, which halves the VALU work of the window
(the kernel is VALU-bound).  Each asm block saves EXEC, narrows it, and restores it before it ends, so the
compiler never sees a modified EXEC.  The window is cut into blocks of 4 (8 / 12 slots) or 8 (32 slots) keys; a block
is skipped, by a wave-uniform branch, when no lane has anything at or above it (windows are mostly part-filled).

  SLOT_SHIFT_ALL(MASK)   lanes in MASK: k0 <- k1 <- ... <- k[KS-1] <- ~0   (pop the smallest key)
  SLOT_INSERT(KEY)       sorted insert of KEY into k0..k[KS-1] (ascending, ~0 = free; the largest key falls off
                         when the window is full; KEY == ~0 is a no-op).  Top-down: with c_i = KEY < k_i,
                         lanes with c_{i-1} take k_{i-1}, lanes with c_i & ~c_{i-1} take KEY; the walk stops
                         (wave-uniformly, between blocks) once every lane has placed its key.

Run:  python3 gen_slots.py > grt_slots_gen.inc
"""
def chunk(ks):
    return 4 if ks <= 12 else 8

def shift_macro(ks):
    ch = chunk(ks)
    blocks = []
    lo = 0
    while lo < ks:
        hi = min(lo + ch, ks) - 1
        lines = ['"s_and_saveexec_b64 %[sv], %[m]\\n\\t"']
        for i in range(lo, hi + 1):
            src = f'%[k{i+1}]' if i + 1 < ks else '-1'
            lines.append(f'"v_mov_b64 %[k{i}], {src}\\n\\t"')
        lines.append('"s_mov_b64 exec, %[sv]"')
        outs = ', '.join(f'[k{i}] "+v"(k{i})' for i in range(lo, hi + 1)) + ', [sv] "=&s"(sv_)'
        ins = '[m] "s"(m_)' + (f', [k{hi+1}] "v"(k{hi+1})' if hi + 1 < ks else '')
        guard = '' if lo == 0 else f'if (wave_any(k{lo} != kKeyInvalid)) /* nothing above an empty slot */ '
        # (s_and_saveexec_b64 writes SCC: the compiler must not keep a compare result alive across the block)
        blocks.append('        ' + guard + 'asm volatile(' + ' '.join(lines) + '\n                     : ' + outs + '\n                     : ' + ins + '\n                     : "scc");')
        lo = hi + 1
    body = '\n'.join(blocks)
    return ('#define SLOT_SHIFT_ALL(MASK)  \\\n    {  \\\n        const uint64_t m_ = (MASK);  \\\n        uint64_t sv_;  \\\n'
            + '  \\\n'.join(body.split('\n')) + '  \\\n    }\n')

def insert_macro(ks):
    """Per block: all the compares first (KEY against the ORIGINAL keys k[hi] .. k[lo-1], independent VALU ops whose
    SGPR results are ready by the time the moves need them), then the EXEC-masked moves."""
    ch = chunk(ks)
    blocks = []
    hi = ks - 1
    while hi >= 0:
        lo = max(hi - ch + 1, 0)
        idx = list(range(hi, lo - 2, -1)) if lo > 0 else list(range(hi, -1, -1))  # masks c_i for these i
        lines = ['"s_mov_b64 %[sv], exec\\n\\t"']
        for i in idx:
            lines.append(f'"v_cmp_lt_u64 %[c{i}], %[nk], %[k{i}]\\n\\t"')
        for i in range(hi, lo - 1, -1):
            if i == 0:
                lines.append('"s_and_b64 exec, %[sv], %[c0]\\n\\t"')
                lines.append('"v_mov_b64 %[k0], %[nk]\\n\\t"')
            else:
                lines.append(f'"s_and_b64 exec, %[sv], %[c{i-1}]\\n\\t"')
                lines.append(f'"v_mov_b64 %[k{i}], %[k{i-1}]\\n\\t"')
                lines.append(f'"s_andn2_b64 %[c{i}], %[c{i}], %[c{i-1}]\\n\\t"')
                lines.append(f'"s_and_b64 exec, %[sv], %[c{i}]\\n\\t"')
                lines.append(f'"v_mov_b64 %[k{i}], %[nk]\\n\\t"')
        if lo > 0:
            lines.append(f'"s_mov_b64 %[go], %[c{lo-1}]\\n\\t"')  # the carry (KEY < k[lo-1]) leaves the block
        lines.append('"s_mov_b64 exec, %[sv]"')
        outs = ', '.join(f'[k{i}] "+v"(k{i})' for i in range(lo, hi + 1))
        outs += ', [sv] "=&s"(sv_), ' + ', '.join(f'[c{i}] "=&s"(m{i - (lo - 1 if lo > 0 else 0)}_)' for i in idx)
        if lo > 0:
            outs += ', [go] "=&s"(go_)'
        ins = '[nk] "v"(nk_)' + (f', [k{lo-1}] "v"(k{lo-1})' if lo > 0 else '')
        guard = 'if (go_ != 0ull) ' if lo == 0 else f'if (go_ != 0ull && wave_any(k{lo-1} != kKeyInvalid)) '
        if hi == ks - 1:
            guard = '' if lo == 0 else f'if (wave_any(k{lo-1} != kKeyInvalid)) '
        blocks.append('        ' + guard + 'asm volatile(' + ' '.join(lines) + '\n                     : ' + outs + '\n                     : ' + ins + '\n                     : "scc");')
        hi = lo - 1
    body = '\n'.join(blocks)
    nm = chunk(ks) + 1
    decl = ', '.join(f'm{j}_' for j in range(nm))
    return ('#define SLOT_INSERT(KEY)  \\\n    {  \\\n        const uint64_t nk_ = (KEY);  \\\n        uint64_t sv_, go_ = ~0ull, ' + decl + ';  \\\n'
            + '  \\\n'.join(body.split('\n')) + '  \\\n        ' + ' '.join(f'(void)m{j}_;' for j in range(nm)) + '  \\\n    }\n')

def insert_macro_chain(ks):
    """ONE asm statement for the whole window (8 / 12 slots), no compiler-visible control flow:

        exec = c[ks-1]:  k[ks-1] = KEY
        exec = c[i]:     k[i+1] = k[i] ; k[i] = KEY          for i = ks-2 .. 0      (c[i] is a subset of c[i+1])

    one EXEC write and two v_mov_b64 per slot (the block form above pays three SALU per slot and the compiler's branch
    logic between blocks; the frame is its instruction count, profiles/r03_sensitivity.json).  The compares of a block of
    four run under the EXEC of the block above's carry: lanes outside it would compare false anyway.  Entry points: the
    highest block some lane occupies (k[4b-1] != ~0 somewhere); exits: between blocks once no lane carries on."""
    assert ks % 4 == 0 and ks <= 12
    nb = ks // 4
    L = []
    A = L.append
    A('s_mov_b64 %[sv], exec')
    # entry: from the top block down, the first block b whose lower neighbour slot k[4b-1] is occupied in some lane
    for b in range(nb - 1, 0, -1):
        A(f'v_cmp_ne_u64 vcc, -1, %[k{4*b-1}]')
        A(f's_cbranch_vccnz .Lgrt_e{b}_%=')
    A('s_branch .Lgrt_e0_%=')
    def cm(i):  # mask register of c_i: the carry of a block keeps m4 (block boundary 4b-1), others m0..m3 by position
        return 'm4' if i % 4 == 3 and i != ks - 1 else f'm{3 - (i % 4)}' if i % 4 != 3 else 'm0'
    for b in range(nb - 1, -1, -1):
        hi, lo = 4 * b + 3, 4 * b
        top = (b == nb - 1)
        # masks: c_hi (top block only: m0 .. ), carry from above otherwise
        A(f'.Lgrt_e{b}_%=:')
        if top:
            names = {hi: 'm0', hi - 1: 'm1', hi - 2: 'm2', hi - 3: 'm3'}
            if lo > 0: names[lo - 1] = 'm4'
            for i in sorted(names, reverse=True):
                A(f'v_cmp_lt_u64 %[{names[i]}], %[nk], %[k{i}]')
            A(f's_mov_b64 exec, %[{names[hi]}]')
            A(f'v_mov_b64 %[k{hi}], %[nk]')
            carry_in = None
        else:
            # entered from the label: slots above are empty everywhere, the block starts with its own top compare
            A(f'v_cmp_lt_u64 %[m4], %[nk], %[k{hi}]')
            A(f's_mov_b64 exec, %[m4]')
            A(f'.Lgrt_c{b}_%=:')  # entered by falling through from the block above: exec = carry already, k[hi+1] = k[hi] done
            A(f'v_mov_b64 %[k{hi}], %[nk]')
            names = {hi - 1: 'm0', hi - 2: 'm1', hi - 3: 'm2'}
            if lo > 0: names[lo - 1] = 'm3'
            for i in sorted(names, reverse=True):
                A(f'v_cmp_lt_u64 %[{names[i]}], %[nk], %[k{i}]')
        for i in range(hi - 1, lo - 1, -1):
            A(f's_mov_b64 exec, %[{names[i]}]')
            A(f'v_mov_b64 %[k{i+1}], %[k{i}]')
            A(f'v_mov_b64 %[k{i}], %[nk]')
        if lo > 0:
            cn = names[lo - 1]
            A(f's_cmp_eq_u64 %[{cn}], 0')
            A(f's_cbranch_scc1 .Lgrt_x_%=')
            A(f's_mov_b64 exec, %[{cn}]')
            A(f'v_mov_b64 %[k{lo}], %[k{lo-1}]')
            if cn != 'm4':
                A(f's_mov_b64 %[m4], %[{cn}]')  # (not reached: the carry register of a lower block is m3 only when lo-1 >= 0)
            A(f's_branch .Lgrt_c{b-1}_%=')
    A('.Lgrt_x_%=:')
    A('s_mov_b64 exec, %[sv]')
    return L

def insert_macro_v2(ks):
    L = insert_macro_chain(ks)
    # the fall-through entry of a lower block expects ITS compares keyed m0..m2 (+m3 carry); the carry that led there may
    # sit in m4 (top block) or m3: both are only read before the block's own compares overwrite m0..m3, so no move is needed
    L = [l for l in L if not l.startswith('s_mov_b64 %[m4], %[m')]
    body = ' '.join('"' + l + '\\n\\t"' for l in L[:-1]) + ' "' + L[-1] + '"'
    outs = ', '.join(f'[k{i}] "+v"(k{i})' for i in range(ks)) + ', [sv] "=&s"(sv_), ' + ', '.join(f'[m{j}] "=&s"(m{j}_)' for j in range(5))
    return ('#define SLOT_INSERT(KEY)  \\\n    {  \\\n        const uint64_t nk_ = (KEY);  \\\n        uint64_t sv_, m0_, m1_, m2_, m3_, m4_;  \\\n'
            '        asm volatile(' + body + '  \\\n                     : ' + outs + '  \\\n                     : [nk] "v"(nk_)  \\\n                     : "vcc", "scc");  \\\n'
            '        (void)m0_; (void)m1_; (void)m2_; (void)m3_; (void)m4_;  \\\n    }\n')

print('// GENERATED by gen_slots.py — do not edit; see that file for what these macros do and why.')
for ks in (8, 12, 32):
    print(f'#if GRT_KS == {ks}')
    print(shift_macro(ks))
    print(insert_macro(ks) if ks > 12 else insert_macro_v2(ks))
    print('#endif')
