"""The sorted-window insert of the tile kernel is ONE generated asm chain (gaussian-ray-tracing_amd/csrc/gen_slots.py:
insert_macro_chain).  Its instruction list is interpreted here lane by lane — EXEC masks, SGPR compare masks, branches —
and compared with a plain sorted insert on random windows, partial EXEC included; and the checked-in grt_slots_gen.inc
must be what the generator prints."""
import contextlib
import importlib.util
import io
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-ray-tracing_amd", "csrc")
INV = (1 << 64) - 1


def _gen():
    spec = importlib.util.spec_from_file_location("gen_slots", os.path.join(CSRC, "gen_slots.py"))
    g = importlib.util.module_from_spec(spec)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        spec.loader.exec_module(g)
    return g, out.getvalue()


def _run(L, ks, keys, nk, active):
    n = len(nk)
    R = {f"k{i}": [keys[l][i] for l in range(n)] for i in range(ks)}
    R["nk"] = list(nk)
    S, exec_, vcc, scc = {}, set(active), set(), 0
    labels = {l[:-1]: i for i, l in enumerate(L) if l.endswith(":")}
    nm = lambda x: x.strip("%[]")
    val = lambda x, l: INV if x == "-1" else R[nm(x)][l]
    pc = 0
    while pc < len(L):
        ins = L[pc]; pc += 1
        if ins.endswith(":"):
            continue
        op, rest = ins.split(" ", 1)
        a = [x.strip() for x in rest.split(",")]
        if op == "s_mov_b64":
            src = exec_ if a[1] == "exec" else S[nm(a[1])]
            if a[0] == "exec": exec_ = set(src)
            else: S[nm(a[0])] = set(src)
        elif op == "v_cmp_ne_u64":
            vcc = {l for l in exec_ if val(a[1], l) != val(a[2], l)}
        elif op == "v_cmp_lt_u64":
            S[nm(a[0])] = {l for l in exec_ if val(a[1], l) < val(a[2], l)}
        elif op == "v_mov_b64":
            src = [val(a[1], l) for l in range(n)]
            for l in exec_: R[nm(a[0])][l] = src[l]
        elif op == "s_cbranch_vccnz":
            if vcc: pc = labels[a[0]]
        elif op == "s_branch":
            pc = labels[a[0]]
        elif op == "s_cmp_eq_u64":
            scc = 0 if S[nm(a[0])] else 1
        elif op == "s_cbranch_scc1":
            if scc: pc = labels[a[0]]
        else:
            raise AssertionError("instruction the interpreter does not know: " + ins)
    assert exec_ == set(active), "EXEC not restored"
    return [[R[f"k{i}"][l] for i in range(ks)] for l in range(n)]


def _ref(ks, key, nk):
    if nk == INV:
        return list(key)
    k = [x for x in key if x != INV]
    k.insert(sum(1 for x in k if x <= nk), nk)  # behind equal keys; the largest falls off a full window
    k = k[:ks]
    return k + [INV] * (ks - len(k))


def test_insert_chain_is_a_sorted_insert():
    g, _ = _gen()
    rng = random.Random(1)
    for ks in (8, 12):
        L = [l for l in g.insert_macro_chain(ks) if not l.startswith("s_mov_b64 %[m4], %[m")]
        for trial in range(1500):
            lanes = 8
            maxfill = rng.choice([0, 1, 3, 4, 5, 7, 8, ks - 1, ks])
            keys = []
            for _ in range(lanes):
                m = rng.randint(0, maxfill)
                keys.append(sorted(rng.randint(0, 40) for _ in range(m)) + [INV] * (ks - m))
            nk = [rng.choice([INV, rng.randint(0, 41)]) for _ in range(lanes)]
            active = {l for l in range(lanes) if rng.random() < 0.8}
            out = _run(L, ks, keys, nk, active)
            for l in range(lanes):
                want = _ref(ks, keys[l], nk[l]) if l in active else keys[l]
                assert out[l] == want, (ks, trial, l, keys[l], nk[l], out[l], want)


def test_checked_in_macros_are_the_generators_output():
    _, printed = _gen()
    assert printed == open(os.path.join(CSRC, "grt_slots_gen.inc")).read()
