#!/usr/bin/env python3
"""spill_lanes.py <file.s> [substring] — per kernel: which v_readlane / v_writelane are SGPR-spill traffic (they move to / from a
VGPR that the kernel also WRITES lanes of: the allocator's spill VGPRs) and which are the source's own cross-lane reads; how many of
each sit at which loop depth (number of backward-branch ranges covering the line)."""
import re, sys, collections

def kernels(lines):
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m: cur = (m.group(1), i)
        elif cur and l.strip().startswith("s_endpgm"):
            yield cur[0], cur[1], i
            cur = None

def loops(body):
    """set of line indices inside some backward-branch range"""
    lab = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m: lab[m.group(1)] = i
    inl = [0] * (len(body) + 1)
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] <= i:
            inl[lab[m.group(1)]] += 1; inl[i + 1] -= 1
    out, d = {}, 0
    for i in range(len(body)):
        d += inl[i]
        out[i] = d  # number of backward-branch ranges that cover the line (~ loop depth)
    return out

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name, i0, i1 in kernels(src):
    if pat not in name: continue
    body = src[i0:i1]
    inl = loops(body)
    spv = set()
    for l in body:
        m = re.match(r"\s+v_writelane_b32 (v\d+)", l)
        if m: spv.add(m.group(1))
    c = collections.Counter()
    for i, l in enumerate(body):
        m = re.match(r"\s+v_readlane_b32 s\d+, (v\d+)", l)
        if m: c[("reload" if m.group(1) in spv else "source readlane", min(inl[i], 4))] += 1
        elif re.match(r"\s+v_writelane_b32", l): c[("spill", min(inl[i], 4))] += 1
    print(name, "spill VGPRs", sorted(spv))
    for k in sorted(c): print("   ", k[0], "covered by %d%s backward branches" % (k[1], "+" if k[1] == 4 else ""), c[k])
