#!/usr/bin/env python3
"""Where a gfx950 .s file (hipcc -save-temps) touches scratch: per kernel, every scratch_load / scratch_store with the loops
(backward branches) that contain it.  A reload inside a hot loop is followed by s_waitcnt vmcnt(0) -- it waits for EVERY vector
memory operation in flight, prefetches included (that was 0.035 ms of the training forward: a spilled store address).
usage: isa_scratch.py file.s [kernel-substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and "; @_Z" in l]
for i in starts:
    name = lines[i].split(":")[0]
    if want not in name:
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm"))
    body = lines[i:end]
    lab = {l.split(":")[0]: j for j, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
    loops = []
    for j, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < j:
            loops.append((lab[m.group(1)], j))
    sc = [(j, l.strip().split(";")[0].strip()) for j, l in enumerate(body) if "scratch_" in l]
    if not sc:
        continue
    mf = sum("v_mfma" in l for l in body)
    print(f"{name[:90]}: {len(body)} lines, {mf} MFMAs, {len(sc)} scratch instructions")
    for j, t in sc:
        inside = sorted({(a, b) for a, b in loops if a <= j <= b}, key=lambda ab: ab[1] - ab[0])
        inner = inside[0] if inside else None
        n_mfma = sum("v_mfma" in l for l in body[inner[0]:inner[1]]) if inner else 0
        print(f"   line {j:5d}  {t:48s} innermost loop {inner} ({n_mfma} MFMAs in it)" if inner else f"   line {j:5d}  {t:48s} (no loop)")
