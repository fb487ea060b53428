#!/usr/bin/env python
"""Finds the loop shape that made block 0 of ppo_reduce_kernel the long pole of its launch (round 4): an inner loop whose body issues
global / buffer loads and waits for ALL of them (s_waitcnt vmcnt(0)) before the backward branch, i.e. one memory round trip per
iteration (`for (...) v += p[i]` compiles to exactly that).  usage: serial_load_loops.py file.s [max loads per iteration = 2]
Prints kernel, loop label, loads per iteration, instructions per iteration."""
import re
import sys

path = sys.argv[1]
max_loads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kernel, lines = None, open(path).read().split("\n")
label_at = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label_at[m.group(1)] = i
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if not m or cur is None:
        continue
    tgt = label_at.get(m.group(1))
    if tgt is None or tgt >= i:
        continue                      # forward branch
    body = [b.strip() for b in lines[tgt + 1:i] if b.startswith("\t") and not b.strip().startswith((";", "."))]
    if any(re.match(r"^\.LBB", b) for b in lines[tgt + 1:i]):
        continue                      # not an innermost single-block loop
    loads = [b for b in body if re.match(r"^(global_load|buffer_load|flat_load)", b)]
    waits0 = [b for b in body if re.match(r"^s_waitcnt.*vmcnt\(0\)", b)]
    if loads and waits0 and len(loads) <= max_loads:
        print(f"{cur[:90]:<90} {m.group(1):<12} loads/iter {len(loads)}  instrs/iter {len(body)}")
