#!/usr/bin/env python
"""Static instruction census of one kernel in hipcc's assembly, split into code inside / outside loops (blocks hipcc annotates
"in Loop" / "Loop Header"), and the dynamic estimate  outside + n_iter * inside  for a kernel whose only hot loop is its tile walk.
usage: isa_dynamic.py file.s mangled_name_substring [n_iter=2]"""
import collections
import re
import sys


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.split(";")[0].rstrip().endswith(":"))
    inside, outside, in_loop = collections.Counter(), collections.Counter(), False
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m:
            in_loop = "Loop" in m.group(2)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        (inside if in_loop else outside)[klass(s.split()[0])] += 1
    dyn = {k: outside[k] + n_iter * inside[k] for k in sorted(set(inside) | set(outside))}
    print("outside loops:", dict(outside))
    print("inside loops :", dict(inside))
    print(f"dynamic (x{n_iter}):", dyn)


if __name__ == "__main__":
    main()
