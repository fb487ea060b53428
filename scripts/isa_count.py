#!/usr/bin/env python
"""Instruction-class census of a gfx950 kernel from hipcc's assembly (-S --cuda-device-only), per basic block.
usage: isa_count.py file.s mangled_kernel_name_substring [min_block_size]"""
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
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "trans"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    minb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.split(";")[0].rstrip().endswith(":"))
    seg, cur = collections.OrderedDict(), "entry"
    seg[cur] = collections.Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        s = s.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":") and not s.startswith("."):
            cur = s[:-1]
            seg[cur] = collections.Counter()
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            cur = s[:-1]
            seg[cur] = collections.Counter()
            continue
        if s.startswith("."):
            continue
        seg[cur][klass(s.split()[0])] += 1
    tot = collections.Counter()
    for k, v in seg.items():
        tot.update(v)
        if sum(v.values()) >= minb:
            print(f"{k:<12}", dict(sorted(v.items())))
    print("TOTAL       ", dict(sorted(tot.items())))


if __name__ == "__main__":
    main()
