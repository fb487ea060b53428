#!/usr/bin/env python
"""VGPR / SGPR / scratch / LDS of every kernel in a hipcc -S assembly file: kernel_regs.py file.s [name substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or sub not in name.group(1):
        continue
    f = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]   # noqa: E731
    print(f"{name.group(1)[:70]:<70} vgpr {f('vgpr_count'):>4} sgpr {f('sgpr_count'):>4} scratch {f('private_segment_fixed_size'):>4} "
          f"spill {f('vgpr_spill_count'):>3} lds {f('group_segment_fixed_size'):>6}")
