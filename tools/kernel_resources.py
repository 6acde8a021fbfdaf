#!/usr/bin/env python3
"""List VGPR / SGPR / LDS per kernel from a `hipcc -S --cuda-device-only` assembly file (development tool)."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
    name = g("name").group(1)
    if pat and not re.search(pat, name):
        continue
    short = re.sub(r"^_ZN2fy12_GLOBAL__N_1\d+", "", name)[:60]
    print(f"{short:62s} vgpr {g('vgpr_count').group(1):>4s} sgpr {g('sgpr_count').group(1):>4s} lds {g('group_segment_fixed_size').group(1):>6s}")
