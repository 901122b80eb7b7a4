#!/usr/bin/env python
"""Per-kernel scratch / spill table from a hipcc -S listing (amdhsa metadata).  usage: python tools/spills.py file.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, name, priv, sg, sgs, vg, vgs = m.groups()
    if flt in name:
        print(f"scratch {priv:>4} B  sgpr {sg:>3} (spilled {sgs:>3})  vgpr {vg:>3} (spilled {vgs:>3})  agpr {ag:>3}  {name[:110]}")
