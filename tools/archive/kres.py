#!/usr/bin/env python
"""Per-kernel resources from a hipcc -save-temps .s file: VGPRs, SGPRs, spills, LDS, occupancy."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = g("name")
    if pat and pat not in name: continue
    print("%-70s vgpr %3s agpr %3s sgpr %3s spill v%s s%s lds %6s" % (name[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size")))
