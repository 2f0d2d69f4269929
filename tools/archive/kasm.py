#!/usr/bin/env python
"""Extract one kernel's ISA from a hipcc -save-temps .s file and print a digest of its memory / sync instructions.
usage: kasm.py file.s kernel-name-substring [full]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.split(";")[0].strip().endswith(":") and pat in l and not l.startswith(".") and not l.startswith(";"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if len(sys.argv) > 3:
    print("\n".join(body)); sys.exit(0)
keys = ("s_cbranch", "scratch_", "s_barrier", "s_waitcnt", "buffer_load", "buffer_store", "global_load", "global_store", "s_endpgm", "ds_read", "ds_write", "ds_bpermute", "s_load")
n_valu = 0
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith(".LBB") or t.startswith("; %bb"):
        print("%5d %s   [valu so far %d]" % (i, t[:110], n_valu)); n_valu = 0
    elif t.startswith("v_"):
        n_valu += 1
    elif any(t.startswith(k) for k in keys):
        print("%5d    %s" % (i, t[:100]))
print("lines", len(body))
