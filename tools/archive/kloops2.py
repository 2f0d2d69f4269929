#!/usr/bin/env python
"""Per inner loop of a kernel (hipcc -save-temps .s): instruction count, VALU / FP64 count, every `s_waitcnt vmcnt(N)`
and scratch access inside it, LDS op count.   usage: kloops2.py file.s kernel-substring"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
i = 0
while i < len(lines):
    head = lines[i].split(";")[0].strip()
    if head.endswith(":") and pat in head and not head.startswith("."):
        j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        body = lines[i:j]
        labels = {b.split(":")[0].strip(): n for n, b in enumerate(body) if re.match(r"\.LBB\d+_\d+:", b)}
        print(head[:-1])
        for k in body[::-1]:
            pass
        meta = [l for l in lines[j:j + 80] if re.search(r"NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize", l)]
        print("   " + " ".join(m.strip("; ").strip() for m in meta))
        for n, b in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", b)
            if m and m.group(1) in labels and labels[m.group(1)] < n and n - labels[m.group(1)] > 100:
                lo = labels[m.group(1)]
                seg = [x.strip() for x in body[lo:n] if x.strip() and not x.strip().startswith(";")]
                vm = [re.search(r"vmcnt\((\d+)\)", x).group(1) for x in seg if "s_waitcnt" in x and "vmcnt" in x]
                print("   loop @%s %s: %d instr, %d valu (%d f64), %d ds, %d vmem, %d salu, %d waits | vmcnt: %s | scratch: %d" % (
                    m.group(1), body[lo][body[lo].find(";"):][:60], len(seg), sum(x.startswith("v_") for x in seg),
                    sum(bool(re.match(r"v_\w+_f64", x)) for x in seg), sum(x.startswith("ds_") for x in seg),
                    sum(x.startswith(("global_", "buffer_")) for x in seg), sum(x.startswith("s_") and not x.startswith("s_waitcnt") for x in seg),
                    sum(x.startswith("s_waitcnt") for x in seg), ",".join(vm), sum(x.startswith("scratch_") for x in seg)))
        i = j
    i += 1
