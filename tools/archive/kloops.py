#!/usr/bin/env python
"""Loops of the hand-counted kernels (fine_dma.h) must contain no vector-memory wait or scratch access of the compiler's
own: lists, per kernel of a hipcc -save-temps .s file, every inner loop with its compiler-inserted `s_waitcnt vmcnt`,
scratch_ instructions and instruction count.   usage: kloops.py file.s [kernel-substring]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "k_fine_dma"
i = 0
while i < len(lines):
    l = lines[i]
    head = l.split(";")[0].strip()
    if head.endswith(":") and pat in head and not head.startswith("."):
        name = head[:-1]
        j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        body = lines[i:j]
        labels = {b.split(":")[0].strip(): n for n, b in enumerate(body) if re.match(r"\.LBB\d+_\d+:", b)}
        in_asm = [False] * len(body)
        f = False
        for n, b in enumerate(body):
            if "#ASMSTART" in b: f = True
            in_asm[n] = f
            if "#ASMEND" in b: f = False
        print(name)
        for n, b in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", b)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                lo = labels[m.group(1)]
                if n - lo < 150: continue
                tag = "masked" if "Lb1EEv" in body[lo] else ("unmasked" if "Lb0EEv" in body[lo] else "?")
                bad = [body[k].strip() for k in range(lo, n) if not in_asm[k] and (("s_waitcnt" in body[k] and "vmcnt" in body[k]) or "scratch_" in body[k])]
                valu = sum(1 for k in range(lo, n) if body[k].strip().startswith("v_"))
                f64 = sum(1 for k in range(lo, n) if re.match(r"\s+v_\w+_f64", body[k]))
                # registers with an asm-issued VGPR load in flight: from the load to the loop end and from the loop top to the
                # last hand-written vmcnt wait no instruction may name them
                def regs_of(tok):
                    m2 = re.match(r"v\[(\d+):(\d+)\]", tok)
                    if m2: return set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                    m2 = re.match(r"v(\d+)$", tok)
                    return {int(m2.group(1))} if m2 else set()
                loads = [k for k in range(lo, n) if in_asm[k] and re.match(r"\s+buffer_load_dwordx[24] v", body[k]) and " lds" not in body[k]]
                touched = "-"
                if loads:
                    R = set()
                    for k in loads: R |= regs_of(body[k].split()[1].rstrip(","))
                    waits = [k for k in range(lo, n) if in_asm[k] and "s_waitcnt vmcnt" in body[k] and k < loads[0]]
                    wpos = waits[-1] if waits else lo
                    hits = []
                    for k in list(range(loads[-1] + 1, n)) + list(range(lo, wpos)):
                        if in_asm[k] or body[k].strip().startswith(";"): continue
                        toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", body[k])
                        if any(regs_of(t) & R for t in toks): hits.append(body[k].strip())
                    touched = "CLEAN" if not hits else "TOUCHED: %s" % hits[:3]
                print("   loop %-9s %4d instr (%3d valu, %3d f64)  compiler vm-waits/scratch: %s   regs in flight: %s" % (tag, n - lo, valu, f64, bad if bad else "none", touched))
        i = j
    i += 1
