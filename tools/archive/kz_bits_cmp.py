import sys, torch
ref = torch.load(sys.argv[1])
for f in sys.argv[2:]:
    o = torch.load(f)
    for i, (a, c) in enumerate(zip(ref, o)):
        d = (a - c).abs()
        bad = torch.nonzero(d > 0).flatten()
        pl = 3 * 17 * 9
        print(f, "case", i, "maxdiff %.3e" % d.max().item(), "n", bad.numel(), "planes", torch.unique(bad // pl).tolist(), "first", [(int(q) // pl, (int(q) % pl) // 51, ((int(q) % pl) % 51) // 3, int(q) % 3) for q in bad[:6]])
