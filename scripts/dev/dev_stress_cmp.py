import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = (a["flags"] != b["flags"]) | (a["steps"] != b["steps"])
d = np.abs(a["end"] - b["end"]).max(1)
print("rays", len(bad), "flag/step mismatches", int(bad.sum()), "end diff max (matching rays) %.3e" % np.nanmax(np.where(bad, 0, d)))
i = np.nonzero(bad)[0]
print("first mismatching rays", i[:20], "flags", a["flags"][i[:20]], b["flags"][i[:20]], "steps", a["steps"][i[:20]], b["steps"][i[:20]])
if len(i):
    print("mismatch index spacing: min", np.diff(i).min() if len(i) > 1 else None, "clusters of 64-aligned batches:", np.unique(i // 64)[:20])
