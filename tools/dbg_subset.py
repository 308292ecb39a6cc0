"""Where k_infer<SUB> and the plain form disagree (diagnostic; GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import common
build, dkw, rkw, ekw = common.PARITY_CASES[sys.argv[1] if len(sys.argv) > 1 else "uniform_150"]
d = build(**dkw)
a = common.run_engine(d, **rkw, **ekw)
b = common.run_engine(d, **rkw, **ekw, tune={"infer_subset": 1})
print("thresholds", a["thresholds"])
print("lookups", a["stats"]["infer_lookups"], b["stats"]["infer_lookups"])
ea, eb = a["infer_errors"], b["infer_errors"]
bad = np.nonzero(ea != eb)[0]
print("mismatching bases", len(bad), "of", len(ea))
off = d["off"].astype(np.int64)
seen = set()
for g in bad[:400]:
    r = int(np.searchsorted(off, g, side="right") - 1)
    if r in seen:
        continue
    seen.add(r)
    L = int(off[r + 1] - off[r])
    pa = "".join("x" if v else "." for v in ea[off[r]:off[r + 1]])
    pb = "".join("x" if v else "." for v in eb[off[r]:off[r + 1]])
    df = "".join("^" if x != y else " " for x, y in zip(pa, pb))
    print("read", r, "len", L)
    print(" plain ", pa)
    print(" subset", pb)
    print("       ", df)
    print(" seq   ", bytes(d["seq"][off[r]:off[r + 1]]).decode())
    print(" lowq  ", "".join("q" if v <= 2 else "." for v in d["qual"][off[r]:off[r + 1]]))
    if len(seen) >= 6:
        break
