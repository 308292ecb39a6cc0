"""Prints the per-kernel averages of bench.py JSON lines side by side: python tools/ab_show.py FILE [FILE ...]
(a bare TAG means gpurun_out/ab_TAG.json)."""
import json
import os
import sys
rows = {}
tags = sys.argv[1:]
for t in tags:
    path = t if os.path.exists(t) else "gpurun_out/ab_%s.json" % t
    d = json.loads(open(path).read().strip().splitlines()[-1])
    rows[t] = dict(step=d["ms_per_step"], digest=d["result"]["recal_qual_sum"], **{k: v["avg_ms"] for k, v in d["kernels"].items()})
keys = []
for t in tags:
    keys += [k for k in rows[t] if k not in keys]
names = [os.path.basename(t).replace(".json", "")[-22:] for t in tags]
print("%-22s" % "" + "".join("%24s" % n for n in names))
for k in keys:
    print("%-22s" % k + "".join("%24.3f" % rows[t].get(k, float("nan")) if k != "digest" else "%24d" % rows[t][k] for t in tags))
