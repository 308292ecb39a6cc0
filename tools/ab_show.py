"""Prints the per-kernel averages of gpurun_out/ab_<tag>.json files side by side."""
import json
import sys
rows = {}
tags = sys.argv[1:]
for t in tags:
    d = json.loads(open("gpurun_out/ab_%s.json" % t).read().strip().splitlines()[-1])
    rows[t] = dict(step=d["ms_per_step"], digest=d["result"]["recal_qual_sum"], **{k: v["avg_ms"] for k, v in d["kernels"].items()})
keys = list(rows[tags[0]].keys())
print("%-18s" % "" + "".join("%14s" % t for t in tags))
for k in keys:
    print("%-18s" % k + "".join("%14.3f" % rows[t].get(k, float("nan")) if k != "digest" else "%14d" % rows[t][k] for t in tags))
