#!/usr/bin/env python3
"""tools/pmc_json.py SUMMARY.txt OUT.json [SUMMARY_FILE_NAME] -- the per-kernel FETCH_SIZE / WRITE_SIZE averages of a
tools/pmc_summary.py table as the JSON bench.py reads (profiles/r03_pmc_latest.json), stamped with the hash of the kernel
sources they were measured on (bench.py: kernel_source_sha16): a later build with other kernels does not quote them."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

kern = {}
for ln in open(sys.argv[1]).read().splitlines()[1:]:
    f = ln.split()
    if len(f) < 5 or f[-4] not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    name = f[0].split("<")[0]
    kern.setdefault(name, {})[f[-4]] = float(f[-1])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace) of `KBBQ_NO_OVERLAP=1 python3 bench.py "
                 "--genome-len 300000000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step`: same launch size as the full-scale run "
                 "(4194304 reads per launch); KiB per launch as rocprofv3 reports them (FETCH_SIZE counts gfx950's 128-byte requests as 64 bytes: double it)",
       "summary_file": sys.argv[3] if len(sys.argv) > 3 else sys.argv[1],
       "kernel_source_sha16": bench.kernel_source_sha16(),
       "kernels": {k: v for k, v in kern.items() if len(v) == 2}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(out["kernels"]), "kernels, sources", out["kernel_source_sha16"])
