"""bench.py keeps the driver's contract: one JSON line with the agreed keys, the roofline object for a kernel
that runs alone, the CPU baseline leg -- checked on a small configuration."""
import json
import os
import subprocess
import sys

import pytest

import common

pytestmark = pytest.mark.gpu


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--genome-len", "30000000",
                          "--cpu-genome-len", "300000", "--pcie-genome-len", "10000000"], cwd=common.ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "recalibrated Gbases/sec" and d["unit"] == "Gbases/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    bases = d["config"]["bases"]
    assert abs(d["value"] - bases / (d["ms_per_step"] / 1e3) / 1e9) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # an exclusive duration: the kernel's own launches of the untimed in-order step, not the time it shared the chip
    dom = d["kernels"][r["kernel"]]
    assert r["duration_measured_in"].startswith("exclusive") and r["avg_launch_ms"] == dom["exclusive_avg_ms"]
    assert r["timed_region_avg_launch_ms"] == dom["avg_ms"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) < 0.01 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Gbases/s" and c["value"] > 0 and "sample" in c
    assert d["result"]["recal_qual_sum"] > 0 and d["result"]["trusted_inserted"] > d["result"]["sampled_inserted"] > 0
    # the boundary's host-batch mode, PCIe inclusive: never the headline, always beside it
    p = d["pcie_inclusive"]
    assert p["host_link"]["h2d_GBps"] > 1 and p["host_link"]["d2h_GBps"] > 1
    for mode in ("resubmit", "upload_once"):
        assert 0 < p[mode]["value"] <= p[mode]["bound_Gbases_per_s"] * 1.05 and len(p[mode]["pass_seconds"]) == 4
        # a bound that is beaten is not a bound: the duplex form prices a one-directional pass at that direction's own rate
        assert p[mode]["fraction_of_duplex_bound"] <= 1.02, (mode, p[mode])
    assert d["ingest_inclusive_value"] == p["upload_once"]["value"] and d["ingest_inclusive_unit"] == "Gbases/s"
    assert r["useful_bytes"] is None or r["useful_bytes"] < r["algorithmic_bytes_per_launch"]


def test_bench_two_ranks_reproduce_one_rank():
    """The N > 1 path of bench.py launched the plain way -- `python bench.py --gpus 2`, no torchrun: the script
    spawns its own ranks before it touches a GPU -- here with both ranks on the one GPU of the box and gloo instead
    of RCCL: shards, ordinals, the three exchange steps and the digest must give the N = 1 answer."""
    common_args = ["--steps", "1", "--warmup", "0", "--genome-len", "30000000", "--no-cpu-baseline", "--no-pcie"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common_args, cwd=common.ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, KBBQ_BENCH_BACKEND="gloo", KBBQ_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common_args,
                         cwd=common.ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    assert len([ln for ln in two.stdout.splitlines() if ln.strip()]) == 1      # the launcher relays rank 0's line only
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 1 and a["ranks_seen"] == 1
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and "x2" in b["config"]["parallelism"]
    assert set(b["exchange_ms"]) == {"filter0", "filter1", "histograms", "broadcast"} and b["exchange_ms"]["filter1"] > 0
    for key in ("sampled_inserted", "trusted_inserted", "fpr", "recal_qual_sum"):
        assert a["result"][key] == b["result"][key], key


def test_bench_exchange_in_the_library_gives_the_same_answer():
    """bench.py --exchange lib --force-exchange: every exchange step of a step runs through the library's own RCCL calls
    (include/kbbq_exchange.h; a group of one rank on this box) at the bench's launch sizes; counts, fpr and digest must be the
    plain run's, and the line says who exchanged."""
    common_args = ["--steps", "1", "--warmup", "0", "--genome-len", "30000000", "--no-cpu-baseline", "--no-pcie", "--no-exclusive-step"]
    one = subprocess.run([sys.executable, "bench.py"] + common_args, cwd=common.ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    lib = subprocess.run([sys.executable, "bench.py", "--force-exchange", "--exchange", "lib"] + common_args, cwd=common.ROOT, capture_output=True, text=True,
                         timeout=900)
    assert lib.returncode == 0, lib.stderr[-3000:]
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in lib.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("sampled_inserted", "trusted_inserted", "fpr", "recal_qual_sum"):
        assert a["result"][key] == b["result"][key], key
    x = b["exchange_one_rank"]
    assert x["exchange_by"].startswith("library") and x["exchange_ms_per_step"]["filter0"] > 0 and x["exchange_ms_per_step"]["filter1"] > 0
