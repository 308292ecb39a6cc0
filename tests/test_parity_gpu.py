"""GPU parity: the HIP engine (through the C ABI) against the oracle, bit for bit,
on seeded inputs small enough for the oracle to finish in seconds.

Run on the GPU box with `pytest -m gpu`.  There is no CPU fallback: without the
HIP library or without a GPU these tests fail.
"""
import ctypes

import numpy as np
import pytest

import common
from kbbq_amd import _lib, synth
from kbbq_amd.engine import Engine, device_tensor, plan_parameters
from kbbq_amd.reads import ReadBatch

pytestmark = pytest.mark.gpu


def test_library_loaded_and_gpu_present():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU: the engine has no CPU path"
    assert _lib.lib() is not None


@pytest.mark.parametrize("name", list(common.PARITY_CASES))
def test_four_passes_bit_exact(name):
    """Every intermediate of the four passes, bit for bit: filters' parameters and bit arrays, insert
    counters, thresholds, infer_read_errors flags, get_errors flags, all four histograms, the delta-Q
    tables and the recalibrated qualities.  tests/test_coverage_cpu.py shows these inputs together reach
    every branch of get_errors (correct_one, anchor adjustment, ties, recursion, over-correction...)."""
    build, dkw, rkw, ekw = common.PARITY_CASES[name]
    d = build(**dkw)
    ora = common.run_oracle(d, **rkw)
    eng = common.run_engine(d, **rkw, **ekw)
    common.assert_same_run(eng, ora)
    assert eng["stats"]["corrected_reads"] > 0


@pytest.mark.parametrize("name", ["uniform_150", "ragged_2rg_paired", "k21_low_alpha", "reads_250", "noisy", "clusters", "k9", "repeat_ties",
                                  "config0_1Mbp_20x", "config4_60x_k21", "softmasked_k9_ragged", "reads_400", "high_qualities_2rg"])
def test_infer_from_a_subset_of_the_lookups_is_bit_exact(name):
    """k_infer<SUB> (the default since round 4; kbbq_engine_tune "infer_subset"): phase 1 decides infer_read_errors from three of
    four lookups (a base whose known-present count already exceeds its threshold, or cannot reach it, is decided), phase 2
    makes the skipped lookups around the undecided bases.  Both forms against the oracle, which computes every flag from
    ALL lookups: flags, insert decisions, the trusted filter and everything behind them -- and where the thresholds leave
    room the subset form must really fetch fewer lines."""
    build, dkw, rkw, ekw = common.PARITY_CASES[name]
    d = build(**dkw)
    ora = common.run_oracle(d, **rkw)
    eng = common.run_engine(d, **rkw, **ekw, tune={"infer_subset": 1})
    common.assert_same_run(eng, ora)
    plain = common.run_engine(d, **rkw, **ekw, tune={"infer_subset": 0})
    common.assert_same_run(plain, ora)
    if name in ("uniform_150", "config0_1Mbp_20x", "reads_250"):
        assert eng["stats"]["infer_lookups"] < plain["stats"]["infer_lookups"], (eng["stats"], plain["stats"])
    else:
        assert eng["stats"]["infer_lookups"] <= plain["stats"]["infer_lookups"]


def test_one_read_per_lane_form_agrees():
    """The k < 3 / diagnostic form of the correction walk (correct.h) against the oracle; a second process
    because the choice is read once from the environment."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, 'tests'); import common\n"
            "for name in ('ragged_2rg_paired', 'clusters', 'k9', 'repeat_ties'):\n"
            "    build, dkw, rkw, ekw = common.PARITY_CASES[name]\n"
            "    d = build(**dkw)\n"
            "    common.assert_same_run(common.run_engine(d, **rkw, **ekw), common.run_oracle(d, **rkw))\n"
            "print('lane form ok')\n")
    env = dict(os.environ, KBBQ_CORRECT="lane")
    out = subprocess.run([sys.executable, "-c", code], cwd=common.ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "lane form ok" in out.stdout, out.stdout + out.stderr


def test_batching_does_not_change_results():
    d = common.make_dataset(seed=5, genome_len=15000, coverage=20)
    one = common.run_engine(d, uniform=True, n_batches=1)
    many = common.run_engine(d, uniform=False, n_batches=7)
    for key in ("sampled_table", "trusted_table", "errors", "recal", "infer_errors"):
        assert np.array_equal(one[key], many[key]), key
    assert one["sampled_inserted"] == many["sampled_inserted"]


def test_empty_and_degenerate_reads():
    """Zero-length reads, reads shorter than k and an all-N read in one ragged batch."""
    d = common.make_dataset(seed=17, genome_len=6000, coverage=20, ragged=True, short_reads=60)
    lens = np.diff(d["off"].astype(np.int64))
    # empty the first, a middle and the last read
    keep = np.ones(len(d["seq"]), dtype=bool)
    off = d["off"].astype(np.int64)
    for r in (0, len(lens) // 2, len(lens) - 1):
        keep[off[r]:off[r + 1]] = False
        lens[r] = 0
    d["seq"], d["qual"] = np.ascontiguousarray(d["seq"][keep]), np.ascontiguousarray(d["qual"][keep])
    d["off"] = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    o2 = d["off"].astype(np.int64)
    r = 5
    d["seq"][o2[r]:o2[r + 1]] = ord("N")
    ora = common.run_oracle(d)
    eng = common.run_engine(d, uniform=False, n_batches=2, max_read_len=150)
    common.assert_same_run(eng, ora)


def test_device_synth_matches_host_twin():
    sp = synth.synth_params(2024, 50000, 3000, 150, n_rg=3, paired=True, n_per_million=2000)
    host = synth.generate(sp, first_read=100, n=1000)
    e = Engine(32, 0.35, 777, 100000, n_rg=3, max_read_len=150)
    dev = e.synth_reads(sp, 100, 1000)
    got = e.download(dev)
    ref = ReadBatch(host["seq"], host["qual"], host["off"], host["rg"], host["second"], uniform=True)
    nb = ref.n_bases
    assert np.array_equal(got["qual"][:nb], ref.qual[:nb])
    assert np.array_equal(got["bases"][:nb // 32], ref.bases[:nb // 32])
    assert np.array_equal(got["nmask"][:nb // 64], ref.nmask[:nb // 64])
    assert np.array_equal(got["rg"], ref.rg)
    assert np.array_equal(got["flags"], ref.flags)
    dev.free()
    e.close()


def test_device_resident_batches_match_host_batches():
    d = common.make_dataset(seed=8, genome_len=12000, coverage=20)
    host_run = common.run_engine(d, uniform=True)
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
    e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
    hb = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    db = e.upload(hb)
    e.subsample_kmers(db, 0)
    assert e.sample_finish() == host_run["sampled_inserted"]
    e.compute_thresholds()
    e.find_trusted_kmers(db)
    assert e.trusted_finish() == host_run["trusted_inserted"]
    e.get_covariatedata(db)
    cov_d = e.covariates()
    assert np.array_equal(cov_d["cycle"], host_run["cov"]["cycle"])
    e.get_dqs()
    import torch
    out = torch.zeros(hb.n_bases + 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # the engine runs on its own stream
    e.recalibrate(db, out.data_ptr())
    e.sync()
    assert np.array_equal(out.cpu().numpy()[:hb.n_bases], host_run["recal"])
    # filters are visible to torch without a copy (what the multi-GPU exchange relies on)
    info = e.filter_info(1)
    assert info["table_bytes"] == info["n_blocks"] * 16          # the engine's 128-bit blocks
    t = device_tensor(e.L.kbbq_filter_device_table(e.h, 1), info["table_bytes"], torch.int64)
    assert np.array_equal(expand_blocks(t.cpu().numpy().view(np.uint64)), host_run["trusted_table"])
    db.free()
    e.close()


def test_host_and_device_batches_interleaved():
    """A caller may hand over some batches as host buffers and keep others resident: the passes' stream and buffer
    bookkeeping (pass 1 and pass 3 run device batches on two streams) must order them all the same."""
    d = common.make_dataset(seed=1717, genome_len=20000, coverage=24, n_per_million=1000, extra_errors=80)
    ref = common.run_engine(d, uniform=True)
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
    e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
    full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    n = full.n_reads
    cuts = [n * i // 6 for i in range(7)]
    host = [full.slice(cuts[i], cuts[i + 1]) for i in range(6)]
    mixed = [b if i % 3 == 1 else e.upload(b) for i, b in enumerate(host)]     # D H D D H D
    ordinal = 0
    for b, hb in zip(mixed, host):
        e.subsample_kmers(b, ordinal)
        ordinal += hb.n_kmer_positions(32)
    assert e.sample_finish() == ref["sampled_inserted"]
    assert np.array_equal(e.filter_table(0), ref["sampled_table"])
    e.compute_thresholds()
    for b in mixed:
        e.find_trusted_kmers(b)
    assert e.trusted_finish() == ref["trusted_inserted"]
    assert np.array_equal(e.filter_table(1), ref["trusted_table"])
    for b in mixed:
        e.get_covariatedata(b)
    c = e.covariates()
    assert np.array_equal(c["cycle"], ref["cov"]["cycle"]) and np.array_equal(c["dinuc"], ref["cov"]["dinuc"])
    for b in mixed:
        if not isinstance(b, ReadBatch):
            b.free()
    e.close()


def test_host_batches_in_reused_page_locked_memory():
    """A host batch belongs to the caller again when the call returns -- also when its memory is page-locked (the
    engine's copies are then truly asynchronous) and the caller refills the same buffers for the next batch at once."""
    import torch
    d = common.make_dataset(seed=3131, genome_len=30000, coverage=24, extra_errors=60)
    ref = common.run_engine(d, uniform=True)
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
    e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
    full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    n = full.n_reads // 4 // 32 * 32
    parts = [full.slice(i * n, (i + 1) * n) for i in range(4)]          # equal sizes: one set of buffers serves all
    keep = []

    def pinned(a):
        t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
        keep.append(t)
        return t.numpy().view(a.dtype)

    pb, pm, pq = pinned(parts[0].bases), pinned(parts[0].nmask), pinned(parts[0].qual)
    shell = parts[0]
    shell.c.bases, shell.c.nmask, shell.c.qual = pb.ctypes.data, pm.ctypes.data, pq.ctypes.data
    arrays = [(p.bases.copy(), p.nmask.copy(), p.qual.copy()) for p in parts]

    def each(fn):
        for i, (b, m, q) in enumerate(arrays):
            pb[:], pm[:], pq[:] = b, m, q          # refill the same page-locked buffers right after the previous call
            fn(i)

    each(lambda i: e.subsample_kmers(shell, i * n * (150 - 32 + 1)))
    e.sample_finish()
    sub = common.run_engine(dict(d, seq=d["seq"][:4 * n * 150], qual=d["qual"][:4 * n * 150], off=d["off"][:4 * n + 1],
                                 rg=d["rg"][:4 * n], second=d["second"][:4 * n]), uniform=True)
    assert np.array_equal(e.filter_table(0), sub["sampled_table"])
    e.compute_thresholds()
    each(lambda i: e.find_trusted_kmers(shell))
    assert e.trusted_finish() == sub["trusted_inserted"]
    assert np.array_equal(e.filter_table(1), sub["trusted_table"])
    each(lambda i: e.get_covariatedata(shell))
    c = e.covariates()
    assert np.array_equal(c["cycle"], sub["cov"]["cycle"]) and np.array_equal(c["dinuc"], sub["cov"]["dinuc"])
    e.close()
    assert ref["sampled_inserted"] >= sub["sampled_inserted"]


def test_asynchronous_submission_of_host_batches():
    """kbbq_*_batch_submit + kbbq_batch_wait: the call only queues the batch; the caller cycles through three sets of
    page-locked buffers and waits for a batch's ticket before it refills that set (and, in pass 4, before it reads the new
    qualities).  Filters, histograms and qualities equal the synchronous run's -- twelve batches, so every staging slot is
    handed out several times, with ragged and uniform layouts."""
    import ctypes
    import torch
    d = common.make_dataset(seed=4141, genome_len=40000, coverage=24, extra_errors=60)
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
    full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    nparts = 12
    n = full.n_reads // nparts // 32 * 32
    used = dict(d, seq=d["seq"][:nparts * n * 150], qual=d["qual"][:nparts * n * 150], off=d["off"][:nparts * n + 1],
                rg=d["rg"][:nparts * n], second=d["second"][:nparts * n])
    want = common.run_engine(used, uniform=True)
    parts = [full.slice(i * n, (i + 1) * n) for i in range(nparts)]
    arrays = [(p.bases.copy(), p.nmask.copy(), p.qual.copy()) for p in parts]
    keep = []

    def pinned(nbytes, dtype):
        t = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        keep.append(t)
        return t.numpy().view(dtype)

    DEPTH = 3
    sets = []
    for j in range(DEPTH):
        shell = full.slice(0, n)
        pb, pm, pq = pinned(parts[0].bases.nbytes, np.uint64), pinned(parts[0].nmask.nbytes, np.uint64), pinned(parts[0].qual.nbytes, np.uint8)
        shell.c.bases, shell.c.nmask, shell.c.qual = pb.ctypes.data, pm.ctypes.data, pq.ctypes.data
        sets.append((shell, pb, pm, pq, pinned(n * 150 + 16, np.uint8)))
    e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
    L = e.L
    recal = np.zeros(nparts * n * 150, dtype=np.uint8)

    def cycle(submit, collect=None):
        tickets = [None] * DEPTH
        for i in range(nparts + DEPTH):
            j = i % DEPTH
            if tickets[j] is not None:                      # the set's previous batch: wait before touching its memory
                _lib.check(L.kbbq_batch_wait(e.h, tickets[j][1]))
                if collect:
                    collect(tickets[j][0], sets[j])
                tickets[j] = None
            if i < nparts:
                shell, pb, pm, pq, out = sets[j]
                pb[:], pm[:], pq[:] = arrays[i]
                t = ctypes.c_uint64()
                _lib.check(submit(i, sets[j], ctypes.byref(t)))
                assert t.value != 0
                tickets[j] = (i, t.value)

    cycle(lambda i, st, t: L.kbbq_sample_batch_submit(e.h, ctypes.byref(st[0].c), i * n * (150 - 32 + 1), t))
    assert e.sample_finish() == want["sampled_inserted"]
    assert np.array_equal(e.filter_table(0), want["sampled_table"])
    e.compute_thresholds()
    cycle(lambda i, st, t: L.kbbq_trusted_batch_submit(e.h, ctypes.byref(st[0].c), t))
    assert e.trusted_finish() == want["trusted_inserted"]
    assert np.array_equal(e.filter_table(1), want["trusted_table"])
    cycle(lambda i, st, t: L.kbbq_errors_batch_submit(e.h, ctypes.byref(st[0].c), t))
    c = e.covariates()
    assert np.array_equal(c["cycle"], want["cov"]["cycle"]) and np.array_equal(c["dinuc"], want["cov"]["dinuc"])
    e.get_dqs()

    def take(i, st):
        recal[i * n * 150:(i + 1) * n * 150] = st[4][:n * 150]

    for piece in (0, 4160):      # pass 4 in one piece per batch and through the piece pipeline
        if piece:
            _lib.check(L.kbbq_engine_tune(e.h, b"pass4_piece", piece))
        recal[:] = 0
        cycle(lambda i, st, t: L.kbbq_recalibrate_batch_submit(e.h, ctypes.byref(st[0].c), st[4].ctypes.data, t), take)
        assert np.array_equal(recal, want["recal"])
    # a device batch gets ticket 0; waiting for it, or for a ticket whose slot has moved on, returns at once
    db = e.upload(parts[0])
    t = ctypes.c_uint64(77)
    _lib.check(L.kbbq_errors_batch_submit(e.h, ctypes.byref(db.c), ctypes.byref(t)))
    assert t.value == 0
    _lib.check(L.kbbq_batch_wait(e.h, 0))
    e.sync()
    db.free()
    e.close()


def test_hint_arrays_do_not_change_results():
    """kbbq_reads.hint_sampled / hint_trusted only skip lookups whose answer is known."""
    import torch
    d = common.make_dataset(seed=88, genome_len=15000, coverage=24, n_per_million=2000, extra_errors=100)
    ref = common.run_engine(d, uniform=True)
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
    e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
    hb = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    db = e.upload(hb)
    nbytes = (hb.n_bases // 64 + 2) * 8
    hints = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    db.set_hints(hints.data_ptr(), hints.data_ptr() + nbytes)
    # two sub-batches through views, like bench.py
    n = db.n_reads
    cut = (n // 2) // 32 * 32
    parts = [db.view(0, cut), db.view(cut, n - cut)]
    for v, o in zip(parts, (0, cut * (150 - 32 + 1))):
        e.subsample_kmers(v, o)
    assert e.sample_finish() == ref["sampled_inserted"]
    assert np.array_equal(e.filter_table(0), ref["sampled_table"])
    e.compute_thresholds()
    for v in parts:
        e.find_trusted_kmers(v)
    assert e.trusted_finish() == ref["trusted_inserted"]
    assert np.array_equal(e.filter_table(1), ref["trusted_table"])
    for v in parts:
        e.get_covariatedata(v)
    c = e.covariates()
    assert np.array_equal(c["cycle"], ref["cov"]["cycle"]) and np.array_equal(c["dinuc"], ref["cov"]["dinuc"])
    hs = hints.cpu().numpy()
    assert hs[:nbytes].any() and hs[nbytes:].any()       # the hints were actually written
    db.free()
    e.close()


def test_device_pattern_tables_equal_the_oracle():
    from oracle import pyoracle
    e = Engine(32, 0.35, 1, 700000)
    o = pyoracle.Oracle(32, 0.35, 1, 700000)
    for which in (0, 1):
        assert np.array_equal(e.filter_patterns(which), o.filter_patterns(which))
    e.close()


def expand_blocks(engine_words):
    """The engine's 128-bit blocks as the reference's 512-bit ones (include/kbbq_engine.h: kbbq_filter_device_table)."""
    engine_words = np.ascontiguousarray(engine_words, dtype=np.uint64)
    out = np.zeros(len(engine_words) * 4, dtype=np.uint64)
    _lib.check(_lib.lib().kbbq_host_blocks_expand(engine_words.ctypes.data_as(_lib.c_u64p), len(engine_words) // 2,
                                                  out.ctypes.data_as(_lib.c_u64p)))
    return out


def test_or_kernels_of_the_exchange_step():
    """kbbq_device_or / kbbq_device_or_pieces / kbbq_filter_or_from: the reduce step of the OR all-reduce."""
    import torch
    e = Engine(32, 0.35, 1, 100000)
    g = torch.Generator(device="cpu").manual_seed(3)
    n, pieces = 4098, 5
    recv = torch.randint(-2 ** 62, 2 ** 62, (pieces * n,), dtype=torch.int64, generator=g).cuda()
    mine = recv[2 * n:3 * n].clone()
    want = recv.view(pieces, n)[0].clone()
    for p in range(1, pieces):
        want |= recv.view(pieces, n)[p]
    torch.cuda.synchronize()
    _lib.check(e.L.kbbq_device_or_pieces(e.h, mine.data_ptr(), recv.data_ptr(), n, pieces, 2))
    e.sync()
    assert torch.equal(mine, want)
    a = recv[:n].clone()
    torch.cuda.synchronize()
    _lib.check(e.L.kbbq_device_or(e.h, a.data_ptr(), recv[n:2 * n].data_ptr(), n))
    e.sync()
    assert torch.equal(a, recv[:n] | recv[n:2 * n])
    # into the filter itself
    info = e.filter_info(0)
    words = info["table_bytes"] // 8
    src = torch.randint(-2 ** 62, 2 ** 62, (words,), dtype=torch.int64, generator=g).cuda()
    torch.cuda.synchronize()
    _lib.check(e.L.kbbq_filter_or_from(e.h, 0, src.data_ptr(), 0, words))
    assert np.array_equal(e.filter_table(0), expand_blocks(src.cpu().numpy().view(np.uint64)))
    assert e.L.kbbq_filter_or_from(e.h, 0, src.data_ptr(), 2, words) == -22      # past the end
    e.close()


def test_exchange_steps_over_rccl_with_one_rank():
    """The N > 1 plumbing on real RCCL, as far as one GPU allows: a one-rank `nccl` group with the
    collectives forced on (all_to_all / all_gather / all_reduce / broadcast over the engine's own device
    memory viewed as tensors) must leave every result unchanged.  Multi-rank equality is proven on the
    CPU in tests/test_dist_cpu.py."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, "tests")
import numpy as np, torch, torch.distributed as dist
import common
from kbbq_amd.dist import EnginePeer, Exchange
from kbbq_amd.engine import Engine
from kbbq_amd.reads import ReadBatch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d = common.make_dataset(seed=8, genome_len=12000, coverage=20, n_rg=2, paired=True, extra_errors=50)
ref = common.run_engine(d, n_rg=2, uniform=True)
alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
e = Engine(32, alpha_ld, 777, approx, n_rg=2, max_read_len=150)
x = Exchange(EnginePeer(e), slab_words=1 << 14, device=torch.device("cuda", 0), force=True)
b = e.upload(ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True))
e.subsample_kmers(b, 0); e.sample_finish()
assert x.filter_done(0) == ref["sampled_inserted"]
assert np.array_equal(e.filter_table(0), ref["sampled_table"])
e.compute_thresholds()
e.find_trusted_kmers(b); e.trusted_finish()
assert x.filter_done(1) == ref["trusted_inserted"]
assert np.array_equal(e.filter_table(1), ref["trusted_table"])
e.get_covariatedata(b)
x.histograms_done()
dq = x.train_and_share()
assert np.array_equal(dq["cycle"], ref["dq"]["cycle"]) and np.array_equal(dq["q"], ref["dq"]["q"])
out = torch.zeros(b.n_bases + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
e.recalibrate(b, out.data_ptr()); e.sync()
assert np.array_equal(out.cpu().numpy()[:b.n_bases], ref["recal"])
dist.destroy_process_group()
print("rccl one-rank exchange ok")
'''
    out = subprocess.run([sys.executable, "-c", code], cwd=common.ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl one-rank exchange ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_exchange_steps_over_rccl_with_two_ranks():
    """Two ranks over RCCL, one GPU each (runs where two devices are visible; the driver's one-GPU box skips it): each rank
    works through its shard of the reads, the filters are OR-ed through the pipelined slab loop (several slabs and a ragged
    last one: slab_words = 2^14), histograms summed, delta-Q broadcast -- filters, inserted counts, covariates and the
    recalibrated qualities of every rank's shard must equal the one-rank run's.  The same equality is proven over gloo on
    the CPU in tests/test_dist_cpu.py."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    code = r'''
import os, sys
sys.path.insert(0, "tests")
import numpy as np, torch, torch.distributed as dist
import common
from kbbq_amd.dist import EnginePeer, Exchange, shard_range
from kbbq_amd.engine import Engine
from kbbq_amd.reads import ReadBatch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
d = common.make_dataset(seed=8, genome_len=12000, coverage=20, n_rg=2, paired=True, extra_errors=50)
ref = common.run_engine(d, n_rg=2, uniform=True)              # the whole input on this rank's GPU, no exchange
alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
e = Engine(32, alpha_ld, 777, approx, n_rg=2, max_read_len=150, device=rank)
x = Exchange(EnginePeer(e), slab_words=1 << 14, device=torch.device("cuda", rank))
n = len(d["off"]) - 1
lo, hi = shard_range(n, rank, world)
off = d["off"]
sl = slice(int(off[lo]), int(off[hi]))
b = e.upload(ReadBatch(d["seq"][sl], d["qual"][sl], off[lo:hi + 1] - off[lo], d["rg"][lo:hi], d["second"][lo:hi], uniform=True))
e.subsample_kmers(b, lo * (150 - 32 + 1)); e.sample_finish()
assert x.filter_done(0) == ref["sampled_inserted"]
assert np.array_equal(e.filter_table(0), ref["sampled_table"])
e.compute_thresholds()
e.find_trusted_kmers(b); e.trusted_finish()
assert x.filter_done(1) == ref["trusted_inserted"]
assert np.array_equal(e.filter_table(1), ref["trusted_table"])
e.get_covariatedata(b)
x.histograms_done()
c = e.covariates()
assert np.array_equal(c["cycle"], ref["cov"]["cycle"]) and np.array_equal(c["dinuc"], ref["cov"]["dinuc"])
dq = x.train_and_share()
assert np.array_equal(dq["cycle"], ref["dq"]["cycle"]) and np.array_equal(dq["q"], ref["dq"]["q"])
out = torch.zeros(b.n_bases + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
e.recalibrate(b, out.data_ptr()); e.sync()
assert np.array_equal(out.cpu().numpy()[:b.n_bases], ref["recal"][sl])
dist.barrier()
dist.destroy_process_group()
print("rccl two-rank exchange ok", rank)
'''
    script = os.path.join(common.ROOT, "gpurun_out", "rccl_two_ranks.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    with open(script, "w") as f:
        f.write(code)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29741", script], cwd=common.ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and out.stdout.count("rccl two-rank exchange ok") == 2, out.stdout[-2000:] + out.stderr[-4000:]


def test_single_rank_exchange_is_a_no_op():
    from kbbq_amd.dist import EnginePeer, Exchange
    d = common.make_dataset(seed=8, genome_len=6000, coverage=20)
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
    e = Engine(32, alpha_ld, 777, approx, max_read_len=150)
    b = ReadBatch(d["seq"], d["qual"], d["off"], uniform=True)
    e.subsample_kmers(b, 0)
    n = e.sample_finish()
    x = Exchange(EnginePeer(e))
    assert x.filter_done(0) == n
    e.close()


def test_pass3_qualities_unseen_by_pass2_are_still_tallied():
    """The tally's LDS tables are laid out over the quality values pass 2 saw; a batch of pass 3 with other
    values (a caller may tally other reads than it trusted) must count them all the same."""
    from oracle import pyoracle
    d = common.make_dataset(seed=808, genome_len=15000, coverage=24, n_rg=3, paired=True, extra_errors=40)
    rng = np.random.RandomState(8)
    qual_b = np.where(d["qual"] <= 2, d["qual"], rng.randint(3, 60, size=len(d["qual"]))).astype(np.uint8)
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
    rg32, sec = np.ascontiguousarray(d["rg"], dtype=np.int32), np.ascontiguousarray(d["second"], dtype=np.uint8)
    o = pyoracle.Oracle(32, alpha_ld, 777, approx)
    o.sample(d["seq"], d["off"])
    o.compute_thresholds()
    o.trusted(d["seq"], d["qual"], d["off"])
    want_err = o.errors(d["seq"], qual_b, d["off"], rg32, sec)
    want = o.covariates()
    e = Engine(32, alpha_ld, 777, approx, n_rg=3, max_read_len=150)
    a = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    b = ReadBatch(d["seq"], qual_b, d["off"], d["rg"], d["second"], uniform=True)
    e.subsample_kmers(a, 0)
    e.sample_finish()
    e.compute_thresholds()
    e.find_trusted_kmers(a)
    e.trusted_finish()
    got_err = e.get_covariatedata(b, want_errors=True)
    got = e.covariates()
    e.close()
    assert np.array_equal(got_err, np.asarray(want_err).astype(got_err.dtype))
    R, C = want["R"], want["C"]
    assert np.array_equal(got["rg"][:R], want["rg"]) and np.array_equal(got["q"][:R], want["q"])
    assert np.array_equal(got["cycle"][:R, :, :, :C], want["cycle"]) and np.array_equal(got["dinuc"][:R], want["dinuc"])


def test_fixed_mode_tally_matches_oracle():
    # --fixed (kbbq.cc:367-378): caller-supplied error flags feed the tally directly
    d = common.make_dataset(seed=77, genome_len=8000, coverage=15, n_rg=2, paired=True)
    rng = np.random.RandomState(1)
    err = (rng.rand(len(d["seq"])) < 0.01).astype(np.uint8)
    from oracle import pyoracle
    o = pyoracle.Oracle(32, 0.35, 1, 10000)
    o.tally(d["seq"], d["qual"], d["off"], np.ascontiguousarray(d["rg"], np.int32), d["second"], err)
    oc = o.covariates()
    from kbbq_amd.reads import pack_bits
    e = Engine(32, 0.35, 1, 10000, n_rg=2, max_read_len=150)
    b = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
    e.tally(b, pack_bits(err))
    ec = e.covariates()
    for key in ("rg", "q", "cycle", "dinuc"):
        assert np.array_equal(ec[key], oc[key]), key
    e.close()


def test_error_codes():
    L = _lib.lib()
    p = _lib.Params()
    p.k, p.alpha, p.seed, p.n_rg, p.approx_kmers = 33, 0.3, 1, 1, 1000
    p.fpr_sampled, p.fpr_trusted, p.bloom_seed, p.max_read_len = 0.01, 0.0005, _lib.DEFAULT_BLOOM_SEED, 150
    h = _lib.c_vp()
    assert L.kbbq_engine_create(ctypes.byref(p), ctypes.byref(h)) == -34   # k > 32 (kbbq.cc:102)
    p.k = 32
    p.bloom_seed = 0
    assert L.kbbq_engine_create(ctypes.byref(p), ctypes.byref(h)) == -22   # invalid bloom parameters (bloom.cc:18-21)
    assert b"Invalid bloom filter parameters" in L.kbbq_last_error()
    e = Engine(32, 0.35, 1, 10000)
    d = common.make_dataset(seed=3, genome_len=2000, coverage=5)
    b = ReadBatch(d["seq"], d["qual"], d["off"], uniform=True)
    with pytest.raises(_lib.KbbqError) as ei:
        e.find_trusted_kmers(b)       # thresholds not set yet
    assert ei.value.code == -1
    with pytest.raises(_lib.KbbqError):
        e.recalibrate(b)              # no delta-Q tables yet
    # the schedule knobs: workgroups per CU 0..16 (17 = the engine's own choice), pass-2 modes 0..2
    for knob, good, bad in ((b"scan_blocks", 4, 18), (b"walk_blocks", 17, 40), (b"infer_blocks", 0, 18), (b"pass2_side", 2, 3)):
        assert L.kbbq_engine_tune(e.h, knob, good) == 0
        assert L.kbbq_engine_tune(e.h, knob, bad) == -22
    assert L.kbbq_engine_tune(e.h, b"no_such_knob", 1) == -22
    e.close()
    p.bloom_seed, p.max_read_len = _lib.DEFAULT_BLOOM_SEED, (1 << 23)
    assert L.kbbq_engine_create(ctypes.byref(p), ctypes.byref(h)) == -34   # longer than KBBQ_MAX_READ_LEN (2^23 - 1)


def test_properties_at_scale_on_device_generated_reads():
    """6·10⁹ bases (1/15 of BASELINE configs[1]; the oracle would need a quarter of an hour), so size-independent
    properties instead of an oracle comparison: the run must not depend on how the reads are cut into batches or
    on the hint arrays; the sampler's insert count must be the Bernoulli sum it is; every base is tallied exactly
    once; the trusted filter only ever receives k-mers whose bases passed; qualities below 6 come out untouched;
    and filters built by two disjoint halves OR to the filter of the whole (what the multi-GPU exchange relies on)."""
    import torch
    G, cov, L, k = 200_000_000, 30, 150, 32
    n_reads = G * cov // L
    n_reads -= n_reads % 64
    alpha_ld, cov, approx = plan_parameters(G, cov, None)
    sp = synth.synth_params(4711, G, n_reads, L, n_rg=1, paired=False, n_per_million=150)
    nk = L - k + 1

    def run(cuts, hints, keep_tables=False):
        e = Engine(k, alpha_ld, 99, approx, n_rg=1, max_read_len=L)
        dev = e.synth_reads(sp, 0, n_reads)
        hbuf = None
        if hints:
            nbytes = (n_reads * L // 64 + 2) * 8
            hbuf = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
            dev.set_hints(hbuf.data_ptr(), hbuf.data_ptr() + nbytes)
        views = [dev.view(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
        for v, a in zip(views, cuts[:-1]):
            e.subsample_kmers(v, a * nk)
        out = dict(sampled=e.sample_finish())
        out["thr"] = e.compute_thresholds()
        for v in views:
            e.find_trusted_kmers(v)
        out["trusted"] = e.trusted_finish()
        for v in views:
            e.get_covariatedata(v)
        out["cov"] = e.covariates()
        out["dq"] = e.get_dqs()
        q_new = torch.zeros(n_reads * L + 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for v, a in zip(views, cuts[:-1]):
            e.recalibrate(v, q_new.data_ptr() + a * L)
        e.sync()
        q_old = device_tensor(dev.c.qual, n_reads * L, torch.uint8)
        out.update(sum=0, changed=0, max=0, low_untouched=True)
        for c0 in range(0, n_reads * L, 1 << 30):      # in pieces: 6e9 elements are beyond 32-bit indexing
            qn, qo = q_new[c0:min(c0 + (1 << 30), n_reads * L)], q_old[c0:min(c0 + (1 << 30), n_reads * L)]
            out["sum"] += int(qn.sum(dtype=torch.int64).item())
            out["changed"] += int((qn != qo).sum().item())
            out["max"] = max(out["max"], int(qn.max().item()))
            out["low_untouched"] &= not bool(((qo < 6) & (qn != qo)).any().item())
        out["pop"] = [int(device_tensor(e.L.kbbq_filter_device_table(e.h, w), e.filter_info(w)["table_bytes"], torch.uint8)
                          .sum(dtype=torch.int64).item()) for w in (0, 1)]   # byte sums as a cheap fingerprint of the bit arrays
        out["stats"] = e.stats()
        if keep_tables:
            out["tables"] = [device_tensor(e.L.kbbq_filter_device_table(e.h, w), e.filter_info(w)["table_bytes"], torch.int64).clone() for w in (0, 1)]
            torch.cuda.synchronize()
        dev.free()
        e.close()
        return out

    whole = run([0, n_reads], hints=True, keep_tables=True)
    third = n_reads // 3 // 64 * 64
    parts = run([0, 64, third, 2 * third + 640, n_reads], hints=False)
    for key in ("sampled", "trusted", "sum", "pop", "changed"):
        assert whole[key] == parts[key], key
    assert whole["thr"][0].tolist() == parts["thr"][0].tolist()
    for key in ("cycle", "dinuc"):
        assert np.array_equal(whole["cov"][key], parts["cov"][key])
    # the sampler: one Bernoulli(alpha) draw per k-mer position, inserted when the k-mer has no N
    n_pos = n_reads * nk
    a = float(alpha_ld)
    assert abs(whole["sampled"] - a * n_pos) < 6 * (n_pos * a * (1 - a)) ** 0.5 + 0.01 * a * n_pos      # N-containing k-mers are < 1 %
    assert 0 < whole["trusted"] <= n_pos
    # every base tallied once (covariateutils.cc:193-202), errors never exceed totals
    cyc = whole["cov"]["cycle"]
    assert int(cyc[..., 1].sum()) == n_reads * L and (cyc[..., 0] <= cyc[..., 1]).all()
    assert int(whole["cov"]["dinuc"][..., 1].sum()) <= n_reads * (L - 1)
    # (the generator's substitution rate is 10^(-q/10) per base, so at this size the model finds the data
    # calibrated and may change nothing at all: "changed" is compared between the runs above, not against zero)
    assert whole["low_untouched"] and whole["max"] <= 93
    # OR of the filters of two disjoint halves of the reads == filter of all reads (exchange step)
    half = n_reads // 2 // 64 * 64
    e = Engine(k, alpha_ld, 99, approx, n_rg=1, max_read_len=L)
    dev = e.synth_reads(sp, 0, n_reads)
    tabs = []
    for a0, b0 in ((0, half), (half, n_reads)):
        e.reset()
        e.subsample_kmers(dev.view(a0, b0 - a0), a0 * nk)
        e.sample_finish()
        tabs.append(device_tensor(e.L.kbbq_filter_device_table(e.h, 0), e.filter_info(0)["table_bytes"], torch.int64).clone())
        torch.cuda.synchronize()      # the copy runs on torch's stream, the next reset on the engine's
    assert torch.equal(tabs[0] | tabs[1], whole["tables"][0])
    dev.free()
    e.close()


def test_overlapped_and_in_order_pass3_agree_on_device_batches():
    """Device-resident batches run the walk and the tally of batch i on a side stream beside the scan of batch
    i+1 (double-buffered scratch, deferred counters); KBBQ_NO_OVERLAP=1 keeps everything in order.  Same oracle
    equality either way, with enough small batches to go round the two sides many times."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, 'tests'); import common, numpy as np, torch\n"
            "from kbbq_amd.engine import Engine, plan_parameters\n"
            "from kbbq_amd.reads import ReadBatch\n"
            "d = common.make_dataset(seed=77, genome_len=30000, coverage=30, extra_errors=300, clusters=100)\n"
            "ora = common.run_oracle(d)\n"
            "alpha, cov, approx = plan_parameters(d['genome_len'], d['coverage'], None)\n"
            "e = Engine(32, alpha, 777, approx, n_rg=1, max_read_len=150)\n"
            "full = ReadBatch(d['seq'], d['qual'], d['off'], d['rg'], d['second'], uniform=True)\n"
            "n = full.n_reads; cuts = [n * i // 11 // 64 * 64 for i in range(11)] + [n]\n"
            "devs = [e.upload(full.slice(a, b)) for a, b in zip(cuts[:-1], cuts[1:])]\n"
            "o = 0\n"
            "for dv, (a, b) in zip(devs, zip(cuts[:-1], cuts[1:])):\n"
            "    e.subsample_kmers(dv, a * 119)\n"
            "e.sample_finish(); e.compute_thresholds()\n"
            "for dv in devs: e.find_trusted_kmers(dv)\n"
            "e.trusted_finish()\n"
            "for rep in range(2):\n"
            "    for dv in devs: e.get_covariatedata(dv)\n"
            "cov = e.covariates()\n"
            "assert np.array_equal(cov['cycle'][:, :, :, :ora['cov']['C']], 2 * ora['cov']['cycle'])\n"
            "assert np.array_equal(cov['dinuc'], 2 * ora['cov']['dinuc'])\n"
            "st = e.stats(); assert st['reads'] == 2 * n, st\n"
            "print('pass3 ok', st['corrected_reads'], st['correction_queries'])\n")
    outs = []
    # (the default caps the scan at four and the walk at two workgroups per CU while both streams are in use; 0 = as many as
    # fit, round 3's launches; 1 / 1: as little room as there is)
    for extra in ({}, {"KBBQ_NO_OVERLAP": "1"}, {"KBBQ_SCAN_BLOCKS": "0", "KBBQ_WALK_BLOCKS": "0"}, {"KBBQ_SCAN_BLOCKS": "1", "KBBQ_WALK_BLOCKS": "1"}):
        out = subprocess.run([sys.executable, "-c", code], cwd=common.ROOT, env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "pass3 ok" in out.stdout, out.stdout + out.stderr
        outs.append(out.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs


def test_qualities_above_93_are_modelled_like_the_reference_s_growing_tables():
    """A quality is a uint8_t and a BAM can hold up to 255 (0xFF = "missing").  The reference's covariate tables grow
    with the largest quality seen (covariateutils.cc:65-76,102-116,147-164), so such a base is tallied, gets delta-Q
    rows of its own (the model's candidates stay 0..93, :49,85,128,175) and only the OUTPUT is clamped to 93
    (readutils.cc:592-594).  Engine and oracle carry 256 quality rows: histograms, delta-Q tables and qualities equal."""
    d = common.make_dataset(seed=919, genome_len=15000, coverage=24, extra_errors=40)
    rng = np.random.RandomState(3)
    q = d["qual"].copy()
    cand = np.nonzero(q == 37)[0]                  # two fifths of the best bases get a quality above 93: enough of them for
    at = rng.choice(cand, size=len(cand) * 2 // 5, replace=False)      # the model to give those rows deltas of their own
    q[at] = rng.choice([94, 95, 120, 200, 255], size=len(at)).astype(np.uint8)
    d2 = dict(d, qual=np.ascontiguousarray(q))
    ora = common.run_oracle(d2)
    assert int(ora["cov"]["q"][0, 94:, 1].sum()) == len(at)          # the oracle did count them
    assert int(np.abs(ora["dq"]["q"][0, 94:]).sum()) > 0             # ... and trained rows for them
    assert (ora["recal"][at] < 93).all()                             # ... so they are not simply clamped
    for kw in (dict(uniform=True, n_batches=2), dict(uniform=False, n_batches=3)):
        eng = common.run_engine(d2, **kw)
        common.assert_same_run(eng, ora)
        assert eng["stats"]["quality_above_93"] is False             # (the rounds-1/2 "left out of the model" flag is gone)
    assert (np.asarray(eng["recal"]) <= 93).all()
    # the command line's flow: batches uploaded once, hint arrays, pass 3 on two streams
    import fuzz_parity
    fuzz_parity.same_resident(fuzz_parity.run_resident(d2, 32, None, 1, True, 3), ora)
    # pass 3 on its own (--fixed mode: no pass 2 has announced the quality values): same histograms
    from kbbq_amd.reads import pack_bits
    from oracle import pyoracle
    alpha_ld, cov, approx = plan_parameters(d2["genome_len"], d2["coverage"], None)
    flags = np.asarray(ora["errors"], dtype=np.uint8)
    o = pyoracle.Oracle(32, alpha_ld, 777, approx)
    o.tally(d2["seq"], d2["qual"], d2["off"], np.ascontiguousarray(d2["rg"], np.int32), d2["second"], flags)
    want = o.covariates()
    for uniform in (True, False):
        e = Engine(32, alpha_ld, 777, approx, n_rg=1, max_read_len=150)
        b = ReadBatch(d2["seq"], d2["qual"], d2["off"], d2["rg"], d2["second"], uniform=uniform)
        e.tally(b, pack_bits(flags))
        ec = e.covariates()
        e.close()
        for key in ("rg", "q", "cycle", "dinuc"):
            assert np.array_equal(ec[key], want[key]), key


@pytest.mark.parametrize("piece", ["64", "4160", "70000"])
def test_pass4_of_a_host_batch_in_pieces(piece):
    """A large host batch with its result in host memory goes through pass 4 in pieces -- copy in, kernel and copy out of
    successive pieces overlap on three streams (engine.hip: recalibrate_impl).  KBBQ_PASS4_PIECE shrinks the pieces so
    that small batches cross many piece boundaries (inside reads, between reads, at 64-base word boundaries): same
    qualities as the oracle's for uniform, ragged and many-read-group batches.  (The switch is read once per process.)"""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); import common\n"
        "for name in ('uniform_150', 'ragged_2rg_paired', 'wide_qualities_6rg_250', 'reads_400'):\n"
        "    maker, dkw, rkw, ekw = common.PARITY_CASES[name]\n"
        "    d = maker(**dkw)\n"
        "    eng = common.run_engine(d, **rkw, **dict(ekw, n_batches=2))\n"
        "    ora = common.run_oracle(d, **rkw)\n"
        "    common.assert_same_run(eng, ora)\n"
        "print('ok')\n" % os.path.dirname(os.path.abspath(__file__))
    )
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KBBQ_PASS4_PIECE=piece), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
