"""The exchange steps in the library (include/kbbq_exchange.h, kbbq_amd/csrc/exchange.hip): OR all-reduce of both filters by
grouped sends / OR kernel / all-gather in slabs, counter and histogram sums, delta-Q broadcast.

  * local group: n engines of ONE process, one host thread per rank (what a single-process multi-device caller uses) --
    here all on the one GPU of the box, n = 2 and 3, slabs far smaller than the filters so that the loop runs many rounds
    and ends in a ragged, zero-padded one.  Every rank must end with the oracle's filters, counters, histograms, delta-Q
    tables and recalibrated qualities.
  * RCCL: a group of one rank (RCCL refuses two ranks on one device): ncclSend/ncclRecv to itself, ncclAllGather,
    ncclAllReduce, ncclBroadcast through the dlopen-ed library -- every wrapper runs, results unchanged; two ranks where two
    GPUs are visible (skipped on the one-GPU boxes).
"""
import ctypes
import threading

import numpy as np
import pytest

import common
from kbbq_amd import _lib
from kbbq_amd.dist import shard_range
from kbbq_amd.engine import Engine, plan_parameters
from kbbq_amd.reads import ReadBatch

pytestmark = pytest.mark.gpu

DATA = dict(seed=321, genome_len=20000, coverage=24, n_rg=2, paired=True, n_per_million=2000, extra_errors=120, clusters=60)


def run_rank(rank, world, group, d, out, device=0, slab_words=1 << 12):
    try:
        import torch
        L = _lib.lib()
        k = 32
        alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
        n_reads = len(d["off"]) - 1
        a, b = shard_range(n_reads, rank, world)
        full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
        e = Engine(k, alpha_ld, 777, approx, n_rg=2, max_read_len=150, device=device)
        cuts = [a + (b - a) * i // 2 for i in range(3)]
        devs, hints = [], []
        for x, y in zip(cuts[:-1], cuts[1:]):
            dv = e.upload(full.slice(x, y))
            nbytes = (dv.n_bases // 64 + 2) * 8
            h = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda:%d" % device)
            dv.set_hints(h.data_ptr(), h.data_ptr() + nbytes)
            devs.append(dv)
            hints.append(h)
        torch.cuda.synchronize()
        nk = 150 - k + 1
        for dv, x in zip(devs, cuts[:-1]):
            e.subsample_kmers(dv, x * nk)           # ordinals are global: the draw stream is one, in file order
        e.sample_finish()
        tot = ctypes.c_uint64()
        _lib.check(L.kbbq_exchange_filter(e.h, 0, group, slab_words, ctypes.byref(tot)))
        sampled = tot.value
        thr, fpr, p_text, too_high = e.compute_thresholds()
        for dv in devs:
            e.find_trusted_kmers(dv)
        e.trusted_finish()
        _lib.check(L.kbbq_exchange_filter(e.h, 1, group, slab_words, ctypes.byref(tot)))
        trusted = tot.value
        for dv in devs:
            e.get_covariatedata(dv)
        _lib.check(L.kbbq_exchange_histograms(e.h, group))
        _lib.check(L.kbbq_exchange_dq(e.h, group))
        dq = e.dq()
        outq = torch.zeros((b - a) * 150 + 16, dtype=torch.uint8, device="cuda:%d" % device)
        torch.cuda.synchronize()
        for dv, x in zip(devs, cuts[:-1]):
            e.recalibrate(dv, outq.data_ptr() + (x - a) * 150)
        e.sync()
        ms = (ctypes.c_double * 4)()
        _lib.check(L.kbbq_exchange_ms(group, ms))
        out[rank] = dict(a=a, b=b, sampled=sampled, trusted=trusted, thr=thr, p_text=p_text, recal=outq.cpu().numpy()[:(b - a) * 150],
                         t0=e.filter_table(0), t1=e.filter_table(1), dq=dq, cov=e.covariates(), ms=list(ms))
        e.close()
    except BaseException as ex:      # a rank that dies must not leave the others in a barrier for ever: report and re-raise in the test
        out[rank] = ex
        raise


def check_against_oracle(parts, d):
    ref = common.run_oracle(d, n_rg=2)
    world = len(parts)
    for p in parts:
        assert not isinstance(p, BaseException), p
    assert parts[0]["a"] == 0 and parts[-1]["b"] == len(d["off"]) - 1 and all(parts[i]["b"] == parts[i + 1]["a"] for i in range(world - 1))
    C = ref["cov"]["C"]
    for p in parts:
        assert p["sampled"] == ref["sampled_inserted"] and p["trusted"] == ref["trusted_inserted"]
        assert np.array_equal(p["thr"], ref["thresholds"]) and p["p_text"] == ref["p_text"]
        assert np.array_equal(p["t0"], ref["sampled_table"]) and np.array_equal(p["t1"], ref["trusted_table"])
        assert np.array_equal(p["cov"]["cycle"][:, :, :, :C], ref["cov"]["cycle"]) and np.array_equal(p["cov"]["dinuc"], ref["cov"]["dinuc"])
        assert np.array_equal(p["dq"]["cycle"][:, :, :, :C], ref["dq"]["cycle"]) and np.array_equal(p["dq"]["q"], ref["dq"]["q"])
        assert np.array_equal(p["dq"]["meanq"], ref["dq"]["meanq"]) and np.array_equal(p["dq"]["dinuc"], ref["dq"]["dinuc"])
        assert all(m > 0 for m in p["ms"])
    assert np.array_equal(np.concatenate([p["recal"] for p in parts]), ref["recal"])


@pytest.mark.parametrize("world,slab_words", [(2, 1 << 12), (3, 1000), (3, 0)])
def test_ranks_of_one_process_reproduce_the_single_engine_run(world, slab_words):
    d = common.make_dataset(**DATA)
    L = _lib.lib()
    groups = (_lib.c_vp * world)()
    _lib.check(L.kbbq_group_local_create(world, groups))
    out = [None] * world
    threads = [threading.Thread(target=run_rank, args=(r, world, groups[r], d, out, 0, slab_words)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads), "a rank hangs in the exchange"
    for r in range(world):
        L.kbbq_group_destroy(groups[r])
    check_against_oracle(out, d)


def test_one_rank_over_rccl_every_wrapper_runs():
    d = common.make_dataset(**DATA)
    L = _lib.lib()
    uid = (ctypes.c_uint8 * 128)()
    _lib.check(L.kbbq_group_rccl_unique_id(uid))
    g = _lib.c_vp()
    _lib.check(L.kbbq_group_rccl_create(uid, 0, 1, 0, ctypes.byref(g)))
    rank, n = ctypes.c_int32(-1), ctypes.c_int32(-1)
    _lib.check(L.kbbq_group_rank(g, ctypes.byref(rank), ctypes.byref(n)))
    assert (rank.value, n.value) == (0, 1)
    out = [None]
    run_rank(0, 1, g, d, out, 0, 1 << 12)
    L.kbbq_group_destroy(g)
    check_against_oracle(out, d)


def test_two_ranks_over_rccl_where_two_gpus_are_visible():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device")
    d = common.make_dataset(**DATA)
    L = _lib.lib()
    uid = (ctypes.c_uint8 * 128)()
    _lib.check(L.kbbq_group_rccl_unique_id(uid))
    groups, out = [None, None], [None, None]

    def rank_main(r):
        g = _lib.c_vp()
        _lib.check(L.kbbq_group_rccl_create(uid, r, 2, r, ctypes.byref(g)))      # collective: both threads inside at once
        groups[r] = g
        run_rank(r, 2, g, d, out, device=r)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads)
    for g in groups:
        if g:
            L.kbbq_group_destroy(g)
    check_against_oracle(out, d)
