"""Multi-process path on the CPU (gloo, world_size 2): the OR all-reduce built from all_to_all +
all_gather, and the whole sharded protocol of kbbq_amd/dist.py (filter OR + counter sum after passes
1 and 2, histogram sum after pass 3, delta-Q broadcast) driven through an oracle-backed peer -- the
result must equal the single-process run bit for bit, which is what makes the GPU path
rank-count invariant by construction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common
from kbbq_amd.dist import Exchange, or_allreduce_, shard_range
from oracle import pyoracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_or(dst, src):
    dst.bitwise_or_(src)


def _worker_or(rank, world, port, sizes, tmp):
    _init(rank, world, port)
    try:
        for n, slab in sizes:
            g = torch.Generator().manual_seed(1000 * n + rank)
            t = torch.randint(-2 ** 62, 2 ** 62, (n,), dtype=torch.int64, generator=g)
            want = None
            for r in range(world):
                gr = torch.Generator().manual_seed(1000 * n + r)
                x = torch.randint(-2 ** 62, 2 ** 62, (n,), dtype=torch.int64, generator=gr)
                want = x if want is None else want | x
            or_allreduce_(t, _cpu_or, slab_words=slab)
            assert torch.equal(t, want), (n, slab)
        open(os.path.join(tmp, "or_ok_%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_or_allreduce_equals_elementwise_or(tmp_path, world):
    sizes = [(8, 1 << 20), (1000, 64), (1001, 100), (4096, 4096), (12345, 1000), (7, 2)]
    mp.spawn(_worker_or, args=(world, _free_port(), sizes, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("or_ok_%d" % r)) for r in range(world))


class OraclePeer:
    """The Exchange protocol's view of one rank, backed by the oracle (CPU tensors)."""

    def __init__(self, o):
        self.o = o
        L = o.L
        import ctypes
        L.ko_filter_table_mut.restype = pyoracle.u64p
        L.ko_filter_table_mut.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ko_filter_set_inserted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
        L.ko_dq_set.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64] + [pyoracle.i32p] * 5
        self.hist = None

    def quiesce(self):
        pass

    def table_tensor(self, which):
        n = self.o.L.ko_filter_bits(self.o.h, which) // 64
        a = np.ctypeslib.as_array(self.o.L.ko_filter_table_mut(self.o.h, which), (n,))
        return torch.from_numpy(a.view(np.int64))      # aliases the oracle's table

    def or_into(self, dst, src):
        dst.bitwise_or_(src)

    def get_inserted(self, which):
        return self.o.filter_info(which)["inserted"]

    def set_inserted(self, which, n):
        self.o.L.ko_filter_set_inserted(self.o.h, which, n)

    def hist_tensor(self):
        c = self.o.covariates()
        self.cov = c
        flat = np.concatenate([c[k].reshape(-1) for k in ("rg", "q", "cycle", "dinuc")]).astype(np.uint64)
        self.hist = torch.from_numpy(flat.view(np.int64))
        return self.hist

    def install_hist(self):
        c, flat, pos = self.cov, self.hist.numpy().view(np.uint64), 0
        for k in ("rg", "q", "cycle", "dinuc"):
            n = c[k].size
            c[k] = flat[pos:pos + n].reshape(c[k].shape).copy()
            pos += n
        self.o.set_covariates(c)

    def train(self):
        return self.o.train()

    def dq_shapes(self):
        c = self.cov
        R, C = c["R"], c["C"]
        return [(R,), (R,), (R, common.NQ), (R, common.NQ, 2, C), (R, common.NQ, 16)]

    def set_dq(self, dq):
        a = [np.ascontiguousarray(dq[k], dtype=np.int32) for k in ("meanq", "rg", "q", "cycle", "dinuc")]
        R, C = self.cov["R"], self.cov["C"]
        self.o.L.ko_dq_set(self.o.h, R, C, *[x.ctypes.data_as(pyoracle.i32p) for x in a])


def _worker_pipeline(rank, world, port, tmp):
    _init(rank, world, port)
    try:
        k = 32
        d = common.make_dataset(seed=321, genome_len=12000, coverage=20, n_rg=2, paired=True, n_per_million=2000,
                                extra_errors=80, clusters=40)
        alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], d["coverage"], None)
        n_reads = len(d["off"]) - 1
        a, b = shard_range(n_reads, rank, world)
        off = d["off"].astype(np.int64)
        s, e = off[a], off[b]
        seq, qual = d["seq"][s:e].copy(), d["qual"][s:e].copy()
        loff = (d["off"][a:b + 1] - d["off"][a]).astype(np.uint64)
        rg = np.ascontiguousarray(d["rg"][a:b], dtype=np.int32)
        second = np.ascontiguousarray(d["second"][a:b], dtype=np.uint8)
        o = pyoracle.Oracle(k, alpha_ld, 777, approx)
        peer = OraclePeer(o)
        xch = Exchange(peer, slab_words=1 << 12)
        # pass 1: this shard's draws start where the previous shards' k-mer positions end
        lens = np.diff(off)
        first_ordinal = int(np.maximum(lens[:a] - k + 1, 0).sum())
        o.L.ko_skip_draws.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_uint64]
        o.L.ko_skip_draws(o.h, first_ordinal)
        o.sample(seq, loff)
        sampled = xch.filter_done(0)
        thr, p_text, too_high = o.compute_thresholds()
        o.trusted(seq, qual, loff)
        trusted = xch.filter_done(1)
        err = o.errors(seq, qual, loff, rg, second, tally=True)
        # both ranks must present histograms of the same shape
        c = o.covariates()
        R = torch.tensor([c["R"], c["C"]], dtype=torch.int64)
        dist.all_reduce(R, op=dist.ReduceOp.MAX)
        assert (c["R"], c["C"]) == (int(R[0]), int(R[1])), "test shards must see every read group and the longest read"
        xch.histograms_done()
        peer.install_hist()
        dq = xch.train_and_share()
        recal = o.recalibrate(seq, qual, loff, rg, second)
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), a=a, b=b, sampled=sampled, trusted=trusted, thr=thr, err=err,
                 recal=recal, t0=o.filter_table(0), t1=o.filter_table(1), dq_cycle=dq["cycle"], dq_q=dq["q"])
    finally:
        dist.destroy_process_group()


def test_sharded_protocol_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker_pipeline, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = common.make_dataset(seed=321, genome_len=12000, coverage=20, n_rg=2, paired=True, n_per_million=2000,
                            extra_errors=80, clusters=40)
    ref = common.run_oracle(d, n_rg=2)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for p in parts:
        assert int(p["sampled"]) == ref["sampled_inserted"] and int(p["trusted"]) == ref["trusted_inserted"]
        assert np.array_equal(p["thr"], ref["thresholds"])
        assert np.array_equal(p["t0"], ref["sampled_table"]) and np.array_equal(p["t1"], ref["trusted_table"])
        assert np.array_equal(p["dq_cycle"], ref["dq"]["cycle"]) and np.array_equal(p["dq_q"], ref["dq"]["q"])
    assert np.array_equal(np.concatenate([p["err"] for p in parts]), ref["errors"])
    assert np.array_equal(np.concatenate([p["recal"] for p in parts]), ref["recal"])
