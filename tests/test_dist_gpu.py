"""Two ranks with real engines on ONE GPU (gloo between the processes, device memory staged through the host for the
collectives): the sharded run of kbbq_amd/dist.py -- contiguous read shards, global k-mer ordinals, OR of the 128-bit
filter arrays, counter and histogram sums, delta-Q broadcast -- must reproduce the single-process oracle bit for bit
on every rank.  The same code runs over RCCL with one GPU per rank (bench.py --gpus N); RCCL itself cannot put two
ranks on one device, which is why this test uses gloo."""
import os
import socket

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    from kbbq_amd.dist import EnginePeer, Exchange, shard_range
    from kbbq_amd.engine import Engine, plan_parameters
    from kbbq_amd.reads import ReadBatch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k = 32
        d = common.make_dataset(seed=321, genome_len=20000, coverage=24, n_rg=2, paired=True, n_per_million=2000,
                                extra_errors=120, clusters=60)
        alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], None)
        n_reads = len(d["off"]) - 1
        a, b = shard_range(n_reads, rank, world)
        full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=True)
        e = Engine(k, alpha_ld, 777, approx, n_rg=2, max_read_len=150)
        # three device-resident batches per rank, each with its own hint arrays
        cuts = [a + (b - a) * i // 3 for i in range(4)]
        devs, hints = [], []
        for x, y in zip(cuts[:-1], cuts[1:]):
            dv = e.upload(full.slice(x, y))
            nbytes = (dv.n_bases // 64 + 2) * 8
            h = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
            dv.set_hints(h.data_ptr(), h.data_ptr() + nbytes)
            devs.append(dv)
            hints.append(h)
        torch.cuda.synchronize()
        xch = Exchange(EnginePeer(e), slab_words=1 << 14, device=None, stage_host=True)
        nk = 150 - k + 1
        for dv, x in zip(devs, cuts[:-1]):
            e.subsample_kmers(dv, x * nk)           # ordinals are global: the draw stream is one, in file order
        e.sample_finish()
        sampled = xch.filter_done(0)
        thr, fpr, p_text, too_high = e.compute_thresholds()
        for dv in devs:
            e.find_trusted_kmers(dv)
        e.trusted_finish()
        trusted = xch.filter_done(1)
        for dv in devs:
            e.get_covariatedata(dv)
        xch.histograms_done()
        dq = xch.train_and_share()
        out = torch.zeros((b - a) * 150 + 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for dv, x in zip(devs, cuts[:-1]):
            e.recalibrate(dv, out.data_ptr() + (x - a) * 150)
        e.sync()
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), a=a, b=b, sampled=sampled, trusted=trusted, thr=thr, p_text=np.frombuffer(p_text.encode(), dtype=np.uint8),
                 recal=out.cpu().numpy()[:(b - a) * 150], t0=e.filter_table(0), t1=e.filter_table(1), dq_cycle=dq["cycle"], dq_q=dq["q"],
                 cov_cycle=e.covariates()["cycle"])
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_engines_sharing_one_gpu_reproduce_the_single_process_run(tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = common.make_dataset(seed=321, genome_len=20000, coverage=24, n_rg=2, paired=True, n_per_million=2000,
                            extra_errors=120, clusters=60)
    ref = common.run_oracle(d, n_rg=2)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert parts[0]["a"] == 0 and parts[-1]["b"] == len(d["off"]) - 1 and all(parts[i]["b"] == parts[i + 1]["a"] for i in range(world - 1))
    for p in parts:
        assert int(p["sampled"]) == ref["sampled_inserted"] and int(p["trusted"]) == ref["trusted_inserted"]
        assert np.array_equal(p["thr"], ref["thresholds"]) and p["p_text"].tobytes().decode() == ref["p_text"]
        assert np.array_equal(p["t0"], ref["sampled_table"]) and np.array_equal(p["t1"], ref["trusted_table"])
        C = ref["cov"]["C"]
        assert np.array_equal(p["cov_cycle"][:, :, :, :C], ref["cov"]["cycle"])
        assert np.array_equal(p["dq_cycle"][:, :, :, :C], ref["dq"]["cycle"]) and np.array_equal(p["dq_q"], ref["dq"]["q"])
    recal = np.concatenate([p["recal"] for p in parts])
    assert np.array_equal(recal, ref["recal"])
