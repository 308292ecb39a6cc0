"""Multi-GPU exchange steps of the k-mer BQSR path (one process per GPU).

Reads shard across ranks as contiguous ranges in file order; every rank keeps
full replicas of both Bloom filters.  The path has exactly three exchange steps
and one broadcast:

  after pass 1   OR  all-reduce of the sampled bit array  + SUM of its insert counter
  after pass 2   OR  all-reduce of the trusted bit array  + SUM of its insert counter
  after pass 3   SUM all-reduce of the covariate histograms (u64)
  after training broadcast of the delta-Q tables from rank 0

RCCL has no bitwise-OR reduction, so the OR all-reduce is built from the
collectives it does have, in the direct (one-hop) form that suits a fully
connected xGMI node: every slab of the bit array is cut into world_size pieces,
one all_to_all hands piece j of every rank to rank j (all 7 links carry 1/8 of
the slab at once), rank j ORs the pieces with the engine's own HIP kernel, and
one all_gather returns the reduced pieces.  Per rank that moves 2*(N-1)/N of the
array instead of the (N-1) arrays of a gather-everything scheme.

The functions take torch tensors and a process group, so the same code runs on
`nccl` (= RCCL) with device tensors and on `gloo` with CPU tensors (tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def or_allreduce_(t, or_into, slab_words=1 << 27, group=None, or_pieces=None, force=False):
    """In-place bitwise-OR all-reduce of a 1-D int64 tensor.

    or_into(dst, src): dst |= src for two equally long int64 tensors (the HIP
    kernel on a GPU, tensor.bitwise_or_ on the CPU).  or_pieces(dst, recv, piece, n, skip), when
    given, ORs all n-1 foreign pieces of `recv` into dst in one launch.
    """
    rank, n = world(group)
    if n == 1 and not force:      # force: run the collectives anyway (single-rank plumbing check)
        return t
    assert t.dim() == 1 and t.dtype == torch.int64 and t.is_contiguous()
    total = t.numel()
    piece = max(2, (min(slab_words, total) + n - 1) // n)
    piece += piece & 1                     # keep pieces 16-byte aligned
    slab = piece * n
    recv = torch.empty(slab, dtype=torch.int64, device=t.device)
    mine = torch.empty(piece, dtype=torch.int64, device=t.device)
    pad = None
    for s in range(0, total, slab):
        ln = min(slab, total - s)
        if ln == slab:
            view = t[s:s + slab]
        else:
            if pad is None:
                pad = torch.zeros(slab, dtype=torch.int64, device=t.device)
            pad[:ln].copy_(t[s:s + ln])
            pad[ln:].zero_()
            view = pad
        dist.all_to_all_single(recv, view, group=group)
        mine.copy_(recv[rank * piece:(rank + 1) * piece])
        if or_pieces is not None:
            or_pieces(mine, recv, piece, n, rank)
        else:
            for j in range(n):
                if j != rank:
                    or_into(mine, recv[j * piece:(j + 1) * piece])
        dist.all_gather_into_tensor(view, mine, group=group)
        if ln != slab:
            t[s:s + ln].copy_(pad[:ln])
    return t


def sum_allreduce_(t, group=None):
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_range(n_reads, rank, n, align=32):
    """Contiguous read range of a rank; boundaries are multiples of `align` reads so that a
    uniform-length shard starts on a 64-base boundary of the packed arrays."""
    per = -(-n_reads // n)
    per = -(-per // align) * align
    a = min(n_reads, rank * per)
    b = min(n_reads, a + per)
    return a, b


class Exchange:
    """The exchange steps bound to one engine-like object.

    `peer` must provide: table_tensor(which) -> int64 tensor aliasing the filter's bit array,
    or_into(dst, src), get_inserted(which), set_inserted(which, n), hist_tensor() -> int64 tensor
    aliasing the tallied histograms, train() -> dict of int32 arrays, set_dq(dict), quiesce().
    """

    def __init__(self, peer, group=None, slab_words=1 << 27, device=None, force=False, stage_host=False):
        self.peer = peer
        self.group = group
        self.slab_words = slab_words
        self.rank, self.n = world(group)
        self.device = device
        self.force = force and dist.is_initialized()   # run the collectives even with one rank
        # stage_host: the peer's tensors live on a GPU but the process group cannot move device memory (gloo):
        # copy to the host, reduce there, copy back.  Lets several ranks share one GPU in the tests.
        self.stage_host = stage_host

    def _reduce(self, t, fn):
        if not self.stage_host:
            return fn(t)
        c = t.cpu()
        fn(c)
        t.copy_(c)
        if t.is_cuda:
            torch.cuda.synchronize()
        return t

    def filter_done(self, which):
        """After pass 1 (which=0) or pass 2 (which=1): make filter and counter global."""
        local = self.peer.get_inserted(which)
        if self.n == 1 and not self.force:
            return local
        self.peer.quiesce()
        if self.stage_host:
            self._reduce(self.peer.table_tensor(which),
                         lambda c: or_allreduce_(c, lambda dst, src: dst.bitwise_or_(src), self.slab_words, self.group, None, force=self.force))
        else:
            or_allreduce_(self.peer.table_tensor(which), self.peer.or_into, self.slab_words, self.group,
                          getattr(self.peer, "or_pieces", None), force=self.force)
        cnt = torch.tensor([local], dtype=torch.int64, device=self.device)
        sum_allreduce_(cnt, self.group)
        total = int(cnt.item())
        self.peer.quiesce()
        self.peer.set_inserted(which, total)
        return total

    def histograms_done(self):
        if self.n == 1 and not self.force:
            return
        self.peer.quiesce()
        self._reduce(self.peer.hist_tensor(), lambda c: sum_allreduce_(c, self.group))
        self.peer.quiesce()

    def train_and_share(self):
        """Rank 0 trains (host, long double), the int32 tables are broadcast and installed everywhere."""
        keys = ("meanq", "rg", "q", "cycle", "dinuc")
        if self.n == 1 and not self.force:
            return self.peer.train()
        if self.rank == 0:
            dq = self.peer.train()
            flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(dq[k], dtype=np.int32).reshape(-1) for k in keys]))
            shapes = [tuple(np.shape(dq[k])) for k in keys]
        else:
            shapes = self.peer.dq_shapes()
            flat = torch.empty(int(sum(int(np.prod(s)) for s in shapes)), dtype=torch.int32)
        if self.device is not None:
            flat = flat.to(self.device)
        dist.broadcast(flat, src=0, group=self.group)
        flat = flat.cpu().numpy()
        out, pos = {}, 0
        for k, s in zip(keys, shapes):
            cnt = int(np.prod(s))
            out[k] = flat[pos:pos + cnt].reshape(s).copy()
            pos += cnt
        if self.rank != 0:
            self.peer.set_dq(out)
        return out


class EnginePeer:
    """Adapter: a kbbq_amd.engine.Engine seen through the Exchange protocol."""

    def __init__(self, engine):
        from .engine import device_tensor
        self.e = engine
        self._dt = device_tensor

    def quiesce(self):
        self.e.sync()
        torch.cuda.synchronize()

    def table_tensor(self, which):
        info = self.e.filter_info(which)
        return self._dt(self.e.L.kbbq_filter_device_table(self.e.h, which), info["table_bytes"], torch.int64,
                        self.e.params.device)

    def or_into(self, dst, src):
        from . import _lib
        torch.cuda.synchronize()
        _lib.check(self.e.L.kbbq_device_or(self.e.h, dst.data_ptr(), src.data_ptr(), dst.numel()))
        self.e.sync()

    def or_pieces(self, dst, recv, piece, n, skip):
        from . import _lib
        torch.cuda.synchronize()
        _lib.check(self.e.L.kbbq_device_or_pieces(self.e.h, dst.data_ptr(), recv.data_ptr(), piece, n, skip))
        self.e.sync()

    def get_inserted(self, which):
        return self.e.filter_info(which)["inserted"]

    def set_inserted(self, which, n):
        from . import _lib
        _lib.check(self.e.L.kbbq_filter_set_inserted(self.e.h, which, n))

    def hist_tensor(self):
        import ctypes
        n = ctypes.c_uint64()
        ptr = self.e.L.kbbq_covariates_device(self.e.h, ctypes.byref(n))
        return self._dt(ptr, n.value * 8, torch.int64, self.e.params.device)

    def train(self):
        return self.e.get_dqs()

    def dq_shapes(self):
        R, C = self.e.n_rg, self.e.max_read_len
        return [(R,), (R,), (R, 94), (R, 94, 2, C), (R, 94, 16)]

    def set_dq(self, dq):
        self.e.set_dq(dq)
