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
import contextlib
import time

import numpy as np
import torch
import torch.distributed as dist

from ._lib import NQ


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def or_allreduce_(t, or_into, slab_words=1 << 26, group=None, or_pieces=None, force=False, stream=None):
    """In-place bitwise-OR all-reduce of a 1-D int64 tensor, software-pipelined over slabs.

    or_into(dst, src): dst |= src for two equally long int64 tensors (the HIP
    kernel on a GPU, tensor.bitwise_or_ on the CPU).  or_pieces(dst, recv, piece, n, skip), when
    given, ORs all n-1 foreign pieces of `recv` into dst in one launch.

    Pipeline (two receive buffers): the all_to_all of slab s+1 is issued BEFORE the OR of slab s, so on RCCL
    (collectives run on the communicator's own stream, in issue order) the links move slab s+1 while the OR kernel
    reduces slab s; the all_gather of slab s follows.  `stream` (a torch stream wrapping the engine's HIP stream)
    is made current for the whole exchange: work.wait() then orders that stream behind a collective and every
    collective behind the kernels queued on it so far -- no host-side synchronisation inside the loop.  On gloo
    (CPU tensors) wait() blocks the host and the same code is simply sequential.

    Device tensors WITHOUT `stream`: the caller's OR kernel runs on a stream this function knows nothing about, so
    nothing orders it against the collectives (which follow torch's current stream).  The loop then falls back to
    the host-synchronised sequence: the device is drained after every collective wait and after every OR.
    """
    rank, n = world(group)
    if n == 1 and not force:      # force: run the collectives anyway (single-rank plumbing check)
        return t
    assert t.dim() == 1 and t.dtype == torch.int64 and t.is_contiguous()
    total = t.numel()
    piece = max(2, (min(slab_words, total) + n - 1) // n)
    piece += piece & 1                     # keep pieces 16-byte aligned
    slab = piece * n
    n_slabs = (total + slab - 1) // slab
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    host_fence = (lambda: torch.cuda.synchronize(t.device)) if (t.is_cuda and stream is None) else (lambda: None)
    with ctx:
        recv = [torch.empty(slab, dtype=torch.int64, device=t.device) for _ in range(min(2, n_slabs))]
        mine = [torch.empty(piece, dtype=torch.int64, device=t.device) for _ in range(min(2, n_slabs))]
        pad = None
        tail = total - (n_slabs - 1) * slab
        if tail != slab:                   # the ragged last slab travels through a zero-padded copy
            pad = torch.zeros(slab, dtype=torch.int64, device=t.device)
            pad[:tail].copy_(t[total - tail:])

        def view_of(s):
            return pad if (s == n_slabs - 1 and pad is not None) else t[s * slab:(s + 1) * slab]

        def start_a2a(s):
            return dist.all_to_all_single(recv[s & 1], view_of(s), group=group, async_op=True)

        gathers = []
        w = start_a2a(0)
        for s in range(n_slabs):
            w_next = start_a2a(s + 1) if s + 1 < n_slabs else None
            w.wait()
            while len(gathers) >= 2:       # mine[s & 1] was the source of slab s-2's all_gather: that one first
                gathers.pop(0).wait()      # (RCCL: a stream dependency, no host wait -- and already implied by the
            r, m = recv[s & 1], mine[s & 1]    #  communicator's issue order; gloo runs its collectives on several threads)
            m.copy_(r[rank * piece:(rank + 1) * piece])
            host_fence()                   # (no stream given: the received pieces and the copy are complete before the OR kernel starts)
            if or_pieces is not None:
                or_pieces(m, r, piece, n, rank)
            else:
                for j in range(n):
                    if j != rank:
                        or_into(m, r[j * piece:(j + 1) * piece])
            host_fence()                   # (... and the OR kernel before the all_gather reads its result)
            gathers.append(dist.all_gather_into_tensor(view_of(s), m, group=group, async_op=True))
            w = w_next
        for g in gathers:
            g.wait()
        host_fence()
        if pad is not None:
            t[total - tail:].copy_(pad[:tail])
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

    def __init__(self, peer, group=None, slab_words=1 << 26, device=None, force=False, stage_host=False):
        self.peer = peer
        self.group = group
        self.slab_words = slab_words
        self.rank, self.n = world(group)
        self.device = device
        self.force = force and dist.is_initialized()   # run the collectives even with one rank
        # stage_host: the peer's tensors live on a GPU but the process group cannot move device memory (gloo):
        # copy to the host, reduce there, copy back.  Lets several ranks share one GPU in the tests.
        self.stage_host = stage_host
        # host wall-clock spent in the exchange steps since the last reset_timers() (each step ends synchronised)
        self.ms = {"filter0": 0.0, "filter1": 0.0, "histograms": 0.0, "broadcast": 0.0}

    def reset_timers(self):
        for k in self.ms:
            self.ms[k] = 0.0

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
        t0 = time.perf_counter()
        if self.stage_host:
            self._reduce(self.peer.table_tensor(which),
                         lambda c: or_allreduce_(c, lambda dst, src: dst.bitwise_or_(src), self.slab_words, self.group, None, force=self.force))
        else:
            stream = self.peer.torch_stream() if hasattr(self.peer, "torch_stream") else None
            or_allreduce_(self.peer.table_tensor(which), self.peer.or_into, self.slab_words, self.group,
                          getattr(self.peer, "or_pieces", None), force=self.force, stream=stream)
        cnt = torch.tensor([local], dtype=torch.int64, device=self.device)
        sum_allreduce_(cnt, self.group)
        total = int(cnt.item())
        self.peer.quiesce()
        self.peer.set_inserted(which, total)
        self.ms["filter%d" % which] += (time.perf_counter() - t0) * 1e3
        return total

    def histograms_done(self):
        if self.n == 1 and not self.force:
            return
        self.peer.quiesce()
        t0 = time.perf_counter()
        self._reduce(self.peer.hist_tensor(), lambda c: sum_allreduce_(c, self.group))
        self.peer.quiesce()
        self.ms["histograms"] += (time.perf_counter() - t0) * 1e3

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
        t0 = time.perf_counter()
        if self.device is not None:
            flat = flat.to(self.device)
        dist.broadcast(flat, src=0, group=self.group)
        flat = flat.cpu().numpy()
        self.ms["broadcast"] += (time.perf_counter() - t0) * 1e3
        out, pos = {}, 0
        for k, s in zip(keys, shapes):
            cnt = int(np.prod(s))
            out[k] = flat[pos:pos + cnt].reshape(s).copy()
            pos += cnt
        if self.rank != 0:
            self.peer.set_dq(out)
        return out


class LibExchange:
    """The same steps through the library's own exchange (include/kbbq_exchange.h, kbbq_amd/csrc/exchange.hip): RCCL called
    directly from C++ on the engine's HIP stream -- grouped ncclSend/ncclRecv, the OR kernel, ncclAllGather per slab;
    ncclAllReduce for counters and histograms; ncclBroadcast for the delta-Q tables.  This class only makes the group
    (rank 0's ncclUniqueId travels through torch.distributed, whatever its backend) and calls the three entry points; a
    C++ caller needs no Python at all.  Same interface as Exchange."""

    def __init__(self, engine, group=None, slab_words=1 << 26, device=0, force=False):
        import ctypes
        from . import _lib
        self.e = engine
        self.L = _lib.lib()
        self._lib = _lib
        self.slab_words = slab_words
        self.rank, self.n = world(group)
        self.force = force
        self.ms = {"filter0": 0.0, "filter1": 0.0, "histograms": 0.0, "broadcast": 0.0}
        self.g = None
        if self.n > 1 or force:
            uid = np.zeros(128, dtype=np.uint8)
            if self.rank == 0:
                _lib.check(self.L.kbbq_group_rccl_unique_id(uid.ctypes.data))
            if self.n > 1:
                box = [uid.tobytes()]
                dist.broadcast_object_list(box, src=0, group=group)      # (works on gloo and nccl alike)
                uid = np.frombuffer(box[0], dtype=np.uint8).copy()
            g = _lib.c_vp()
            _lib.check(self.L.kbbq_group_rccl_create(uid.ctypes.data, self.rank, self.n, device, ctypes.byref(g)))
            self.g = g
        self._ctypes = ctypes

    def close(self):
        if self.g:
            self.L.kbbq_group_destroy(self.g)
            self.g = None

    def reset_timers(self):
        for k in self.ms:
            self.ms[k] = 0.0

    def _add_ms(self):
        out = (self._ctypes.c_double * 4)()
        self._lib.check(self.L.kbbq_exchange_ms(self.g, out))
        return list(out)

    def filter_done(self, which):
        if not self.g:
            return self.e.filter_info(which)["inserted"]
        tot = self._ctypes.c_uint64()
        self._lib.check(self.L.kbbq_exchange_filter(self.e.h, which, self.g, self.slab_words, self._ctypes.byref(tot)))
        self.ms["filter%d" % which] += self._add_ms()[which]
        return tot.value

    def histograms_done(self):
        if not self.g:
            return
        self._lib.check(self.L.kbbq_exchange_histograms(self.e.h, self.g))
        self.ms["histograms"] += self._add_ms()[2]

    def train_and_share(self):
        if not self.g:
            return self.e.get_dqs()
        self._lib.check(self.L.kbbq_exchange_dq(self.e.h, self.g))
        self.ms["broadcast"] += self._add_ms()[3]
        return self.e.dq()


class EnginePeer:
    """Adapter: a kbbq_amd.engine.Engine seen through the Exchange protocol."""

    def __init__(self, engine):
        from .engine import device_tensor
        self.e = engine
        self._dt = device_tensor
        self._stream = None

    def quiesce(self):
        self.e.sync()
        torch.cuda.synchronize()

    def table_tensor(self, which):
        info = self.e.filter_info(which)
        return self._dt(self.e.L.kbbq_filter_device_table(self.e.h, which), info["table_bytes"], torch.int64,
                        self.e.params.device)

    def torch_stream(self):
        """The engine's HIP stream as a torch stream: collectives issued under it are ordered against the engine's
        kernels by events, not by host-side waits."""
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(self.e.stream_ptr(), device=torch.device("cuda", self.e.params.device))
        return self._stream

    # the OR kernels are queued on the engine's stream; or_allreduce_ runs with that stream current
    def or_into(self, dst, src):
        from . import _lib
        _lib.check(self.e.L.kbbq_device_or(self.e.h, dst.data_ptr(), src.data_ptr(), dst.numel()))

    def or_pieces(self, dst, recv, piece, n, skip):
        from . import _lib
        _lib.check(self.e.L.kbbq_device_or_pieces(self.e.h, dst.data_ptr(), recv.data_ptr(), piece, n, skip))

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
        return [(R,), (R,), (R, NQ), (R, NQ, 2, C), (R, NQ, 16)]

    def set_dq(self, dq):
        self.e.set_dq(dq)
