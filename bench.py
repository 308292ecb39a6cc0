#!/usr/bin/env python3
"""bench.py -- recalibrated Gbases/s of the MI355X k-mer BQSR engine.

Workload (BASELINE.json configs[1]): seeded synthetic 30x human-WGS-scale reads
(genome 3e9, 150 bp, k=32 -> 6e8 reads, 9e10 bases), generated on the device and
resident in HBM together with both Bloom filters (25.2 GB + 41.5 GB).  One
"step" is one complete run of the hot path over the whole data set:

    reset -> pass 1 sample+insert -> [exchange] -> thresholds -> pass 2 trusted ->
    [exchange] -> pass 3 errors+tally -> [exchange] -> delta-Q model (host) ->
    pass 4 apply

With N > 1 ranks (torchrun, one per GPU) the reads are sharded by contiguous
ranges (strong scaling: the data set is fixed) and the three exchange steps of
kbbq_amd/dist.py run over RCCL.

Prints ONE JSON line on rank 0.  The `cpu_baseline` leg (N=1 only) times the
oracle -- the single-threaded CPU restatement of the reference -- on a bounded
sample of the same workload; it is a reported baseline, never the product path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch, numpy and the engine library are imported in _imports(), AFTER the decision to spawn ranks: the parent of a
# self-launched multi-rank run must never load a GPU runtime (it only relays rank 0's line).
np = torch = _lib = synth = EnginePeer = Exchange = shard_range = Engine = plan_parameters = None


# The ROCm runtime spreads a process's streams over four hardware queues unless told otherwise; with torch's and RCCL's streams
# beside the engine's three, two of the engine's can share one and then run in order (seen in the command line: DESIGN.md
# section 4, "Hardware queues").  Read by the runtime at the first HIP call, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _imports():
    global np, torch, _lib, synth, EnginePeer, Exchange, shard_range, Engine, plan_parameters
    import numpy as np_
    import torch as torch_
    from kbbq_amd import _lib as lib_, synth as synth_
    from kbbq_amd.dist import EnginePeer as EP, Exchange as EX, shard_range as SR
    from kbbq_amd.engine import Engine as EN, plan_parameters as PP
    np, torch, _lib, synth, EnginePeer, Exchange, shard_range, Engine, plan_parameters = np_, torch_, lib_, synth_, EP, EX, SR, EN, PP

K = int(os.environ.get("KBBQ_BENCH_K", 32))          # BASELINE configs[1]: 32; configs[4] (60x, k=21, -a 0.05) via the KBBQ_BENCH_* knobs
ALPHA = os.environ.get("KBBQ_BENCH_ALPHA") or None     # text, parsed to long double like the command line's --alpha
READ_LEN = 150
SEED_DATA = 12345
SEED_SAMPLER = 777
BATCH_READS = int(os.environ.get("KBBQ_BENCH_BATCH", 1 << 22))   # reads per engine call
PCIE_BATCH_READS = 1 << 20     # host batches of the PCIe-inclusive leg: what the command line hands over (kbbq_cli.cc)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha16():
    """Identity of the kernels a PMC summary belongs to: the sources of the pass kernels (a counter summary taken on other
    sources must not be quoted next to fresh timings)."""
    import hashlib
    h = hashlib.sha256()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbbq_amd", "csrc")
    for f in ("kernels.h", "device_common.h", "bucket.h", "correct_wave.h", "long_reads.h"):
        try:
            h.update(open(os.path.join(here, f), "rb").read())
        except OSError:
            h.update(b"missing:" + f.encode())
    return h.hexdigest()[:16]


# kernels that share the chip with a kernel of the neighbouring batch on the other stream (pass 1: draw beside insert;
# pass 3: walk + tally beside scan): their event durations are not exclusive costs
PASS3_KERNELS = ("k_draw_mask", "k_insert_sampled", "k_emit_sampled", "k_scan_trusted", "k_compact", "k_correct_wave", "k_correct", "k_tally")
# the emits of pass 2 run on the side stream beside k_infer (KBBQ_PASS2_SIDE=2, the default since round 4; 1: split and apply
# as well, round 3's default; 0: everything in order): those kernels' durations are shared too
_side = os.environ.get("KBBQ_PASS2_SIDE", "2")
if _side not in ("", "0"):
    PASS3_KERNELS += ("k_infer", "k_emit_trusted") + (("k_split_trusted", "k_apply_trusted") if _side == "1" else ())


def run_step(e, xch, batches, ordinals, out_buf, hints, pass_ms=None):
    """One complete pass of the hot path over this rank's shard.  pass_ms (a list, diagnostic runs only): receives the wall
    time of each of the four passes in ms -- the passes end in a wait of their own anyway (the *_finish calls, the histograms)."""
    marks = []

    def mark(wait=False):
        if pass_ms is not None:
            if wait:
                e.sync()
            marks.append(time.perf_counter())
    e.reset()
    hints.zero_()
    torch.cuda.synchronize()
    mark()
    for b, o in zip(batches, ordinals):
        e.subsample_kmers(b, o)
    e.sample_finish()
    sampled = xch.filter_done(0)
    thr, fpr, p_text, too_high = e.compute_thresholds()
    mark()
    for b in batches:
        e.find_trusted_kmers(b)
    e.trusted_finish()
    trusted = xch.filter_done(1)
    mark()
    for b in batches:
        e.get_covariatedata(b)
    xch.histograms_done()
    xch.train_and_share()
    mark(wait=True)
    for b in batches:
        e.recalibrate(b, out_buf.data_ptr())
    e.sync()
    mark()
    if pass_ms is not None:
        pass_ms[:] = [round((marks[i + 1] - marks[i]) * 1e3, 1) for i in range(4)]
    return dict(sampled_inserted=sampled, trusted_inserted=trusted, fpr=fpr, fpr_too_high=too_high)


def exchange_one_rank(e, xch, steps, device):
    """--force-exchange: what the exchange steps cost on one rank at full size (the transport itself needs the 8-GPU node),
    the rate of the OR kernel at an 8-rank slab shape, and the HBM left beside filters, reads and record buffers."""
    import ctypes
    out = {"exchange_ms_per_step": {k: round(v / steps, 2) for k, v in xch.ms.items()},
           "filter_bytes": [e.filter_info(0)["table_bytes"], e.filter_info(1)["table_bytes"]],
           "slab_bytes": xch.slab_words * 8}
    # the reduce step of an 8-rank exchange: 7 received pieces of a 512 MB slab OR-ed into the rank's own piece
    n, piece = 8, (xch.slab_words // 8) & ~1
    recv = torch.randint(0, 2 ** 62, (n * piece,), dtype=torch.int64, device="cuda")
    mine = torch.zeros(piece, dtype=torch.int64, device="cuda")
    peer = getattr(xch, "peer", None) or EnginePeer(e)      # (LibExchange has no peer object: the library holds the engine)
    out["exchange_by"] = "library (include/kbbq_exchange.h: RCCL from C++)" if not hasattr(xch, "peer") else "torch.distributed (kbbq_amd/dist.py)"
    with torch.cuda.stream(peer.torch_stream()):
        peer.or_pieces(mine, recv, piece, n, 3)
        e.sync()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            peer.or_pieces(mine, recv, piece, n, 3)
        e.sync()
        dt = (time.perf_counter() - t0) / reps
    out["or_pieces_8_ranks"] = {"piece_bytes": piece * 8, "ms": round(dt * 1e3, 4),
                                "GBps_read_and_written": round(((n - 1) + 2) * piece * 8 / dt / 1e9, 1)}
    del recv, mine
    free_b, total_b = ctypes.c_uint64(), ctypes.c_uint64()
    e.L.kbbq_device_memory(device, ctypes.byref(free_b), ctypes.byref(total_b))
    out["hbm_free_GB"] = round(free_b.value / 1e9, 1)
    out["hbm_total_GB"] = round(total_b.value / 1e9, 1)
    out["note"] = ("one rank: all_to_all and all_gather move every slab through RCCL to the rank itself, so exchange_ms shows the "
                   "software path at full size (slab loop, event ordering, OR launches), not xGMI transport")
    return out


def emulate_rank(e, batches, ordinals, mine, out_buf, hints, emu, args, G, cov):
    """Per-rank compute of rank R in an N-rank strong-scaling job, measured on ONE GPU (diagnostic, not the metric).

    What a rank does between the exchanges is timed on its shard with the filters in the state they have at that
    point of the real job: pass 1 into EMPTY filters (the young-filter regime of the inserts), pass 2 against the
    GLOBAL sampled filter (the other shards' pass 1 runs untimed in between: that is what the OR all-reduce
    delivers) into a trusted filter that holds this shard only, pass 3 against the global trusted filter, the
    model, pass 4.  The exchanges themselves are not in it (priced from bytes and link rates in DESIGN.md)."""
    t = {}
    kern = {}

    def timed(name, fn):
        e.sync()
        e.profile_reset()
        t0 = time.perf_counter()
        fn()
        e.sync()
        t[name] = t.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        for k, (n, ms) in e.profile().items():
            a = kern.setdefault(k, [0, 0.0])
            a[0] += n
            a[1] += ms

    sel = [(b, o) for b, o, m in zip(batches, ordinals, mine) if m]
    rest = [(b, o) for b, o, m in zip(batches, ordinals, mine) if not m]
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            t.clear()
            kern.clear()
        e.reset()
        hints.zero_()
        torch.cuda.synchronize()
        timed("pass1", lambda: ([e.subsample_kmers(b, o) for b, o in sel], e.sample_finish()))
        for b, o in rest:
            e.subsample_kmers(b, o)
        e.sample_finish()
        e.compute_thresholds()
        timed("pass2", lambda: ([e.find_trusted_kmers(b) for b, _ in sel], e.trusted_finish()))
        for b, _ in rest:
            e.find_trusted_kmers(b)
        e.trusted_finish()
        timed("pass3", lambda: [e.get_covariatedata(b) for b, _ in sel])
        for b, _ in rest:
            e.get_covariatedata(b)
        timed("model", lambda: e.get_dqs())
        timed("pass4", lambda: [e.recalibrate(b, out_buf.data_ptr()) for b, _ in sel])
    n = args.steps
    shard_bases = sum(b.n_bases for b, _ in sel)
    total = sum(t.values()) / n
    print(json.dumps({
        "metric": "DIAGNOSTIC: per-rank compute of rank %d/%d of the strong-scaling job, emulated on one GPU "
                  "(filters in their real state per pass, exchanges excluded)" % emu,
        "value": None, "shard_bases": shard_bases, "shard_batches": len(sel), "rank_compute_ms": round(total, 2),
        "pass_ms": {k: round(v / n, 2) for k, v in t.items()},
        "ideal_ms_note": "1/N of the N=1 step",
        "overlap": not os.environ.get("KBBQ_NO_OVERLAP"), "bucketed_inserts": e.stats()["bucket_capacity"] > 0,
        "kernels": {k: {"launches": v[0] // n, "avg_ms": round(v[1] / max(1, v[0]), 4)} for k, v in kern.items() if v[0]},
        "config": {"workload": "%dx synthetic WGS reads, genome %d bp, %d bp reads, k=%d" % (cov, G, READ_LEN, K)}}), flush=True)


def kernel_model(name, bases, nk, alpha, f_t, walk_queries=0.0):
    """Algorithmic HBM bytes one launch over `bases` bases / `nk` k-mer positions moves
    (packed bases 0.25 B, quals 1 B, Bloom query 64 B, insert 64 B + 64 B).
    The correction walk is data dependent: its figure is the number of Bloom queries the kernel
    counted (x 64 B) -- SURVEY excludes it from the per-base formula."""
    return {
        "k_correct_wave": walk_queries * 64.0,
        "k_correct": walk_queries * 64.0,
        "k_draw_mask": nk / 8.0,
        "k_insert_sampled": bases * 0.25 + nk / 8.0 + nk * alpha * 128.0,
        # slice-bucketed inserts (bucket.h): the emit kernels write one 8-byte record per insert
        "k_emit_sampled": bases * 0.25 + nk / 8.0 + nk * alpha * 8.0,
        "k_emit_trusted": bases * 0.25 + bases / 8.0 + nk * f_t * 8.0,
        "k_infer": bases * 1.25 + nk * 64.0,
        "k_insert_trusted": bases * 0.25 + nk * f_t * 128.0,
        "k_scan_trusted": bases * 0.25 + nk * 64.0,
        "k_tally": bases * 1.25 + bases / 8.0,
        "k_recalibrate": bases * 2.25,
    }.get(name)


class HostPacked:
    """A packed batch in PAGE-LOCKED host memory (kbbq_host_alloc): what a driver that decodes on the CPU hands over
    the boundary (caller-owned pinned SoA buffers)."""

    def __init__(self, e, dev):
        from kbbq_amd import _lib
        h = e.download(dev)
        nb = dev.n_bases
        self.pins = [_lib.PinnedArray(nb + 16, np.uint8), _lib.PinnedArray(nb // 32 + 2, np.uint64), _lib.PinnedArray(nb // 64 + 2, np.uint64)]
        self.qual, self.bases, self.nmask = (p.array for p in self.pins)
        self.qual[:nb] = h["qual"]
        self.qual[nb:] = 0
        self.bases[:len(h["bases"])] = h["bases"]
        self.bases[len(h["bases"]):] = 0
        self.nmask[:len(h["nmask"])] = h["nmask"]
        self.nmask[len(h["nmask"]):] = 0
        self.n_reads, self.n_bases = dev.n_reads, dev.n_bases
        self.c = _lib.Reads()
        self.c.n_reads, self.c.n_bases = dev.n_reads, dev.n_bases
        self.c.bases, self.c.nmask, self.c.qual = self.bases.ctypes.data, self.nmask.ctypes.data, self.qual.ctypes.data
        self.c.offsets = None
        self.c.flags = None
        self.c.rg = None
        self.c.read_len = READ_LEN
        self.c.on_device = 0

    def free(self):
        self.qual = self.bases = self.nmask = None
        for p in self.pins:
            p.free()


def pcie_inclusive(genome_len, coverage, local_rank):
    """The boundary's other mode on a bounded sample: batches handed over as HOST buffers (page-locked), reported
    beside `value`, never as `value`.  Two ways over the same boundary:
      resubmit     every pass re-submits every batch (what the reference's four re-reads of its input imply):
                   bases + N mask four times, qualities three times = 4.5 B/base H2D, new qualities 1 B/base D2H
      upload_once  kbbq_reads_upload once (while pass 1 of the batch before runs), batches and hint arrays resident for
                   passes 2-4, new qualities back through kbbq_recalibrate_batch_host: 1.4 B/base H2D, 1 B/base D2H
    with the host link's measured rates (one 1 GiB pinned copy each way) and, from them, each mode's bound
    sum over passes of max(transfer, resident compute) -- transfers of passes 1 and 4 cannot hide behind passes 2-3."""
    from kbbq_amd import _lib
    L = _lib.lib()
    up, dn = ctypes.c_double(), ctypes.c_double()
    _lib.check(L.kbbq_measure_host_link(local_rank, 1 << 30, ctypes.byref(up), ctypes.byref(dn)))
    up2, dn2 = ctypes.c_double(), ctypes.c_double()
    _lib.check(L.kbbq_measure_host_link_duplex(local_rank, 1 << 30, ctypes.byref(up2), ctypes.byref(dn2)))
    n_reads = genome_len * coverage // READ_LEN
    alpha_ld, cov, approx = plan_parameters(genome_len, coverage, None)
    e = Engine(K, alpha_ld, SEED_SAMPLER, approx, n_rg=1, max_read_len=READ_LEN, device=local_rank)
    sp = synth.synth_params(SEED_DATA, genome_len, n_reads, READ_LEN, n_rg=1, paired=False, n_per_million=100)
    dev = e.synth_reads(sp, 0, n_reads)
    host, ordinals = [], []
    PB = PCIE_BATCH_READS
    for s in range(0, n_reads, PB):
        n = min(PB, n_reads - s)
        host.append(HostPacked(e, dev.view(s, n)))
        ordinals.append(s * (READ_LEN - K + 1))
    nb = n_reads * READ_LEN

    def passes(run_batches):
        """time the four passes separately (each ends synchronised)"""
        t = []
        for fn in run_batches:
            e.sync()
            t0 = time.perf_counter()
            fn()
            e.sync()
            t.append(time.perf_counter() - t0)
        return t

    # resident reference for the bound: the same batches from HBM
    hint_bytes = (nb // 64 + 2) * 8
    hints = torch.zeros(2 * hint_bytes, dtype=torch.uint8, device="cuda")
    dev.set_hints(hints.data_ptr(), hints.data_ptr() + hint_bytes)
    res = [dev.view(s, min(PB, n_reads - s)) for s in range(0, n_reads, PB)]
    out_dev = torch.empty(PB * READ_LEN + 16, dtype=torch.uint8, device="cuda")
    t_res = passes([lambda: ([e.subsample_kmers(b, o) for b, o in zip(res, ordinals)], e.sample_finish(), e.compute_thresholds()),
                    lambda: ([e.find_trusted_kmers(b) for b in res], e.trusted_finish()),
                    lambda: ([e.get_covariatedata(b) for b in res], e.get_dqs()),
                    lambda: [e.recalibrate(b, out_dev.data_ptr()) for b in res]])
    dev.free()
    del hints
    out_pin = _lib.PinnedArray(PB * READ_LEN + 16, np.uint8)
    out = out_pin.array
    _lib.check(e.L.kbbq_engine_reset(e.h))
    # The asynchronous form of the boundary (kbbq_*_batch_submit + kbbq_batch_wait): a caller that cycles through DEPTH
    # page-locked batches waits for batch i - DEPTH (its memory free again, its qualities back) before it hands over
    # batch i, so the copies of consecutive batches are queued back to back and pass 4's two directions overlap.
    DEPTH = 3
    outs = [_lib.PinnedArray(PB * READ_LEN + 16, np.uint8) for _ in range(DEPTH)]

    def in_flight(submit):
        tickets = []
        for i, b in enumerate(host):
            if len(tickets) >= DEPTH:
                _lib.check(e.L.kbbq_batch_wait(e.h, tickets.pop(0)))
            t = ctypes.c_uint64()
            _lib.check(submit(i, b, ctypes.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            _lib.check(e.L.kbbq_batch_wait(e.h, t))

    t_sub = passes([lambda: (in_flight(lambda i, b, t: e.L.kbbq_sample_batch_submit(e.h, ctypes.byref(b.c), ordinals[i], t)), e.sample_finish(), e.compute_thresholds()),
                    lambda: (in_flight(lambda i, b, t: e.L.kbbq_trusted_batch_submit(e.h, ctypes.byref(b.c), t)), e.trusted_finish()),
                    lambda: (in_flight(lambda i, b, t: e.L.kbbq_errors_batch_submit(e.h, ctypes.byref(b.c), t)), e.get_dqs()),
                    lambda: in_flight(lambda i, b, t: e.L.kbbq_recalibrate_batch_submit(e.h, ctypes.byref(b.c), outs[i % DEPTH].array.ctypes.data, t))])
    e.sync()
    sub_digest = int(outs[(len(host) - 1) % DEPTH].array[:host[-1].c.n_bases].astype(np.int64).sum())
    # the same four passes through the plain (synchronous) calls, for the record and as a check of the asynchronous ones
    _lib.check(e.L.kbbq_engine_reset(e.h))
    t_sync = passes([lambda: ([e.subsample_kmers(b, o) for b, o in zip(host, ordinals)], e.sample_finish(), e.compute_thresholds()),
                     lambda: ([e.find_trusted_kmers(b) for b in host], e.trusted_finish()),
                     lambda: ([e.get_covariatedata(b) for b in host], e.get_dqs()),
                     lambda: [_lib.check(e.L.kbbq_recalibrate_batch(e.h, ctypes.byref(b.c), out.ctypes.data)) for b in host]])
    if int(out[:host[-1].c.n_bases].astype(np.int64).sum()) != sub_digest:
        raise SystemExit("bench.py: asynchronous and synchronous host-batch passes disagree")
    for o in outs:
        o.free()
    _lib.check(e.L.kbbq_engine_reset(e.h))
    resident = []

    def upload_and_sample():
        for b, o in zip(host, ordinals):
            d = _lib.Reads()
            _lib.check(e.L.kbbq_reads_upload(e.h, ctypes.byref(b.c), ctypes.byref(d)))
            _lib.check(e.L.kbbq_reads_alloc_hints(ctypes.byref(d)))
            _lib.check(e.L.kbbq_sample_batch(e.h, ctypes.byref(d), o))
            resident.append(d)
        e.sample_finish()
        e.compute_thresholds()

    t_once = passes([upload_and_sample,
                     lambda: ([_lib.check(e.L.kbbq_trusted_batch(e.h, ctypes.byref(d), None)) for d in resident], e.trusted_finish()),
                     lambda: ([_lib.check(e.L.kbbq_errors_batch(e.h, ctypes.byref(d), None)) for d in resident], e.get_dqs()),
                     lambda: [_lib.check(e.L.kbbq_recalibrate_batch_host(e.h, ctypes.byref(d), out.ctypes.data)) for d in resident]])
    for d in resident:
        _lib.check(e.L.kbbq_reads_free_hints(ctypes.byref(d)))
        _lib.check(e.L.kbbq_reads_free(e.h, ctypes.byref(d)))
    e.close()
    for b in host:
        b.free()
    out = None
    out_pin.free()
    h2d, d2h = up.value * 1e9, dn.value * 1e9
    h2d_dx, d2h_dx = up2.value * 1e9, dn2.value * 1e9      # with the other direction busy as well
    # per pass: bytes each way per base (bases 0.25 + N mask 0.125 + qualities 1)
    sub_bytes = [(0.375, 0.0), (1.375, 0.0), (1.375, 0.0), (1.375, 1.0)]
    once_bytes = [(1.375, 0.0), (0.0, 0.0), (0.0, 0.0), (0.0, 1.0)]

    def bound(bytes_per_pass, duplex=False):
        # a pass cannot be faster than its transfers or than its resident time; `duplex`: the two directions of pass 4 fully
        # overlapped (the link carries both at once; the engine pipelines a host batch's pass 4 in pieces), otherwise summed
        # (duplex: both directions at once, each at the rate measured WITH the other one running; a pass that moves bytes
        # in one direction only is priced at that direction's rate alone)
        return sum(max(max(nb * u / (h2d_dx if dwn else h2d), nb * dwn / (d2h_dx if u else d2h)) if duplex else nb * u / h2d + nb * dwn / d2h, c)
                   for (u, dwn), c in zip(bytes_per_pass, t_res))

    def mode(t, bytes_per_pass, text):
        b, bd = bound(bytes_per_pass), bound(bytes_per_pass, True)
        return dict(value=round(nb / sum(t) / 1e9, 4), unit="Gbases/s", seconds=round(sum(t), 3), pass_seconds=[round(x, 3) for x in t],
                    bound_Gbases_per_s=round(nb / b / 1e9, 4), fraction_of_bound=round(b / sum(t), 3),
                    duplex_bound_Gbases_per_s=round(nb / bd / 1e9, 4), fraction_of_duplex_bound=round(bd / sum(t), 3), traffic=text)

    return dict(host_link=dict(h2d_GBps=round(up.value, 2), d2h_GBps=round(dn.value, 2), how="one 1 GiB copy each way, page-locked host memory",
                               duplex_h2d_GBps=round(up2.value, 2), duplex_d2h_GBps=round(dn2.value, 2),
                               duplex_how="three 1 GiB copies each way at the same time on two streams"),
                sample="%d reads x %d bp = %.3g bases as page-locked host batches of %d reads (the command line's batch size)" % (n_reads, READ_LEN, nb, PB),
                resident_pass_seconds=[round(x, 3) for x in t_res],
                bound="per mode: sum over the four passes of max(bytes up / measured H2D rate + bytes down / measured D2H rate, the pass's resident time); duplex bound: the two directions of a pass overlapped",
                resubmit=mode(t_sub, sub_bytes, "4.5 B/base H2D (pass 1 sends no qualities), 1 B/base D2H; asynchronous submission, three batches in flight"),
                resubmit_synchronous_calls=mode(t_sync, sub_bytes, "the same through the plain calls (each returns when its batch's copy has landed)"),
                upload_once=mode(t_once, once_bytes, "1.4 B/base H2D, 1 B/base D2H"))


def host_cpu():
    """Model name, online cores and last-level cache of the box's host CPU (for the cpu_baseline line)."""
    model, llc = "unknown", "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for idx in (3, 2):
        try:
            llc = open("/sys/devices/system/cpu/cpu0/cache/index%d/size" % idx).read().strip()
            break
        except OSError:
            continue
    return model, os.cpu_count() or 0, llc


def cpu_baseline(e, genome_len, coverage):
    """Oracle (CPU restatement, 1 thread, built here with the reference's -O2 -march=native) on a bounded sample of
    the same workload."""
    from oracle import pyoracle
    native = pyoracle.use_native()
    n_reads = genome_len * coverage // READ_LEN
    sp = synth.synth_params(SEED_DATA, genome_len, n_reads, READ_LEN, n_rg=1, paired=False, n_per_million=100)
    dev = e.synth_reads(sp, 0, n_reads)
    h = e.download(dev)
    dev.free()
    nb = n_reads * READ_LEN
    codes = ((h["bases"].view(np.uint8)[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(-1)[:nb]
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    nm = np.unpackbits(h["nmask"].view(np.uint8), bitorder="little")[:nb].astype(bool)
    seq = np.where(nm, np.uint8(ord("N")), seq).astype(np.uint8)
    qual = np.ascontiguousarray(h["qual"][:nb])
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(READ_LEN)
    alpha_ld, cov, approx = plan_parameters(genome_len, coverage, None)
    o = pyoracle.Oracle(K, alpha_ld, SEED_SAMPLER, approx)
    filt_mb = (o.filter_info(0)["bits"] + o.filter_info(1)["bits"]) / 8e6
    t0 = time.perf_counter()
    o.run_all(seq, qual, off, None, None)
    dt = time.perf_counter() - t0
    model, cores_online, llc = host_cpu()
    return dict(value=nb / dt / 1e9, unit="Gbases/s", cores=1, kind="port", cpu_model=model, host_cores_online=cores_online,
                host_llc=llc, flags="-O2 -march=native" if native else "-O2 (portable build: no compiler on this host)",
                sample="oracle (CPU restatement of the reference, single-threaded like the reference's compute path), %d reads x %d bp = "
                       "%.3g bases, genome %d x %dx, k=%d, both Bloom filters %.0f MB in the reference's 512-bit layout (past the "
                       "last-level cache: %s), %.1f s, I/O excluded; the 100 Mbp slice of BASELINE.md would take ~7 min and is "
                       "available as --cpu-genome-len 100000000"
                       % (n_reads, READ_LEN, nb, genome_len, coverage, K, filt_mb, llc, dt))


def spawn_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, RCCL over xGMI unless
    KBBQ_BENCH_BACKEND says otherwise), relay rank 0's JSON line to stdout, return the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    log("[launcher] %s" % " ".join(cmd))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for ln in proc.stdout:
        if ln.startswith("{"):
            lines.append(ln)
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and len(lines) != 1:
        log("[launcher] expected one JSON line from rank 0, got %d" % len(lines))
        rc = 1
    for ln in lines[-1:]:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("KBBQ_BENCH_GENOME", 3_000_000_000)))
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--cpu-genome-len", type=int, default=8_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the bounded PCIe-inclusive leg (host batches over the boundary)")
    ap.add_argument("--pcie", action="store_true", help="(kept for older command lines: the leg now runs by default)")
    ap.add_argument("--pcie-genome-len", type=int, default=100_000_000)
    ap.add_argument("--emulate-shard", default=None, metavar="R/N",
                    help="diagnostic: one process runs rank R's shard of an N-rank job (fresh filters, no collectives) -- "
                         "the per-rank compute of the strong-scaling curve, measured on one GPU")
    ap.add_argument("--no-exclusive-step", action="store_true",
                    help="skip the extra untimed in-order step that gives the exclusive kernel durations of the roofline object")
    ap.add_argument("--force-exchange", action="store_true",
                    help="diagnostic (N = 1): a one-rank RCCL group and every exchange step at full size -- all_to_all, OR "
                         "kernel and all_gather over the whole 6.3 + 10.4 GB of filters in 512 MB slabs, histogram sum, delta-Q "
                         "broadcast -- beside the record buffers of the bucketed inserts; adds an `exchange_one_rank` object")
    ap.add_argument("--exchange", choices=("torch", "lib"), default=os.environ.get("KBBQ_EXCHANGE", "torch"),
                    help="who runs the exchange steps: torch.distributed collectives + the engine's OR kernel (kbbq_amd/dist.py: Exchange, "
                         "the default) or the library's own RCCL calls (include/kbbq_exchange.h through dist.py: LibExchange)")
    ap.add_argument("--ab-tune", default=None, metavar="KNOB[,ROUNDS]",
                    help="diagnostic (N = 1): A/B of one kbbq_engine_tune switch inside ONE process on the same resident reads -- "
                         "KNOB=0 / KNOB=1 alternated ROUNDS times (default 4), each an overlapped step (wall time) and an in-order "
                         "step (exclusive kernel durations); prints one JSON object and exits")
    ap.add_argument("--ab-set", default=None, metavar="K=V,K=V;K=V,...[;...]",
                    help="diagnostic (N = 1): several settings of kbbq_engine_tune knobs alternated inside ONE process on the same "
                         "resident reads (--ab-rounds times each, in turn): wall time of an overlapped step and of its four passes per "
                         "setting; a knob a setting does not name has its default; prints one JSON object and exits")
    ap.add_argument("--ab-rounds", type=int, default=2)
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not imported torch or the
        # engine library and never touches a GPU; the ranks are fresh CHILD processes (no exec of this one).
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a line for a different rank count"
                         % (args.gpus, world))
    _imports()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # KBBQ_BENCH_BACKEND=gloo with KBBQ_BENCH_ONE_GPU=1 runs every rank on device 0 and stages the collectives
    # through the host: a functional check of the N > 1 path on a one-GPU box (tests/test_bench_contract_gpu.py),
    # not a measurement.  The driver's runs use the default: one rank per GPU over RCCL.
    backend = os.environ.get("KBBQ_BENCH_BACKEND", "nccl")
    if os.environ.get("KBBQ_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        if world == 1:      # --force-exchange: a group of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29561")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    G, cov = args.genome_len, args.coverage
    n_reads_total = G * cov // READ_LEN
    alpha_ld, cov, approx = plan_parameters(G, cov, ALPHA)
    a, b = shard_range(n_reads_total, rank, world)
    emu = None
    if args.emulate_shard:
        if world != 1:
            raise SystemExit("--emulate-shard is a one-process diagnostic")
        emu = tuple(int(x) for x in args.emulate_shard.split("/"))
        emu_range = shard_range(n_reads_total, emu[0], emu[1])      # all reads are generated; the shard is a sub-range
    n_local = b - a
    nk_per_read = READ_LEN - K + 1
    log("[rank %d] reads %d..%d of %d, alpha %.6f, approx_kmers %d" % (rank, a, b, n_reads_total, float(alpha_ld), approx))

    e = Engine(K, alpha_ld, SEED_SAMPLER, approx, n_rg=1, max_read_len=READ_LEN, device=local_rank, profile=True)
    sp = synth.synth_params(SEED_DATA, G, n_reads_total, READ_LEN, n_rg=1, paired=False, n_per_million=100)
    t0 = time.perf_counter()
    shard = e.synth_reads(sp, a, n_local)
    log("[rank %d] generated %.3g bases on the device in %.1f s" % (rank, n_local * READ_LEN, time.perf_counter() - t0))
    # hint bit arrays (include/kbbq_engine.h: kbbq_reads.hint_*): 2 x 1 bit per base, zeroed every step
    hint_bytes = (n_local * READ_LEN // 64 + 2) * 8
    hints = torch.zeros(2 * hint_bytes, dtype=torch.uint8, device="cuda")
    shard.set_hints(hints.data_ptr(), hints.data_ptr() + hint_bytes)
    batches, ordinals, mine = [], [], []
    cuts = [0, n_local] if not emu else sorted({0, emu_range[0], emu_range[1], n_local})
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        for s in range(lo, hi, BATCH_READS):
            n = min(BATCH_READS, hi - s)
            batches.append(shard.view(s, n))
            ordinals.append((a + s) * nk_per_read)
            mine.append(bool(emu) and emu_range[0] <= s < emu_range[1])
    out_buf = torch.empty(min(BATCH_READS, n_local) * READ_LEN + 16, dtype=torch.uint8, device="cuda")
    if args.exchange == "lib" and backend == "nccl":
        from kbbq_amd.dist import LibExchange
        xch = LibExchange(e, device=local_rank, force=args.force_exchange)
    else:
        xch = Exchange(EnginePeer(e), device=torch.device("cuda", local_rank) if backend == "nccl" else None,
                       stage_host=backend != "nccl", force=args.force_exchange)

    def barrier():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if emu:
        emulate_rank(e, batches, ordinals, mine, out_buf, hints, emu, args, G, cov)
        shard.free()
        e.close()
        return
    info = None
    for _ in range(args.warmup):
        info = run_step(e, xch, batches, ordinals, out_buf, hints)

    def exclusive_step():
        """One untimed step with every kernel in order on one stream: exclusive kernel durations (see below)."""
        from kbbq_amd import _lib as _l
        _l.check(e.L.kbbq_engine_tune(e.h, b"no_overlap", 1))
        e.profile_reset()
        keep = dict(xch.ms)
        run_step(e, xch, batches, ordinals, out_buf, hints)
        barrier()
        p = e.profile()
        xch.ms.clear()
        xch.ms.update(keep)      # the exchange times quoted are the timed region's alone
        _l.check(e.L.kbbq_engine_tune(e.h, b"no_overlap", 0))
        return p

    if args.ab_set is not None:
        from kbbq_amd import _lib as _l
        settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in part.split(",") if kv) for part in args.ab_set.split(";")]
        knobs = sorted({k for st in settings for k in st})
        defaults = {"scan_blocks": 17, "walk_blocks": 17, "infer_subset": 1, "pass2_side": 2}      # (17: the engine's own choice)
        rows = []
        for i in range(args.ab_rounds):
            for st in settings:
                for k in knobs:
                    _l.check(e.L.kbbq_engine_tune(e.h, k.encode(), st.get(k, defaults.get(k, 0))))
                barrier()
                passes = []
                e.profile_reset()
                t0 = time.perf_counter()
                info = run_step(e, xch, batches, ordinals, out_buf, hints, pass_ms=passes)
                barrier()
                wall = (time.perf_counter() - t0) * 1e3
                # (event durations of kernels that share the chip with another stream's: from eligible to done, not their own cost)
                shared = {k: round(ms / n, 3) for k, (n, ms) in e.profile().items() if n}
                rows.append(dict(round=i, setting=st, step_ms=round(wall, 1), pass_ms=passes, trusted_inserted=info["trusted_inserted"],
                                 event_avg_ms=shared))
                log("[ab] %s round %d: step %.0f ms, passes %s" % (st or "default", i, wall, passes))
        for k in knobs:
            _l.check(e.L.kbbq_engine_tune(e.h, k.encode(), defaults.get(k, 0)))
        digest = 0
        for bt in batches:
            e.recalibrate(bt, out_buf.data_ptr())
            e.sync()
            digest += int(out_buf[:bt.n_bases].to(torch.int64).sum().item())
        print(json.dumps(dict(ab_set=args.ab_set, genome_len=G, coverage=cov, recal_qual_sum=digest, rows=rows)), flush=True)
        shard.free()
        e.close()
        return
    if args.ab_tune:
        # the card's clock moves a kernel by +-7 % between boxes and within minutes: both settings alternate in one job
        knob, _, rounds = args.ab_tune.partition(",")
        rounds = int(rounds or 4)
        from kbbq_amd import _lib as _l
        rows = []
        for i in range(rounds):
            for v in (0, 1):
                _l.check(e.L.kbbq_engine_tune(e.h, knob.encode(), v))
                barrier()
                t0 = time.perf_counter()
                info = run_step(e, xch, batches, ordinals, out_buf, hints)
                barrier()
                wall = (time.perf_counter() - t0) * 1e3
                p = exclusive_step()
                st = e.stats()
                rows.append(dict(round=i, value=v, step_ms=round(wall, 1), lookups=st["infer_lookups"],
                                 trusted_inserted=info["trusted_inserted"],
                                 exclusive_avg_ms={k: round(ms / n, 4) for k, (n, ms) in p.items() if n}))
                log("[ab] %s=%d round %d: step %.0f ms, k_infer %.3f ms, lookups %d" % (knob, v, i, wall, rows[-1]["exclusive_avg_ms"].get("k_infer", 0), st["infer_lookups"]))
        _l.check(e.L.kbbq_engine_tune(e.h, knob.encode(), 0))
        digest = 0
        for bt in batches:
            e.recalibrate(bt, out_buf.data_ptr())
            e.sync()
            digest += int(out_buf[:bt.n_bases].to(torch.int64).sum().item())
        print(json.dumps(dict(ab_tune=knob, genome_len=G, coverage=cov, recal_qual_sum=digest, rows=rows)), flush=True)
        shard.free()
        e.close()
        return
    # (the card slows down under sustained load -- the same kernel measures 9.0 ms on a cold card and 9.8-10.3 ms a
    # minute later -- so the exclusive step is taken on both sides of the timed region: the roofline quotes the one behind)
    want_excl = not os.environ.get("KBBQ_NO_OVERLAP") and not args.no_exclusive_step
    prof_excl_before = exclusive_step() if want_excl and args.warmup > 0 else None
    e.profile_reset()
    xch.reset_timers()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = run_step(e, xch, batches, ordinals, out_buf, hints)
    barrier()
    dt = time.perf_counter() - t0
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    prof = e.profile()
    stats = e.stats()
    # Untimed, same process, same resident reads: one more step with every kernel in order on one stream.  In the timed
    # region passes 1-3 keep two streams busy, so a kernel's event duration there includes time it shared the chip; the
    # roofline line of the dominant kernel is quoted on its EXCLUSIVE duration from this step (both are printed).
    prof_excl = exclusive_step() if want_excl else None
    # untimed: digest of the recalibrated qualities (rank-count invariant)
    digest = 0
    for bt in batches:
        e.recalibrate(bt, out_buf.data_ptr())
        e.sync()
        digest += int(out_buf[:bt.n_bases].to(torch.int64).sum().item())
    if world > 1:
        dg = torch.tensor([digest], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(dg)
        digest = int(dg.item())

    if rank == 0:
        bases_total = n_reads_total * READ_LEN
        ms_per_step = dt * 1000.0 / args.steps
        value = bases_total * args.steps / dt / 1e9
        nk_total = n_reads_total * nk_per_read
        f_t = info["trusted_inserted"] / nk_total
        alpha = float(alpha_ld)
        kernels = {}
        for name, (launches, ms) in prof.items():
            if not launches:
                continue
            ent = dict(launches=launches, total_ms=round(ms, 3), avg_ms=round(ms / launches, 4))
            per_launch_bases = n_local * READ_LEN * args.steps / launches
            per_launch_nk = n_local * nk_per_read * args.steps / launches
            model = kernel_model(name, per_launch_bases, per_launch_nk, alpha, f_t,
                                 stats["correction_queries"] / max(1, launches // args.steps))
            if model:
                ent["alg_bytes_per_launch"] = model
                ent["achieved_GBps"] = round(model / (ms / launches) / 1e6, 1)
            # passes 1 and 3 run on two streams (draw of batch i+1 beside insert of batch i; walk + tally of batch i beside
            # scan + fast path of batch i+1): the event duration of such a kernel includes time it shared the chip
            if name in PASS3_KERNELS and not os.environ.get("KBBQ_NO_OVERLAP"):
                ent["overlapped"] = True
            kernels[name] = ent
        # exclusive durations (the in-order step above) beside the timed region's
        if prof_excl:
            for name, (launches, ms) in prof_excl.items():
                if launches and name in kernels:
                    kernels[name]["exclusive_avg_ms"] = round(ms / launches, 4)
                    if "alg_bytes_per_launch" in kernels[name]:
                        kernels[name]["exclusive_GBps"] = round(kernels[name]["alg_bytes_per_launch"] / (ms / launches) / 1e6, 1)
            dom = max((k for k in kernels if "exclusive_GBps" in kernels[k]), key=lambda k: kernels[k]["exclusive_avg_ms"] * kernels[k]["launches"])
            dom_ms, dom_GBps = kernels[dom]["exclusive_avg_ms"], kernels[dom]["exclusive_GBps"]
            dom_how = ("exclusive: an untimed in-order step of this same run behind the timed region (kbbq_engine_tune no_overlap), HIP events on "
                       "the engine's stream; avg_launch_ms_before_timed_region: the same step taken behind the warm-up, on a cooler card")
        else:
            dom = max((k for k in kernels if "achieved_GBps" in kernels[k] and not kernels[k].get("overlapped")),
                      key=lambda k: kernels[k]["total_ms"])
            dom_ms, dom_GBps = kernels[dom]["avg_ms"], kernels[dom]["achieved_GBps"]
            dom_how = "the timed region (every kernel in order on one stream: KBBQ_NO_OVERLAP)"
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure
        # is the committed rocprofv3 --pmc summary of this same command at the same launch size
        traffic, traffic_note = None, "no PMC summary for this launch size"
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_pmc_latest.json")))
            k = pmc["kernels"].get(dom)
            launches_per_step = kernels[dom]["launches"] // args.steps
            full_launches = (n_local // BATCH_READS) >= 1 and READ_LEN == 150 and BATCH_READS == 1 << 22
            if pmc.get("kernel_source_sha16") != kernel_source_sha16():
                traffic_note = ("the committed PMC summary (%s) was taken on other kernel sources than the ones this run was built from: "
                                "not quoted" % pmc.get("summary_file"))
            elif k and full_launches:
                # gfx950: FETCH_SIZE tallies the L2's 128-byte memory requests at 64 bytes each (MI355X_MICROARCH.md, HBM
                # section; tools/probe_req: every random lookup is one TCC_EA0_RDREQ_128B) -- doubled here
                traffic = round((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0)
                traffic_note = ("2 x FETCH_SIZE + WRITE_SIZE per launch of %d reads, %s (rocprofv3 --pmc of this command on these kernel sources; "
                                "%d launches per step here, the last one partial); FETCH_SIZE counts the 128-byte requests of gfx950's L2 as "
                                "64 bytes" % (BATCH_READS, pmc["summary_file"], launches_per_step))
        except (OSError, ValueError, KeyError):
            pass
        # bytes the dominant kernel NEEDS: its share of the reads plus 16 bytes (one 128-bit block) per Bloom lookup it
        # really issued (counted by the kernel); the L2 fetches a 128-byte line for each, which is what `traffic` shows
        useful = None
        if dom == "k_infer" and stats.get("infer_lookups"):
            per_launch = kernels[dom]["launches"] // args.steps
            useful = round(n_local * READ_LEN * 1.25 / per_launch + 16.0 * stats["infer_lookups"] / per_launch)
        roof = dict(bound="hbm", kernel=dom, useful_bytes=useful, achieved=dom_GBps, peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(dom_GBps / HBM_PEAK_GBPS, 4), traffic=traffic,
                    algorithmic_bytes_per_launch=kernels[dom]["alg_bytes_per_launch"], avg_launch_ms=dom_ms,
                    duration_measured_in=dom_how,
                    timed_region_avg_launch_ms=kernels[dom]["avg_ms"],
                    avg_launch_ms_before_timed_region=(round(prof_excl_before[dom][1] / prof_excl_before[dom][0], 4)
                                                       if prof_excl and prof_excl_before and prof_excl_before.get(dom, (0, 0))[0] else None),
                    timed_region_note="in the timed region the kernel shares the chip with the insert side of pass 2 on the side stream; "
                                      "its event duration there is not its own cost",
                    traffic_note=traffic_note,
                    traffic_GBps=None if traffic is None else round(traffic / dom_ms / 1e6, 1),
                    note="achieved = SURVEY 8d's algorithmic bytes (64 per Bloom query) over the launch time.  The engine's blocks are 16 bytes, "
                         "but the L2 fetches a whole 128-byte line per random lookup (tools/probe_req: TCC_EA0_RDREQ_128B = lookups), so the "
                         "HBM traffic is above the algorithmic figure although 23 % of the queries are answered by hint bits; in those real "
                         "bytes the kernel runs at traffic / avg_launch_ms of HBM bandwidth, and the probe's ceiling for this access pattern is "
                         "54 G lookups/s = 6.9 TB/s.  dominant = largest total among the kernels that run alone (pass-3 kernels overlap on "
                         "two streams, flagged `overlapped`)")
        line = {
            "metric": "recalibrated Gbases/sec", "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "exchange_ms": {k: round(v / args.steps, 2) for k, v in xch.ms.items()},
            "config": {"workload": "%dx synthetic WGS reads, genome %d bp, %d bp reads, k=%d%s (%s)"
                                   % (cov, G, READ_LEN, K, "" if ALPHA is None else ", alpha " + ALPHA,
                                      "BASELINE configs[1]" if (K, ALPHA, cov) == (32, None, 30) else "another shape than BASELINE configs[1]"),
                       "reads": n_reads_total, "bases": bases_total, "batch_reads": BATCH_READS,
                       "parallelism": "reads sharded x%d, Bloom OR all-reduce + histogram sum over RCCL" % world
                       if world > 1 else "1 GPU, reads resident in HBM"},
            "roofline": roof,
            "kernels": kernels,
            "result": {"sampled_inserted": info["sampled_inserted"], "trusted_inserted": info["trusted_inserted"],
                       "fpr": info["fpr"], "corrected_reads_per_step": stats["corrected_reads"],
                       "correction_queries_per_step": stats["correction_queries"],
                       "recal_qual_sum": digest},
            # slice-bucketed inserts (kbbq_amd/csrc/bucket.h): records gathered per flush, flushes per step and filter,
            # records that overflowed a region and were inserted directly
            "bucketed_inserts": {"records_per_flush": stats["bucket_capacity"], "flushes_per_step": list(stats["bucket_flushes"]),
                                 "direct_fallback_records": stats["bucket_direct"]},
        }
        if args.force_exchange:
            line["exchange_one_rank"] = exchange_one_rank(e, xch, args.steps, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(e, args.cpu_genome_len, cov)
        if world == 1 and not args.no_pcie:
            shard.free()      # the leg needs HBM for its own resident copy
            shard = None
            line["pcie_inclusive"] = pcie_inclusive(args.pcie_genome_len, cov, local_rank)
            # the product's other shape beside `value` (reads resident): host batches uploaded once over the pinned link
            line["ingest_inclusive_value"] = line["pcie_inclusive"]["upload_once"]["value"]
            line["ingest_inclusive_unit"] = "Gbases/s"
        print(json.dumps(line), flush=True)
    if shard is not None:
        shard.free()
    e.close()
    if world > 1 or args.force_exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
