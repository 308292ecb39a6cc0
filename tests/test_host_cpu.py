"""CPU tests of the product's host side: the C-ABI library loads without a GPU and exports every
symbol the header declares; its host-only entry points (filter construction, thresholds, delta-Q
model, RNG jump-ahead, packing, synthetic reads) agree with the oracle.  No device compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

import common
from kbbq_amd import _lib, synth
from kbbq_amd.dist import shard_range
from kbbq_amd.engine import long_double_text, plan_parameters, rng_state_at
from kbbq_amd.reads import ReadBatch, pack_bits, unpack_bits
from oracle import pyoracle


def test_library_exports_every_declared_symbol():
    header = "".join(open(os.path.join(common.ROOT, "include", h)).read() for h in ("kbbq_engine.h", "kbbq_bgzf.h", "kbbq_exchange.h"))
    declared = set(re.findall(r"\b(kbbq_[a-z0-9_]+)\s*\(", header))
    declared -= {"kbbq_engine", "kbbq_params", "kbbq_reads", "kbbq_bgzf", "kbbq_group"}
    assert len(declared) >= 50
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, "not exported: %s" % missing
    unbound = [n for n in sorted(declared) if n not in _lib.SYMBOLS]
    assert not unbound, "declared in the header but not bound in _lib.SYMBOLS: %s" % unbound
    assert _lib.lib() is not None


def test_struct_layouts_match_the_header(tmp_path):
    # the C compiler's view of the ABI structs against the ctypes mirrors
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "kbbq_engine.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(kbbq_params), sizeof(kbbq_reads), sizeof(kbbq_filter_info),
           sizeof(kbbq_synth_params), sizeof(kbbq_profile_entry), sizeof(kbbq_covariates), sizeof(kbbq_dq),
           offsetof(kbbq_params, max_read_len), offsetof(kbbq_reads, read_len), offsetof(kbbq_filter_info, salt));
    return 0;
}
''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(common.ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.Params), ctypes.sizeof(_lib.Reads), ctypes.sizeof(_lib.FilterInfo),
            ctypes.sizeof(_lib.SynthParams), ctypes.sizeof(_lib.ProfileEntry), ctypes.sizeof(_lib.Covariates),
            ctypes.sizeof(_lib.Dq), _lib.Params.max_read_len.offset, _lib.Reads.read_len.offset, _lib.FilterInfo.salt.offset]
    assert got == want


def test_pack_bases_and_bits():
    rng = np.random.RandomState(0)
    seq = rng.choice(np.frombuffer(b"ACGTNacgtRYn=0123", dtype=np.uint8), size=1003).astype(np.uint8)
    qual = rng.randint(0, 256, size=1003).astype(np.uint8)
    off = np.array([0, 100, 100, 400, 1003], dtype=np.uint64)
    b = ReadBatch(seq, qual, off)
    L = pyoracle.lib()
    codes = np.array([L.ko_base_code(int(c)) for c in seq])
    got = ((b.bases.view(np.uint8)[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(-1)[:1003]
    assert np.array_equal(got[codes < 4], codes[codes < 4])
    assert np.all(got[codes == 4] == 0)
    assert np.array_equal(unpack_bits(b.nmask, 1003), (codes == 4).astype(np.uint8))
    bits = rng.randint(0, 2, size=777).astype(np.uint8)
    assert np.array_equal(unpack_bits(pack_bits(bits), 777), bits)
    assert b.max_len == 603 and b.n_kmer_positions(32) == 69 + 0 + 269 + 572


def test_rng_jump_ahead_matches_the_serial_stream():
    L = pyoracle.lib()
    n = 5000
    outs = (ctypes.c_uint64 * n)()
    L.ko_rng_outputs(777, n, outs)
    M = (1 << 64) - 1

    def first_output(s):
        x = (s[1] * 5) & M
        x = ((x << 7) | (x >> 57)) & M
        return (x * 9) & M
    for ordinal in (0, 1, 2, 63, 64, 65, 255, 256, 1023, 1024, 4095, 4999):
        assert first_output(rng_state_at(777, ordinal)) == outs[ordinal]
    # composition: jumping a then b equals jumping a+b (checked through a far offset computed two ways)
    a = rng_state_at(12345, (1 << 40) + 12345)
    b = rng_state_at(12345, (1 << 40) + 12346)

    def step(s):
        s = list(s)
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M
        return s
    assert step(a) == b


def test_bernoulli_threshold_is_the_exact_integer_form_of_the_draw_rule():
    L, O = _lib.lib(), pyoracle.lib()
    for p in (0.35, 7.0 / 30.0, 0.05, 0.5, 1e-12, 0.9999999999999999):
        always = ctypes.c_int32()
        T = L.kbbq_host_bernoulli_threshold(p, ctypes.byref(always))
        assert not always.value
        for u in (0, 1, T - 2, T - 1, T, T + 1, (1 << 64) - 1, (1 << 63), int(p * 2 ** 64)):
            if 0 <= u < (1 << 64):
                assert bool(O.ko_bernoulli_one(u, p)) == (u < T), (p, u, T)
    always = ctypes.c_int32()
    L.kbbq_host_bernoulli_threshold(1.0, ctypes.byref(always))
    assert always.value == 1
    assert L.kbbq_host_bernoulli_threshold(0.0, ctypes.byref(always)) == 0 and not always.value


@pytest.mark.parametrize("approx,fpr", [(700000, float(np.longdouble("0.01"))), (700000, float(np.longdouble("0.0005"))),
                                         (21_000_000_000, float(np.longdouble("0.0005"))), (123457, 0.05)])
def test_host_filter_spec_equals_oracle(approx, fpr):
    L = _lib.lib()
    info = _lib.FilterInfo()
    pats = np.zeros(65536 * 8, dtype=np.uint64)
    small = approx < 10 ** 9
    _lib.check(L.kbbq_host_filter_spec(approx, fpr, _lib.DEFAULT_BLOOM_SEED, ctypes.byref(info),
                                       pats.ctypes.data_as(_lib.c_u64p)))
    if small:
        o = pyoracle.Oracle(32, 0.35, 1, approx, fpr, fpr)
        ref = o.filter_info(0)
        assert (info.bits, info.bits_unblocked, info.n_hash, info.n_salt, info.random_seed) == (
            ref["bits"], ref["bits_unblocked"], ref["nhash"], ref["nsalt"], ref["random_seed"])
        assert list(info.salt[:info.n_salt]) == [int(x) for x in ref["salts"]]
        assert np.array_equal(pats, o.filter_patterns(0))
    else:
        nh, bits = ctypes.c_uint32(), ctypes.c_uint64()
        pyoracle.lib().ko_optimal_parameters(approx, fpr, ctypes.byref(nh), ctypes.byref(bits))
        assert (info.n_hash, info.bits_unblocked) == (nh.value, bits.value)
        assert info.bits % 512 == 0 and 0 <= info.bits - info.bits_unblocked < 512
        # the 30x WGS trusted filter is ~3.322e11 bits (41.5 GB)
        assert abs(info.bits / 3.322e11 - 1) < 0.01


def test_invalid_filter_parameters_are_rejected():
    L = _lib.lib()
    info = _lib.FilterInfo()
    for approx, fpr, seed in ((0, 0.01, 1), (1000, -0.5, 1), (1000, 0.01, 0), (1000, 0.01, 0xFFFFFFFFFFFFFFFF)):
        assert L.kbbq_host_filter_spec(approx, fpr, seed, ctypes.byref(info), None) == -22


def test_host_thresholds_equal_oracle():
    L, O = _lib.lib(), pyoracle.lib()
    cases = [(32, 67150848, 5553859, 7, np.longdouble("0.35")), (32, 67150848, 5553859, 7, np.longdouble(7) / np.longdouble(30)),
             (21, 10 ** 9, 31234567, 7, np.longdouble("0.05")), (32, 2 * 10 ** 11, 16607190899, 7, np.longdouble(7) / np.longdouble(30)),
             (15, 5 * 10 ** 6, 4 * 10 ** 6, 7, np.longdouble("0.09")),
             (32, 2048, 14413, 7, np.longdouble("0.7"))]      # saturated: more elements than bits, fpr reported as 1
    for k, bits, ins, nsalt, alpha in cases:
        thr = np.zeros(k + 1, dtype=np.int32)
        fpr = ctypes.c_double()
        buf = ctypes.create_string_buffer(64)
        rc = L.kbbq_host_thresholds(k, bits, ins, nsalt, long_double_text(alpha), thr.ctypes.data_as(_lib.c_i32p),
                                    ctypes.byref(fpr), buf, 64)
        assert rc in (0, 1)
        ref_fpr = ctypes.create_string_buffer(64)
        O.ko_effective_fpp_text(bits, ins, nsalt, ref_fpr, 64)
        assert float(ref_fpr.value) == fpr.value
        ref_p = ctypes.create_string_buffer(64)
        O.ko_phit_text(bits, ins, nsalt, long_double_text(alpha), ref_p, 64)
        assert buf.value == ref_p.value
        ref_thr = (ctypes.c_int32 * (k + 1))()
        O.ko_thresholds(k, ref_p.value, ref_thr)
        assert list(thr) == list(ref_thr)
        assert rc == (1 if fpr.value > 0.15 else 0)
    # the SURVEY KAT through the product's own code
    thr = np.zeros(33, dtype=np.int32)
    buf = ctypes.create_string_buffer(64)
    fpr = ctypes.c_double()
    L.kbbq_host_thresholds(32, 67150848, 5553859, 7, b"0.35", thr.ctypes.data_as(_lib.c_i32p), ctypes.byref(fpr), buf, 64)
    assert buf.value == b"0.579494101768392170918"
    assert list(thr)[-5:] == [23, 23, 24, 25, 25]


def test_host_model_equals_oracle_on_random_histograms():
    L = _lib.lib()
    rng = np.random.RandomState(11)
    R, C = 2, 40
    NQ = _lib.NQ
    cyc = np.zeros((R, NQ, 2, C, 2), dtype=np.uint64)
    di = np.zeros((R, NQ, 16, 2), dtype=np.uint64)
    for r in range(R):
        for q in (2, 7, 12, 22, 33, 37, 41, 93, 94, 130, 255):      # (above KBBQ_MAXQ = 93: rows the reference's growing tables would hold too)
            for s in range(2):
                n_c = rng.randint(5, C + 1)
                tot = rng.randint(0, 5000, size=n_c)
                err = (tot * 10 ** (-q / 10.0) * rng.uniform(0.3, 3.0, size=n_c)).astype(np.int64)
                cyc[r, q, s, :n_c, 1] = tot
                cyc[r, q, s, :n_c, 0] = np.minimum(err, tot)
            if q >= 6:
                tot = rng.randint(0, 20000, size=16)
                di[r, q, :, 1] = tot
                di[r, q, :, 0] = (tot * 10 ** (-q / 10.0) * rng.uniform(0.3, 3.0, size=16)).astype(np.int64)
    q_tot = cyc.sum(axis=(2, 3))
    cov = dict(R=R, C=C, rg=q_tot.sum(axis=1), q=q_tot, cycle=cyc, dinuc=di)
    o = pyoracle.Oracle(32, 0.35, 1, 1000)
    o.set_covariates(cov)
    ref = o.train()
    c = _lib.Covariates()
    c.n_rg, c.n_cycle = R, C
    c.cycle, c.dinuc = cyc.ctypes.data, di.ctypes.data
    out = dict(meanq=np.zeros(R, np.int32), rg=np.zeros(R, np.int32), q=np.zeros((R, NQ), np.int32),
               cycle=np.zeros((R, NQ, 2, C), np.int32), dinuc=np.zeros((R, NQ, 16), np.int32))
    d = _lib.Dq()
    d.meanq, d.rgdq, d.qdq, d.cycledq, d.dinucdq = (out[k].ctypes.data for k in ("meanq", "rg", "q", "cycle", "dinuc"))
    _lib.check(L.kbbq_host_train(ctypes.byref(c), ctypes.byref(d)))
    for key in ("meanq", "rg", "q", "cycle", "dinuc"):
        assert np.array_equal(out[key], ref[key]), key
    assert np.abs(out["cycle"]).max() > 0 and np.abs(out["dinuc"]).max() > 0


def test_plan_parameters_follow_the_cli():
    alpha, cov, approx = plan_parameters(3_000_000_000, 30)
    assert cov == 30 and approx == 21_000_000_000 and abs(float(alpha) - 7 / 30) < 1e-15
    alpha, cov, approx = plan_parameters(1_000_000, 0, seqlen=20_000_000)     # kbbq.cc:244-251
    assert cov == 20 and approx == 7_000_000
    # kbbq.cc:122,254-256: alpha parsed by std::stold, coverage = uint(7.0l / alpha) in long double
    alpha, cov, approx = plan_parameters(1_000_000, 0, alpha="0.05")
    assert cov == 140 and approx == 7_000_000
    alpha, cov, approx = plan_parameters(1_000_000, 0, alpha="0.25")
    assert cov == 28 and approx == 7_000_000


def test_synthetic_reads_are_seeded_and_sliceable():
    sp = synth.synth_params(42, 100000, 5000, 150, n_rg=3, paired=True, n_per_million=500)
    a = synth.generate(sp, 0, 600)
    b = synth.generate(sp, 200, 400)
    assert np.array_equal(a["seq"][200 * 150:], b["seq"]) and np.array_equal(a["qual"][200 * 150:], b["qual"])
    assert np.array_equal(a["rg"][200:], b["rg"]) and np.array_equal(a["second"][200:], b["second"])
    assert set(np.unique(a["qual"])) <= {2, 12, 22, 32, 37}
    assert a["rg"].max() == 2 and set(a["second"]) == {0, 1}
    sp2 = synth.synth_params(43, 100000, 5000, 150)
    assert not np.array_equal(synth.generate(sp2, 0, 10)["seq"], a["seq"][:1500])


def test_shard_ranges_partition_the_reads():
    for n_reads, n in ((600_000_000, 8), (1001, 4), (31, 2), (64, 1), (5, 8)):
        cover = []
        for r in range(n):
            a, b = shard_range(n_reads, r, n)
            assert a % 32 == 0 or a == n_reads
            cover.append((a, b))
        assert cover[0][0] == 0 and cover[-1][1] == n_reads
        assert all(cover[i][1] == cover[i + 1][0] for i in range(n - 1))


def test_engine_block_layout_holds_everything_the_reference_can_set():
    """The engine keeps 128 of the 512 bits of a block (include/kbbq_engine.h: kbbq_filter_device_table):
    every pattern of both pattern tables, and every table the oracle builds, must survive squeeze -> expand
    unchanged, and a bit outside the reachable set must be refused."""
    L = _lib.lib()
    info = _lib.FilterInfo()
    for fpr in (0.01, 0.0005):
        pat = np.zeros(65536 * 8, dtype=np.uint64)
        assert L.kbbq_host_filter_spec(700000, fpr, _lib.DEFAULT_BLOOM_SEED, ctypes.byref(info), pat.ctypes.data_as(_lib.c_u64p)) == 0
        assert info.table_bytes == info.n_blocks * 16
        small = np.zeros(65536 * 2, dtype=np.uint64)
        assert L.kbbq_host_blocks_squeeze(pat.ctypes.data_as(_lib.c_u64p), 65536, small.ctypes.data_as(_lib.c_u64p)) == 0
        back = np.zeros_like(pat)
        assert L.kbbq_host_blocks_expand(small.ctypes.data_as(_lib.c_u64p), 65536, back.ctypes.data_as(_lib.c_u64p)) == 0
        assert np.array_equal(back, pat)
        # 512 sampled bit numbers fold 4-to-1 onto 128 places, so some patterns have fewer than n_salt bits
        assert 65536 * 2 < int(np.unpackbits(small.view(np.uint8)).sum()) == int(np.unpackbits(pat.view(np.uint8)).sum()) < 65536 * info.n_salt
    # the bit positions: word w keeps bytes w&3 and 4+(w&3)
    one = np.zeros(8, dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    for w in range(8):
        for bit in range(64):
            one[:] = 0
            one[w] = np.uint64(1) << np.uint64(bit)
            rc = L.kbbq_host_blocks_squeeze(one.ctypes.data_as(_lib.c_u64p), 1, out.ctypes.data_as(_lib.c_u64p))
            reachable = ((bit >> 3) & 3) == (w & 3)
            assert rc == (0 if reachable else -34), (w, bit)
            if reachable:
                c = ((bit >> 5) << 3) | (bit & 7)
                assert int(out[w >> 2]) == 1 << (16 * (w & 3) + c) and int(out[1 - (w >> 2)]) == 0
    # an oracle run's tables
    import common
    d = common.make_dataset(seed=3, genome_len=5000, coverage=20, read_len=100)
    ora = common.run_oracle(d)
    for key in ("sampled_table", "trusted_table"):
        t = ora[key]
        small = np.zeros(len(t) // 4, dtype=np.uint64)
        assert L.kbbq_host_blocks_squeeze(t.ctypes.data_as(_lib.c_u64p), len(t) // 8, small.ctypes.data_as(_lib.c_u64p)) == 0
        back = np.zeros_like(t)
        L.kbbq_host_blocks_expand(small.ctypes.data_as(_lib.c_u64p), len(t) // 8, back.ctypes.data_as(_lib.c_u64p))
        assert np.array_equal(back, t) and t.any()


def test_block_index_is_the_exact_remainder():
    """hash % n_blocks (get_block, bloom.hh:99-105) through the two-multiply Barrett form the kernels use for
    n_blocks <= 2^31 and the 64-bit fastmod above that: every divisor class, hashes at the edges."""
    L = _lib.lib()
    rng = np.random.RandomState(7)
    divisors = [1, 2, 3, 4, 5, 7, 255, 256, 257, 13116, 21630, 131154, 216294, 393216000, 649000000, 2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1,
                3 * 2 ** 30 + 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 40]
    divisors += [int(x) for x in rng.randint(1, 2 ** 31, size=200, dtype=np.int64)] + [int(x) for x in rng.randint(2 ** 31, 2 ** 33, size=50, dtype=np.int64)]
    for d in divisors:
        hs = [0, 1, d - 1 if d <= 2 ** 32 else 5, d % 2 ** 32, (d + 1) % 2 ** 32, (2 * d - 1) % 2 ** 32, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1]
        hs += [int(x) for x in rng.randint(0, 2 ** 32, size=300, dtype=np.int64)]
        hs += [(2 ** 32 // d) * d % 2 ** 32, ((2 ** 32 // d) * d - 1) % 2 ** 32] if d <= 2 ** 32 else []
        for h in hs:
            assert L.kbbq_host_block_index(h, d) == h % d, (h, d)


def test_bench_refuses_a_rank_count_it_was_not_launched_with():
    """bench.py --gpus N under a launcher with another WORLD_SIZE must not print a line (checked before any GPU use)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr and not out.stdout.strip()
