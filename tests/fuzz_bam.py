"""Randomised differential run of the BAM path on the device against the host codec (kbbq_amd/csrc/bam_io.cc through
`kbbq --io-test bam`): random record counts, read lengths (1 .. 70 000: records longer than a segment), tags, aligned and
unaligned records, header sizes, BGZF block sizes, chunk cuts, --use-oq; batches compared word for word, and pass 4's
rewrite (random new qualities, --set-oq or not) against the rules.   python tests/fuzz_bam.py N_CASES SEED"""
import ctypes
import os
import sys
import tempfile
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bamutil  # noqa: E402
import common  # noqa: E402,F401
import test_bam_gpu as T  # noqa: E402
from kbbq_amd import _lib, bgzf  # noqa: E402
from test_bgzf_cpu import bgzf_blocks  # noqa: E402


def one_case(rng, tmp):
    import torch
    use_oq = bool(rng.randint(0, 2))
    set_oq = bool(rng.randint(0, 2))
    shape = rng.randint(0, 4)
    if shape == 0:
        lens, n = [int(rng.randint(1, 400)) for _ in range(7)], int(rng.randint(50, 3000))
    elif shape == 1:
        lens, n = [150], int(rng.randint(500, 6000))
    elif shape == 2:
        lens, n = [int(rng.choice([5, 150, 33000, 40000, 70000, 1])) for _ in range(5)], int(rng.randint(5, 60))
    else:
        lens, n = None, int(rng.randint(100, 4000))
    recs = T.make_records(n, seed=int(rng.randint(0, 1 << 30)), lens=lens, oq_every=1 if use_oq else int(rng.randint(0, 4)),
                          big_quals=bool(rng.randint(0, 2)) and shape != 2)
    n_refs = int(rng.choice([2, 2, 50, 3000]))
    refs = tuple(("contig_%05d" % i, 1000 + i) for i in range(n_refs))
    text = T.HEADER_TEXT + ("".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) if rng.randint(0, 2) else "")
    comp, head_len, n_ref = T.bam_file(recs, text=text, refs=refs, ragged=int(rng.randint(0, 1 << 20)) if rng.randint(0, 3) else None)
    path = os.path.join(tmp, "f.bam")
    open(path, "wb").write(comp)
    rows, rc = T.host_rows(path, use_oq)
    assert rc == -1 and len(rows) == len(recs), (rc, len(rows), len(recs))
    n_cuts = int(rng.randint(0, 4))
    # (a piece must hold at least one whole BGZF block: the caller's contract; the command line's pieces are 64 KB and more)
    cuts = []
    for x in sorted(int(x) for x in rng.randint(70000, max(70001, len(comp) - 70000), n_cuts)) if len(comp) > 210000 else []:
        if not cuts or x - cuts[-1] >= 70000:
            cuts.append(x)
    reader = bgzf.BamReader(head_len, n_ref, T.RG_IDS, use_oq=use_oq)
    writer = bgzf.BgzfWriter()
    newq = rng.randint(0, 94, sum(len(r["seq"]) for r in recs)).astype(np.uint8)
    got, at_rec, at_base = [], 0, 0
    for info in T.feed(reader, comp, cuts):
        assert info["flags"] == 0, info
        if not info["n_records"]:
            continue
        d = reader.batch()
        got.append((info, T.download_batch(d), T.download_rg(d)))
        _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
        nrec, nb = info["n_records"], info["n_bases"]
        dq = torch.from_numpy(newq[at_base:at_base + nb].copy()).cuda()
        torch.cuda.synchronize()
        reader.write(writer, dq.data_ptr(), set_oq=set_oq)
        blob, _ = writer.collect()
        assert b"".join(bgzf_blocks(blob)) == T.rewritten(recs[at_rec:at_rec + nrec], newq[at_base:at_base + nb], set_oq), "rewrite differs"
        at_rec += nrec
        at_base += nb
    T.check_batches(got, rows)
    reader.close()
    writer.close()
    return dict(n=n, shape=int(shape), use_oq=use_oq, set_oq=set_oq, cuts=len(cuts), comp=len(comp))


def main():
    n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.RandomState(seed)
    failures = 0
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(n_cases):
            try:
                info = one_case(rng, tmp)
                if c % 10 == 0:
                    print("case", c, info, flush=True)
            except Exception:
                failures += 1
                print("case", c, "FAILED", flush=True)
                traceback.print_exc()
    print("failures", failures)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
