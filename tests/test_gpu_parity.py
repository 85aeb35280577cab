"""GPU parity proper: the HIP path (through the C-ABI) against the CPU oracle on seeded synthetic inputs, edge cases
of the reference's own tests (empty / ragged input, N, lower case, table growth, several files), and
size-independent properties at larger sizes.  Integer / byte work: everything must be bit-exact."""
import gzip
import os

import numpy as np
import pytest

from jasper_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def KT(hip):
    from jasper_amd import KmerTable
    return KmerTable


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def workload(seed, G, k, cov=30, rl=150, err=0.003, asm_err=1e-3):
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(rng, G)
    reads = synth.make_reads_stream(rng, genome, cov, rl, err)
    asm = synth.make_assembly(rng, genome, err=asm_err, n_every=max(G // 3, 1000), n_len=60)
    return genome, reads.tobytes(), asm.tobytes().decode()


def histo_rows(h):
    return [(m, h[m]) for m in range(1, 10002) if h[m]]


@pytest.mark.parametrize("k,G,seed", [(37, 1_200_000, 1), (25, 150_000, 2), (31, 60_000, 3), (32, 60_000, 4), (33, 60_000, 5),
                                       (15, 30_000, 6), (41, 50_000, 7),
                                       # k >= 38 in a table of few slots: the remainder does not fit the tag word and its low 64
                                       # bits live in the second array (kmer.hpp: wide_rem); `jasper.sh -k` takes any k
                                       # (src/jasper.sh:89-92, JF::include/jellyfish/mer_dna.hpp:660-669)
                                       (38, 60_000, 8), (45, 60_000, 9), (51, 400_000, 10), (63, 60_000, 11)])
def test_count_histogram_lookup_vs_oracle(KT, O, k, G, seed):
    genome, reads, asm = workload(seed, G, k)
    t = KT(k, min_slots=1 << 16)          # far too small on purpose: exercises growth / rehash between launches
    t.count_bases(reads)
    db = O.OracleDB(k)
    n_o = db.count_bases(reads)
    info = t.info()
    assert info["occurrences"] == n_o
    assert info["distinct"] == db.distinct()
    assert t.histogram() == db.histo()
    rng = np.random.default_rng(seed)
    # lookups of assembly windows incl. N runs, windows running off the end, and short / empty strings
    pos = rng.integers(0, len(asm), 3000)
    qs = [asm[p:p + k] for p in pos] + ["", "A", "N", asm[:k - 1], asm[5:5 + k].lower(), asm[:k] + "GGG"]
    assert t.lookup(qs) == [db.query(q) for q in qs]
    t.close()


def test_k64_counts_against_python_integers(KT):
    """k = 64 is the widest k-mer the 128-bit arithmetic holds (the C oracle stops at 63): counts of a small read set against
    a dictionary of Python integers -- 2-bit codes, first base in the top pair, canonical = min(mer, revcomp)
    (JF::include/jellyfish/mer_dna.hpp:38-55,428-431,525-542)"""
    import collections
    k = 64
    rng = np.random.default_rng(64)
    genome = synth.make_genome(rng, 3000, repeat_frac=0)
    reads = synth.make_reads_stream(rng, genome, 12, 120, 0.002).tobytes().decode()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    want = collections.Counter()
    mask = (1 << (2 * k)) - 1
    for r in reads.split("N"):
        for i in range(len(r) - k + 1):
            f = 0
            for ch in r[i:i + k]:
                f = (f << 2) | code[ch]
            rc = 0
            for ch in reversed(r[i:i + k]):
                rc = (rc << 2) | (3 - code[ch])
            want[min(f, rc) & mask] += 1
    t = KT(k, min_slots=1 << 12)
    t.count_bases(reads.encode())
    info = t.info()
    assert info["occurrences"] == sum(want.values()) and info["distinct"] == len(want)
    h = t.histogram()
    hw = collections.Counter(want.values())
    assert all(h[m] == hw.get(m, 0) for m in range(1, 200))
    g = genome.tobytes().decode()
    qs = [g[i:i + k] for i in range(0, len(g) - k, 37)] + ["A" * k, g[:k - 1], g[7:7 + k].lower()]

    def cnt(q):
        q = q.upper()
        q = (q + "A" * k)[:k] if all(c in code for c in q) else None
        f = 0
        for ch in q:
            f = (f << 2) | code[ch]
        rc = 0
        for ch in reversed(q):
            rc = (rc << 2) | (3 - code[ch])
        return want.get(min(f, rc), 0)
    assert t.lookup(qs) == [cnt(q) for q in qs]
    t.close()


def test_encoder_known_answers_through_the_lookup_entry_point(KT):
    """tests/golden/mer_kats.json = 162 answers of the REAL SWIG `MerDNA(s)` / `.get_canonical()` (lower case, N at various
    offsets, short and empty strings; JF::swig/mer_dna.i:12-19, JF::include/jellyfish/mer_dna.hpp:525-542).  Through the
    C-ABI: a table of each k holds every KAT's canonical k-mer exactly as often as it occurs among the KATs, and
    jasper_lookup of the RAW string s (truncate at the first non-ACGT byte, right-fill with 'A', canonicalise on the GPU)
    must return that multiplicity -- a wrong bit in the device encoder / reverse complement lands on another key (0)."""
    import collections
    import json
    import os
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mer_kats.json")))
    by_k = collections.defaultdict(list)
    for e in kats:
        by_k[e["k"]].append(e)
    assert len(by_k) >= 2
    for k, es in sorted(by_k.items()):
        mult = collections.Counter(e["canonical"] for e in es)
        t = KT(k, min_slots=1 << 12)
        t.count_bases("N".join(c for c, m in mult.items() for _ in range(m)).encode())
        assert t.info()["distinct"] == len(mult)
        got = t.lookup([e["s"] for e in es])
        assert got == [mult[e["canonical"]] for e in es], (k, [(e["s"], g) for e, g in zip(es, got) if g != mult[e["canonical"]]])
        # the uncanonicalised mer the binding reports is the same key's other strand (or itself)
        assert t.lookup([e["mer"] for e in es]) == [mult[e["canonical"]] for e in es]
        t.close()


@pytest.mark.parametrize("k,G,seed,thre,passes", [(37, 300_000, 11, 3, 2), (25, 200_000, 12, 3, 2), (31, 100_000, 13, 4, 3),
                                                    (21, 80_000, 14, 3, 1), (37, 120_000, 15, 5, 4),
                                                    (45, 200_000, 18, 3, 2), (63, 150_000, 19, 2, 2),      # wide remainders
                                                    # chunks of 230 kb: later passes find no sync points and cut at clean
                                                    # zones instead (segments chained through their arrival positions)
                                                    (37, 1_600_000, 16, 3, 2), (25, 1_200_000, 17, 3, 3)])
def test_polish_vs_oracle(KT, O, k, G, seed, thre, passes):
    from jasper_amd import polisher
    genome, reads, asm = workload(seed, G, k, asm_err=2e-3)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    db = O.OracleDB(k)
    db.count_bases(reads)
    bs = max(2 * k + 5, G // 7)
    recs = synth.chunk_records("ctg", len(asm), bs)
    names = [r[0] for r in recs] + ["tiny:0", "empty:0"]
    seqs = [asm[a:b] for _, a, b in recs] + [asm[:k + 3], ""]
    fixed_o, rows_o, qv_o, _ = db.polish_batch(names, seqs, thre, passes)
    fixed, rows, qv, res = polisher.polish_batch(t, names, seqs, thre, passes)
    assert qv == qv_o
    assert fixed == fixed_o
    for it in range(passes):
        assert polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it]
    assert sum(len(r) for r in rows) > 10
    if G >= 200_000:   # chunks long enough to be cut at sync points: the walk really ran as concurrent segments
        assert res.segments > (passes + 1) * len(seqs), (res.segments, res.respeculated)
    # without --fix (src/jasper.py:114,120: nothing is written, nothing is changed; counters still differ from the
    # fixing run because that one re-scans the text it has just edited)
    res2 = t.polish_batch(seqs, thre, passes, fix=False)
    fixed_n, rows_n, qv_n, _ = db.polish_batch(names, seqs, thre, passes, fix=False)
    assert res2.seqs == seqs == fixed_n and res2.qv == qv_n and not res2.records
    t.close()


@pytest.mark.parametrize("lanes,tight", [(2, False), (3, False), (4, False), (3, True)])
def test_polish_in_lanes_equals_one_lane_and_oracle(KT, O, lanes, tight):
    """a batch run as several lanes (groups of chunk records with a stream, a host thread and a workspace each; the default for
    batches of 4 MB and more) gives what one lane gives: text, every fix row in batch order, QV counters per chunk -- also when
    a lane's first attempt runs out of room and the whole call is repeated with more (JASPER_POLISH_TIGHT)"""
    from jasper_amd import polisher
    k, G, thre, passes = 37, 900_000, 3, 2
    genome, reads, asm = workload(31, G, k, asm_err=2e-3)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    db = O.OracleDB(k)
    db.count_bases(reads)
    # ragged chunk lengths, a tiny and an empty record in the middle of the batch
    cuts = [0, 40_000, 41_000, 250_000, 250_050, 250_050, 600_000, 820_000, len(asm)]
    seqs = [asm[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    names = ["c:%d" % a for a in cuts[:-1]]
    fixed_o, rows_o, qv_o, _ = db.polish_batch(names, seqs, thre, passes)
    os.environ["JASPER_POLISH_LANES"] = "1"
    try:
        one = t.polish_batch(seqs, thre, passes)
        os.environ["JASPER_POLISH_LANES"] = str(lanes)
        if tight:
            os.environ["JASPER_POLISH_TIGHT"] = "1"
        fixed, rows, qv, res = polisher.polish_batch(t, names, seqs, thre, passes)
    finally:
        del os.environ["JASPER_POLISH_LANES"]
        os.environ.pop("JASPER_POLISH_TIGHT", None)
    assert fixed == fixed_o == one.seqs and qv == qv_o == one.qv
    for it in range(passes):
        assert polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it]
    assert res.records == one.records and res.segments == one.segments and res.lookups == one.lookups
    assert [res.qv_chunk(i) for i in range(len(seqs))] == [one.qv_chunk(i) for i in range(len(seqs))]
    assert bool(res.retried) == tight
    t.close()


def test_path_search_scratch_slots_handed_from_wave_to_wave(KT):
    """the scratch of a path search (src/jasper.py:527-583 base_extension) is one of a pool of slots in device memory, taken and
    given back by the searching wave.  With few slots and busy memory (several lanes) a slot changes hands all the time; a
    store of the wave that gave it back must never arrive in the arrays of the wave that took it (found in round 3: the release
    was a workgroup-scope fence, and a path search returned different patches now and then -- 118 of 150 runs with 16 slots)."""
    k, thre, passes = 37, 3, 2
    genome, reads, asm = workload(31, 900_000, k, asm_err=2e-3)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    cuts = [0, 40_000, 41_000, 250_000, 250_050, 250_050, 600_000, 820_000, len(asm)]
    seqs = [asm[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    one = t.polish_batch(seqs, thre, passes)
    want = (one.seqs, one.qv, one.records, one.segments, one.lookups)
    os.environ.update(JASPER_POLISH_LANES="3", JASPER_POLISH_ROOMY="1", JASPER_POLISH_TEST_NSLOTS="8")
    try:
        for it in range(25):
            r = t.polish_batch(seqs, thre, passes)
            assert (r.seqs, r.qv, r.records, r.segments, r.lookups) == want, it
    finally:
        for v in ("JASPER_POLISH_LANES", "JASPER_POLISH_ROOMY", "JASPER_POLISH_TEST_NSLOTS"):
            os.environ.pop(v, None)
    t.close()


def test_very_many_small_chunk_records(KT, O):
    """a fragmented assembly: thousands of short records in one batch (their texts cross PCIe packed, one transfer each way, above
    256 records) -- the oracle's result, and the same as with a transfer per record"""
    from jasper_amd import polisher
    k, thre, passes = 37, 3, 2
    genome, reads, asm = workload(41, 400_000, k, asm_err=3e-3)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    db = O.OracleDB(k)
    db.count_bases(reads)
    rng = np.random.default_rng(9)
    n = 3000
    starts = rng.integers(0, len(asm) - 900, n)
    lens = rng.integers(0, 800, n)                 # (empty and shorter-than-k records among them)
    seqs = [asm[a:a + l] for a, l in zip(starts, lens)]
    names = ["f%d:0" % i for i in range(n)]
    fixed_o, rows_o, qv_o, _ = db.polish_batch(names, seqs, thre, passes)
    fixed, rows, qv, res = polisher.polish_batch(t, names, seqs, thre, passes)
    assert qv == qv_o and fixed == fixed_o
    for it in range(passes):
        assert polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it]
    os.environ["JASPER_POLISH_NO_PACKED_IO"] = "1"
    try:
        one = t.polish_batch(seqs, thre, passes)
    finally:
        del os.environ["JASPER_POLISH_NO_PACKED_IO"]
    assert one.seqs == fixed and one.qv == qv and one.records == res.records
    t.close()


def test_reads_files_formats_and_gzip(KT, O, tmp_path):
    """`zcat -f R1 R2 | jellyfish count`: one stream, format from the first byte, plain and gzip mixed"""
    k = 21
    rng = np.random.default_rng(5)
    genome = synth.make_genome(rng, 20000, repeat_frac=0)
    stream = synth.make_reads_stream(rng, genome, 10, 80, 0.01).tobytes().decode()
    reads = [r for r in stream.split("N") if r]
    half = len(reads) // 2
    fq1 = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads[:half]))
    fq2 = "".join("@s%d x\r\n%s\r\n+\r\n%s\r\n" % (i, r, "#" * len(r)) for i, r in enumerate(reads[half:]))
    p1, p2 = tmp_path / "r1.fq", tmp_path / "r2.fq.gz"
    p1.write_text(fq1)
    with gzip.open(p2, "wb") as f:
        f.write(fq2.encode())
    t = KT(k, min_slots=1 << 16)
    t.count_files([str(p1), str(p2)])
    db = O.OracleDB(k)
    db.count_text(fq1 + fq2)
    assert t.histogram() == db.histo() and t.info()["occurrences"] == sum(c for _, c in db.items())
    t.close()
    # FASTA, multi-line, with an empty record and lower case
    fa = ">a\nACGTACGTACGTACGTACGTACGTAC\nGTACGTAAACCCGGGTTT\n>b\n>c desc\nacgtacgtacgtacgtacgtacgtacgtacgtaaa\n"
    t = KT(k, min_slots=1 << 16)
    t.count_text(fa)
    db = O.OracleDB(k)
    db.count_text(fa)
    assert t.histogram() == db.histo()
    t.close()


def test_one_large_gzip_file_inflated_by_many_threads(KT, O, tmp_path, monkeypatch):
    """a .gz large enough for the many-thread reader (pgunzip.hpp: cut at block boundaries found by search, each piece inflated
    twice with two made-up windows, stitched in order) gives the same table as the text itself; with one thread the plain
    zlib reader does"""
    k = 31
    rng = np.random.default_rng(55)
    genome = synth.make_genome(rng, 400_000)
    stream = synth.make_reads_stream(rng, genome, 40, 120, 0.004).tobytes().decode()
    reads = [r for r in stream.split("N") if r]
    q = "FFFFFFFFF:,F#"
    idx = rng.integers(0, len(q), (len(reads), 120))
    fq = "".join("@SIM:%d:%d 1:N:0\n%s\n+\n%s\n" % (i // 997, i % 997, r, "".join(q[j] for j in idx[i])) for i, r in enumerate(reads))
    p = tmp_path / "big.fq.gz"
    with gzip.open(p, "wb", compresslevel=6) as f:
        f.write(fq.encode())
    assert os.path.getsize(p) > 8 << 20
    db = O.OracleDB(k)
    db.count_text(fq)
    for threads in ("6", "1"):
        monkeypatch.setenv("JASPER_INGEST_GZ_THREADS", threads)
        t = KT(k, min_slots=1 << 20)
        t.count_files([str(p)])
        assert t.histogram() == db.histo() and t.info()["distinct"] == db.distinct()
        t.close()


@pytest.mark.parametrize("ahead_mb", [None, "1"])
def test_several_gzip_files_inflated_ahead(KT, O, tmp_path, monkeypatch, ahead_mb):
    """every gzip file of a call is inflated by its own thread, ahead of the parser by a bounded budget (a tiny one here makes
    the threads block and resume many times); what the parser sees is still ONE stream, the files in order"""
    k = 25
    if ahead_mb:
        monkeypatch.setenv("JASPER_INGEST_AHEAD_MB", ahead_mb)
    monkeypatch.setenv("JASPER_INGEST_CHUNK", str(1 << 20))
    rng = np.random.default_rng(8)
    genome = synth.make_genome(rng, 120_000, repeat_frac=0)
    stream = synth.make_reads_stream(rng, genome, 200, 150, 0.005).tobytes().decode()
    reads = [r for r in stream.split("N") if r]
    cuts = [0, len(reads) // 5, len(reads) // 2, len(reads) * 3 // 4, len(reads)]
    texts, paths = [], []
    for j in range(4):
        txt = "".join("@f%d_%d\n%s\n+\n%s\n" % (j, i, r, "F" * len(r)) for i, r in enumerate(reads[cuts[j]:cuts[j + 1]]))
        texts.append(txt)
        p = tmp_path / ("part%d.fq%s" % (j, "" if j == 1 else ".gz"))        # plain file in the middle of the gzip ones
        if j == 1:
            p.write_text(txt)
        else:
            with gzip.open(p, "wb", compresslevel=1) as f:
                f.write(txt.encode())
        paths.append(str(p))
    assert min(len(x) for x in texts) > 9 << 20                               # every file is longer than the smallest look-ahead (2 x 4 MB)
    t = KT(k, min_slots=1 << 21)
    t.count_files(paths)
    ref = KT(k, min_slots=1 << 21)
    ref.count_text("".join(texts))
    assert t.info()["occurrences"] == ref.info()["occurrences"] and t.info()["distinct"] == ref.info()["distinct"]
    assert t.histogram() == ref.histogram()
    t.close()
    ref.close()


def test_counting_thread_stops_when_asked(KT, tmp_path):
    """jasper_request_cancel: a count_files call running in another thread returns "cancelled" at its next chunk (what the CLI's
    atexit hook uses when it leaves on an error elsewhere while the reads are being counted)"""
    import threading
    from jasper_amd import _lib
    rng = np.random.default_rng(5)
    genome = synth.make_genome(rng, 300_000, repeat_frac=0)
    reads = [r for r in synth.make_reads_stream(rng, genome, 60, 150, 0.003).tobytes().split(b"N") if r]
    fn = tmp_path / "r.fq"
    with open(fn, "wb") as f:
        for j, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (j, r, b"I" * len(r)))
    os.environ["JASPER_INGEST_CHUNK"] = str(64 << 10)          # (hundreds of chunks: there is a next one to stop at)
    t = KT(25, min_slots=1 << 22)
    out = {}
    try:
        _lib.lib().jasper_request_cancel(1)
        th = threading.Thread(target=lambda: out.setdefault("err", _try(lambda: t.count_files([str(fn)]))))
        th.start()
        th.join(60)
        assert not th.is_alive() and "cancelled" in str(out["err"])
        _lib.lib().jasper_request_cancel(0)                     # re-armed: the same call goes through
        t2 = KT(25, min_slots=1 << 22)
        t2.count_files([str(fn)])
        assert t2.info()["occurrences"] == sum(max(0, len(r) - 24) for r in reads)
        t2.close()
    finally:
        _lib.lib().jasper_request_cancel(0)
        del os.environ["JASPER_INGEST_CHUNK"]
    t.close()


def _try(f):
    try:
        return f()
    except Exception as e:          # noqa: BLE001
        return e


def test_reads_from_named_pipes(KT, tmp_path):
    """`-r <(zcat a.gz)`-style input: a pipe cannot be sized, seeked or looked at twice; plain and gzip content both work"""
    import threading
    k = 31
    rng = np.random.default_rng(12)
    genome = synth.make_genome(rng, 30_000, repeat_frac=0)
    reads = [r for r in synth.make_reads_stream(rng, genome, 20, 100, 0.01).tobytes().decode().split("N") if r]
    txt = "".join("@p%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
    half = txt.index("@p%d\n" % (len(reads) // 2))
    ref = KT(k, min_slots=1 << 16)
    ref.count_text(txt)
    f1, f2 = str(tmp_path / "a.fifo"), str(tmp_path / "b.fifo")
    os.mkfifo(f1)
    os.mkfifo(f2)

    def feed(path, data):
        with open(path, "wb") as f:
            f.write(data)
    ws = [threading.Thread(target=feed, args=(f1, txt[:half].encode())), threading.Thread(target=feed, args=(f2, gzip.compress(txt[half:].encode())))]
    for w in ws:
        w.start()
    t = KT(k, min_slots=1 << 16)
    t.count_files([f1, f2])
    for w in ws:
        w.join(timeout=60)
        assert not w.is_alive()
    assert t.histogram() == ref.histogram() and t.info()["occurrences"] == ref.info()["occurrences"]
    t.close()
    ref.close()


def test_k_limits(KT):
    """every k of `jasper.sh -k` up to 64 (two 64-bit words, JF::include/jellyfish/mer_dna.hpp:660-669 has no limit of its own);
    a table of few slots keeps the remainder bits the tag word cannot hold in its second array and moves to tags alone
    when it has grown enough (kmer.hpp: wide_rem); tables that other GPUs read need whole remainders in the tags: k <= 43"""
    from jasper_amd._lib import JasperHipError
    for k in (44, 51, 64):
        t = KT(k, min_slots=1 << 12)
        t.count_bases(b"ACGTTGCATGCAAGTCCGATAGGCTAACGTTTGACCATGACAGATTACAGGATCCATTGACCGTAAGGCTTAACGTA" * 2)
        assert t.info()["occurrences"] == 2 * 77 - k + 1
        with pytest.raises(JasperHipError, match="k <= 43"):
            t.ipc_handle()
        t.close()
    with pytest.raises(JasperHipError, match=r"\[1,64\]"):
        KT(0, min_slots=1 << 20)
    with pytest.raises(JasperHipError, match=r"\[1,64\]"):
        KT(65, min_slots=1 << 20)


def test_format_errors(KT):
    from jasper_amd._lib import JasperHipError
    t = KT(21, min_slots=1 << 16)
    with pytest.raises(JasperHipError, match="Unsupported format"):
        t.count_text("ACGT\n")
    with pytest.raises(JasperHipError, match="Invalid fastq"):
        t.count_text("@r\nACGTACGT\n+\nIIII\n")
    t.count_text("")                       # empty input: nothing counted, no error
    assert t.info()["occurrences"] == 0 and t.histogram() == [0] * 10002
    t.close()
    with pytest.raises(JasperHipError):
        KT(70)                             # k out of range


def test_merge_by_export_import_is_keywise_sum(KT, O):
    """the multi-GPU merge primitive: counts(A) (+) counts(B) == counts(A ++ B)   (jellyfish merge semantics)"""
    k = 37
    _, reads, _ = workload(21, 100_000, k)
    cut = reads.index(b"N", len(reads) // 2) + 1
    a, b = reads[:cut], reads[cut:]
    ta, tb = KT(k, min_slots=1 << 16), KT(k, min_slots=1 << 22)
    ta.count_bases(a)
    tb.count_bases(b)
    ent = tb.export_entries()
    assert ent.shape[0] == tb.info()["distinct"]
    ta.import_entries(ent)
    db = O.OracleDB(k)
    db.count_bases(reads)
    assert ta.histogram() == db.histo() and ta.info()["distinct"] == db.distinct()
    # linearity: importing the same entries again doubles exactly those counts
    ta.close()
    tb.import_entries(ent)
    h2 = tb.histogram()
    db2 = O.OracleDB(k)
    db2.count_bases(b)
    db2.count_bases(b)
    assert h2 == db2.histo()
    tb.close()


@pytest.mark.parametrize("k", [37, 25])      # both layouts of the 16-byte exchange entry (count beside / above the hash)
def test_packed_partition_exchange_primitives(KT, O, k):
    """the pieces of dist.merge_tables on one GPU: slot-range partitions are a disjoint cover; add-import sums; set-import
    overwrites partial counts with final ones"""
    import torch
    _, reads, _ = workload(23, 120_000, k)
    cut = reads.index(b"N", len(reads) // 2) + 1
    ta, tb, full = KT(k, min_slots=1 << 21), KT(k, min_slots=1 << 21), KT(k, min_slots=1 << 21)
    ta.count_bases(reads[:cut])
    tb.count_bases(reads[cut:])
    full.count_bases(reads)
    nparts = 3
    sizes = [tb.export_packed(0, 0, p, nparts) for p in range(nparts)]
    assert sum(sizes) == tb.info()["distinct"] == tb.export_packed(0, 0)
    # "rank a" owns partition 1: it receives b's entries of partition 1 (add) -> final there
    buf = torch.zeros((max(sizes), 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()          # the library runs on its own stream: torch's fill must be done first
    assert tb.export_packed(buf.data_ptr(), buf.shape[0], 1, nparts) == sizes[1]
    ta.import_packed(buf.data_ptr(), sizes[1], 0)
    # then publishes its final partition 1 and b SETs it
    n1 = ta.export_packed(0, 0, 1, nparts)
    out = torch.zeros((n1, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ta.export_packed(out.data_ptr(), n1, 1, nparts)
    tb.import_packed(out.data_ptr(), n1, 1)
    # partition 1 of b now equals partition 1 of the full table
    ref = torch.zeros((n1, 2), dtype=torch.int64, device="cuda")
    got = torch.zeros((n1, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert full.export_packed(ref.data_ptr(), n1, 1, nparts) == n1
    assert tb.export_packed(got.data_ptr(), n1, 1, nparts) == n1
    key = lambda x: sorted(map(tuple, x.cpu().tolist()))
    assert key(got) == key(ref) == key(out)
    for t in (ta, tb, full):
        t.close()


def test_partition_histograms_sum_to_the_full_histogram(KT):
    k = 37
    _, reads, _ = workload(33, 200_000, k)
    t = KT(k, min_slots=1 << 18)
    t.count_bases(reads)
    full = t.histogram()
    for nparts in (1, 2, 3, 8):
        acc = [0] * 10002
        for p in range(nparts):
            acc = [a + b for a, b in zip(acc, t.histogram_part(p, nparts))]
        assert acc == full, nparts
    t.close()


def test_device_resident_stream_equals_host_stream(KT):
    """jasper_count_bases_device on an HBM-resident (and deliberately misaligned) buffer == host path"""
    import torch
    k = 37
    _, reads, _ = workload(31, 150_000, k)
    t1, t2 = KT(k, min_slots=1 << 16), KT(k, min_slots=1 << 16)
    t1.count_bases(reads)
    buf = torch.empty(len(reads) + 64, dtype=torch.uint8, device="cuda")
    for shift in (0, 3):
        view = buf[shift:shift + len(reads)]
        view.copy_(torch.frombuffer(bytearray(reads), dtype=torch.uint8))
        torch.cuda.synchronize()
        t2.clear()
        t2.count_bases_device(view.data_ptr(), len(reads))
        assert t2.histogram() == t1.histogram() and t2.info()["occurrences"] == t1.info()["occurrences"]
    t1.close()
    t2.close()


def test_large_host_buffer_counts_like_resident_bases(KT):
    """jasper_count_bases with more than 1 GiB of bases in host memory: brought over in super-pieces (the cut falls inside a
    read, so k-mers span it) and counted by the atomic-free paths -- same table as the same bases resident in HBM"""
    import torch
    k = 37
    G = 37_000_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    genome = synth.torch_genome(gen, G, dev)
    nreads = G * 30 // 150
    reads = synth.torch_reads_stream(gen, genome, nreads)
    torch.cuda.synchronize()
    assert reads.numel() > (1 << 30) and (1 << 30) % 151 != 0
    a = KT(k, min_slots=1 << 28)
    a.count_bases_device(reads.data_ptr(), reads.numel())
    b = KT(k, min_slots=1 << 28)
    b.count_bases(reads.cpu().numpy().tobytes())
    assert b.info()["occurrences"] == a.info()["occurrences"] == nreads * (150 - k + 1)
    assert b.info()["distinct"] == a.info()["distinct"]
    assert b.histogram() == a.histogram()
    assert b.count_stages()[1] >= 1                              # (the first GiB took an atomic-free path, not the direct kernel on 64-MiB pieces)
    qs = [bytes(reads[i * 151:i * 151 + k].cpu().numpy()).decode() for i in range(0, nreads, nreads // 500)]
    assert a.lookup(qs) == b.lookup(qs)
    a.close()
    b.close()


def test_properties_at_scale(KT):
    """size-independent properties on a multi-Mb workload (the oracle would take minutes here)"""
    import torch
    k = 37
    G = 8_000_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    genome = synth.torch_genome(gen, G, dev)
    nreads = G * 30 // 150
    reads = synth.torch_reads_stream(gen, genome, nreads)
    torch.cuda.synchronize()
    t = KT(k, min_slots=1 << 20)
    t.count_bases_device(reads.data_ptr(), reads.numel())
    info = t.info()
    assert info["occurrences"] == nreads * (150 - k + 1)          # every window of every read, nothing across the 'N's
    h = t.histogram()
    assert sum(h) == info["distinct"]
    assert sum(m * c for m, c in enumerate(h[:10001])) <= info["occurrences"]
    # counting the same reads again doubles every count: histogram bins move from m to 2m
    t.count_bases_device(reads.data_ptr(), reads.numel())
    h2 = t.histogram()
    assert all(h2[2 * m] == h[m] for m in range(1, 5000)) and all(h2[m] == 0 for m in range(1, 10000, 2))
    t.clear()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    rng = np.random.default_rng(1)
    asm = synth.make_assembly(rng, genome.cpu().numpy(), err=2e-4).tobytes().decode()
    bs = synth.jasper_batch_size(len(asm), 16)
    seqs = [asm[a:b] for _, a, b in synth.chunk_records("c", len(asm), bs)]
    res = t.polish_batch(seqs, 2, 2)
    assert res.segments > 100 and res.respeculated <= len(seqs)
    assert res.qv[1] == sum(len(s) - k + 1 for s in seqs)
    assert res.qv[2] < res.qv[0] / 20                              # polishing removes >95 % of the bad k-mers
    # idempotence of the QV pass: scanning the polished text again reports exactly the final counters, and a
    # sequence that needs no fix is returned unchanged
    res2 = t.polish_batch(res.seqs, 2, 0, fix=False)
    assert res2.qv[0] == res.qv[2] and res2.qv[1] == res.qv[3] and res2.seqs == res.seqs
    # length bookkeeping: every record explains one unit of length change
    delta = 0
    for r in res.records:
        if r["kind"] == "i":
            delta -= r["rep"]
        elif r["kind"] == "d":
            delta += r["rep"]
        elif r["kind"] == "x":
            delta += len(r["patch"]) - len(r["orig"])
    assert sum(len(s) for s in res.seqs) - sum(len(s) for s in seqs) == delta
    t.close()


def test_polish_device_resident_equals_host_call(KT, O):
    """jasper_polish_batch_device (chunk records and polished text in HBM) gives what jasper_polish_batch gives, which
    the oracle checks; its text survives the next polish call on the same table (fetched to the host before reuse)"""
    import torch
    k = 31
    genome, reads, asm = workload(41, 400_000, k, asm_err=2e-3)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    bs = 90_000
    recs = synth.chunk_records("c", len(asm), bs)
    seqs = [asm[a:b].encode() for _, a, b in recs] + [b"", asm[:k - 1].encode()]
    offs = [0]
    for s in seqs:
        offs.append(offs[-1] + len(s))
    d = torch.frombuffer(bytearray(b"".join(seqs)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    rh = t.polish_batch(seqs, 3, 2)
    rd = t.polish_batch_device(d, offs, 3, 2)
    assert [rd.seq_len(i) for i in range(len(seqs))] == [len(s) for s in rh.seqs]
    # still in HBM: read it back through torch from the device pointer the result reports
    p0, n0 = rd.seq_device(0)
    assert n0 == len(rh.seqs[0]) and p0
    # a second call on the same table reuses the workspace: the library must have saved rd's text first
    rd2 = t.polish_batch_device(d, offs, 3, 2)
    with pytest.raises(Exception):
        rd.seq_device(0)
    for r in (rd, rd2):
        assert r.seqs == rh.seqs and r.qv == rh.qv and r.n_records == rh.n_records
        assert (r._raw == rh._raw).all() and r.aux == rh.aux
    # destroying the table first must not lose a device-resident result
    rd3 = t.polish_batch_device(d, offs, 3, 2)
    t.close()
    assert rd3.seqs == rh.seqs
    db = O.OracleDB(k)
    db.count_bases(reads)
    fixed_o, _, qv_o, _ = db.polish_batch(["c%d" % i for i in range(len(seqs))], [s.decode() for s in seqs], 3, 2)
    assert [s.decode() for s in rh.seqs] == fixed_o and rh.qv == qv_o


def test_histogram_fused_into_counting_pass_equals_table_scan(KT):
    """one partitioned counting pass over the whole input into an empty table bins the final counts while it writes
    them (region_insert_kernel); that histogram must equal the one histo_kernel reads back from the table"""
    import torch
    k = 37
    G = 12_000_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(17)
    genome = synth.torch_genome(gen, G, dev)
    nreads = G * 30 // 150
    reads = synth.torch_reads_stream(gen, genome, nreads, 150, 0.003)
    torch.cuda.synchronize()
    t = KT(k, min_slots=int(1.25 * nreads * 150 * 2.1 / 10))
    for rep in range(2):                      # second round: lazily cleared table (region images start from zeros)
        t.clear()
        t.count_bases_device(reads.data_ptr(), reads.numel())
        assert t.count_stages()[1] == 1, "expected one partitioned launch"
        assert t.histogram_is_fused()
        h_fused = t.histogram()
        t.count_bases_device(reads.data_ptr(), 0)        # any counting call drops the cached histogram
        assert not t.histogram_is_fused()
        h_scan = t.histogram()
        assert h_fused == h_scan and sum(h_fused) == t.info()["distinct"]
    # a second pass over the same reads is not "the whole input into an empty table": no fused histogram
    t.count_bases_device(reads.data_ptr(), reads.numel())
    assert not t.histogram_is_fused()
    h2 = t.histogram()
    assert all(h2[2 * m] == h_scan[m] for m in range(1, 5000))
    t.close()


WIDE_PARTITIONED = True       # tables whose slots carry a second word (B - s > 53) through the partition passes


@pytest.mark.parametrize("k,log2_slots", [(15, 0), (21, 0), (25, 0), (27, 0), (31, 0), (32, 0), (33, 0), (35, 0), (37, 0), (38, 0), (39, 25), (41, 29), (41, 0),
                                          (45, 0), (48, 0), (49, 0), (51, 0), (63, 0), (64, 0)])
def test_atomic_free_counting_equals_direct_counting(KT, k, log2_slots):
    """the two ways a table is filled must build the same table for every k (one- to four-word k-mers): one record per occurrence
    through the partition passes and LDS images (count_part.hip: part1 -> part2 -> region_insert; 8-byte records for k <= 37,
    where the hash bits below the 2^10 first-level buckets fit them, 16-byte records above) and the direct insert kernel
    (global atomics).  log2_slots: a table large enough that a slot's remainder fits its tag word (B - s <= 53) for this k."""
    import torch
    G = 1_500_000 if log2_slots < 28 else 3_200_000      # (a piece takes the partition passes when it is large relative to the table)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(100 + k)
    genome = synth.torch_genome(gen, G, dev)
    nreads = G * 30 // 150
    reads = synth.torch_reads_stream(gen, genome, nreads, 150, 0.004)
    # low-complexity stretches, a read of all-N
    reads[1000:1600] = ord("A")
    reads[5000:5400] = torch.tensor(list(b"ACACACACACACACACACAC" * 20), dtype=torch.uint8, device=dev)
    reads[9000:9300] = ord("N")
    reads[2_000_000:2_050_000] = ord("N")         # whole 16384-base tiles of the partition pass without a single k-mer
    torch.cuda.synchronize()
    slots = max(int(1.25 * nreads * 150 * 2.1 / 10), 1 << log2_slots)
    tp = KT(k, min_slots=slots)
    tp.count_bases_device(reads.data_ptr(), reads.numel())
    if k <= 37 or WIDE_PARTITIONED or 2 * k - (tp.info()["slots"].bit_length() - 1) <= 53:
        assert tp.count_stages()[1] >= 1 and tp.count_path() == 1, "partitioned path not taken"
    else:
        assert tp.count_stages()[1] == 0 and tp.count_path() == 0
    os.environ["JASPER_COUNT_DIRECT"] = "1"
    try:
        td = KT(k, min_slots=slots)
        td.count_bases_device(reads.data_ptr(), reads.numel())
        assert td.count_stages()[1] == 0
    finally:
        del os.environ["JASPER_COUNT_DIRECT"]
    idr = td.info()
    g = genome[:200_000].cpu().numpy().tobytes().decode()
    qs = [g[i:i + k] for i in range(0, len(g) - k, 997)] + ["A" * k, "ACGT" * 16, "AC" * 32]
    hd, ld = td.histogram(), td.lookup(qs)
    ip = tp.info()
    assert ip["occurrences"] == idr["occurrences"] and ip["distinct"] == idr["distinct"], (tp.count_path(), ip, idr)
    assert tp.histogram() == hd and tp.lookup(qs) == ld
    tp.close()
    # a second call adds to the table that is already there (images loaded, not started from zeros)
    t2 = KT(k, min_slots=slots)
    half = (reads.numel() // 2) // 151 * 151
    t2.count_bases_device(reads.data_ptr(), half)
    t2.count_bases_device(reads.data_ptr() + half, reads.numel() - half)
    assert t2.info()["distinct"] == idr["distinct"] and t2.histogram() == hd and t2.lookup(qs) == ld
    t2.close()
    td.close()


@pytest.mark.parametrize("k", [25, 37, 41])
def test_partition_passes_with_one_level_of_lists(KT, k):
    """a table so small (2^22 slots) that the first-level lists ARE the region lists (no second pass: region_insert reads the
    level-1 slices, whose fill counts part1 keeps slice-major and a small kernel transposes): deep coverage of a small genome"""
    import torch
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(300 + k)
    genome = synth.torch_genome(gen, 250_000, dev)
    nreads = 250_000 * 80 // 150                # (deep enough that a piece of >= 8 M bases fits the table's free room)
    reads = synth.torch_reads_stream(gen, genome, nreads, 150, 0.001)
    torch.cuda.synchronize()
    assert reads.numel() >= 16 << 20
    os.environ["JASPER_COUNT_DEBUG"] = "2"
    try:
        tp = KT(k, min_slots=1 << 22)
        tp.count_bases_device(reads.data_ptr(), reads.numel())
    finally:
        del os.environ["JASPER_COUNT_DEBUG"]
    assert tp.info()["slots"] == 1 << 22 and tp.count_stages()[1] >= 1, "no piece took the partition passes"
    os.environ["JASPER_COUNT_DIRECT"] = "1"
    try:
        td = KT(k, min_slots=1 << 22)
        td.count_bases_device(reads.data_ptr(), reads.numel())
    finally:
        del os.environ["JASPER_COUNT_DIRECT"]
    assert tp.info() == td.info() and tp.histogram() == td.histogram()
    tp.close()
    td.close()


@pytest.mark.parametrize("frac,k", [(0.02, 37), (0.10, 37), (0.5, 37), (0.02, 41), (0.10, 41), (0.02, 51), (0.5, 51)])
def test_one_kmer_that_makes_up_much_of_the_input(KT, frac, k, capfd):
    """reads of one repeated base: all their k-mers are ONE key, all its records go to one region list.  A few per cent of the
    input overflow that list into the deferred list (and reach the table through the direct path, a wave's 64 equal entries as one
    add); more than the deferred list holds makes the piece abandon itself before it touches the table, and the call counts it --
    and what follows -- through the direct kernel.  Either way the table is the one direct counting builds."""
    import torch
    dev = torch.device("cuda", 0)                 # (k = 41: 16-byte records; k = 51: those into a table with a second word per slot)
    gen = torch.Generator(device=dev).manual_seed(5)
    genome = synth.torch_genome(gen, 1_500_000, dev)
    nreads = 1_500_000 * 30 // 150
    reads = synth.torch_reads_stream(gen, genome, nreads, 150, 0.004)
    m = int(reads.numel() * frac) // 151 * 151
    reads[:m].view(-1, 151)[:, :150] = ord("A")
    torch.cuda.synchronize()
    slots = int(1.25 * nreads * 150 * 2.1 / 10)
    os.environ["JASPER_COUNT_DEBUG"] = "1"
    try:
        t = KT(k, min_slots=slots)
        t.count_bases_device(reads.data_ptr(), reads.numel())
    finally:
        del os.environ["JASPER_COUNT_DEBUG"]
    log = capfd.readouterr().err
    assert ("piece abandoned" in log) == (frac >= 0.10), log[-500:]
    assert t.count_path() == (1 if frac < 0.10 else 0)
    os.environ["JASPER_COUNT_DIRECT"] = "1"
    try:
        td = KT(k, min_slots=slots)
        td.count_bases_device(reads.data_ptr(), reads.numel())
    finally:
        del os.environ["JASPER_COUNT_DIRECT"]
    assert t.info() == td.info() and t.histogram() == td.histogram()
    assert t.lookup(["A" * k, "T" * k]) == td.lookup(["A" * k, "T" * k]) == [(m // 151) * (150 - k + 1)] * 2
    # the table keeps counting (a second call adds to it), still equal to direct counting
    t.count_bases_device(reads.data_ptr(), reads.numel())
    td.count_bases_device(reads.data_ptr(), reads.numel())
    assert t.info() == td.info() and t.histogram() == td.histogram()
    t.close()
    td.close()


def _ingest_cases():
    rng = np.random.default_rng(77)
    genome = synth.make_genome(rng, 30000, repeat_frac=0)
    stream = synth.make_reads_stream(rng, genome, 12, 90, 0.01).tobytes().decode()
    reads = [r for r in stream.split("N") if r]
    quals = lambda r, i: ("@" if i % 3 == 0 else "I") + "I" * (len(r) - 1)      # quality strings that start with '@' too
    fq = "".join("@r%d some text\n%s\n+%s\n%s\n" % (i, r, "r%d" % i if i % 2 else "", quals(r, i)) for i, r in enumerate(reads))
    fa = "".join(">r%d\n%s\n" % (i, "\n".join(r[j:j + 37] for j in range(0, len(r), 37))) for i, r in enumerate(reads))
    multi = "".join("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:40], r[40:], "I" * 40, "I" * (len(r) - 40)) for i, r in enumerate(reads[:200]))
    cases = {
        "fastq4": ([fq], "gpu"),
        "fastq4_no_final_newline": ([fq[:-1]], "gpu+tail"),
        "fastq4_two_files_joined_mid_record": ([fq[:len(fq) // 2 + 7], fq[len(fq) // 2 + 7:]], "gpu"),
        "fasta_multiline": ([fa], "gpu"),
        "fasta_with_empty_lines_and_empty_records": ([">x\n\nACGTACGTACGTACGTACGTACGTACG\n\n>y\n>z\n" + fa], "gpu"),
        "fastq_with_N_lower_and_short_reads": (["".join("@q%d\n%s\n+\n%s\n" % (i, s, "#" * len(s)) for i, s in
                                                        enumerate([r.lower() if i % 5 == 0 else r[:30] + "N" + r[31:] if i % 7 == 0 else r[:i % 25]
                                                                   for i, r in enumerate(reads)]))], "gpu"),
        "fastq_multiline_records": ([multi + fq], "host"),
        "fastq_crlf": ([fq.replace("\n", "\r\n")], "host"),
        "fastq_then_broken_record": ([fq + "@bad\nACGTACGTACGTACGTACGTACGTAC\n+\nIIII\n"], "error"),
        "fastq_empty_sequence_record": ([fq[:len(fq) // 2].rsplit("@r", 1)[0] + "@e\n\n+\n\n" + fq], "gpu"),
        "leading_blank_line": (["\n" + fq], "unsupported"),
    }
    return cases


@pytest.mark.parametrize("name", sorted(_ingest_cases()))
def test_gpu_text_ingest_equals_host_parser_and_oracle(KT, O, tmp_path, name):
    """jasper_count_reads_files parses FASTA/FASTQ text on the GPU (ingest_gpu.hip) and hands anything that is not plain
    4-line FASTQ / FASTA to the host state machine; both must give the table the oracle's parser rules give"""
    from jasper_amd._lib import JasperHipError
    k = 21
    files, expect = _ingest_cases()[name]
    paths = []
    for i, text in enumerate(files):
        p = tmp_path / ("f%d.txt" % i)
        p.write_bytes(text.encode())
        paths.append(str(p))
    whole = "".join(files)
    db = O.OracleDB(k)
    t = KT(k, min_slots=1 << 16)
    if expect in ("error", "unsupported"):
        with pytest.raises(JasperHipError, match="Invalid fastq" if expect == "error" else "Unsupported format"):
            t.count_files(paths)
        with pytest.raises(RuntimeError):
            db.count_text(whole)                       # the oracle's parser rejects it too
        t.close()
        return
    db.count_text(whole)
    t.count_files(paths)
    g, h = t.last_ingest()
    assert g + h == len(whole)
    if expect == "gpu":
        assert g > 0.9 * len(whole), (g, h)
    elif expect == "gpu+tail":
        assert g > 0.9 * len(whole) and h > 0
    else:
        assert g == 0 and h == len(whole)
    assert t.histogram() == db.histo() and t.info()["occurrences"] == sum(c for _, c in db.items())
    os.environ["JASPER_INGEST_HOST"] = "1"
    try:
        t2 = KT(k, min_slots=1 << 16)
        t2.count_files(paths)
    finally:
        del os.environ["JASPER_INGEST_HOST"]
    assert t2.histogram() == t.histogram()
    # the same stream in many small chunks: records and lines straddle chunk ends and are carried over -- with the next chunk's text
    # copied to the second device buffer while this one is parsed (the default), with the chunks taking turns on one buffer, and with
    # the files read by pread instead of out of a mapping
    for chunk, switches in ((4096, {}), (50_000, {}), (4096, {"JASPER_INGEST_OVERLAP": "0"}), (50_000, {"JASPER_INGEST_MMAP": "0"}),
                            (4096, {"JASPER_INGEST_OVERLAP": "0", "JASPER_INGEST_MMAP": "0"})):
        os.environ["JASPER_INGEST_CHUNK"] = str(chunk)
        os.environ.update(switches)
        try:
            t3 = KT(k, min_slots=1 << 16)
            t3.count_files(paths)
        finally:
            del os.environ["JASPER_INGEST_CHUNK"]
            for name_ in switches:
                del os.environ[name_]
        g3, h3 = t3.last_ingest()
        assert g3 + h3 == len(whole) and t3.histogram() == t.histogram(), (chunk, g3, h3)
        if expect == "gpu":
            assert g3 > 0.9 * len(whole)
        t3.close()
    t.close()
    t2.close()


def test_polish_many_small_chunks_vs_oracle(KT, O):
    """a batch of hundreds of short chunk records (contigs of 0.1-6 kb, some shorter than k, some empty): the batched
    kernels index chunks through blockIdx.y and the candidate list is shared by the whole batch"""
    from jasper_amd import polisher
    k = 25
    genome, reads, asm = workload(23, 400_000, k, asm_err=3e-3)
    rng = np.random.default_rng(9)
    names, seqs, pos = [], [], 0
    while pos < len(asm) and len(seqs) < 400:
        n = int(rng.choice([0, 7, k - 1, k, k + 1, 120, 800, 2500, 6000]))
        names.append("c%d" % len(seqs))
        seqs.append(asm[pos:pos + n])
        pos += max(n, 1)
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    db = O.OracleDB(k)
    db.count_bases(reads)
    fixed_o, rows_o, qv_o, _ = db.polish_batch(names, seqs, 3, 2)
    fixed, rows, qv, res = polisher.polish_batch(t, names, seqs, 3, 2)
    assert qv == qv_o and fixed == fixed_o
    for it in range(2):
        assert polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it]
    assert sum(len(r) for r in rows) > 50
    t.close()


def test_polish_retries_with_more_room_when_a_bound_is_exceeded(KT, O):
    """slack for growing text, record and scratch bounds are guesses; a call that exceeds one fails cleanly inside the
    library and is repeated with 8x the room -- the caller sees the same result as always"""
    from jasper_amd import polisher
    k = 25
    genome, reads, asm = workload(29, 200_000, k, asm_err=3e-3)
    recs = synth.chunk_records("c", len(asm), 30_000)
    names = [r[0] for r in recs]
    seqs = [asm[a:b] for _, a, b in recs]
    t = KT(k, min_slots=1 << 20)
    t.count_bases(reads)
    want = polisher.polish_batch(t, names, seqs, 3, 2)
    os.environ["JASPER_POLISH_TIGHT"] = "1"        # gap buffers with 2 bytes of slack: any net insertion of 3+ bases overflows
    try:
        got = polisher.polish_batch(t, names, seqs, 3, 2)
    finally:
        del os.environ["JASPER_POLISH_TIGHT"]
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert got[3].retried and not want[3].retried
    t.close()
