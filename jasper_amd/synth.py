"""Synthetic workloads of SURVEY.md 8(d): genome, Illumina-like reads, draft assembly with planted errors.

Two generators with the same statistical recipe:
  * numpy (CPU)  -- small inputs for tests and for the bounded cpu_baseline sample
  * torch (GPU)  -- the bench workload, generated straight into HBM so the timed region starts with the inputs
                    resident (there is no network for real genomes; sizes match BASELINE.json configs)
Recipe: genome = i.i.d. uniform ACGT, 3 % of bases overwritten by copies of 40 repeat units (300-6000 bp, 1 %
divergence); reads = 150 bp, uniform start, random strand, 0.3 % uniform substitutions, one 'N' between reads
(what Jellyfish's parser inserts, JF::include/jellyfish/mer_overlap_sequence_parser.hpp:175,205); assembly =
genome with errors at 1e-4/base (60 % substitutions, 20 % 1-bp insertions, 20 % 1-bp deletions) and a 500-bp
N-run every 10 Mb.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_genome(rng, n, repeat_frac=0.03):
    g = ACGT[rng.integers(0, 4, n)]
    if n >= 20000 and repeat_frac > 0:
        units = [ACGT[rng.integers(0, 4, int(rng.integers(300, 6001)))] for _ in range(40)]
        target = int(n * repeat_frac)
        placed = 0
        while placed < target:
            u = units[int(rng.integers(0, 40))].copy()
            if len(u) >= n:
                break
            mut = rng.random(len(u)) < 0.01
            u[mut] = ACGT[rng.integers(0, 4, int(mut.sum()))]
            p = int(rng.integers(0, n - len(u)))
            g[p:p + len(u)] = u
            placed += len(u)
    return g


def make_reads_stream(rng, genome, coverage, read_len=150, err=0.003):
    """uint8 array: read_0 'N' read_1 'N' ...   (k-mers never span reads)"""
    n = len(genome)
    nreads = int(n * coverage / read_len)
    out = np.empty((nreads, read_len + 1), dtype=np.uint8)
    out[:, read_len] = ord("N")
    bs = 200000
    ar = np.arange(read_len)
    for a in range(0, nreads, bs):
        m = min(bs, nreads - a)
        starts = rng.integers(0, n - read_len + 1, m)
        r = genome[starts[:, None] + ar[None, :]]
        e = rng.random((m, read_len)) < err
        ne = int(e.sum())
        if ne:
            # substitute with a different base
            cur = r[e]
            code = np.searchsorted(ACGT, cur) % 4
            r[e] = ACGT[(code + rng.integers(1, 4, ne)) % 4]
        flip = rng.random(m) < 0.5
        r[flip] = _COMP[r[flip][:, ::-1]]
        out[a:a + m, :read_len] = r
    return out.reshape(-1)


def make_assembly(rng, genome, err=1e-4, n_every=10_000_000, n_len=500):
    g = genome.copy()
    n = len(g)
    for p in range(n_every, n - n_len, n_every):
        g[p:p + n_len] = ord("N")
    ne = int(rng.poisson(n * err))
    pos = np.sort(rng.choice(n, size=min(ne, n), replace=False)) if ne else np.zeros(0, dtype=np.int64)
    kind = rng.random(len(pos))
    pieces = []
    last = 0
    for p, kd in zip(pos.tolist(), kind.tolist()):
        if g[p] == ord("N"):
            continue
        pieces.append(g[last:p])
        if kd < 0.6:      # substitution
            c = int(np.searchsorted(ACGT, g[p]))
            pieces.append(ACGT[[(c + int(rng.integers(1, 4))) % 4]])
            last = p + 1
        elif kd < 0.8:    # insertion before p
            pieces.append(ACGT[[int(rng.integers(0, 4))]])
            last = p
        else:             # deletion of p
            last = p + 1
    pieces.append(g[last:])
    return np.concatenate(pieces)


def chunk_records(name, seq_len, batch_size):
    """(record name, start, end) of src/jasper.sh:155 for one contig"""
    return [("%s:%d" % (name, ci), ci, min(seq_len, ci + batch_size)) for ci in range(0, seq_len, batch_size)]


def jasper_batch_size(total_bases, threads, user_batch=0, max_batch=25000000):
    """src/jasper.sh:132-138"""
    bs = int(total_bases / threads * .9)
    b = user_batch
    if bs > b:
        b = bs
        if b > max_batch:
            b = max_batch
    return b


# ---- torch / GPU versions (bench) ---------------------------------------------------------------------
def torch_genome(gen, n, device, repeat_frac=0.03):
    import torch
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    g = lut[torch.randint(0, 4, (n,), generator=gen, device=device)]
    if n >= 20000 and repeat_frac > 0:
        cpu = torch.Generator().manual_seed(int(torch.randint(0, 2**31 - 1, (1,), generator=gen, device=device).item()))
        ulen = torch.randint(300, 6001, (40,), generator=cpu).tolist()
        units = [lut[torch.randint(0, 4, (l,), generator=gen, device=device)] for l in ulen]
        target, placed = int(n * repeat_frac), 0
        choice = torch.randint(0, 40, (target // 300 + 1,), generator=cpu).tolist()
        ci = 0
        while placed < target and ci < len(choice):
            u = units[choice[ci]].clone()
            ci += 1
            if len(u) >= n:
                break
            mut = torch.rand(len(u), generator=gen, device=device) < 0.01
            u[mut] = lut[torch.randint(0, 4, (int(mut.sum().item()),), generator=gen, device=device)]
            p = int(torch.randint(0, n - len(u), (1,), generator=cpu).item())
            g[p:p + len(u)] = u
            placed += len(u)
    return g


def torch_reads_stream(gen, genome, nreads, read_len=150, err=0.003, block=1 << 20):
    """uint8 tensor of nreads*(read_len+1) bytes in HBM: read 'N' read 'N' ..."""
    import torch
    dev = genome.device
    n = genome.numel()
    out = torch.empty((nreads, read_len + 1), dtype=torch.uint8, device=dev)
    out[:, read_len] = ord("N")
    ar = torch.arange(read_len, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    code = torch.full((256,), 0, dtype=torch.int64, device=dev)
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for i, (a, b) in enumerate(zip(b"ACGT", b"TGCA")):
        code[a] = i
        comp[a] = b
    for a in range(0, nreads, block):
        m = min(block, nreads - a)
        starts = torch.randint(0, n - read_len + 1, (m,), generator=gen, device=dev)
        r = genome[starts[:, None] + ar[None, :]]
        e = torch.rand((m, read_len), generator=gen, device=dev) < err
        shift = torch.randint(1, 4, (m, read_len), generator=gen, device=dev)
        sub = lut[(code[r.long()] + shift) % 4]
        isb = (r != ord("N"))
        r = torch.where(e & isb, sub, r)
        flip = torch.rand(m, generator=gen, device=dev) < 0.5
        rc = comp[r.flip(1).long()]
        r = torch.where(flip[:, None], rc, r)
        out[a:a + m, :read_len] = r
    return out.reshape(-1)


def _write_fastq(path, reads, read_len):
    n = reads.shape[0]
    rec = np.empty((n, 2 * read_len + 7), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + read_len] = reads
    rec[:, 3 + read_len:6 + read_len] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + read_len:6 + 2 * read_len] = ord("I")
    rec[:, 6 + 2 * read_len] = ord("\n")
    rec.tofile(path)
    return n


def _write_reads_fastq(path, rng, genomes, coverage, read_len, err, workers=None):
    """make_reads_stream() of every array of `genomes` in turn + _write_fastq() of their concatenation, byte for byte and
    draw for draw (the full-size fixtures of tests/golden/ were made from these bytes) -- as a pipeline: ONE thread makes the random
    draws of every block of 200 000 reads in the generator's order (they depend on nothing but the generator and each other),
    a few threads turn a block's draws into FASTQ records (the gathers and substitutions: numpy releases the GIL in them), and the
    caller's thread writes the blocks in order.  Returns the number of reads."""
    import os, queue, threading
    if workers is None:
        workers = max(2, min(6, (os.cpu_count() or 4) // 2))
    bs = 200000
    ar = np.arange(read_len)
    tasks, done = queue.Queue(maxsize=2 * workers), queue.Queue()
    total = sum(int(len(g) * coverage / read_len) for g in genomes)
    failed = []

    def produce():
        seq = 0
        try:
            for g in genomes:
                n = len(g)
                nreads = int(n * coverage / read_len)
                for a in range(0, nreads, bs):
                    m = min(bs, nreads - a)
                    starts = rng.integers(0, n - read_len + 1, m)
                    e = rng.random((m, read_len)) < err
                    ei, ej = np.nonzero(e)
                    subs = rng.integers(1, 4, len(ei)) if len(ei) else None
                    flip = rng.random(m) < 0.5
                    tasks.put((seq, g, starts, ei, ej, subs, flip))
                    seq += 1
        except BaseException as ex:          # noqa: BLE001 -- handed to the caller
            failed.append(ex)
        for _ in range(workers):
            tasks.put(None)

    def work():
        try:
            while True:
                tk = tasks.get()
                if tk is None:
                    break
                seq, g, starts, ei, ej, subs, flip = tk
                m = len(starts)
                r = g[starts[:, None] + ar[None, :]]
                if subs is not None:
                    code = np.searchsorted(ACGT, r[ei, ej]) % 4
                    r[ei, ej] = ACGT[(code + subs) % 4]
                r[flip] = _COMP[r[flip][:, ::-1]]
                rec = np.empty((m, 2 * read_len + 7), dtype=np.uint8)
                rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
                rec[:, 3:3 + read_len] = r
                rec[:, 3 + read_len:6 + read_len] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                rec[:, 6 + read_len:6 + 2 * read_len] = ord("I")
                rec[:, 6 + 2 * read_len] = ord("\n")
                done.put((seq, rec))
        except BaseException as ex:          # noqa: BLE001
            failed.append(ex)
        done.put(None)

    threads = [threading.Thread(target=produce, daemon=True)] + [threading.Thread(target=work, daemon=True) for _ in range(workers)]
    for th in threads:
        th.start()
    held, nxt, ended, written = {}, 0, 0, 0
    with open(path, "wb") as f:
        while ended < workers:
            item = done.get()
            if item is None:
                ended += 1
                continue
            held[item[0]] = item[1]
            while nxt in held:
                rec = held.pop(nxt)
                rec.tofile(f)
                written += rec.shape[0]
                nxt += 1
    for th in threads:
        th.join()
    if failed:
        raise failed[0]
    assert not held and written == total
    return written


def contig_lengths(total, contigs):
    """deterministic contig sizes: one contig = everything; otherwise weights falling linearly 5 : 1 from the first to
    the last contig (SURVEY 8d cfg 4: "24 contigs 50-250 Mb"; cfg 3's 7 contigs get the same shape)"""
    if contigs <= 1:
        return [int(total)]
    w = np.linspace(5.0, 1.0, contigs)
    lens = np.floor(w / w.sum() * total).astype(np.int64)
    lens[0] += int(total) - int(lens.sum())
    return [int(x) for x in lens]


def write_cli_inputs(d, genome_mb=47.0, seed=2, coverage=30, read_len=150, err=0.003, contigs=1, populations=1, snp=0.001):
    """reads.fq (4-line FASTQ, header "@r", quality 'I') + asm.fa (60 columns) in directory d; byte-identical for the same
    arguments wherever numpy is the same -- the build container runs the REAL reference on them
    (tests/golden/ref_fullsize.py), the GPU box runs the drop-in, and the outputs are compared by digest.

    contigs > 1: the genome is cut into `contigs` records chr1..chrN (contig_lengths), reads never span contigs, every
    contig's assembly gets its own errors and N-runs (SURVEY 8d cfg 3 / cfg 4 shape).
    populations > 1: `populations` individuals, each the genome with its own substitutions at rate `snp`, each sequenced
    at `coverage` into its own file reads_<i>.fq (SURVEY 8d cfg 5 shape: "10 individuals x 30x each carrying 0.1 %
    private SNPs"); the draft assembly is made from the common genome.
    Returns (number of reads, assembly bases); the read files are reads.fq, or reads_0.fq .. reads_{P-1}.fq."""
    import os
    rng = np.random.default_rng(seed)
    genome = make_genome(rng, int(genome_mb * 1e6))
    if contigs <= 1 and populations <= 1:      # (the round-1 inputs: same generator calls in the same order)
        n = _write_reads_fastq(os.path.join(d, "reads.fq"), rng, [genome], coverage, read_len, err)
        asm = make_assembly(rng, genome)
        a = asm.tobytes()
        with open(os.path.join(d, "asm.fa"), "wb") as f:
            f.write(b">chr1\n")
            f.write(b"\n".join(a[i:i + 60] for i in range(0, len(a), 60)))
            f.write(b"\n")
        return n, len(asm)
    lens = contig_lengths(len(genome), contigs)
    starts = np.concatenate([[0], np.cumsum(lens)]).tolist()
    n = 0
    for p in range(populations):
        g = genome
        if populations > 1:
            g = genome.copy()
            m = rng.random(len(g)) < snp
            code = np.searchsorted(ACGT, g[m]) % 4
            g[m] = ACGT[(code + rng.integers(1, 4, int(m.sum()))) % 4]
        n += _write_reads_fastq(os.path.join(d, "reads.fq" if populations <= 1 else "reads_%d.fq" % p), rng, [g[starts[c]:starts[c + 1]] for c in range(contigs)],
                                coverage, read_len, err)
    total = 0
    with open(os.path.join(d, "asm.fa"), "wb") as f:
        for c in range(contigs):
            a = make_assembly(rng, genome[starts[c]:starts[c + 1]]).tobytes()
            total += len(a)
            f.write(b">chr%d some description\n" % (c + 1))      # (the reference keeps only the first token of a header)
            f.write(b"\n".join(a[i:i + 60] for i in range(0, len(a), 60)))
            f.write(b"\n")
    return n, total


def read_files(populations=1):
    return ["reads.fq"] if populations <= 1 else ["reads_%d.fq" % p for p in range(populations)]


def output_digests(d, asm_name="asm.fa", k=37):
    """what identifies a run's results: sha256 of the polished FASTA (records sorted by name: the reference's contig order
    is perl-hash order), of fixes.csv, of the histogram file, and the threshold"""
    import hashlib, os
    recs, name, seq = {}, None, []
    for ln in open(os.path.join(d, asm_name + ".polished.fasta")):
        ln = ln.rstrip("\n")
        if ln.startswith(">"):
            if name is not None:
                recs[name] = "".join(seq)
            name, seq = ln, []
        else:
            seq.append(ln)
    if name is not None:
        recs[name] = "".join(seq)
    h = hashlib.sha256()
    for nm in sorted(recs):
        h.update(nm.encode() + b"\n" + recs[nm].encode() + b"\n")
    out = {"polished_fasta_sha256": h.hexdigest(), "polished_bases": sum(len(v) for v in recs.values())}
    out["fixes_csv_sha256"] = hashlib.sha256(open(os.path.join(d, asm_name + ".fixes.csv"), "rb").read()).hexdigest()
    out["fixes_csv_lines"] = sum(1 for _ in open(os.path.join(d, asm_name + ".fixes.csv"), "rb"))
    out["jfhisto_sha256"] = hashlib.sha256(open(os.path.join(d, "jfhisto%d.csv" % k), "rb").read()).hexdigest()
    out["threshold"] = open(os.path.join(d, "threshold.txt")).read()
    return out
