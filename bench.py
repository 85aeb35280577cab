#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: assembly Mbp/s polished + Gk-mers/s counted, k=37.

    python bench.py --gpus N --steps K --warmup W
    (N>1: starts its own ranks, or runs under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass of the whole hot path over one batch of synthetic input that is already resident in HBM:
    new table -> count canonical 37-mers of the read shard (K1+K2) -> [N>1: the counts brought together by key owner
    over RCCL, the table kept key-sharded over the GPUs: --count] -> histogram (K3) -> threshold (src/jellyfish.py) -> P fixing passes + 1 QV pass over this rank's
    chunk records (K4-K6) -> polished text (left in HBM, like the inputs) + fix records and QV counters on the host.
After the timed region the same polish call is repeated with host buffers in and out (`polish_host_io_ms`, the
PCIe-inclusive figure) and its text is compared with the HBM-resident result.
Workload at N=1 = BASELINE.json configs[1]: "human chr21"-sized synthetic genome (47 Mb) + 30x 150-bp reads, k=37,
2 passes, chunked as `jasper.sh -t 16` would (BATCH_SIZE = int(47e6/16*.9)).  For N>1 the genome, the reads and
the assembly grow with N (weak scaling): every rank counts 1/N of the reads of the N x 47 Mb genome and polishes
its share of the chunks, with the exchange in between: the reads leave the partition passes of the counting pipeline as
region lists grouped by key owner and deduplicated, one all_to_all delivers them, the owners insert them into their shards
(--count exchange; --count local: a table per GPU whose (key, count) entries are sent to the owners instead), and the
polishing kernels then read each key from its owner's HBM (their own, or a peer's over xGMI).

Prints ONE JSON line (rank 0). `value` = assembly bases polished per second of whole-job wall time (Mbp/s);
the counting rate, the polishing-only rate, the roofline of the dominant kernels (the counting pipeline, HIP-event timed
on the table's own stream; `roofline_polish`) and a CPU baseline (the C restatement, multi-threaded, on the WHOLE workload of
a step with its results compared with the GPU's; rank 0 at N=1 only) ride along.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 37
PASSES = 2
THREADS_FOR_BATCH_RULE = 16
READ_LEN = 150
COVERAGE = 30
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_KMER = 33            # 1 B base + 16 B slot read + 16 B slot write


def build_workload(torch, dev, rank, world, genome_mb, seed):
    from jasper_amd import synth
    import numpy as np
    G = int(genome_mb * 1e6)
    # every rank builds the same N genomes (same seeds) so that read shards come from the whole genome
    genomes = []
    for j in range(world):
        gen = torch.Generator(device=dev).manual_seed(seed * 1000 + j)
        genomes.append(synth.torch_genome(gen, G, dev))
    whole = torch.cat(genomes) if world > 1 else genomes[0]
    nreads_total = int(G * world * COVERAGE / READ_LEN)
    lo = rank * nreads_total // world
    hi = (rank + 1) * nreads_total // world
    gen = torch.Generator(device=dev).manual_seed(seed * 1000 + 500 + rank)
    reads = synth.torch_reads_stream(gen, whole, hi - lo, READ_LEN, 0.003)
    # this rank's draft assembly = its own genome with planted errors; chunked like jasper.sh -t 16
    rng = np.random.default_rng(seed * 1000 + 900 + rank)
    asm = synth.make_assembly(rng, genomes[rank].cpu().numpy())
    del genomes, whole
    bs = synth.jasper_batch_size(len(asm), THREADS_FOR_BATCH_RULE)
    recs = synth.chunk_records(">chr%d" % rank, len(asm), bs)
    asm_b = asm.tobytes()
    names = [r[0][1:] for r in recs]
    seqs = [asm_b[a:b] for _, a, b in recs]
    # the chunk records back to back in HBM (what a GPU-side FASTA splitter would hand over) + their boundaries
    d_asm = torch.from_numpy(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()).to(dev)
    offs = [0]
    for q in seqs:
        offs.append(offs[-1] + len(q))
    torch.cuda.synchronize(dev)
    return reads, names, seqs, (d_asm, offs), len(asm), bs, hi - lo


def one_step(torch, dev_index, local, reads, d_chunks, world, timers, shard=None):
    """local: {"table": this rank's own table or None, "make": creates it, "exchange": try the exchange of region lists};
    shard: this rank's owner table when the merged table is kept key-sharded over the GPUs (N>1), else None"""
    from jasper_amd import polisher, dist as jdist
    t0 = time.perf_counter()
    xinfo = None
    if shard is not None and local["exchange"]:
        # no table per GPU at all: reads -> region lists grouped by owner -> ONE all_to_all -> owners insert into their shards
        # (the shard is emptied inside, once no peer can still be reading it: part of the timed path like clear() below)
        xinfo = jdist.count_sharded(shard, reads.data_ptr(), reads.numel(), torch.device("cuda", dev_index), clear=True)
        if xinfo is None:        # (decided by all ranks together) this table / input has no exchange geometry
            local["exchange"] = False
    if xinfo is not None:
        t0b = t0
        t1 = t2 = time.perf_counter()
        table = shard
        merged = xinfo["wire_bytes"]
    else:
        if local["table"] is None:
            local["table"] = local["make"]()
        table = local["table"]
        table.clear()          # a step starts from an empty table (zeroing 16 B/slot is part of the timed path)
        table.sync()
        t0b = time.perf_counter()
        table.count_bases_device(reads.data_ptr(), reads.numel())
        table.sync()
        t1 = time.perf_counter()
    kms, launches = table.count_timing()
    stages, part_launches = table.count_stages()
    path = table.count_path()
    if xinfo is None:
        merged = 0
        if world > 1:
            if shard is not None:    # one all_to_all: owner o ends up with the summed counts of the keys it owns
                merged = jdist.shard_tables(table, shard, torch.device("cuda", dev_index))
            else:                    # reduce-scatter + all-gather: the merged table on every GPU
                merged = jdist.merge_tables(table, torch.device("cuda", dev_index))
                table.sync()
        t2 = time.perf_counter()
    if world > 1:      # every rank bins the keys it owns, the bins are summed over ranks
        if shard is not None:
            h = jdist.histogram_sharded(shard, torch.device("cuda", dev_index))
        else:
            h = jdist.histogram_merged(table, torch.device("cuda", dev_index))
        rows = [(m, h[m]) for m in range(1, 10002) if h[m]]
    else:
        rows = table.histo_rows()
    txt, status = polisher.threshold_from_histo_rows(rows)
    if status != 0 or not txt:
        raise RuntimeError("synthetic histogram has no usable local minimum (threshold script would abort)")
    thr = int(txt)
    t3 = time.perf_counter()
    lookup_table = shard if shard is not None else table      # (sharded: lookups read the owner's HBM, own or peer)
    res = lookup_table.polish_batch_device(d_chunks[0], d_chunks[1], thr, PASSES, fix=True)   # returns when the GPU is done
    t4 = time.perf_counter()
    info = table.info()
    timers.append(dict(count=t1 - t0, clear=t0b - t0, merge=t2 - t1, histo=t3 - t2, polish=t4 - t3, kernel_ms=kms, launches=launches, stages=stages, part_launches=part_launches, path=path,
                       exchange=xinfo,
                       polish_dev=res.seconds, thr=thr, qv=res.qv, nfix=res.n_records, merged=merged,
                       distinct=info["distinct"], occurrences=info["occurrences"], slots=info["slots"], lookups=res.lookups,
                       segments=res.segments, respeculated=res.respeculated,
                       shard_distinct=shard.info()["distinct"] if shard is not None else None,
                       shard_slots=shard.info()["slots"] if shard is not None else None))
    return res


def host_cpu():
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return cores, model


def cpu_baseline(seed, reads_np, names, seqs, gpu):
    """The CPU restatement of the reference's algorithm (oracle/jasper_oracle.c, kind "port") on the host cores of this box,
    divided the way the reference divides its work: counting like `jellyfish count -t N`, chunk records like `xargs -P N`
    (src/jasper.sh:177,212).  With >= 8 cores it runs the WHOLE workload of the timed steps (the same reads and chunk
    records, copied out of HBM) and its results must equal the GPU's: histogram, threshold, polished text, number of fix
    records, QV counters -- the full-size check of the bench's own chunking against the oracle.  With fewer cores a
    12 Mb sample of the same recipe is timed instead (no comparison)."""
    import numpy as np
    from jasper_amd import synth, polisher
    from oracle import oracle as O
    cores, model = host_cpu()
    host_cores = cores
    cores = min(cores, 64)       # the counting threads hand keys over through threads x threads buffers: more than 64 only adds overhead
    full = cores >= 8 and reads_np is not None
    if full:
        reads = reads_np
        G = sum(len(q) for q in seqs)
        chunk_names, chunk_seqs = names, [q.decode() for q in seqs]
    else:
        rng = np.random.default_rng(seed)
        G = 12_000_000
        genome = synth.make_genome(rng, G)
        reads = synth.make_reads_stream(rng, genome, COVERAGE, READ_LEN, 0.003)
        asm = synth.make_assembly(rng, genome, err=1e-4, n_every=10_000_000)
        bs = synth.jasper_batch_size(len(asm), THREADS_FOR_BATCH_RULE)
        recs = synth.chunk_records("s", len(asm), bs)
        b = asm.tobytes().decode()
        chunk_names, chunk_seqs = [r[0] for r in recs], [b[a:e] for _, a, e in recs]
    t0 = time.perf_counter()
    db = O.OracleDB(K, threads=cores)
    nk = db.count_bases(reads)
    t1 = time.perf_counter()
    h = db.histo()
    rows = [(m, h[m]) for m in range(1, 10002) if h[m]]
    txt, status = polisher.threshold_from_histo_rows(rows)
    thr = int(txt) if (status == 0 and txt) else 2
    fixed, csv_rows, qv, _ = db.polish_batch(chunk_names, chunk_seqs, thr, PASSES)
    t2 = time.perf_counter()
    out = dict(value=round(G / 1e6 / (t2 - t0), 4), unit="Mbp/s", cores=cores, host_threads_available=host_cores, cpu=model, kind="port",
               sample=("the whole workload of a timed step" if full else "12 Mb sample of the same recipe") +
                      ": %.1f Mb assembly, %dx %d-bp reads (%d k-mers), k=%d, %d passes; oracle/jasper_oracle.c, %d threads "
                      "(reads divided like jellyfish count -t, chunk records like xargs -P)" % (G / 1e6, COVERAGE, READ_LEN, nk, K, PASSES, cores),
               count_Mkmers_per_s=round(nk / 1e6 / (t1 - t0), 3), polish_Mbp_per_s=round(G / 1e6 / (t2 - t1), 3),
               seconds=round(t2 - t0, 2))
    if full:
        same = dict(histogram=rows == gpu["rows"], threshold=thr == gpu["thr"], qv_counters=tuple(qv) == tuple(gpu["qv"]),
                    fix_csv_rows=["Contig Base_coord Original Mutation\r\n" + r for r in csv_rows] == gpu["csv"],
                    polished_text=all(f.encode() == g for f, g in zip(fixed, gpu["text"])))
        out["fix_csv_rows"] = sum(r.count("\n") for r in csv_rows)
        out["gpu_result_equals_oracle"] = same
        if not all(same.values()):
            raise RuntimeError("GPU result differs from the CPU oracle on the bench workload: %r" % (same,))
    return out


def e2e_cli():
    """UNTIMED leg (never part of `value`): the drop-in as a user runs it -- BASELINE configs[1] as FILES on local disk (2.9 GB of
    FASTQ + 47 Mb of FASTA, the deterministic inputs of tests/golden/fullsize_cfg2_t16.json), `python -m jasper_amd.cli` with
    jasper.sh's flags in a child process, files out.  Reports the wall time of that process, its own stage marks, the rate at
    which the read text became a table, and whether the outputs have the digests of the REAL reference's run on the same files
    (src/jasper.sh with Jellyfish 2.3.0: 113.6 s on the 8 vCPU of the build container)."""
    import shutil
    import subprocess
    import tempfile
    from jasper_amd import synth
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_cfg2_t16.json")))
    d = tempfile.mkdtemp(prefix="jasper_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        t0 = time.perf_counter()
        nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"], coverage=ref["coverage"])
        t_gen = time.perf_counter() - t0
        fastq = os.path.getsize(os.path.join(d, "reads.fq"))
        args = [sys.executable, "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]), "-p", str(ref["passes"])]
        runs = {}
        for label, extra in (("with_database_file", {}), ("no_database_file", {"JASPER_AMD_NO_JF": "1"})) * 2:      # (each twice: the faster run counts)
            for fn in os.listdir(d):
                if fn not in ("reads.fq", "asm.fa"):
                    os.remove(os.path.join(d, fn))
            # (device memory that a process frees is cleared in the background, and the next allocation of it waits for that:
            #  ~28 ms per GB -- a run started right behind another one's 40 GB measures the other's clean-up; docs/experiments.md)
            time.sleep(3.0)
            t0 = time.perf_counter()
            t0_epoch = time.time()
            p = subprocess.run(args, cwd=d, env=dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", **extra), capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - t0
            if p.returncode:
                raise RuntimeError("jasper_amd.cli exit %d: %s" % (p.returncode, p.stderr[-400:]))
            marks = {}
            for ln in p.stderr.splitlines():
                m = re.match(r"\[timing\] (.*?)\s+([0-9.]+) s$", ln)
                if m:
                    marks[m.group(1)] = float(m.group(2))
            got = synth.output_digests(d, k=ref["k"])
            keys = ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256")
            this = {"seconds": round(wall, 3), "seconds_until_outputs_complete": _outputs_complete_s(p.stderr, t0_epoch), "stage_seconds": marks,
                    "outputs_equal_reference": all(got[k] == ref[k] for k in keys)}
            prev = runs.get(label)
            if prev is not None:
                this["outputs_equal_reference"] = this["outputs_equal_reference"] and prev["outputs_equal_reference"]
                if prev["seconds"] < this["seconds"]:
                    this = dict(prev, outputs_equal_reference=this["outputs_equal_reference"])
            runs[label] = this
        best = runs["no_database_file"]
        count_s = next((v for k, v in best["stage_seconds"].items() if k.startswith("count reads")), None)
        return {"seconds": best["seconds"], "seconds_until_outputs_complete": best["seconds_until_outputs_complete"],
                "seconds_with_database_file": runs["with_database_file"]["seconds"],
                "outputs_equal_reference": all(r["outputs_equal_reference"] for r in runs.values()),
                "ingest_text_GBps": round(fastq / 1e9 / count_s, 2) if count_s else None, "stage_seconds": best["stage_seconds"],
                "stage_seconds_with_database_file": runs["with_database_file"]["stage_seconds"],
                "input": "%.2f GB FASTQ (%d reads) + %.1f Mb FASTA on %s (page cache warm: written %.0f s before), flags -k %d -t %d -p %d" % (
                    fastq / 1e9, nreads, asm_len / 1e6, d.rsplit("/", 1)[0], t_gen, ref["k"], ref["threads"], ref["passes"]),
                "reference_seconds_build_container_8_vcpu": ref["reference_wall_seconds"],
                "note": "untimed leg; `seconds` = wall time of the child process to its END (the faster of two runs each; like jasper.sh the command returns when its device memory has been released), seconds_until_outputs_complete = to the moment every output file was in place: interpreter start, split, count (files -> table), histogram, threshold, polish of the batch files, join, QV; the database file is written beside the stages after the counting"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _outputs_complete_s(stderr, t_start_epoch):
    """seconds from the start of the child to the moment its run() returned (every output file complete and under its final name);
    what follows in the child is the release of tens of GB of device memory at process end -- part of `seconds`, as for jasper.sh"""
    m = re.search(r"\[timing-abs\] run\(\) returned at ([0-9.]+)", stderr)
    return round(float(m.group(1)) - t_start_epoch, 3) if m else None


def e2e_cli_big(fixture="fullsize_cfg3", need_gb=40):
    """Further UNTIMED legs at larger driver-visible sizes, through `python -m jasper_amd.cli` as ONE process, against the digests
    (and QV column sums) of the real reference's run on the same deterministic files (tests/golden/<fixture>.json):
      fullsize_cfg3        BASELINE configs[2] exactly as stated -- 140 Mb in 7 contigs, 40x reads (37.3 M reads, 11.5 GB of FASTQ), -t 16
      fullsize_cfg4_share  one rank's share of configs[3] (CHM13 on 8 GPUs): 390 Mb in 3 contigs, 30x (78 M reads, 24 GB of FASTQ), -t 64:
                           a 2^32-slot table filled in 8 pieces, 36 batch files
    Skipped, with the reason, when the scratch directory has no room for the files."""
    import shutil
    import subprocess
    import tempfile
    from jasper_amd import synth
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", fixture + ".json")))
    base = os.environ.get("TMPDIR", "/tmp")
    free = shutil.disk_usage(base).free
    if free < need_gb << 30:
        return {"seconds": None, "skipped": "%s has %.0f GB free: the read files + working files need %d GB" % (base, free / 1e9, need_gb)}
    d = tempfile.mkdtemp(prefix="jasper_e2e3_", dir=base)
    try:
        t0 = time.perf_counter()
        nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"], coverage=ref["coverage"], contigs=ref["contigs"])
        t_gen = time.perf_counter() - t0
        fastq = sum(os.path.getsize(os.path.join(d, fn)) for fn in synth.read_files(1))
        args = [sys.executable, "-m", "jasper_amd.cli", "-r", " ".join(synth.read_files(1)), "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]), "-p", str(ref["passes"])]
        # twice, the faster run counts and the first one's time is reported beside it: on a GPU whose memory has never been handed out
        # the driver clears what it gives (~28 ms per GB: 2-3 s of a run that allocates 100 GB), which a second run does not pay
        # (what the first one freed is cleared in the background while this process sleeps; docs/experiments.md)
        inputs = set(os.listdir(d))
        walls = []
        keys = ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256")
        all_equal = True
        for attempt in range(3):
            # (a third run only when the first two are more than a quarter apart: the sign of allocations that waited for the driver --
            #  on some boxes BOTH of two runs back to back do, 2.1 and 3.8 s against 0.8 s)
            if attempt == 2 and max(walls) <= 1.25 * min(walls):
                break
            for fn in set(os.listdir(d)) - inputs:
                os.remove(os.path.join(d, fn))
            time.sleep(3.0 if attempt == 0 else 8.0)
            t0 = time.perf_counter()
            t0_epoch_i = time.time()
            pi = subprocess.run(args, cwd=d, env=dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", JASPER_AMD_NO_JF="1"), capture_output=True, text=True, timeout=900)
            wi = time.perf_counter() - t0
            if pi.returncode:
                raise RuntimeError("jasper_amd.cli exit %d: %s" % (pi.returncode, pi.stderr[-400:]))
            walls.append(round(wi, 3))
            goti = synth.output_digests(d, k=ref["k"])
            all_equal = all_equal and all(goti[k] == ref[k] for k in keys)      # (BOTH runs' outputs are checked)
            if attempt == 0 or wi < wall:
                p, wall, t0_epoch, got = pi, wi, t0_epoch_i, goti
        marks = {m.group(1): float(m.group(2)) for m in re.finditer(r"\[timing\] (.*?)\s+([0-9.]+) s", p.stderr)}
        qm = re.search(r"^\[qv\] before (\d+) (\d+) after (\d+) (\d+)$", p.stderr, re.M)
        count_s = next((v for k, v in marks.items() if k.startswith("count reads")), None)
        split_s = marks.get("split", 0.0)
        return {"seconds": round(wall, 3), "seconds_until_outputs_complete": _outputs_complete_s(p.stderr, t0_epoch), "seconds_of_both_runs": walls,      # (two, or three: see above)
                "outputs_equal_reference": all_equal,
                "qv_sums_equal_reference": bool(qm) and [int(qm.group(1)), int(qm.group(2))] == ref.get("qv_before") and [int(qm.group(3)), int(qm.group(4))] == ref.get("qv_after"),
                "stage_seconds": marks, "ingest_text_GBps": round(fastq / 1e9 / (count_s + split_s), 2) if count_s else None,
                "input": "%.2f GB FASTQ (%d reads) + %.1f Mb FASTA in %d contigs on %s (written %.0f s before), flags -k %d -t %d -p %d" % (
                    fastq / 1e9, nreads, asm_len / 1e6, ref["contigs"], base, t_gen, ref["k"], ref["threads"], ref["passes"]),
                "reference_seconds_build_container_8_vcpu": ref["reference_wall_seconds"]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=4)      # (a table polishes in three lanes from its second call on; the lanes' buffers are settled two calls later)
    ap.add_argument("--genome-mb", type=float, default=47.0, help="assembly size per GPU in Mb (47 = chr21-sized, BASELINE configs[1])")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selftest", action="store_true", help="N > 1: skip the transport self-test in throw-away processes")
    ap.add_argument("--no-e2e", action="store_true", help="skip the untimed files-in / files-out run of the drop-in CLI (e2e_cli)")
    ap.add_argument("--no-e2e-cfg3", action="store_true", help="skip the second untimed leg (configs[2]-sized files through the CLI)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: all ranks share GPU 0")
    ap.add_argument("--count", choices=("auto", "exchange", "local"), default="auto",
                    help="N>1 with a sharded table: 'exchange' = reads become region lists grouped by key owner, one all_to_all, owners "
                         "insert (no table per GPU); 'local' = count into a table per GPU, then sum by owner; auto = whichever the "
                         "bytes-per-link model of dist.prefer_exchange expects to be faster (with deduplicated lists: the exchange)")
    ap.add_argument("--table", choices=("auto", "sharded", "replicated"), default="auto",
                    help="N>1: keep the merged table key-sharded over the GPUs (lookups read the owner's HBM over xGMI) or replicate "
                         "it on every GPU; auto = sharded, replicated only if the peers' memory cannot be mapped")
    a = ap.parse_args()

    if os.environ.get("JASPER_BENCH_WATCHDOG"):      # debugging aid: where is every thread, every N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["JASPER_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU, RCCL rendezvous on
        # 127.0.0.1, a free port) and leave with their exit code.  Decided before anything here touches the GPU -- this
        # process never initialises HIP, and nothing is exec'ed.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    # the files-in / files-out run of the drop-in (untimed, N = 1 only) comes FIRST: its child process then has the GPU to itself,
    # as a user's run has (beside this process's tables and streams the same child took 2.1 s instead of 0.8)
    # (the larger leg first, on a GPU nobody has used yet: device memory that a process frees is cleared in the background, ~28 ms
    #  per GB, and an allocation that gets such memory waits for it -- the configs[2] run allocates ~100 GB and took 3.1 s instead
    #  of 1.3 s right behind the four configs[1] runs, all of it in its first allocations; docs/experiments.md)
    e2e3 = e2e4 = None
    if a.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.genome_mb == 47.0 and not a.no_e2e and not a.no_e2e_cfg3:
        try:        # one rank's share of configs[3] (24 GB of FASTQ, a 2^32-slot table): the largest case, on the untouched GPU
            e2e4 = e2e_cli_big("fullsize_cfg4_share", need_gb=70)
        except Exception as e:
            e2e4 = {"seconds": None, "failed": "%r" % (e,)}
        time.sleep(8.0)
        try:
            e2e3 = e2e_cli_big("fullsize_cfg3", need_gb=40)
        except Exception as e:
            e2e3 = {"seconds": None, "failed": "%r" % (e,)}
        time.sleep(6.0)
    e2e = None
    if a.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.genome_mb == 47.0 and not a.no_e2e:
        try:
            e2e = e2e_cli()
        except Exception as e:      # a report, like the CPU baseline: never a reason to lose the measurement
            e2e = {"seconds": None, "failed": "%r" % (e,)}
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, world))
        sys.exit(2)
    # N > 1 over RCCL: first contact with the transport happens in throw-away processes under a time limit (dist.transport_selftest)
    # BEFORE this rank touches its GPU.  Rank 0 runs it, a gloo group (TCP, host memory) tells everybody the verdict, and a
    # transport that hangs or delivers wrong words makes the whole job use gloo for its collectives instead of never returning.
    selftest = None
    if world > 1 and a.backend == "nccl" and not a.no_selftest:
        from jasper_amd import dist as jdist0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [jdist0.transport_selftest(world, "nccl", a.one_gpu) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        dist.barrier()
        dist.destroy_process_group()
        selftest = box[0]
        if not selftest["ok"]:
            if rank == 0:
                sys.stderr.write("bench.py: RCCL self-test failed (%s) -- collectives over gloo instead\n" % selftest.get("error"))
            a.backend = "gloo"
        elif not selftest.get("pipeline_ok", False):
            # RCCL works, the pattern of the pipelined exchange (an asynchronous all_to_all with agreements on a control group beside it)
            # did not: the rounds of the exchange one stage after the other, as before round 5
            if rank == 0:
                sys.stderr.write("bench.py: the pipelined exchange failed its self-test (%s) -- stages one after the other\n" % selftest.get("pipeline_error"))
            os.environ["JASPER_AMD_EXCHANGE_PIPELINE"] = "0"
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the product has no CPU path\n")
        sys.exit(2)
    if a.one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    reads, names, seqs, d_chunks, asm_len, bs, nreads = build_workload(torch, dev, rank, world, a.genome_mb, a.seed)
    from jasper_amd import polisher
    # size hint like jasper.sh: JF_SIZE = FASTQ bytes / 10 (src/jasper.sh:82); FASTQ ~ 2.1 bytes per base
    from jasper_amd import KmerTable, dist as jdist
    sharded = world > 1 and a.table != "replicated"
    # (a sharded run's local table only ever holds this rank's read shard; a replicated one holds everything)
    jf_size = int(nreads * (1 if sharded else world) * READ_LEN * 2.1 / 10)
    min_slots = max(1 << 21, int(1.25 * jf_size))
    exchange = sharded and a.count != "local"
    # the owner's shard: with the exchange it is the only table and is sized like the reference's -s hash for the keys it will own
    # (1/N of an N times larger genome); behind local tables it grows to its size in the first warm-up step
    shard = KmerTable(K, min_slots=(min_slots if exchange else 1 << 21), device=local) if sharded else None
    if sharded and a.count == "auto":
        # bytes per xGMI link decide (dist.prefer_exchange): a read shard of the N-fold genome at 30/N-fold coverage has
        # about N x G x (1 - exp(-lambda/N)) genomic k-mers + one k-mer in ten with a read error
        import math
        occ = nreads * (READ_LEN - K + 1)
        lam = COVERAGE * (READ_LEN - K + 1) / READ_LEN
        plan = shard.exchange_plan(reads.numel(), world)
        dedup = plan is not None and plan["p2"] >= 1 and jdist.dedupe_pays(world)
        exchange = plan is not None and jdist.prefer_exchange(world, occ, world * a.genome_mb * 1e6 * (1.0 - math.exp(-lam / world)) + 0.103 * occ, deduplicated=dedup)
        if not exchange:
            shard.close()
            shard = KmerTable(K, min_slots=1 << 21, device=local)
    loc = {"table": None, "exchange": exchange, "make": lambda: KmerTable(K, min_slots=min_slots, device=local)}
    if not exchange:
        loc["table"] = loc["make"]()        # allocated once, like the reference's -s sized hash

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    timers = []

    def step():
        nonlocal shard
        try:
            try:
                return one_step(torch, local, loc, reads, d_chunks, world, timers, shard)
            except jdist.CollectiveCountError as e:      # (count_sharded raises it on every rank together; nothing else is answered with a fallback)
                if not loc["exchange"] or a.count == "exchange":
                    raise
                if rank == 0:
                    sys.stderr.write("bench.py: %s -- counting into a table per GPU instead\n" % e)
                loc["exchange"] = False
                return one_step(torch, local, loc, reads, d_chunks, world, timers, shard)
        except jdist.ShardAttachError as e:     # raised on every rank together
            if a.table == "sharded":
                raise
            if rank == 0:
                sys.stderr.write("bench.py: %s -- replicating the merged table instead\n" % e)
            shard.close()
            shard = None
            loc["exchange"] = False
            if loc["table"] is None:
                loc["table"] = loc["make"]()
            loc["table"].reserve(max(1 << 21, int(1.25 * nreads * world * READ_LEN * 2.1 / 10)))
            return one_step(torch, local, loc, reads, d_chunks, world, timers, None)

    for i in range(a.warmup):
        step()
        if i == 0 and shard is not None and loc["table"] is not None:
            # the size hint (FASTQ bytes / 10, as jasper.sh passes to `jellyfish count -s`) is low for a read shard of an
            # N times larger genome, and growing on demand overshoots: settle the local table at load <= 1/2 once
            loc["table"].fit(0.5)
    if a.count == "exchange" and sharded and not loc["exchange"]:
        raise RuntimeError("--count exchange: this table / input size has no exchange geometry")
    # (the table's FIRST polish call runs in one lane -- what `python -m jasper_amd.cli` reaches on a genome of <= ~1 Gbase, one call
    #  per run; the timed steps are the steady state of a table that is polished repeatedly, three lanes: both are reported)
    first_call = dict(polish_ms=round(timers[0]["polish"] * 1e3, 2), polish_device_ms=round(timers[0]["polish_dev"] * 1e3, 2)) if timers else None
    timers.clear()
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(a.steps):
        res = None          # the consumer is done with the previous batch's result before the next batch starts
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    # untimed: the same batch with host buffers in and out (PCIe-inclusive), and a check that both give the same text
    host_ms = []
    for _ in range(3):
        th = time.perf_counter()
        res_h = (shard if shard is not None else loc["table"]).polish_batch(seqs, timers[-1]["thr"], PASSES, fix=True)
        host_ms.append((time.perf_counter() - th) * 1e3)
    same_text = all(bytes(res.seq_view(i)) == bytes(res_h.seq_view(i)) for i in range(len(seqs)))
    if not same_text or res.qv != res_h.qv or res.n_records != res_h.n_records:
        raise RuntimeError("HBM-resident and host-buffer polish calls disagree")
    del res_h
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([asm_len, timers[-1]["occurrences"] if world == 1 else 0, reads.numel()], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        asm_total = int(tot[0].item())
    else:
        asm_total = asm_len
    steps = max(a.steps, 1)
    ms_per_step = dt / steps * 1e3
    T = timers[-1]
    mean = lambda key: sum(t[key] for t in timers) / len(timers)
    # k-mer occurrences this rank counted per step (own shard): reads x (READ_LEN - K + 1)
    kmers_rank = nreads * (READ_LEN - K + 1)
    kernel_s = mean("kernel_ms") * 1e-3
    launches = max(int(T["launches"]), 1)
    achieved = BYTES_PER_KMER * kmers_rank / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # HBM bytes of the counting pipeline per launch, from the PMC passes committed with the same build (rocprofv3 cannot
    # collect FETCH_SIZE / WRITE_SIZE inside this process); null when that file is absent
    traffic, traffic_src = None, None
    pol_traffic, pol_traffic_src = None, None
    from jasper_amd._lib import kernel_source_digest
    for rnd in (("round5", "round4") if (world == 1 and a.genome_mb == 47.0) else ()):      # (the committed counters are those of the N=1 configs[1] run)
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", rnd, "bench_hbm_counters.json")))
            if pj.get("kernel_source_sha256") != kernel_source_digest():
                # counters of OTHER kernels are not evidence: nothing is cited (tools/prof_bench.sh + tools/summarize_prof.py re-collect them)
                traffic_src = pol_traffic_src = "profiles/%s/bench_hbm_counters.json was taken from other kernel sources than this build's: not cited" % rnd
                continue
            if "polishing" in pj:      # per polish CALL (tools/summarize_prof.py divides by the calls the profiled process made, not by any kernel's dispatches)
                pol_traffic = int(pj["polishing"]["hbm_bytes_per_call"])
                pol_traffic_src = ("profiles/%s/bench_hbm_counters.json: FETCH_SIZE + WRITE_SIZE of the polishing kernels as reported / %d polish calls of the profiled run "
                                   "(their reads are mostly 16-byte slot probes that each bring a 64-byte sector: not the wide streaming reads whose FETCH_SIZE gfx950 halves)"
                                   % (rnd, pj["polishing"]["polish_calls_profiled"]))
            cp = pj["counting_pipeline"]
            if "hbm_bytes_per_step_corrected" in cp:
                traffic = int(cp["hbm_bytes_per_step_corrected"] / launches)
                traffic_src = ("profiles/%s/bench_hbm_counters.json: 2 x FETCH_SIZE + WRITE_SIZE per step / launches (separate rocprofv3 --pmc passes; "
                               "FETCH_SIZE doubled because gfx950 reports half of the bytes of wide coalesced streaming reads, MI355X_MICROARCH.md)" % rnd)
            else:
                traffic = int(cp["hbm_bytes_per_step"] / launches)
                traffic_src = "profiles/%s/bench_hbm_counters.json (FETCH_SIZE+WRITE_SIZE as reported, separate rocprofv3 --pmc passes, per step / launches)" % rnd
            break
        except Exception:
            continue
    out = {
        "metric": "assembly Mbp/s polished + Gk-mers/s counted, k=37",
        "value": round(asm_total / 1e6 / (dt / steps), 3),
        "unit": "Mbp/s",
        "n_gpus": world, "rccl_world_size": (dist.get_world_size() if world > 1 else 1), "backend": (a.backend if world > 1 else None),
        "rccl_selftest": selftest,      # N > 1: the transport's first contact, made in throw-away processes under a time limit (null at N = 1)
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: %.0f Mb synthetic genome x %d GPU(s) + %dx %d-bp reads, k=%d, %d passes, "
                               "chunked as jasper.sh -t %d (BATCH_SIZE %d)" % ("configs[1] chr21-sized" if a.genome_mb == 47.0 else "custom size", a.genome_mb, world, COVERAGE, READ_LEN, K, PASSES,
                                                                               THREADS_FOR_BATCH_RULE, bs),
                   "chunks_per_gpu": len(seqs), "reads_per_gpu": nreads, "table_slots": T["slots"], "distinct_kmers": T["distinct"],
                   "threshold": T["thr"],
                   "parallelism": "single GPU" if world == 1 else
                                  ("read shards + chunk shards; reads partitioned into region lists by key owner, ONE all_to_all of the lists (8 B per k-mer "
                                   "occurrence), owners insert into their shards -- no table per GPU") if (shard is not None and T["exchange"]) else
                                  "read shards + chunk shards; counts summed by key owner in one all_to_all over RCCL" if shard is not None else
                                  "read shards + chunk shards; table merge (reduce-scatter + all-gather by key range) over RCCL",
                   "table": "whole" if world == 1 else
                            "owner-sharded: each GPU keeps 1/N of the keys (%d keys in %d slots here), lookups read the owner's HBM over xGMI"
                            % (T["shard_distinct"], T["shard_slots"]) if shard is not None else "replicated on every GPU"},
        "kmers_counted_Gk_per_s": round(kmers_rank * world / mean("count") / 1e9, 3),
        "polish_only_Mbp_per_s": round(asm_total / 1e6 / mean("polish"), 3),
        "phase_ms": {k: round(mean(k) * 1e3, 2) for k in ("clear", "count", "merge", "histo", "polish")},
        "count_exchange": T["exchange"],      # N>1: rounds, bytes this rank put on the wire per step, list geometry (null: tables per GPU, summed by owner)
        "polish_device_ms": round(mean("polish_dev") * 1e3, 2),
        "polish_first_call": first_call,       # one lane, buffers not yet allocated (the drop-in CLI's case); the timed steps: lanes settled
        "polish_calls_in_process": a.warmup + a.steps + 3,      # (tools/summarize_prof.py divides the polishing kernels' counters by this)
        "io": "reads and chunk records resident in HBM; polished text left in HBM; fix records, histogram and QV counters on the host",
        "polish_host_io_ms": round(min(host_ms), 2),
        "value_pcie_inclusive_polish": round(asm_total / 1e6 / (dt / steps + (min(host_ms) * 1e-3 - mean("polish"))), 3),
        "qv_counters": list(T["qv"]), "fix_records": T["nfix"], "polish_lookups": T["lookups"],
        "polish_segments": T["segments"], "polish_chunks_redone_unsegmented": T["respeculated"],
        "roofline": {"bound": "hbm",
                     "kernel": {3: "k-mer counting = part1_kernel + part2_kernel<by owner> (+ list_dedupe_kernel) on the sender, region_insert_kernel<exchange> on the owner, per round",
                                1: "k-mer counting = part1_kernel + part2f_kernel + region_insert_kernel (+ import3h_kernel for deferred records) per piece",
                                0: "count_kernel"}[T["path"]],
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": int(BYTES_PER_KMER * kmers_rank / launches),
                     "launches_per_step": launches, "partitioned_launches": T["part_launches"],
                     "avg_launch_ms": round(mean("kernel_ms") / launches, 3),
                     "kernel_ms_per_step": {n: round(sum(t["stages"][i] for t in timers) / len(timers), 3)
                                            for i, n in enumerate(KmerTable.STAGE_NAMES[T["path"]])}},
    }
    # the polishing phase against the dense model of SURVEY 8d: every base read once per scan with one 16-byte slot probe
    # ((P+1) scans x 17 B) plus one byte of output.  The walk itself is sparse (it looks at ~0.06 slots per base and scan),
    # so this figure says how the phase compares with a dense formulation at the HBM roofline, not how full the memory pipe is.
    polish_bytes = asm_len * ((PASSES + 1) * 17 + 1)
    pol_s = mean("polish_dev")
    out["roofline_polish"] = {"bound": "hbm", "kernel": "scan_classify + find_sync (pass 0), seg_walk x %d, seg_stitch, rescan" % (PASSES + 1),
                              "achieved": round(polish_bytes / pol_s / 1e9, 1) if pol_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(polish_bytes / pol_s / 1e9 / HBM_PEAK_GBS, 4) if pol_s > 0 else 0.0,
                              "algorithmic_bytes_per_step": int(polish_bytes), "device_ms": round(pol_s * 1e3, 3),
                              "lookups_per_base_and_scan": round(T["lookups"] / max(asm_len * (PASSES + 1), 1), 4), "traffic": pol_traffic,
                              "traffic_source": pol_traffic_src}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            # the last step's fix records as the rows jasper.py would write (per pass, chunk order), like polisher.polish_batch
            per_pass = [[] for _ in range(PASSES)]
            for r in res.records:
                per_pass[r["pass_"]].append((r["chunk"], r["seqno"], polisher.rows_from_record(names[r["chunk"]], r)))
            csv_gpu = [polisher.fix_csv_text([row for _, _, rr in sorted(pp, key=lambda x: (x[0], x[1])) for row in rr]) for pp in per_pass]
            gpu = dict(rows=loc["table"].histo_rows(), thr=T["thr"], qv=T["qv"], csv=csv_gpu, text=[bytes(res.seq_view(i)) for i in range(len(seqs))])
            reads_np = reads.cpu().numpy()
            try:
                out["cpu_baseline"] = cpu_baseline(a.seed, reads_np, names, seqs, gpu)
            except RuntimeError:
                raise               # a result that differs from the oracle voids the measurement
            except Exception as e:  # anything else: the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "Mbp/s", "cores": host_cpu()[0], "kind": "port", "sample": "failed: %r" % (e,)}
        if e2e is not None:
            out["e2e_cli"] = e2e
        if e2e3 is not None:
            out["e2e_cli_cfg3"] = e2e3
        if e2e4 is not None:
            out["e2e_cli_cfg4_share"] = e2e4
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()           # nobody unmaps or frees a shard that a peer may still be reading
        if shard is not None:
            shard.detach()  # every rank lets go of its peers' memory, and only then is any of it freed
            barrier()
            shard.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
