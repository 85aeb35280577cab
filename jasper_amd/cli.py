"""Drop-in driver for `jasper.sh` (src/jasper.sh): same flags, batch/pass semantics, working-directory artefacts,
sentinel files and log lines; the Jellyfish + per-batch python processes are replaced by the HBM table and the GPU
polisher.  Lines are cited as src/jasper.sh:N.

    python -m jasper_amd.cli -r 'R1.fq R2.fq' -a asm.fa -k 37 -t 16 -p 2 [--gpus N]

Differences that are deliberate and documented in DESIGN.md:
  * contigs are written to <asm>.polished.fasta in input order (the reference's order is perl-hash random, :220)
  * column sums for the QV are exact integers (gawk behaviour)
  * `mer_counts$K.jf` is written like the reference does (:177 `tee $JF_DB`) -- as a Jellyfish binary/sorted file that
    jellyfish 2.3.0 and the reference's own jasper.py read -- unless JASPER_AMD_NO_JF=1; an existing one, or -j, is read
    into HBM
"""
import datetime
import glob
import os
import re
import sys

from . import polisher, qv
from .table import KmerTable

MAX_BATCH_SIZE = 25000000  # :9


def _tty():
    try:
        return os.isatty(1)
    except Exception:
        return False


_T0 = [None]


def _timing(label):
    """JASPER_AMD_TIMING=1: seconds since the previous mark, on stderr (not part of the reference's output)"""
    import time
    if not os.environ.get("JASPER_AMD_TIMING"):
        return
    now = time.perf_counter()
    if _T0[0] is not None:
        sys.stderr.write("[timing] %-28s %.3f s\n" % (label, now - _T0[0]))
    _T0[0] = now


_QUIET = [False]      # multi-GPU runs: only rank 0 talks


def log(msg):  # :30-33
    if _QUIET[0]:
        return
    d = datetime.datetime.now().astimezone().strftime("%a %b %d %H:%M:%S %Z %Y")
    if _tty():
        print("\033[0;32m[%s]\033[0m %s" % (d, msg), flush=True)
    else:
        print("[%s] %s" % (d, msg), flush=True)


def error_exit(msg, code=1):  # :35-39
    d = datetime.datetime.now().astimezone().strftime("%a %b %d %H:%M:%S %Z %Y")
    if not _QUIET[0]:
        sys.stderr.write("[%s] %s\n" % (d, msg))
    sys.exit(code)


USAGE = """JASPER version 1.0.2
Usage: jasper.sh [options]
Options:
Options (default value in (), *required):
-b, --batch=uint64              Desired batch size for the query (default value based on number of threads and assembly size)
-t, --threads=uint32            Number of threads (2)
-a, --assembly=path             *Assembly file
-j, --jf=path                   Jellyfish k-mer count database file. Required if --reads is not provided
-r|--reads=path                 File(s) containing the polishing reads. If two or more files are provided, please enclose the list with single-quotes, e.g. -r '/path_to/file1.fa /path_to/file2.fa'. Required if -j (--jf) is not provided
-k|--kmer=uint64                k-mer size (37)
-p|--num_passes=uint16          Number of polishing iterations (3), not recommended to increase much past 4
-h|--help                       This message
-v|--verbose                    Verbose (False)
-d|--debug                      Debug mode. If supplied, all intermediate output files are kept"""


class Options:
    def __init__(self):
        self.num_threads = "2"       # :6
        self.batch_size = "0"        # :8
        self.passes = "2"            # :10
        self.kmer = "37"             # :11
        self.jf_size = 0
        self.debug = False
        self.query = "random.fa"
        self.query_fn = "random.fa"
        self.reads = "random.fastq"
        self.jf_db = None
        self.verbose = False
        self.device = 0


def parse_args(argv):
    """the `case` parser of src/jasper.sh:58-110, including `-d` swallowing the following argument (:93-96)"""
    o = Options()
    i = 0
    while i < len(argv):
        key = argv[i]
        nxt = argv[i + 1] if i + 1 < len(argv) else ""
        if key in ("-b", "--batch"):
            o.batch_size = nxt; i += 1
        elif key in ("-t", "--threads"):
            o.num_threads = nxt; i += 1
        elif key in ("-a", "--assembly"):
            o.query = nxt; o.query_fn = os.path.basename(nxt); i += 1
        elif key in ("-j", "--jf"):
            o.jf_db = nxt; i += 1
        elif key in ("-r", "--reads"):
            o.reads = nxt
            tot = 0
            for f in nxt.split():
                try:
                    tot += os.stat(f).st_size
                except OSError:
                    pass
            o.jf_size = int(tot / 10)                                  # :82
            i += 1
        elif key in ("-p", "--num_passes"):
            o.passes = nxt; i += 1
        elif key in ("-k", "--kmer"):
            o.kmer = nxt; i += 1
        elif key in ("-d", "--debug"):
            o.debug = True; i += 1                                     # the extra `shift`
        elif key in ("-v", "--verbose"):
            o.verbose = True
        elif key in ("-h", "--help", "-u", "--usage"):
            print(USAGE)
            sys.exit(0)
        elif key == "--device":                                        # extension: which GPU
            o.device = int(nxt); i += 1
        elif key == "--gpus":                                          # extension: handled by main() (one process per GPU)
            i += 1
        else:
            print("Unknown option %s" % key)
            sys.exit(1)
        i += 1
    return o


_SAFE_BODY = bytes(range(0x21, 0x7f)) + b"\n"     # printable ASCII and '\n': anything else in a sequence line and the line-by-line rules decide


def _header_spans(data):
    """(start, end) of every line of `data` that starts with '>' (end = past its '\n', or the end of the data)"""
    spans = []
    pos = 0 if data.startswith(b">") else data.find(b"\n>") + 1
    while pos > 0 or (pos == 0 and data.startswith(b">") and not spans):
        end = data.find(b"\n", pos)
        end = len(data) if end < 0 else end + 1
        spans.append((pos, end))
        nxt = data.find(b"\n>", end - 1)
        if nxt < 0:
            break
        pos = nxt + 1
    return spans


def _fasta_events(path, fast=True):
    """The lines of a FASTA-like file as the perl one-liners of src/jasper.sh:155,220 see them: an event per non-empty line,
    ("h", first whitespace token) when that token starts with '>', else ("s", first whitespace token); consecutive "s" events
    come merged into one.  Fast path for the ordinary file (sequence lines of printable ASCII without blanks): whole bodies
    between header lines at once; anything else -- blanks or tabs in sequence lines, '\r', a '>' after leading blanks,
    non-ASCII bytes -- goes line by line through str.split(), like the text-mode reader this replaces."""
    if fast:
        with open(path, "rb") as f:
            data = f.read()
        import locale
        enc = locale.getpreferredencoding(False)     # (what open(path, "r") decodes with)
        events, pos, ok = [], 0, b"\r" not in data    # (text mode also ends lines at a lone '\r': line by line then)

        def body_event(body):
            if not body:
                return True
            if body.translate(None, _SAFE_BODY):
                return False
            body = body.translate(None, b"\n")
            if body:
                events.append(("s", body.decode("ascii")))
            return True

        for hs, he in _header_spans(data) if ok else ():
            if not body_event(data[pos:hs]):
                ok = False
                break
            F = data[hs:he].decode(enc, "replace").split()
            if not F or not F[0].startswith(">"):     # (cannot happen: the line starts with '>')
                ok = False
                break
            events.append(("h", F[0]))
            pos = he
        if ok and body_event(data[pos:]):
            return events
    events, parts = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            F = line.split()
            if not F:
                continue
            if F[0].startswith(">"):
                if parts:
                    events.append(("s", "".join(parts)))
                    parts = []
                events.append(("h", F[0]))
            else:
                parts.append(F[0])
    if parts:
        events.append(("s", "".join(parts)))
    return events


def read_assembly(path, fast=True):
    """perl #1 of src/jasper.sh:155: header = first whitespace token of a '>' line, sequence = first tokens of other lines"""
    contigs = []
    name, seq = None, ""
    for kind, tok in _fasta_events(path, fast):
        if kind == "h":
            if name is not None and seq:
                contigs.append((name, seq))
            name, seq = tok, ""
        else:
            seq += tok                        # (events come merged: at most one "s" between two headers)
    if name is not None and seq:
        contigs.append((name, seq))
    return contigs


def sequence_bytes(path, fast=True):
    """`grep -v '^>' $QUERY | tr -d '\\n' | wc` third column (src/jasper.sh:132)"""
    if fast:
        with open(path, "rb") as f:
            data = f.read()
        n = len(data) - data.count(b"\n")
        for hs, he in _header_spans(data):                        # header lines do not count (their '\n' was taken off above)
            n -= (he - hs) - (1 if data[he - 1:he] == b"\n" else 0)
        return n
    n = 0
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                n += len(line) - (1 if line.endswith(b"\n") else 0)
    return n


def split_batches(contigs, batch_size, query_fn):
    """src/jasper.sh:155-156: chunk records '>name:offset' of <= batch_size bases; a new batch file starts at a
    header once more than batch_size bases have been written to the current one"""
    bs = int(batch_size)
    records = []
    for name, seq in contigs:
        if bs <= 0:
            # perl's `for($ci=0;$ci<length;$ci+=0)` would never end; jasper.sh guarantees bs>0 for non-empty input
            records.append(("%s:0" % name, seq))
            continue
        for ci in range(0, len(seq), bs):
            records.append(("%s:%d" % (name, ci), seq[ci:ci + bs]))
    files = []
    batch_index, output = 0, 0
    f = open("%s.batch.%d.fa" % (query_fn, batch_index), "w")
    files.append(f.name)
    held = {}                                 # what polisher.parse_fasta would make of the file being written
    plain = True

    def close():
        f.close()
        if plain:
            polisher.remember(f.name, "records", dict(held))
    for hdr, seq in records:
        if output > bs:
            close()
            batch_index += 1
            f = open("%s.batch.%d.fa" % (query_fn, batch_index), "w")
            files.append(f.name)
            output = 0
            held, plain = {}, True
        f.write(hdr + "\n")
        f.write(seq + "\n")
        output += len(seq)
        # (parse_fasta: a line that starts with '>' names a record by its first token; other lines, '\n' taken off, are its text)
        plain = plain and hdr.isascii() and seq.isascii() and hdr.startswith(">") and len(hdr.split()) == 1 and not seq.startswith(">") and "\n" not in seq and "\r" not in seq
        held[hdr.split()[0][1:] if hdr.split() else ""] = seq
    close()
    return files


def join_polished(fixed_files, batch_size, contig_order, fast=True):
    """perl of src/jasper.sh:220: chunks keyed '>name:offset', emitted per contig by walking offsets 0, bs, 2bs ..."""
    bs = int(batch_size)
    if bs <= 0:
        bs = 1
    h = {}
    ctg, seq = "", ""                         # (perl's undef $ctg is the hash key "")
    for path in fixed_files:                  # (one stream: a sequence may go on in the next file, as for `cat`)
        events = polisher.recall(path, "events") if fast else None        # (written by this very process a moment ago: polisher.main_many)
        for kind, tok in (events if events is not None else _fasta_events(path, fast)):
            if kind == "h":
                if seq:
                    h[ctg] = seq
                    seq = ""
                ctg = tok
            else:
                seq += tok
    h[ctg] = seq
    out = []
    keys = [c for c in h if c.endswith(":0")]
    rank = {n: i for i, n in enumerate(contig_order)}
    keys.sort(key=lambda c: rank.get(c[:c.rfind(":")], len(rank)))
    for c in keys:
        base = c[:c.rfind(":")]
        out.append(base + "\n")
        b = 0
        while (base + ":%d" % b) in h:
            out.append(h[base + ":%d" % b])
            b += bs
        out.append("\n")
    return "".join(out)


def merge_fix_csvs(csv_files):
    """src/jasper.sh:222-226 (awk | awk -F: | sort -k1,1 -k2,2n -k3,3n | awk); byte order for the name key"""
    lines = []
    for n, path in enumerate(csv_files):
        with open(path, "r", newline="") as f:
            content = f.read().split("\n")
        if content and content[-1] == "":
            content.pop()
        for fnr, ln in enumerate(content, start=1):
            if (n == 0 and fnr == 1) or fnr > 1:
                lines.append(ln)
    def awk_fields(rec):
        """awk's default field splitting: runs of blank/tab/newline separate fields ('\r' is NOT a separator)"""
        t = rec.strip(" \t\n")
        return re.split(r"[ \t\n]+", t) if t else []

    rows = []
    for ln in lines:                                   # awk -F ':' '{print $1" "$2}'
        parts = ln.split(":")
        rows.append(parts[0] + " " + (parts[1] if len(parts) > 1 else ""))

    def num(x):                                        # sort -n: leading integer, 0 if none
        m = re.match(r"[ \t]*-?\d+", x)
        return int(m.group(0)) if m else 0

    # (a row of plain ASCII without the other characters str.split() takes for blanks -- every row this program writes -- splits the
    #  same way with the built-in, whose fields are reused for the output line: the regular expressions cost 7 us a row)
    odd = re.compile(r"[^\x20-\x7e\t]")

    def fields(s):
        # (the rows this program writes end with the CSV's '\r', which awk keeps on the last field -- or as a field of its own
        #  behind a blank)
        if s.endswith("\r") and not odd.search(s, 0, len(s) - 1):
            body = s[:-1]
            F = body.split()
            if not F or body[-1] in " \t":
                F.append("\r")
            else:
                F[-1] += "\r"
            return F
        return awk_fields(s) if odd.search(s) else s.split()

    def key_of(s, F):                                  # sort -k1,1 -k2,2n -k3,3n, last resort: whole line
        f1 = F[1] if len(F) > 1 else ""
        f2 = F[2] if len(F) > 2 else ""
        return (F[0].encode() if F else b"", int(f1) if f1.isascii() and f1.isdigit() else num(f1), int(f2) if f2.isascii() and f2.isdigit() else num(f2), s.encode())

    keyed = []
    for s in rows:
        F = fields(s)
        keyed.append((key_of(s, F), F))
    keyed.sort(key=lambda kf: kf[0])
    out = []
    for _, F in keyed:                                 # awk '{print $1":"$2" "$3" "$4" "$5}'
        F = F + [""] * (5 - len(F))
        out.append("%s:%s %s %s %s\n" % (F[0], F[1], F[2], F[3], F[4]))
    return "".join(out)


def write_merged_fix_csvs(csv_files, out_path):
    """merge_fix_csvs(csv_files) into out_path: natively (libjasper_hip.so: jasper_merge_fix_csvs, the same rules on bytes) for rows
    of printable ASCII, by the restatement above for anything else"""
    try:
        import ctypes as C
        from . import _lib
        arr = (C.c_char_p * max(len(csv_files), 1))(*[os.fsencode(p) for p in csv_files])
        rc = _lib.lib().jasper_merge_fix_csvs(arr, len(csv_files), os.fsencode(out_path))
        if rc == 0:
            return
    except Exception:           # noqa: BLE001 -- no library here: the rules in Python
        pass
    with open(out_path, "w", newline="") as f:
        f.write(merge_fix_csvs(csv_files))


def _write_histo(path, rows):
    with open(path + ".tmp", "w") as f:
        for m, n in rows:
            f.write("%d %d\n" % (m, n))
    os.replace(path + ".tmp", path)


def _run_multi(o, rank, world, dev, batch_size, passes, kmer, job=None):
    """the stages of run() below with the work of one node's GPUs divided as SURVEY.md 8e says: every rank counts its byte
    ranges of the read files (or its record range of an existing database) into a local table, dist.shard_tables sums the
    counts by key owner, and every rank polishes its share of the batch files with lookups served from the owners' HBM.
    Rank 0 alone splits, joins, writes the histogram / threshold / sentinels and talks; the stage decisions (which
    sentinels exist) are taken by rank 0 and shared, so that no rank sees a file another one is just creating.
    mer_counts$K.jf is written by all GPUs together (dist.write_jf_sharded); an existing one, or -j, is read in record ranges."""
    import torch
    import torch.distributed as tdist
    from . import dist as jdist
    is0 = rank == 0
    last_it = passes - 1
    qfn = o.query_fn

    def bar():
        torch.cuda.synchronize(dev)
        tdist.barrier()

    def decide(cond):
        return bool(jdist.all_reduce_ints([1 if (is0 and cond()) else 0], device=dev)[0])

    def together(work, msg):
        """a stage that every rank runs on its own share: a rank that fails (an exception, or the reference's own sys.exit(1))
        must not leave the others waiting in the next collective -- the outcome is agreed, and either all ranks go on or all
        leave with the message jasper.sh prints for that stage (only rank 0 talks)"""
        failed, why, out = 0, "", None
        try:
            out = work()
        except SystemExit as e:
            failed = 1 if e.code not in (0, None) else 0
        except BaseException as e:                      # noqa: BLE001 -- whatever it was, the others must hear of it
            failed, why = 1, "%s: %s" % (type(e).__name__, e)
        if why:
            sys.stderr.write("jasper_amd: rank %d: %s\n" % (rank, why))
        if jdist.all_reduce_ints([failed], device=dev)[0]:
            error_exit(msg)
        return out

    keep_fixed = bool(os.environ.get("JASPER_AMD_KEEP_INTERMEDIATES"))
    job_split = job_polished = False
    file_owner = None
    pinner = None

    def split_done():
        """the job's batch files are complete on every rank (or "Splitting files failed" on all): jasper.split.success"""
        if job_split and decide(lambda: not os.path.exists("jasper.split.success")):
            together(job.split_wait, "Splitting files failed, do you have enough disk space?")
            bar()
            if is0:
                if os.path.exists("jasper.correct.success"):
                    os.remove("jasper.correct.success")
                open("jasper.split.success", "w").close()

    if decide(lambda: not os.path.exists("jasper.split.success")):      # :152-159
        log("Splitting query into batches for parallel execution")
        # every rank holds the assembly in a job of its own (the same records and batch files everywhere: the plan is a function of
        # the file and the batch size); a rank writes the batch files it will polish, on a thread, while the reads are counted --
        # nobody splits alone behind a barrier.  Any rank without a job (not an ordinary FASTA): rank 0 splits in Python, as before.
        use_job = bool(jdist.all_reduce_ints([1 if (job is not None and batch_size > 0 and job.n_contigs) else 0], device=dev, op="min")[0])
        if is0:
            for p in glob.glob("%s.batch.*.fa" % glob.escape(qfn)):
                os.remove(p)
        bar()
        if use_job:
            def plan_and_write():
                nonlocal file_owner
                job.split(batch_size, qfn, write_files=False)
                order = sorted(range(job.n_files), key=job.batch_file_name)                    # `ls` order, as the polishing stage lists them
                own = jdist.assign_chunks([job.file_bytes[f] for f in order], world)
                file_owner = {f: ow for f, ow in zip(order, own)}
                job.split(batch_size, qfn, write_files=True, only_files=[f for f in order if file_owner[f] == rank])
            together(plan_and_write, "Splitting files failed, do you have enough disk space?")
            job_split = True
            pinner = _in_thread(lambda: job.pin(o.device))
        else:
            if is0:
                try:
                    split_batches(read_assembly(o.query), batch_size, qfn)
                except OSError:
                    error_exit("Splitting files failed, do you have enough disk space?")
                if os.path.exists("jasper.correct.success"):
                    os.remove("jasper.correct.success")
                open("jasper.split.success", "w").close()
            bar()

    histo_file = "jfhisto%d.csv" % kmer
    counted = False
    if o.jf_db is None:                                                 # :162-185
        reads = o.reads.split()
        for fn in reads:
            if not (os.path.isfile(fn) and os.path.getsize(fn) > 0):
                error_exit("The reads file  %s does not exist. Please supply a series of valid reads files separated by space and wrapped in one pair of quotation marks." % fn)
        jf_file = "mer_counts%d.jf" % kmer
        if decide(lambda: os.path.isfile(jf_file) and os.path.getsize(jf_file) > 0):     # :171-173
            log("Using existing jellyfish database %s" % jf_file)
            if is0 and os.path.exists("jasper.no_cat.success"):
                os.remove("jasper.no_cat.success")
            local = together(lambda: KmerTable.from_jf_part(jf_file, rank, world, device=o.device),
                             "Computing mer counts histogram from %s failed, please make sure that %s is a valid Jellyfish mer counts file" % (jf_file, jf_file))
        else:
            _timing("split")
            log("Creating jellyfish database mer_counts%d.jf" % kmer)
            fail_msg = "Computing mer counts histogram from mer_counts%d.jf failed, please make sure that mer_counts%d.jf is a valid Jellyfish mer counts file" % (kmer, kmer)
            my_ranges = jdist.plan_read_shards(reads, world)[rank]
            # No table per GPU when the key owners' table has a geometry for it (dist.count_sharded): the file reader feeds batches
            # of bases, every batch is partitioned into region lists by key owner, ONE all_to_all moves the lists, the owners insert.
            sharded = None
            how = os.environ.get("JASPER_AMD_COUNT", "auto")
            if how != "local":
                # (sized like the reference's `-s $JF_SIZE` hash, for the keys one owner will hold; JASPER_AMD_SHARD_SLOTS overrides)
                shard_slots = int(os.environ.get("JASPER_AMD_SHARD_SLOTS", max(1 << 21, int(1.25 * o.jf_size / world))))
                sharded = together(lambda: KmerTable(kmer, min_slots=shard_slots, device=o.device), fail_msg)
                plan = sharded.exchange_plan(1 << 26, world)
                take = plan is not None
                if take and how == "auto":   # bytes per link decide (dist.prefer_exchange): FASTQ is ~2.1 bytes per base; -s is the expected number of distinct k-mers
                    occ = sum((e if e >= 0 else os.path.getsize(p)) - b for p, b, e in my_ranges) / 2.1
                    dedup = plan["p2"] >= 1 and jdist.dedupe_pays(world)
                    take = jdist.prefer_exchange(world, occ, o.jf_size, deduplicated=dedup)
                if not jdist.all_reduce_ints([1 if take else 0], device=dev, op="min")[0]:
                    sharded.close()
                    sharded = None
            if sharded is not None:
                feeder = KmerTable(kmer, min_slots=1 << 10, device=o.device)      # lends its device buffers to the reader
                together(lambda: feeder.feed_start(my_ranges), fail_msg)
                try:
                    info = jdist.count_sharded(sharded, 0, 0, dev, feeder=feeder)
                except jdist.ShardAttachError as e:         # (raised on every rank together, after all lists were inserted)
                    sharded._attach_failed = str(e)
                    info = dict(rounds=-1)
                except jdist.CollectiveCountError as e:     # (raised on every rank together: e.g. shards sized from a hint that was far too small;
                                                            #  anything else is this rank's own failure and ends it -- no fallback the peers do not take)
                    if is0:
                        sys.stderr.write("jasper_amd: %s -- counting into a table per GPU instead\n" % e)
                    info = None
                finally:
                    feeder.close()
                if info is None:                            # start over the round-1 way (the read files are read again)
                    sharded.detach()
                    bar()
                    sharded.close()
                    sharded = None
                else:
                    local = None
                    _timing("count reads (file ranges -> region lists -> owners' shards, %d rounds)" % info["rounds"])
            if sharded is None:
                def count_my_ranges():
                    t = KmerTable(kmer, min_slots=max(1 << 20, int(1.25 * o.jf_size / world)), device=o.device)
                    t.count_file_ranges(my_ranges)
                    return t
                local = together(count_my_ranges, fail_msg)
                _timing("count reads (file ranges -> local table)")
            counted = True
    else:
        local = together(lambda: KmerTable.from_jf_part(o.jf_db, rank, world, device=o.device),
                         "Computing mer counts histogram from %s failed, please make sure that %s is a valid Jellyfish mer counts file" % (o.jf_db, o.jf_db))
    # key-wise sum over the GPUs; the result stays sharded by key owner unless the peers' HBM cannot be mapped
    write_db = counted and os.environ.get("JASPER_AMD_NO_JF", "") not in ("1", "true", "yes")
    db_cmdline = ["count", "-C", "-t", str(o.num_threads), "-s", str(o.jf_size), "-m", str(kmer), "-o", "mer_counts%d.jf" % kmer] + (o.reads.split() if counted else [])
    # sharded by owner, or a copy of the whole table on every GPU?  dist.prefer_replicated: it must fit and the gather must cost less than
    # the remote lookups it saves -- with one polish call per counted table it does not (JASPER_AMD_TABLE=replicated|sharded overrides)
    how_table = os.environ.get("JASPER_AMD_TABLE", "auto")
    replicate = how_table == "replicated"
    if how_table == "auto":
        try:
            import ctypes as C
            from . import _lib
            free_b, total_b = C.c_uint64(0), C.c_uint64(0)
            _lib.check(_lib.lib().jasper_device_mem_info(int(o.device), C.byref(free_b), C.byref(total_b)))
            asm_bases = job.n_bases if job is not None else os.path.getsize(o.query)
            replicate = jdist.prefer_replicated(world, max(o.jf_size, 1), asm_bases / world, passes + 1, free_b.value)
        except Exception:           # noqa: BLE001 -- no answer: the default
            replicate = False
    replicate = bool(jdist.all_reduce_ints([1 if replicate else 0], device=dev, op="min")[0])
    try:
        if replicate:
            raise jdist.ShardAttachError("a copy of the whole table on every GPU was asked for (or is expected to pay)")
        if local is None:       # counted straight into the owners' shards
            table = sharded
            local = table       # (what the fallback below merges: the shards are disjoint, their key-wise sum is the whole table)
            if getattr(table, "_attach_failed", None):
                raise jdist.ShardAttachError(table._attach_failed)
        else:
            table = KmerTable(local.k, min_slots=1 << 21, device=o.device)
            jdist.shard_tables(local, table, dev)
            local.close()
        if write_db:       # :177 `... | tee $JF_DB | ...`: every GPU sorts and writes one consecutive piece of the file
            jdist.write_jf_sharded(table, "mer_counts%d.jf" % kmer, db_cmdline, dev)
            _timing("write mer_counts.jf")
        h = jdist.histogram_sharded(table, dev)
    except jdist.ShardAttachError as e:
        if is0:
            sys.stderr.write("jasper_amd: %s -- replicating the merged table on every GPU instead\n" % e)
        if replicate:
            table = local if local is not None else sharded
            local = table
        table.detach()          # (whatever was mapped is unmapped on every rank before anybody frees its slot array)
        bar()
        if table is not local:
            table.close()
        table = local
        jdist.merge_tables(table, dev)
        if write_db:
            if is0:
                table.write_jf("mer_counts%d.jf.tmp" % kmer, db_cmdline)
                os.replace("mer_counts%d.jf.tmp" % kmer, "mer_counts%d.jf" % kmer)
            bar()
        h = jdist.histogram_merged(table, dev)
    rows = [(m, h[m]) for m in range(1, 10002) if h[m]]
    _timing("sum counts over the GPUs + histogram")
    if counted and is0:
        _write_histo(histo_file, rows)
        open("jasper.no_cat.success", "w").close()
        open("jasper.histo.success", "w").close()
        if os.path.exists("jasper.correct.success"):
            os.remove("jasper.correct.success")
    bar()

    split_done()
    if decide(lambda: not os.path.exists("jasper.histo.success") or not (os.path.isfile(histo_file) and os.path.getsize(histo_file) > 0)):   # :187-193
        log("Computing K-mer histogram")
        if is0:
            _write_histo(histo_file, rows)
            if os.path.exists("jasper.correct.success"):
                os.remove("jasper.correct.success")
            open("jasper.histo.success", "w").close()
        bar()

    if decide(lambda: not os.path.exists("jasper.correct.success")):    # :195-216
        log("Polishing")
        if is0:
            txt, status = polisher.threshold_from_histo_file(histo_file)
            if status == 0:
                with open("threshold.txt.tmp", "w") as f:
                    f.write(txt)
                os.replace("threshold.txt.tmp", "threshold.txt")
        bar()
        if not (os.path.isfile("threshold.txt") and os.path.getsize("threshold.txt") > 0):
            error_exit("Local min of kmer counts is smaller than 4. The input read data is not suitable for polishing.")
        thresh = int(open("threshold.txt").read().split()[0])
        log("Lower threshold for unreliable kmers is %d" % thresh)
        group, group_bytes = [], 0
        if job_split:
            # (as in run(): record text from the job's arena, polished text back into the job, no `_iter*.fixed.fa` unless asked for,
            #  and jasper.correct.success only once the join below has made the polished FASTA)
            def flush_group():
                if group:
                    polisher.main_many_job(job, list(group), kmer, True, True, table, thresh, passes, keep_fixed=keep_fixed)
                    if keep_fixed:
                        for f in group:
                            bf = job.batch_file_name(f)
                            os.replace("_iter%d_%s.fixed.fa.tmp" % (last_it, bf), "_iter%d_%s.fixed.fa" % (last_it, bf))
                    del group[:]
            def polish_my_batches():
                nonlocal group_bytes
                if pinner is not None:
                    pinner.join()
                for f in sorted(range(job.n_files), key=job.batch_file_name):
                    if file_owner[f] != rank:
                        continue
                    group.append(f)
                    group_bytes += job.file_bytes[f]
                    if group_bytes > (1 << 30):
                        flush_group()
                        group_bytes = 0
                flush_group()
            together(polish_my_batches, "Polishing failed")                          # :215
            job_polished = True
            bar()
            if is0 and os.path.exists("jasper.join.success"):
                os.remove("jasper.join.success")
            bar()
        else:
            batch_files = sorted(glob.glob("%s.batch.*.fa" % glob.escape(qfn)))   # `ls` order
            owner = jdist.assign_chunks([os.path.getsize(bf) for bf in batch_files], world)

            def flush_group():
                if group:
                    polisher.main_many(group, kmer, True, True, table, thresh, passes)
                    for bf in group:
                        os.replace("_iter%d_%s.fixed.fa.tmp" % (last_it, bf), "_iter%d_%s.fixed.fa" % (last_it, bf))
                    del group[:]
            def polish_my_batches():
                nonlocal group_bytes
                for bf, ow in zip(batch_files, owner):
                    if ow != rank:
                        continue
                    group.append(bf)
                    group_bytes += os.path.getsize(bf)
                    if group_bytes > (1 << 30):
                        flush_group()
                        group_bytes = 0
                flush_group()
            together(polish_my_batches, "Polishing failed")                          # :215
            bar()
            if is0:
                if os.path.exists("jasper.join.success"):
                    os.remove("jasper.join.success")
                open("jasper.correct.success", "w").close()
            bar()

    if decide(lambda: not os.path.exists("jasper.join.success")):       # :218-232
        _timing("polish batches")
        log("Joining")
        if job_polished:
            # every rank writes the records it polished straight into their places of ONE file: a record's place follows from the
            # polished lengths of the records before it (a sum over ranks of a short vector), so no text moves between ranks and
            # nobody reads the assembly or the fixed files again (src/jasper.sh:220)
            lens, have = job.polished_lens()
            all_lens = jdist.all_reduce_ints([int(v) for v in lens], device=dev)
            held = jdist.all_reduce_ints([int(v) for v in have], device=dev)
            tmp = qfn + ".fixed.fasta.tmp"

            def create():
                if min(held, default=1) != 1 or max(held, default=1) != 1:
                    raise RuntimeError("a chunk record was polished by no rank, or by two")
                if is0:
                    job.join(tmp, all_lens=all_lens, mode=1)
            together(create, "Joining failed")
            bar()
            together(lambda: job.join(tmp, all_lens=all_lens, mode=2), "Joining failed")
            bar()
            if is0:
                os.replace(tmp, qfn + ".polished.fasta")
                open("jasper.correct.success", "w").close()
                _join_and_merge(o, qfn, batch_size, last_it, None, fasta_done=True)
        elif is0:
            _join_and_merge(o, qfn, batch_size, last_it, read_assembly(o.query))
        bar()
    if is0:
        _qv_block(passes, kmer)
    _timing("join + QV")
    log("Polished sequence is in %s.polished.fasta" % qfn)
    bar()                       # nobody unmaps or frees a shard that a peer may still be reading ...
    table.detach()              # ... every rank lets go of its peers' memory ...
    bar()                       # ... and only then is any of it freed
    table.close()
    bar()
    tdist.destroy_process_group()
    return 0


def _stop_thread(th, leftover=None):
    """atexit: a thread of ours that is still inside the library -- an exit taken on an error elsewhere, e.g. "Splitting files
    failed" while the reads are being counted -- is asked to stop at its next chunk / block (jasper_request_cancel; src/jasper.sh:23-28
    kills its children on the way out) and waited for: tearing the interpreter down under a thread that uses the GPU can hang."""
    if th.is_alive():
        try:
            from . import _lib
            _lib.lib().jasper_request_cancel(1)
        except Exception:          # noqa: BLE001 -- no library: nothing of ours can be running
            pass
        th.join()
        if leftover and os.path.exists(leftover):
            try:
                os.remove(leftover)
            except OSError:
                pass


class _EarlyTable:
    """KmerTable(k, min_slots) created -- and the read files counted into it -- by a thread (the library calls release the GIL)
    while the caller splits the assembly; get() hands the table over, or raises what the thread raised"""

    def __init__(self, k, min_slots, device, reads=None):
        import threading
        self.out, self.err = None, None

        def work():
            try:
                import time
                t0 = time.perf_counter()
                t = KmerTable(k, min_slots=min_slots, device=device)
                t1 = time.perf_counter()
                if reads:
                    t.count_files(reads)
                if os.environ.get("JASPER_AMD_TIMING"):
                    sys.stderr.write("[timing-thread] library + GPU runtime + table %.3f s, files -> table %.3f s\n" % (t1 - t0, time.perf_counter() - t1))
                self.out = t
            except BaseException as e:          # noqa: BLE001 -- handed to the caller of get()
                self.err = e

        self.th = threading.Thread(target=work, daemon=True)
        self.th.start()
        # an exit taken while the thread is still inside the GPU driver (the split stage that runs beside it calls error_exit ->
        # sys.exit on an unreadable assembly or a full disk) must wait for it: tearing the interpreter down under a thread that
        # initialises HIP can hang or crash instead of giving the reference's clean "Splitting files failed" exit code
        import atexit
        atexit.register(_stop_thread, self.th)

    def get(self):
        self.th.join()
        if self.err is not None:
            raise self.err
        return self.out


def _in_thread(fn):
    """fn() on a daemon thread; what it raises is dropped (optional work: pinning buffers)"""
    import threading

    def work():
        try:
            fn()
        except BaseException:           # noqa: BLE001
            pass
    th = threading.Thread(target=work, daemon=True)
    th.start()
    return th


class _JobJoin:
    """job.join(tmp) by a thread (src/jasper.sh:220 from the job's memory; the library call releases the GIL); finish() waits and
    raises what the thread raised"""

    def __init__(self, job, tmp):
        import threading
        self.err = None

        def work():
            try:
                job.join(tmp)
            except BaseException as e:          # noqa: BLE001 -- handed to the caller of finish()
                self.err = e

        self.th = threading.Thread(target=work, daemon=True)
        self.th.start()

    def finish(self):
        self.th.join()
        if self.err is not None:
            raise self.err


class _JfWriter:
    """table.write_jf(tmp) + rename to `final`, by a thread, while the caller goes on to the histogram and the polishing (the
    database file is `tee`'s by-product in the reference, src/jasper.sh:177: nothing in the same run reads it, and writing 14 bytes
    per distinct k-mer takes longer than all the polishing).  finish() waits and raises what the thread raised; the file gets its
    name only when it is complete."""

    def __init__(self, table, tmp, final, cmdline):
        import threading
        self.err = None
        self.cmdline = cmdline

        def work():
            try:
                table.write_jf(tmp, cmdline)
                os.replace(tmp, final)
            except BaseException as e:          # noqa: BLE001 -- handed to the caller of finish()
                self.err = e

        self.th = threading.Thread(target=work, daemon=True)
        self.th.start()
        import atexit
        atexit.register(_stop_thread, self.th, tmp)   # (an exit taken meanwhile waits for the thread, as for _EarlyTable; its unfinished file goes)

    def finish(self):
        self.th.join()
        if self.err is not None:
            raise self.err


def _jf_write_fits_beside_polishing(table, device, qfn, text_bytes=None):
    """does the device have room for table.write_jf (entries + keys for the sort + the sort's temporaries + the formatted
    records: ~64 bytes per distinct k-mer, measured 48 + rocPRIM's temporaries) AND the polisher's workspaces for the largest group
    of batch files (two text arenas, classes, segment buffers, records: ~8 bytes per base of a group of <= 1 Gbase) at once?"""
    import ctypes as C
    from . import _lib
    free, total = C.c_uint64(0), C.c_uint64(0)
    try:
        _lib.check(_lib.lib().jasper_device_mem_info(int(device), C.byref(free), C.byref(total)))
        distinct = table.info()["distinct"]
        text = text_bytes if text_bytes is not None else sum(os.path.getsize(p) for p in glob.glob("%s.batch.*.fa" % glob.escape(qfn)))
    except Exception:                                  # noqa: BLE001 -- no answer: the safe order
        return False
    need = 64 * distinct + 8 * min(text, 1 << 30) + (2 << 30)
    if os.environ.get("JASPER_AMD_TEST_JF_SERIAL"):      # (tests: take the serial order whatever the device has)
        return False
    return need < free.value


def _join_and_merge(o, qfn, batch_size, last_it, contigs, fasta_done=False):
    """src/jasper.sh:218-232 (fasta_done: the polished FASTA is already in place, written from an AssemblyJob)"""
    if not fasta_done:
        fixed_files = sorted(glob.glob("_iter%d_%s.batch.*.fa.fixed.fa" % (last_it, glob.escape(qfn))))
        text = join_polished(fixed_files, batch_size, [c[0] for c in contigs])
        with open(qfn + ".fixed.fasta.tmp", "w") as f:
            f.write(text)
        os.replace(qfn + ".fixed.fasta.tmp", qfn + ".polished.fasta")
    for p in glob.glob("_iter*_%s.batch.*.fa.fixed.fa" % glob.escape(qfn)) + glob.glob("_iter*_%s.batch.*.fa.fixed.fa.tmp" % glob.escape(qfn)):
        os.remove(p)
    csvs = sorted(glob.glob("_iter*_%s.batch.*.fa.fix.csv" % glob.escape(qfn)))
    write_merged_fix_csvs(csvs, qfn + ".fixes.csv.tmp")
    os.replace(qfn + ".fixes.csv.tmp", qfn + ".fixes.csv")
    open("jasper.join.success", "w").close()
    if not o.debug:
        for p in csvs + glob.glob("%s.batch.*.fa" % glob.escape(qfn)):
            if os.path.exists(p):
                os.remove(p)


def _qv_block(passes, kmer):
    """src/jasper.sh:235-257"""
    def colsum(path):
        a = b = 0
        with open(path) as f:
            for ln in f:
                F = ln.split()
                if len(F) >= 2:
                    a += int(F[0]); b += int(F[1])
        return a, b
    if os.path.exists("0qValCalcHelper.csv") and os.path.exists("%dqValCalcHelper.csv" % passes):
        b0, t0 = colsum("0qValCalcHelper.csv")
        b1, t1 = colsum("%dqValCalcHelper.csv" % passes)
        log("Before Polishing: Q value = %s" % qv.q_value(b0, t0, kmer))
        log("After Polishing: Q value = %s" % qv.q_value(b1, t1, kmer))
        if os.environ.get("JASPER_AMD_TIMING"):       # (the column sums themselves: the reference removes the files it takes them from, :258)
            sys.stderr.write("[qv] before %d %d after %d %d\n" % (b0, t0, b1, t1))
        for p in glob.glob("*qValCalcHelper.csv"):
            os.remove(p)


def _init_multi(o):
    """one process per GPU under `python -m torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the environment):
    returns (rank, world, torch device) after joining the process group (RCCL; JASPER_AMD_DIST_BACKEND=gloo and
    JASPER_AMD_ONE_GPU=1 rehearse it on one GPU)"""
    from . import dist as jdist
    rank, world, local = jdist.env_world()
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as tdist
    if os.environ.get("JASPER_AMD_ONE_GPU", "") not in ("1", "true", "yes"):
        o.device = local
    torch.cuda.set_device(o.device)
    dev = torch.device("cuda", o.device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("JASPER_AMD_DIST_BACKEND", "nccl")
    if backend == "nccl":
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        tdist.init_process_group(backend, rank=rank, world_size=world)
    _QUIET[0] = rank != 0
    return rank, world, dev


def run(argv):
    if os.environ.get("JASPER_AMD_TIMING"):
        import time
        sys.stderr.write("[timing-abs] run() entered at %.6f\n" % time.time())
    o = parse_args(argv)
    rank, world, dev = _init_multi(o)
    multi = world > 1
    if not (os.path.isfile(o.query) and os.path.getsize(o.query) > 0):
        error_exit("The query file does not exist. Please supply a valid fasta file to be polished with -a option.")
    # The counting stage -- the start of the GPU runtime, the table's allocation and reads -> table: everything of src/jasper.sh:177
    # but the database file -- is the work of a thread that starts NOW, before this one even sizes the batches: the two do not
    # depend on each other, counting is the longest stage of a run, and the log lines keep the reference's order.  Only when
    # counting WILL happen and the flags are the ones run() accepts below (no exit while the thread is in the driver).
    early = None
    def _flags_ok():
        try:
            return (re.match(r"^-?[0-9]+$", str(o.kmer)) and int(o.kmer) - 1 >= 0 and re.match(r"^-?[0-9]+$", str(o.passes)) and int(o.passes) - 1 >= 0
                    and float(o.num_threads) > 0)
        except ValueError:
            return False
    if (not multi and _flags_ok()
            and o.jf_db is None and not (os.path.isfile("mer_counts%d.jf" % int(o.kmer)) and os.path.getsize("mer_counts%d.jf" % int(o.kmer)) > 0)
            and not os.environ.get("JASPER_AMD_NO_EARLY_TABLE") and o.reads.split() and all(os.path.isfile(fn) and os.path.getsize(fn) > 0 for fn in o.reads.split())):
        early = _EarlyTable(int(o.kmer), max(1 << 20, int(1.25 * o.jf_size)), o.device, reads=o.reads.split())
    batch_size = o.batch_size
    if not re.match(r"^[0-9]+$", str(batch_size)):
        log("BATCH SIZE supplied is not a positive integer. Calculating BATCH SIZE from QUERY SIZE")
        batch_size = "0"
    batch_size = int(batch_size)
    # The assembly is read ONCE, natively and by several threads, into a host arena that the split, the polisher and the join all
    # work from (assembly.AssemblyJob); a file that is not an ordinary FASTA -- '\r', blanks in sequence lines, non-ASCII bytes,
    # text before the first '>', a name that occurs twice -- gives None and takes the line-by-line rules below, in Python.
    job = None
    if not os.environ.get("JASPER_AMD_NO_NATIVE_ASM"):
        try:
            from .assembly import AssemblyJob
            job = AssemblyJob.open(o.query)
        except Exception:           # noqa: BLE001 -- no library: the table's constructor reports it
            job = None
    try:
        nthreads = float(o.num_threads)
        bs = int((job.sequence_bytes if job is not None else sequence_bytes(o.query)) / nthreads * .9)               # :132
    except (ValueError, ZeroDivisionError):
        error_exit("The number of threads supplied by -t must be a positive integer")
    if bs > batch_size:                                                 # :133-138
        batch_size = bs
        if batch_size > MAX_BATCH_SIZE:
            batch_size = MAX_BATCH_SIZE
    _timing("start")
    log("Using BATCH SIZE %d" % batch_size)
    if not re.match(r"^-?[0-9]+$", str(o.passes)) or int(o.passes) - 1 < 0:
        error_exit("The number of passes supplied by -p must be a positive integer")
    if not re.match(r"^-?[0-9]+$", str(o.kmer)) or int(o.kmer) - 1 < 0:
        error_exit("The k-mer size supplied by -k must be a positive integer")
    passes, kmer = int(o.passes), int(o.kmer)
    last_it = passes - 1
    qfn = o.query_fn
    contigs = None
    if multi:
        return _run_multi(o, rank, world, dev, batch_size, passes, kmer, job)

    jf_writer = None
    keep_fixed = bool(os.environ.get("JASPER_AMD_KEEP_INTERMEDIATES"))
    job_split = False          # the batch files are the job's (being written by its thread until _split_done())
    job_polished = False       # the polished records are in the job's memory
    join_writer = []           # [_JobJoin]: the polished FASTA being written from them
    pinner = None              # thread: job.pin()

    def _split_done():
        nonlocal job_split
        if job_split and not os.path.exists("jasper.split.success"):
            try:
                job.split_wait()
            except Exception:           # noqa: BLE001 -- a full disk, a directory that went away
                error_exit("Splitting files failed, do you have enough disk space?")
            if os.path.exists("jasper.correct.success"):
                os.remove("jasper.correct.success")
            open("jasper.split.success", "w").close()

    if not os.path.exists("jasper.split.success"):                      # :152-159
        log("Splitting query into batches for parallel execution")
        for p in glob.glob("%s.batch.*.fa" % glob.escape(qfn)):
            os.remove(p)
        if job is not None and batch_size > 0 and job.n_contigs:
            # perl #1 + perl #2 on the arena; the files are written by a thread of the job while the reads are counted, and
            # jasper.split.success appears when they are complete (_split_done, before "Polishing")
            try:
                job.split(batch_size, qfn, write_files=True)
            except Exception:           # noqa: BLE001
                error_exit("Splitting files failed, do you have enough disk space?")
            job_split = True
            pinner = _in_thread(lambda: job.pin(o.device))      # (waits for the GPU runtime, pins the arena: beside the counting)
        else:
            try:
                contigs = read_assembly(o.query)
                split_batches(contigs, batch_size, qfn)
            except OSError:
                error_exit("Splitting files failed, do you have enough disk space?")
            if os.path.exists("jasper.correct.success"):
                os.remove("jasper.correct.success")
            open("jasper.split.success", "w").close()

    table = None
    histo_file = "jfhisto%d.csv" % kmer
    if o.jf_db is None:                                                 # :162-185
        reads = o.reads.split()
        for fn in reads:
            if not (os.path.isfile(fn) and os.path.getsize(fn) > 0):
                error_exit("The reads file  %s does not exist. Please supply a series of valid reads files separated by space and wrapped in one pair of quotation marks." % fn)
        jf_file = "mer_counts%d.jf" % kmer
        if os.path.isfile(jf_file) and os.path.getsize(jf_file) > 0:     # :171-173
            log("Using existing jellyfish database %s" % jf_file)
            if os.path.exists("jasper.no_cat.success"):
                os.remove("jasper.no_cat.success")
            table = KmerTable.from_jf(jf_file, device=o.device)
        else:
            _timing("split")
            log("Creating jellyfish database mer_counts%d.jf" % kmer)
            if early is not None:
                try:                                    # (what the later stages import, while this thread only waits)
                    import numpy                        # noqa: F401 -- 0.1 s that histo_rows / the fix records would otherwise spend
                    import csv, io                      # noqa: F401,E401
                except ImportError:
                    pass
                table = early.get()                     # (counted while the assembly was split)
                early = None
            else:
                table = KmerTable(kmer, min_slots=max(1 << 20, int(1.25 * o.jf_size)), device=o.device)
                table.count_files(reads)
            _timing("count reads (files -> table)")
            # (the histogram first, on this thread: whatever a lazily cleared table still owes its slots is settled before a
            #  second thread looks at them)
            with open(histo_file + ".tmp", "w") as f:
                for m, n in table.histo_rows():
                    f.write("%d %d\n" % (m, n))
            os.replace(histo_file + ".tmp", histo_file)
            if os.environ.get("JASPER_AMD_NO_JF", "") not in ("1", "true", "yes"):
                # :177 `... | tee $JF_DB | ...`: leave the database behind for reruns and for other Jellyfish tools.  Written by a
                # thread beside the polishing when the device has room for both (the writer holds ~48 bytes per distinct k-mer plus
                # its sort's workspace, the polisher several times its batch's text); otherwise first the file, then the polishing,
                # as the reference orders them.
                jf_cmdline = ["count", "-C", "-t", str(o.num_threads), "-s", str(o.jf_size), "-m", str(kmer), "-o", jf_file] + reads
                if _jf_write_fits_beside_polishing(table, o.device, qfn, sum(job.file_bytes) if job_split else None):      # (the job's files may still be being written)
                    jf_writer = _JfWriter(table, jf_file + ".tmp", jf_file, jf_cmdline)
                else:
                    try:
                        table.write_jf(jf_file + ".tmp", jf_cmdline)
                        os.replace(jf_file + ".tmp", jf_file)
                    except Exception as e:             # noqa: BLE001 -- what `set -o pipefail` makes of a failing tee (src/jasper.sh:177-181)
                        error_exit("Creating jellyfish database mer_counts%d.jf failed (%s)" % (kmer, e))
                    _timing("write mer_counts.jf (before the polishing: not enough device memory for both at once)")
            open("jasper.no_cat.success", "w").close()
            open("jasper.histo.success", "w").close()
            if os.path.exists("jasper.correct.success"):
                os.remove("jasper.correct.success")
    else:
        # -j: an existing Jellyfish database; its header decides k (JF::swig/mer_file.i:23 -- the DB wins over -k)
        try:
            table = KmerTable.from_jf(o.jf_db, device=o.device)
        except Exception as e:
            error_exit("Computing mer counts histogram from %s failed, please make sure that %s is a valid Jellyfish mer counts file (%s)"
                       % (o.jf_db, o.jf_db, e))

    _split_done()
    if not os.path.exists("jasper.histo.success") or not (os.path.isfile(histo_file) and os.path.getsize(histo_file) > 0):   # :187-193
        log("Computing K-mer histogram")
        with open(histo_file + ".tmp", "w") as f:
            for m, n in table.histo_rows():
                f.write("%d %d\n" % (m, n))
        os.replace(histo_file + ".tmp", histo_file)
        if os.path.exists("jasper.correct.success"):
            os.remove("jasper.correct.success")
        open("jasper.histo.success", "w").close()

    if not os.path.exists("jasper.correct.success"):                    # :195-216
        _timing("histogram")
        log("Polishing")
        txt, status = polisher.threshold_from_histo_file(histo_file)
        if status == 0:
            with open("threshold.txt.tmp", "w") as f:
                f.write(txt)
            os.replace("threshold.txt.tmp", "threshold.txt")
        if not (os.path.isfile("threshold.txt") and os.path.getsize("threshold.txt") > 0):
            error_exit("Local min of kmer counts is smaller than 4. The input read data is not suitable for polishing.")
        thresh = int(open("threshold.txt").read().split()[0])
        log("Lower threshold for unreliable kmers is %d" % thresh)
        # the reference starts one jasper.py process per batch file (:207-212); chunk records are independent, so all
        # files go through the GPU in groups (<= ~1 Gbase of text per call) and leave the same per-file artefacts
        group, group_bytes = [], 0
        if job_split:
            # the job's own batch files, in `ls` order; record text goes from the arena to the GPU and the polished text back into
            # the job.  The `_iter*.fixed.fa` files have one reader, the join below, which then works from memory: they are not
            # written (JASPER_AMD_KEEP_INTERMEDIATES=1 writes them), and so jasper.correct.success -- "the fixed files are
            # complete" -- appears only once the join has made the polished FASTA from them (a run that dies in between starts the
            # polishing over instead of joining files that are not there).
            if pinner is not None:
                pinner.join()
            groups = [[]]
            for f in sorted(range(job.n_files), key=job.batch_file_name):
                groups[-1].append(f)
                group_bytes += job.file_bytes[f]
                if group_bytes > (1 << 30):
                    groups.append([])
                    group_bytes = 0
            groups = [g for g in groups if g]
            if os.path.exists("jasper.join.success"):
                os.remove("jasper.join.success")
            for gi, g in enumerate(groups):
                # (the moment the last group's polished text is in the job, a thread starts writing the polished FASTA from it,
                #  src/jasper.sh:220, while this one still turns fix records into CSV rows)
                last = gi == len(groups) - 1
                polisher.main_many_job(job, g, kmer, True, True, table, thresh, passes, keep_fixed=keep_fixed,
                                       on_taken=(lambda: join_writer.append(_JobJoin(job, qfn + ".fixed.fasta.tmp"))) if last else None)
                if keep_fixed:
                    for f in g:
                        bf = job.batch_file_name(f)
                        os.replace("_iter%d_%s.fixed.fa.tmp" % (last_it, bf), "_iter%d_%s.fixed.fa" % (last_it, bf))
            job_polished = True
        else:
            batch_files = sorted(glob.glob("%s.batch.*.fa" % glob.escape(qfn)))   # `ls` order
            def flush_group():
                if group:
                    polisher.main_many(group, kmer, True, True, table, thresh, passes)
                    for bf in group:
                        os.replace("_iter%d_%s.fixed.fa.tmp" % (last_it, bf), "_iter%d_%s.fixed.fa" % (last_it, bf))
                    del group[:]
            for bf in batch_files:
                group.append(bf)
                group_bytes += os.path.getsize(bf)
                if group_bytes > (1 << 30):
                    flush_group()
                    group_bytes = 0
            flush_group()
            if os.path.exists("jasper.join.success"):
                os.remove("jasper.join.success")
            open("jasper.correct.success", "w").close()

    if not os.path.exists("jasper.join.success"):                       # :218-232
        _timing("polish batches")
        log("Joining")
        if job_polished:
            try:
                if join_writer:
                    join_writer[0].finish()
                else:
                    job.join(qfn + ".fixed.fasta.tmp")
            except Exception:           # noqa: BLE001
                error_exit("Joining failed")
            os.replace(qfn + ".fixed.fasta.tmp", qfn + ".polished.fasta")
            open("jasper.correct.success", "w").close()
            _timing("  polished FASTA complete")
        else:
            if contigs is None:
                contigs = read_assembly(o.query)
            fixed_files = sorted(glob.glob("_iter%d_%s.batch.*.fa.fixed.fa" % (last_it, glob.escape(qfn))))
            text = join_polished(fixed_files, batch_size, [c[0] for c in contigs])
            with open(qfn + ".fixed.fasta.tmp", "w") as f:
                f.write(text)
            os.replace(qfn + ".fixed.fasta.tmp", qfn + ".polished.fasta")
        for p in glob.glob("_iter*_%s.batch.*.fa.fixed.fa" % glob.escape(qfn)) + glob.glob("_iter*_%s.batch.*.fa.fixed.fa.tmp" % glob.escape(qfn)):
            os.remove(p)
        csvs = sorted(glob.glob("_iter*_%s.batch.*.fa.fix.csv" % glob.escape(qfn)))
        write_merged_fix_csvs(csvs, qfn + ".fixes.csv.tmp")
        os.replace(qfn + ".fixes.csv.tmp", qfn + ".fixes.csv")
        open("jasper.join.success", "w").close()
        if not o.debug:
            for p in csvs + glob.glob("%s.batch.*.fa" % glob.escape(qfn)):
                if os.path.exists(p):
                    os.remove(p)

    _qv_block(passes, kmer)      # (:235-257)
    _timing("join + QV")
    if jf_writer is not None:
        try:
            try:
                jf_writer.finish()
            except Exception as e1:            # noqa: BLE001 -- e.g. a device allocation that failed beside the polisher's: once more, alone
                if "alloc" not in str(e1).lower() and "memory" not in str(e1).lower():
                    raise
                table.write_jf("mer_counts%d.jf.tmp" % kmer, jf_writer.cmdline)
                os.replace("mer_counts%d.jf.tmp" % kmer, "mer_counts%d.jf" % kmer)
        except Exception as e:                 # noqa: BLE001 -- what `set -o pipefail` makes of a failing tee (src/jasper.sh:177-181)
            error_exit("Creating jellyfish database mer_counts%d.jf failed (%s)" % (kmer, e))
        _timing("mer_counts.jf complete (written beside the stages above)")
    log("Polished sequence is in %s.polished.fasta" % qfn)
    if table is not None:
        table.close()
    return 0


def _front_process():
    """OPT-IN (JASPER_AMD_FRONT=1; the default is what jasper.sh does: the command returns when everything, device memory
    included, has been released).  The end of a process that holds tens of GB of device memory takes ~0.1 s in the kernel
    (tools/probes/exit_probe.py: 0.04 s with nothing allocated, 0.10 s with 48 GB), spent after every output file is complete.  With
    the front, the command the user waits for never touches the GPU: it forks the worker before anything is loaded, waits for the
    worker's word that the outputs are complete (one byte on a pipe) and ends at once; the worker then ends on its own time -- a
    GPU job started right afterwards may find that memory not yet free, which is why a caller has to ask for this.  A worker that
    fails, is killed or exits with a status ends without the byte: the front waits for it and passes its status on.  SIGTERM /
    SIGHUP reaching the front are handed to the worker (which leaves through SystemExit, so its atexit clean-up runs); SIGINT is
    not forwarded -- a terminal's Ctrl-C reaches the whole foreground process group, worker included -- unless the worker is in
    another process group.
    Returns the pipe's write end in the worker, None when there is no front."""
    if (not hasattr(os, "fork") or "WORLD_SIZE" in os.environ or os.environ.get("JASPER_AMD_FRONT", "") not in ("1", "true", "yes")
            or os.environ.get("JASPER_AMD_SLOW_EXIT")):
        return None
    try:
        with open("/proc/self/maps") as f:
            maps = f.read()
        if "libamdhip64" in maps or "libhsa-runtime" in maps:
            return None
    except OSError:
        return None
    import signal
    sys.stdout.flush()
    sys.stderr.flush()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)

        def _leave(signum, _frame):         # (SIGTERM / SIGHUP: out through SystemExit -- atexit asks the GPU threads to stop and removes unfinished files)
            sys.exit(128 + signum)
        for sig in (signal.SIGTERM, signal.SIGHUP):
            try:
                signal.signal(sig, _leave)
            except (OSError, ValueError):
                pass
        return w
    os.close(w)

    def _hand_on(signum, _frame):
        try:
            if signum != signal.SIGINT or os.getpgid(pid) != os.getpgid(0):
                os.kill(pid, signum)
        except OSError:
            pass
    for sig in (signal.SIGINT, signal.SIGTERM, signal.SIGHUP):
        try:
            signal.signal(sig, _hand_on)
        except (OSError, ValueError):
            pass
    while True:
        try:
            word = os.read(r, 1)
            break
        except InterruptedError:
            continue
    if word:
        os._exit(0)
    while True:
        try:
            _, st = os.waitpid(pid, 0)
            break
        except InterruptedError:
            continue
        except ChildProcessError:
            os._exit(1)
    if os.WIFSIGNALED(st):
        signal.signal(os.WTERMSIG(st), signal.SIG_DFL)
        os.kill(os.getpid(), os.WTERMSIG(st))
        os._exit(128 + os.WTERMSIG(st))
    os._exit(os.WEXITSTATUS(st))


def main():
    argv = sys.argv[1:]
    if "--gpus" in argv and "WORLD_SIZE" not in os.environ:
        # extension: `--gpus N` = this driver as one process per GPU. Nothing has touched the GPU yet,
        # so the launcher is simply started as a child with the same arguments.
        import subprocess
        i = argv.index("--gpus")
        try:
            n = int(argv[i + 1])
        except (IndexError, ValueError):
            print("--gpus needs a number")
            sys.exit(1)
        if n > 1:
            port = os.environ.get("MASTER_PORT")
            if not port:                      # a free port, so that two runs on one node do not meet
                import socket
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
                sk.close()
            sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                                      "--master-addr", "127.0.0.1", "--master-port", port, "-m", "jasper_amd.cli"] + argv))
    done_fd = _front_process()
    rc = run(argv)
    if os.environ.get("JASPER_AMD_TIMING"):
        import time
        sys.stderr.write("[timing-abs] run() returned at %.6f\n" % time.time())
    if done_fd is not None and not rc:
        # (the worker of a front process, see _front_process: the outputs are complete -- tell the front, let go of the terminal's
        #  files, and leave the release of the device memory to this process's own end)
        sys.stdout.flush()
        sys.stderr.flush()
        try:
            os.write(done_fd, b"\0")
            for fd in (0, 1, 2, done_fd):
                os.close(fd)
        except OSError:
            pass
        os._exit(0)
    # Every output file is closed and under its final name.  A normal interpreter exit would now free tens of GB of device
    # memory allocation by allocation (hipFree of the table, the list workspaces, the pinned buffers: 0.1 s of a 0.7-s run);
    # the driver releases all of it when the process ends anyway.
    if not rc and "WORLD_SIZE" not in os.environ and not os.environ.get("JASPER_AMD_SLOW_EXIT"):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    sys.exit(rc)


if __name__ == "__main__":
    main()
