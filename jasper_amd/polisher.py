"""Host-side mirror of the reference's per-batch polisher `src/jasper.py` and of `src/jellyfish.py`.

Same names, argument meaning, output files and error behaviour as the reference; the scan/lookup/fix loop itself
runs on the GPU (libjasper_hip.so: jasper_polish_batch).  Reference lines are cited as src/jasper.py:N.
"""
import csv
import io
import math
import os
import sys

from .table import KmerTable

FIX_FIELDS = ['Contig', 'Base_coord', 'Original', 'Mutation']  # src/jasper.py:115


# What this process has itself just written to a file, kept so that the next stage need not read and parse it again (the file
# is still written: it is the stage's artefact and the restart point).  Valid only while the file's size and mtime are what they
# were when it was written.
_WRITTEN = {}


def remember(path, kind, value, final_path=None):
    """kind: "records" = what parse_fasta(path) returns, "events" = what cli._fasta_events(path) returns"""
    try:
        st = os.stat(path)
    except OSError:
        return
    _WRITTEN[(os.path.abspath(final_path or path), kind)] = (st.st_size, st.st_mtime_ns, value)


def recall(path, kind):
    e = _WRITTEN.get((os.path.abspath(path), kind))
    if e is None:
        return None
    try:
        st = os.stat(path)
    except OSError:
        return None
    return e[2] if (st.st_size, st.st_mtime_ns) == e[:2] else None


def wrap_lines(seq, num_per_line=60):
    """the text split_output's lines give when joined with and ended by a newline, for ASCII text; None otherwise"""
    try:
        raw = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
    except UnicodeEncodeError:
        return None
    if not raw:
        return b""
    import numpy as np
    a = np.frombuffer(raw, dtype=np.uint8)
    full = len(raw) // num_per_line
    out = np.empty((full, num_per_line + 1), dtype=np.uint8)
    out[:, :num_per_line] = a[:full * num_per_line].reshape(full, num_per_line)
    out[:, num_per_line] = 10
    tail = raw[full * num_per_line:]
    return out.tobytes() + (tail + b"\n" if tail else b"")


def parse_fasta(query_file):
    """src/jasper.py:615-631 -- ordered dict name -> sequence; name = first token without '>', only '\\n' stripped"""
    seq = {}
    name = None
    temp = []
    with open(query_file, "r") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    seq[name] = "".join(temp)
                name = line.split()[0][1:]
                temp = []
            else:
                if name is None:
                    # the reference stores this under "placeholder" and pops it again (:619,629)
                    continue
                temp.append(line.replace('\n', ''))
    if name is not None:
        seq[name] = "".join(temp)
    return seq


def split_output(seq, num_per_line=60):
    """src/jasper.py:142-147"""
    return [seq[num_per_line * i:num_per_line * (i + 1)] for i in range(math.ceil(len(seq) / num_per_line))]


def globalms_first(a, b):
    """Stand-in for Bio.pairwise2.align.globalms(a, b, 0, -1, -1, -1)[0][:2]  (src/jasper.py:309).

    PARITY UNPINNED: Biopython is a third-party dependency that is not part of the reference tree. Global alignment
    with match 0 and mismatch/gap -1; traceback from the end preferring gap-in-a, then diagonal, then gap-in-b, never
    a gap-in-a directly after a gap-in-b, first complete path wins.  It only decides which rows the ">k bad k-mers"
    branch writes to the fix CSV, never the polished sequence or the QV counters.
    """
    n, m = len(a), len(b)
    S = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        S[i][0] = -i
    for j in range(1, m + 1):
        S[0][j] = -j
    for i in range(1, n + 1):
        ai = a[i - 1]
        Si, Sp = S[i], S[i - 1]
        for j in range(1, m + 1):
            d = Sp[j - 1] + (0 if ai == b[j - 1] else -1)
            u = Sp[j] - 1
            l = Si[j - 1] - 1
            Si[j] = d if d >= u and d >= l else (u if u >= l else l)
    stack = [(n, m, 0, False, 0)]
    oa, ob = [], []
    while stack:
        i, j, opt, colgap, olen = stack.pop()
        del oa[olen:], ob[olen:]
        if i == 0 and j == 0:
            break
        cur = S[i][j]
        pushed = False
        while opt < 3 and not pushed:
            o = opt
            opt += 1
            if o == 0 and j > 0 and cur == S[i][j - 1] - 1 and not colgap:
                stack.append((i, j, opt, colgap, olen))
                oa.append('-'); ob.append(b[j - 1])
                stack.append((i, j - 1, 0, False, olen + 1)); pushed = True
            elif o == 1 and i > 0 and j > 0 and cur == S[i - 1][j - 1] + (0 if a[i - 1] == b[j - 1] else -1):
                stack.append((i, j, opt, colgap, olen))
                oa.append(a[i - 1]); ob.append(b[j - 1])
                stack.append((i - 1, j - 1, 0, False, olen + 1)); pushed = True
            elif o == 2 and i > 0 and cur == S[i - 1][j] - 1:
                stack.append((i, j, opt, colgap, olen))
                oa.append(a[i - 1]); ob.append('-')
                stack.append((i - 1, j, 0, True, olen + 1)); pushed = True
    return "".join(reversed(oa)), "".join(reversed(ob))


def rows_from_record(seqname, r):
    """fix records of one handle_bad_kmers call -> rows appended to fixed_bases_list (src/jasper.py:218-222,232-329)"""
    kind = r["kind"]
    if kind == "s":
        return [[seqname, r["index"], r["newc"], "s" + r["oldc"]]]
    if kind == "i":
        return [[seqname, r["index"], "-", "i" + r["oldc"] * r["rep"]]]
    if kind == "d":
        return [[seqname, r["index"], r["newc"] * r["rep"], "d-"]]
    if kind == "x":
        # the ">k bad k-mers" branch: the patch aligned against the text it replaced, one (coordinate, new, tag + old) triple per
        # column where the two differ -- a gap in the patch is a removed base ("i" + base, new = '-'), a gap in the old text a
        # restored one ("d-"), anything else a substitution ("s" + base); coordinates are alignment columns counted from the
        # segment's start (src/jasper.py:309-329).  handle_bad_kmers then writes ONE row: with a single difference the three
        # python LISTS go into the row as they are, otherwise the first two differences (src/jasper.py:218-222).
        new_row, old_row = globalms_first(r["patch"], r["orig"])
        diffs = []
        for col, (b_new, b_old) in enumerate(zip(new_row, old_row)):
            if b_new != b_old:
                tag = ("i" + b_old) if b_new == "-" else "d-" if b_old == "-" else ("s" + b_old)
                diffs.append((col + r["index"], b_new, tag))
        if len(diffs) == 1:
            at, b_new, tag = diffs[0]
            return [[seqname, at, [b_new], [tag]]]
        # (fewer than two differences left after that: the reference indexes an empty list, prints and exits 1 -- so does this)
        return [[seqname, at, b_new, tag] for at, b_new, tag in (diffs[0], diffs[1])]
    raise ValueError("unknown fix record kind %r" % kind)


def fix_csv_text(rows):
    """what csv.writer(f, delimiter=' ') writes for header + rows (src/jasper.py:116-119)"""
    buf = io.StringIO(newline="")
    w = csv.writer(buf, delimiter=' ')
    w.writerow(FIX_FIELDS)
    w.writerows(rows)
    return buf.getvalue()


def polish_batch(table, names, seqs, thre, num_iter, fix=True):
    """run one batch; returns (polished seqs, [rows of pass 0, rows of pass 1, ...], (bad0,total0,badP,totalP), result)"""
    res = table.polish_batch(seqs, thre, num_iter, fix=fix)
    rows = [[] for _ in range(num_iter)]
    for r in res.records:  # already ordered by chunk, then pass, then emission
        rows[r["pass_"]].append((r["chunk"], r["seqno"], rows_from_record(names[r["chunk"]], r)))
    out_rows = []
    for p in range(num_iter):
        flat = []
        for _, _, rr in sorted(rows[p], key=lambda x: (x[0], x[1])):
            flat.extend(rr)
        out_rows.append(flat)
    return res.seqs, out_rows, res.qv, res


def main(contigs, query_path, k, test, fix, fout, fixedout, db, thre, num_iter):
    """src/jasper.py:12-32 main() + :35-137 iteration(): same arguments; `db` is a KmerTable (HBM) instead of a .jf path.

    Writes, in the current directory, exactly the files the reference writes:
      _iter{i}_<fout> (CSV, CRLF), _iter{P-1}_<fixedout> (FASTA, 60 columns), {0,P}qValCalcHelper.csv (appended).
    Any failure prints and exits 1 like the reference's bare `except` (:27-32).
    """
    try:
        if not isinstance(db, KmerTable):
            raise TypeError("db must be a jasper_amd.KmerTable resident in HBM")
        if db.k != k:
            # QueryMerFile sets the global k from the DB header (JF::swig/mer_file.i:23): the DB wins
            k = db.k
        seq_dict = parse_fasta(query_path)
        names = list(seq_dict.keys())
        seqs = [seq_dict[n] for n in names]
        do_fix = bool(fix)
        fixed, rows, qv, _ = polish_batch(db, names, seqs, thre, num_iter, fix=do_fix)
        if test:                                                            # :107-111
            with open("0qValCalcHelper.csv", 'a') as f:
                f.write("{} {}\n".format(qv[0], qv[1]))
            if num_iter != 0:
                with open(str(num_iter) + "qValCalcHelper.csv", 'a') as f:
                    f.write("{} {}\n".format(qv[2], qv[3]))
        if do_fix:
            fo = os.path.split(fout)
            for ite in range(num_iter):                                     # :48-49,114-119
                with open(fo[0] + "_iter" + str(ite) + "_" + fo[1], 'w', newline='') as csvf:
                    csvf.write(fix_csv_text(rows[ite]))
            ff = os.path.split(fixedout)                                    # :39-40
            out_path = ff[0] + "_iter" + str(num_iter - 1) + "_" + ff[1]
            with open(out_path, 'w') as of:                                 # :120-128
                for seqname, seq in zip(names, fixed):
                    of.write(">{}\n".format(seqname))
                    lines = split_output(seq, 60)
                    if lines:
                        of.write("\n".join(lines) + "\n")   # (the reference writes them one by one: same bytes)
            return out_path
        return None
    except SystemExit:
        raise
    except BaseException:
        # what the reference's bare `except` leaves on stdout -- the line number, then the exc_info triple -- and exit status 1
        # (src/jasper.py:27-32; jasper.sh only looks at the status)
        info = sys.exc_info()
        print(info[2].tb_lineno)
        print(info)
        sys.exit(1)


def main_many(query_paths, k, test, fix, db, thre, num_iter):
    """what `xargs -P $THREADS jasper.py ...` (src/jasper.sh:207-212) leaves behind for a LIST of batch files, computed in
    ONE GPU call (chunk records are independent): per file the same artefacts as main() -- `_iter{i}_<file>.fix.csv`,
    `_iter{P-1}_<file>.fixed.fa.tmp`, one line per file in {0,P}qValCalcHelper.csv, in the order of `query_paths`."""
    try:
        if not isinstance(db, KmerTable):
            raise TypeError("db must be a jasper_amd.KmerTable resident in HBM")
        per_file = []
        names, seqs = [], []
        for qp in query_paths:
            d = recall(qp, "records")
            if d is None:
                d = parse_fasta(qp)
            per_file.append((qp, len(names), len(d)))
            names.extend(d.keys())
            seqs.extend(d.values())
        do_fix = bool(fix)
        res = db.polish_batch(seqs, thre, num_iter, fix=do_fix)         # (polish_batch() above would build every CSV row a second time)
        fixed = res.seqs
        rows_by_chunk = {}
        if do_fix:
            for r in res.records:
                rows_by_chunk.setdefault((r["pass_"], r["chunk"]), []).append((r["seqno"], rows_from_record(names[r["chunk"]], r)))
        outs = []
        for qp, first, n in per_file:
            if test:                                                            # :107-111, one line per process
                q = [0, 0, 0, 0]
                for c in range(first, first + n):
                    qc = res.qv_chunk(c)
                    q = [a + b for a, b in zip(q, qc)]
                with open("0qValCalcHelper.csv", 'a') as f:
                    f.write("{} {}\n".format(q[0], q[1]))
                if num_iter != 0:
                    with open(str(num_iter) + "qValCalcHelper.csv", 'a') as f:
                        f.write("{} {}\n".format(q[2], q[3]))
            if do_fix:
                fo = os.path.split(qp + ".fix.csv")
                for ite in range(num_iter):
                    flat = []
                    for c in range(first, first + n):
                        for _, rr in sorted(rows_by_chunk.get((ite, c), []), key=lambda x: x[0]):
                            flat.extend(rr)
                    with open(fo[0] + "_iter" + str(ite) + "_" + fo[1], 'w', newline='') as csvf:
                        csvf.write(fix_csv_text(flat))
                ff = os.path.split(qp + ".fixed.fa.tmp")
                out_path = ff[0] + "_iter" + str(num_iter - 1) + "_" + ff[1]
                events = []
                with open(out_path, 'w') as of:
                    for seqname, seq in zip(names[first:first + n], fixed[first:first + n]):
                        of.write(">{}\n".format(seqname))
                        wrapped = wrap_lines(seq, 60)
                        if wrapped is not None:
                            of.flush()
                            of.buffer.write(wrapped)
                        else:
                            lines = split_output(seq, 60)
                            if lines:
                                of.write("\n".join(lines) + "\n")
                        events.append(("h", ">" + seqname))
                        if seq:
                            events.append(("s", seq if isinstance(seq, str) else bytes(seq).decode("ascii", "replace")))
                # (what cli.join_polished would read back from the file under its final name)
                if all(isinstance(q, str) and q.isascii() and not any(c.isspace() for c in nm) and nm.isascii()
                       for nm, q in zip(names[first:first + n], fixed[first:first + n])):
                    remember(out_path, "events", events, final_path=out_path[:-4] if out_path.endswith(".tmp") else out_path)
                outs.append(out_path)
        return outs
    except SystemExit:
        raise
    except BaseException:
        # what the reference's bare `except` leaves on stdout -- the line number, then the exc_info triple -- and exit status 1
        # (src/jasper.py:27-32; jasper.sh only looks at the status)
        info = sys.exc_info()
        print(info[2].tb_lineno)
        print(info)
        sys.exit(1)


def _rows_by_pass_and_chunk(res, name_of):
    """{(pass, chunk): CSV rows in emission order} of a PolishResult, as rows_from_record makes them -- the plain kinds ('s', 'i',
    'd': all but a handful) straight from the record array's columns, without a dict per record"""
    import numpy as np
    raw = res._raw
    out = {}
    if not len(raw):
        return out
    order = np.lexsort((raw["seqno"], raw["pass_"], raw["chunk"]))          # by chunk, pass, emission
    chunk = raw["chunk"][order].tolist()
    pas = raw["pass_"][order].tolist()
    kind = raw["kind"][order].tolist()
    index = raw["index"][order].tolist()
    newc = raw["newc"][order].tolist()
    oldc = raw["oldc"][order].tolist()
    rep = raw["rep"][order].tolist()
    which = order.tolist()
    S, I, D = ord("s"), ord("i"), ord("d")
    names = {}
    for j in range(len(chunk)):
        c = chunk[j]
        nm = names.get(c)
        if nm is None:
            nm = names[c] = name_of(c)
        kd = kind[j]
        if kd == S:
            rows = [[nm, index[j], chr(newc[j]), "s" + chr(oldc[j])]]
        elif kd == I:
            rows = [[nm, index[j], "-", "i" + chr(oldc[j]) * rep[j]]]
        elif kd == D:
            rows = [[nm, index[j], chr(newc[j]) * rep[j], "d-"]]
        else:
            rows = rows_from_record(nm, res.record(which[j]))      # (the 'x' records carry aux bytes: the general decoder, for these alone)
        key = (pas[j], c)
        lst = out.get(key)
        if lst is None:
            out[key] = lst = []
        lst.extend(rows)
    return out


def main_many_job(job, files, k, test, fix, db, thre, num_iter, keep_fixed=False, on_taken=None):
    """main_many() for batch files of an assembly.AssemblyJob (numbers in `files`): the record text goes from the job's arena to the
    GPU and the polished text back into the job (job.take) without becoming Python objects; what is left per file is what
    main_many leaves -- `_iter{i}_<file>.fix.csv`, one line per file in {0,P}qValCalcHelper.csv -- except that
    `_iter{P-1}_<file>.fixed.fa.tmp` is written (natively, src/jasper.py:120-128) only when keep_fixed: its one reader is the join
    (src/jasper.sh:220), which a job does from memory."""
    try:
        if not isinstance(db, KmerTable):
            raise TypeError("db must be a jasper_amd.KmerTable resident in HBM")
        do_fix = bool(fix)
        import time
        tm = [time.perf_counter()]

        def mark(what):
            if os.environ.get("JASPER_AMD_TIMING"):
                now = time.perf_counter()
                sys.stderr.write("[timing-polish] %-34s %.3f s\n" % (what, now - tm[0]))
                tm[0] = now
        res = job.polish(db, files, thre, num_iter, fix=do_fix)
        mark("GPU call (%d files, device %.3f s)" % (len(files), res.seconds))
        if do_fix:
            job.take(res, files)
            if on_taken is not None:
                on_taken()          # (the caller may start writing the polished text while the rows below are made)
        recs = job.records_of(files)                  # result chunk i = record recs[i]
        rows_by_chunk = {}
        if do_fix:
            rows_by_chunk = _rows_by_pass_and_chunk(res, lambda c: job.chunk_name(recs[c]))
        mark("fix records -> rows (%d)" % res.n_records)
        at = 0
        outs = []
        for f in files:
            n = job.file_first[f + 1] - job.file_first[f]
            qp = job.batch_file_name(f)
            if test:                                                            # :107-111, one line per process
                q = [0, 0, 0, 0]
                for c in range(at, at + n):
                    q = [a + b for a, b in zip(q, res.qv_chunk(c))]
                with open("0qValCalcHelper.csv", 'a') as fh:
                    fh.write("{} {}\n".format(q[0], q[1]))
                if num_iter != 0:
                    with open(str(num_iter) + "qValCalcHelper.csv", 'a') as fh:
                        fh.write("{} {}\n".format(q[2], q[3]))
            if do_fix:
                fo = os.path.split(qp + ".fix.csv")
                for ite in range(num_iter):
                    flat = []
                    for c in range(at, at + n):
                        flat.extend(rows_by_chunk.get((ite, c), ()))
                    with open(fo[0] + "_iter" + str(ite) + "_" + fo[1], 'w', newline='') as csvf:
                        csvf.write(fix_csv_text(flat))
                ff = os.path.split(qp + ".fixed.fa.tmp")
                outs.append(ff[0] + "_iter" + str(num_iter - 1) + "_" + ff[1])
            at += n
        mark("QV lines + fix CSVs")
        if do_fix and keep_fixed:
            job.write_fixed(files, outs)
            mark("fixed files")
        return outs
    except SystemExit:
        raise
    except BaseException:
        # what the reference's bare `except` leaves on stdout -- the line number, then the exc_info triple -- and exit status 1
        # (src/jasper.py:27-32; jasper.sh only looks at the status)
        info = sys.exc_info()
        print(info[2].tb_lineno)
        print(info)
        sys.exit(1)


def threshold_from_histo_rows(rows):
    """The solid-k-mer threshold rule of src/jellyfish.py:8-22, on histogram rows (multiplicity, ..., n_distinct).

    The histogram is followed downhill from its first row; the first row whose n_distinct RISES ends the descent, and the
    threshold is half the multiplicity (rounded down) of the last row of the descent (0 if the descent was the first row
    alone).  Returns (what the script writes to stdout, its exit status): ("", 1) when that threshold is below 2,
    ("", 0) when the histogram never rises.
    """
    it = iter(rows)
    first = next(it, None)
    if first is None:
        return "", 0
    floor_n, half = int(first[-1]), 0
    for row in it:
        n = int(row[-1])
        if n > floor_n:
            return ("", 1) if half < 2 else (str(half), 0)
        floor_n, half = n, int(row[0]) // 2
    return "", 0


def threshold_from_histo_file(path):
    rows = []
    with open(path, 'r') as histo:
        for row in csv.reader(histo, delimiter=' '):
            rows.append(row)
    return threshold_from_histo_rows(rows)
