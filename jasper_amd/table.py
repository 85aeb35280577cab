"""KmerTable: the HBM-resident canonical k-mer count table and the operations of the hot path on it.

Host-side mirror of what the reference does through `jellyfish count/histo` (src/jasper.sh:177,189) and the SWIG
module `dna_jellyfish` (JF::swig/mer_file.i, JF::swig/mer_dna.i).  All compute happens in libjasper_hip.so;
nothing here falls back to the CPU.
"""
import ctypes as C

from . import _lib
from ._lib import FixRec, check


class PolishResult:
    """what one `jasper.py` process produces for one batch file (src/jasper.py:107-128).

    The polished texts stay in the C result object until asked for: `seq_view(i)` is a zero-copy memoryview,
    `seqs` materialises python objects (str if the inputs were str, else bytes) on first use."""

    def __init__(self, lib, handle, n_chunks, want_str, raw_records, aux, qv, lookups, seconds):
        self._L = lib
        self._h = handle
        self._n = n_chunks
        self._want_str = want_str
        self._seqs = None
        self._raw = raw_records     # numpy structured array (FixRec layout), ordered by chunk, pass, emission
        self._records = None
        self.aux = aux              # per chunk: bytes referenced by its 'x' records
        self.qv = qv                # (bad0, total0, badP, totalP)
        self.lookups = lookups
        self.seconds = seconds
        self.segments = 0
        self.respeculated = 0
        self.retried = False

    def __del__(self):
        try:
            if self._h:
                self._L.jasper_result_free(self._h)
                self._h = None
        except Exception:
            pass

    def seq_view(self, i):
        p = C.c_void_p()
        ln = C.c_int64(0)
        check(self._L.jasper_result_seq(self._h, i, C.byref(p), C.byref(ln)))
        if not ln.value:
            return memoryview(b"")
        return memoryview((C.c_char * ln.value).from_address(p.value)).cast("B")

    def seq_len(self, i):
        ln = C.c_int64(0)
        check(self._L.jasper_result_seq_len(self._h, i, C.byref(ln)))
        return ln.value

    def qv_chunk(self, i):
        """(bad0, total0, badP, totalP) of chunk record i alone"""
        q = (C.c_int64 * 4)()
        check(self._L.jasper_result_qv_chunk(self._h, i, q))
        return tuple(q)

    def seq_device(self, i):
        """(device pointer, length) of polished chunk i while the text is still in HBM (polish_batch_device results, until
        the next polish call on the same table or the first host access)"""
        p = C.c_void_p()
        ln = C.c_int64(0)
        check(self._L.jasper_result_seq_device(self._h, i, C.byref(p), C.byref(ln)))
        return (p.value or 0), ln.value

    @property
    def seqs(self):
        if self._seqs is None:
            out = []
            for i in range(self._n):
                raw = bytes(self.seq_view(i))
                out.append(raw.decode("latin-1") if self._want_str else raw)
            self._seqs = out
        return self._seqs

    @property
    def n_records(self):
        return len(self._raw)

    def record(self, i):
        """record i as a dict (chunk, pass_, seqno, kind, index, newc, oldc, rep[, patch, orig])"""
        e = self._raw[i]
        d = dict(chunk=int(e["chunk"]), pass_=int(e["pass_"]), seqno=int(e["seqno"]), kind=chr(int(e["kind"])), index=int(e["index"]),
                 newc=chr(int(e["newc"])), oldc=chr(int(e["oldc"])), rep=int(e["rep"]))
        if d["kind"] == "x":
            a = self.aux[d["chunk"]]
            o, n = int(e["aux_off"]), int(e["aux_len"])
            d["patch"] = a[o:o + n].decode("latin-1")
            d["orig"] = a[o + n:o + n + d["rep"]].decode("latin-1")
        return d

    @property
    def records(self):
        """list of dicts (see record()); decoded on first use"""
        if self._records is None:
            self._records = [self.record(i) for i in range(len(self._raw))]
        return self._records


FIXREC_DTYPE = [("index", "<i8"), ("chunk", "<u4"), ("seqno", "<u4"), ("pass_", "u1"), ("kind", "u1"), ("newc", "u1"), ("oldc", "u1"),
                ("rep", "<u4"), ("aux_off", "<u4"), ("aux_len", "<u4")]


class KmerTable:
    def __init__(self, k, min_slots=1 << 20, device=0):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.k = int(k)
        self.device = int(device)
        check(self._L.jasper_table_create(self.k, int(min_slots), self.device, C.byref(self._h)))

    @classmethod
    def from_jf(cls, path, device=0):
        """jf.QueryMerFile(path): open a Jellyfish binary/sorted DB into HBM; k comes from the file (JF::swig/mer_file.i:18-36)"""
        self = cls.__new__(cls)
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.device = int(device)
        check(self._L.jasper_table_load_jf(path.encode(), self.device, C.byref(self._h)))
        self.k = self.info()["k"]
        return self

    @classmethod
    def from_jf_part(cls, path, part, nparts, device=0):
        """records [n*part/nparts, n*(part+1)/nparts) of a Jellyfish DB: one GPU's shard of it"""
        self = cls.__new__(cls)
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.device = int(device)
        check(self._L.jasper_table_load_jf_part(path.encode(), self.device, int(part), int(nparts), C.byref(self._h)))
        self.k = self.info()["k"]
        return self

    def write_jf(self, path, cmdline=()):
        """write the table as a Jellyfish binary/sorted DB (what `jellyfish count -o` produces, src/jasper.sh:177)"""
        args = [a.encode() for a in cmdline]
        arr = (C.c_char_p * max(len(args), 1))(*args)
        check(self._L.jasper_table_write_jf(self._h, path.encode(), arr, len(args)))

    def write_jf_piece(self, path, cmdline, size_log2, what):
        """what: 0 = whole file, 1 = sorted records only (one GPU's piece), 2 = header only; `size` of the file = 2^size_log2"""
        args = [a.encode() for a in cmdline]
        arr = (C.c_char_p * max(len(args), 1))(*args)
        check(self._L.jasper_table_write_jf_piece(self._h, path.encode(), arr, len(args), int(size_log2), int(what)))

    def export_file_ranges(self, dev_ptr, cap_entries, n_ranges, size_log2):
        """entries grouped by their range in the order of a binary/sorted file of size 2^size_log2 (layout as export_owner)"""
        counts = (C.c_uint64 * int(n_ranges))()
        check(self._L.jasper_table_export_file_ranges(self._h, C.c_void_p(dev_ptr), int(cap_entries), int(n_ranges), int(size_log2), counts))
        return [int(c) for c in counts]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.jasper_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- counting (jellyfish count -C) -------------------------------------------------------------
    def count_files(self, paths):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        check(self._L.jasper_count_reads_files(self._h, arr, len(paths)))

    def count_file_ranges(self, ranges):
        """ranges: list of (path, begin, end) byte ranges (end < 0: to the end of the file) -- one GPU's shard of the reads"""
        n = len(ranges)
        arr = (C.c_char_p * max(n, 1))(*[r[0].encode() for r in ranges])
        b = (C.c_int64 * max(n, 1))(*[int(r[1]) for r in ranges])
        e = (C.c_int64 * max(n, 1))(*[int(r[2]) for r in ranges])
        check(self._L.jasper_count_reads_file_ranges(self._h, arr, b, e, n))

    # ---- the read files as a feed of base batches in HBM (include/jasper_hip.h, jasper_read_feed_*) ----
    def feed_start(self, ranges):
        """ranges as for count_file_ranges; this table only lends its device and buffers"""
        n = len(ranges)
        arr = (C.c_char_p * max(n, 1))(*[r[0].encode() for r in ranges])
        b = (C.c_int64 * max(n, 1))(*[int(r[1]) for r in ranges])
        e = (C.c_int64 * max(n, 1))(*[int(r[2]) for r in ranges])
        check(self._L.jasper_read_feed_start(self._h, arr, b, e, n))

    def feed_next(self):
        """(device pointer, bytes) of the next batch of bases; bytes == 0: the stream has ended"""
        p, n = C.c_void_p(0), C.c_uint64(0)
        check(self._L.jasper_read_feed_next(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def feed_release(self):
        check(self._L.jasper_read_feed_release(self._h))

    def last_ingest(self):
        """(text bytes parsed on the GPU, text bytes parsed by the host state machine) of the last count_files call"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.jasper_last_ingest(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def count_text(self, text):
        if isinstance(text, str):
            text = text.encode()
        check(self._L.jasper_count_reads_text(self._h, text, len(text)))

    def count_bases(self, bases):
        if isinstance(bases, str):
            bases = bases.encode()
        check(self._L.jasper_count_bases(self._h, bases, len(bases)))

    def count_bases_device(self, dev_ptr, n):
        check(self._L.jasper_count_bases_device(self._h, C.c_void_p(dev_ptr), int(n)))

    def count_timing(self):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        check(self._L.jasper_last_count_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    STAGE_NAMES = {1: ("part1_kernel", "part2_kernel", "region_insert_kernel", "deferred_import3h_kernel"),
                   3: ("part1_kernel", "part2_by_owner_kernel", "region_insert_kernel", "(unused)", "deferred_import3h_kernels"),
                   0: ()}

    def count_stages(self):
        """(ms per kernel stage of the atomic-free counting path the last piece took, launches that took such a path);
        count_path() names the path, STAGE_NAMES[path] the stages"""
        ms = (C.c_double * 8)()
        n = C.c_uint64(0)
        path = C.c_int(0)
        check(self._L.jasper_last_count_stages(self._h, ms, C.byref(n), C.byref(path)))
        self._count_path = path.value
        return list(ms)[:len(self.STAGE_NAMES.get(path.value, ()))] or list(ms)[:5], n.value

    def count_path(self):
        """3 = region lists exchanged between GPUs, 1 = one record per occurrence through partition passes and LDS images, 0 = count_kernel"""
        self.count_stages()
        return self._count_path

    def clear(self):
        check(self._L.jasper_table_clear(self._h))

    def sync(self):
        check(self._L.jasper_table_sync(self._h))

    def info(self):
        k = C.c_int(0)
        slots = C.c_uint64(0)
        distinct = C.c_uint64(0)
        occ = C.c_uint64(0)
        check(self._L.jasper_table_info(self._h, C.byref(k), C.byref(slots), C.byref(distinct), C.byref(occ)))
        return dict(k=k.value, slots=slots.value, distinct=distinct.value, occurrences=occ.value)

    # ---- jellyfish histo ---------------------------------------------------------------------------
    def histogram_is_fused(self):
        """True if the last counting call already produced the histogram (binned while the final counts were written)"""
        return bool(self._L.jasper_histogram_is_fused(self._h))

    def histogram(self):
        out = (C.c_uint64 * 10002)()
        check(self._L.jasper_histogram(self._h, out))
        return list(out)

    def histogram_part(self, part, nparts):
        """multiplicity histogram of the keys of owner partition part/nparts only"""
        out = (C.c_uint64 * 10002)()
        check(self._L.jasper_histogram_part(self._h, int(part), int(nparts), out))
        return list(out)

    def histo_rows(self):
        """non-zero rows (multiplicity, n_distinct) as `jellyfish histo` prints them (JF::sub_commands/histo_main.cc:82-84)"""
        import numpy as np
        out = (C.c_uint64 * 10002)()
        check(self._L.jasper_histogram(self._h, out))
        h = np.frombuffer(out, dtype=np.uint64)
        nz = np.flatnonzero(h[1:]) + 1
        return list(zip(nz.tolist(), h[nz].tolist()))

    # ---- qf[MerDNA(s).get_canonical()] -------------------------------------------------------------
    def lookup(self, strings):
        n = len(strings)
        if n == 0:
            return []
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
        offs = (C.c_int64 * (n + 1))()
        tot = 0
        for i, b in enumerate(bs):
            offs[i] = tot
            tot += len(b)
        offs[n] = tot
        out = (C.c_uint32 * n)()
        check(self._L.jasper_lookup(self._h, b"".join(bs), offs, n, out))
        return list(out)

    # ---- merge support -----------------------------------------------------------------------------
    def export_entries(self):
        """numpy uint64 array [n,3]: mixed-hash hi, lo, count"""
        import numpy as np
        n = C.c_uint64(0)
        check(self._L.jasper_table_export(self._h, C.byref(n), None))
        arr = np.zeros((max(int(n.value), 1), 3), dtype=np.uint64)
        cap = C.c_uint64(arr.shape[0])
        check(self._L.jasper_table_export(self._h, C.byref(cap), arr.ctypes.data_as(C.POINTER(C.c_uint64))))
        return arr[: int(cap.value)]

    def import_entries(self, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        if arr.size == 0:
            return
        check(self._L.jasper_table_import(self._h, arr.ctypes.data_as(C.POINTER(C.c_uint64)), arr.shape[0]))

    def export_device(self):
        n = C.c_uint64(0)
        p = C.c_void_p()
        check(self._L.jasper_table_export_device(self._h, C.byref(n), C.byref(p)))
        return p.value, int(n.value)

    def export_to(self, dev_ptr, cap_entries):
        n = C.c_uint64(0)
        check(self._L.jasper_table_export_to(self._h, C.c_void_p(dev_ptr), int(cap_entries), C.byref(n)))
        return int(n.value)

    def export_packed(self, dev_ptr, cap_entries, part=0, nparts=1):
        """16-byte exchange entries of slot-range partition part/nparts into device memory; returns how many exist"""
        n = C.c_uint64(0)
        check(self._L.jasper_table_export_packed(self._h, C.c_void_p(dev_ptr), int(cap_entries), C.byref(n), int(part), int(nparts)))
        return int(n.value)

    def import_packed(self, dev_ptr, n, mode=0):
        check(self._L.jasper_table_import_packed(self._h, C.c_void_p(dev_ptr), int(n), int(mode)))

    def import_packed_multi(self, dev_ptrs, counts):
        """add several entry lists (device pointers, entry counts) in one sweep over the table"""
        n = len(dev_ptrs)
        ps = (C.c_void_p * n)(*[int(p) for p in dev_ptrs])
        cs = (C.c_uint64 * n)(*[int(c) for c in counts])
        check(self._L.jasper_table_import_packed_multi(self._h, ps, cs, n))

    def reserve(self, min_slots):
        check(self._L.jasper_table_reserve(self._h, int(min_slots)))

    def fit(self, max_load=0.5):
        check(self._L.jasper_table_fit(self._h, float(max_load)))

    # ---- counting as an exchange of region lists (include/jasper_hip.h, jasper_count_exchange_*) -------
    def exchange_plan(self, piece_max, n_owners, records_max=0):
        """None when this table / piece size / k has no exchange geometry, else a dict of buffer sizes"""
        out = (C.c_uint64 * 8)()
        rc = self._L.jasper_count_exchange_plan(self._h, int(piece_max), int(records_max), int(n_owners), out)
        if rc == 1:
            return None
        check(rc)
        names = ("records_per_owner", "counts_per_owner", "deferred_cap", "p1", "p2", "region_bits", "slices", "slice_cap")
        d = dict(zip(names, (int(v) for v in out)))
        d["p2"], d["p2_owner"] = d["p2"] & 0xFF, d["p2"] >> 8      # second-level bits split by the senders / left to the owner's extra pass
        return d

    def exchange_scan(self, d_bases, n, pos, end, piece_max, n_owners, d_deferred, deferred_cap):
        """first pass (returns when it is done): the number of k-mer occurrences found in [pos, end)"""
        rec = C.c_uint64(0)
        check(self._L.jasper_count_exchange_scan(self._h, C.c_void_p(d_bases), int(n), int(pos), int(end), int(piece_max), int(n_owners), C.c_void_p(d_deferred),
                                                 int(deferred_cap), C.byref(rec)))
        return rec.value

    def exchange_partition(self, piece_max, records_max, n_owners, d_send, d_send_counts, d_deferred, deferred_cap):
        check(self._L.jasper_count_exchange_partition(self._h, int(piece_max), int(records_max), int(n_owners), C.c_void_p(d_send), C.c_void_p(d_send_counts),
                                                      C.c_void_p(d_deferred), int(deferred_cap)))

    def exchange_dedupe(self, piece_max, records_max, n_owners, d_send, d_send_counts):
        """the lists of the send buffers deduplicated in place: (records in the fullest list, count bits), or None when the
        geometry has no bits for the counts (nothing done)"""
        mx, cb = C.c_uint32(0), C.c_int(0)
        rc = self._L.jasper_count_exchange_dedupe(self._h, int(piece_max), int(records_max), int(n_owners), C.c_void_p(d_send), C.c_void_p(d_send_counts),
                                                  C.byref(mx), C.byref(cb))
        if rc == 1:
            return None
        check(rc)
        return mx.value, cb.value

    def exchange_insert(self, d_recv, d_recv_counts, piece_max, records_max, n_owners, self_index, d_deferred_all=0, n_deferred_all=0, whole_input=False,
                        slice_cap=0, count_bits=0):
        check(self._L.jasper_count_exchange_insert(self._h, C.c_void_p(d_recv), C.c_void_p(d_recv_counts), int(piece_max), int(records_max), int(n_owners),
                                                   int(self_index), C.c_void_p(d_deferred_all or None), int(n_deferred_all), 1 if whole_input else 0,
                                                   int(slice_cap), int(count_bits)))

    # ---- owner-sharded table (include/jasper_hip.h, "Owner-sharded table") ----------------------------
    def export_owner(self, dev_ptr, cap_entries, n_owners):
        """all entries grouped by owner into device memory (segment o at dev_ptr + o*cap*16); returns the n counts"""
        counts = (C.c_uint64 * int(n_owners))()
        check(self._L.jasper_table_export_owner(self._h, C.c_void_p(dev_ptr), int(cap_entries), int(n_owners), counts))
        return [int(c) for c in counts]

    def ipc_handle(self):
        buf = C.create_string_buffer(64)
        check(self._L.jasper_table_ipc_handle(self._h, buf))
        return buf.raw

    def attach_ipc(self, handles, self_index):
        """handles: list of 64-byte IPC handles in owner order (the entry at self_index is not used)"""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == 64 * len(handles)
        check(self._L.jasper_table_attach_ipc(self._h, blob, len(handles), int(self_index)))

    def attach_tables(self, shards, self_index):
        """shards: the owners' Table objects (same process, same device) in owner order"""
        arr = (C.c_void_p * len(shards))(*[t._h for t in shards])
        check(self._L.jasper_table_attach_tables(self._h, arr, len(shards), int(self_index)))
        self._shard_refs = list(shards)      # keep the owners alive while their slot arrays are read through this table

    def release_retired(self):
        """free slot arrays this table has outgrown after their IPC handles were given out (call once every owner has attached
        to the new ones)"""
        check(self._L.jasper_table_release_retired(self._h))

    def detach(self):
        check(self._L.jasper_table_detach(self._h))
        self._shard_refs = None

    def import_device(self, dev_ptr, n):
        check(self._L.jasper_table_import_device(self._h, C.c_void_p(dev_ptr), int(n)))

    def device_free(self, dev_ptr):
        check(self._L.jasper_device_free(self._h, C.c_void_p(dev_ptr)))

    # ---- one batch through the polisher -------------------------------------------------------------
    def polish_batch(self, seqs, solid_thre, passes, fix=True):
        n = len(seqs)
        want_str = any(isinstance(s, str) for s in seqs)       # bytes in -> bytes out (no 1-byte-per-char round trip)
        bs = [s.encode("latin-1") if isinstance(s, str) else (s if isinstance(s, bytes) else bytes(s)) for s in seqs]
        cs = (C.c_char_p * max(n, 1))(*bs)
        lens = (C.c_int64 * max(n, 1))(*[len(b) for b in bs])
        res = C.c_void_p()
        rc = self._L.jasper_polish_batch(self._h, n, cs, lens, int(solid_thre), int(passes), 1 if fix else 0, C.byref(res))
        return self._wrap_result(rc, res, n, want_str)

    def polish_batch_device(self, d_text, offsets, solid_thre, passes, fix=True):
        """chunk records already in HBM: d_text is a device pointer (int) or an object with .data_ptr() holding the chunk
        texts back to back, offsets the n+1 chunk boundaries.  The polished text stays in HBM (PolishResult.seq_device)
        and is copied to the host on first use of seq_view / seqs."""
        n = len(offsets) - 1
        ptr = d_text.data_ptr() if hasattr(d_text, "data_ptr") else int(d_text)
        offs = (C.c_int64 * (n + 1))(*[int(o) for o in offsets])
        res = C.c_void_p()
        rc = self._L.jasper_polish_batch_device(self._h, n, C.c_void_p(ptr), offs, int(solid_thre), int(passes), 1 if fix else 0, C.byref(res))
        return self._wrap_result(rc, res, n, False)

    def _wrap_result(self, rc, res, n, want_str):
        try:
            check(rc)
            import numpy as np
            aux = []
            for i in range(n):
                ap = C.c_void_p()
                an = C.c_uint64(0)
                check(self._L.jasper_result_aux(res, i, C.byref(ap), C.byref(an)))
                aux.append(C.string_at(ap, an.value) if an.value else b"")
            rp = C.POINTER(FixRec)()
            rn = C.c_uint64(0)
            check(self._L.jasper_result_records(res, C.byref(rp), C.byref(rn)))
            if rn.value:
                raw = np.frombuffer(C.string_at(rp, rn.value * C.sizeof(FixRec)), dtype=FIXREC_DTYPE).copy()
            else:
                raw = np.zeros(0, dtype=FIXREC_DTYPE)
            qv = (C.c_int64 * 4)()
            check(self._L.jasper_result_qv(res, qv))
            nl = C.c_uint64(0)
            check(self._L.jasper_result_lookups(res, C.byref(nl)))
            secs = self._L.jasper_result_seconds(res)
            nseg, nredo = C.c_uint64(0), C.c_uint64(0)
            check(self._L.jasper_result_segments(res, C.byref(nseg), C.byref(nredo)))
            pr = PolishResult(self._L, res, n, want_str, raw, aux, tuple(qv), nl.value, secs)
            res = None   # owned by the PolishResult from here on
            pr.segments, pr.respeculated = nseg.value, nredo.value
            pr.retried = bool(self._L.jasper_result_retried(pr._h))
            return pr
        finally:
            if res:
                self._L.jasper_result_free(res)
