"""The assembly side of `jasper.sh` held natively (libjasper_hip.so: jasper_asm_*, jasper_amd/csrc/asmio.cpp): the FASTA is read
once into one host arena by several threads, chunk records (src/jasper.sh:155) are slices of it, the batch files (:156) are
written from it beside the counting, the polisher reads record text from it, and the polished records are written to
`$QUERY_FN.polished.fasta` (:220) -- and, on request, to the `_iter{P-1}_<batch>.fixed.fa` files (src/jasper.py:120-128) --
without ever becoming Python strings.  Only the ordinary file takes this path: AssemblyJob.open returns None for anything
else and cli.py applies the reference's line-by-line rules in Python, as before."""
import ctypes as C

from ._lib import check, lib


class AssemblyJob:
    def __init__(self, handle):
        self._L = lib()
        self._h = handle
        sb, nc, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self._L.jasper_asm_info(self._h, C.byref(sb), C.byref(nc), C.byref(nb)))
        self.sequence_bytes, self.n_contigs, self.n_bases = sb.value, nc.value, nb.value
        self.n_chunks = self.n_files = 0
        self._names = None

    @classmethod
    def open(cls, path, threads=0):
        """None when the file is not an ordinary FASTA (see include/jasper_hip.h: jasper_asm_open) -- or cannot be opened at all:
        the caller's own reader then gives the reference's message for it"""
        h = C.c_void_p()
        rc = lib().jasper_asm_open(str(path).encode(), int(threads), C.byref(h))
        if rc != 0 or not h:
            return None
        return cls(h)

    def close(self):
        if self._h:
            self._L.jasper_asm_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # noqa: BLE001 -- interpreter teardown
            pass

    # ---- contigs -------------------------------------------------------------------------------------------
    def contig_names(self):
        """first whitespace token of every header line that has a sequence, with its '>' (what perl -ane sees as $F[0])"""
        if self._names is None:
            out = []
            p, n = C.c_void_p(), C.c_uint64(0)
            for i in range(self.n_contigs):
                check(self._L.jasper_asm_contig(self._h, i, C.byref(p), C.byref(n), None))
                out.append(C.string_at(p, n.value).decode("ascii"))
            self._names = out
        return self._names

    # ---- src/jasper.sh:155-156 ------------------------------------------------------------------------------
    def split(self, batch_size, prefix, write_files=True, only_files=None):
        import numpy as np
        nchunks, nfiles = C.c_uint64(0), C.c_uint64(0)
        only = None
        if only_files is not None:
            only = np.ascontiguousarray(only_files, dtype=np.uint32)
        check(self._L.jasper_asm_split(self._h, int(batch_size), str(prefix).encode(), only.ctypes.data if only is not None else None,
                                       len(only) if only is not None else 0, 1 if write_files else 0, 0, C.byref(nchunks), C.byref(nfiles)))
        self.n_chunks, self.n_files = nchunks.value, nfiles.value
        self.prefix = str(prefix)
        self.chunk_contig = np.zeros(self.n_chunks, dtype=np.uint32)
        self.chunk_offset = np.zeros(self.n_chunks, dtype=np.uint64)
        self.chunk_len = np.zeros(self.n_chunks, dtype=np.uint64)
        self.chunk_file = np.zeros(self.n_chunks, dtype=np.uint32)
        check(self._L.jasper_asm_chunks(self._h, self.chunk_contig.ctypes.data, self.chunk_offset.ctypes.data, self.chunk_len.ctypes.data,
                                        self.chunk_file.ctypes.data))
        fb = np.zeros(self.n_files, dtype=np.uint64)
        check(self._L.jasper_asm_file_bytes(self._h, fb.ctypes.data))
        self.file_bytes = [int(v) for v in fb]
        # file f = records [file_first[f], file_first[f + 1])
        self.file_first = np.searchsorted(self.chunk_file, np.arange(self.n_files + 1, dtype=np.uint32)).tolist()
        return self.n_chunks, self.n_files

    def split_wait(self):
        """the batch files are complete (raises what their writer met: a full disk ...)"""
        check(self._L.jasper_asm_split_wait(self._h))

    def batch_file_name(self, f):
        return "%s.batch.%d.fa" % (self.prefix, f)

    def chunk_name(self, c):
        """the record's name as src/jasper.py:615-631 parse_fasta reads it from the batch file: 'name:offset' without the '>'"""
        return "%s:%d" % (self.contig_names()[int(self.chunk_contig[c])][1:], int(self.chunk_offset[c]))

    def records_of(self, files):
        out = []
        for f in files:
            out.extend(range(self.file_first[f], self.file_first[f + 1]))
        return out

    def chunk_text(self, c, polished=False):
        p, n = C.c_void_p(), C.c_uint64(0)
        check(self._L.jasper_asm_chunk_text(self._h, int(c), 1 if polished else 0, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    # ---- src/jasper.sh:207-212 ------------------------------------------------------------------------------
    def _files(self, files):
        import numpy as np
        return np.ascontiguousarray(list(files), dtype=np.uint32)

    def polish(self, table, files, solid_thre, passes, fix=True):
        """the records of the listed batch files through the polisher (KmerTable.polish_batch on text read from the arena);
        result chunk i = the i-th record of records_of(files)"""
        fl = self._files(files)
        res = C.c_void_p()
        rc = self._L.jasper_asm_polish(table._h, self._h, fl.ctypes.data, len(fl), int(solid_thre), int(passes), 1 if fix else 0, C.byref(res))
        return table._wrap_result(rc, res, len(self.records_of(files)), False)

    def pin(self, device=0):
        """optional: the arena registered with the GPU runtime + one pinned buffer for the polished text (waits for the runtime to
        start: meant for a thread beside the counting)"""
        check(self._L.jasper_asm_pin(self._h, int(device)))

    def take(self, result, files):
        """keep the polished text of `result` (of polish(table, files, ...)) in the job: result.seqs is empty afterwards"""
        fl = self._files(files)
        check(self._L.jasper_asm_take(self._h, result._h, fl.ctypes.data, len(fl)))

    def put(self, c, text):
        """the polished text of record c, given by the caller"""
        if isinstance(text, str):
            text = text.encode("latin-1")
        check(self._L.jasper_asm_put(self._h, int(c), text, len(text)))

    # ---- src/jasper.py:120-128 / src/jasper.sh:220 ----------------------------------------------------------
    def write_fixed(self, files, out_paths):
        fl = self._files(files)
        arr = (C.c_char_p * max(len(fl), 1))(*[str(p).encode() for p in out_paths])
        check(self._L.jasper_asm_write_fixed(self._h, fl.ctypes.data, arr, len(fl), 0))

    def polished_lens(self):
        import numpy as np
        lens = np.zeros(self.n_chunks, dtype=np.uint64)
        have = np.zeros(self.n_chunks, dtype=np.uint8)
        check(self._L.jasper_asm_polished_lens(self._h, lens.ctypes.data, have.ctypes.data))
        return lens, have

    def join(self, out_path, all_lens=None, mode=3):
        import numpy as np
        al = None
        if all_lens is not None:
            al = np.ascontiguousarray(all_lens, dtype=np.uint64)
            assert al.size == self.n_chunks
        check(self._L.jasper_asm_join(self._h, str(out_path).encode(), al.ctypes.data if al is not None else None, int(mode), 0))
