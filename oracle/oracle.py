"""ctypes front end of the CPU oracle (oracle/jasper_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only
as the checker. The product (jasper_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjasper_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "jasper_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.jo_db_new.restype = C.c_void_p
        L.jo_db_new.argtypes = [C.c_int]
        L.jo_db_free.argtypes = [C.c_void_p]
        L.jo_db_k.argtypes = [C.c_void_p]
        L.jo_db_distinct.restype = C.c_uint64
        L.jo_db_distinct.argtypes = [C.c_void_p]
        L.jo_db_count_bases.restype = C.c_uint64
        L.jo_db_count_bases.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.jo_db_count_text.restype = C.c_int
        L.jo_db_count_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.jo_db_query.restype = C.c_uint32
        L.jo_db_query.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.jo_db_add_kmer.restype = C.c_int
        L.jo_db_add_kmer.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.jo_db_histo.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.jo_db_next.restype = C.c_int
        L.jo_db_next.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_uint64)]
        L.jo_encode.restype = C.c_int
        L.jo_encode.argtypes = [C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_uint64)]
        L.jo_revcomp.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.jo_canonical.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.jo_threshold.restype = C.c_int
        L.jo_threshold.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t]
        L.jo_polish_batch.restype = C.c_int
        L.jo_polish_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                      C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_uint64)]
        L.jo_free.argtypes = [C.c_void_p]
        L.jo_mt_db_new.restype = C.c_void_p
        L.jo_mt_db_new.argtypes = [C.c_int, C.c_int]
        L.jo_mt_count_bases.restype = C.c_uint64
        L.jo_mt_count_bases.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.jo_mt_polish_batch.restype = C.c_int
        L.jo_mt_polish_batch.argtypes = L.jo_polish_batch.argtypes + [C.c_int]
        L.malloc_copy = None
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


class OracleDB:
    """canonical k-mer -> count map built the way `jellyfish count -C` defines it"""

    def __init__(self, k, threads=0):
        """threads > 0: the multi-threaded driver (jo_mt_*): counting and polishing divided over that many host threads,
        the map split by key owner; same results as the plain map"""
        self.k = k
        self.threads = threads
        self._h = lib().jo_mt_db_new(k, threads) if threads > 0 else lib().jo_db_new(k)
        if not self._h:
            raise ValueError("bad k")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().jo_db_free(self._h)
            self._h = None

    def count_bases(self, b):
        if isinstance(b, str):
            b = b.encode()
        if self.threads > 0:
            buf = C.c_char_p(b) if isinstance(b, bytes) else None
            ptr = C.cast(buf, C.c_void_p) if buf is not None else C.c_void_p(b.ctypes.data)      # bytes, or a numpy uint8 array
            return lib().jo_mt_count_bases(self._h, ptr, len(b))
        return lib().jo_db_count_bases(self._h, b, len(b))

    def count_text(self, t):
        if isinstance(t, str):
            t = t.encode()
        n = C.c_uint64(0)
        rc = lib().jo_db_count_text(self._h, t, len(t), C.byref(n))
        if rc == -1:
            raise RuntimeError("Unsupported format")
        if rc == -2:
            raise RuntimeError("Invalid fastq sequence")
        return n.value

    def add_kmer(self, kmer, count):
        if lib().jo_db_add_kmer(self._h, kmer.encode() if isinstance(kmer, str) else kmer, count) != 0:
            raise ValueError("bad k-mer")

    def query(self, s):
        if isinstance(s, str):
            s = s.encode()
        return lib().jo_db_query(self._h, s, len(s))

    def distinct(self):
        return lib().jo_db_distinct(self._h)

    def histo(self):
        out = (C.c_uint64 * 10002)()
        lib().jo_db_histo(self._h, out)
        return list(out)

    def items(self):
        cur = C.c_uint64(0)
        buf = C.create_string_buffer(self.k + 1)
        cnt = C.c_uint64(0)
        while lib().jo_db_next(self._h, C.byref(cur), buf, C.byref(cnt)):
            yield buf.value.decode(), cnt.value

    def polish_batch(self, names, seqs, solid_thre, passes, fix=True):
        """returns (fixed seqs, [csv rows text per fixing pass], (bad0,total0,badP,totalP), n_lookups)"""
        n = len(seqs)
        cn = (C.c_char_p * max(n, 1))(*[x.encode() for x in names])
        cs = (C.c_void_p * max(n, 1))()
        for i, s in enumerate(seqs):
            b = s.encode() if isinstance(s, str) else s
            p = _libc.malloc(len(b) + 1)
            C.memmove(p, b + b"\0", len(b) + 1)
            cs[i] = p
        csv = (C.c_void_p * max(passes, 1))()
        qv = (C.c_int64 * 4)()
        nl = C.c_uint64(0)
        if self.threads > 0:
            rc = lib().jo_mt_polish_batch(self._h, self.k, n, cn, cs, solid_thre, passes, 1 if fix else 0, csv, qv, C.byref(nl), self.threads)
        else:
            rc = lib().jo_polish_batch(self._h, self.k, n, cn, cs, solid_thre, passes, 1 if fix else 0, csv, qv, C.byref(nl))
        out = []
        for i in range(n):
            out.append(C.string_at(cs[i]).decode())
            lib().jo_free(cs[i])
        rows = []
        for p in range(passes):
            if csv[p]:
                rows.append(C.string_at(csv[p]).decode())
                lib().jo_free(csv[p])
            else:
                rows.append("")
        if rc != 0:
            raise RuntimeError("reference would exit(1): code %d" % rc)
        return out, rows, tuple(qv), nl.value


def encode(k, s):
    out = (C.c_uint64 * 2)()
    b = s.encode() if isinstance(s, str) else s
    t = lib().jo_encode(k, b, len(b), out)
    return t, out[0] | (out[1] << 64)


def revcomp(k, v):
    i = (C.c_uint64 * 2)(v & (2**64 - 1), v >> 64)
    o = (C.c_uint64 * 2)()
    lib().jo_revcomp(k, i, o)
    return o[0] | (o[1] << 64)


def canonical(k, v):
    i = (C.c_uint64 * 2)(v & (2**64 - 1), v >> 64)
    o = (C.c_uint64 * 2)()
    lib().jo_canonical(k, i, o)
    return o[0] | (o[1] << 64)


def threshold(rows):
    """rows: list of (multiplicity, n_distinct). Returns int threshold, None (nothing printed), or raises SystemExit(1)"""
    n = len(rows)
    m = (C.c_uint64 * max(n, 1))(*[r[0] for r in rows])
    d = (C.c_uint64 * max(n, 1))(*[r[1] for r in rows])
    t = lib().jo_threshold(m, d, n)
    if t == -1:
        raise SystemExit(1)
    return None if t == 0 else t
