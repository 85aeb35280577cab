"""A `dna_jellyfish`-shaped module (the SWIG binding the reference imports, src/jasper.py:10) served from the HBM table.

    import jasper_amd.compat.dna_jellyfish as jf          # or put jasper_amd/compat on PYTHONPATH: `import dna_jellyfish as jf`
    qf = jf.QueryMerFile("mer_counts37.jf")               # src/jasper.py:15   (or QueryMerFile(a KmerTable that was counted into))
    occ = qf[jf.MerDNA(window).get_canonical()]           # src/jasper.py:70-71

What is restated here (the classes' names, constructor arguments, return types and error wording):
  * JF::swig/mer_file.i:12-43  QueryMerFile(path): RuntimeError "Can't open file '<path>'" (:21) / "Unsupported format '<f>'" (:34);
    opening a DB sets the process-wide k-mer length from its header (:23); qf[mer] -> unsigned int, 0 for an absent k-mer.
  * JF::swig/mer_dna.i:12-19   MerDNA(const char*): no validation -- the string is cut at the first character that is not one of
    ACGTacgt (or at k characters) and right-filled with 'A' (JF::include/jellyfish/mer_dna.hpp:525-542); k is a class-wide
    setting (JF::include/jellyfish/mer_dna.hpp:660-669), 22 until something sets it.
Only string handling happens on the host; every count comes from `jasper_lookup` (include/jasper_hip.h) on the GPU -- there is
no CPU table.  One lookup per `qf[mer]` is one kernel launch: a caller that has many k-mers uses `qf.counts(mers)`, and the
polisher itself does not come through here at all (jasper_polish_batch runs the whole scan on the device).
"""
from .. import _lib
from ..table import KmerTable

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
_k = [22]            # JF::include/jellyfish/mer_dna.hpp:660-669 (static k_, default 22)


class MerDNA:
    """a k-mer of the current class-wide length, held as its upper-case string"""
    __slots__ = ("_s",)

    def __init__(self, s=None):
        if isinstance(s, MerDNA):
            self._s = s._s
            return
        k = _k[0]
        if s is None:                                   # MerDNA(): all zero bits = poly-A
            self._s = "A" * k
            return
        if isinstance(s, bytes):
            s = s.decode("latin-1")
        out = []
        for ch in s[:k]:
            u = ch.upper()
            if u not in _COMP:                          # the first negative code ends the copy (mer_dna.hpp:525-542)
                break
            out.append(u)
        self._s = "".join(out) + "A" * (k - len(out))

    @staticmethod
    def k(new_k=None):
        """MerDNA.k() -> current length; MerDNA.k(n) sets it and returns it (JF::swig/mer_dna.i: static unsigned int k(unsigned int))"""
        if new_k is not None:
            _k[0] = int(new_k)
        return _k[0]

    def get_reverse_complement(self):
        m = MerDNA.__new__(MerDNA)
        m._s = "".join(_COMP[c] for c in reversed(self._s))
        return m

    def get_canonical(self):
        """the numerically smaller of the mer and its reverse complement (A<C<G<T: the same order as the strings')"""
        r = self.get_reverse_complement()
        return r if r._s < self._s else MerDNA(self)

    def reverse_complement(self):
        self._s = self.get_reverse_complement()._s

    def canonicalize(self):
        self._s = self.get_canonical()._s

    def polyA(self): self._s = "A" * len(self._s)
    def polyC(self): self._s = "C" * len(self._s)
    def polyG(self): self._s = "G" * len(self._s)
    def polyT(self): self._s = "T" * len(self._s)

    def is_homopolymer(self):
        return len(set(self._s)) <= 1

    def shift_left(self, c):
        """"ACGT".shift_left('A') -> "CGTA", returns the base that fell off ('A')"""
        out = self._s[0]
        self._s = self._s[1:] + c.upper()
        return out

    def shift_right(self, c):
        out = self._s[-1]
        self._s = c.upper() + self._s[:-1]
        return out

    def __str__(self): return self._s
    def __repr__(self): return "MerDNA(%r)" % self._s
    def __len__(self): return len(self._s)
    def __eq__(self, o): return isinstance(o, MerDNA) and self._s == o._s
    def __lt__(self, o): return self._s < o._s
    def __gt__(self, o): return self._s > o._s
    def __hash__(self): return hash(self._s)


class QueryMerFile:
    """random access to a k-mer database: `qf[mer]` -> count (JF::swig/mer_file.i:12-43)"""

    def __init__(self, path_or_table, device=0):
        if isinstance(path_or_table, KmerTable):          # counts that never were a file: the table `jellyfish count` would have dumped
            self._t = path_or_table
        else:
            path = path_or_table
            try:
                with open(path, "rb") as f:
                    head = f.read(9)
            except OSError:
                raise RuntimeError("Can't open file '%s'" % path) from None
            try:
                self._t = KmerTable.from_jf(path, device=device)
            except _lib.JasperHipError as e:
                if e.code == _lib.JASPER_ERR_FORMAT or not head.isdigit():
                    raise RuntimeError("Unsupported format '%s'" % _header_format(path)) from None
                raise RuntimeError(str(e)) from None
        MerDNA.k(self._t.k)                               # opening a DB sets the process-wide k (mer_file.i:23)

    def __getitem__(self, mer):
        return self._t.lookup([str(mer)])[0]

    def counts(self, mers):
        """the counts of many mers with one kernel launch (not in the SWIG module: `[qf[m] for m in mers]` gives the same list)"""
        return self._t.lookup([str(m) for m in mers])

    @property
    def table(self):
        return self._t


def _header_format(path):
    """the `format` field of a Jellyfish file header (JF::include/jellyfish/generic_file_header.hpp:88-111), '' if there is none"""
    import json
    try:
        with open(path, "rb") as f:
            n = int(f.read(9))
            return str(json.loads(f.read(n).split(b"\0")[0].decode("latin-1")).get("format", ""))
    except Exception:             # noqa: BLE001 -- not a Jellyfish header at all
        return ""
