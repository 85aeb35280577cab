"""helpers to load the golden cases (tests/golden/cases/*) produced by tests/golden/make_golden.py"""
import glob
import gzip
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases")


def case_names():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*")) if os.path.isdir(p))


class Case:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        self.k = self.meta["k"]
        self.passes = self.meta["passes"]
        self.thre = self.meta["thre"]

    def reads_text(self):
        p = glob.glob(os.path.join(self.dir, "reads.*.gz"))[0]
        return gzip.open(p, "rb").read()

    def dump(self):
        """dict canonical k-mer -> count as printed by `jellyfish dump -c`"""
        d = {}
        for ln in gzip.open(os.path.join(self.dir, "dump.txt.gz"), "rt"):
            a, b = ln.split()
            d[a] = int(b)
        return d

    def histo_rows(self):
        return [tuple(int(x) for x in ln.split()) for ln in open(os.path.join(self.dir, "histo.csv")) if ln.strip()]

    def batch(self):
        """ordered (name, seq) as src/jasper.py:615-631 parse_fasta reads the batch file"""
        names, seqs = [], []
        for ln in open(os.path.join(self.dir, "batch.fa")):
            if ln.startswith(">"):
                names.append(ln.split()[0][1:])
                seqs.append("")
            else:
                seqs[-1] += ln.replace("\n", "")
        return names, seqs

    def fix_csv(self, it):
        return open(os.path.join(self.dir, "iter%d.fix.csv" % it), "rb").read().decode()

    def fixed_fa(self):
        return open(os.path.join(self.dir, "fixed.fa")).read()

    def qv(self):
        a = [int(x) for x in self.meta["qv0"].split()]
        b = [int(x) for x in self.meta["qvP"].split()]
        return (a[0], a[1], b[0], b[1])


def fasta60(names, seqs):
    """src/jasper.py:120-128,142-147"""
    out = []
    for n, s in zip(names, seqs):
        out.append(">%s\n" % n)
        for i in range(0, len(s), 60):
            out.append(s[i:i + 60] + "\n")
    return "".join(out)


CSV_HEADER = "Contig Base_coord Original Mutation\r\n"
