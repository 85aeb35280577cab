#!/usr/bin/env python3
"""Validate .jf files written by jasper_table_write_jf with the REAL jellyfish 2.3.0 (build container only).

Step 1 (GPU box):   python tests/golden/check_jf_writer.py write gpurun_out/jfw      -> <case>.jf for the golden cases
Step 2 (here):      python tests/golden/check_jf_writer.py verify gpurun_out/jfw     -> `jellyfish dump -c`, `histo` and
                    `query` of those files == the golden dump / histogram the reference produced for the same reads.
"""
import gzip, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = ["simple_k25", "simple_k37"]
JF = os.environ.get("JELLYFISH", "/tmp/jf_install/bin/jellyfish")


def write(outdir):
    from golden_util import Case
    from jasper_amd import KmerTable
    os.makedirs(outdir, exist_ok=True)
    for name in CASES:
        c = Case(name)
        t = KmerTable(c.k, min_slots=1 << 16)
        t.count_text(c.reads_text())
        t.write_jf(os.path.join(outdir, name + ".jf"), ["count", "-C", "-m", str(c.k), "(libjasper_hip)"])
        t.close()
        print("wrote", name)


def verify(outdir):
    from golden_util import Case
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(os.path.dirname(os.path.dirname(JF)), "lib"))
    for name in CASES:
        c = Case(name)
        p = os.path.join(outdir, name + ".jf")
        dump = subprocess.run([JF, "dump", "-c", p], env=env, capture_output=True, text=True, check=True).stdout
        got = dict((a, int(b)) for a, b in (l.split() for l in dump.splitlines()))
        want = c.dump()
        assert got == want, (name, len(got), len(want))
        histo = subprocess.run([JF, "histo", p], env=env, capture_output=True, text=True, check=True).stdout
        assert histo == open(os.path.join(c.dir, "histo.csv")).read(), name
        kmers = sorted(want)[::37][:300] + ["A" * c.k]
        q = subprocess.run([JF, "query", p] + kmers, env=env, capture_output=True, text=True, check=True).stdout
        for line, km in zip(q.splitlines(), kmers):
            a, b = line.split()
            assert int(b) == want.get(km, 0), (name, km, b)
        print(name, "ok: dump (%d k-mers), histo and %d queries agree with the reference" % (len(got), len(kmers)))
        # the reference's own polisher (unmodified src/jasper.py through its SWIG QueryMerFile) on OUR file
        import json, shutil, tempfile
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import make_golden as mg
        meta = json.load(open(os.path.join(c.dir, "meta.json")))
        work = tempfile.mkdtemp(prefix="jfw_")
        shutil.copy(p, os.path.join(work, "db.jf"))
        shutil.copy(os.path.join(c.dir, "batch.fa"), os.path.join(work, "batch.fa"))
        args = dict(query="batch.fa", k=meta["k"], fout="batch.fa.fix.csv", ff="batch.fa.fixed.fa.tmp", db="db.jf", thre=meta["thre"],
                    passes=meta["passes"], debug=False)
        drv = os.path.join(work, "drv.py")
        open(drv, "w").write(mg.DRIVER % dict(jfpy=mg.JF_PY, ref=mg.REF))
        r = subprocess.run([sys.executable, drv, json.dumps(args)], cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        P = meta["passes"]
        assert open(os.path.join(work, "_iter%d_batch.fa.fixed.fa.tmp" % (P - 1))).read() == open(os.path.join(c.dir, "fixed.fa")).read()
        for it in range(P):
            assert open(os.path.join(work, "_iter%d_batch.fa.fix.csv" % it), "rb").read() == open(os.path.join(c.dir, "iter%d.fix.csv" % it), "rb").read()
        assert open(os.path.join(work, "0qValCalcHelper.csv")).read() == meta["qv0"]
        assert open(os.path.join(work, "%dqValCalcHelper.csv" % P)).read() == meta["qvP"]
        shutil.rmtree(work)
        print(name, "ok: unmodified src/jasper.py run on our .jf reproduces the golden fixed FASTA, fix CSVs and QV counters")
        # `jellyfish merge` (JF::jellyfish/merge_files.cc:96-150) takes files of one size and one hash matrix: two files written by this
        # library (identity matrix in the header, same size) merge, and the merged counts are the sums
        merged = os.path.join(outdir, name + ".merged.jf")
        subprocess.run([JF, "merge", "-o", merged, p, p], env=env, check=True)
        dump2 = subprocess.run([JF, "dump", "-c", merged], env=env, capture_output=True, text=True, check=True).stdout
        assert dict((a, int(b)) for a, b in (l.split() for l in dump2.splitlines())) == {k: 2 * v for k, v in want.items()}, name
        os.remove(merged)
        print(name, "ok: jellyfish merge of two of our files gives twice the counts")


if __name__ == "__main__":
    {"write": write, "verify": verify}[sys.argv[1]](sys.argv[2])
