#!/usr/bin/env python3
"""Build container only: the inputs of Jellyfish's OWN counting tests, made by the reference's own generator.

JF::tests/generate_sequence.sh:6-7 runs `generate_sequence` (built from the vendored tarball by SURVEY.md Appendix C's recipe;
the binary is `jellyfish-2.3.0/bin/generate_sequence` of the build tree) with fixed seeds:

    generate_sequence -v -o seq10m -m 10 -m 22 -s 3141592653 10000000                       -> seq10m.fa  (2 records, 70 columns)
    generate_sequence -v -o seq1m -s 1040104553 1000000 1000000 1000000 1000000 1000000     -> seq1m_{0..4}.fa
    gzip -c seq1m_$i.fa > seq1m_$i.fa.gz                                                    (:9-11)

and JF::tests/parallel_hashing.sh:6-20, JF::tests/multi_file.sh:6-9 hold the md5 of `jellyfish histo` after
`jellyfish count -C -m 15` on them -- the only golden vectors the reference itself HOLDS on the counting path:

    864c0b0826854bdc72a85d170549b64b   seq10m.fa (also with DOS line ends: unix2dos -n seq10m.fa seq10mDOS.fa)
    d93b7678037814c256d1d9120a0e6422   seq1m_0 seq1m_1 seq1m_2 seq10m seq1m_3 seq1m_4 (plain, and seq10m.fa + the five .gz)

This script copies the generator's outputs (data, not source) into tests/golden/jf_tests/ -- seq10m.fa gzipped here, the five
seq1m_*.fa.gz as the reference's script made them -- and records the md5 of every plain file in jf_tests/inputs.md5, so that a
test can tell a damaged fixture from a wrong histogram.  tests/test_jf_held_vectors.py uses them (oracle on the CPU, the HIP
path on the GPU).

usage: python3 tests/golden/make_jf_test_vectors.py [/tmp/jf_build/jellyfish-2.3.0]     (after `make check` or tests/generate_sequence.sh there)"""
import gzip, hashlib, os, shutil, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
build = sys.argv[1] if len(sys.argv) > 1 else "/tmp/jf_build/jellyfish-2.3.0"
data = os.path.join(build, "tests-data")
if not os.path.exists(os.path.join(data, "seq10m.fa")):
    os.makedirs(data, exist_ok=True)
    gen = os.path.join(build, "bin", "generate_sequence")
    subprocess.check_call([gen, "-v", "-o", "seq10m", "-m", "10", "-m", "22", "-s", "3141592653", "10000000"], cwd=data)
    subprocess.check_call([gen, "-v", "-o", "seq1m", "-s", "1040104553"] + ["1000000"] * 5, cwd=data)
    for i in range(5):
        with open(os.path.join(data, "seq1m_%d.fa" % i), "rb") as f, open(os.path.join(data, "seq1m_%d.fa.gz" % i), "wb") as g:
            subprocess.check_call(["gzip", "-c"], stdin=f, stdout=g)
out = os.path.join(HERE, "jf_tests")
os.makedirs(out, exist_ok=True)
sums = []
with open(os.path.join(data, "seq10m.fa"), "rb") as f:
    raw = f.read()
sums.append((hashlib.md5(raw).hexdigest(), "seq10m.fa"))
with open(os.path.join(out, "seq10m.fa.gz"), "wb") as g:
    with gzip.GzipFile(filename="", mode="wb", fileobj=g, compresslevel=9, mtime=0) as z:
        z.write(raw)
for i in range(5):
    shutil.copy(os.path.join(data, "seq1m_%d.fa.gz" % i), out)
    sums.append((hashlib.md5(gzip.open(os.path.join(out, "seq1m_%d.fa.gz" % i), "rb").read()).hexdigest(), "seq1m_%d.fa" % i))
with open(os.path.join(out, "inputs.md5"), "w") as f:
    for s, n in sums:
        f.write("%s  %s\n" % (s, n))
print(open(os.path.join(out, "inputs.md5")).read())
