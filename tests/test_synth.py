"""CPU: the pipelined FASTQ writer of jasper_amd.synth produces the bytes -- and leaves the generator in the state -- of the plain
composition it replaces (make_reads_stream + _write_fastq): the full-size fixtures under tests/golden/ were made by the real
reference from exactly those bytes."""
import hashlib

import numpy as np

from jasper_amd import synth


def test_pipelined_fastq_writer_equals_the_plain_composition(tmp_path):
    seed, read_len, cov, err = 11, 150, 30, 0.003
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(rng, 2_300_000)
    cuts = [0, 1_500_000, 1_500_400, 2_300_000]          # three "contigs": several blocks of 200 000 reads, a tiny one, one partial block
    parts = [genome[a:b] for a, b in zip(cuts, cuts[1:])]
    state = rng.bit_generator.state
    plain = [synth.make_reads_stream(rng, g, cov, read_len, err).reshape(-1, read_len + 1)[:, :read_len] for g in parts]
    n_plain = synth._write_fastq(str(tmp_path / "plain.fq"), np.concatenate(plain), read_len)
    after_plain = rng.integers(0, 1 << 62)
    rng.bit_generator.state = state
    n_piped = synth._write_reads_fastq(str(tmp_path / "piped.fq"), rng, parts, cov, read_len, err, workers=3)
    assert n_piped == n_plain > 400_000
    assert rng.integers(0, 1 << 62) == after_plain
    md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
    assert md5(tmp_path / "piped.fq") == md5(tmp_path / "plain.fq")
