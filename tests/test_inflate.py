"""CPU: one gzip stream inflated by many threads (jasper_amd/csrc/pgunzip.hpp through the host-only C-ABI entry
jasper_inflate_file) against Python's zlib: the role of `zcat -f` in `zcat -f $READS | jellyfish count /dev/stdin`
(src/jasper.sh:177).  Every compression level, members glued together, many small members (bgzf-like), stored blocks,
binary data, cuts that fall into headers and trailers, and damaged files (which must fail, never return other text)."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from jasper_amd import _lib
    return _lib.lib()


def inflate(L, path, out, threads=4, chunk=1 << 16):
    n = C.c_uint64(0)
    par = C.c_int(0)
    rc = L.jasper_inflate_file(str(path).encode(), threads, chunk, str(out).encode() if out else None, C.byref(n), C.byref(par))
    return rc, n.value, par.value


def fastq_text(seed, nreads, rl=150):
    rng = np.random.default_rng(seed)
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60_000)]
    q = np.frombuffer(b"FFFFFFFF:F,F#", dtype=np.uint8)
    out = []
    for i in range(nreads):
        s = int(rng.integers(0, len(g) - rl))
        out.append(b"@SIM:1:FC:%d:%d 1:N:0:ACGT\n" % (i // 1000, i % 1000) + g[s:s + rl].tobytes() + b"\n+\n" + q[rng.integers(0, len(q), rl)].tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("threads", [2, 5])
def test_levels_and_thread_counts(L, tmp_path, level, threads):
    text = fastq_text(level, 30_000)
    p = tmp_path / "r.fq.gz"
    p.write_bytes(gzip.compress(text, compresslevel=level, mtime=0))
    out = tmp_path / "r.fq"
    rc, n, par = inflate(L, p, out, threads=threads)
    assert rc == 0 and par == 1 and n == len(text)
    assert out.read_bytes() == text


def test_members_glued_together_and_small_members(L, tmp_path):
    a, b, c = fastq_text(1, 8000), fastq_text(2, 5000), b"short tail\n"
    p = tmp_path / "cat.gz"
    p.write_bytes(gzip.compress(a, 6, mtime=0) + gzip.compress(b, 1, mtime=0) + gzip.compress(b"", 6, mtime=0) + gzip.compress(c, 9, mtime=0))
    out = tmp_path / "o"
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and par == 1 and out.read_bytes() == a + b + c
    # bgzf-like: thousands of members of a few KB (every cut lands near a member start: the window is known there)
    text = fastq_text(3, 20_000)
    parts = [text[i:i + 40_000] for i in range(0, len(text), 40_000)]
    p.write_bytes(b"".join(gzip.compress(x, 6, mtime=0) for x in parts))
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and par == 1 and out.read_bytes() == text
    # a member header with a file name and a comment (FNAME, FCOMMENT) and an extra field
    hdr = b"\x1f\x8b\x08\x1c\0\0\0\0\0\x03" + b"\x04\x00ABCD" + b"reads.fq\0" + b"a comment\0"
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(a) + co.flush()
    member = hdr + body + (zlib.crc32(a) & 0xFFFFFFFF).to_bytes(4, "little") + (len(a) & 0xFFFFFFFF).to_bytes(4, "little")
    p.write_bytes(member + gzip.compress(b, 6, mtime=0))
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and par == 1 and out.read_bytes() == a + b


def test_stored_blocks_and_binary_data(L, tmp_path):
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes()          # incompressible: stored blocks
    text = fastq_text(4, 6000)
    mixed = text[:300_000] + noise + text[300_000:] + bytes(range(256)) * 2000 + noise[:100_000] + text
    p = tmp_path / "m.gz"
    out = tmp_path / "o"
    for level in (1, 6):
        p.write_bytes(gzip.compress(mixed, level, mtime=0))
        rc, n, par = inflate(L, p, out, threads=3)
        assert rc == 0 and par == 1 and out.read_bytes() == mixed
    # level 0: nothing but stored blocks (no dynamic block to cut at: the whole file is one chunk chain)
    p.write_bytes(gzip.compress(mixed, 0, mtime=0))
    rc, n, par = inflate(L, p, out, threads=3)
    assert rc == 0 and out.read_bytes() == mixed


def test_small_and_odd_inputs_go_to_zlib(L, tmp_path):
    p = tmp_path / "s.gz"
    out = tmp_path / "o"
    p.write_bytes(gzip.compress(b"@r\nACGT\n+\nIIII\n", mtime=0))
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and par == 0 and out.read_bytes() == b"@r\nACGT\n+\nIIII\n"
    p.write_bytes(b"")                                                        # empty file: zlib's reader returns nothing
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and n == 0
    text = fastq_text(5, 4000)
    p.write_bytes(text)                                                       # plain text: `zcat -f` passes it through
    rc, n, par = inflate(L, p, out)
    assert rc == 0 and par == 0 and out.read_bytes() == text
    rc, n, par = inflate(L, tmp_path / "missing.gz", out)
    assert rc != 0


def test_damaged_files_fail(L, tmp_path):
    """a truncated file, a flipped bit in the middle, a wrong CRC, a wrong length: an error -- never other text"""
    text = fastq_text(6, 25_000)
    z = gzip.compress(text, 6, mtime=0)
    p = tmp_path / "d.gz"
    out = tmp_path / "o"
    for cut in (len(z) // 2, len(z) - 4, len(z) - 9):
        p.write_bytes(z[:cut])
        rc, n, par = inflate(L, p, out)
        assert par == 1 and rc != 0
    crc_bad = bytearray(z)
    crc_bad[-6] ^= 0x10
    p.write_bytes(bytes(crc_bad))
    assert inflate(L, p, out)[0] != 0
    len_bad = bytearray(z)
    len_bad[-2] ^= 0x01
    p.write_bytes(bytes(len_bad))
    assert inflate(L, p, out)[0] != 0
    rng = np.random.default_rng(8)
    bad = 0
    for trial in range(12):
        flipped = bytearray(z)
        pos = int(rng.integers(len(z) // 10, len(z) - 100))
        flipped[pos] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(flipped))
        rc, n, par = inflate(L, p, out)
        if rc == 0:
            assert out.read_bytes() == text       # (cannot happen: the CRC covers every byte; kept as the statement of intent)
        else:
            bad += 1
    assert bad == 12


def test_many_chunk_sizes_against_zlib(L, tmp_path):
    """cuts of many sizes, so that boundaries fall everywhere: inside stored blocks, next to member ends, near the end"""
    rng = np.random.default_rng(9)
    text = fastq_text(10, 12_000)
    z = gzip.compress(text[:1_500_000], 6, mtime=0) + gzip.compress(text[1_500_000:], 2, mtime=0)
    p = tmp_path / "c.gz"
    out = tmp_path / "o"
    p.write_bytes(z)
    for chunk in (1 << 16, 70_001, 131_072, 250_000, len(z) // 5):
        for threads in (2, 3, 7):
            rc, n, par = inflate(L, p, out, threads=threads, chunk=chunk)
            assert rc == 0 and out.read_bytes() == text, (chunk, threads)
