"""GPU: one large single-GPU run at the geometry of BASELINE configs[3] (CHM13 on 8 GPUs: shards of 2^32 slots, rounds of 2^31
bases) -- the sizes at which 32-bit arithmetic on slot numbers, list offsets or byte positions would first go wrong.

  * a table of 2^32 slots (69 GB; slot indices need 33 bits, 2^20 regions, 1024 second-level lists per bucket)
  * three pieces of 2^31 bases each (14.2 M reads; every piece one launch of part1 / part2 / region_insert: the first into the
    lazily cleared table with 12-byte LDS slots, the others into the filled table with its images loaded from HBM)
  * checked by what does not depend on size: the reads are sampled from a random genome with KNOWN start positions, so the exact
    count of the k-mer at every genome position is a difference of prefix sums of the start histogram -- 200 000 of them are
    looked up -- plus occurrences, distinct keys and the histogram's two sums; then polishing through the large table.
The 8-rank exchange protocol at this shard size cannot be played on one 288 GB GPU (eight shards are 550 GB); its owner-side
extra split pass is forced at small size in test_gpu_exchange.py.  Role of the reference: JF::sub_commands/count_main.cc:152-184
at `-s 4G`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reads_piece(torch, gen, genome, nreads, L, starts_hist, err):
    """nreads reads of length L (+ 'N') from random starts, random strand; adds the starts to starts_hist (error-free pieces only)"""
    dev = genome.device
    n = genome.numel()
    out = torch.empty((nreads, L + 1), dtype=torch.uint8, device=dev)
    out[:, L] = ord("N")
    ar = torch.arange(L, device=dev)
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    code = torch.zeros(256, dtype=torch.int64, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for i, (a, b) in enumerate(zip(b"ACGT", b"TGCA")):
        comp[a] = b
        code[a] = i
    block = 1 << 20
    for a in range(0, nreads, block):
        m = min(block, nreads - a)
        starts = torch.randint(0, n - L + 1, (m,), generator=gen, device=dev)
        r = genome[starts[:, None] + ar[None, :]]
        if err > 0:
            e = torch.rand((m, L), generator=gen, device=dev) < err
            shift = torch.randint(1, 4, (m, L), generator=gen, device=dev)
            r = torch.where(e, lut[(code[r.long()] + shift) % 4], r)
        else:
            starts_hist += torch.bincount(starts, minlength=n)
        flip = torch.rand(m, generator=gen, device=dev) < 0.5
        r = torch.where(flip[:, None], comp[r.flip(1).long()], r)
        out[a:a + m, :L] = r
    return out.reshape(-1)


def test_table_of_2_32_slots_and_pieces_of_2_31_bases(hip):
    import torch
    sys.path.insert(0, ROOT)
    from jasper_amd import KmerTable, synth
    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info(dev)
    if free < (200 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    k, L, G = 37, 150, 64_000_000
    gen = torch.Generator(device=dev).manual_seed(20261004)
    genome = synth.torch_genome(gen, G, dev, repeat_frac=0)
    starts_hist = torch.zeros(G, dtype=torch.int64, device=dev)
    nreads = (1 << 31) // (L + 1)
    t = KmerTable(k, min_slots=1 << 32, device=0)
    assert t.info()["slots"] == 1 << 32
    occ = 0
    for piece in range(3):
        reads = _reads_piece(torch, gen, genome, nreads, L, starts_hist, err=(0.0 if piece < 2 else 0.003))
        assert reads.numel() > (1 << 31) - 200
        torch.cuda.synchronize()
        t.count_bases_device(reads.data_ptr(), reads.numel())
        assert t.count_stages()[1] >= 1 and t.count_path() == 1, "the partitioned path was not taken"
        occ += nreads * (L - k + 1)
        info = t.info()
        assert info["occurrences"] == occ and info["slots"] == 1 << 32
        if piece == 1:
            # exact counts after the two error-free pieces: k-mer at genome position p <- reads starting in [p - (L - k), p]
            cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(starts_hist, 0)])
            pos = torch.randint(0, G - k + 1, (200_000,), generator=gen, device=dev)
            want = (cs[pos + 1] - cs[torch.clamp(pos - (L - k), min=0)]).cpu().tolist()
            g = genome.cpu().numpy().tobytes()
            got = t.lookup([g[p:p + k].decode() for p in pos.cpu().tolist()])
            assert got == want
            covered = int(((cs[torch.arange(G - k + 1, device=dev) + 1] - cs[torch.clamp(torch.arange(G - k + 1, device=dev) - (L - k), min=0)]) > 0).sum().item())
            assert info["distinct"] == covered                     # (a random 64 Mb genome repeats no 37-mer)
            h = t.histogram()
            assert sum(h) == info["distinct"] and sum(m * c for m, c in enumerate(h)) == occ
            del cs, pos
        del reads
    info = t.info()
    h = t.histogram()
    assert sum(h) == info["distinct"] and info["distinct"] > covered + 100_000_000      # the third piece's read errors: ~1.6e8 new keys
    assert sum(m * c for m, c in enumerate(h)) == occ                                    # (no count reaches the last bin here)
    # polishing through the large table: a draft with planted errors comes back (almost) clean, QV counters consistent
    rng = np.random.default_rng(5)
    asm = synth.make_assembly(rng, np.frombuffer(g[:3_000_000], dtype=np.uint8), err=2e-4).tobytes().decode()
    seqs = [asm[a:b] for _, a, b in synth.chunk_records("c", len(asm), 600_000)]
    res = t.polish_batch(seqs, 2, 2)
    assert res.qv[1] == sum(len(s) - k + 1 for s in seqs) and res.qv[2] < res.qv[0] / 20
    t.close()
