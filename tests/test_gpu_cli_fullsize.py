"""GPU: the drop-in at BASELINE configs[0..4] shape against the REAL reference.

tests/golden/fullsize_*.json hold what the unmodified reference (bash src/jasper.sh, Jellyfish 2.3.0, jasper.py, on the 8
vCPU of the build container; tests/golden/ref_fullsize.py) produced for the deterministic synthetic inputs of
jasper_amd.synth.write_cli_inputs: digests of the polished FASTA (records sorted by name), of fixes.csv, of the histogram
file, the threshold and the log lines.  The same input is regenerated here (numpy), `python -m jasper_amd.cli` runs with
the same flags -- as one process, or as two ranks (the multi-GPU form of the driver) -- and every digest must be identical.

  fullsize_cfg1          configs[0]: 4.6 Mb, 30x, k=25, 1 pass, -t 8
  fullsize_cfg2          configs[1]: 47 Mb, 30x, k=37, 2 passes, -t 8      (also as two ranks)
  fullsize_cfg2_t16      configs[1] chunked as -t 16 (the chunking bench.py uses)
  fullsize_cfg2_k41/_k51 the same files polished with -k 41 / -k 51 (counting through the 16-byte-record passes; a wide table at 51)
  fullsize_cfg3_quarter  configs[2] shape at 1/4 scale: 35 Mb in 7 contigs, 40x, two ranks (read shards + table merge)
  fullsize_cfg4_scaled   configs[3] shape at 1/64 scale: 48.4 Mb in 24 contigs of 0.7-3.4 Mb, 30x, -t 64 -> BATCH_SIZE below the
                         contig sizes: several chunk records per contig, ~70 batch files (src/jasper.sh:132-139,155-156)
  fullsize_cfg5_scaled   configs[4] shape, scaled: 4 Mb in 3 contigs, 10 read sets (individuals with 0.1 % private SNPs) x 30x,
                         4 passes (the rolling-threshold path sees 300x counts, src/jasper.py:80-93)
  fullsize_cfg5_lowcov   the same shape with 10 x 10x on 10 Mb: the private-SNP k-mers put the histogram's first local minimum
                         below 4 and the REFERENCE aborts ("Local min of kmer counts is smaller than 4", src/jasper.sh:200-202) --
                         the drop-in must abort the same way, with the same log lines
  fullsize_cfg3          configs[2] EXACTLY as stated: 140 Mb in 7 contigs, 40x = 37.3 M reads (11.5 GB of FASTQ), two ranks
  fullsize_cfg4_share    one rank's share of configs[3] (CHM13 on 8 GPUs): 390 Mb in 3 contigs, 30x = 78 M reads (24 GB of FASTQ),
                         -t 64: a 2^32-slot table filled in 8 pieces, 36 batch files (reference: 884 s)
  fullsize_cfg5_chr21    configs[4]'s shape at chr21 size: 47 Mb, 10 read sets (individuals with 0.1 % private SNPs) x 30x = 94 M reads
                         (29 GB of FASTQ, 10.7 G k-mer occurrences, counts in the hundreds), 4 passes, -t 16 (reference: 850 s):
                         the rolling threshold (src/jasper.py:80-93) at 300x
  on request (JASPER_TEST_BIG=1): fullsize_cfg3 as one process, fullsize_cfg3like (140 Mb, one contig, 30x)"""
import json
import os
import re
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIG = bool(os.environ.get("JASPER_TEST_BIG"))
_inputs = {}


def _ref(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))


def _input_dir(ref, tmp_path_factory):
    """the inputs of one shape are generated once per session and linked into every run directory"""
    from jasper_amd import synth
    key = (ref["genome_mb"], ref["seed"], ref.get("coverage", 30), ref.get("contigs", 1), ref.get("populations", 1))
    if key not in _inputs:
        d = str(tmp_path_factory.mktemp("inputs"))
        # (the read files of the large cases are 11-29 GB: a scratch disk without room for them is a reason to skip, not to fail)
        import shutil
        need = int(ref["reads"] * 307 * 1.25) + (4 << 30)
        free = shutil.disk_usage(d).free
        if free < need:
            pytest.skip("%s has %.0f GB free, the inputs of this case need %.0f GB" % (d, free / 1e9, need / 1e9))
        nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"], coverage=key[2], contigs=key[3], populations=key[4])
        assert nreads == ref["reads"] and asm_len == ref["assembly_bases"]
        _inputs[key] = d
    return _inputs[key]


def _input_key(ref):
    return (ref["genome_mb"], ref["seed"], ref.get("coverage", 30), ref.get("contigs", 1), ref.get("populations", 1))


_uses_left = {}          # input key -> runs that still need it (the large inputs -- tens of GB -- go as soon as their last run is over)


def _done_with(ref):
    import shutil
    key = _input_key(ref)
    if key in _uses_left:
        _uses_left[key] -= 1
        if _uses_left[key] <= 0 and key in _inputs:
            shutil.rmtree(_inputs.pop(key), ignore_errors=True)


def _run(ref, tmp_path_factory, ranks, count=None, extra_env=None, expect_stderr=None):
    try:
        _run_case(ref, tmp_path_factory, ranks, count, extra_env, expect_stderr)
    finally:
        _done_with(ref)


def _run_case(ref, tmp_path_factory, ranks, count=None, extra_env=None, expect_stderr=None):
    from jasper_amd import synth
    src = _input_dir(ref, tmp_path_factory)
    d = str(tmp_path_factory.mktemp("run"))
    files = synth.read_files(ref.get("populations", 1))
    for fn in files + ["asm.fa"]:
        os.symlink(os.path.join(src, fn), os.path.join(d, fn))
    args = ["-r", " ".join(files), "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]), "-p", str(ref["passes"])]
    env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_NO_JF="1", JASPER_AMD_TIMING="1")
    if ranks == 1:
        cmd = [sys.executable, "-m", "jasper_amd.cli"] + args
    else:
        # the multi-GPU form of the driver, rehearsed on the one GPU of the test box with gloo as the transport (with two
        # GPUs, tests/test_gpu_nccl.py runs the same driver over RCCL)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env.update(JASPER_AMD_DIST_BACKEND="gloo", JASPER_AMD_ONE_GPU="1", JASPER_AMD_TIMING="1")
        env.update(extra_env or {})
        if count:             # how the ranks' counts reach the key owners (default: cli picks by the bytes-per-link model, `local` at two ranks)
            env["JASPER_AMD_COUNT"] = count
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "jasper_amd.cli"] + args
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    strip = lambda ls: [re.sub(r"Q value = .*", "Q value =", re.sub(r"^\[[^\]]*\]", "[DATE]", ln)) for ln in ls]
    mine = [ln for ln in p.stdout.splitlines() if re.match(r"^\[\w{3} \w{3} +\d", ln)]      # (gloo prints its own, sometimes interleaved, lines)
    if ref["exit"] != 0:
        # the reference gave up on this input (src/jasper.sh:35-39: message on stderr, exit 1): so must the drop-in, at the same point
        assert p.returncode != 0, p.stdout + p.stderr
        assert strip(mine) == strip(ref["stdout"])
        assert "Local min of kmer counts is smaller than 4" in p.stderr
        assert not os.path.exists(os.path.join(d, "asm.fa.polished.fasta"))
        return
    assert p.returncode == 0, p.stdout + p.stderr
    if expect_stderr:
        assert expect_stderr in p.stderr, p.stderr
    elif ranks > 1:
        assert ("region lists -> owners' shards" in p.stderr) == (count == "exchange"), p.stderr
    got = synth.output_digests(d, k=ref["k"])
    for key in ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256"):
        assert got[key] == ref[key], (key, got[key], ref[key])
    # same log lines (dates and Q digits aside: bc is missing in the build container, so the reference printed "Inf")
    assert strip(mine) == strip(ref["stdout"])
    # The QV INPUTS are pinned by the reference: the column sums of its own {0,P}qValCalcHelper.csv (src/jasper.py:107-111), which
    # tests/golden/ref_fullsize.py copies before src/jasper.sh:258 removes them.  The Q strings the drop-in prints must be
    # qv.py (the bc restatement) of exactly those sums.
    m = re.search(r"^\[qv\] before (\d+) (\d+) after (\d+) (\d+)$", p.stderr, re.M)
    assert m, p.stderr
    assert [int(m.group(1)), int(m.group(2))] == ref["qv_before"], (m.groups(), ref["qv_before"])
    assert [int(m.group(3)), int(m.group(4))] == ref["qv_after"], (m.groups(), ref["qv_after"])
    from jasper_amd import qv
    qlines = [re.sub(r"^\[[^\]]*\] ", "", ln) for ln in mine if "Q value" in ln]
    assert qlines == ["Before Polishing: Q value = %s" % qv.q_value(ref["qv_before"][0], ref["qv_before"][1], ref["k"]),
                      "After Polishing: Q value = %s" % qv.q_value(ref["qv_after"][0], ref["qv_after"][1], ref["k"])], qlines
    print("%d-rank drop-in wall %.1f s%s vs reference %.1f s on %s" % (ranks, wall, " (counts by exchange of region lists)" if count == "exchange" else "",
                                                                     ref["reference_wall_seconds"], ref["host"]))
    for fn in os.listdir(d):          # (GBs of intermediates per case)
        if not os.path.islink(os.path.join(d, fn)):
            os.remove(os.path.join(d, fn))


CASES = [("fullsize_cfg1", 1), ("fullsize_cfg2", 1), ("fullsize_cfg2_t16", 1), ("fullsize_cfg2", 2),
         ("fullsize_cfg3_quarter", 2), ("fullsize_cfg3_quarter", 1), ("fullsize_cfg4_scaled", 1), ("fullsize_cfg4_scaled", 2),
         ("fullsize_cfg5_scaled", 1), ("fullsize_cfg5_scaled", 2), ("fullsize_cfg5_lowcov", 1),
         ("fullsize_cfg2_k41", 1), ("fullsize_cfg2_k51", 1),      # configs[1] files at k = 41 / 51: keys of 82 / 102 bits -> 16-byte records, a wide table at k = 51 (src/jasper.sh:89-90 takes any -k)
         ("fullsize_cfg3", 2),          # configs[2] exactly as stated (11.5 GB of FASTQ, as two ranks on one GPU)
         ("fullsize_cfg4_share", 1),    # ONE rank's share of configs[3]: 390 Mb in 3 contigs + 30x (78 M reads, 24 GB of FASTQ), -t 64: a 2^32-slot table,
                                        # multi-piece counting, 36 batch files (also tools/run_big_case.py, which prints while it works)
         ("fullsize_cfg5_chr21", 1)]    # configs[4]'s shape at chr21 size: 10 x 30x on 47 Mb (94 M reads, 29 GB of FASTQ), 4 passes
if BIG:
    CASES += [("fullsize_cfg3", 1), ("fullsize_cfg3like", 1)]
EXCHANGE_CASES = [("fullsize_cfg2", 2), ("fullsize_cfg3_quarter", 2), ("fullsize_cfg5_scaled", 2), ("fullsize_cfg3", 2)]
for _name, _ranks in CASES + EXCHANGE_CASES + [("fullsize_cfg1", 2)]:          # (the last: test_cli_exchange_falls_back_...)
    _k = _input_key(_ref(_name))
    _uses_left[_k] = _uses_left.get(_k, 0) + 1


@pytest.mark.parametrize("name,ranks", CASES)
def test_cli_fullsize_matches_real_reference(hip, tmp_path_factory, name, ranks):
    """(several ranks: counts reach the key owners as the entries of per-GPU tables here, as region lists in the test below)"""
    _run(_ref(name), tmp_path_factory, ranks, count=("local" if ranks > 1 else None))


@pytest.mark.parametrize("name,ranks", EXCHANGE_CASES)
def test_cli_fullsize_counts_by_exchange_of_region_lists(hip, tmp_path_factory, name, ranks):
    """the same digests with no table per GPU: file reader -> batches of bases -> region lists grouped by key owner -> one
    all_to_all per batch -> owners' shards (dist.count_sharded; what `auto` picks whenever the table has a geometry for it)"""
    _run(_ref(name), tmp_path_factory, ranks, count="exchange")


def test_cli_exchange_falls_back_when_the_shards_are_far_too_small(hip, tmp_path_factory):
    """owners' shards of 2^16 slots for 6 M keys each (the reference's hash grows on demand, JF::jellyfish/mer_counter.cc; a size
    hint that low would do this): the lists overflow many-fold, every rank hears of it, and the run starts over with a table
    per GPU -- same digests"""
    _run(_ref("fullsize_cfg1"), tmp_path_factory, 2, count="exchange", extra_env={"JASPER_AMD_SHARD_SLOTS": "65536"},
         expect_stderr="counting into a table per GPU instead")
