"""GPU: the drop-in at BASELINE configs[1] size against the REAL reference.

tests/golden/fullsize_cfg2.json holds what the unmodified reference (bash src/jasper.sh, Jellyfish 2.3.0, jasper.py; 107 s on
the 8 vCPU of the build container) produced for the 47 Mb / 30x / k=37 / 2-pass synthetic input of
jasper_amd.synth.write_cli_inputs(seed 2): digests of the polished FASTA (records sorted by name), of fixes.csv, of the
histogram file, and the threshold.  The same input is regenerated here (deterministic numpy), `python -m jasper_amd.cli`
runs with the same flags, and every digest must be identical."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the 140 Mb case (11 GB of FASTQ) runs only on request: JASPER_TEST_BIG=1
@pytest.mark.parametrize("name", ["fullsize_cfg1", "fullsize_cfg2"] + (["fullsize_cfg3like"] if os.environ.get("JASPER_TEST_BIG") else []))
def test_cli_fullsize_matches_real_reference(hip, tmp_path, name):
    """cfg1 = BASELINE configs[0] (4.6 Mb, k=25, 1 pass: the reference's own CPU-runnable case), cfg2 = configs[1]"""
    from jasper_amd import synth
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    d = str(tmp_path)
    nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"])
    assert nreads == ref["reads"] and asm_len == ref["assembly_bases"]
    env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_NO_JF="1")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]),
                        "-p", str(ref["passes"])], cwd=d, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert p.returncode == 0, p.stdout + p.stderr
    got = synth.output_digests(d, k=ref["k"])
    for key in ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256"):
        assert got[key] == ref[key], (key, got[key], ref[key])
    # same log lines (dates and Q digits aside: bc is missing in the build container, so the reference printed "Inf")
    import re
    strip = lambda ls: [re.sub(r"Q value = .*", "Q value =", re.sub(r"^\[[^\]]*\]", "[DATE]", ln)) for ln in ls]
    assert strip(p.stdout.splitlines()) == strip(ref["stdout"])
    print("drop-in wall %.1f s vs reference %.1f s on %s" % (wall, ref["reference_wall_seconds"], ref["host"]))


def test_cli_fullsize_two_ranks_matches_real_reference(hip, tmp_path):
    """configs[1] through the multi-GPU form of the driver (2 ranks rehearsed on the one GPU, gloo transport): the 2.9 GB
    FASTQ is cut into two byte ranges at record boundaries, the counts are summed by key owner, each rank polishes half of
    the batch files through IPC-mapped owner tables -- same digests as the real reference's run"""
    import re
    import socket
    from jasper_amd import synth
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_cfg2.json")))
    d = str(tmp_path)
    nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"])
    assert nreads == ref["reads"] and asm_len == ref["assembly_bases"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_DIST_BACKEND="gloo", JASPER_AMD_ONE_GPU="1", JASPER_AMD_NO_JF="1")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", str(ref["k"]),
                        "-t", str(ref["threads"]), "-p", str(ref["passes"])], cwd=d, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert p.returncode == 0, p.stdout + p.stderr
    got = synth.output_digests(d, k=ref["k"])
    for key in ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256"):
        assert got[key] == ref[key], (key, got[key], ref[key])
    strip = lambda ls: [re.sub(r"Q value = .*", "Q value =", re.sub(r"^\[[^\]]*\]", "[DATE]", ln)) for ln in ls]
    mine = [ln for ln in p.stdout.splitlines() if re.match(r"^\[\w{3} \w{3} +\d", ln)]      # (gloo prints its own, sometimes interleaved, lines)
    assert strip(mine) == strip(ref["stdout"])
    print("2-rank drop-in wall %.1f s (both ranks on one GPU, gloo) vs reference %.1f s" % (wall, ref["reference_wall_seconds"]))
