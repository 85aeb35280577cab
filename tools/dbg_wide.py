import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from jasper_amd import KmerTable, synth
k = int(sys.argv[1]); G = int(sys.argv[2]); ms = int(sys.argv[3])
rng = np.random.default_rng(7)
genome = synth.make_genome(rng, G)
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003).tobytes()
t = KmerTable(k, min_slots=1 << ms)
print(t.info())
try:
    t.count_bases(reads)
    print("ok", t.info())
except Exception as e:
    print("ERR", e, t.info())
