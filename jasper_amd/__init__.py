"""jasper_amd -- MI355X-native k-mer polishing engine (drop-in for the hot path of alguoo314/JASPER).

    KmerTable          HBM-resident canonical k-mer count table (count / histogram / lookup / polish)
    polisher           host-side mirror of src/jasper.py and src/jellyfish.py
    cli                drop-in for src/jasper.sh
    dist               read sharding + table merge across GPUs (one process per GPU, RCCL)
"""
from .table import KmerTable, PolishResult  # noqa: F401

__all__ = ["KmerTable", "PolishResult"]
