"""ctypes binding of libjasper_hip.so (C-ABI declared in include/jasper_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (JASPER_AMD_LIB: another build of the same library, for A/B timing of two builds in one run -- tools/ only)
LIB_PATH = os.environ.get("JASPER_AMD_LIB") or os.path.join(_HERE, "libjasper_hip.so")

JASPER_OK = 0
JASPER_ERR = -1
JASPER_ERR_CAPACITY = -2
JASPER_ERR_FORMAT = -3
JASPER_ERR_REFERENCE_EXIT = -4


class FixRec(C.Structure):
    _fields_ = [("index", C.c_int64), ("chunk", C.c_uint32), ("seqno", C.c_uint32), ("pass_", C.c_uint8),
                ("kind", C.c_uint8), ("newc", C.c_uint8), ("oldc", C.c_uint8), ("rep", C.c_uint32),
                ("aux_off", C.c_uint32), ("aux_len", C.c_uint32)]


assert C.sizeof(FixRec) == 32

# every symbol include/jasper_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "jasper_last_error": (C.c_char_p, []),
    "jasper_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "jasper_request_cancel": (C.c_int, [C.c_int]),
    "jasper_device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_table_create": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(_P)]),
    "jasper_table_load_jf": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "jasper_table_destroy": (None, [_P]),
    "jasper_table_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_table_sync": (C.c_int, [_P]),
    "jasper_table_clear": (C.c_int, [_P]),
    "jasper_count_bases": (C.c_int, [_P, C.c_char_p, C.c_uint64]),
    "jasper_count_bases_device": (C.c_int, [_P, _P, C.c_uint64]),
    "jasper_count_reads_text": (C.c_int, [_P, C.c_char_p, C.c_uint64]),
    "jasper_count_reads_files": (C.c_int, [_P, C.POINTER(C.c_char_p), C.c_int]),
    "jasper_count_reads_file_ranges": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "jasper_table_load_jf_part": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "jasper_table_write_jf": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]),
    "jasper_debug_mix": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jasper_last_ingest": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_histogram_part": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "jasper_histogram_is_fused": (C.c_int, [_P]),
    "jasper_histogram": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "jasper_lookup": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_uint32)]),
    "jasper_table_export": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_table_import": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint64]),
    "jasper_table_export_device": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(_P)]),
    "jasper_table_import_device": (C.c_int, [_P, _P, C.c_uint64]),
    "jasper_table_export_to": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jasper_device_free": (C.c_int, [_P, _P]),
    "jasper_table_export_packed": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32]),
    "jasper_table_import_packed": (C.c_int, [_P, _P, C.c_uint64, C.c_int]),
    "jasper_table_import_packed_multi": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64), C.c_uint32]),
    "jasper_table_reserve": (C.c_int, [_P, C.c_uint64]),
    "jasper_table_fit": (C.c_int, [_P, C.c_double]),
    "jasper_read_feed_start": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "jasper_read_feed_next": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "jasper_read_feed_release": (C.c_int, [_P]),
    "jasper_count_exchange_plan": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "jasper_count_exchange_scan": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jasper_count_exchange_partition": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, _P, _P, _P, C.c_uint64]),
    "jasper_count_exchange_dedupe": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, _P, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "jasper_count_exchange_insert": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_int, C.c_uint32, C.c_int]),
    "jasper_table_export_owner": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "jasper_table_export_file_ranges": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]),
    "jasper_table_write_jf_piece": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int]),
    "jasper_table_ipc_handle": (C.c_int, [_P, _P]),
    "jasper_table_attach_ipc": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "jasper_table_attach_tables": (C.c_int, [_P, C.POINTER(_P), C.c_uint32, C.c_uint32]),
    "jasper_table_detach": (C.c_int, [_P]),
    "jasper_ipc_probe": (C.c_int, [C.c_int, _P, C.c_uint32, C.c_uint32]),
    "jasper_table_release_retired": (C.c_int, [_P]),
    "jasper_inflate_file": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "jasper_owner_of": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint32]),
    "jasper_polish_batch": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "jasper_result_num_chunks": (C.c_int, [_P]),
    "jasper_polish_batch_device": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "jasper_result_seq": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "jasper_result_seq_len": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "jasper_result_seq_device": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "jasper_result_records": (C.c_int, [_P, C.POINTER(C.POINTER(FixRec)), C.POINTER(C.c_uint64)]),
    "jasper_result_aux": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "jasper_result_qv": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "jasper_result_qv_chunk": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "jasper_result_lookups": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "jasper_result_seconds": (C.c_double, [_P]),
    "jasper_result_segments": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_result_retried": (C.c_int, [_P]),
    "jasper_result_free": (None, [_P]),
    "jasper_asm_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "jasper_asm_close": (None, [_P]),
    "jasper_asm_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_asm_contig": (C.c_int, [_P, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_asm_split": (C.c_int, [_P, C.c_uint64, C.c_char_p, _P, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jasper_asm_split_wait": (C.c_int, [_P]),
    "jasper_asm_chunks": (C.c_int, [_P, _P, _P, _P, _P]),
    "jasper_asm_file_bytes": (C.c_int, [_P, _P]),
    "jasper_asm_chunk_text": (C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "jasper_asm_polish": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "jasper_asm_pin": (C.c_int, [_P, C.c_int]),
    "jasper_asm_take": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "jasper_asm_put": (C.c_int, [_P, C.c_uint64, C.c_char_p, C.c_uint64]),
    "jasper_asm_write_fixed": (C.c_int, [_P, _P, C.POINTER(C.c_char_p), C.c_uint32, C.c_int]),
    "jasper_asm_polished_lens": (C.c_int, [_P, _P, _P]),
    "jasper_asm_join": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int]),
    "jasper_merge_fix_csvs": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p]),
    "jasper_last_count_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "jasper_last_count_stages": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
}

_lib = None


class JasperHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libjasper_hip: %s (code %d)" % (msg, code))
        self.code = code


def _share_hip_runtime_with_torch():
    """One process must use ONE HIP/HSA runtime. PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64;
    if libjasper_hip.so pulled in /opt/rocm's copies first, a later `import torch` finds "No HIP GPUs" (the second
    HSA runtime cannot open the device). So when torch is installed but not imported yet, load ITS runtime first;
    libjasper_hip.so's NEEDED libamdhip64.so.7 then resolves to the already-loaded copy. Without torch the system
    ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """load libjasper_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `make -C jasper_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise JasperHipError(rc, lib().jasper_last_error().decode(errors="replace"))


def kernel_source_digest():
    """sha256 over the native sources (jasper_amd/csrc/*.hip|hpp|cpp, include/*.h), file names and contents in sorted order:
    what a set of hardware counters under profiles/ was taken from.  bench.py cites committed counters only when this digest is
    the one stored with them (tools/summarize_prof.py) -- counters of other kernels are not evidence."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "jasper_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "jasper_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for fn in files:
        h.update(os.path.basename(fn).encode() + b"\0")
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()
