"""ctypes binding of libreverie_amd.so (the C-ABI declared in include/reverie_amd.h).

There is NO CPU fallback: if the HIP library is missing, or no gfx950 device is visible,
every compute entry point raises.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RV_LIB_PATH: an alternative build of the same library (A/B measurements of kernel variants, tools/)
LIB_PATH = os.environ.get("RV_LIB_PATH") or os.path.join(_HERE, "_build", "libreverie_amd.so")

RV_OK = 0
ERRORS = {
    1: "WITNESS_INVALID", 2: "WITNESS_SHORT", 3: "WIRE_OOB", 4: "PROOF_MALFORMED", 5: "BAD_OP",
    6: "NOMEM", 7: "DEVICE", 8: "UNSUPPORTED", 9: "ARG",
}


class ReverieError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        self.name = ERRORS.get(code, "?")
        super().__init__(f"reverie_amd error {code} ({self.name}){': ' + detail if detail else ''}")


class ShardParts(C.Structure):
    _fields_ = [
        ("gf2_online", C.c_void_p), ("gf2_pre", C.c_void_p), ("z64_online", C.c_void_p), ("z64_pre", C.c_void_p),
        ("gf2_online_len", C.c_size_t), ("gf2_pre_len", C.c_size_t), ("z64_online_len", C.c_size_t),
        ("z64_pre_len", C.c_size_t), ("n_online", C.c_uint32), ("n_pre", C.c_uint32),
    ]


class CircuitInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_ops", "gf2_inputs", "gf2_muls", "gf2_asserts", "gf2_linear", "gf2_masks", "z64_inputs", "z64_muls",
        "z64_asserts", "z64_linear", "z64_masks", "b2a", "levels", "device_bytes", "scratch_bytes", "compile_us", "upload_us",
        "gf2_operand_rows", "gf2_rows_written")]


class BristolInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_gates", "n_wires", "n_inputs", "n_outputs", "n_and", "n_xor", "n_inv",
                                          "n_other", "gf2_wires")]


class StreamInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_ops", "chunks", "levels", "gf2_masks", "z64_masks", "gf2_muls", "z64_muls", "wire_store_bytes",
                                          "peak_chunk_bytes", "hash_state_bytes", "proof_bytes")] + [("pass_", C.c_uint32), ("kept_mib", C.c_uint32)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8), ("calls", C.c_uint64)]


PHASES = ["setup", "masks", "interp", "hash", "join", "open"]


# every symbol include/reverie_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "rv_strerror", "rv_last_error", "rv_abi_version", "rv_ctx_create", "rv_ctx_destroy", "rv_ctx_sync",
    "rv_circuit_compile", "rv_circuit_compile_ex", "rv_hook_compile_info", "rv_hook_compile_compare", "rv_circuit_destroy", "rv_circuit_get_info", "rv_circuit_early_staging_bytes", "rv_ctx_ops_cache_clear", "rv_prove", "rv_verify", "rv_free",
    "rv_shard_commit", "rv_shard_digests_device", "rv_shard_digests", "rv_shard_open", "rv_shard_destroy",
    "rv_shard_open_device", "rv_combine_digests", "rv_challenge", "rv_assemble_proof", "rv_verify_shard",
    "rv_verify_finish", "rv_hook_prg_blocks", "rv_hook_expand_seed", "rv_hook_sharegen_gf2", "rv_hook_sharegen_z64",
    "rv_hook_blake3", "rv_hook_shard_stream_digests", "rv_ctx_profile", "rv_shard_digests_to_device", "rv_shard_open_size", "rv_shard_open_into", "rv_shard_open_self", "rv_shard_open_gathered",
    "rv_bristol_parse", "rv_circuit_record_sizes", "rv_program_from_bincode", "rv_program_to_bincode", "rv_prove_batch", "rv_prove_device",
    "rv_verify_ex", "rv_verify_shard_ex", "rv_verify_finish_ex", "rv_verify_batch",
    "rv_hook_gf2_reconstruct", "rv_hook_z64_reconstruct", "rv_hook_early_proofs", "rv_hook_open_direct_proofs", "rv_hook_ops_cache_hits", "rv_hook_ops_same", "rv_hook_overlap_commits", "rv_hook_early_plan", "rv_hook_verify_vc_count",
    "rv_stream_begin", "rv_stream_feed", "rv_stream_commit", "rv_stream_finish", "rv_stream_abort", "rv_stream_get_info", "rv_stream_same_cuts",
    "rv_prove_streaming", "rv_prove_ops", "rv_verify_ops", "rv_stream_verify_begin", "rv_stream_verify_finish", "rv_verify_streaming",
    "rv_comm_unique_id", "rv_comm_create", "rv_comm_create_all", "rv_comm_destroy", "rv_prove_sharded", "rv_prove_multi",
]
RV_VERIFY_STRICT = 1
RV_COMPILE_WHOLE_PROVER = 1
RV_VERIFY_REFERENCE_COMPAT = 2  # the reference verifier's two unchecked conditions stay unchecked

_lib = None


def _pin_hip_runtime():
    """A process must run ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64 next to torch/lib, this library
    links the system one; whichever is loaded first wins the SONAME, and if that is the system copy a later
    `import torch` finds no GPU ("No HIP GPUs are available") or corrupts the heap at exit.  So when PyTorch is
    installed its copy is loaded first (without importing torch), and both sides then share it — the arrangement
    bench.py and the multi-GPU path (which import torch first anyway) have always run in."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ReverieError(7, f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(make -C reverie_amd/csrc); there is no CPU fallback")
        _pin_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.rv_strerror.restype = C.c_char_p
        L.rv_last_error.restype = C.c_char_p
        L.rv_abi_version.restype = C.c_uint32
        for name in SYMBOLS:
            fn = getattr(L, name)
            if name in ("rv_ctx_destroy", "rv_circuit_destroy", "rv_shard_destroy", "rv_free", "rv_stream_abort", "rv_comm_destroy"):
                fn.restype = None
            elif name in ("rv_hook_early_proofs", "rv_hook_open_direct_proofs", "rv_hook_verify_vc_count", "rv_hook_ops_cache_hits", "rv_hook_overlap_commits"):
                fn.restype = C.c_uint64
            elif name not in ("rv_strerror", "rv_last_error", "rv_abi_version"):
                fn.restype = C.c_int
        _lib = L
    return _lib


def check(rc: int):
    if rc != RV_OK:
        raise ReverieError(rc, lib().rv_last_error().decode() if rc in (6, 7) else "")
