"""Program files of the reference CLI: bincode(`Vec<mcircuit::CombineOperation>`), read at
/root/reference/src/main.rs:66,98,122.  The (de)serialiser is C++ behind the C-ABI
(`rv_program_from_bincode` / `rv_program_to_bincode`, reverie_amd/csrc/program.cpp); this is the ctypes
wrapper.  The enum's variant order is recalled, not verifiable here — see the header."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .ops import OP_DTYPE, program


def loads(data: bytes) -> np.ndarray:
    """bincode bytes -> rv_op array"""
    data = bytes(data)
    ops = C.c_void_p()
    n = C.c_size_t()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    _lib.check(_lib.lib().rv_program_from_bincode(buf, C.c_size_t(len(data)), C.byref(ops), C.byref(n)))
    prog = np.frombuffer(C.string_at(ops, n.value * OP_DTYPE.itemsize), dtype=OP_DTYPE).copy()
    _lib.lib().rv_free(ops)
    return prog


def dumps(prog) -> bytes:
    """rv_op array (or list of op tuples) -> bincode bytes"""
    prog = program(prog)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(_lib.lib().rv_program_to_bincode(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.byref(out), C.byref(n)))
    data = C.string_at(out, n.value)
    _lib.lib().rv_free(out)
    return data
