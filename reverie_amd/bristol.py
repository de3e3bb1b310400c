"""Bristol / Bristol Fashion front end (host only): text -> rv_op program.

The reference README advertises Bristol-format circuits but leaves parsing to the
un-vendored `mcircuit` crate (/root/reference/README.md:14-16, src/lib.rs:6-7).  The
parser itself is C++ behind the C-ABI (`rv_bristol_parse`, reverie_amd/csrc/bristol.cpp);
this is the ctypes wrapper.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .ops import OP_DTYPE


def parse(text: str | bytes, expected_outputs: Optional[Sequence[int]] = None, fmt: int = 0) -> Tuple[np.ndarray, dict]:
    """-> (program, info).  With expected_outputs the program ends with one AddConst+AssertZero
    per output wire, i.e. it states "this witness drives the circuit to these outputs".
    info["wire_counts"] is the (z64, gf2) tuple Proof.new / Proof.verify take.  A wrong number of expected outputs
    is RV_E_ARG (checked by the parser before it reads any of them)."""
    data = text.encode() if isinstance(text, str) else bytes(text)
    exp = None
    if expected_outputs is not None:
        exp = np.ascontiguousarray(np.asarray(expected_outputs, dtype=np.uint8))
    ops = C.c_void_p()
    n = C.c_size_t()
    info = _lib.BristolInfo()
    _lib.check(_lib.lib().rv_bristol_parse(data, C.c_size_t(len(data)), C.c_int(fmt),
                                           exp.ctypes.data_as(C.c_void_p) if exp is not None else None,
                                           C.c_size_t(len(exp) if exp is not None else 0),
                                           C.byref(ops), C.byref(n), C.byref(info)))
    prog = np.frombuffer(C.string_at(ops, n.value * OP_DTYPE.itemsize), dtype=OP_DTYPE).copy()
    _lib.lib().rv_free(ops)
    d = {k: int(getattr(info, k)) for k, _ in info._fields_}
    d["wire_counts"] = (0, d["gf2_wires"])
    return prog, d
