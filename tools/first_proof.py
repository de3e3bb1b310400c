"""Where the time of the FIRST proof of a circuit goes (rv_prove_ops on the 10^7-gate benchmark circuit): the Python mirror's
call, and inside it compile / upload / prove / release (RV_COMPILE_STATS=1 prints the library's laps on stderr)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import circuits  # noqa: E402
import reverie_amd  # noqa: E402
from reverie_amd import _lib  # noqa: E402

prog, wit, wc, st = circuits.layered_gf2()
seeds = np.arange(4096, dtype=np.uint32).astype(np.uint8).reshape(256, 16)
ctx = reverie_amd.Context(0)
L = _lib.lib()
g = np.ascontiguousarray(np.asarray(wit, np.uint8))
for it in range(4):
    ctx.sync()
    t0 = time.perf_counter()
    c = C.c_void_p()
    _lib.check(L.rv_circuit_compile_ex(ctx.handle, prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(wc[0]), C.c_size_t(wc[1]),
                                       C.c_uint32(1), C.byref(c)))
    t1 = time.perf_counter()
    out, n = C.c_void_p(), C.c_size_t()
    _lib.check(L.rv_prove(ctx.handle, c, g.ctypes.data_as(C.c_void_p), C.c_size_t(len(g)), None, C.c_size_t(0), seeds.ctypes.data_as(C.c_void_p),
                          C.byref(out), C.byref(n)))
    t2 = time.perf_counter()
    L.rv_circuit_destroy(c)
    t3 = time.perf_counter()
    L.rv_free(out)
    print("compile+upload %.1f ms, prove %.1f ms, destroy %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
    t0 = time.perf_counter()
    p = reverie_amd.Proof.new(prog, wit, [], wc, seeds=seeds, ctx=ctx)
    print("Proof.new(ops) %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
