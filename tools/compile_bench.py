"""Host-side compile time of the benchmark circuits (no GPU work): sequential compiler vs the parallel one at several
thread counts (RV_COMPILE_THREADS), with the per-pass laps of RV_COMPILE_STATS on stderr.
usage: python tools/compile_bench.py [gf2|z64] [threads ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import circuits  # noqa: E402
from reverie_amd import _lib  # noqa: E402

L = _lib.lib()
for f in ("enabled", "defrag"):
    try:
        print("transparent_hugepage/%s:" % f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as e:
        print(e)
which = sys.argv[1] if len(sys.argv) > 1 else "gf2"
threads = [int(x) for x in sys.argv[2:]] or [8, 16, 32, 64]
prog, wit, wc, st = circuits.layered_gf2() if which == "gf2" else circuits.layered_z64()
a = np.zeros(60_000_000, dtype=np.uint32)
t = time.perf_counter()
a[:] = 1
print("first touch of 240 MB: %.3f s" % (time.perf_counter() - t), flush=True)
del a


def compile_once(flags):
    ci = _lib.CircuitInfo()
    t = time.perf_counter()
    rc = L.rv_hook_compile_info(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])),
                                C.c_uint32(flags), C.c_size_t(0), C.byref(ci))
    assert rc == 0
    return time.perf_counter() - t


os.environ["RV_COMPILE_SEQ"] = "1"
print("%s %d ops  sequential: %s" % (which, len(prog), ["%.3f" % compile_once(1) for _ in range(2)]), flush=True)
os.environ["RV_COMPILE_SEQ"] = "0"
for th in threads:
    os.environ["RV_COMPILE_THREADS"] = str(th)
    print("threads %3d: %s" % (th, ["%.3f" % compile_once(1) for _ in range(3)]), flush=True)
d = C.c_int(-7)
rc = L.rv_hook_compile_compare(prog.ctypes.data_as(C.c_void_p), C.c_size_t(len(prog)), C.c_size_t(int(wc[0])), C.c_size_t(int(wc[1])), C.c_uint32(1),
                               C.c_int(threads[-1]), C.byref(d))
print("parallel == sequential:", rc == 0 and d.value == 0)
