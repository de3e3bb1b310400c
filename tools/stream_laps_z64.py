"""config 5 (10^6 Z64 MUL, recycled wire indices) through rv_prove_streaming, four calls (RV_STREAM_STATS=1 for the laps)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, circuits, reverie_amd
from reverie_amd.stream import prove_streaming
ctx = reverie_amd.Context(0)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
prog, w64, wc, st = circuits.layered_z64(n_mul=int(os.environ.get("Z64_MULS", "1000000")), recycle=True)
for i in range(3):
    t0 = time.perf_counter()
    proof, info = prove_streaming(prog, [], w64, wc, seeds=seeds, max_chunk_ops=int(os.environ.get("CHUNK", 1 << 16)), ctx=ctx)
    print("call", i, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), info["chunks"], "kept MiB", info["kept_mib"], file=sys.stderr)
