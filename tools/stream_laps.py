import os, sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, circuits, reverie_amd
from reverie_amd.stream import prove_streaming
ctx = reverie_amd.Context(0)
seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
rprog, rwit, rwc, rst = circuits.layered_gf2(layers=153, p_and=0.5, recycle=True)
for i in range(4):
    t0 = time.perf_counter()
    proof, info = prove_streaming(rprog, rwit, [], rwc, seeds=seeds, max_chunk_ops=int(os.environ.get("CHUNK", 1 << 18)), ctx=ctx)
    print("call", i, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), info["chunks"], file=sys.stderr)
