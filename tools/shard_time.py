#!/usr/bin/env python3
"""Per-rank time of a sharded proof on ONE GPU: what each of N GPUs would spend on its 256/N repetitions of the
headline circuit (commit -> digests to host -> host challenge -> open), without the collective.  Gives the
strong-scaling ceiling of bench.py --gpus N before RCCL and the rank-0 gather are added."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import circuits  # noqa: E402
import reverie_amd  # noqa: E402
from reverie_amd.dist import HipShardBackend  # noqa: E402
from reverie_amd.proof import challenge, combine_digests  # noqa: E402


def main():
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    layers = int(os.environ.get("LAYERS", 153))
    prog, wit, wc, st = circuits.layered_gf2(layers=layers)
    seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
    ctx = reverie_amd.Context(0)
    c = reverie_amd.Circuit(prog, wc, ctx)
    be = HipShardBackend(c)
    base = None
    for world in (1, 2, 4, 8):
        count = 256 // world
        import ctypes as C

        from reverie_amd import _lib
        ts = []
        _lib.lib().rv_ctx_profile(ctx.handle, 1, 1, None)
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            shard = be.commit(wit, [], seeds[:count], 0, count)
            h = be.digests(shard)
            # stand-in for the all-gather: the other ranks' digests are whatever; the challenge only has to be a valid map
            allh = np.zeros((256, 32), np.uint8)
            allh[:count] = h
            omit = challenge(combine_digests(allh))
            lens = be.open_sizes(shard, omit)
            buf = torch.empty(max(sum(lens), 1), dtype=torch.uint8, device="cuda")
            be.open_into(shard, omit, buf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            be.destroy(shard)
        prof = _lib.Profile()
        _lib.lib().rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
        ph = {nm: round(prof.ms[i] / 5, 3) for i, nm in enumerate(_lib.PHASES)}
        t = min(ts[1:])
        base = base or t
        print(json.dumps({"gpus": world, "reps_per_gpu": count, "ms_per_rank": t * 1e3, "and_per_s_if_perfectly_parallel": st["and"] / t,
                          "strong_scaling_ceiling": base / (t * world), "device_phase_ms": ph}), flush=True)


if __name__ == "__main__":
    main()
