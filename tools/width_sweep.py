"""Interpreter time per dependency level versus level width (isolated phases, RV_PIPELINE=0).

Separates the per-level fixed cost (launch + fill/drain of the latency chain
gate record -> operand rows -> stores) from the per-gate streaming cost of k_interp_full:
    t_level(W) ~= t0 + W * bytes_per_gate / BW
Run on the GPU box:  python tools/width_sweep.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RV_PIPELINE"] = "0"

import numpy as np  # noqa: E402

import circuits  # noqa: E402
import reverie_amd  # noqa: E402
from reverie_amd import _lib  # noqa: E402
from reverie_amd.dist import HipShardBackend, prove_sharded  # noqa: E402


def main():
    import torch

    torch.cuda.set_device(0)  # torch's HIP runtime must initialise before the library's
    torch.zeros(1, device="cuda")
    seeds = np.random.default_rng(0x5EED).integers(0, 256, (256, 16), dtype=np.uint8)
    L = _lib.lib()
    total = int(os.environ.get("SWEEP_GATES", 1 << 22))
    for width in (8192, 16384, 32768, 65536, 131072, 262144, 524288):
        layers = max(total // width, 4)
        prog, wit, wc, st = circuits.layered_gf2(width=width, layers=layers)
        ctx = reverie_amd.Context(0)
        c = reverie_amd.Circuit(prog, wc, ctx)
        b = HipShardBackend(c)
        prove_sharded(b, wit, [], seeds, device_resident=True)
        L.rv_ctx_profile(ctx.handle, 1, 1, None)
        n = 3
        for _ in range(n):
            prove_sharded(b, wit, [], seeds, device_resident=True)
        prof = _lib.Profile()
        L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
        ms = {nm: prof.ms[i] / n for i, nm in enumerate(_lib.PHASES)}
        lv = c.info["levels"]
        print(f"width {width:7d} layers {layers:4d} levels {lv:4d} gates {st['gates']:9d} and {st['and']:9d}  "
              f"interp {ms['interp']:.3f} ms  = {ms['interp'] * 1e3 / lv:.2f} us/level  {ms['interp'] * 1e6 / st['gates']:.3f} ns/gate  "
              f"masks {ms['masks']:.3f} hash {ms['hash']:.3f} open {ms['open']:.3f}", flush=True)
        c.close()
        ctx.close()


if __name__ == "__main__":
    main()
