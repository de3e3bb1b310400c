"""Does a copy-engine transfer beside the interpreter's level launches slow them, and in which direction?  rv_prove (RV_EARLY=0)
of the 10^7-gate circuit with a torch thread keeping device-to-host, host-to-device or device-to-device copies of 64 MB in flight
on a stream of its own; per-phase HIP-event times.   python tools/copy_beside.py"""
import os, sys, threading, time
import ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["RV_EARLY"] = "0"
import numpy as np, torch
import reverie_amd as rv
from reverie_amd import _lib
import circuits

prog, wit, wc, st = circuits.layered_gf2()
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
c = rv.Circuit(prog, wc, whole_prover=True)
L = _lib.lib(); ctx = rv.Context.default()
n = 64 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda"); dev2 = torch.empty_like(dev)
host = torch.empty(n, dtype=torch.uint8).pin_memory()
side = torch.cuda.Stream()

def run(kind):
    stop = threading.Event()
    def pump():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(4):
                    if kind == "d2h": host.copy_(dev, non_blocking=True)
                    elif kind == "h2d": dev.copy_(host, non_blocking=True)
                    elif kind == "d2d": dev2.copy_(dev, non_blocking=True)
                side.synchronize()
    th = threading.Thread(target=pump) if kind != "none" else None
    if th: th.start()
    for _ in range(3): rv.Proof.new(c, wit, [], seeds=seeds)
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    t = time.perf_counter(); N = 12
    for _ in range(N): rv.Proof.new(c, wit, [], seeds=seeds)
    dt = (time.perf_counter() - t) / N * 1e3
    prof = _lib.Profile(); L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    stop.set()
    if th: th.join()
    print(f"{kind:5s}: {dt:.3f} ms/proof  " + "  ".join(f"{nm} {prof.ms[i] / max(prof.calls, 1):.3f}" for i, nm in enumerate(_lib.PHASES)), flush=True)

for k in ("none", "d2h", "h2d", "d2d", "none"):
    run(k)
