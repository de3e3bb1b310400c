"""Shader clock / power while proofs run back to back (is the co-resident schedule power-limited?): python tools/clock_watch.py [seconds]
Samples rocm-smi in a thread while the main thread proves; prints min / median / max of sclk and average socket power per variant."""
import os, subprocess, sys, threading, time, re, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import reverie_amd as rv
import circuits

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
prog, wit, wc, st = circuits.layered_gf2()
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
c = rv.Circuit(prog, wc, whole_prover=True)
for _ in range(3):
    p = rv.Proof.new(c, wit, [], seeds=seeds)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), str(e)))


def idle_sample(tag):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
    print(tag, out[:600])


idle_sample("idle:")
th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    p = rv.Proof.new(c, wit, [], seeds=seeds)
    n += 1
dt = time.perf_counter() - t0
stop = True
th.join()
print(f"{n} proofs in {dt:.2f} s = {dt / n * 1e3:.3f} ms per proof back to back; {len(samples)} samples")
for t, out in samples[:: max(1, len(samples) // 8)]:
    print(f"  t={t - t0:6.2f}s", out[:400].replace("\n", " "))
