#!/usr/bin/env python3
"""Headline benchmark: prover AND-gates/sec on a GF(2) circuit (BASELINE.json metric), on the boundary SURVEY §8(d)
defines: `rv_prove` from "compiled gate stream resident on the GPU + witness bytes on the host" to "bincode(Proof)
bytes on the host" -- seeds, AES-CTR masks, interpreter, BLAKE3 commitments, Fiat-Shamir challenge, openings AND the
device-to-host copy of the proof (what the reference's bench_prover times around Proof::new,
/root/reference/src/proof/mod.rs:346-353, plus the PCIe leg a GPU prover has to pay).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[3] — the synthetic 10^7-gate layered random AND/XOR GF(2) circuit of
SURVEY §8d (SplitMix64 seed 0x5EED000000000004, 4096 inputs, 153 layers x 65536 gates, p(AND)=1/2, folded + asserted
tail), full KKW parameters (256 repetitions, 8 players, 40 opened).  It fits one GPU, and it is the configuration the
metric (and the north-star 10^9 AND/s target) is quoted on.  One step = one complete proof, host to host.  With N GPUs
the 256 repetitions are split N ways (strong scaling of ONE proof), one RCCL all-gather of digests per proof, and the
step ends with the assembled proof bytes on rank 0's host.

Next to `value` the line carries, all outside the timed region and each with its own parity flag: `device_resident`
(round 1's headline: the same proof left in HBM), `prove_batch_host` (rv_prove_batch, host to host), `all_and`
(the 10^7-AND variant), `secondary` (AES-128 / SHA-256 Bristol circuits, single proof and 64 / 256 proofs per call;
the 10^6-MUL Z64 circuit), `streaming` (the bounded-memory prover), `verifier`, and `cpu_baseline` (the CPU oracle,
median of five proofs of the same workload, CPU model and core counts stated).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PROFILE_TAG = "r06"    # profiles/<tag>_pmc_traffic.json feeds roofline.traffic


def rule_seeds():
    rng = np.random.default_rng(0x5EED)
    return rng.integers(0, 256, (256, 16), dtype=np.uint8)


def host_info():
    """CPU model, logical CPUs, physical cores (distinct (physical id, core id) pairs) of this host"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return {"cpu_model": model, "logical_cpus": logical, "physical_cores": len(cores) or logical, "cpu_quota": cpu_quota()}


def cpu_quota():
    """CPUs this process's cgroup may use at a time (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1), None when unlimited or
    unreadable.  The benchmark boxes grant 16 of their 256 logical CPUs: threads beyond the quota only take turns."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(prog, w2, w64, wc, seeds, units, unit, what, runs=5):
    """The CPU oracle (a C port of the reference algorithm: packed u64 groups, AES-NI CTR, AVX2 movemask transpose,
    eight-chunk AVX2 BLAKE3 behind a 64 KiB staged hasher, one thread per packed group like rayon at
    proof/mod.rs:128) timed on this host: median of `runs` proofs.  -> (record, proof bytes)"""
    import oracle_lib

    hi = host_info()
    threads = max(1, min(32, hi["physical_cores"]))
    if hi["cpu_quota"]:
        threads = max(1, min(threads, int(hi["cpu_quota"] + 0.5)))  # (more threads than granted CPUs would only take turns)
    times, proof = [], None
    for _ in range(runs):
        t0 = time.perf_counter()
        proof = oracle_lib.prove(prog, w2, w64, wc, seeds, threads=threads)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {
        "value": units / med, "unit": unit, "cores": threads, "kind": "port",
        "sample": f"{what}: {runs} proofs, median {med:.3f} s (min {min(times):.3f}, max {max(times):.3f}); "
                  f"{threads} threads over the 32 packed groups (one each at most), capped by the physical cores and by the cgroup's CPU quota",
        "cpu_model": hi["cpu_model"], "physical_cores": hi["physical_cores"], "logical_cpus": hi["logical_cpus"], "cpu_quota": hi["cpu_quota"],
        "simd": "AES-NI, AVX2 (movemask bit transpose, 8-way BLAKE3)", "proof_bytes": len(proof),
    }, proof


def shard_times(ctx, prog, wc, wit, seeds, n_and):
    """Per-rank time of a sharded proof on ONE GPU (tools/shard_time.py): what each of N GPUs would spend on its 256 / N
    repetitions of the workload -- commit -> digests to the host -> challenge -> openings left in HBM -- without the collective.
    The strong-scaling ceiling of --gpus N follows from it: t(256) / (N * t(256 / N)).  Neutral gate stream (no prover hint), as
    the multi-GPU path compiles it."""
    import torch

    import reverie_amd
    from reverie_amd.dist import HipShardBackend
    from reverie_amd.proof import challenge, combine_digests

    c = reverie_amd.Circuit(prog, wc, ctx)
    be = HipShardBackend(c)
    out = {}
    for count in (256, 128, 64, 32):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            shard = be.commit(wit, [], seeds[:count], 0, count)
            h = be.digests(shard)
            allh = np.zeros((256, 32), np.uint8)  # (the other ranks' digests: the challenge only has to be a valid map)
            allh[:count] = h
            omit = challenge(combine_digests(allh))
            lens = be.open_sizes(shard, omit)
            buf = torch.empty(max(sum(lens), 1), dtype=torch.uint8, device="cuda")
            be.open_into(shard, omit, buf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            be.destroy(shard)
        out[str(count)] = min(ts[1:]) * 1e3
    c.close()
    base = out["256"]
    return {"ms": out, "strong_scaling_ceiling": {str(256 // int(k)): base / (v * (256 // int(k))) for k, v in out.items() if k != "256"},
            "note": "one rank's share of a sharded proof (commit, digests to the host, host challenge, openings left in HBM), min of 4 after a "
                    "warm-up, no collective; ceiling[N] = t(256) / (N * t(256 / N)) = the best strong-scaling efficiency N GPUs can reach"}


class HostProver:
    """rv_prove through ctypes with as little Python in the loop as the FFI allows: witness and seeds are numpy arrays
    made once, every call returns the library's page-locked proof buffer, which is handed back with rv_free"""

    def __init__(self, circuit, w2, w64, seeds):
        from reverie_amd import _lib

        self.L = _lib.lib()
        self.check = _lib.check
        self.circuit = circuit
        self.g = np.ascontiguousarray(np.asarray(w2, dtype=np.uint8))
        self.z = np.ascontiguousarray(np.asarray(w64, dtype=np.uint64))
        self.s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(256, 16)
        self.args = (circuit.ctx.handle, circuit.handle, self.g.ctypes.data_as(C.c_void_p) if self.g.size else None, C.c_size_t(len(self.g)),
                     self.z.ctypes.data_as(C.c_void_p) if self.z.size else None, C.c_size_t(len(self.z)), self.s.ctypes.data_as(C.c_void_p))

    def prove(self):
        out, n = C.c_void_p(), C.c_size_t()
        self.check(self.L.rv_prove(*self.args, C.byref(out), C.byref(n)))
        return out, n.value

    def free(self, p):
        self.L.rv_free(p)

    def run_each(self, steps, warm=2):
        """every proof timed on its own (rv_prove returns with the bytes on the host) after `warm` untimed ones
        -> (sorted seconds per proof, bytes of the last proof): the secondary records quote the MEDIAN with min / max, so that a
        one-off (a buffer first mapped inside the loop, a neighbour's burst) cannot pass for the rate"""
        last, ts = None, []
        for i in range(warm + steps):
            t0 = time.perf_counter()
            p, n = self.prove()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
            if last is not None:
                self.free(last[0])
            last = (p, n)
        data = C.string_at(last[0], last[1])
        self.free(last[0])
        return sorted(ts), data

    def run(self, steps):
        """-> (seconds, bytes of the last proof)"""
        last = None
        t0 = time.perf_counter()
        for _ in range(steps):
            p, n = self.prove()
            if last is not None:
                self.free(last[0])
            last = (p, n)
        dt = time.perf_counter() - t0
        data = C.string_at(last[0], last[1])
        self.free(last[0])
        return dt, data


def bristol_case(name):
    import hashlib

    import bristol_gen
    from reverie_amd import bristol

    bits = lambda d: [(b >> (7 - k)) & 1 for b in d for k in range(8)]  # noqa: E731
    if name == "aes128":
        key = bytes(range(16)); pt = bytes.fromhex("00112233445566778899aabbccddeeff")
        prog, info = bristol.parse(bristol_gen.aes128(), expected_outputs=bits(bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")))
        return prog, bits(key) + bits(pt), info["wire_counts"], info["n_and"]
    block = b"abc" + b"\x80" + bytes(52) + (24).to_bytes(8, "big")
    prog, info = bristol.parse(bristol_gen.sha256_block(), expected_outputs=bits(hashlib.sha256(b"abc").digest()))
    return prog, bits(block), info["wire_counts"], info["n_and"]


def phase_ms(ctx, run, n):
    """per-phase HIP-event times (ms per proof) of `run()`, which makes n proofs"""
    from reverie_amd import _lib

    L = _lib.lib()
    L.rv_ctx_profile(ctx.handle, 1, 1, None)
    run()
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    return {nm: prof.ms[i] / n for i, nm in enumerate(_lib.PHASES)}


def secondary_records(ctx, seeds, quick):
    """Configs 2, 3 and 5 through the host-bytes entry points; every record carries its parity flag"""
    import circuits
    import oracle_lib
    import reverie_amd

    out = {}
    for name in ("aes128", "sha256"):
        prog, w2, wc, n_and = bristol_case(name)
        circ = reverie_amd.Circuit(prog, wc, ctx)
        hp = HostProver(circ, w2, [], seeds)
        hp.run(3)
        lat = []
        for _ in range(10 if quick else 30):
            dt, data = hp.run(1)
            lat.append(dt)
        want = oracle_lib.prove(prog, w2, [], wc, seeds, threads=8)
        rec = {"and_gates": n_and, "levels": circ.info["levels"], "proof_bytes": len(data),
               "single_proof_ms": statistics.median(lat) * 1e3, "single_proof_and_per_s": n_and / statistics.median(lat),
               "bit_exact_vs_cpu": data == want, "verifies_strict": bool(reverie_amd.Proof(data).verify(circ)),
               "phase_ms": phase_ms(ctx, lambda: hp.run(10), 10),
               "note": "narrow levels run as LDS runs (csrc/ldsrun.*: 32 slices of 8 repetitions, live wires in LDS); "
                       "RV_LDS_RUN=0 gives the one-workgroup row interpreter of round 1"}
        pv = reverie_amd.Proof(data)
        pv.verify(circ)
        tv = []
        for _ in range(10):
            t0 = time.perf_counter()
            pv.verify(circ)
            tv.append(time.perf_counter() - t0)
        rec["single_verify_ms"] = statistics.median(tv) * 1e3
        del pv
        for B in (8, 64, 256):
            rng = np.random.default_rng(B)
            bs = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
            bs[0] = seeds
            bw = np.tile(np.asarray(w2, np.uint8), (B, 1))
            reverie_amd.Proof.new_batch(circ, bw, seeds=bs)
            t0 = time.perf_counter()
            n_calls = 2 if quick else 4
            for _ in range(n_calls):
                proofs = reverie_amd.Proof.new_batch(circ, bw, seeds=bs)
            dt = (time.perf_counter() - t0) / n_calls
            reverie_amd.verify_batch(circ, proofs)
            t0 = time.perf_counter()
            oks = reverie_amd.verify_batch(circ, proofs)
            tv = time.perf_counter() - t0
            rec[f"batch{B}"] = {"ms_per_call": dt * 1e3, "us_per_proof": dt / B * 1e6, "and_per_s": n_and * B / dt,
                                "first_proof_bit_exact_vs_cpu": bytes(proofs[0]) == want,
                                "verify_batch_all_ok": all(oks), "verify_us_per_proof": tv / B * 1e6}
            del proofs
        t0 = time.perf_counter()
        for _ in range(3):
            oracle_lib.prove(prog, w2, [], wc, seeds, threads=32)
        rec["cpu_oracle_and_per_s"] = n_and * 3 / (time.perf_counter() - t0)
        out[name] = rec
        circ.close()
    # config 5: Z64, 10^6 MUL (the full size on the GPU, byte-compared with the CPU oracle's proof of the same circuit once per run
    # when the host has the memory for it; a 10^5-MUL sample of the same generator besides)
    n_mul = 100_000 if quick else 1_000_000
    prog, w64, wc, st = circuits.layered_z64(n_mul=n_mul)
    circ = reverie_amd.Circuit(prog, wc, ctx)
    hp = HostProver(circ, [], w64, seeds)
    # warm-up of TWO proofs: run() holds the previous proof's buffer while the next one is made, and the second 640 MB page-locked
    # output buffer takes 45 ms to map the first time it is needed (with one warm-up proof that landed in the timed three: +15 ms
    # per proof in every earlier record of this number)
    steps = 5
    tz, data = hp.run_each(steps, warm=2)
    dt = tz[len(tz) // 2] * steps  # (median per proof)
    p = reverie_amd.Proof(data)
    ok = bool(p.verify(circ))  # (the first call also sizes the context's buffer cache for the verifier's rows)
    t0 = time.perf_counter()
    ok = bool(p.verify(circ)) and ok
    tv = time.perf_counter() - t0
    rec = {"mul_gates": st["mul"], "levels": circ.info["levels"], "proof_bytes": len(data), "ms_per_proof": dt / steps * 1e3,
           "ms_min_max": [tz[0] * 1e3, tz[-1] * 1e3],
           "mul_per_s": st["mul"] * steps / dt, "verifies_strict": ok, "verify_ms": tv * 1e3,
           "phase_ms": phase_ms(ctx, lambda: hp.run(2), 2),
           "note": "host to host; the 640 MB proof alone is ~11 ms of PCIe, of which the early-corrections path (the corrections vectors of all "
                   "256 repetitions cross PCIe in twelve chunks while the kernels run: csrc/api.hip, 2 GB of page-locked staging) hides ~5; the prover runs the mask generator inside the "
                   "interpreter's level launches (k_z64_fused, csrc/aes.hip; RV_Z64_FUSED=0 = k_aes_z64_masks then k_interp64: 26.9 + 26.4 ms); "
                   f"profiles/{PROFILE_TAG}_z64_* hold the kernel trace and the PMC traffic of both (fused: 119.2 GB and 38.5 ms per proof, "
                   "two kernels: 182 GB), DESIGN.md the analysis (2.05e9 cipher blocks at the VALU rate = 27 ms)"}
    circ.close()
    del p
    # the FULL circuit through the CPU oracle once (a 640 MB proof, ~20 GB of host memory, ~10 s on 32 threads): the byte-for-byte
    # check of the timed configuration itself, and the CPU baseline on it; skipped (with the 10^5-MUL sample standing in) when the
    # host is short of memory
    mem_gb = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                mem_gb = int(line.split()[1]) / 1e6
    except OSError:
        pass
    if not quick and mem_gb >= 64:
        base, full = cpu_baseline(prog, [], w64, wc, seeds, st["mul"], "Z64 MUL gates/s", "the whole timed workload (10^6 MUL)", runs=1)
        rec["full_size_bit_exact_vs_cpu"] = data == full
        rec["cpu_baseline"] = base
        del full
    else:
        rec["full_size_bit_exact_vs_cpu"] = None
        rec["full_size_note"] = f"host has {mem_gb:.0f} GB available (or --quick): the oracle's full-size proof was not made"
    del data
    sprog, sw64, swc, sst = circuits.layered_z64(n_mul=100_000)
    scirc = reverie_amd.Circuit(sprog, swc, ctx)
    dts, sdata = HostProver(scirc, [], sw64, seeds).run(1)
    sbase, sproof = cpu_baseline(sprog, [], sw64, swc, seeds, sst["mul"], "Z64 MUL gates/s", "10^5-MUL sample of the same generator", runs=3)
    rec["sample_1e5_bit_exact_vs_cpu"] = sdata == sproof
    rec.setdefault("cpu_baseline", sbase)
    rec["cpu_baseline_1e5_sample"] = sbase
    scirc.close()
    out["z64"] = rec
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=153, help="circuit depth (153 = the BASELINE workload)")
    ap.add_argument("--p-and", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the timed configuration (rocprofv3 runs whose summary should contain nothing else)")
    ap.add_argument("--profile-run", action="store_true",
                    help="only the timed proofs and the parity check (tools/profile_round.sh: the kernel counts of a profiled run must "
                         "be those of warmup + steps proofs)")
    ap.add_argument("--quick", action="store_true", help="fewer repetitions in the secondary records, Z64 at 10^5 MUL")
    ap.add_argument("--device-resident", action="store_true",
                    help="time rv_prove_device (openings left in HBM: round 1's headline) instead of the host-to-host rv_prove")
    ap.add_argument("--no-two-in-flight", action="store_true", help="(accepted for old command lines; implied by --no-secondary)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # RV_BENCH_BACKEND=gloo lets several ranks share one GPU (used only to exercise the N>1 code path
    # on a single-GPU box; the real multi-GPU run uses RCCL)
    backend_name = os.environ.get("RV_BENCH_BACKEND", "nccl")
    if backend_name != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend_name)

    import circuits
    import reverie_amd
    from reverie_amd import _lib
    from reverie_amd.dist import HipShardBackend, LibComm, assemble_device_parts, prove_sharded

    ctx = reverie_amd.Context(local)
    prog, wit, wc, st = circuits.layered_gf2(layers=args.layers, p_and=args.p_and)
    t0 = time.perf_counter()
    # one GPU: the circuit serves whole proofs (the RV_COMPILE_WHOLE_PROVER hint); repetition shards run fastest on the
    # neutral gate stream
    circuit = reverie_amd.Circuit(prog, wc, ctx, whole_prover=(world == 1))
    compile_s = time.perf_counter() - t0
    info = circuit.info
    backend = HipShardBackend(circuit)
    seeds = rule_seeds()
    n_and = info["gf2_muls"]
    L = _lib.lib()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    hp = HostProver(circuit, wit, [], seeds)
    dev_buf = torch.empty(max(sum(backend.single_shard_sizes()), 1), dtype=torch.uint8, device="cuda")
    last_bytes = [None]

    # N > 1: the library's own communicator (rv_comm_* / rv_prove_sharded: RCCL all-gather on the library's stream,
    # openings to rank 0 by ncclSend/ncclRecv, one D2H).  If it cannot be set up, or its first proof fails, on ANY rank,
    # all ranks fall back to the torch.distributed orchestration of reverie_amd/dist.py (RV_BENCH_COMM=torch forces it).
    lib_comm, comm_note = None, None
    if world > 1:
        ok = 1
        if os.environ.get("RV_BENCH_COMM", "lib") == "lib":
            try:
                lib_comm = LibComm(circuit)
                r = lib_comm.prove(wit, [], seeds)
                if r:
                    L.rv_free(r[0])
            except Exception as e:  # noqa: BLE001
                ok, comm_note = 0, f"library communicator failed ({e}); torch.distributed path used"
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            lib_comm = None

    def step():
        if world > 1 and lib_comm is not None:
            r = lib_comm.prove(wit, [], seeds)
            if r:
                if isinstance(last_bytes[0], tuple):
                    hp.free(last_bytes[0][0])
                last_bytes[0] = r
        elif world > 1:
            # sharded proof; rank 0 ends the step with the assembled bincode(Proof) bytes in host memory
            out = prove_sharded(backend, wit, [], seeds, device_resident=True, gather=True)
            if rank == 0:
                comm, bufs, all_lens = out
                last_bytes[0] = assemble_device_parts(comm, bufs, all_lens)
        elif args.device_resident:
            backend.prove_device(wit, [], seeds, dev_buf)
        else:
            p, n = hp.prove()
            if last_bytes[0] is not None:
                hp.free(last_bytes[0][0])
            last_bytes[0] = (p, n)

    for _ in range(args.warmup):
        step()
    # HIP events around the dominant phase (the interpreter's: generator || levels) of every timed step; the other phases are timed on
    # PHASE_PASS proofs right behind the timed region -- every event in the stream costs a proof ~5 us of idle GPU, and seven of them
    # per step were 0.5 % of the headline spent on its own measurement
    interp_only = world == 1 and not args.device_resident and not args.profile_run  # (a profiled run makes exactly warmup + steps proofs: its summaries divide by that)
    L.rv_ctx_profile(ctx.handle, 2 if interp_only else 1, 1, None)
    sync_all()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        step()
        step_marks.append(time.perf_counter())
    sync_all()
    dt = time.perf_counter() - t0
    if os.environ.get("RV_BENCH_STEP_TIMES") and rank == 0:  # (diagnostic: the K steps one by one; the timed region is unchanged)
        prev = [t0] + step_marks[:-1]
        print("step ms: " + " ".join("%.3f" % ((b - a) * 1e3) for a, b in zip(prev, step_marks)), file=sys.stderr)
    prof = _lib.Profile()
    L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof))
    PHASE_PASS = 10
    prof_all = None
    if interp_only and rank == 0:
        L.rv_ctx_profile(ctx.handle, 1, 1, None)
        for _ in range(PHASE_PASS):
            step()
        sync_all()
        prof_all = _lib.Profile()
        L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof_all))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if isinstance(last_bytes[0], tuple):
        p, n = last_bytes[0]
        last_bytes[0] = C.string_at(p, n)
        hp.free(p)

    result = None
    if rank == 0:
        phases = {n: prof.ms[i] / max(args.steps, 1) for i, n in enumerate(_lib.PHASES)}
        launches = {n: int(prof.launches[i] // max(args.steps, 1)) for i, n in enumerate(_lib.PHASES)}
        if prof_all is not None:
            # (the interpreter's phase: the timed steps' own events; the others: the PHASE_PASS proofs behind them)
            for i, n in enumerate(_lib.PHASES):
                if n != "interp":
                    phases[n] = prof_all.ms[i] / PHASE_PASS
                    launches[n] = int(prof_all.launches[i] // PHASE_PASS)
        # the roofline object is the interpreter's (the HBM-bound kernel, and the longest phase in every profile under
        # profiles/); the mask and digest phases are integer-VALU-bound, which the contract's two bounds do not describe
        dom = "interp"
        row = 256 // world  # bytes per transcript / mask row on this rank
        # algorithmic HBM bytes per proof of each phase.  The prover of a pure GF(2) circuit keeps one
        # cleartext value byte per share row instead of the corr-bit rows (MODE_PROVE_V, the interpreter's template
        # argument 2)
        vclr = os.environ.get("RV_VCLR", "1") != "0" and os.environ.get("RV_PIPELINE", "0") == "0" and row in (256, 128, 64, 32)
        per_row_corr = 1 if vclr else row // 8  # bytes of corr bits / cleartext value next to every share row touched
        n_dev_gates = info["gf2_muls"] + info["gf2_inputs"] + info["gf2_asserts"] + info["gf2_rows_written"]
        alg = {
            "masks": info["gf2_masks"] * row,  # writes every mask row once
            # 48 B gate records; operand rows in (as compiled: with the RV_COMPILE_WHOLE_PROVER hint a Mul reads up to
            # three rows per operand and fewer Xor gates exist at all); a Mul's two fresh mask rows in (its result's mask IS
            # one of them); materialised rows out; online rows and preprocessing bits out
            "interp": (n_dev_gates * 48
                       + (info["gf2_operand_rows"] + 2 * info["gf2_muls"] + info["gf2_inputs"]) * (row + per_row_corr)
                       + info["gf2_rows_written"] * (row + per_row_corr) + (info["gf2_muls"] + info["gf2_inputs"]) * per_row_corr
                       + (info["gf2_muls"] + info["gf2_inputs"] + info["gf2_asserts"]) * row + info["gf2_muls"] * row // 8),
            "hash": (info["gf2_muls"] + info["gf2_inputs"] + info["gf2_asserts"]) * row + info["gf2_muls"] * row // 8,
        }
        interp_arg = "2" if vclr else "0"
        kname = {"masks": "rv::k_aes_gf2_masks_col4", "interp": f"rv::k_interp_full<{interp_arg}, {row // 4}, {'true' if world == 1 else 'false'}>", "hash": "rv::k_b3_chunks<4>"}[dom]
        kname_masks = "rv::k_aes_gf2_masks_col4"
        ach = alg[dom] / (phases[dom] * 1e-3) / 1e9 if phases[dom] > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic.json")
        if world == 1 and args.layers == 153 and args.p_and == 0.5 and os.path.exists(pmc_path):
            pk = json.load(open(pmc_path))["kernels"]
            prefix = {"masks": "rv::k_aes_gf2_masks", "interp": f"rv::k_interp_full<{interp_arg}", "hash": "rv::k_b3_chunks<"}[dom]
            hits = [v["hbm_bytes_per_proof"] for k, v in pk.items() if k.startswith(prefix)]
            if hits:
                traffic = sum(hits) / max(launches[dom], 1)
        n_l = max(launches[dom], 1)
        # rv_prove's early-corrections path (csrc/api.hip) puts a packing kernel and a progress stamp per chunk of the corrections
        # vectors between the interpreter's level launches: their time is inside the interpreter's phase, their launches are
        # counted apart (rv_profile slot 6), and the copy-engine transfers they feed run beside the levels
        early_launches = int(prof.launches[6] // max(args.steps, 1))
        kernel_alone = None
        # round 5: the mask generator runs BESIDE the level launches (RV_OVERLAP, csrc/api.hip: its own stream, chunk by chunk), so the
        # interpreter's phase of the timed proofs holds the level launches, their waits for mask chunks and -- on the other stream --
        # the cipher's 2.57 GB of stores; the masks' own phase slot is empty
        overlapped = world == 1 and os.environ.get("RV_OVERLAP", "1") != "0" and phases["masks"] < 0.25 * phases["interp"]
        if (early_launches or overlapped) and world == 1 and not args.device_resident and not args.profile_run:
            # the same kernel alone (RV_EARLY=0: no packing kernels in the phase, no copy engine beside it; RV_OVERLAP=0: the mask
            # generator before the first level, nothing beside the levels), measured the same way right after the timed region
            os.environ["RV_EARLY"] = "0"
            os.environ["RV_OVERLAP"] = "0"
            timed_bytes, last_bytes[0] = last_bytes[0], None  # (step() frees the previous proof's buffer: keep the timed proof's bytes aside)
            try:
                n_alone = max(min(args.steps, 10), 3)
                for _ in range(2):
                    step()
                sync_all()
                L.rv_ctx_profile(ctx.handle, 1, 1, None)
                t1 = time.perf_counter()
                for _ in range(n_alone):
                    step()
                sync_all()
                dt1 = time.perf_counter() - t1
                prof1 = _lib.Profile()
                L.rv_ctx_profile(ctx.handle, 0, 0, C.byref(prof1))
                ms1 = prof1.ms[_lib.PHASES.index(dom)] / n_alone
                nl1 = max(int(prof1.launches[_lib.PHASES.index(dom)] // n_alone), 1)
                kernel_alone = {"achieved": alg[dom] / (ms1 * 1e-3) / 1e9, "frac": alg[dom] / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "phase_ms": ms1, "launches_per_proof": nl1, "avg_launch_us": ms1 * 1e3 / nl1, "ms_per_proof_host_to_host": dt1 / n_alone * 1e3,
                                "phase_ms_all": {n: prof1.ms[i] / n_alone for i, n in enumerate(_lib.PHASES)},
                                "note": f"RV_EARLY=0 RV_OVERLAP=0, {n_alone} proofs after the timed region: the mask generator runs before the first "
                                        "level, the interpreter's phase holds the level launches only and no copy engine runs beside them; the proof "
                                        "then pays the whole 50 MB device-to-host copy behind its last kernel (this is the round-4 schedule)"}
            finally:
                os.environ.pop("RV_EARLY", None)
                os.environ.pop("RV_OVERLAP", None)
            if isinstance(last_bytes[0], tuple):
                hp.free(last_bytes[0][0])
            last_bytes[0] = timed_bytes
        roofline = {
            "bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "note": "dominant phase by HIP-event time on the library's own stream (a proof's phases run back to back on one "
                    "stream); achieved = algorithmic HBM bytes per launch (DESIGN.md §4) / average launch duration, both from the "
                    "timed run itself; traffic = PMC-measured HBM bytes per launch of that kernel "
                    f"(profiles/{PROFILE_TAG}_pmc_traffic.json). The mask and hash phases are integer-VALU-bound (bitsliced AES, "
                    "BLAKE3): no MFMA on this path.",
            "launches_per_proof": launches[dom], "avg_launch_us": phases[dom] * 1e3 / n_l,
            "algorithmic_bytes_per_launch": alg[dom] / n_l,
            "kernel_alone": kernel_alone,
            "concurrent": {"kernel": kname_masks, "stream": "the context's mask stream, chunk by chunk beside the level launches (RV_OVERLAP)",
                           "algorithmic_bytes_per_proof": int(alg["masks"]),
                           "phase_achieved_incl_masks": (alg[dom] + alg["masks"]) / (phases[dom] * 1e-3) / 1e9,
                           "phase_frac_incl_masks": (alg[dom] + alg["masks"]) / (phases[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "the interpreter's phase (HIP events on the main stream) spans the level launches, their waits for mask chunks "
                                   "and, beside them, the lane-distributed mask generator with its own 2.57 GB of row stores: achieved / frac above "
                                   "price the LEVEL kernel's bytes over that whole span (it runs at about 0.75 of its stand-alone rate beside the "
                                   "cipher), phase_*_incl_masks price both kernels' bytes over it; kernel_alone is the level kernel with nothing beside it"} if overlapped else None,
            "early_corrections": {"kernels_in_phase": early_launches, "kernel_alone": kernel_alone,
                                  "note": "the timed proofs take rv_prove's early-corrections path: the interpreter's phase also holds this many small kernels "
                                          "(k_pack_corr_all ~25 us + k_publish ~5 us per chunk of the corrections vectors, not counted in launches_per_proof) while the "
                                          "copy engine moves 160 MB to the host beside the levels; achieved / frac above are "
                                          "the whole phase over the level launches, kernel_alone is the same kernel without the path"} if early_launches else None,
            "phase_ms": phases, "phase_launches": launches, "algorithmic_bytes_per_proof": {k: int(v) for k, v in alg.items()},
            "gpu_ms_per_proof": sum(phases.values()),
            "phase_ms_note": ("interp: HIP events around the phase in every timed step; the other phases: the same events on %d proofs right behind the "
                              "timed region (a stream event costs a proof ~5 us of idle GPU: the timed steps carry two, not seven)" % PHASE_PASS) if prof_all is not None else None,
        }
        # SURVEY 8(d) names integer VALU throughput as the binding roofline of the mask and hash phases: instruction-issue
        # fraction = wavefront-level VALU instructions per proof (SQ_INSTS_VALU of the profiled run, a property of the circuit)
        # / this run's phase time / the chip's issue peak (1024 SIMDs x 1/2 instruction per clock x 2.4 GHz)
        sq_path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_sq_counters.json")
        if world == 1 and args.layers == 153 and args.p_and == 0.5 and os.path.exists(sq_path):
            sk = json.load(open(sq_path))["kernels"]
            peak_issue = 1024 * 0.5 * 2.4e9
            valu = {"peak": peak_issue, "unit": "wavefront VALU instructions/s", "kernels": {}}
            mask_pf, interp_pf = ("rv::k_aes_gf2_masks_col4",), ("rv::k_interp_full<2,",)  # (the timed proofs' kernels: whole proofs run the interpreter as MODE_PROVE_V)
            # (the prover's hash launches; k_b3_chunks_pair_uni<1> / <2> are the one verification of the profiled run)
            hash_pf = ("rv::k_b3_chunks_uni", "rv::k_b3_chunks_bits_uni", "rv::k_b3_chunks_pair_uni<0>", "rv::k_b3_reduce_pair", "rv::k_b3_tree_tail_pair_big")
            # overlapped: the cipher and the level launches share the interpreter's phase (and the SIMDs): one entry for both
            for phase, prefixes, ms in ((("masks+interp", mask_pf + interp_pf, phases["interp"]),) if overlapped else (("masks", mask_pf, phases["masks"]),)) + (("hash", hash_pf, phases["hash"]),):
                insts = sum(v.get("SQ_INSTS_VALU_per_proof", 0.0) for k, v in sk.items() if k.startswith(prefixes))
                if insts and ms > 0:
                    valu["kernels"][phase] = {"kernels": [k for k in sk if k.startswith(prefixes)], "valu_insts_per_proof": insts,
                                              "phase_ms": ms, "issue_frac": insts / (ms * 1e-3) / peak_issue}
            valu["note"] = ("half of the BLAKE3 / bitsliced-AES instruction mix are 3-source VOP3 (v_bitop3, v_perm, v_add3, v_alignbit) "
                            "that issue at a quarter of a wavefront per clock, not half: tools/mb/valu_mb.hip; DESIGN.md section 4")
            # the ceiling the instruction mix itself allows (tools/valu_mix.py over the shipped ISA: N / (N_full + 2 N_half))
            mix_path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_valu_mix.json")
            mix = json.load(open(mix_path))["kernels"] if os.path.exists(mix_path) else {}
            valu["mix"] = {k: {"half_rate_share": v["half_rate_share"], "issue_ceiling": v["issue_ceiling"]} for k, v in mix.items()}
            roofline["valu"] = valu
            # the line's roofline names the binding resource of the dominant phase.  Beside each other the cipher and
            # the level launches are paced by integer-VALU issue (the cipher takes the longer share of the phase and is at the ceiling
            # of its mix when alone); the level kernel's HBM view stays as roofline.hbm
            ph = valu["kernels"].get("masks+interp")
            if overlapped and ph:
                c4 = "rv::k_aes_gf2_masks_col4"
                c4_insts = sum(v.get("SQ_INSTS_VALU_per_proof", 0.0) for k, v in sk.items() if k.startswith(mask_pf))
                ceil_c4 = mix.get(c4, {}).get("issue_ceiling")
                ceil_lv = mix.get(kname, {}).get("issue_ceiling")
                lv_insts = ph["valu_insts_per_proof"] - c4_insts
                # (the phase's ceiling: both kernels' instructions at their own mixes' rates)
                ceil_phase = (ph["valu_insts_per_proof"] / (c4_insts / ceil_c4 + lv_insts / ceil_lv)) if ceil_c4 and ceil_lv else None
                c4_traffic = None
                if world == 1 and args.layers == 153 and args.p_and == 0.5 and os.path.exists(pmc_path):
                    # (the generator's chunk launches are counted by the profiled run itself: they run on the mask stream, outside the phase counters)
                    hit = json.load(open(pmc_path))["kernels"].get(c4)
                    c4_traffic = hit["hbm_bytes_per_proof"] / max(hit["launches_per_proof"], 1.0) if hit else None
                hbm_view = roofline
                roofline = {
                    "bound": "valu", "kernel": c4, "achieved": ph["valu_insts_per_proof"] / (ph["phase_ms"] * 1e-3) / 1e9, "peak": peak_issue / 1e9,
                    "unit": "G wavefront VALU instructions/s", "frac": ph["issue_frac"], "ceiling_frac": ceil_phase,
                    "ceiling_frac_kernel": ceil_c4, "traffic": c4_traffic,
                    "note": "the proof's dominant phase = the lane-distributed mask generator (bitsliced AES-128-CTR, the kernel with the largest share of "
                            "the trace) on its own stream BESIDE the interpreter's level launches; both are paced by integer-VALU issue: achieved = "
                            "wavefront VALU instructions of both kernels per proof (SQ_INSTS_VALU, profiles/%s_sq_counters.json) / the phase's HIP-event "
                            "time in THIS run; peak = 1024 SIMDs x 1/2 instruction per clock x 2.4 GHz; ceiling_frac = what the instruction mix allows "
                            "(61 %% of the cipher's instructions are 3-source VOP3 at half rate: profiles/%s_valu_mix.json) -- the cipher alone reaches it, "
                            "the pair does not because the level launches are latency-bound (roofline.hbm.kernel_alone); no MFMA on this path.  "
                            "traffic = PMC HBM bytes per generator launch (2.70 GB per proof in 17 - 18 chunk launches: its row stores, nothing re-read)" % (PROFILE_TAG, PROFILE_TAG),
                    "phase_ms": phases, "phase_launches": launches, "gpu_ms_per_proof": sum(phases.values()), "phase_ms_note": hbm_view.get("phase_ms_note"),
                    "valu": valu, "hbm": hbm_view,
                }
                # (kept at the top level for the tools that read them)
                for k in ("kernel_alone", "concurrent", "early_corrections", "launches_per_proof", "avg_launch_us", "algorithmic_bytes_per_proof"):
                    roofline[k] = hbm_view.get(k)
        boundary = (("rv_prove_sharded (library communicator: RCCL all-gather of digests on the library's stream, ncclSend/ncclRecv of "
                     "the openings); rank 0 ends with bincode(Proof) bytes in host memory" if lib_comm is not None else
                     "sharded rv_shard_* + torch.distributed all-gather; rank 0 ends with bincode(Proof) bytes in host memory"
                     + (f" [{comm_note}]" if comm_note else "")) if world > 1 else
                    "rv_prove_device: openings left in HBM (NOT the SURVEY 8(d) boundary)" if args.device_resident else
                    "rv_prove: witness bytes on the host -> bincode(Proof) bytes on the host (page-locked), D2H of the proof included")
        result = {
            "metric": "prover AND-gates/sec on GF(2) Bristol circuit; 1/2/4/8-GPU; proof bit-exact",
            "value": n_and * args.steps / dt, "unit": "AND gates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"synthetic layered AND/XOR GF(2) circuit, {st['gates']} gates ({st['and']} AND), "
                                   f"{st['inputs']} inputs, {args.layers} layers x 65536, p_and={args.p_and}, 256 reps x 8 players, 40 online",
                       "boundary": boundary, "parallelism": f"reps/{world}", "levels": info["levels"], "compile_s": compile_s,
                       "compile_hint": "RV_COMPILE_WHOLE_PROVER" if world == 1 else "none",
                       "gate_stream_upload_ms": info["upload_us"] / 1e3, "gate_stream_bytes": info["device_bytes"]},
            "roofline": roofline,
        }

    # ---- parity gate and the secondary records, outside the timed region (rank 0; the other ranks wait at the barrier)
    if rank == 0:
        if world == 1 and args.device_resident:
            last = reverie_amd.Proof.new(circuit, wit, [], seeds=seeds)
        else:
            last = reverie_amd.Proof(last_bytes[0])
        # (a profiled run verifies nothing: its kernel summaries are divided by the number of proofs it makes)
        parity = {"last_timed_proof_verifies_strict": None if args.profile_run else bool(last.verify(circuit)), "proof_bytes": len(last)}
        if world > 1:
            single = reverie_amd.Proof.new(circuit, wit, [], seeds=seeds)
            parity["sharded_proof_equals_single_gpu_proof"] = bytes(single) == bytes(last)
        result["parity"] = parity
        if world == 1 and not args.profile_run:
            # the first proof of a circuit the library has not seen: rv_prove_ops from the raw op list (the reference's
            # Proof::new walks the raw ops, proof/mod.rs:150-152) = compile on the host threads + upload + prove
            fp = []
            os.environ["RV_OPS_CACHE"] = "0"  # cold: nothing kept between the calls (what the FIRST Proof::new on a circuit costs)
            try:
                for _ in range(3):
                    ctx.sync()
                    tf = time.perf_counter()
                    first = reverie_amd.Proof.new(prog, wit, [], wc, seeds=seeds, ctx=ctx)
                    fp.append((time.perf_counter() - tf) * 1e3)
                vo = time.perf_counter()
                ok_ops = bool(first.verify(prog, wc, ctx=ctx))
                vo = (time.perf_counter() - vo) * 1e3
            finally:
                os.environ.pop("RV_OPS_CACHE", None)
            # ... and what every later Proof::new on the same op list costs: the context keeps the compiled circuit by content
            # (csrc/api.hip ops_cache_get: 128 bits hashed from the 240 MB op array on host threads, then rv_prove)
            rp = []
            again = reverie_amd.Proof.new(prog, wit, [], wc, seeds=seeds, ctx=ctx)  # (fills the cache)
            for _ in range(7):
                tf = time.perf_counter()
                again = reverie_amd.Proof.new(prog, wit, [], wc, seeds=seeds, ctx=ctx)
                rp.append((time.perf_counter() - tf) * 1e3)
            rp.sort()
            again_ok = bytes(again) == bytes(last)
            del again
            _lib.lib().rv_ctx_ops_cache_clear(ctx.handle)
            result["first_proof"] = {
                "first_proof_ms": sorted(fp)[1], "runs_ms": fp, "and_per_s": n_and / (sorted(fp)[1] * 1e-3),
                "repeat_proof_ms": rp[len(rp) // 2], "repeat_ms_min_max": [rp[0], rp[-1]], "repeat_and_per_s": n_and / (rp[len(rp) // 2] * 1e-3),
                "repeat_bit_exact_vs_timed_proof": again_ok,
                "bit_exact_vs_timed_proof": bytes(first) == bytes(last), "verify_ops_ms": vo, "verify_ops_ok": ok_ops,
                "compile_threads": int(os.environ.get("RV_COMPILE_THREADS", "0")) or min(16, os.cpu_count() or 1),
                "note": "rv_prove_ops: raw rv_op list + witness bytes on the host -> bincode(Proof) bytes on the host; the gate stream is "
                        "levelised by the parallel compiler (csrc/compile_par.cpp), uploaded and proved once, then released "
                        "(median of 3 on the warm context, RV_OPS_CACHE=0); rv_verify_ops likewise (one run); repeat_proof_ms = the same call on "
                        "the same op list once the context has seen it (content-addressed cache: hash + proof, the Python mirror's array checks included)"}
        if world == 1 and not args.no_secondary:
            # verifier: rv_verify (strict) from host proof bytes, second call timed; the verifying party
            # compiles the circuit for itself (no prover hint)
            host_proof = reverie_amd.Proof.new(circuit, wit, [], seeds=seeds)
            vcirc = reverie_amd.Circuit(prog, wc, ctx)
            host_proof.verify(vcirc)
            tps, okp = [], True
            for _ in range(5):
                tp = time.perf_counter()
                okp = bool(host_proof.verify(vcirc)) and okp
                tps.append(time.perf_counter() - tp)
            tp = sorted(tps)[2]
            vcirc.close()
            result["verifier"] = {"value": n_and / tp, "unit": "AND gates/s", "ms": tp * 1e3, "ms_min_max": [min(tps) * 1e3, max(tps) * 1e3], "strict_ok": okp,
                                  "note": "rv_verify (strict), host proof bytes (page-locked, as rv_prove returned them) in, median of 5 calls after a first one; "
                                          "circuit compiled by rv_circuit_compile (the prover's has the RV_COMPILE_WHOLE_PROVER hint)"}
            parity["rv_prove_is_deterministic"] = bytes(host_proof) == bytes(last)
            del host_proof
            # what a rank of a sharded proof costs (the strong-scaling ceiling from a driver record)
            try:
                result["shard_ms"] = shard_times(ctx, prog, wc, wit, seeds, n_and)
            except Exception as e:  # noqa: BLE001
                result["shard_ms"] = {"error": repr(e)}
            # the same proof left in HBM (round 1's headline) -- or, with --device-resident, the host-to-host one
            n2 = max(args.steps // 2, 5)
            if args.device_resident:
                hp.run(2)
                d2, data2 = hp.run(n2)
                result["host_to_host"] = {"value": n_and * n2 / d2, "unit": "AND gates/s", "ms_per_proof": d2 / n2 * 1e3,
                                          "bit_exact_vs_timed_proof": data2 == bytes(last)}
            else:
                # (shard_times() above created and destroyed shards of four sizes: two warm-ups, every proof timed on its own, median)
                for _ in range(2):
                    backend.prove_device(wit, [], seeds, dev_buf)
                sync_all()
                t2s = []
                for _ in range(n2):
                    t0 = time.perf_counter()
                    comm, omit, lens = backend.prove_device(wit, [], seeds, dev_buf)
                    sync_all()
                    t2s.append(time.perf_counter() - t0)
                t2s.sort()
                d2m = t2s[len(t2s) // 2]
                dev_proof = assemble_device_parts(comm, [dev_buf], [lens])
                result["device_resident"] = {"value": n_and / d2m, "unit": "AND gates/s", "ms_per_proof": d2m * 1e3,
                                             "ms_min_max": [t2s[0] * 1e3, t2s[-1] * 1e3], "proofs": n2,
                                             "bit_exact_vs_timed_proof": dev_proof == bytes(last),
                                             "note": "rv_prove_device: the proof's openings stay in HBM (no D2H, no framing); median of the "
                                                     "proofs timed one by one after two warm-ups"}
            # rv_prove_batch, host to host: several proofs of the circuit per call
            B = 8
            rng = np.random.default_rng(7)
            bs = rng.integers(0, 256, (B, 256, 16), dtype=np.uint8)
            bs[0] = seeds
            bw = np.tile(np.asarray(wit, np.uint8), (B, 1))
            proofs = None
            for _ in range(2):  # (held like in the timed loop: the second page-locked slab is mapped here, not inside a timed call)
                proofs = reverie_amd.Proof.new_batch(circuit, bw, seeds=bs)
            d3s = []
            for _ in range(12):
                t0 = time.perf_counter()
                proofs = reverie_amd.Proof.new_batch(circuit, bw, seeds=bs)
                d3s.append(time.perf_counter() - t0)
            d3 = statistics.median(d3s)
            result["prove_batch_host"] = {"value": n_and * B / d3, "unit": "AND gates/s", "ms_per_proof": d3 / B * 1e3, "proofs_per_call": B,
                                          "first_proof_bit_exact_vs_timed_proof": bytes(proofs[0]) == bytes(last),
                                          "last_proof_verifies_strict": bool(proofs[-1].verify(circuit)),
                                          "ms_per_proof_min_max": [min(d3s) / B * 1e3, max(d3s) / B * 1e3],
                                          "note": "rv_prove_batch: witness bytes on the host -> B proofs' bytes on the host, one call (median of 12 calls after two warm-ups; worker streams on hardware queues of their own, csrc/batch.inc)"}
            del proofs
    if world > 1:
        # informational, outside the timed region: the same N GPUs proving N INDEPENDENT statements, one whole proof
        # (256 repetitions) per rank and no collective -- weak scaling, what a proving service with a queue of
        # statements would run.  `value` above stays the north-star's sharded single proof (strong scaling).
        n_ind = max(args.steps // 2, 3)
        hp.run(1)
        sync_all()
        ti = time.perf_counter()
        hp.run(n_ind)
        sync_all()
        ti = torch.tensor([time.perf_counter() - ti], dtype=torch.float64, device="cuda")
        dist.all_reduce(ti, op=dist.ReduceOp.MAX)
        if rank == 0:
            result["independent_proofs"] = {"value": n_and * n_ind * world / float(ti.item()), "unit": "AND gates/s", "scaling": "weak",
                                            "ms_per_proof_per_gpu": float(ti.item()) / n_ind * 1e3, "proofs": n_ind * world,
                                            "note": "one whole rv_prove (host to host) per GPU at a time, no collective"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        what = f"the whole timed workload ({st['gates']} gates, {st['and']} AND), same seeds"
        base, cproof = cpu_baseline(prog, wit, [], wc, seeds, st["and"], "AND gates/s", what, runs=5)
        result["cpu_baseline"] = base
        # the oracle proved the very workload that was timed, with the same seeds: the last timed proof must be its
        # proof, byte for byte
        result["parity"]["timed_proof_bit_exact_vs_cpu"] = bytes(last) == cproof
        del cproof
    if rank == 0 and world == 1 and not args.no_secondary:
        # the all-AND variant (the north-star's "10^7-gate" phrasing read as 10^7 AND gates)
        circuit.close()
        aprog, awit, awc, ast = circuits.layered_gf2(layers=args.layers, p_and=1.0)
        acirc = reverie_amd.Circuit(aprog, awc, ctx)
        ahp = HostProver(acirc, awit, [], seeds)
        na = 7
        tas, adata = ahp.run_each(na, warm=2)
        da = tas[len(tas) // 2]
        rec = {"value": ast["and"] / da, "unit": "AND gates/s", "ms_per_proof": da * 1e3, "ms_min_max": [tas[0] * 1e3, tas[-1] * 1e3], "and_gates": ast["and"],
               "proof_bytes": len(adata), "verifies_strict": bool(reverie_amd.Proof(adata).verify(acirc)),
               "note": "rv_prove host to host on the p_and = 1 variant of the workload (median of 7 proofs timed one by one after two warm-ups)"}
        if not args.no_cpu_baseline:
            import oracle_lib

            t0 = time.perf_counter()
            want = oracle_lib.prove(aprog, awit, [], awc, seeds, threads=32)
            rec["cpu_oracle_and_per_s"] = ast["and"] / (time.perf_counter() - t0)
            rec["bit_exact_vs_cpu"] = adata == want
            del want
        result["all_and"] = rec
        acirc.close()
        del aprog, adata
        result["secondary"] = secondary_records(ctx, seeds, args.quick)
        # the streaming prover / verifier in a process of their own: what a caller of the library sees (this process holds 20+ GB of
        # host arrays and several thread pools by now, and the chunks' host-side compile runs ~1.5x slower in it)
        try:
            import subprocess

            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_bench.py"), "--bench-record"], capture_output=True, text=True, timeout=300)
            lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            result["streaming"] = json.loads(lines[-1])
        except Exception:  # noqa: BLE001
            try:
                from tools.stream_bench import streaming_record

                result["streaming"] = streaming_record(ctx, prog, wit, wc, st, seeds, bytes(last))
            except ImportError:
                pass
    if rank == 0:
        flags = [v for k, v in result["parity"].items() if k != "proof_bytes" and v is not None]  # (None: not checked in a profiled run)
        if not all(flags):
            result["value"] = 0.0
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
