"""Multi-GPU proving: repetitions shard across ranks, one process per GPU.

The reference proves all 256 repetitions in one process (32 rayon tasks,
/root/reference/src/proof/mod.rs:127-157) and has exactly one point where every
repetition's result meets: `combine_hashes` over the 256 per-repetition digests before the
Fiat-Shamir challenge (proof/mod.rs:160-172).  Here rank g owns repetitions
[g*256/G, (g+1)*256/G); the gate stream and witness are replicated; the ONE data-path
collective is an all-gather of 32-byte digests (8 KiB total — RCCL over xGMI with the
`nccl` backend, gloo in CPU tests).  Every rank then derives the same challenge, opens its
own repetitions, and rank 0 concatenates the openings in ascending repetition order
(proof/mod.rs:200-221) — a point-to-point collection of output, not a reduction.

`backend` supplies the per-shard compute.  The product backend is HipShardBackend (C-ABI,
GPU).  Tests inject an oracle-backed stand-in to exercise this orchestration on CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .ops import TOTAL_REPS
from .proof import Circuit, _ptr, challenge, combine_digests


def shard_range(rank: int, world: int) -> Tuple[int, int]:
    if TOTAL_REPS % (8 * world):
        raise ValueError("world size must divide 32 packed groups")
    n = TOTAL_REPS // world
    return rank * n, n


class HipShardBackend:
    """rv_shard_commit / rv_shard_open over the C-ABI on this rank's GPU."""

    device_type = "cuda"

    def __init__(self, circuit: Circuit):
        self.circuit = circuit
        self._sizes = None      # record sizes of the circuit (constant)
        self._gather_bufs = {}  # (count, device) -> (own digests, all digests): reused from proof to proof

    def _record_sizes(self):
        if self._sizes is None:
            self._sizes = self.circuit.record_sizes()
        return self._sizes

    def commit(self, wit_gf2, wit_z64, seeds, rep_begin, rep_count):
        g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
        z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(rep_count, 16)
        h = C.c_void_p()
        _lib.check(_lib.lib().rv_shard_commit(self.circuit.ctx.handle, self.circuit.handle, _ptr(g), C.c_size_t(len(g)),
                                              _ptr(z), C.c_size_t(len(z)), _ptr(s), C.c_uint32(rep_begin),
                                              C.c_uint32(rep_count), C.byref(h)))
        return (h, rep_count)

    def digests(self, shard) -> np.ndarray:
        h, n = shard
        out = np.zeros((n, 32), np.uint8)
        _lib.check(_lib.lib().rv_shard_digests(h, _ptr(out)))
        return out

    def digests_into(self, shard, tensor):
        """device-to-device into a torch CUDA tensor (the all-gather input)"""
        _lib.check(_lib.lib().rv_shard_digests_to_device(shard[0], C.c_void_p(tensor.data_ptr())))

    def all_open_sizes(self, omit: np.ndarray, world: int) -> List[List[int]]:
        """blob sizes of every rank's shard for this challenge — a pure function of the challenge and the
        circuit, so no collective is needed to learn them"""
        sz2, sz64 = self._record_sizes()
        out = []
        for r in range(world):
            b, n = shard_range(r, world)
            n_on = int((omit[b:b + n] < 8).sum())
            out.append([n_on * sz2, (n - n_on) * 48, n_on * sz64, (n - n_on) * 48])
        return out

    def open_sizes(self, shard, omit: np.ndarray) -> List[int]:
        lens = (C.c_size_t * 4)()
        _lib.check(_lib.lib().rv_shard_open_size(shard[0], _ptr(omit), lens))
        return [int(x) for x in lens]

    def open_into(self, shard, omit: np.ndarray, tensor) -> List[int]:
        lens = (C.c_size_t * 4)()
        _lib.check(_lib.lib().rv_shard_open_into(shard[0], _ptr(omit), C.c_void_p(tensor.data_ptr()), lens))
        return [int(x) for x in lens]

    def open_self(self, shard, tensor):
        """Single-shard proofs: commitment, challenge and openings on the device (rv_shard_open_self).
        -> (comm bytes, omit[256], lens[4])"""
        lens = (C.c_size_t * 4)()
        comm = np.zeros(32, np.uint8)
        omit = np.zeros(TOTAL_REPS, np.uint8)
        _lib.check(_lib.lib().rv_shard_open_self(shard[0], C.c_void_p(tensor.data_ptr()), _ptr(comm), _ptr(omit), lens))
        return comm.tobytes(), omit, [int(x) for x in lens]

    def gathered_capacity(self, count: int) -> int:
        """bytes open_gathered may write for a shard of `count` repetitions (every one of up to 40 of them opened)"""
        sz2, sz64 = self._record_sizes()
        return min(40, count) * (sz2 + sz64) + 2 * count * 48

    def open_gathered(self, shard, all_digests, tensor):
        """rv_shard_open_gathered: challenge and this shard's openings on the device from the gathered digests
        (`all_digests`: device tensor of 256 x 32 bytes) -> (comm bytes, omit[256], lens[4])"""
        lens = (C.c_size_t * 4)()
        comm = np.zeros(32, np.uint8)
        omit = np.zeros(TOTAL_REPS, np.uint8)
        _lib.check(_lib.lib().rv_shard_open_gathered(shard[0], C.c_void_p(all_digests.data_ptr()), C.c_void_p(tensor.data_ptr()),
                                                     _ptr(comm), _ptr(omit), lens))
        return comm.tobytes(), omit, [int(x) for x in lens]

    def single_shard_sizes(self) -> List[int]:
        """blob sizes of a proof whose one shard holds all 256 repetitions (40 opened, 216 not)"""
        sz2, sz64 = self._record_sizes()
        return [40 * sz2, 216 * 48, 40 * sz64, 216 * 48]

    def prove_device(self, wit_gf2, wit_z64, seeds, tensor):
        """rv_prove_device: the whole prover with one host synchronisation, openings into `tensor` (device)
        -> (comm bytes, omit[256], lens[4])"""
        g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
        z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(TOTAL_REPS, 16)
        lens = (C.c_size_t * 4)()
        comm = np.zeros(32, np.uint8)
        omit = np.zeros(TOTAL_REPS, np.uint8)
        _lib.check(_lib.lib().rv_prove_device(self.circuit.ctx.handle, self.circuit.handle, _ptr(g), C.c_size_t(len(g)), _ptr(z),
                                              C.c_size_t(len(z)), _ptr(s), C.c_void_p(tensor.data_ptr()), _ptr(comm), _ptr(omit), lens))
        return comm.tobytes(), omit, [int(x) for x in lens]

    def open(self, shard, omit: np.ndarray):
        """-> (blob bytes [gf2_on | gf2_pre | z64_on | z64_pre], lens[4], n_online, n_pre)"""
        parts = _lib.ShardParts()
        _lib.check(_lib.lib().rv_shard_open(shard[0], _ptr(omit), C.byref(parts)))
        lens = [parts.gf2_online_len, parts.gf2_pre_len, parts.z64_online_len, parts.z64_pre_len]
        ptrs = [parts.gf2_online, parts.gf2_pre, parts.z64_online, parts.z64_pre]
        blob = b"".join(C.string_at(p, n) for p, n in zip(ptrs, lens))
        for p in ptrs:
            _lib.lib().rv_free(C.c_void_p(p))
        return blob, lens, int(parts.n_online), int(parts.n_pre)

    def destroy(self, shard):
        _lib.lib().rv_shard_destroy(shard[0])


def assemble_device_parts(comm: bytes, bufs, all_lens) -> bytes:
    """bincode(Proof) from the (tensor, lens) pairs prove_sharded(device_resident=True) returns on rank 0"""
    return assemble(comm, [(bytes(b.cpu().numpy().tobytes()[:sum(l)]), l) for b, l in zip(bufs, all_lens)])


def assemble(comm: bytes, parts: List[Tuple[bytes, List[int]]]) -> bytes:
    """bincode(Proof) from per-shard blobs ordered by rep_begin."""
    out = [comm]
    n_on = 0
    n_pre = 0
    split = []
    for blob, lens in parts:
        o = 0
        pieces = []
        for n in lens:
            pieces.append(blob[o:o + n])
            o += n
        split.append(pieces)
    # records are fixed-size within a section, so the counts follow from the challenge; the
    # caller passes them through `lens`-derived sections only, counts are 40 / 216 overall
    for dom in (0, 2):
        out.append((40).to_bytes(8, "little"))
        out.extend(p[dom] for p in split)
        out.append((216).to_bytes(8, "little"))
        out.extend(p[dom + 1] for p in split)
    del n_on, n_pre
    return b"".join(out)


def prove_sharded(backend, wit_gf2, wit_z64, seeds, group=None, device_resident: bool = False, gather: bool = True):
    """One proof over all ranks of `group`.  Returns bincode(Proof) bytes on rank 0 (None on
    other ranks); with device_resident=True returns the openings left in HBM instead
    (bench.py: no PCIe copy inside the timed region): (comm, [tensor per rank], [lens per rank]) on rank 0.
    gather=False (device_resident only) leaves every rank's openings in that rank's HBM — the sharded
    counterpart of the single-GPU device-resident proof — and returns (comm, [own tensor], [own lens]) on
    every rank."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1

    def g_rank(r: int) -> int:
        # `rank` / `r` are ranks INSIDE `group`; the point-to-point calls and gather_object(dst=) take global ranks
        return dist.get_global_rank(group, r) if (group is not None and world > 1) else r

    begin, count = shard_range(rank, world)
    seeds = np.asarray(seeds, dtype=np.uint8).reshape(TOTAL_REPS, 16)
    if world == 1 and device_resident and backend.device_type == "cuda" and hasattr(backend, "prove_device"):
        # one shard holds every repetition: commitment, Fiat-Shamir and openings stay on the device and the host
        # waits for it once (rv_prove_device)
        lens = backend.single_shard_sizes()
        buf = torch.empty(max(sum(lens), 1), dtype=torch.uint8, device="cuda")
        comm, _, lens = backend.prove_device(wit_gf2, wit_z64, seeds, buf)
        return comm, [buf], [lens]
    shard = backend.commit(wit_gf2, wit_z64, seeds[begin:begin + count], begin, count)
    try:
        # ---- the one collective: all-gather of per-repetition digests
        if world == 1:
            h = backend.digests(shard)
            on_gpu = False
        else:
            # RCCL ("nccl") moves device tensors over xGMI; with gloo (CPU tests, or several ranks sharing one
            # GPU) the 8 KiB travel through host memory instead
            # RV_DIST_DEVICE_PATH=1 takes the device-tensor branch with any backend that accepts CUDA tensors (the
            # tests use it with gloo, several ranks on one GPU, to exercise the RCCL code path without RCCL)
            on_gpu = backend.device_type == "cuda" and (dist.get_backend(group) == "nccl" or os.environ.get("RV_DIST_DEVICE_PATH") == "1")
            dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
            cache = getattr(backend, "_gather_bufs", None)
            key = (count, str(dev))
            if cache is not None and key in cache:
                mine, allh = cache[key]  # both are consumed before this function returns
            else:
                mine = torch.empty(count * 32, dtype=torch.uint8, device=dev)
                allh = torch.empty(TOTAL_REPS * 32, dtype=torch.uint8, device=dev)
                if cache is not None:
                    cache[key] = (mine, allh)
            if on_gpu:
                backend.digests_into(shard, mine)
            else:
                mine.copy_(torch.from_numpy(backend.digests(shard).reshape(-1)))
            dist.all_gather_into_tensor(allh, mine, group=group)
            if on_gpu and device_resident and hasattr(backend, "open_gathered"):
                # the gathered digests are on this GPU: commitment, challenge and this shard's openings without a
                # host round trip (rv_shard_open_gathered); the buffer holds the worst case, `lens` what was written
                torch.cuda.current_stream().synchronize()  # the collective ran on torch's stream, the library has its own
                buf = torch.empty(max(backend.gathered_capacity(count), 1), dtype=torch.uint8, device="cuda")
                comm, omit, lens = backend.open_gathered(shard, allh, buf)
                if not gather:
                    return comm, [buf], [lens]
                all_lens = backend.all_open_sizes(omit, world)
                assert all_lens[rank] == lens
                if rank == 0:
                    bufs = [buf] + [torch.empty(max(sum(l), 1), dtype=torch.uint8, device="cuda") for l in all_lens[1:]]
                    reqs = [dist.irecv(bufs[r], src=g_rank(r), group=group) for r in range(1, world)]
                    for q in reqs:
                        q.wait()
                    return comm, bufs, all_lens
                dist.send(buf[:max(sum(lens), 1)].contiguous(), dst=g_rank(0), group=group)
                return comm, None, None
            h = allh.cpu().numpy().reshape(TOTAL_REPS, 32)
        comm = combine_digests(h)  # every rank derives the same challenge
        omit = challenge(comm)
        # ---- open own repetitions; rank 0 collects in rank (= repetition) order
        if device_resident and backend.device_type == "cuda":
            lens = backend.open_sizes(shard, omit)
            buf = torch.empty(max(sum(lens), 1), dtype=torch.uint8, device="cuda")
            backend.open_into(shard, omit, buf)
            if world > 1 and not gather:
                return comm, [buf], [lens]
            if world > 1:
                all_lens = backend.all_open_sizes(omit, world)
                assert all_lens[rank] == lens
                if not on_gpu:  # gloo has no device point-to-point: stage through the host
                    if rank == 0:
                        bufs = [buf.cpu()] + [torch.empty(max(sum(l), 1), dtype=torch.uint8) for l in all_lens[1:]]
                        for r in range(1, world):
                            dist.recv(bufs[r], src=g_rank(r), group=group)
                        return comm, bufs, all_lens
                    dist.send(buf.cpu(), dst=g_rank(0), group=group)
                    return comm, None, None
                if rank == 0:
                    bufs = [buf] + [torch.empty(max(sum(l), 1), dtype=torch.uint8, device="cuda") for l in all_lens[1:]]
                    reqs = [dist.irecv(bufs[r], src=g_rank(r), group=group) for r in range(1, world)]
                    for q in reqs:
                        q.wait()
                    return comm, bufs, all_lens
                dist.send(buf, dst=g_rank(0), group=group)
                return comm, None, None
            return comm, [buf], [lens]
        blob, lens, _, _ = backend.open(shard, omit)
        if world == 1:
            return assemble(comm, [(blob, lens)])
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((blob, lens), gathered, dst=g_rank(0), group=group)
        return assemble(comm, gathered) if rank == 0 else None
    finally:
        backend.destroy(shard)


class LibComm:
    """This rank's place in a group of GPUs proving together INSIDE the library (rv_comm_*, rv_prove_sharded): the
    library owns the RCCL communicator, the digests' all-gather runs on its own stream and the openings go to rank 0
    with ncclSend/ncclRecv.  torch.distributed (any backend) is only used once, to hand rank 0's communicator id to
    the other ranks; with world == 1 nothing is exchanged."""

    def __init__(self, circuit: Circuit, group=None):
        import torch.distributed as dist

        self.circuit = circuit
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        uid = np.zeros(128, np.uint8)
        if self.rank == 0:
            _lib.check(_lib.lib().rv_comm_unique_id(_ptr(uid)))
        if self.world > 1:
            box = [uid.tobytes()]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = np.frombuffer(box[0], np.uint8).copy()
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().rv_comm_create(circuit.ctx.handle, C.c_int(self.world), C.c_int(self.rank), _ptr(uid), C.byref(self.handle)))

    def prove(self, wit_gf2, wit_z64, seeds):
        """rv_prove_sharded (collective).  -> (pointer, length) of the library's proof buffer on rank 0 (hand it to
        Proof(_owned=...) or rv_free), None on the other ranks"""
        g = np.ascontiguousarray(np.asarray(wit_gf2, dtype=np.uint8))
        z = np.ascontiguousarray(np.asarray(wit_z64, dtype=np.uint64))
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(TOTAL_REPS, 16)
        out, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().rv_prove_sharded(self.handle, self.circuit.handle, _ptr(g), C.c_size_t(len(g)), _ptr(z), C.c_size_t(len(z)),
                                               _ptr(s), C.byref(out), C.byref(n)))
        return (C.c_void_p(out.value), n.value) if out.value else None

    def close(self):
        if self.handle:
            if self.circuit.ctx.handle:
                _lib.lib().rv_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
