"""Oracle-backed stand-in for reverie_amd.dist.HipShardBackend — TEST INFRASTRUCTURE.

Lets the multi-rank orchestration (sharding, digest all-gather, challenge, ordered
collection, assembly) run on CPU with gloo.  It computes a whole oracle proof per call and
slices out the shard's repetitions; never used by the product."""
import numpy as np

import oracle_lib


def _parse_single(buf, pos):
    n = int.from_bytes(buf[pos:pos + 8], "little"); pos += 8
    on = []
    for _ in range(n):
        start = pos
        pos += 1 + 128
        for _ in range(3):
            ln = int.from_bytes(buf[pos:pos + 8], "little"); pos += 8 + ln
        on.append(buf[start:pos])
    n = int.from_bytes(buf[pos:pos + 8], "little"); pos += 8
    pre = [buf[pos + 48 * i:pos + 48 * (i + 1)] for i in range(n)]
    return on, pre, pos + 48 * n


class OracleShardBackend:
    device_type = "cpu"

    def __init__(self, prog, wire_counts):
        self.prog, self.wc = prog, wire_counts

    def commit(self, wit_gf2, wit_z64, seeds, rep_begin, rep_count):
        # the oracle has no shard API: rebuild all 256 seeds deterministically from the caller's
        # full seed table (tests pass it via .all_seeds) and keep only this shard's rows
        full = self.all_seeds
        assert (np.asarray(seeds) == full[rep_begin:rep_begin + rep_count]).all()
        h, st, comm = oracle_lib.commit(self.prog, wit_gf2, wit_z64, self.wc, full, threads=2)
        proof = oracle_lib.prove(self.prog, wit_gf2, wit_z64, self.wc, full, threads=2)
        return {"begin": rep_begin, "count": rep_count, "h": h, "proof": proof}

    def digests(self, shard):
        return shard["h"][shard["begin"]:shard["begin"] + shard["count"]].copy()

    def open(self, shard, omit):
        proof = shard["proof"]
        on2, pre2, pos = _parse_single(proof, 32)
        on64, pre64, _ = _parse_single(proof, pos)
        assert proof[:32] == bytes(__import__("reverie_amd").combine_digests(shard["h"]))
        b, n = shard["begin"], shard["count"]
        k_on = int((omit[:b] < 8).sum())
        k_pre = b - k_on
        n_on = int((omit[b:b + n] < 8).sum())
        n_pre = n - n_on
        parts = [b"".join(on2[k_on:k_on + n_on]), b"".join(pre2[k_pre:k_pre + n_pre]),
                 b"".join(on64[k_on:k_on + n_on]), b"".join(pre64[k_pre:k_pre + n_pre])]
        return b"".join(parts), [len(p) for p in parts], n_on, n_pre

    def destroy(self, shard):
        pass
