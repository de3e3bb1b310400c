"""The library's multi-rank prover with MORE THAN ONE rank on the 1-GPU box: RCCL refuses two ranks on one device, so a fresh
process binds the rccl test shim (tests/rccl_shim: ranks = host threads sharing the GPU, the collectives as device copies
between them) through RV_RCCL_PATH and runs rv_comm_create_all + rv_prove_multi with 2, 4 and 8 ranks.  What this covers that
nothing else does: rv_prove_sharded's all-gather offsets, the shard form of the device-side challenge on gathered digests, the
section bookkeeping and the grouped send / recv of every rank's openings into rank 0's framed proof (csrc/comm.inc; the
reference meets at the same point, proof/mod.rs:160-172)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_SRC = os.path.join(ROOT, "tests", "rccl_shim", "rccl_shim.cpp")
SHIM = os.path.join(ROOT, "tests", "rccl_shim", "_build", "librccl_shim.so")


def build_shim():
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < os.path.getmtime(SHIM_SRC):
        os.makedirs(os.path.dirname(SHIM), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-std=c++17", SHIM_SRC, "-o", SHIM])
    return SHIM


@pytest.mark.parametrize("sendrecv", ["0", "1"])
@pytest.mark.parametrize("worlds", [("2", "4"), ("8",)])
def test_rv_prove_multi_with_several_ranks_on_one_gpu(worlds, sendrecv):
    """sendrecv = 1: the gather rv_prove_sharded uses between PROCESSES (sections to rank 0 by ncclSend / ncclRecv, RV_MULTI_SENDRECV);
    0: rv_prove_multi's own (round 6: every rank copies its sections into one page-locked buffer)"""
    env = dict(os.environ, RV_RCCL_PATH=build_shim(), RV_MULTI_SENDRECV=sendrecv)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), *worlds], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res and all(res.values()), res
    for n in worlds:
        assert res["mixed/%s/0" % n] and res["layered/%s/1" % n] and res["mixed/%s/invalid-witness" % n]


def test_rv_prove_multi_full_size_config4():
    """BASELINE config 4 at full size (10^7 gates, 5 014 185 AND) through the multi-rank C path with 2 and 8 ranks sharing the GPU:
    every rank holds the replicated gate stream (0.3 GB) and a 128- / 32-repetition shard; rank 0's framed 50 MB proof must be
    the oracle's and rv_prove's, byte for byte (reference: proof/mod.rs:127-172)"""
    env = dict(os.environ, RV_RCCL_PATH=build_shim(), MULTI_FULL="1", RV_MULTI_SENDRECV="1")  # (the send / recv gather at full size; the shared buffer: the test above)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), "2", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res == {"config4/2/0": True, "config4/8/0": True}, res
