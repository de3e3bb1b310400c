"""N>1 path on CPU: world_size-2 and -4 gloo runs of reverie_amd.dist.prove_sharded with the
oracle-backed shard backend; the assembled proof must equal the single-process oracle proof."""
import json
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_path):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    import oracle_lib
    from oracle_backend import OracleShardBackend
    from reverie_amd.dist import prove_sharded, shard_range
    from reverie_amd.ops import OP_DTYPE, program

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = json.load(open(os.path.join(GOLDEN, "proofs.json")))[name]
    prog = program([tuple(o) for o in m["ops"]]) if m["ops"] else np.zeros(0, OP_DTYPE)
    wc = tuple(m["wire_counts"])
    seeds = np.random.default_rng(77).integers(0, 256, (256, 16), dtype=np.uint8)
    be = OracleShardBackend(prog, wc)
    be.all_seeds = seeds
    assert shard_range(rank, world) == (rank * 256 // world, 256 // world)
    proof = prove_sharded(be, m["wit_gf2"], [int(x) for x in m["wit_z64"]], seeds)
    if rank == 0:
        want = oracle_lib.prove(prog, m["wit_gf2"], [int(x) for x in m["wit_z64"]], wc, seeds, threads=2)
        ok = proof == want and oracle_lib.verify(prog, wc, proof, threads=2)
        open(out_path, "w").write("ok" if ok else "MISMATCH")
    else:
        assert proof is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "adder64"), (4, "ref_test")])
def test_sharded_prove_gloo(tmp_path, world, name):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_shard_range_rejects_bad_world():
    from reverie_amd.dist import shard_range

    with pytest.raises(ValueError):
        shard_range(0, 3)
    assert shard_range(7, 8) == (224, 32)
