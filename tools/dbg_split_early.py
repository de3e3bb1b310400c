import os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
os.environ.update(RV_FLAT=sys.argv[1], RV_EARLY="2", RV_EARLY_MIN="1000", RV_EARLY_POISON="1", RV_FLAT_BANDS=sys.argv[2], RV_EARLY_CHUNKS=sys.argv[3])
import reverie_amd as rv, circuits, oracle_lib
oracle_lib.build()
prog, wit, wc, st = circuits.layered_gf2(n_in=256, width=4096, layers=20)
seeds = np.frombuffer(bytes(range(256)) * 16, np.uint8).reshape(256, 16)
want = oracle_lib.prove(prog, wit, [], wc, seeds, threads=8)
c = rv.Circuit(prog, wc, whole_prover=True)
for it in range(3):
    p = bytes(rv.Proof.new(c, wit, [], seeds=seeds))
    a = np.frombuffer(p, np.uint8); b = np.frombuffer(want, np.uint8)
    d = np.nonzero(a != b)[0] if len(a) == len(b) else None
    print(it, len(p), len(want), "equal" if d is not None and len(d) == 0 else ("diff at %d..%d n=%d" % (d[0], d[-1], len(d)) if d is not None else "len"))
