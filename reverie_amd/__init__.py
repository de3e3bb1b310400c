"""reverie_amd — MI355X-native KKW (MPC-in-the-head) prover/verifier hot path.

Public surface mirrors the reference's `Proof` API (see proof.py) over the C-ABI in
include/reverie_amd.h; `ops` builds gate streams.
"""
from .ops import B2A, GF2, Z64, SizeHint, largest_wires, program  # noqa: F401
from .proof import Circuit, Context, Proof, challenge, combine_digests, verify_batch  # noqa: F401
from ._lib import ReverieError  # noqa: F401
