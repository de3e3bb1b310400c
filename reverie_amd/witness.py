"""Witness text parser — /root/reference/src/witness.rs:12-60: the file is scanned byte by
byte, '0' and '1' are witness bits, every other byte is ignored."""
from __future__ import annotations

import numpy as np


def parse_witness(data: bytes | str) -> np.ndarray:
    if isinstance(data, str):
        data = data.encode()
    a = np.frombuffer(data, dtype=np.uint8)
    keep = (a == 0x30) | (a == 0x31)
    return (a[keep] - 0x30).astype(np.uint8)
