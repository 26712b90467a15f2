import hashlib
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def hex_f64(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def hex_list(hs):
    return np.array([hex_f64(h) for h in hs], dtype=np.float64)


def load_kats():
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


def load_cases():
    with open(os.path.join(HERE, "golden", "scan_cases.json")) as f:
        return json.load(f)["cases"]


def case_rows(case, orc):
    """Packed rows of a golden case: stored bytes, or regenerated from the seed and
    checked against the stored SHA-256."""
    from_seed = orc.synth_rows(case["seed"], 0, case["n"], case["dim"], case["bits"])
    if case["rows_hex"] is not None:
        rows = np.frombuffer(bytes.fromhex(case["rows_hex"]), dtype=np.uint8).reshape(case["n"], -1)
        assert (rows == from_seed).all()
    else:
        rows = from_seed
    assert hashlib.sha256(rows.tobytes()).hexdigest() == case["rows_sha256"]
    return rows


def same_f64(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
