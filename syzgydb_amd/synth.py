"""Synthetic benchmark inputs: counter-based splitmix64, element
e of stream `seed` = U[-1,1) = (mix(seed + e) >> 11) * 2^-52 - 1.  The HIP
library generates the corpus with the same function on the device
(szg_index_synth); this numpy version makes the float64 queries on the host.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synth_vectors(seed, first_row, n_rows, dim):
    """[n_rows, dim] float64, rows first_row.. of stream `seed`."""
    with np.errstate(over="ignore"):
        idx = (np.uint64(seed) + (np.uint64(first_row) * np.uint64(dim))
               + np.arange(int(n_rows) * int(dim), dtype=np.uint64))
    m = splitmix64(idx) >> np.uint64(11)
    v = m.astype(np.float64) * (1.0 / 4503599627370496.0) - 1.0
    return v.reshape(int(n_rows), int(dim))
