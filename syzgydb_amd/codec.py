"""Host-side vector codec: float64 vectors <-> the reference's packed row bytes.

Mirrors quantize/dequantize (quantization.go:5-36), encodeDocument
(collection.go:713-743), decodeVector (collection.go:768-794) and
getVectorSize (collection.go:796-811) with numpy, so the Python host layer can
build the byte stream the HIP scan reads.  (The parity oracle has its own,
independent C restatement; tests compare the two.)
"""
import numpy as np

SUPPORTED_BITS = (4, 8, 16, 32, 64)


def vector_size(bits, dim):
    """getVectorSize, collection.go:796-811."""
    if bits == 4:
        return (dim + 1) // 2
    if bits == 8:
        return dim
    if bits == 16:
        return dim * 2
    if bits == 32:
        return dim * 4
    if bits == 64:
        return dim * 8
    raise ValueError("Unsupported quantization level")  # the reference panics


def _round_half_away(x):
    # math.Round for x >= 0 without the floor(x + 0.5) double-rounding trap
    f = np.floor(x)
    return f + ((x - f) >= 0.5)


def quantize(values, bits):
    """quantization.go:5-23 for an array; returns uint64 codes."""
    v = np.asarray(values, dtype=np.float64)
    if bits == 32:
        return v.astype(np.float32).view(np.uint32).astype(np.uint64)
    if bits == 64:
        return v.view(np.uint64).copy()
    if bits not in (4, 8, 16):
        raise ValueError("Unsupported quantization level")
    max_int = float((1 << bits) - 1)
    c = np.where(v < -1, -1.0, np.where(v > 1, 1.0, v))
    q = (c + 1) / 2 * max_int
    return _round_half_away(q).astype(np.uint64)


def dequantize(codes, bits):
    """quantization.go:25-36 for an array of uint64 codes."""
    c = np.asarray(codes, dtype=np.uint64)
    if bits == 32:
        return c.astype(np.uint32).view(np.float32).astype(np.float64)
    if bits == 64:
        return c.view(np.float64).copy()
    max_int = float((1 << bits) - 1)
    return (c.astype(np.float64) / max_int) * 2 - 1


def encode_rows(vectors, bits):
    """encodeDocument for a [n, dim] float64 matrix -> [n, vector_size] uint8."""
    V = np.atleast_2d(np.asarray(vectors, dtype=np.float64))
    n, dim = V.shape
    if bits == 64:
        return np.ascontiguousarray(V.astype(">f8")).view(np.uint8).reshape(n, dim * 8)
    if bits == 32:
        with np.errstate(over="ignore"):
            f = V.astype(np.float32)
        return np.ascontiguousarray(f.astype(">f4")).view(np.uint8).reshape(n, dim * 4)
    q = quantize(V, bits)
    if bits == 16:
        return np.ascontiguousarray(q.astype(">u2")).view(np.uint8).reshape(n, dim * 2)
    if bits == 8:
        return q.astype(np.uint8)
    # 4-bit: even index -> high nibble, odd index -> low nibble (collection.go:724-729)
    out = np.zeros((n, (dim + 1) // 2), dtype=np.uint8)
    out[:, :] = (q[:, 0::2].astype(np.uint8) << 4)
    odd = q[:, 1::2].astype(np.uint8) & 0x0F
    out[:, : odd.shape[1]] |= odd
    return out


def decode_rows(data, dim, bits):
    """decodeVector for [n, vector_size] uint8 rows -> [n, dim] float64."""
    rb = vector_size(bits, dim)
    D = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, rb)
    n = D.shape[0]
    if bits == 64:
        return D.view(">f8").astype(np.float64).reshape(n, dim)
    if bits == 32:
        return D.view(">f4").astype(np.float64).reshape(n, dim)
    if bits == 16:
        codes = D.view(">u2").astype(np.uint64).reshape(n, dim)
    elif bits == 8:
        codes = D.astype(np.uint64)
    else:
        codes = np.zeros((n, dim), dtype=np.uint64)
        codes[:, 0::2] = (D >> 4)[:, : (dim + 1) // 2]
        codes[:, 1::2] = (D & 0x0F)[:, : dim // 2]
    return dequantize(codes, bits)
