"""SpanfilePager: open a SyzgyDB collection file (.dat) through the C++ pager
(include/syzgy_pager.h) -- the read side of spanfile.go restated, no Go needed."""
import ctypes

import numpy as np

from . import _lib


class SpanfilePager:
    def __init__(self, path, n_threads=0):
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        rc = self._L.szg_pager_open(ctypes.byref(self._h), str(path).encode(), int(n_threads))
        if rc != _lib.SZG_OK:
            raise _lib.SzgError(rc, "szg_pager_open", {_lib.SZG_E_IO: "cannot open/map file",
                                                        _lib.SZG_E_FORMAT: "no usable header record"}.get(rc, ""))
        d, q, m = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._L.szg_pager_options(self._h, ctypes.byref(d), ctypes.byref(q), ctypes.byref(m))
        self.dim, self.quant_bits, self.metric = d.value, q.value, m.value
        self.row_bytes = int(self._L.szg_row_bytes(self.quant_bits, self.dim))

    def close(self):
        if self._h:
            self._L.szg_pager_close(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def count(self):
        return int(self._L.szg_pager_count(self._h))

    @property
    def skipped(self):
        return int(self._L.szg_pager_skipped(self._h))

    def ids(self):
        out = np.zeros(self.count, dtype=np.uint64)
        _lib.check(self._L.szg_pager_ids(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))),
                   "szg_pager_ids")
        return out

    def vectors(self):
        out = np.zeros((self.count, self.row_bytes), dtype=np.uint8)
        _lib.check(self._L.szg_pager_vectors(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                             out.size), "szg_pager_vectors")
        return out

    def metadata(self, row):
        p = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_uint64(0)
        _lib.check(self._L.szg_pager_metadata(self._h, int(row), ctypes.byref(p), ctypes.byref(n)),
                   "szg_pager_metadata")
        return ctypes.string_at(p, n.value) if n.value else b""

    def load_into(self, index):
        _lib.check(self._L.szg_pager_load(self._h, index._h), "szg_pager_load")
