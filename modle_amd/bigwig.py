"""ctypes view of the bigWig writer (include/modle_bigwig.h, in libmodle_cooler.so): the 1-D LEF
occupancy track the reference writes next to the cooler (simulation.cpp:170-197)."""
import ctypes
import os

import numpy as np

from . import cooler as _cooler


class BigWigError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"modle_bigwig error {code}: {message}")
        self.code = code


_bound = False


def lib():
    global _bound
    lb = _cooler.lib()
    if not _bound:
        lb.modle_bw_create.restype = ctypes.c_int
        lb.modle_bw_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                       ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        lb.modle_bw_write_range.restype = ctypes.c_int
        lb.modle_bw_write_range.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32,
                                            ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        lb.modle_bw_write_occupancy.restype = ctypes.c_int
        lb.modle_bw_write_occupancy.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32,
                                                ctypes.c_char_p, ctypes.c_size_t]
        lb.modle_bw_close.restype = ctypes.c_int
        lb.modle_bw_close.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _bound = True
    return lb


class BigWigWriter:
    """`chroms`: list of (name, size) in genome order; ranges are appended in genome order."""

    def __init__(self, path, chroms, force_overwrite=False):
        names = (ctypes.c_char_p * len(chroms))(*[n.encode() for n, _ in chroms])
        sizes = (ctypes.c_uint32 * len(chroms))(*[int(s) for _, s in chroms])
        self._h = ctypes.c_void_p()
        self._err = ctypes.create_string_buffer(512)
        rc = lib().modle_bw_create(os.fsencode(path), int(force_overwrite), names, sizes, len(chroms),
                                   ctypes.byref(self._h), self._err, len(self._err))
        if rc != 0:
            self._h = None
            raise BigWigError(rc, self._err.value.decode())
        self._index = {n: i for i, (n, _) in enumerate(chroms)}

    def _cid(self, chrom):
        return self._index[chrom] if isinstance(chrom, str) else int(chrom)

    def write_range(self, chrom, values, span, step, offset=0):
        v = np.ascontiguousarray(values, dtype=np.float32)
        rc = lib().modle_bw_write_range(self._h, self._cid(chrom), v.ctypes.data, len(v), int(span),
                                        int(step), int(offset), self._err, len(self._err))
        if rc != 0:
            raise BigWigError(rc, self._err.value.decode())

    def write_occupancy(self, chrom, occupancy, bin_size, offset_bp=0):
        """counts / max(counts) as float32, span = step = bin size (write_lef_occupancy_to_bwig)"""
        o = np.ascontiguousarray(occupancy, dtype=np.uint64)
        rc = lib().modle_bw_write_occupancy(self._h, self._cid(chrom), o.ctypes.data, len(o),
                                            int(bin_size), int(offset_bp), self._err, len(self._err))
        if rc != 0:
            raise BigWigError(rc, self._err.value.decode())

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            rc = lib().modle_bw_close(h, self._err, len(self._err))
            if rc != 0:
                raise BigWigError(rc, self._err.value.decode())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
