"""ctypes view of the cooler (v3) writer, include/modle_cooler.h.  The product path is the C
library (modle_amd/libmodle_cooler.so, built by `make -C modle_amd/csrc cooler`); this module
mirrors how the reference's IO thread uses it (simulation.cpp:117-168, 217-269): create the file
with every chromosome, append the interval matrices in genome order, close."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmodle_cooler.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C modle_amd/csrc cooler` "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        lb = ctypes.CDLL(path)
        lb.modle_cool_create.restype = ctypes.c_int
        lb.modle_cool_create.argtypes = [
            ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
            ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p,
            ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p,
            ctypes.c_size_t]
        lb.modle_cool_append_matrix.restype = ctypes.c_int
        lb.modle_cool_append_matrix.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
            ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]
        lb.modle_cool_close.restype = ctypes.c_int
        lb.modle_cool_close.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _LIB = lb
    return _LIB


class CoolerError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"modle_cooler error {code}: {message}")
        self.code = code


class CoolerWriter:
    """`chroms`: list of (name, size); matrices are appended in ascending chromosome order."""

    def __init__(self, path, chroms, bin_size, assembly="unknown", generated_by="modle-hip",
                 metadata_json="", force_overwrite=False):
        names = (ctypes.c_char_p * len(chroms))(*[n.encode() for n, _ in chroms])
        sizes = (ctypes.c_uint32 * len(chroms))(*[int(s) for _, s in chroms])
        self._h = ctypes.c_void_p()
        self._err = ctypes.create_string_buffer(512)
        rc = lib().modle_cool_create(os.fsencode(path), int(force_overwrite), names, sizes,
                                     len(chroms), int(bin_size), assembly.encode(),
                                     generated_by.encode(), metadata_json.encode(),
                                     ctypes.byref(self._h), self._err, len(self._err))
        if rc != 0:
            self._h = None
            raise CoolerError(rc, self._err.value.decode())
        self._index = {n: i for i, (n, _) in enumerate(chroms)}

    def append(self, chrom, band, nrows, ncols, offset_bp=0):
        """band: uint32 array of nrows * ncols (+1) words in the layout of the HIP library"""
        import numpy as np
        band = np.ascontiguousarray(band, dtype=np.uint32)
        if band.size < nrows * ncols:
            raise ValueError("band matrix smaller than nrows * ncols")
        cid = self._index[chrom] if isinstance(chrom, str) else int(chrom)
        rc = lib().modle_cool_append_matrix(self._h, cid, int(offset_bp), band.ctypes.data,
                                            int(nrows), int(ncols), self._err, len(self._err))
        if rc != 0:
            raise CoolerError(rc, self._err.value.decode())

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            rc = lib().modle_cool_close(h, self._err, len(self._err))
            if rc != 0:
                raise CoolerError(rc, self._err.value.decode())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
