"""Genome import in front of the simulation path: chrom.sizes + extrusion-barrier BED6
(+ optional BED3 of genomic intervals) -> the interval dictionaries `driver.plan_genome` takes.

The parsing rules live in the native library (modle_amd/csrc/genome_io.cpp behind
include/modle_genome.h; reference: src/libmodle/internal/genome.cpp:299-469,
src/libmodle_io/bed.cpp, src/libmodle_io/chrom_sizes.cpp); this module reads the files
(plain, .gz, .bz2 or .xz like the reference's compressed_io reader) and marshals."""
import bz2
import ctypes as C
import gzip
import lzma

import numpy as np

from ._lib import lib
from .params import Config


class GenomeError(ValueError):
    pass


class _Interval(C.Structure):
    _fields_ = [("id", C.c_uint64), ("chrom_id", C.c_uint64), ("start", C.c_uint64),
                ("end", C.c_uint64), ("num_barriers", C.c_uint64)]


_bound = False


def _bind():
    global _bound
    L = lib()
    if not _bound:
        P = C.POINTER
        L.modle_genome_import.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                          C.c_size_t, P(Config), C.c_int, P(C.c_void_p), C.c_char_p,
                                          C.c_size_t]
        L.modle_genome_free.argtypes = [C.c_void_p]
        L.modle_genome_free.restype = None
        L.modle_genome_num_chromosomes.argtypes = [C.c_void_p]
        L.modle_genome_num_chromosomes.restype = C.c_size_t
        L.modle_genome_chromosome.argtypes = [C.c_void_p, C.c_size_t, P(C.c_char_p), P(C.c_uint64)]
        L.modle_genome_num_intervals.argtypes = [C.c_void_p]
        L.modle_genome_num_intervals.restype = C.c_size_t
        L.modle_genome_interval_info.argtypes = [C.c_void_p, C.c_size_t, P(_Interval)]
        L.modle_genome_interval_barriers.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]
        L.modle_genome_barrier_counts.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]
        L.modle_genome_barrier_counts.restype = None
        _bound = True
    return L


def read_text(path):
    """file contents as bytes; gzip / bzip2 / xz are detected by their magic numbers"""
    with open(path, "rb") as fh:
        head = fh.read(6)
    if head[:2] == b"\x1f\x8b":
        opener = gzip.open
    elif head[:3] == b"BZh":
        opener = bz2.open
    elif head[:6] == b"\xfd7zXZ\x00":
        opener = lzma.open
    else:
        opener = open
    with opener(path, "rb") as fh:
        return fh.read()


def import_genome_text(cfg, chrom_sizes, barriers_bed, intervals_bed=None,
                       interpret_name_as_not_bound_stp=False):
    """Returns (chromosomes [(name, size)], intervals [dict], stats).  Interval dicts carry name,
    size (of the chromosome), start, end, bar_pos, bar_dir, bar_stp_active, bar_stp_inactive in
    import order (the simulation library sorts the barriers)."""
    L = _bind()
    as_bytes = lambda t: t if isinstance(t, (bytes, bytearray)) else (t or "").encode()
    cs, bb, ib = as_bytes(chrom_sizes), as_bytes(barriers_bed), as_bytes(intervals_bed)
    handle = C.c_void_p()
    err = C.create_string_buffer(1024)
    rc = L.modle_genome_import(cs, len(cs), bb, len(bb), ib if ib else None, len(ib), C.byref(cfg),
                               int(bool(interpret_name_as_not_bound_stp)), C.byref(handle), err,
                               len(err))
    if rc != 0:
        raise GenomeError(err.value.decode(errors="replace"))
    try:
        chroms = []
        for i in range(L.modle_genome_num_chromosomes(handle)):
            name, size = C.c_char_p(), C.c_uint64()
            L.modle_genome_chromosome(handle, i, C.byref(name), C.byref(size))
            chroms.append((name.value.decode(), size.value))
        intervals = []
        for i in range(L.modle_genome_num_intervals(handle)):
            info = _Interval()
            L.modle_genome_interval_info(handle, i, C.byref(info))
            n = info.num_barriers
            pos = np.zeros(n, dtype=np.uint64)
            dirs = np.zeros(n, dtype=np.uint8)
            sa = np.zeros(n, dtype=np.float64)
            si = np.zeros(n, dtype=np.float64)
            L.modle_genome_interval_barriers(handle, i, pos.ctypes.data, dirs.ctypes.data,
                                             sa.ctypes.data, si.ctypes.data)
            name, size = chroms[info.chrom_id]
            intervals.append({"name": name, "size": size, "start": int(info.start),
                              "end": int(info.end), "bar_pos": pos, "bar_dir": dirs,
                              "bar_stp_active": sa, "bar_stp_inactive": si})
        imported, dropped = C.c_uint64(), C.c_uint64()
        L.modle_genome_barrier_counts(handle, C.byref(imported), C.byref(dropped))
        return chroms, intervals, {"barriers_imported": imported.value,
                                   "barriers_without_strand": dropped.value}
    finally:
        L.modle_genome_free(handle)


def import_genome(cfg, path_to_chrom_sizes, path_to_barriers, path_to_intervals=None,
                  interpret_name_as_not_bound_stp=False):
    return import_genome_text(cfg, read_text(path_to_chrom_sizes), read_text(path_to_barriers),
                              read_text(path_to_intervals) if path_to_intervals else None,
                              interpret_name_as_not_bound_stp)
