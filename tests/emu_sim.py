"""Whole-cell simulation on the product's device code under the CPU lane emulator."""
import ctypes as C
import os

import numpy as np

from modle_amd.params import CellResult, Config, Task
from phase_backend import emu_lib

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def simulate_interval(cfg, start, end, bar_pos, bar_dir, stp_active, stp_inactive, tasks, nrows,
                      ncols, track_occupancy=True, variant=None):
    from phase_backend import emu_size_class

    # MODLE_EMU_VARIANT=w12: campaigns on the geometry of the 12-wave kernels (tools/emu_fuzz_campaign.py)
    if variant is None and os.environ.get("MODLE_EMU_VARIANT"):
        variant = os.environ["MODLE_EMU_VARIANT"]

    # the size class the product would run these cells in (NARROW builds refuse WIDE set-ups)
    if variant in (None, "philox") and emu_size_class(cfg, max(int(t.num_lefs) for t in tasks)) != 0:
        variant = "wide" if variant is None else "philox_wide"
    L = emu_lib(variant)
    L.emu_simulate_interval.argtypes = [C.POINTER(Config), C.c_uint64, C.c_uint64, u64p, u8p,
                                        f64p, f64p, C.c_size_t, C.POINTER(Task), C.c_size_t,
                                        u32p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64),
                                        C.c_void_p, C.POINTER(CellResult)]
    L.emu_simulate_interval.restype = C.c_int
    contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32)
    occ = np.zeros(ncols, dtype=np.uint64) if track_occupancy else None
    missed = C.c_uint64(0)
    n = len(tasks)
    results = (CellResult * n)()
    rc = L.emu_simulate_interval(
        C.byref(cfg), start, end, np.ascontiguousarray(bar_pos, dtype=np.uint64),
        np.ascontiguousarray(bar_dir, dtype=np.uint8),
        np.ascontiguousarray(stp_active, dtype=np.float64),
        np.ascontiguousarray(stp_inactive, dtype=np.float64), len(bar_pos), tasks, n, contacts,
        nrows, ncols, C.byref(missed), occ.ctypes.data if occ is not None else None, results)
    assert rc == 0, f"emu_simulate_interval failed: {rc}"
    return contacts, missed.value, occ, results
