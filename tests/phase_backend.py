"""KAT backend that drives the product's device code through its phase-level entry point.

`phases(cfg, mask, st, prng_state) -> raws_consumed` is either the CPU lane emulator
(tests/wave_emu, runs everywhere) or `modle_hip_test_phases` (the real GPU path).  The hook ->
phase-mask mapping mirrors the reference's `Simulation::test_*` hooks
(reference: src/libmodle/cpu/include/modle/simulation.hpp:413-567).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from modle_amd.params import Config

PH_RANK, PH_RANK_INIT, PH_ADJUST, PH_CLAMP = 0x001, 0x002, 0x004, 0x008
PH_BOUNDARIES, PH_LEF_BAR, PH_PRIMARY = 0x010, 0x020, 0x040
PH_CORRECT_LEF_BAR, PH_CORRECT_PRIMARY = 0x080, 0x100
PH_SECONDARY, PH_FIX_SECONDARY, PH_USE_BOUNDARY_COUNTS = 0x200, 0x400, 0x800

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

_EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wave_emu")


def splitmix_seed(seed):
    """SplitMix64 x4 (reference: random.hpp:26-30); pure-python so tests need no library for it."""
    out = []
    mask = (1 << 64) - 1
    for _ in range(4):
        seed = (seed + 0x9E3779B97F4A7C15) & mask
        z = seed
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        out.append(z ^ (z >> 31))
    return out


class PhaseBackend:
    def __init__(self, phases):
        self._phases = phases

    def make_config(self, c):
        cfg = Config()
        cfg.bin_size = 1
        cfg.rev_extrusion_speed = c.get("rev_speed", 0)
        cfg.fwd_extrusion_speed = c.get("fwd_speed", 0)
        cfg.probability_of_extrusion_unit_bypass = c["bypass"]
        cfg.lef_bar_major_collision_pblock = c["major_pblock"]
        cfg.lef_bar_minor_collision_pblock = c["minor_pblock"]
        cfg.burnin_history_length = 100
        cfg.burnin_smoothing_window_size = 5
        return cfg

    def make_prng(self, seed):
        # mutable [state, consumed]; every call resumes the stream where the previous one stopped
        return {"seed_state": splitmix_seed(seed), "consumed": 0}

    def _run(self, cfg, mask, st, rng=None):
        if cfg is None:
            cfg = self.make_config({"bypass": 0.0, "major_pblock": 1.0, "minor_pblock": 0.0})
        state = list(rng["seed_state"]) if rng else [1, 2, 3, 4]
        skip = rng["consumed"] if rng else 0
        consumed = self._phases(cfg, mask, st, state, skip)
        if rng:
            rng["consumed"] += consumed

    def rank_lefs(self, st, init_buffers=False):
        self._run(None, PH_RANK | (PH_RANK_INIT if init_buffers else 0), st)

    def adjust_and_clamp_moves(self, st):
        self._run(None, PH_ADJUST | PH_CLAMP, st)

    def detect_units_at_interval_boundaries(self, st):
        self._run(None, PH_BOUNDARIES, st)

    def detect_lef_bar_collisions(self, cfg, st, rng):
        self._run(cfg, PH_LEF_BAR, st, rng)

    def correct_moves_for_lef_bar_collisions(self, st):
        self._run(None, PH_CORRECT_LEF_BAR, st)

    def detect_primary_lef_lef_collisions(self, cfg, st, rng):
        self._run(cfg, PH_PRIMARY, st, rng)

    def process_lef_lef_collisions(self, cfg, st, rng):
        self._run(cfg, PH_PRIMARY | PH_CORRECT_PRIMARY | PH_SECONDARY, st, rng)

    def process_collisions(self, cfg, st, rng):
        self._run(cfg, PH_BOUNDARIES | PH_USE_BOUNDARY_COUNTS | PH_LEF_BAR | PH_PRIMARY |
                  PH_CORRECT_LEF_BAR | PH_CORRECT_PRIMARY | PH_SECONDARY, st, rng)

    def fix_secondary_lef_lef_collisions(self, st):
        self._run(None, PH_FIX_SECONDARY, st)


def _advance(state, n):
    """state of xoshiro256++ after n outputs (pure python; n is tiny in the KATs)"""
    mask = (1 << 64) - 1
    s = list(state)
    for _ in range(n):
        t = (s[1] << 17) & mask
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & mask
    return s


_emu = None
_emu_variants = {}


def emu_lib(variant=None):
    """the lane emulator build of the device code; variant "philox" = PHILOX generator policy"""
    global _emu
    if variant is not None:
        if variant not in _emu_variants:
            so = f"libmodle_emu_{variant}.so"
            subprocess.run(["make", "-C", _EMU_DIR, so], check=True, capture_output=True)
            _emu_variants[variant] = C.CDLL(os.path.join(_EMU_DIR, so))
        return _emu_variants[variant]
    if _emu is None:
        subprocess.run(["make", "-C", _EMU_DIR], check=True, capture_output=True)
        L = C.CDLL(os.path.join(_EMU_DIR, "libmodle_emu.so"))
        L.emu_test_phases.argtypes = ([C.POINTER(Config), C.c_uint32, C.c_uint64, C.c_uint64,
                                       C.c_size_t] + [u64p] * 9 +
                                      [C.c_size_t, u64p, u8p, u8p, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)])
        L.emu_test_phases.restype = C.c_int
        L.emu_set_lane_schedule.argtypes = [C.c_uint]
        L.emu_set_lane_schedule.restype = None
        # lane schedule of the emulator (0 = ascending, 1 = descending, n = random permutation n)
        L.emu_set_lane_schedule(int(os.environ.get("MODLE_EMU_SCHEDULE", "0")))
        _emu = L
    return _emu


def emu_size_class(cfg, max_lefs, moves=()):
    """0 = NARROW, 1 = WIDE: the size class the product runs a set-up in (modle_hip_size_class; the emulator's
    build of host_logic.cpp answers), WIDE also when a caller-supplied move does not fit 16 bits"""
    L = emu_lib()
    L.modle_hip_size_class.argtypes = [C.POINTER(Config), C.c_uint64]
    wide = L.modle_hip_size_class(C.byref(cfg), int(max_lefs)) != 0
    for m in moves:
        wide = wide or (len(m) != 0 and int(np.max(m)) > 65533)
    return 1 if wide else 0


def _declare_phases(L):
    L.emu_test_phases.argtypes = ([C.POINTER(Config), C.c_uint32, C.c_uint64, C.c_uint64,
                                   C.c_size_t] + [u64p] * 9 +
                                  [C.c_size_t, u64p, u8p, u8p, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)])
    L.emu_test_phases.restype = C.c_int
    return L


def emu_phases(cfg, mask, st, state, skip):
    L = emu_lib()
    if emu_size_class(cfg, st.n, (st.rev_moves, st.fwd_moves)) != 0:
        L = _declare_phases(emu_lib("wide"))
    prng = (C.c_uint64 * 4)(*_advance(state, skip))
    consumed = C.c_uint64(0)
    rc = L.emu_test_phases(C.byref(cfg), mask, st.start, st.end, st.n, st.rev_pos, st.fwd_pos,
                           st.epoch, st.rev_rank, st.fwd_rank, st.rev_moves, st.fwd_moves,
                           st.rev_coll, st.fwd_coll, len(st.bar_pos), st.bar_pos, st.bar_dir,
                           st.bar_active, prng, C.byref(consumed))
    assert rc == 0, f"emu_test_phases failed: {rc}"
    return consumed.value
