"""Thin Python view of the C ABI (include/modle_hip.h).

Host-side logic (Config defaults, derived parameters, task generation) and the device path are
both implemented natively in libmodle_hip.so; this module only marshals arguments.  Names follow
the reference: `Config` (simulation_config.hpp), `Task` / `State` results (simulation.hpp:59-135),
`run_simulate`'s task generation (scheduler_simulate.cpp:104-160).
"""
import ctypes as C

import numpy as np

from ._lib import lib
from .params import CellResult, Config, LaunchInfo, Task


Config = Config  # re-exported: `api.Config` is the ctypes mirror of modle_hip_config


ERR_ARG, ERR_DEVICE, ERR_UNSUPPORTED, ERR_STATE, ERR_CANCELLED, ERR_TIMEOUT = -1, -2, -3, -4, -5, -6


class ModleHipError(RuntimeError):
    """a negative return code of the C ABI (`code`: MODLE_HIP_ERR_*)"""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


def _check(rc, err):
    if rc < 0:
        raise ModleHipError(f"modle_hip error {rc}: {err.value.decode(errors='replace')}", rc)
    return rc


def _errbuf():
    return C.create_string_buffer(512)


# ---------------------------------------------------------------------------------------------
# host logic
# ---------------------------------------------------------------------------------------------
def default_config(**overrides):
    """Config with the reference defaults; `overrides` set raw (pre-transform) fields."""
    cfg = Config()
    lib().modle_hip_config_default(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"unknown Config field {k}")
        setattr(cfg, k, v)
    return cfg


def transform_config(cfg):
    """Cli::transform_args: derive speeds, release probabilities, burn-in parameters."""
    err = _errbuf()
    _check(lib().modle_hip_config_transform(C.byref(cfg), err, len(err)), err)
    return cfg


def make_config(**overrides):
    return transform_config(default_config(**overrides))


def interval_hash(name, chrom_size, start, end, seed):
    return lib().modle_hip_interval_hash(name.encode(), chrom_size, start, end, seed)


def prng_seed(seed):
    st = (C.c_uint64 * 4)()
    lib().modle_hip_prng_seed(seed, st)
    return [int(x) for x in st]


def prng_jump(state):
    st = (C.c_uint64 * 4)(*state)
    lib().modle_hip_prng_jump(st)
    return [int(x) for x in st]


def compute_num_lefs(cfg, size_bp):
    return lib().modle_hip_compute_num_lefs(C.byref(cfg), size_bp)


def compute_contacts_per_epoch(cfg, nlefs):
    return lib().modle_hip_compute_contacts_per_epoch(C.byref(cfg), nlefs)


def matrix_shape(cfg, size_bp):
    nr, nc = C.c_uint64(), C.c_uint64()
    lib().modle_hip_matrix_shape(C.byref(cfg), size_bp, C.byref(nr), C.byref(nc))
    return nr.value, nc.value


def make_tasks(cfg, name, chrom_size, start, end, first_task_id=0):
    tasks = (Task * int(cfg.num_cells))()
    rc = lib().modle_hip_make_tasks(C.byref(cfg), name.encode(), chrom_size, start, end,
                                    first_task_id, tasks)
    if rc < 0:
        raise ModleHipError(f"modle_hip_make_tasks failed: {rc}")
    return tasks


def stp_active_from_occupancy(stp_inactive, occupancy):
    return lib().modle_hip_stp_active_from_occupancy(stp_inactive, occupancy)


def barrier_stps(cfg, occupancy):
    """Per-barrier self-transition probabilities from BED scores (reference: genome.cpp:260-271)."""
    occupancy = np.asarray(occupancy, dtype=np.float64)
    stp_inactive = np.full(len(occupancy), cfg.barrier_not_occupied_stp, dtype=np.float64)
    stp_active = np.array(
        [stp_active_from_occupancy(cfg.barrier_not_occupied_stp, o) if o != 0.0
         else cfg.barrier_occupied_stp for o in occupancy], dtype=np.float64)
    return stp_active, stp_inactive


def sort_barriers(bar_pos, bar_dir, stp_active, stp_inactive):
    """ExtrusionBarriers::sort: the four arrays ordered by position (copies)."""
    pos = np.array(bar_pos, dtype=np.uint64)
    dirs = np.array(bar_dir, dtype=np.uint8)
    sa = np.array(stp_active, dtype=np.float64)
    si = np.array(stp_inactive, dtype=np.float64)
    lib().modle_hip_sort_barriers(pos, dirs, sa, si, len(pos))
    return pos, dirs, sa, si


def slice_tasks(tasks, lo, hi):
    n = hi - lo
    out = (Task * n)()
    for i in range(n):
        C.memmove(C.byref(out[i]), C.byref(tasks[lo + i]), C.sizeof(Task))
    return out


# ---------------------------------------------------------------------------------------------
# device path
# ---------------------------------------------------------------------------------------------
class Simulator:
    """One simulation context bound to one GPU (one process per GPU)."""

    def __init__(self, cfg, device=0):
        err = _errbuf()
        self._L = lib()
        self.cfg = cfg
        self._h = self._L.modle_hip_create(C.byref(cfg), device, err, len(err))
        if not self._h:
            raise ModleHipError(f"modle_hip_create failed: {err.value.decode(errors='replace')}")
        self._n_submitted = {}

    def close(self):
        if self._h:
            self._L.modle_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._L.modle_hip_reset(self._h)
        self._n_submitted = {}

    def add_interval(self, start, end, bar_pos, bar_dir, stp_active, stp_inactive,
                     d_contacts=None, d_occupancy=None):
        err = _errbuf()
        rc = self._L.modle_hip_add_interval(
            self._h, start, end, np.ascontiguousarray(bar_pos, dtype=np.uint64),
            np.ascontiguousarray(bar_dir, dtype=np.uint8),
            np.ascontiguousarray(stp_active, dtype=np.float64),
            np.ascontiguousarray(stp_inactive, dtype=np.float64), len(bar_pos), d_contacts,
            d_occupancy, err, len(err))
        iv = _check(rc, err)
        self._n_submitted[iv] = 0
        return iv

    def submit(self, interval_id, tasks):
        err = _errbuf()
        _check(self._L.modle_hip_submit_tasks(self._h, interval_id, tasks, len(tasks), err,
                                              len(err)), err)
        self._n_submitted[interval_id] += len(tasks)

    def launch(self, stream=None):
        err = _errbuf()
        _check(self._L.modle_hip_launch(self._h, stream, err, len(err)), err)

    def wait(self):
        """collects the launch; raises ModleHipError with code ERR_TIMEOUT when the launch ran into
        the deadline (set_wait_timeout / MODLE_HIP_WAIT_TIMEOUT_S) and was aborted"""
        err = _errbuf()
        _check(self._L.modle_hip_wait(self._h, err, len(err)), err)

    def set_wait_timeout(self, seconds):
        if self._L.modle_hip_set_wait_timeout(self._h, float(seconds)) < 0:
            raise ModleHipError("modle_hip_set_wait_timeout: the deadline must be positive", ERR_ARG)

    def launch_info(self):
        """layout of the last launch (modle_hip_launch_info) as a dict"""
        info = LaunchInfo()
        if self._L.modle_hip_last_launch_info(self._h, C.byref(info)) < 0:
            raise ModleHipError("modle_hip_last_launch_info failed")
        return {name: int(getattr(info, name)) for name, _ in LaunchInfo._fields_}

    def enable_state_log(self, max_epochs_per_task):
        """--log-model-internal-state: needs the diagnostic build (MODLE_HIP_LIB=
        libmodle_hip_statelog.so); the default build refuses"""
        err = _errbuf()
        _check(self._L.modle_hip_enable_state_log(self._h, int(max_epochs_per_task), err, len(err)), err)
        self._state_log_cap = int(max_epochs_per_task)

    def state_log(self, interval_id, task_index):
        """records of one task of the last launch: uint64 array (n_epochs, 10), see
        MODLE_HIP_STATE_LOG_WORDS in include/modle_hip.h"""
        cap = getattr(self, "_state_log_cap", 0)
        rec = np.zeros((cap, 10), dtype=np.uint64)
        n = C.c_size_t(0)
        err = _errbuf()
        _check(self._L.modle_hip_get_state_log(self._h, interval_id, task_index, rec.ctypes.data, cap,
                                               C.byref(n), err, len(err)), err)
        return rec[:n.value]

    def interval_done(self, interval_id):
        """True when the launch in flight has finished every task of this interval."""
        rc = self._L.modle_hip_interval_done(self._h, interval_id)
        if rc < 0:
            raise ModleHipError(f"modle_hip_interval_done failed: {rc}")
        return rc == 1

    def cancel(self):
        err = _errbuf()
        _check(self._L.modle_hip_cancel(self._h, err, len(err)), err)

    def kernel_ms(self):
        ms = C.c_float(0)
        self._L.modle_hip_last_kernel_ms(self._h, C.byref(ms))
        return ms.value

    def results(self, interval_id):
        n = self._n_submitted[interval_id]
        res = (CellResult * n)()
        rc = self._L.modle_hip_get_results(self._h, interval_id, res, n)
        if rc < 0:
            raise ModleHipError(f"modle_hip_get_results failed: {rc}")
        return res

    def outputs(self, interval_id):
        dc, do = C.c_void_p(), C.c_void_p()
        nr, nc = C.c_uint64(), C.c_uint64()
        self._L.modle_hip_interval_outputs(self._h, interval_id, C.byref(dc), C.byref(do),
                                           C.byref(nr), C.byref(nc))
        return dc.value, do.value, nr.value, nc.value

    def copy_outputs(self, interval_id, want_contacts=True):
        _, d_occ, nrows, ncols = self.outputs(interval_id)
        contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32) if want_contacts else None
        occ = np.zeros(ncols, dtype=np.uint64) if d_occ else None
        missed = C.c_uint64(0)
        err = _errbuf()
        _check(self._L.modle_hip_copy_outputs(
            self._h, interval_id, contacts.ctypes.data if contacts is not None else None,
            C.byref(missed), occ.ctypes.data if occ is not None else None, err, len(err)), err)
        return contacts, missed.value, occ

    def simulate_interval(self, start, end, bar_pos, bar_dir, stp_active, stp_inactive, tasks):
        """One-call seam (modle_hip_simulate_interval): returns contacts, missed, occupancy,
        results."""
        nrows, ncols = matrix_shape(self.cfg, end - start)
        contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32)
        occ = np.zeros(ncols, dtype=np.uint64)
        missed = C.c_uint64(0)
        res = (CellResult * len(tasks))()
        err = _errbuf()
        _check(self._L.modle_hip_simulate_interval(
            self._h, start, end, np.ascontiguousarray(bar_pos, dtype=np.uint64),
            np.ascontiguousarray(bar_dir, dtype=np.uint8),
            np.ascontiguousarray(stp_active, dtype=np.float64),
            np.ascontiguousarray(stp_inactive, dtype=np.float64), len(bar_pos), tasks, len(tasks),
            contacts, nrows, ncols, C.byref(missed), occ.ctypes.data, res, err, len(err)), err)
        return contacts, missed.value, occ, res

    def test_units(self, what, pairs, nrows=0, ncols=0, contacts=None, missed=0):
        """Unit-level entry point (modle_hip_test_units); returns (out words, contacts, missed)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint64).reshape(-1)
        n = len(pairs) // 2
        out = np.zeros(2 * n, dtype=np.uint64)
        m = C.c_uint64(missed)
        err = _errbuf()
        _check(self._L.modle_hip_test_units(
            self._h, what, pairs, n, nrows, ncols,
            contacts.ctypes.data if contacts is not None else None, C.byref(m), out, err,
            len(err)), err)
        return out, contacts, m.value

    def test_phases(self, mask, st, prng_state):
        """Phase-level entry point (mirrors Simulation::test_* hooks) on KatState-like arrays."""
        prng = (C.c_uint64 * 4)(*prng_state)
        consumed = C.c_uint64(0)
        err = _errbuf()
        _check(self._L.modle_hip_test_phases(
            self._h, mask, st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.epoch, st.rev_rank,
            st.fwd_rank, st.rev_moves, st.fwd_moves, st.rev_coll, st.fwd_coll, len(st.bar_pos),
            st.bar_pos, st.bar_dir, st.bar_active, prng, C.byref(consumed), err, len(err)), err)
        return consumed.value
