"""The two size classes of the kernels (modle_amd/csrc/sim_types.h; round 5): NARROW keeps LEF ids and moves as
16-bit values, WIDE as 32-bit ones; `modle_hip_size_class` picks per launch and results do not depend on it.

CPU: the rule (`launch_common.hpp: size_class_required`) at its borders, and whole cells on the lane emulator in
both classes against the oracle.  GPU: the same cells through the C ABI in the class the library picks and with
WIDE forced, word for word; a configuration whose extrusion speed rules NARROW out is classed WIDE by itself."""
import ctypes as C

import numpy as np
import pytest

from parity_cases import assert_same_outputs, assert_same_results, build_case


def _size_class(cfg, max_lefs):
    from modle_amd import _lib
    from modle_amd.params import Config

    L = _lib.lib()
    L.modle_hip_size_class.argtypes = [C.POINTER(Config), C.c_uint64]
    return L.modle_hip_size_class(C.byref(cfg), int(max_lefs))


def test_the_rule_at_its_borders(monkeypatch):
    from modle_amd import api

    monkeypatch.delenv("MODLE_HIP_SIZE_CLASS", raising=False)
    default = api.make_config()
    # every real chromosome at the reference's defaults (chr1: 4 979 LEFs), BASELINE configs[4] (15 933) too
    assert _size_class(default, 4979) == 0
    dense = api.make_config(number_of_lefs_per_mbp=64.0, lef_bar_minor_collision_pblock=0.3,
                            soft_stall_lef_stability_multiplier=2.0)
    assert _size_class(dense, 15933) == 0
    # ids: 65 536 LEFs and more need 32 bits
    assert _size_class(default, 65535) in (0, 1) and _size_class(default, 65536) == 1
    # moves: speed + 40 sigma + LEFs + 2 must stay within 65 533
    speed = int(max(default.rev_extrusion_speed, default.fwd_extrusion_speed))
    sd = float(max(default.rev_extrusion_speed_std, default.fwd_extrusion_speed_std))
    room = int(65533 - 2 - speed - 40.0 * sd)
    assert _size_class(default, room) == 0 and _size_class(default, room + 1) == 1
    fast = api.make_config(rev_extrusion_speed=70000, fwd_extrusion_speed=70000, rev_extrusion_speed_set=1,
                           fwd_extrusion_speed_set=1)
    assert _size_class(fast, 100) == 1
    monkeypatch.setenv("MODLE_HIP_SIZE_CLASS", "wide")
    assert _size_class(default, 100) == 1


@pytest.mark.parametrize("name", ["chr20mb_barriers", "many_rebinds_per_epoch"])
def test_emulated_cells_are_identical_in_both_classes(oracle, monkeypatch, name):
    import emu_sim
    from modle_amd import api

    case = build_case(name)
    cfg, ch = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, 1)
    track = bool(cfg.track_1d_lef_position)
    ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                   case["stp_inactive"], tasks, nthreads=2, track_occupancy=track)
    for forced in (None, "wide"):
        if forced:
            monkeypatch.setenv("MODLE_HIP_SIZE_CLASS", forced)
        got = emu_sim.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                        case["stp_inactive"], tasks, case["nrows"], case["ncols"], track_occupancy=track)
        assert_same_results(ref[3], got[3], f"{name}, emulator, class {forced or 'as classed'}")
        assert_same_outputs(ref[:3], got[:3], f"{name}, emulator, class {forced or 'as classed'}")


def test_a_move_the_narrow_class_cannot_hold_ends_the_cell_with_an_error(monkeypatch):
    """The host never classes such a launch NARROW; should the rule ever be wrong, the kernel must say so
    (ERR_MOVE_RANGE -> MODLE_HIP_ERR_STATE) instead of truncating a move.  The NARROW emulator build is made
    to run a configuration whose moves are 70 kb per epoch."""
    import ctypes as C

    from modle_amd import api, synthetic
    from modle_amd.params import CellResult, Config, Task
    from phase_backend import emu_lib

    cfg = api.make_config(num_cells=4, rev_extrusion_speed=70000, fwd_extrusion_speed=70000,
                          rev_extrusion_speed_set=1, fwd_extrusion_speed_set=1, max_burnin_epochs=50)
    ch = synthetic.synthetic_chromosome("chrFast", 6_000_000, seed=3)
    stp_a, stp_i = api.barrier_stps(cfg, ch["bar_occupancy"])
    tasks = api.slice_tasks(api.make_tasks(cfg, ch["name"], ch["size"], 0, ch["size"]), 0, 1)
    nrows, ncols = api.matrix_shape(cfg, ch["size"])
    L = emu_lib()  # the NARROW build
    u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    L.emu_simulate_interval.argtypes = [C.POINTER(Config), C.c_uint64, C.c_uint64, u64p, u8p, f64p, f64p, C.c_size_t,
                                        C.POINTER(Task), C.c_size_t, u32p, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(CellResult)]
    L.emu_simulate_interval.restype = C.c_int
    contacts = np.zeros(nrows * ncols + 1, dtype=np.uint32)
    missed = C.c_uint64(0)
    res = (CellResult * 1)()

    def run():
        return L.emu_simulate_interval(C.byref(cfg), 0, ch["size"], np.ascontiguousarray(ch["bar_pos"], dtype=np.uint64),
                                       np.ascontiguousarray(ch["bar_dir"], dtype=np.uint8),
                                       np.ascontiguousarray(stp_a), np.ascontiguousarray(stp_i), len(ch["bar_pos"]),
                                       tasks, 1, contacts, nrows, ncols, C.byref(missed), None, res)

    assert run() == api.ERR_ARG  # refused: the set-up is of the WIDE class
    monkeypatch.setenv("MODLE_EMU_FORCE_NARROW", "1")
    assert run() == api.ERR_STATE  # run all the same: the first move beyond the limit ends the cell


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["chr20mb_barriers", "dense_barriers_trials", "many_rebinds_per_epoch"])
def test_gpu_cells_are_identical_in_both_classes(oracle, monkeypatch, name):
    from modle_amd import api

    case = build_case(name)
    cfg, ch = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, min(6, len(case["tasks"])))
    track = bool(cfg.track_1d_lef_position)
    ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                   case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)
    for forced, expect in ((None, 0), ("wide", 1)):
        if forced:
            monkeypatch.setenv("MODLE_HIP_SIZE_CLASS", forced)
        sim = api.Simulator(cfg, 0)
        try:
            gc, gm, go, gres = sim.simulate_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"],
                                                     case["stp_active"], case["stp_inactive"], tasks)
            info = sim.launch_info()
        finally:
            sim.close()
        assert info["size_class"] == expect, info
        assert_same_results(ref[3], gres, f"{name}, class {expect}")
        assert_same_outputs(ref[:3], (gc, gm, go if track else None), f"{name}, class {expect}")


@pytest.mark.parametrize("name", ["chr20mb_barriers", "many_rebinds_per_epoch", "rebinds_beyond_sort_buffer"])
def test_emulated_cells_with_the_geometry_of_the_12_wave_kernels(oracle, name):
    """The kernels exist for 8 and for 12 waves per workgroup (sim_launch.h; the host picks per launch); the
    12-wave builds halve the PRNG blocks and the LDS key buffers.  The same device code with that geometry on
    the lane emulator -- one of the cases re-inserts more units per epoch than the small buffers hold -- against
    the oracle."""
    import emu_sim
    from modle_amd import api

    case = build_case(name)
    cfg, ch = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, 1)
    track = bool(cfg.track_1d_lef_position)
    ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                   case["stp_inactive"], tasks, nthreads=2, track_occupancy=track)
    got = emu_sim.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                    case["stp_inactive"], tasks, case["nrows"], case["ncols"], track_occupancy=track,
                                    variant="w12")
    assert_same_results(ref[3], got[3], f"{name}, emulator, 12-wave geometry")
    assert_same_outputs(ref[:3], got[:3], f"{name}, emulator, 12-wave geometry")


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["8", "12"])
@pytest.mark.parametrize("klass", [None, "wide"])
def test_gpu_all_four_kernels_agree_with_the_oracle(oracle, monkeypatch, waves, klass):
    """narrow / wide x 8 / 12 waves per workgroup, forced by name, on a launch that fills no slot and on one
    that leaves tail helpers work to do"""
    from modle_amd import api

    monkeypatch.setenv("MODLE_HIP_WAVES", waves)
    if klass:
        monkeypatch.setenv("MODLE_HIP_SIZE_CLASS", klass)
    for name, n_cells in (("chr20mb_barriers", 40), ("many_rebinds_per_epoch", 4), ("dense_barriers_trials", 8)):
        case = build_case(name)
        cfg, ch = case["cfg"], case["chrom"]
        tasks = api.slice_tasks(case["tasks"], 0, min(n_cells, len(case["tasks"])))
        track = bool(cfg.track_1d_lef_position)
        ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                       case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)
        sim = api.Simulator(cfg, 0)
        try:
            gc, gm, go, gres = sim.simulate_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"],
                                                     case["stp_active"], case["stp_inactive"], tasks)
            info = sim.launch_info()
        finally:
            sim.close()
        assert info["waves_per_workgroup"] == int(waves) and info["size_class"] == (1 if klass else 0), info
        assert_same_results(ref[3], gres, f"{name}, {waves} waves, class {klass}")
        assert_same_outputs(ref[:3], (gc, gm, go if track else None), f"{name}, {waves} waves, class {klass}")


@pytest.mark.gpu
def test_gpu_a_fast_extrusion_speed_is_classed_wide_by_itself(oracle, monkeypatch):
    """moves of 70 kb per epoch do not fit 16 bits: the library runs the 32-bit kernels without being told"""
    from modle_amd import api, synthetic

    monkeypatch.delenv("MODLE_HIP_SIZE_CLASS", raising=False)
    cfg = api.make_config(num_cells=8, rev_extrusion_speed=70000, fwd_extrusion_speed=70000,
                          rev_extrusion_speed_set=1, fwd_extrusion_speed_set=1,
                          target_contact_density=0.05, max_burnin_epochs=300)
    ch = synthetic.synthetic_chromosome("chrFast", 9_000_000, seed=3)
    stp_a, stp_i = api.barrier_stps(cfg, ch["bar_occupancy"])
    tasks = api.slice_tasks(api.make_tasks(cfg, ch["name"], ch["size"], 0, ch["size"]), 0, 4)
    sim = api.Simulator(cfg, 0)
    try:
        gc, gm, go, gres = sim.simulate_interval(0, ch["size"], ch["bar_pos"], ch["bar_dir"], stp_a, stp_i, tasks)
        assert sim.launch_info()["size_class"] == 1
    finally:
        sim.close()
    oc, om, oo, ores = oracle.simulate_interval(cfg, 0, ch["size"], ch["bar_pos"], ch["bar_dir"], stp_a, stp_i, tasks,
                                                nthreads=4)
    assert np.array_equal(gc, oc) and gm == om and np.array_equal(go, oo)
    assert_same_results(ores, gres, "70 kb per epoch")
