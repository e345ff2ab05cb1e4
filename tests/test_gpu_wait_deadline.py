"""Every wait is bounded (include/modle_hip.h: modle_hip_wait, modle_hip_set_wait_timeout).

The reference's workers never wait for each other and poll `_ctx` once per epoch
(simulation.cpp:933, scheduler_simulate.cpp:264-270); here a cell may be served by up to three
waves that hand work to each other through words in LDS (sim_pair.h, sim_helper.h), and a lost
hand-over would leave a wave spinning for ever.  So: every spin loop reads the host's abort word,
and modle_hip_wait raises it when the launch has run into its deadline.

* a helper that withholds a signal (MODLE_HIP_TEST_FAULT=stuck_helper: test-only switch, read at
  the launch) costs the launch, not the box: modle_hip_wait returns MODLE_HIP_ERR_TIMEOUT soon after
  the deadline, the kernel has drained, and the same handle simulates the same cells correctly
  after modle_hip_reset;
* the deadline alone (no fault, a launch that simply takes longer) behaves the same way.
"""
import os
import time

import numpy as np
import pytest

from parity_cases import assert_same_outputs, assert_same_results, build_case

pytestmark = pytest.mark.gpu


def _run(sim, case, tasks):
    ch = case["chrom"]
    iid = sim.add_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                           case["stp_inactive"])
    sim.submit(iid, tasks)
    sim.launch()
    sim.wait()
    return iid


@pytest.mark.parametrize("paired", ["1", "0"])
def test_stuck_helper_costs_the_launch_not_the_box(oracle, monkeypatch, paired):
    """paired = "1": fixed trios (main / helper / PRNG producer); "0": one wave per cell, the idle
    waves of this small launch attach as tail helpers at once."""
    from modle_amd import api

    case = build_case("chr20mb_barriers")
    tasks = api.slice_tasks(case["tasks"], 0, 12)
    cfg, ch = case["cfg"], case["chrom"]
    monkeypatch.setenv("MODLE_HIP_PAIRED", paired)
    sim = api.Simulator(cfg, 0)
    try:
        deadline = 4.0
        sim.set_wait_timeout(deadline)
        monkeypatch.setenv("MODLE_HIP_TEST_FAULT", "stuck_helper")
        t0 = time.time()
        with pytest.raises(api.ModleHipError) as e:
            _run(sim, case, tasks)
        waited = time.time() - t0
        assert e.value.code == api.ERR_TIMEOUT, str(e.value)
        assert "deadline" in str(e.value)
        # the abort word is read every 1024 naps of a spin loop: the drain takes milliseconds
        assert deadline <= waited < deadline + 5.0, waited
        # the launch is over: the handle accepts a reset and simulates the same cells correctly
        monkeypatch.delenv("MODLE_HIP_TEST_FAULT")
        sim.reset()
        sim.set_wait_timeout(600.0)
        iid = _run(sim, case, tasks)
        got = sim.copy_outputs(iid)
        gres = sim.results(iid)
    finally:
        sim.close()
    oc, om, oo, ores = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"],
                                                case["stp_active"], case["stp_inactive"], tasks, nthreads=4)
    assert_same_outputs(got, (oc, om, oo), f"after the aborted launch (paired={paired})")
    assert_same_results(gres, ores, f"after the aborted launch (paired={paired})")
    print(f"paired={paired}: stuck helper, wait returned ERR_TIMEOUT after {waited:.2f} s (deadline {deadline} s)")


def test_deadline_alone_aborts_a_long_launch_and_the_handle_survives(oracle):
    from modle_amd import api, driver, synthetic

    genome = [synthetic.synthetic_chromosome("chrBig", 80_000_000, seed=5)]
    cfg = api.make_config(num_cells=32768, seed=11)
    entry = driver.plan_genome(cfg, genome)[0]
    sim = api.Simulator(cfg, 0)
    try:
        sim.set_wait_timeout(0.3)
        ids = driver.enqueue_plan(sim, cfg, [dict(entry, tasks=api.slice_tasks(entry["tasks"], 0, 24000))])
        t0 = time.time()
        sim.launch()  # seconds of kernel when left alone
        with pytest.raises(api.ModleHipError) as e:
            sim.wait()
        waited = time.time() - t0
        assert e.value.code == api.ERR_TIMEOUT and waited < 3.0, (str(e.value), waited)
        # what was registered before the abort is accounted for, like after modle_hip_cancel
        c, missed, _ = sim.copy_outputs(ids[0])
        res = sim.results(ids[0])
        assert int(c.astype(np.int64).sum()) + missed == sum(r.num_contacts for r in res)
        sim.reset()
        sim.set_wait_timeout(600.0)
        small = synthetic.synthetic_chromosome("chrSmall", 4_000_000, seed=6)
        stp_a, stp_i = api.barrier_stps(cfg, small["bar_occupancy"])
        tasks = api.slice_tasks(api.make_tasks(cfg, small["name"], small["size"], 0, small["size"]), 0, 6)
        gc, gm, go, gres = sim.simulate_interval(0, small["size"], small["bar_pos"], small["bar_dir"], stp_a,
                                                 stp_i, tasks)
    finally:
        sim.close()
    oc, om, oo, ores = oracle.simulate_interval(cfg, 0, small["size"], small["bar_pos"], small["bar_dir"], stp_a,
                                                stp_i, tasks, nthreads=4)
    assert np.array_equal(gc, oc) and gm == om and np.array_equal(go, oo)
    assert [(r.epochs, r.raws_consumed) for r in gres] == [(r.epochs, r.raws_consumed) for r in ores]


def test_destroy_with_a_launch_in_flight_is_bounded(monkeypatch):
    """modle_hip_destroy used to hipStreamSynchronize a launch that was still in flight: a caller going
    down on an error path (or an interpreter shutting down with a Simulator alive) then waited for ever
    behind a stuck kernel.  Now: abort word, bounded drain, and only then the frees."""
    from modle_amd import api, driver, synthetic

    genome = [synthetic.synthetic_chromosome("chrBig", 80_000_000, seed=5)]
    cfg = api.make_config(num_cells=32768, seed=11)
    entry = driver.plan_genome(cfg, genome)[0]
    sim = api.Simulator(cfg, 0)
    driver.enqueue_plan(sim, cfg, [dict(entry, tasks=api.slice_tasks(entry["tasks"], 0, 24000))])
    sim.launch()  # seconds of kernel when left alone; never waited for
    t0 = time.time()
    sim.close()
    closed = time.time() - t0
    assert closed < 2.0, closed
    # ... and with a helper that withholds its signal (the spin loops leave on the abort word)
    monkeypatch.setenv("MODLE_HIP_TEST_FAULT", "stuck_helper")
    monkeypatch.setenv("MODLE_HIP_PAIRED", "1")
    case = build_case("chr20mb_barriers")
    sim = api.Simulator(case["cfg"], 0)
    ch = case["chrom"]
    iid = sim.add_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                           case["stp_inactive"])
    sim.submit(iid, api.slice_tasks(case["tasks"], 0, 12))
    sim.launch()
    time.sleep(0.5)
    t0 = time.time()
    sim.close()
    closed_stuck = time.time() - t0
    assert closed_stuck < 2.0, closed_stuck
    print(f"destroy with a launch in flight: {closed:.3f} s (long launch), {closed_stuck:.3f} s (stuck helper)")
