"""The launch modes of the kernel (modle_amd/csrc/sim_pair.h) give the same results.

A cell is simulated by one wave, or -- when the launch leaves wave slots empty, and in the tail of
every launch -- by a main wave and its helper (plus a PRNG producer wave in the smallest launches).
The modes consume the cell's one PRNG stream at the same positions, so every output word and every
per-cell counter must be identical.  The whole-cell parity cases, the fuzz seeds and the
whole-genome sample test compare both modes with the oracle on small launches; the launches here
are the ones the oracle cannot follow: more tasks than wave slots (the queue drains, idle waves
attach to running cells as helpers), and the sizes around the switch between the modes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _launch(cfg, genome, env):
    from modle_amd import api, driver

    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        plan = driver.plan_genome(cfg, genome, 0, 1)
        sim = api.Simulator(cfg, 0)
        try:
            ids = driver.enqueue_plan(sim, cfg, plan)
            sim.launch()
            sim.wait()  # raises if any task reports a non-zero device status
            outs = []
            for entry, iid in zip(plan, ids):
                c, missed, occ = sim.copy_outputs(iid)
                res = sim.results(iid)
                outs.append((c, missed, occ, [(r.epochs, r.burnin_epochs, r.num_contacts, r.raws_consumed,
                                               tuple(r.prng_final)) for r in res]))
        finally:
            sim.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return outs


def _same(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[0], y[0]), f"{what}: interval {i}: contact matrix"
        assert x[1] == y[1], f"{what}: interval {i}: missed updates"
        assert np.array_equal(x[2], y[2]), f"{what}: interval {i}: occupancy"
        assert x[3] == y[3], f"{what}: interval {i}: per-cell results"


@pytest.mark.parametrize("cells", [1100, 2600])
def test_tail_helpers_do_not_change_the_results(cells):
    """More tasks than the fixed-role mode takes (1024), and more than wave slots (2048): every
    wave runs cells; the waves that find the queue empty attach to the cells still running."""
    from modle_amd import api, synthetic

    genome = [synthetic.synthetic_chromosome("chrA", 14_000_000, seed=11),
              synthetic.synthetic_chromosome("chrB", 5_000_000, seed=12)]
    cfg = api.make_config(num_cells=cells // 2, seed=3, track_1d_lef_position=1)
    plain = _launch(cfg, genome, {"MODLE_HIP_PAIRED": "0", "MODLE_HIP_TAIL_HELPERS": "0"})
    tail = _launch(cfg, genome, {"MODLE_HIP_PAIRED": "0", "MODLE_HIP_TAIL_HELPERS": "1"})
    auto = _launch(cfg, genome, {})
    _same(plain, tail, f"{cells} tasks, tail helpers")
    _same(plain, auto, f"{cells} tasks, default mode")


@pytest.mark.parametrize("cells", [7, 300, 700])
def test_fixed_roles_do_not_change_the_results(cells):
    """One, two and three-to-four main waves per workgroup (with and without the PRNG producer)."""
    from modle_amd import api, synthetic

    genome = [synthetic.synthetic_chromosome("chrA", 16_000_000, seed=21)]
    cfg = api.make_config(num_cells=cells, seed=5)
    plain = _launch(cfg, genome, {"MODLE_HIP_PAIRED": "0", "MODLE_HIP_TAIL_HELPERS": "0"})
    fixed = _launch(cfg, genome, {"MODLE_HIP_PAIRED": "1"})
    auto = _launch(cfg, genome, {})
    _same(plain, fixed, f"{cells} tasks, fixed roles")
    _same(plain, auto, f"{cells} tasks, default mode")
