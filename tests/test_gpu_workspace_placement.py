"""The placement search of the per-wave workspace (modle_hip.hip: place_workspace; round 5).

A launch runs up to 6 % faster or slower depending on which physical pages the driver handed out for the
workspace, and a streaming probe with the launch's geometry tells the placements apart; the library therefore
allocates a new workspace up to MODLE_HIP_WORKSPACE_TRIES times and keeps the candidate the probe likes best
(profiles/r05zr/workspace_placement_probe.txt).  Here: the search runs once per workspace, reports what it did
through `modle_hip_launch_info`, can be switched off, and -- the probe scribbles over the workspace it times --
changes no result."""
import pytest

from parity_cases import assert_same_outputs, assert_same_results, build_case


@pytest.mark.gpu
def test_gpu_the_search_runs_once_per_workspace_and_changes_no_result(oracle, monkeypatch):
    from modle_amd import api

    case = build_case("chr20mb_barriers")
    cfg, ch = case["cfg"], case["chrom"]
    tasks = api.slice_tasks(case["tasks"], 0, min(24, len(case["tasks"])))
    track = bool(cfg.track_1d_lef_position)
    ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                   case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)

    def run(sim):
        out = sim.simulate_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                    case["stp_inactive"], tasks)
        return out, sim.launch_info()

    monkeypatch.delenv("MODLE_HIP_WORKSPACE_TRIES", raising=False)
    sim = api.Simulator(cfg, 0)
    try:
        (gc, gm, go, gres), info = run(sim)
        assert 1 <= info["workspace_tries"] <= 24, info
        assert 0 < info["workspace_probe_us"] <= info["workspace_probe_worst_us"], info
        assert_same_results(ref[3], gres, "with the placement search")
        assert_same_outputs(ref[:3], (gc, gm, go if track else None), "with the placement search")
        # the second launch of the handle finds its workspace: nothing is probed again
        (gc2, gm2, go2, gres2), info2 = run(sim)
        assert (info2["workspace_tries"], info2["workspace_probe_us"]) == (info["workspace_tries"], info["workspace_probe_us"])
        assert_same_results(ref[3], gres2, "second launch")
    finally:
        sim.close()

    monkeypatch.setenv("MODLE_HIP_WORKSPACE_TRIES", "1")
    sim = api.Simulator(cfg, 0)
    try:
        (gc, gm, go, gres), info = run(sim)
        assert info["workspace_tries"] == 0 and info["workspace_probe_us"] == 0, info
        assert_same_results(ref[3], gres, "without the placement search")
        assert_same_outputs(ref[:3], (gc, gm, go if track else None), "without the placement search")
    finally:
        sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["chr20mb_barriers", "dense_barriers_trials", "many_rebinds_per_epoch"])
def test_gpu_cells_do_not_depend_on_what_the_workspace_held(oracle, monkeypatch, name):
    """Fresh device memory is usually zero, the placement probe writes into the candidates it times, earlier
    cells leave their state behind: MODLE_HIP_POISON_WORKSPACE=1 sets every byte of the workspace to 0xA5 before
    the launch, in both launch modes and with 8 and 12 waves per workgroup; results against the oracle."""
    from modle_amd import api

    case = build_case(name)
    cfg, ch = case["cfg"], case["chrom"]
    track = bool(cfg.track_1d_lef_position)
    monkeypatch.setenv("MODLE_HIP_POISON_WORKSPACE", "1")
    for n_cells, waves in ((6, "8"), (40, "8"), (40, "12")):
        tasks = api.slice_tasks(case["tasks"], 0, min(n_cells, len(case["tasks"])))
        ref = oracle.simulate_interval(cfg, ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"], case["stp_active"],
                                       case["stp_inactive"], tasks, nthreads=8, track_occupancy=track)
        monkeypatch.setenv("MODLE_HIP_WAVES", waves)
        sim = api.Simulator(cfg, 0)
        try:
            gc, gm, go, gres = sim.simulate_interval(ch["start"], ch["end"], ch["bar_pos"], ch["bar_dir"],
                                                     case["stp_active"], case["stp_inactive"], tasks)
        finally:
            sim.close()
        what = f"{name}, {len(tasks)} cells, {waves} waves, poisoned workspace"
        assert_same_results(ref[3], gres, what)
        assert_same_outputs(ref[:3], (gc, gm, go if track else None), what)
