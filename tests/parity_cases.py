"""Shared inputs for the whole-cell parity tests (oracle vs emulated device code vs GPU)."""
import numpy as np

from modle_amd import api, synthetic

# name -> (chrom size, Config overrides, with barriers, barrier spacing)
CASES = {
    # BASELINE.json configs[0]: 5 Mb, 64 cells, 16 LEFs/Mb, no barriers
    "config0_5mb_nobarriers": dict(size=5_000_000, barriers=False,
                                   cfg=dict(num_cells=64, number_of_lefs_per_mbp=16.0,
                                            simulate_chromosomes_wo_barriers=1)),
    # barriers + default probabilities (bypass 0.1 => secondary / fix paths are exercised)
    "chr20mb_barriers": dict(size=20_000_000, barriers=True, cfg=dict(num_cells=512)),
    # collision-heavy stress shaped like BASELINE.json configs[4]
    "chr12mb_dense_softstall": dict(size=12_000_000, barriers=True,
                                    cfg=dict(num_cells=512, number_of_lefs_per_mbp=64.0,
                                             lef_bar_minor_collision_pblock=0.3,
                                             soft_stall_lef_stability_multiplier=2.0)),
    # no noise, loop contacts only, no 1-D occupancy track, bypass disabled
    "chr8mb_loop_only": dict(size=8_000_000, barriers=True,
                             cfg=dict(num_cells=256, contact_sampling_strategy=4,
                                      track_1d_lef_position=0,
                                      probability_of_extrusion_unit_bypass=0.0)),
    # stopping on epochs instead of contact density is not derived by transform_config here;
    # burn-in skipped: every LEF is bound in epoch 0 (full-sort path of the ranking)
    "chr6mb_skip_burnin": dict(size=6_000_000, barriers=True,
                               cfg=dict(num_cells=128, skip_burnin=1)),
    # edge cases ------------------------------------------------------------------------------
    # 60 kb: compute_num_lefs rounds to a single LEF (ranking, scans and windows of width 1).
    # Burn-in is skipped: the loop-size history of a single LEF does not settle, and with the
    # default max_burnin_epochs (none) the reference's burn-in would not end either.
    "tiny_single_lef": dict(size=60_000, barriers=True,
                            cfg=dict(num_cells=8, simulate_chromosomes_wo_barriers=1,
                                     skip_burnin=1)),
    # more cells than target contacts: most cells get num_target_contacts = 0 and stop at the
    # first stop-condition check after init_states (scheduler_simulate.cpp:129-141, 234)
    "zero_target_cells": dict(size=400_000, barriers=True,
                              cfg=dict(num_cells=4096, target_contact_density=0.05)),
    # stop on a number of epochs instead of a contact density, TAD contacts only.  (With burn-in
    # the unsigned `epoch - num_burnin_epochs` of the reference's stop condition underflows as
    # soon as the first epoch needs two burn-in rounds, simulation.cpp:930 + 866-894, and the
    # cell ends after one epoch; that behaviour is reproduced, but it makes a poor test.)
    "epochs_stop_tad_only": dict(size=3_000_000, barriers=True,
                                 cfg=dict(num_cells=16, target_contact_density=-1.0,
                                          target_simulation_epochs=40, skip_burnin=1,
                                          contact_sampling_strategy=3)),
    # Bernoulli trials of LEF-BAR detection beyond one PRNG block per batch of 64 units (a barrier
    # every 100 bp, fractional blocking probabilities): the batch is resolved in rounds
    "dense_barriers_trials": dict(size=2_000_000, barriers=True, spacing=100,
                                  cfg=dict(num_cells=8, number_of_lefs_per_mbp=64.0, skip_burnin=1,
                                           target_contact_density=0.02,
                                           lef_bar_major_collision_pblock=0.7,
                                           lef_bar_minor_collision_pblock=0.4)),
    # ... and beyond one block for a single unit (a barrier every 3 bp: ~800 barriers within one
    # move): that unit is replayed sequentially
    "ultra_dense_barriers_trials": dict(size=300_000, barriers=True, spacing=3,
                                        cfg=dict(num_cells=4, number_of_lefs_per_mbp=40.0,
                                                 skip_burnin=1, target_contact_density=0.05,
                                                 diagonal_width=300_000,
                                                 lef_bar_major_collision_pblock=0.5,
                                                 lef_bar_minor_collision_pblock=0.5)),
    # more LEFs released in one epoch (~1500 of 2400: processivity of 8 kb against 5 kb moved per
    # epoch) than the LDS list of released LEFs holds (1024): the overflow path of release_lefs
    # and the sweeping form of select_and_bind_lefs
    "mass_release": dict(size=120_000_000, barriers=True,
                         cfg=dict(num_cells=4, skip_burnin=1, avg_lef_processivity=8000,
                                  target_contact_density=0.0004)),
    # 2 200 LEFs with a burn-in (ended by --max-burnin-epochs after ~75 evaluations of the loop-size
    # statistics): the statistics restore the LEF-id order in LDS in windows of 1 024 ids (round 4) --
    # here two full windows and a partial one, the fold running across their borders
    "burnin_three_windows": dict(size=110_000_000, barriers=True,
                                 cfg=dict(num_cells=2048, max_burnin_epochs=260)),
    # 300-450 LEFs released and bound again per epoch (1 300 LEFs at a processivity of 25 kb): more keys
    # than the one-sweep rank update held before round 4 (256), within what it holds now (511, 16-bit
    # per-key counts): the regime of BASELINE configs[4] on the large chromosomes
    "many_rebinds_per_epoch": dict(size=65_000_000, barriers=True,
                                   cfg=dict(num_cells=4, skip_burnin=1, avg_lef_processivity=25000,
                                            target_contact_density=0.004)),
    # 600-900 re-inserted units per epoch and direction (2 400 LEFs at a processivity of 25 kb): more
    # keys than the LDS sort buffer holds -- the one-sweep rank update keeps them in the generator's
    # ring, whose contents wait in device memory and must come back bit for bit (the draws that follow
    # read them): what BASELINE configs[4] does on chr1-chr12
    "rebinds_beyond_sort_buffer": dict(size=120_000_000, barriers=True,
                                       cfg=dict(num_cells=4, skip_burnin=1, avg_lef_processivity=25000,
                                                target_contact_density=0.012)),
    # the same through a burn-in (ended by --max-burnin-epochs): in helper-wave mode the generator and
    # its ring are with the helper while the main wave ranks the units of a burn-in epoch -- the ring is
    # not the main wave's to borrow, the update takes the general form -- and with one wave per cell the
    # statistics' windows and the borrowed ring alternate in the same LDS
    "rebinds_beyond_sort_buffer_burnin": dict(size=120_000_000, barriers=True,
                                              cfg=dict(num_cells=4, avg_lef_processivity=25000,
                                                       max_burnin_epochs=60, target_contact_density=0.002)),
    # the same with BASELINE configs[4]'s parameters on 190 Mb (12 160 LEFs): ~300 released LEFs AND
    # several hundred units that went past a stalled neighbour per epoch and direction
    "dense_stress_rebinds_and_displaced": dict(size=190_000_000, barriers=True,
                                               cfg=dict(num_cells=4, skip_burnin=1, number_of_lefs_per_mbp=64.0,
                                                        lef_bar_minor_collision_pblock=0.3,
                                                        soft_stall_lef_stability_multiplier=2.0,
                                                        target_contact_density=0.025)),
    # more LEFs than the LDS id filters have bits (32768): ids that share a bit pass the filters
    # together -- the release candidates' and the partner lookups' sweeps then store a few ranks
    # nobody asked for, and nothing else may change
    "many_lefs_hashed_filters": dict(size=50_000_000, barriers=True,
                                     cfg=dict(num_cells=4, skip_burnin=1, number_of_lefs_per_mbp=700.0,
                                              target_contact_density=0.02)),
    # a 4 Mb window that ends 10 Mb below the 32-bit position limit of the device layout, on a
    # chromosome longer than any real one: positions above 2^31 (the 64-bit scans of the move
    # adjustment, saturating key arithmetic in LEF-BAR detection)
    "window_near_position_limit": dict(size=4_290_000_000, barriers=True, window=(4_280_000_000, 4_284_000_000),
                                       cfg=dict(num_cells=64, diagonal_width=1_000_000)),
}


def build_case(name):
    spec = CASES[name]
    cfg = api.make_config(**spec["cfg"])
    if "window" in spec:
        # barriers only where the window is (a genome-scale barrier set is not needed)
        start, end = spec["window"]
        local = synthetic.synthetic_chromosome("chrT", end - start, with_barriers=spec["barriers"])
        chrom = dict(name="chrT", size=spec["size"], start=start, end=end,
                     bar_pos=local["bar_pos"] + np.uint64(start), bar_dir=local["bar_dir"],
                     bar_occupancy=local["bar_occupancy"])
        stp_active, stp_inactive = api.barrier_stps(cfg, chrom["bar_occupancy"])
        tasks = api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"], chrom["end"])
        nrows, ncols = api.matrix_shape(cfg, end - start)
        return dict(cfg=cfg, chrom=chrom, stp_active=stp_active, stp_inactive=stp_inactive,
                    tasks=tasks, nrows=nrows, ncols=ncols)
    chrom = synthetic.synthetic_chromosome("chrT", spec["size"], with_barriers=spec["barriers"],
                                           spacing=spec.get("spacing", synthetic.BARRIER_SPACING_BP))
    stp_active, stp_inactive = api.barrier_stps(cfg, chrom["bar_occupancy"])
    tasks = api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"], chrom["end"])
    nrows, ncols = api.matrix_shape(cfg, chrom["end"] - chrom["start"])
    return dict(cfg=cfg, chrom=chrom, stp_active=stp_active, stp_inactive=stp_inactive,
                tasks=tasks, nrows=nrows, ncols=ncols)


def assert_same_results(res_a, res_b, what):
    assert len(res_a) == len(res_b)
    for i, (a, b) in enumerate(zip(res_a, res_b)):
        for field in ("epochs", "burnin_epochs", "num_contacts", "raws_consumed",
                      "sum_active_lefs", "sampling_events", "sim_epochs"):
            assert getattr(a, field) == getattr(b, field), f"{what}: cell {i}: {field}"
        # State::rand_eng at return (SURVEY.md section 8d lists the final PRNG state in the gate)
        assert list(a.prng_final) == list(b.prng_final), f"{what}: cell {i}: prng_final"


def assert_same_outputs(a, b, what):
    ca, ma, oa = a
    cb, mb, ob = b
    assert ma == mb, f"{what}: missed updates {ma} != {mb}"
    assert np.array_equal(ca, cb), f"{what}: contact matrices differ"
    if oa is not None and ob is not None:
        assert np.array_equal(oa, ob), f"{what}: 1-D occupancy differs"


def launch_modes():
    """The two ways the kernel runs a cell (MODLE_HIP_PAIRED, read at every launch): "0" one wave
    per cell, "1" a main wave and its helper (modle_amd/csrc/sim_pair.h).  Left alone the library
    picks by the number of tasks, so a parity test states the mode and runs both: yields the mode
    with the variable set, and restores it."""
    import os

    old = os.environ.get("MODLE_HIP_PAIRED")
    try:
        for mode in ("0", "1"):
            os.environ["MODLE_HIP_PAIRED"] = mode
            yield mode
    finally:
        if old is None:
            os.environ.pop("MODLE_HIP_PAIRED", None)
        else:
            os.environ["MODLE_HIP_PAIRED"] = old
