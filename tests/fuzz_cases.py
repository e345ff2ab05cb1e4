"""Seeded random simulation set-ups for the differential tests (GPU / emulator vs oracle).

`random_case(seed)` is the generator the committed regression seeds refer to; `random_case_v2`
widens the space (intervals that do not start at 0, barrier density, extrusion speeds and
their spread, LEF processivity, sampling interval, noise parameters)."""
import numpy as np

from modle_amd import api, synthetic


def _finish(cfg, chrom, kw, size):
    stp_active, stp_inactive = api.barrier_stps(cfg, chrom["bar_occupancy"])
    tasks = api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"], chrom["end"])
    nrows, ncols = api.matrix_shape(cfg, chrom["end"] - chrom["start"])
    return dict(cfg=cfg, chrom=chrom, stp_active=stp_active, stp_inactive=stp_inactive,
                tasks=tasks, nrows=nrows, ncols=ncols, kw=kw, size=size)


def random_case(seed):
    rng = np.random.default_rng(seed)
    size = int(rng.integers(300_000, 9_000_000))
    cfg_kw = dict(
        num_cells=int(rng.integers(4, 40)),
        seed=int(rng.integers(0, 2**31)),
        number_of_lefs_per_mbp=float(rng.choice([8.0, 20.0, 40.0, 80.0])),
        probability_of_extrusion_unit_bypass=float(rng.choice([0.0, 0.1, 0.5])),
        lef_bar_major_collision_pblock=float(rng.choice([1.0, 0.9])),
        lef_bar_minor_collision_pblock=float(rng.choice([0.0, 0.2])),
        soft_stall_lef_stability_multiplier=float(rng.choice([1.0, 3.0])),
        hard_stall_lef_stability_multiplier=float(rng.choice([5.0, 1.0])),
        contact_sampling_strategy=int(rng.choice([7, 6, 5, 4, 3, 2])),
        tad_to_loop_contact_ratio=float(rng.choice([5.0, 0.0, 1.0])),
        track_1d_lef_position=int(rng.integers(0, 2)),
        target_contact_density=float(rng.choice([1.0, 0.2])),
        diagonal_width=int(rng.choice([3_000_000, 500_000])),
        bin_size=int(rng.choice([5000, 2000, 20000])),
        skip_burnin=int(rng.random() < 0.25),
        simulate_chromosomes_wo_barriers=1,
        # the reference has no default bound; a cell with very few LEFs may never settle
        max_burnin_epochs=1500,
    )
    with_barriers = bool(rng.random() < 0.85)
    cfg = api.make_config(**cfg_kw)
    chrom = synthetic.synthetic_chromosome(f"chrF{seed}", size, seed=seed, with_barriers=with_barriers)
    return _finish(cfg, chrom, cfg_kw, size)


def random_case_v2(seed, pblock_pairs=None, spacings=(79_564, 25_000, 300_000),
                   lef_densities=(5.0, 20.0, 33.3, 64.0)):
    rng = np.random.default_rng(seed ^ 0x5EED)
    chrom_size = int(rng.integers(400_000, 12_000_000))
    # simulated interval: the whole chromosome, or a window that starts / ends inside it
    start, end = 0, chrom_size
    if rng.random() < 0.5:
        start = int(rng.integers(0, chrom_size // 2))
        end = int(rng.integers(start + chrom_size // 4, chrom_size + 1))
    cfg_kw = dict(
        num_cells=int(rng.integers(4, 24)),
        seed=int(rng.integers(0, 2**31)),
        number_of_lefs_per_mbp=float(rng.choice(list(lef_densities))),
        probability_of_extrusion_unit_bypass=float(rng.choice([0.0, 0.05, 0.1, 0.3])),
        lef_bar_major_collision_pblock=float(rng.choice([1.0, 0.95, 0.7])),
        lef_bar_minor_collision_pblock=float(rng.choice([0.0, 0.1, 0.3])),
        soft_stall_lef_stability_multiplier=float(rng.choice([1.0, 2.0])),
        hard_stall_lef_stability_multiplier=float(rng.choice([5.0, 2.0])),
        contact_sampling_strategy=int(rng.choice([7, 6, 5, 4, 3, 2])),
        tad_to_loop_contact_ratio=float(rng.choice([5.0, 0.0, 2.0])),
        track_1d_lef_position=int(rng.integers(0, 2)),
        target_contact_density=float(rng.choice([0.5, 0.1])),
        diagonal_width=int(rng.choice([3_000_000, 1_000_000])),
        bin_size=int(rng.choice([5000, 10000])),
        skip_burnin=int(rng.random() < 0.25),
        simulate_chromosomes_wo_barriers=1,
        max_burnin_epochs=1200,
        avg_lef_processivity=int(rng.choice([300_000, 100_000, 800_000])),
        contact_sampling_interval=int(rng.choice([50_000, 20_000, 150_000])),
        genextreme_sigma=float(rng.choice([12_500.0, 4_000.0])),
        genextreme_xi=float(rng.choice([0.001, 0.0, 0.2])),
    )
    speed = rng.choice(["default", "slow_fwd", "no_spread"])
    if speed == "slow_fwd":
        cfg_kw.update(rev_extrusion_speed=int(cfg_kw["bin_size"] // 2), rev_extrusion_speed_set=1,
                      fwd_extrusion_speed=int(cfg_kw["bin_size"] // 5), fwd_extrusion_speed_set=1)
    elif speed == "no_spread":
        cfg_kw.update(rev_extrusion_speed_std=0.0, fwd_extrusion_speed_std=0.0)
    if pblock_pairs is not None:
        major, minor = pblock_pairs[int(rng.integers(0, len(pblock_pairs)))]
        cfg_kw.update(lef_bar_major_collision_pblock=float(major),
                      lef_bar_minor_collision_pblock=float(minor))
    cfg = api.make_config(**cfg_kw)
    spacing = int(rng.choice(list(spacings)))
    full = synthetic.synthetic_chromosome(f"chrG{seed}", chrom_size, seed=seed,
                                          with_barriers=bool(rng.random() < 0.9), spacing=spacing)
    inside = (full["bar_pos"] >= start) & (full["bar_pos"] < end)
    chrom = dict(name=full["name"], size=chrom_size, start=start, end=end,
                 bar_pos=full["bar_pos"][inside], bar_dir=full["bar_dir"][inside],
                 bar_occupancy=full["bar_occupancy"][inside])
    return _finish(cfg, chrom, dict(cfg_kw, start=start, end=end, spacing=spacing), end - start)


def random_case_v3(seed):
    """blocking probabilities in {0, 1} only (no Bernoulli trial at the barriers: the device code
    takes its compacted-barrier path), barriers from dense to sparse, LEFs from very few to many"""
    return random_case_v2(seed ^ 0x7EA7, pblock_pairs=[(1, 0), (1, 0), (1, 1), (0, 1), (0, 0)],
                          spacings=(79_564, 8_000, 2_500, 300_000),
                          lef_densities=(1.0, 5.0, 20.0, 64.0))


def random_case_v4(seed):
    """the burn-in machinery and the stopping rules, which the first three generators keep at their
    defaults: minimum burn-in length, history length and smoothing window of the stability test,
    the activation ramp, the burn-in speed coefficient; TAD-to-loop ratio at 0 / very large;
    stopping on a number of epochs WITH burn-in (the unsigned `epoch - num_burnin_epochs` of
    simulation.cpp:930, see parity_cases.py); stall multipliers that make a release probability
    exactly zero (a LEF that draws nothing: the general form of release_lefs on the device) or
    larger than the unstalled one; windows that end at the chromosome's end or start at 0"""
    rng = np.random.default_rng(seed ^ 0xB0A71)
    chrom_size = int(rng.integers(300_000, 8_000_000))
    start, end = 0, chrom_size
    kind = rng.integers(0, 4)
    if kind == 1:  # window [0, x)
        end = int(rng.integers(chrom_size // 3, chrom_size))
    elif kind == 2:  # window [x, size)
        start = int(rng.integers(1, 2 * chrom_size // 3))
    elif kind == 3:  # inner window
        start = int(rng.integers(0, chrom_size // 2))
        end = int(rng.integers(start + chrom_size // 4, chrom_size + 1))
    hist = int(rng.choice([100, 12, 40, 250]))
    window = int(rng.choice([w for w in (5, 2, 9, 30) if w + 2 < hist]))
    cfg_kw = dict(
        num_cells=int(rng.integers(4, 16)),
        seed=int(rng.integers(0, 2**31)),
        number_of_lefs_per_mbp=float(rng.choice([5.0, 20.0, 40.0])),
        probability_of_extrusion_unit_bypass=float(rng.choice([0.0, 0.1, 0.3])),
        soft_stall_lef_stability_multiplier=float(rng.choice([1.0, 0.6, 2.0, np.inf])),
        hard_stall_lef_stability_multiplier=float(rng.choice([5.0, 1.0, np.inf])),
        contact_sampling_strategy=int(rng.choice([7, 6, 5, 4, 3, 2])),
        tad_to_loop_contact_ratio=float(rng.choice([5.0, 0.0, 1.0e9, 0.25])),
        track_1d_lef_position=int(rng.integers(0, 2)),
        diagonal_width=int(rng.choice([3_000_000, 600_000])),
        bin_size=int(rng.choice([5000, 10000])),
        simulate_chromosomes_wo_barriers=1,
        min_burnin_epochs=int(rng.choice([0, 0, 60, 400])),
        max_burnin_epochs=int(rng.choice([1000, 350, 120])),
        burnin_history_length=hist,
        burnin_smoothing_window_size=window,
        burnin_target_epochs_for_lef_activation=int(rng.choice([187, 20, 1, 600])),
        burnin_speed_coefficient=float(rng.choice([1.0, 0.5, 2.0])),
    )
    if cfg_kw["min_burnin_epochs"] > cfg_kw["max_burnin_epochs"]:
        cfg_kw["min_burnin_epochs"] = cfg_kw["max_burnin_epochs"] // 2
    if rng.random() < 0.3:
        # epochs mode, with burn-in in two cases out of three
        cfg_kw.update(target_contact_density=-1.0, target_simulation_epochs=int(rng.choice([25, 80, 200])),
                      skip_burnin=int(rng.random() < 0.33))
    else:
        cfg_kw.update(target_contact_density=float(rng.choice([0.3, 0.08])), skip_burnin=int(rng.random() < 0.15))
    # a LEF that can never be released once it is stalled may stop a cell with a handful of LEFs
    # from ever reaching its contact target (in the reference too): infinite multipliers only with
    # enough LEFs around, and never both
    few_lefs = cfg_kw["number_of_lefs_per_mbp"] * (end - start) / 1.0e6 < 20
    if np.isinf(cfg_kw["soft_stall_lef_stability_multiplier"]) and (
            few_lefs or np.isinf(cfg_kw["hard_stall_lef_stability_multiplier"])):
        cfg_kw["soft_stall_lef_stability_multiplier"] = 50.0
    if np.isinf(cfg_kw["hard_stall_lef_stability_multiplier"]) and few_lefs:
        cfg_kw["hard_stall_lef_stability_multiplier"] = 50.0
    cfg = api.make_config(**cfg_kw)
    spacing = int(rng.choice([79_564, 20_000, 250_000]))
    full = synthetic.synthetic_chromosome(f"chrH{seed}", chrom_size, seed=seed,
                                          with_barriers=bool(rng.random() < 0.9), spacing=spacing)
    inside = (full["bar_pos"] >= start) & (full["bar_pos"] < end)
    chrom = dict(name=full["name"], size=chrom_size, start=start, end=end,
                 bar_pos=full["bar_pos"][inside], bar_dir=full["bar_dir"][inside],
                 bar_occupancy=full["bar_occupancy"][inside])
    return _finish(cfg, chrom, dict(cfg_kw, start=start, end=end, spacing=spacing), end - start)
