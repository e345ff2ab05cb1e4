"""Seeded random simulation set-ups for the differential tests (GPU / emulator vs oracle)."""
import numpy as np

from modle_amd import api, synthetic


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
    stp_active, stp_inactive = api.barrier_stps(cfg, chrom["bar_occupancy"])
    tasks = api.make_tasks(cfg, chrom["name"], chrom["size"], chrom["start"], chrom["end"])
    nrows, ncols = api.matrix_shape(cfg, chrom["end"] - chrom["start"])
    return dict(cfg=cfg, chrom=chrom, stp_active=stp_active, stp_inactive=stp_inactive,
                tasks=tasks, nrows=nrows, ncols=ncols, kw=cfg_kw, size=size)
