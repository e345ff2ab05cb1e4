"""`modle simulate`-shaped front end over the C ABI.

    python -m modle_amd simulate -c hg38.chrom.sizes -b barriers.bed.xz -o out/prefix [options]

Option names, defaults and the derivation of the dependent parameters follow the reference's
`modle simulate` (reference: src/modle/cli.cpp:53-602 options, :886-1016 transform_args); the
task derivation is `run_simulate`'s (src/libmodle/cpu/scheduler_simulate.cpp:43-170) and the
outputs are the reference's: `<prefix>.cool` and, with the 1-D LEF position track on,
`<prefix>_lef_1d_occupancy.bw` (cli.cpp:867-882).  Everything heavy is native: parsing and
task generation in libmodle_hip.so (host), the simulation on the MI355X (one process per GPU;
under torch.distributed.run the cells are sharded over the ranks and the matrices are summed
with RCCL), the writers in libmodle_cooler.so.  `-t/--threads` is accepted and ignored."""
import argparse
import json
import os
import sys
import time

import numpy as np

from . import api, driver, genome
from .params import CS_LOOP, CS_NOISIFY, CS_TAD

STRATEGIES = {  # cli.hpp:63-72
    "tad-only": CS_TAD, "loop-only": CS_LOOP, "tad-plus-loop": CS_TAD | CS_LOOP,
    "tad-only-with-noise": CS_TAD | CS_NOISIFY, "loop-only-with-noise": CS_LOOP | CS_NOISIFY,
    "tad-plus-loop-with-noise": CS_TAD | CS_LOOP | CS_NOISIFY,
}


_DISTANCE_UNITS = {"bp": 1, "k": 10**3, "kb": 10**3, "kbp": 10**3, "m": 10**6, "mb": 10**6,
                   "mbp": 10**6, "g": 10**9, "gb": 10**9, "gbp": 10**9}  # cli_utils.hpp:86-96


def genomic_distance(text):
    """`20kb`, `1.5Mbp`, `3000000`: a number of base pairs with an optional unit, the reference's
    AsGenomicDistance transform (src/common/cli_utils_impl.hpp:304-362; the unit is case
    insensitive and the product must be a whole number)"""
    k = len(text)
    while k > 0 and text[k - 1].isalpha():
        k -= 1
    num, unit = text[:k], text[k:].lower()
    if not num:
        raise argparse.ArgumentTypeError(f"value {text} could not be converted")
    if not unit:
        try:
            v = int(num)
        except ValueError:
            raise argparse.ArgumentTypeError(f"unable to convert {text} to a number")
        if v < 0:
            raise argparse.ArgumentTypeError(f"unable to convert {text} to a number")
        return v
    if unit not in _DISTANCE_UNITS:
        raise argparse.ArgumentTypeError(f"{text[k:]} unit not recognized; valid units: "
                                         + ", ".join(sorted(_DISTANCE_UNITS)))
    try:
        m = float(num) * _DISTANCE_UNITS[unit]
    except ValueError:
        raise argparse.ArgumentTypeError(f"unable to convert {num} to a number")
    if m != int(m) or m < 0:
        raise argparse.ArgumentTypeError(f"Unable to convert {text} to a number of base-pairs "
                                         f"({m} is not an integral number)")
    return int(m)


def build_parser():
    ap = argparse.ArgumentParser(prog="modle_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("simulate", aliases=["sim"], help="simulate loop extrusion and write a .cool")
    io = p.add_argument_group("input / output")
    io.add_argument("-c", "--chrom-sizes", required=True)
    io.add_argument("-b", "--extrusion-barrier-file", required=True)
    io.add_argument("-g", "--genomic-intervals", "--chrom-subranges", default=None)
    io.add_argument("-f", "--force", action="store_true")
    io.add_argument("-o", "--output-prefix", required=True)
    io.add_argument("--assembly-name", default="unknown")
    io.add_argument("-q", "--quiet", action="store_true")
    io.add_argument("-v", "--verbose", action="store_true", help="accepted (the log is short anyway)")
    io.add_argument("--skip-output", action="store_true")
    io.add_argument("--log-model-internal-state", action="store_true",
                    help="write <prefix>_internal_state.log.gz: one line of statistics per task and "
                         "epoch (uses the diagnostic build libmodle_hip_statelog.so)")
    io.add_argument("--internal-state-max-epochs", type=int, default=4096,
                    help="epochs recorded per task with --log-model-internal-state")
    io.add_argument("--simulate-chromosomes-wo-barriers", dest="wo_barriers", action="store_true")
    io.add_argument("--skip-chromosomes-wo-barriers", dest="wo_barriers", action="store_false")
    io.add_argument("-t", "--threads", type=int, default=None, help="ignored (the GPU does the work)")
    io.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK or 0)")
    g = p.add_argument_group("model parameters (reference names; omitted => reference default)")
    for flags, dest, typ in [
        (("--lef-density", "--lefs-per-mbp"), "number_of_lefs_per_mbp", float),
        (("--avg-lef-processivity",), "avg_lef_processivity", genomic_distance),
        (("--probability-of-lef-bypass",), "probability_of_extrusion_unit_bypass", float),
        (("--extrusion-barrier-occupancy",), "extrusion_barrier_occupancy", float),
        (("--hard-stall-lef-stability-multiplier",), "hard_stall_lef_stability_multiplier", float),
        (("--soft-stall-lef-stability-multiplier",), "soft_stall_lef_stability_multiplier", float),
        (("--fwd-extrusion-speed",), "fwd_extrusion_speed", genomic_distance),
        (("--rev-extrusion-speed",), "rev_extrusion_speed", genomic_distance),
        (("--fwd-extrusion-speed-std",), "fwd_extrusion_speed_std", float),
        (("--rev-extrusion-speed-std",), "rev_extrusion_speed_std", float),
        (("--lef-bar-major-collision-prob",), "lef_bar_major_collision_pblock", float),
        (("--lef-bar-minor-collision-prob",), "lef_bar_minor_collision_pblock", float),
        (("--extrusion-barrier-bound-stp",), "barrier_occupied_stp", float),
        (("--extrusion-barrier-not-bound-stp",), "barrier_not_occupied_stp", float),
        (("--contact-sampling-interval",), "contact_sampling_interval", genomic_distance),
        (("-r", "--resolution"), "bin_size", genomic_distance),
        (("-w", "--diagonal-width"), "diagonal_width", genomic_distance),
        (("--tad-to-loop-contact-ratio",), "tad_to_loop_contact_ratio", float),
        (("--mu", "--genextr-location"), "genextreme_mu", float),
        (("--sigma", "--genextr-scale"), "genextreme_sigma", float),
        (("--xi", "--genextr-shape"), "genextreme_xi", float),
        (("--target-number-of-epochs",), "target_simulation_epochs", int),
        (("--target-contact-density",), "target_contact_density", float),
        (("--ncells",), "num_cells", int),
        (("--seed",), "seed", int),
        (("--burnin-target-epochs-for-lef-activation",), "burnin_target_epochs_for_lef_activation", int),
        (("--burnin-history-length",), "burnin_history_length", int),
        (("--burnin-smoothing-window-size",), "burnin_smoothing_window_size", int),
        (("--min-burnin-epochs",), "min_burnin_epochs", int),
        (("--max-burnin-epochs",), "max_burnin_epochs", int),
        (("--burnin-extr-speed-coefficient",), "burnin_speed_coefficient", float),
        (("--probability-normalization-factor",), "probability_normalization_factor", genomic_distance),
    ]:
        g.add_argument(*flags, dest=dest, type=typ, default=None)
    g.add_argument("--contact-sampling-strategy", choices=sorted(STRATEGIES), default=None)
    g.add_argument("-s", "--stopping-criterion", choices=["contact-density", "simulation-epochs"],
                   default="contact-density")
    g.add_argument("--track-1d-lef-position", dest="track_1d", action="store_true", default=None)
    g.add_argument("--no-track-1d-lef-position", dest="track_1d", action="store_false")
    g.add_argument("--skip-burnin", action="store_true")
    g.add_argument("--interpret-extrusion-barrier-name-as-not-bound-stp", dest="name_as_stp",
                   action="store_true")
    g.add_argument("--normalize-probabilities", dest="normalize", action="store_true", default=None)
    g.add_argument("--no-normalize-probabilities", dest="normalize", action="store_false")
    p.set_defaults(wo_barriers=False)
    e = sub.add_parser("evaluate", aliases=["eval"],
                       help="compare two .cool files stripe by stripe (modle_tools evaluate)")
    e.add_argument("-i", "--input-cooler", required=True, help="the matrix under test")
    e.add_argument("-r", "--reference-cooler", required=True)
    e.add_argument("-c", "--chrom-sizes", required=True, help="chromosomes to compare")
    e.add_argument("-w", "--diagonal-width", type=int, default=3_000_000)
    e.add_argument("-m", "--metric", choices=["pearson", "spearman", "rmse", "eucl_dist"],
                   default="pearson")
    e.add_argument("--exclude-zero-pixels", action="store_true")
    return ap


def config_from_args(a):
    """reference defaults, the options given on the command line, then Cli::transform_args"""
    over = {}
    for k, v in vars(a).items():
        if v is not None and hasattr(api.Config, k):
            over[k] = v
    if a.fwd_extrusion_speed is not None:
        over["fwd_extrusion_speed_set"] = 1
    if a.rev_extrusion_speed is not None:
        over["rev_extrusion_speed_set"] = 1
    if a.extrusion_barrier_occupancy is not None:
        if a.barrier_occupied_stp is not None:
            raise SystemExit("--extrusion-barrier-occupancy excludes --extrusion-barrier-bound-stp")
        over["extrusion_barrier_occupancy_set"] = 1
    if a.name_as_stp and a.barrier_not_occupied_stp is not None:
        raise SystemExit("--interpret-extrusion-barrier-name-as-not-bound-stp excludes "
                         "--extrusion-barrier-not-bound-stp")
    if a.contact_sampling_strategy is not None:
        over["contact_sampling_strategy"] = STRATEGIES[a.contact_sampling_strategy]
    if a.track_1d is not None:
        over["track_1d_lef_position"] = int(a.track_1d)
    if a.normalize is not None:
        over["normalize_probabilities"] = int(a.normalize)
    over["skip_burnin"] = int(a.skip_burnin)
    over["simulate_chromosomes_wo_barriers"] = int(a.wo_barriers)
    if a.stopping_criterion == "simulation-epochs":
        # cli.cpp:783-797 + simulation.cpp:1058-1074: epochs mode switches the density target off
        if a.target_simulation_epochs is None:
            raise SystemExit("--stopping-criterion=simulation-epochs requires --target-number-of-epochs")
        if a.target_contact_density is not None:
            raise SystemExit("--target-contact-density excludes --target-number-of-epochs")
        over["target_contact_density"] = -1.0
    elif a.target_simulation_epochs is not None:
        raise SystemExit("--stopping-criterion=contact-density excludes --target-number-of-epochs")
    mn, mx = over.get("min_burnin_epochs"), over.get("max_burnin_epochs")
    if mn is not None and mx is not None and mn > mx:
        raise SystemExit(f"--min-burnin-epochs={mn} cannot be greater than --max-burnin-epochs={mx}.")
    return api.make_config(**over)


def output_paths(prefix):
    return prefix + ".cool", prefix + "_lef_1d_occupancy.bw"


def state_log_path(prefix):
    return prefix + "_internal_state.log.gz"  # cli.cpp:871-875


def simulate(a, log=print):
    cfg = config_from_args(a)
    cool_path, bw_path = output_paths(a.output_prefix)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    if not a.skip_output and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(cool_path)), exist_ok=True)
        for p in (cool_path, bw_path if cfg.track_1d_lef_position else None):
            if p and os.path.exists(p) and not a.force:
                raise SystemExit(f"refusing to overwrite {p}: pass --force to overwrite")
    t0 = time.time()
    chroms, intervals, stats = genome.import_genome(cfg, a.chrom_sizes, a.extrusion_barrier_file,
                                                    a.genomic_intervals, a.name_as_stp)
    log(f"imported {len(chroms)} chromosomes, {len(intervals)} intervals, "
        f"{stats['barriers_imported']} barriers ({stats['barriers_without_strand']} without strand dropped)")
    plan = driver.plan_genome(cfg, intervals, rank, world)
    use_dist = world > 1
    if use_dist:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    sim = api.Simulator(cfg, device)
    try:
        if a.log_model_internal_state and not a.skip_output:
            # (with --skip-output the log would not be written: nothing is recorded, and the
            # default build of the library serves, like the reference accepts the combination)
            sim.enable_state_log(a.internal_state_max_epochs)
        ids = driver.enqueue_plan(sim, cfg, plan)
        n_tasks = sum(len(e["tasks"]) for e in plan if not e["skipped"])
        log(f"simulating {n_tasks} (interval, cell) tasks on device {device} (rank {rank} of {world})")
        sim.launch()
        sim.wait()
        log(f"simulation kernel: {sim.kernel_ms() / 1e3:.2f} s")
        if a.log_model_internal_state and not a.skip_output:
            import gzip

            path = state_log_path(a.output_prefix) if world == 1 else \
                f"{a.output_prefix}_internal_state.rank{rank}.log.gz"
            with gzip.open(path, "wt") as fh:
                fh.write(driver.STATE_LOG_HEADER)
                for entry, iid in zip(plan, ids):
                    if iid is None:
                        continue
                    iv = entry["interval"]
                    for k, task in enumerate(entry["tasks"]):
                        fh.writelines(driver.format_state_log(task, iv, len(iv["bar_pos"]),
                                                              sim.state_log(iid, k)))
            log(f"written {path}")
        matrices, occupancies = [], []
        for entry, iid in zip(plan, ids):
            if iid is None:
                matrices.append(None)
                occupancies.append(None)
                continue
            c, missed, occ = sim.copy_outputs(iid)
            if use_dist:
                import torch

                dev = torch.device("cuda", device)
                tc = torch.from_numpy(c.view(np.int32)).to(dev)
                to = torch.from_numpy(occ.view(np.int64)).to(dev) if occ is not None else None
                dist.reduce(tc, dst=0, op=dist.ReduceOp.SUM)
                if to is not None:
                    dist.reduce(to, dst=0, op=dist.ReduceOp.SUM)
                c = tc.cpu().numpy().view(np.uint32)
                occ = to.cpu().numpy().view(np.uint64) if to is not None else None
            total = int(c[:entry["nrows"] * entry["ncols"]].astype(np.int64).sum())
            if rank == 0 and total + missed > 0 and missed / (total + missed) >= 0.01:
                log(f"warning: {100.0 * missed / (total + missed):.2f}% missing interactions for "
                    f"{entry['interval']['name']}")  # simulation.cpp:153-157
            matrices.append(c)
            occupancies.append(occ)
    finally:
        sim.close()
    if rank == 0 and not a.skip_output:
        meta = json.dumps({k: v for k, v in vars(a).items() if v is not None and k != "command"},
                          sort_keys=True)
        driver.write_cooler(cool_path, cfg, plan, matrices, assembly=a.assembly_name,
                            generated_by="modle_amd (MI355X)", metadata_json=meta,
                            force_overwrite=a.force, chroms=chroms)
        log(f"written {cool_path}")
        if cfg.track_1d_lef_position:
            driver.write_bigwig(bw_path, cfg, plan, occupancies, chroms, force_overwrite=a.force)
            log(f"written {bw_path}")
    if use_dist:
        import torch.distributed as dist

        dist.destroy_process_group()
    log(f"done in {time.time() - t0:.1f} s")
    return 0


def evaluate_cmd(a):
    """prints one JSON object: {chromosome: {vertical: summary, horizontal: summary}}"""
    from . import evaluate as ev

    cfg = api.make_config()
    chroms, _, _ = genome.import_genome_text(cfg, genome.read_text(a.chrom_sizes), b"")
    res = ev.compare_coolers(a.reference_cooler, a.input_cooler, [n for n, _ in chroms],
                             a.diagonal_width, a.metric, a.exclude_zero_pixels)
    print(json.dumps({"metric": a.metric, "chromosomes": res}, indent=1))
    return 0


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.command in ("evaluate", "eval"):
        return evaluate_cmd(a)
    if a.log_model_internal_state and not a.skip_output:
        # the recording code lives in a diagnostic build of the library, chosen at import time
        from . import _lib

        if _lib._lib is not None and "statelog" not in _lib.SO_PATH:
            raise SystemExit("--log-model-internal-state needs MODLE_HIP_LIB=libmodle_hip_statelog.so "
                             "to be set before modle_amd is imported")
        if _lib._lib is None:
            _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libmodle_hip_statelog.so")
    log = (lambda *x: None) if a.quiet else (lambda *x: print(*x, file=sys.stderr, flush=True))
    return simulate(a, log)
