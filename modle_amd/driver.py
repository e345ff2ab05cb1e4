"""Host driver standing in for `Simulation::run_simulate` (reference:
src/libmodle/cpu/scheduler_simulate.cpp:43-170) on top of the C ABI.

For every interval in genome order it derives the per-cell tasks exactly like the reference
(seed hash, one PRNG jump per cell, target-contact split), keeps the cells of this rank's shard,
registers the interval with the simulator and enqueues the tasks.  One launch then processes the
tasks of all intervals together (the reference's workers also drain one queue that mixes
intervals).  Cells are independent, so sharding them over ranks needs no data-path collective;
the per-rank contact matrices are summed afterwards (`reduce_outputs`).
"""
from . import api


def shard_bounds(num_cells, rank, world):
    """Contiguous cell-id range [lo, hi) owned by `rank` (any partition gives the same result:
    every cell's PRNG state is fixed by its cell id)."""
    base, rem = divmod(num_cells, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_genome(cfg, genome, rank=0, world=1):
    """Task derivation for every interval of `genome` (list of dicts, see synthetic.py).

    Returns a list of {interval, tasks (this rank's shard), nrows, ncols, skipped}."""
    plan = []
    task_id = 0
    for iv in genome:
        size = iv["end"] - iv["start"]
        nrows, ncols = api.matrix_shape(cfg, size)
        entry = {"interval": iv, "nrows": nrows, "ncols": ncols, "tasks": None, "skipped": False}
        if not cfg.simulate_chromosomes_wo_barriers and len(iv["bar_pos"]) == 0:
            # scheduler_simulate.cpp:111-124: intervals without barriers are skipped
            entry["skipped"] = True
            plan.append(entry)
            continue
        tasks = api.make_tasks(cfg, iv["name"], iv["size"], iv["start"], iv["end"], task_id)
        task_id += int(cfg.num_cells)
        lo, hi = shard_bounds(int(cfg.num_cells), rank, world)
        entry["tasks"] = api.slice_tasks(tasks, lo, hi)
        plan.append(entry)
    return plan


def interval_stps(cfg, iv):
    """self-transition probabilities of an interval's barriers: as imported (genome.py), or
    derived from BED-score-like occupancies (synthetic.py)"""
    if "bar_stp_active" in iv:
        return iv["bar_stp_active"], iv["bar_stp_inactive"]
    return api.barrier_stps(cfg, iv["bar_occupancy"])


def enqueue_plan(sim, cfg, plan, device_buffers=None):
    """Registers intervals + tasks of a plan with `sim`.  `device_buffers`: optional list of
    (contacts_ptr, occupancy_ptr) device pointers per plan entry (e.g. torch tensors)."""
    ids = []
    for k, entry in enumerate(plan):
        if entry["skipped"]:
            ids.append(None)
            continue
        iv = entry["interval"]
        stp_active, stp_inactive = interval_stps(cfg, iv)
        dc, do = (None, None) if device_buffers is None else device_buffers[k]
        iid = sim.add_interval(iv["start"], iv["end"], iv["bar_pos"], iv["bar_dir"], stp_active,
                               stp_inactive, dc, do)
        if len(entry["tasks"]) != 0:
            sim.submit(iid, entry["tasks"])
        ids.append(iid)
    return ids


def algorithmic_bytes(results, n_barriers, track_1d=True):
    """ALGORITHMIC HBM bytes of a set of finished cells of one interval:
    per simulated epoch 40 B per active LEF (read+write of rev/fwd position, binding epoch and
    the two ranks), 22 B per barrier (position, two transition probabilities, state read+write)
    and 64 B of PRNG state; per sampling event 8 B (contact read-modify-write) and, with the 1-D
    occupancy track, 16 B."""
    total = 0
    for r in results:
        total += 40 * r.sum_active_lefs + (22 * n_barriers + 64) * r.sim_epochs
        total += (8 + (16 if track_1d else 0)) * r.sampling_events
    return total


def read_missed(sim, ids):
    """missed-update counters of every registered interval (they accumulate over launches)"""
    return [None if iid is None else sim.copy_outputs(iid, want_contacts=False)[1] for iid in ids]


def verify_outputs(sim, cfg, plan, ids, matrix_sums, missed_delta, occupancy_sums=None):
    """Self-check of one launch's outputs (what a caller can assert without an oracle):

    * every registered contact is in the band matrix or was counted as a missed update
      (ContactMatrixDense::add, contact_matrix_dense_safe_impl.hpp:55-68);
    * with a target contact density, every cell stops exactly on its share of the target
      (stop condition, simulation.cpp:925-931; split, scheduler_simulate.cpp:129-141);
    * the 1-D occupancy track holds at most two entries per sampling event, an even number;
    * burn-in: all LEFs activated before it can end, and at least one epoch simulated.

    `matrix_sums[k]` / `occupancy_sums[k]`: sum of the interval's matrix / occupancy words of this
    launch (this rank's shard, before any reduce); `missed_delta[k]`: missed updates of this
    launch.  Uses the results of the LAST launch.  Raises AssertionError; returns a summary."""
    n_tasks = n_contacts = n_missed = 0
    for k, (entry, iid) in enumerate(zip(plan, ids)):
        if iid is None:
            continue
        name = entry["interval"]["name"]
        ntasks = len(entry["tasks"])
        res = sim.results(iid)
        last = res[len(res) - ntasks:]
        contacts = sum(r.num_contacts for r in last)
        assert int(matrix_sums[k]) + int(missed_delta[k]) == contacts, \
            f"{name}: {int(matrix_sums[k])} matrix + {int(missed_delta[k])} missed != {contacts} contacts"
        if cfg.target_contact_density >= 0:
            got = [r.num_contacts for r in last]
            want = [t.num_target_contacts for t in entry["tasks"]]
            assert got == want, f"{name}: per-cell contacts differ from the target split"
        if occupancy_sums is not None and cfg.track_1d_lef_position:
            occ = int(occupancy_sums[k])
            assert occ % 2 == 0 and occ <= 2 * sum(r.sampling_events for r in last), \
                f"{name}: occupancy track inconsistent with the sampling events"
        for r, t in zip(last, entry["tasks"]):
            if t.num_target_contacts != 0 and not cfg.skip_burnin and cfg.target_contact_density >= 0:
                assert r.burnin_epochs >= 1 and r.epochs >= r.sim_epochs >= 1, f"{name}: epoch counters"
        n_tasks += ntasks
        n_contacts += contacts
        n_missed += int(missed_delta[k])
    return {"tasks": n_tasks, "contacts": n_contacts, "missed_updates": n_missed}


def _chroms_of_plan(plan):
    chroms = []
    for entry in plan:
        iv = entry["interval"]
        if not chroms or chroms[-1][0] != iv["name"]:
            chroms.append((iv["name"], int(iv["size"])))
    return chroms


def write_cooler(path, cfg, plan, matrices, assembly="unknown", generated_by="modle-hip",
                 metadata_json="", force_overwrite=False, chroms=None):
    """Writes the (reduced) contact matrices of a plan to a cooler file the way the reference's
    IO thread does (simulation.cpp:117-168, 217-269): every chromosome of the genome is in the
    file, intervals are appended in genome order, skipped intervals and intervals without a
    matrix contribute no pixels.  `matrices[k]`: band matrix of plan entry k (uint32,
    nrows * ncols [+1] words, layout of modle_hip_interval_outputs) or None.
    `chroms`: [(name, size)] of the WHOLE genome (chrom.sizes order).  The reference creates the
    file from genome.chromosomes() (init_cooler_file, simulation.cpp:205-214), so a chromosome that
    --genomic-intervals leaves out still has its bins; without `chroms` only the chromosomes of
    the plan are known."""
    from . import cooler

    if chroms is None:
        chroms = _chroms_of_plan(plan)
    with cooler.CoolerWriter(path, chroms, int(cfg.bin_size), assembly=assembly,
                             generated_by=generated_by, metadata_json=metadata_json,
                             force_overwrite=force_overwrite) as w:
        for entry, m in zip(plan, matrices):
            if entry["skipped"] or m is None:
                continue
            iv = entry["interval"]
            w.append(iv["name"], m, entry["nrows"], entry["ncols"], offset_bp=int(iv["start"]))


def write_bigwig(path, cfg, plan, occupancies, chroms=None, force_overwrite=False):
    """Writes the 1-D LEF occupancy of every simulated interval the way the reference's IO thread
    does (simulation.cpp:130-141, 170-197): every chromosome of the genome in the header, one
    range per interval in genome order with values = counts / max(counts) as float32, span = step =
    bin size.  `occupancies[k]`: uint64 counts of plan entry k (ncols words) or None."""
    from . import bigwig

    if chroms is None:
        chroms = _chroms_of_plan(plan)
    with bigwig.BigWigWriter(path, chroms, force_overwrite=force_overwrite) as w:
        for entry, occ in zip(plan, occupancies):
            if entry["skipped"] or occ is None or entry["ncols"] == 0:
                continue
            iv = entry["interval"]
            w.write_occupancy(iv["name"], occ[:entry["ncols"]], int(cfg.bin_size), int(iv["start"]))


STATE_LOG_HEADER = ("task_id\tepoch\tcell_id\tchrom\tstart\tend\tburnin\tbarrier_occupancy\t"
                    "num_active_lefs\tnum_stalls_rev\tnum_stalls_fwd\tnum_stalls_both\t"
                    "num_lef_bar_collisions\tnum_primary_lef_lef_collisions\t"
                    "num_secondary_lef_lef_collisions\tavg_loop_size\n")


def format_state_log(task, interval, n_barriers, records):
    """lines of the model-internal-state log for one task, in the column order of
    Simulation::dump_stats (simulation.cpp:1040-1054): ids, interval, burn-in flag, effective
    barrier occupancy (occupied / all), active LEFs, units stalled rev / fwd, LEFs stalled at both
    ends, LEF-BAR / primary / secondary collisions, mean loop size"""
    out = []
    for rec in records:
        epoch = int(rec[0]) & ((1 << 63) - 1)
        burnin = bool(int(rec[0]) >> 63)
        n = int(rec[2])
        occ = (int(rec[1]) / n_barriers) if n_barriers else float("nan")
        avg = int(rec[9]) / n if n else float("nan")
        out.append(f"{task.id}\t{epoch}\t{task.cell_id}\t{interval['name']}\t{interval['start']}\t"
                   f"{interval['end']}\t{'True' if burnin else 'False'}\t{occ!r}\t{n}\t{int(rec[3])}\t"
                   f"{int(rec[4])}\t{int(rec[5])}\t{int(rec[6])}\t{int(rec[7])}\t{int(rec[8])}\t{avg!r}\n")
    return out


# ---------------------------------------------------------------------------------------------
# what an N > 1 bench line says about itself (bench.py)
# ---------------------------------------------------------------------------------------------
def check_distinct_devices(devices, world):
    """`devices`: one (hostname, device index or uuid) pair per rank, all-gathered.  An RCCL job of N ranks
    on N GPUs has N distinct pairs; two ranks that ended up on one GPU (a launcher that did not set
    LOCAL_RANK, a CUDA_VISIBLE_DEVICES mask) would still finish and report N x the cells."""
    if len(devices) != world:
        raise RuntimeError(f"{len(devices)} device records for {world} ranks")
    if len(set(devices)) != world:
        raise RuntimeError(f"ranks share a GPU: {devices}")
    return True


def scaling_report(world, scaling, total_cells, cells_per_gpu, reduce_ms, prediction=None):
    """The keys an N > 1 line adds.  `prediction`: the parsed profiles/r05zs/scale_prediction.json
    (one-GPU rehearsal of the strong-scaling job: every rank's shard run one after the other), or None.
    Strong scaling at the predicted job size: the predicted speed-up over N = 1 (with the reduce serial
    behind the kernel); weak scaling: N minus the same serial reduce, i.e. what `value` should be a multiple of
    the N = 1 value by.  Always labelled a prediction: the measured one is value(N) / value(1), which only the
    driver, holding both lines, can compute."""
    out = {"scaling": scaling, "total_cells": int(total_cells), "cells_per_gpu": int(cells_per_gpu),
           "reduce_ms": reduce_ms,
           "scaling_means": ("strong: total_cells fixed, each GPU simulates total_cells / N cells of every chromosome "
                             "(north_star's '>= 6 x at 8 vs 1'; BASELINE configs[3] with --total-cells 16384)"
                             if scaling == "strong" else
                             "weak: cells_per_gpu fixed, the job grows with N; the reference splits a fixed number of "
                             "target contacts over the cells, so a cell of the N-GPU job samples 1/N of the contacts "
                             "(compare cell_epochs_per_s, not only value)")}
    pred = None
    if prediction is not None and world > 1:
        w = prediction.get("worlds", {}).get(str(world))
        if w is not None:
            if scaling == "strong" and int(prediction.get("total_cells", -1)) == int(total_cells):
                pred = {"predicted_speedup": w["predicted_speedup_with_serial_reduce"],
                        "predicted_speedup_kernel_only": w["predicted_speedup_kernel_only"]}
            elif scaling == "weak":
                single = prediction.get("single", {})
                k1 = single.get("kernel_ms") if isinstance(single, dict) else None
                if k1:
                    red = prediction.get("reduce_ms_at_one_xgmi_link", 0.0)
                    # (weak scaling of independent cells: N x, less the serial reduce of one launch)
                    pred = {"predicted_speedup": world * w["max_kernel_ms"] / (w["max_kernel_ms"] + red),
                            "predicted_speedup_kernel_only": float(world)}
            if pred is not None:
                pred["prediction_source"] = ("profiles/r05zs/scale_prediction.json (one-GPU rehearsal, "
                                             "unmeasured on multi-GPU hardware)")
    out["predicted"] = pred
    return out
