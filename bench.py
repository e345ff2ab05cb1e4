#!/usr/bin/env python3
"""Benchmark of the loop-extrusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload grch38|chr1] [--cells C]

A "step" is one pass of the hot path over one batch of synthetic input: every (chromosome, cell)
task of this rank's shard is simulated by the HIP kernel (one wavefront per cell) and, for N > 1,
the per-rank contact matrices are summed with RCCL.  Default workload = BASELINE.json configs[2]:
the whole GRCh38-shaped genome (24 chromosomes, synthetic barriers with the bundled BED's
statistics), 2048 cells per GPU, all parameters at the reference defaults, seed 0.  Cells are
sharded over ranks.  Default = weak scaling, 2048 cells per GPU (configs[3], 16384 cells, is the
8-GPU point); the reference splits a fixed number of target contacts over the cells
(scheduler_simulate.cpp:129-141), so a cell of the 8-GPU point samples an eighth of the contacts
of a cell of the 1-GPU point: `cell_epochs_per_s` is reported next to `value` for that reason.
`--scaling strong --total-cells N` keeps the workload fixed (N cells in total at every GPU count).

After the timed region the outputs of the last step are verified (driver.verify_outputs:
contact conservation per interval, per-cell targets, occupancy, device status): `"checked": true`.

Prints ONE JSON line on rank 0 (contract in the task statement): metric = simulated
genome-cells/s (whole job), plus `roofline` (algorithmic HBM bytes of the simulation kernel over
its HIP-event duration, against 8 TB/s) and `cpu_baseline` (the CPU oracle timed on the host
cores over a bounded sample of the same workload; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def launch_mode(info):
    """How the last launch ran its tasks, from the library's own record of it
    (modle_hip_last_launch_info; the rule is in modle_amd/csrc/modle_hip.hip),
    for the bench line's `config`."""
    if info["helper_waves"]:
        return ("main wave + helper + PRNG producer" if info["prng_producer_waves"] else "main wave + helper") + \
               " (launch leaves wave slots empty)"
    return "one wave per cell" + (", idle waves help in the tail of the launch" if info["tail_helpers"] else "")


def workspace_placement(info):
    """what the library's placement search did when it allocated the per-wave workspace (include/modle_hip.h:
    modle_hip_launch_info; once per handle, outside the timed region unless --warmup 0)"""
    tries = info.get("workspace_tries", 0)
    if not tries:
        return "first allocation (no search)"
    return {"candidates_probed": tries, "probe_ms_kept": info.get("workspace_probe_us", 0) / 1e3,
            "probe_ms_slowest": info.get("workspace_probe_worst_us", 0) / 1e3}


def size_class(info):
    return "wide (32-bit LEF ids and moves)" if info.get("size_class") else "narrow (16-bit LEF ids and moves)"


def measured_traffic(workload_key):
    """HBM bytes per launch of the simulation kernel, from the PMC passes committed under
    profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same workload,
    see profiles/README.md).  bench.py cannot collect hardware counters itself; None when no
    committed measurement matches the workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    entry = table.get(workload_key) if isinstance(table, dict) else None
    if not entry:
        return None, None
    # The figure is a committed measurement of ANOTHER run (PMC counters cannot be collected from
    # inside the benchmark): it describes the kernel it was measured on.  The entry records a hash
    # of the device sources (tools/csrc_hash.py); when the sources of this run hash differently the
    # figure is stale and the line says `traffic: null` rather than quote it.
    measured_on = entry.get("csrc_sha256")
    if measured_on is None or measured_on != csrc_hash():
        return None, (f"stale: {entry.get('source', 'profiles/')} was measured on device sources "
                      f"{str(measured_on)[:12]}, this run is {csrc_hash()[:12]}")
    return entry["bytes_per_launch"], f"{entry.get('source', 'profiles/')}; {entry.get('kernel', '')}"


def csrc_hash():
    """sha256 over the device sources of the simulation kernel (what `roofline.traffic` is tied to)"""
    from tools.csrc_hash import csrc_sha256

    return csrc_sha256(ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["grch38", "chr1", "grch38-dense"], default="grch38",
                    help="grch38: BASELINE configs[2] (the headline); chr1: configs[1]; grch38-dense: configs[4], the "
                         "collision-heavy stress (4096 cells, 64 LEFs/Mb, minor-collision trials, soft stalls) -- "
                         "with one GPU the shard of rank 0 of 8 (512 cells of every chromosome), which is what one "
                         "GPU of the 8-GPU node runs")
    ap.add_argument("--chrom", default=None,
                    help="diagnostic: restrict the grch38 workload to one chromosome (e.g. chr21), or synth:<bp>")
    ap.add_argument("--cells", type=int, default=None, help="cells per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--total-cells", type=int, default=None,
                    help="cells of the whole job with --scaling strong (default 2048 / 512)")
    ap.add_argument("--rng", choices=["exact", "philox"], default="exact",
                    help="exact: the reference's xoshiro256++ stream (default, every parity claim); "
                         "philox: the counter-based generator policy, a separate line that is not "
                         "bit-comparable with the reference")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL, default): the matrices are reduced on the GPUs over xGMI.  gloo: "
                         "the same per-interval ordering with the reduce done on host copies -- for "
                         "rehearsing the N > 1 path with several ranks on ONE GPU (tests)")
    ap.add_argument("--checksum-out", default=None,
                    help="rank 0 writes {interval: [sum, position-weighted sum]} of the final (reduced) "
                         "matrices and occupancy tracks of the last step to this JSON file")
    ap.add_argument("--poll-timeout", type=float, default=900.0,
                    help="deadline of one launch in seconds, single-GPU and distributed paths alike: past "
                         "it the launch is aborted (modle_hip_set_wait_timeout: the abort word is raised, "
                         "the kernel drains) and the job fails instead of hanging the box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=None)
    return ap.parse_args()


def usable_cores():
    """CPU cores this process can really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(cfg, genome, unit, sample_cells):
    """Times the CPU oracle (oracle/, kind "port") with one thread per usable host core over a
    bounded sample: `sample_cells` cells (default: three per thread, at most 256) of every chromosome
    of the workload, all (chromosome, cell) tasks in one thread pool."""
    from modle_amd import api
    from oracle import binding as oracle

    cores = usable_cores()
    sample = sample_cells or max(2, min(3 * cores, 256))  # ~3 cells per thread and chromosome
    jobs = []
    for iv in genome:
        if len(iv["bar_pos"]) == 0 and not cfg.simulate_chromosomes_wo_barriers:
            continue
        tasks = api.make_tasks(cfg, iv["name"], iv["size"], iv["start"], iv["end"])
        n = min(sample, len(tasks))
        stp_a, stp_i = api.barrier_stps(cfg, iv["bar_occupancy"])
        jobs.append((iv, api.slice_tasks(tasks, 0, n), stp_a, stp_i))
    oracle.lib().mo_set_rng_policy(1 if os.environ.get("MODLE_HIP_LIB") == "libmodle_hip_philox.so" else 0)
    t0 = time.perf_counter()
    epochs = 0
    for iv, tasks, stp_a, stp_i in jobs:
        _, _, _, res = oracle.simulate_interval(cfg, iv["start"], iv["end"], iv["bar_pos"],
                                                iv["bar_dir"], stp_a, stp_i, tasks,
                                                nthreads=cores,
                                                track_occupancy=bool(cfg.track_1d_lef_position))
        epochs += sum(r.epochs for r in res)
    dt = time.perf_counter() - t0
    return {
        "value": sample / dt,
        "unit": unit,
        "cores": cores,
        "kind": "port",
        "sample": f"{sample} cells of each of the {len(jobs)} chromosomes "
                  f"({sample * len(jobs)} tasks, {epochs} epochs) in {dt:.1f} s",
    }


def main():
    args = parse_args()
    if args.rng == "philox":
        # the policy is a separate build of the library, chosen when modle_amd is imported
        os.environ["MODLE_HIP_LIB"] = "libmodle_hip_philox.so"
    # stdout carries exactly one line, the JSON result: everything libraries print there while
    # the job runs (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch

    from modle_amd import api, driver, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for N > 1")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()  # (rehearsal: the ranks may share a GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run the RCCL path runs even with one rank (same code as N > 1)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # small collectives
    if use_dist:
        # the job really is N ranks (and, over RCCL, N different GPUs): a launcher that lost LOCAL_RANK would
        # still finish and report N x the cells
        import socket

        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")
        props = torch.cuda.get_device_properties(dev)
        mine = (socket.gethostname(), str(getattr(props, "uuid", "")) or str(torch.cuda.current_device()),
                torch.cuda.current_device())
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if args.dist_backend == "nccl":
            driver.check_distinct_devices([(h, u) for h, u, _ in everyone], world)
            driver.check_distinct_devices([(h, i) for h, _, i in everyone], world)

    def reduce_to_rank0(t):
        """sum of the ranks' tensors into rank 0's (issued on the current stream)"""
        if args.dist_backend == "nccl":
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        else:
            h = t.cpu()  # (synchronises the current stream: the interval's cells have finished)
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                t.copy_(h)

    default_cells = 2048 if args.workload == "grch38" else 512
    cfg_overrides = {}
    plan_rank, plan_world = rank, world
    if args.workload == "grch38-dense":
        # BASELINE configs[4] is an 8-GPU job with a fixed number of cells: always strong scaling;
        # a single GPU runs the shard of rank 0 of 8 (tests/test_gpu_baseline_configs.py checks
        # exactly that launch against the oracle)
        args.scaling = "strong"
        args.total_cells = args.total_cells or 4096
        cfg_overrides = dict(number_of_lefs_per_mbp=64.0, lef_bar_minor_collision_pblock=0.3,
                             soft_stall_lef_stability_multiplier=2.0)
        if world == 1:
            plan_rank, plan_world = 0, 8
    if args.scaling == "strong":
        total_cells = args.total_cells or default_cells
        if total_cells < world:
            raise SystemExit("--total-cells must be at least the number of GPUs")
        cells_per_gpu = -(-total_cells // plan_world)  # largest shard
        cells_txt = f"{total_cells} cells in total (strong scaling)"
    else:
        cells_per_gpu = args.cells or default_cells
        total_cells = cells_per_gpu * world
        cells_txt = f"{cells_per_gpu} cells per GPU"
    if args.workload == "grch38" and args.chrom and args.chrom.startswith("synth:"):
        # diagnostic: one synthetic chromosome of the given length, barriers at the bundled file's mean spacing
        genome = [synthetic.synthetic_chromosome("chrS", int(args.chrom.split(":")[1]))]
        workload = f"synthetic {genome[0]['size']} bp interval (diagnostic), {cells_txt}, reference defaults"
        unit = "cells/s"
    elif args.workload == "grch38" and args.chrom:
        genome = synthetic.grch38_like(seed=42, chroms={args.chrom})
        workload = f"{args.chrom}-shaped interval only (diagnostic), {cells_txt}, reference defaults"
        unit = f"{args.chrom}-cells/s"
    elif args.workload == "grch38":
        genome = synthetic.grch38_like(seed=42)
        workload = (f"GRCh38-shaped genome (24 chromosomes, synthetic H1-like barriers: the bundled file's "
                    f"count per chromosome), "
                    f"{cells_txt}, reference defaults (BASELINE configs[2]/[3])")
        unit = "genome-cells/s"
    elif args.workload == "grch38-dense":
        genome = synthetic.grch38_like(seed=42)
        workload = (f"GRCh38-shaped genome, collision-heavy stress (BASELINE configs[4]): {cells_txt}, 64 LEFs/Mb, "
                    f"lef_bar_minor_collision_pblock 0.3, soft_stall_lef_stability_multiplier 2"
                    + (f"; this GPU runs the shard of rank 0 of 8 ({cells_per_gpu} cells of every chromosome)"
                       if world == 1 else ""))
        unit = "genome-cells/s"
    else:
        genome = synthetic.grch38_like(seed=42, chroms={"chr1"})
        workload = (f"chr1-shaped interval (248 956 422 bp, {len(genome[0]['bar_pos'])} synthetic barriers), "
                    f"{cells_txt}, reference defaults (BASELINE configs[1])")
        unit = "chr1-cells/s"
    cfg = api.make_config(num_cells=total_cells, seed=0, **cfg_overrides)

    plan = driver.plan_genome(cfg, genome, plan_rank, plan_world)
    # outputs live in torch tensors so that RCCL can reduce them in place
    buffers, tensors = [], []
    for entry in plan:
        if entry["skipped"]:
            buffers.append((None, None))
            tensors.append(None)
            continue
        c = torch.zeros(entry["nrows"] * entry["ncols"] + 1, dtype=torch.int32, device=dev)
        o = torch.zeros(entry["ncols"], dtype=torch.int64, device=dev)
        buffers.append((c.data_ptr(), o.data_ptr()))
        tensors.append((c, o))
    sim = api.Simulator(cfg, local_rank)
    sim.set_wait_timeout(args.poll_timeout)  # (a launch that hangs fails the job: api.ERR_TIMEOUT)
    ids = driver.enqueue_plan(sim, cfg, plan, buffers)  # uploads barriers, enqueues step 0
    stream = torch.cuda.current_stream(dev)

    kernel_ms = []
    reduce_ms = []      # per step: time the side stream spent in this rank's reduces (events around each pair)
    reduce_events = []  # of the step under way
    check = {}
    reduce_stream = torch.cuda.Stream(device=dev) if use_dist else None
    # largest (LEFs + barriers) first: the order in which modle_hip_launch starts the tasks
    reduce_order = sorted(
        (k for k, iid in enumerate(ids) if iid is not None),
        key=lambda k: -(int(plan[k]["tasks"][0].num_lefs if len(plan[k]["tasks"]) else 0)
                        + len(plan[k]["interval"]["bar_pos"])))

    timing = os.environ.get("MODLE_BENCH_TIMING", "") not in ("", "0")  # (diagnostic: host time per part of a step)

    # diagnostic: a switch the library reads at every launch, changed from step to step INSIDE one process --
    # two processes on one box differ by 2 % whatever they run (profiles/r04z), the steps of one by 0.1 %.
    #   MODLE_BENCH_ALTERNATE="MODLE_HIP_EXP=0,1"     (MODLE_BENCH_ALTERNATE_TAIL=1: "MODLE_HIP_TAIL_HELPERS=1,0")
    alternate = os.environ.get("MODLE_BENCH_ALTERNATE", "")
    if os.environ.get("MODLE_BENCH_ALTERNATE_TAIL", "") not in ("", "0"):
        alternate = "MODLE_HIP_TAIL_HELPERS=1,0"
    alt_var, alt_values = (alternate.split("=")[0], alternate.split("=")[1].split(",")) if alternate else (None, [])
    n_step = [0]

    def step(first, last=False):
        if alternate:
            os.environ[alt_var] = alt_values[n_step[0] % len(alt_values)]
            n_step[0] += 1
        t_a = time.perf_counter()
        if not first:
            for entry, iid in zip(plan, ids):
                if iid is not None:
                    sim.submit(iid, entry["tasks"])
        for t in tensors:
            if t is not None:
                t[0].zero_()
                t[1].zero_()
        if last:
            check["missed_before"] = driver.read_missed(sim, ids)
            check["matrix"] = [None] * len(tensors)
            check["occ"] = [None] * len(tensors)

        def own_sums(k):
            # this rank's own outputs, before the reduce folds the other ranks' into them (two
            # device-side sums per interval: ~1 ms of reads in all against seconds of simulation)
            check["matrix"][k] = tensors[k][0].sum(dtype=torch.int64)
            check["occ"][k] = tensors[k][1].sum()

        t_b = time.perf_counter()
        sim.launch(stream.cuda_stream)
        t_c = time.perf_counter()
        if use_dist:
            import torch.distributed as dist

            # Reduce every interval's matrix as soon as its last cell has finished, on a side
            # stream, while the kernel goes on with the smaller intervals.
            # The completion counters are host-mapped words: polling them touches no stream.
            # Collectives must be issued in the same order on every rank: the order is fixed (the
            # launch order of the tasks, largest interval first, which is also roughly the order
            # in which the intervals complete) and each rank waits for the next interval in it.
            deadline = time.monotonic() + args.poll_timeout
            for k in reduce_order:
                while not sim.interval_done(ids[k]):
                    if time.monotonic() > deadline:
                        # a faulted kernel never counts its intervals down: fail instead of spinning
                        # (the peers then fail in their collective instead of waiting for ever)
                        # ... and leave without the unbounded waits of an orderly shutdown: the abort word is
                        # raised (a kernel that still listens drains), the process exits non-zero at once
                        sim.cancel()
                        print(f"rank {rank}: interval {plan[k]['interval']['name']} not finished after "
                              f"{args.poll_timeout:.0f} s", file=sys.stderr, flush=True)
                        os._exit(1)
                    time.sleep(0.0005)
                with torch.cuda.stream(reduce_stream):
                    if last:
                        own_sums(k)
                    if args.dist_backend == "nccl":
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record(reduce_stream)
                        reduce_to_rank0(tensors[k][0])
                        reduce_to_rank0(tensors[k][1])
                        ev[1].record(reduce_stream)
                        reduce_events.append(ev)
                    else:
                        t_r = time.perf_counter()  # (gloo rehearsal: host copies + host reduce)
                        reduce_to_rank0(tensors[k][0])
                        reduce_to_rank0(tensors[k][1])
                        reduce_events.append(1e3 * (time.perf_counter() - t_r))
            sim.wait()
            stream.wait_stream(reduce_stream)
            if args.dist_backend == "nccl":
                reduce_stream.synchronize()
                reduce_ms.append(sum(a.elapsed_time(b) for a, b in reduce_events))
            else:
                reduce_ms.append(sum(reduce_events))
            reduce_events.clear()
        else:
            sim.wait()
            # (single GPU: nothing folds into the outputs after the launch, so the sums of the self-check
            # are taken behind the timed region -- verify_after_timing -- instead of inside the last step)
        kernel_ms.append(sim.kernel_ms())
        if timing:
            t_d = time.perf_counter()
            print(f"[bench timing] submit + zero {1e3 * (t_b - t_a):.1f} ms, launch {1e3 * (t_c - t_b):.1f} ms, "
                  f"wait + collect {1e3 * (t_d - t_c):.1f} ms (kernel {kernel_ms[-1]:.1f} ms)"
                  + (f" {alt_var}={os.environ[alt_var]}" if alternate else ""), file=sys.stderr)

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize(dev)

    first = True
    for _ in range(args.warmup):
        step(first)
        first = False
    kernel_ms.clear()
    reduce_ms.clear()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(first, last=(k == args.steps - 1))
        first = False
    sync()
    dt = time.perf_counter() - t0
    if not use_dist:
        for k, t in enumerate(tensors):  # verify_after_timing: the last step's outputs, untouched since
            if t is not None:
                check["matrix"][k] = t[0].sum(dtype=torch.int64)
                check["occ"][k] = t[1].sum()
    if use_dist:
        import torch.distributed as dist

        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # algorithmic bytes of the LAST step of this rank (results accumulate per step)
    n_tasks = 0
    step_bytes = 0
    epochs = 0
    lef_epochs = 0  # sum over cells and simulated epochs of the active LEFs
    longest = 0  # epochs of the longest cell: a launch with fewer tasks than wave slots lasts as long as it does
    for entry, iid in zip(plan, ids):
        if iid is None:
            continue
        res = sim.results(iid)
        k = len(entry["tasks"])
        last = res[len(res) - k:]
        step_bytes += driver.algorithmic_bytes(last, len(entry["interval"]["bar_pos"]),
                                               bool(cfg.track_1d_lef_position))
        epochs += sum(r.epochs for r in last)
        lef_epochs += sum(r.sum_active_lefs for r in last)
        longest = max(longest, max((r.epochs for r in last), default=0))
        n_tasks += k
    import math

    # (an event that could not be recorded leaves NaN: the line then says null instead of NaN,
    # which is not JSON)
    timed = [ms for ms in kernel_ms if math.isfinite(ms) and ms > 0]
    avg_kernel_s = (sum(timed) / len(timed)) / 1e3 if timed else None
    achieved = step_bytes / avg_kernel_s / 1e9 if avg_kernel_s else None

    # verify the outputs of the last step (every rank checks its own shard; a violation raises
    # and fails the job: the number printed below comes from a launch whose results were looked at)
    missed_after = driver.read_missed(sim, ids)
    missed_delta = [None if a is None else a - b for a, b in zip(missed_after, check["missed_before"])]
    msum = [None if x is None else int(x.item()) for x in check["matrix"]]
    osum = [None if x is None else int(x.item()) for x in check["occ"]]
    if os.environ.get("MODLE_BENCH_NO_VERIFY", "") not in ("", "0"):
        # (measurement builds that leave the output increments out: profiles/r05*/write_accounting.txt)
        verified = {"skipped": "MODLE_BENCH_NO_VERIFY"}
    else:
        verified = driver.verify_outputs(sim, cfg, plan, ids, msum, missed_delta, osum)
    if use_dist:
        import torch.distributed as dist

        tot = torch.tensor([epochs, n_tasks], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tot)
        job_epochs, job_tasks = int(tot[0].item()), int(tot[1].item())
    else:
        job_epochs, job_tasks = epochs, n_tasks
    # the kernel time of every rank (the longest shard bounds the job: a real multi-GPU run shows its
    # imbalance here)
    kernel_ms_per_rank = None
    if use_dist:
        import torch.distributed as dist

        mine = torch.tensor([avg_kernel_s * 1e3 if avg_kernel_s else float("nan")], dtype=torch.float64,
                            device=coll_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        kernel_ms_per_rank = [float(g.item()) for g in gathered]
        kernel_ms_per_rank = [x if math.isfinite(x) else None for x in kernel_ms_per_rank]
        mine_r = torch.tensor([sum(reduce_ms) / len(reduce_ms) if reduce_ms else float("nan")],
                              dtype=torch.float64, device=coll_dev)
        gathered_r = [torch.zeros_like(mine_r) for _ in range(world)]
        dist.all_gather(gathered_r, mine_r)
        reduce_ms_per_rank = [float(g.item()) if math.isfinite(float(g.item())) else None for g in gathered_r]
    # cells the job simulated per step: all of them, except where one GPU stands for one rank of a
    # larger job (grch38-dense on a single GPU: its shard only)
    job_cells = total_cells if plan_world == world else cells_per_gpu

    if rank == 0 and args.checksum_out:
        sums = {}
        for entry, t in zip(plan, tensors):
            if t is None:
                continue
            iv = entry["interval"]
            words = []
            for x in t:
                x64 = x.to(torch.int64)
                weights = torch.arange(1, x64.numel() + 1, dtype=torch.int64, device=dev) % 1000003
                words += [int(x64.sum().item()), int((x64 * weights).sum().item())]
            sums[f"{iv['name']}:{iv['start']}-{iv['end']}"] = words
        with open(args.checksum_out, "w") as f:
            json.dump(sums, f)

    if rank == 0:
        traffic_bytes, traffic_source = measured_traffic(f"{args.workload}:{cells_per_gpu}")
        if args.workload == "grch38-dense" and args.cpu_sample_cells is None:
            args.cpu_sample_cells = max(2, usable_cores())  # (a dense cell costs several default ones)
        out = {
            "metric": "simulated cells/sec (whole node), GRCh38 default barriers"
                      + (" [PHILOX generator policy: statistically equivalent output, not the "
                         "reference's stream]" if args.rng == "philox" else ""),
            "rng": args.rng,
            "value": job_cells * args.steps / dt,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "checked": "skipped" not in verified,
            "check": dict(verified, what="rank 0's shard of the last step: matrix + missed == contacts "
                                         "per interval, per-cell targets, occupancy parity, status 0"),
            "cell_epochs_per_s": job_epochs * args.steps / dt,
            "tasks_per_s": job_tasks * args.steps / dt,
            "vs_baseline": None,
            "dtype": "u32/f64",
            "data": "synthetic",
            "config": {"workload": workload, "cells_per_gpu": cells_per_gpu,
                       "total_cells": total_cells, "tasks_per_gpu": n_tasks,
                       "cell_epochs_per_gpu_step": epochs, "lef_epochs_per_gpu_step": lef_epochs,
                       "longest_cell_epochs": longest, "mean_cell_epochs": epochs / max(n_tasks, 1), "seed": 0,
                       "waves_per_cell": launch_mode(sim.launch_info()),
                       "size_class": size_class(sim.launch_info()),
                       "waves_per_workgroup": sim.launch_info().get("waves_per_workgroup"),
                       "workspace_placement": workspace_placement(sim.launch_info()),
                       "parallelism": f"cells sharded over {world} GPU(s); per-interval "
                                      + ("RCCL" if args.dist_backend == "nccl" else "gloo (host copies)")
                                      + " sum-reduce issued on a side stream as intervals complete"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved is not None else None,
                         "traffic": traffic_bytes, "traffic_source": traffic_source,
                         # (what the committed counters say this kernel really moves per second of THIS run's launches:
                         # the launch is limited by that rate -- 5.7 TB/s is what its 3 072 waves reach when they only
                         # stream through their slots, profiles/r05zs/probe_counters.txt -- while `frac` prices the
                         # algorithmic bytes)
                         "traffic_GBps": (traffic_bytes / avg_kernel_s / 1e9
                                          if traffic_bytes is not None and avg_kernel_s else None),
                         "kernel": ("modle_simulate_cells_wide" if sim.launch_info().get("size_class")
                                    else "modle_simulate_cells_narrow")
                                   + ("12" if sim.launch_info().get("waves_per_workgroup") == 12 else ""),
                         "kernel_ms": avg_kernel_s * 1e3 if avg_kernel_s else None,
                         "kernel_ms_per_rank": kernel_ms_per_rank,
                         "algorithmic_bytes_per_launch": step_bytes},
        }
        if use_dist:
            prediction = None
            try:
                with open(os.path.join(ROOT, "profiles", "r05zs", "scale_prediction.json")) as f:
                    prediction = json.load(f)
            except (OSError, ValueError):
                pass
            out["multi_gpu"] = driver.scaling_report(
                world, args.scaling, total_cells, cells_per_gpu,
                sum(reduce_ms) / len(reduce_ms) if reduce_ms else None, prediction)
            out["multi_gpu"]["reduce_ms_per_rank"] = reduce_ms_per_rank
            out["multi_gpu"]["ranks"] = [{"host": h, "device": i, "uuid": u} for h, u, i in everyone]
            out["multi_gpu"]["backend"] = "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, genome, unit, args.cpu_sample_cells)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    sim.close()
    if use_dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
