// sim_epoch.h -- part of sim_device.h (included by it, in this order): the epoch loop of one cell (Simulation::simulate_one_cell).
#pragma once

namespace modle_dev {

// =============================================================================================
// Cell driver (reference: simulation.cpp:896-986)
// =============================================================================================
MODLE_DEV_NOINLINE void reset_cell_buffers(Cell& c) {
  // State::reset_buffers (reference: simulation.cpp:617-627)
  Workspace& ws = c.ws;
  const u32 L = wave::uniform(c.n_lefs);
  const u32 lane = wave::lane();
  for (u32 base = 0; base < L; base += 64) {
    const u32 i = base + lane;
    if (i < L) {
      ws.r_pos[i] = UNBOUND;
      ws.f_pos[i] = UNBOUND;
      ws.epoch[i] = UNBOUND;
      ws.r_id[i] = i;
      ws.f_id[i] = i;
      ws.r_rank[i] = i;
      ws.f_rank[i] = i;
      ws.r_move[i] = 0;
      ws.f_move[i] = 0;
      ws.r_coll[i] = 0;
      ws.f_coll[i] = 0;
      ws.stall[i] = 0;
    }
  }
  wave::sync_mem();
}

// LEFs n_old .. n_new-1 become active.  They have never been ranked: their slots are the
// identity (the reference's iota-initialised rank buffers, simulation.cpp:617-620); the id
// arrays are re-initialised here because they are double-buffered by rank_update.
MODLE_DEV void activate_lefs(Cell& c, u32 n_old, u32 n_new) {
  const u32 lane = wave::lane();
  for (u32 base = n_old; base < n_new; base += 64) {
    const u32 k = base + lane;
    if (k < n_new) {
      c.ws.r_id[k] = k;
      c.ws.f_id[k] = k;
      c.ws.r_rank[k] = k;
      c.ws.f_rank[k] = k;
    }
  }
  c.n_active = n_new;
  wave::sync_mem();
}

MODLE_DEV void init_cell(Cell& c, const Params& p, const Interval& iv, const Workspace& ws,
                         const WaveLds& lds, u32 n_lefs, const u64 prng[4]) {
  c.p = &p;
  c.iv = &iv;
  c.ws = ws;
  c.lds = lds;
  c.n_lefs = n_lefs;
  c.n_active = 0;
  c.hist_len = 0;
  c.hist_head = 0;
  c.error = 0;
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  c.n_rel = 0;
  c.rel_valid = false;  // the epoch loop turns the list on; the phase-level hooks sweep
  c.keys_valid = false;
  c.n_keys = 0;
  c.n_disp[0] = 0;
  c.n_disp[1] = 0;
  c.disp_valid = true;  // (nothing has been ranked yet: nothing can be out of order)
  c.max_fwd_move = 0xFFFFFFFFu;
  c.n_bound = 0;
  c.inv_valid[0] = true;  // (reset_cell_buffers / run_test_phases write complete permutations)
  c.inv_valid[1] = true;
  c.filter_on = false;
  c.by_id_valid = false;
  // (the helper keeps counting the requests across the tasks of its main wave)
  c.pair_seq = lds.mbox != nullptr ? wave::uniform(lds.mbox[PAIR_REQ]) : 0u;
  c.pair_on = false;
  c.ring_lent = false;
#ifdef MODLE_PHASE_TIMERS
  for (int i = 0; i < 16; ++i) c.ph[i] = 0;
#endif
  c.g.ring = lds.ring;
  c.g.jump = lds.jump_table;
  c.g.state = lds.rng_state;
  c.g.snap = lds.rng_snap;
  rng_init(c.g, prng);
}

// Diagnostic trace (enabled by the host with MODLE_HIP_TRACE_SHM): after selected phases of every
// epoch, order-sensitive checksums of the unit arrays and the PRNG position are stored, so that
// a run on the GPU can be compared phase by phase with a run under the CPU lane emulator.
constexpr u32 TRACE_STAGES = 8;
constexpr u32 TRACE_WORDS_PER_STAGE = 6;
// Compiled in only with MODLE_STAGE_TRACE (`make trace`, the emulator build): seven inlined copies
// of this function are a seventh of the kernel's code, all of it dead weight in the instruction
// cache of a normal run.
#ifndef MODLE_STAGE_TRACE
MODLE_DEV void trace_stage(Cell&, u64, u32) {}
#else
MODLE_DEV_NOINLINE void trace_stage(Cell& c, u64 epoch, u32 stage) {
  u64* tr = c.lds.trace;
  if (tr == nullptr || epoch >= c.lds.trace_cap) return;
  const u32 lane = wave::lane();
  u64 s[4] = {0, 0, 0, 0};
  for (u32 base = 0; base < c.n_active; base += 64) {
    const u32 k = base + lane;
    if (k < c.n_active) {
      const u64 w = k + 1;
      s[0] += w * c.ws.r_pos[k] + c.ws.r_id[k];
      s[1] += w * c.ws.f_pos[k] + c.ws.f_id[k];
      s[2] += w * c.ws.r_move[k];
      s[3] += w * c.ws.f_move[k];
    }
  }
#pragma unroll
  for (u32 q = 0; q < 4; ++q) {
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
      const u64 o = wave::shfl_down(s[q], d);
      if (lane + d < 64) s[q] += o;
    }
  }
  if (lane == 0) {
    u64* rec = tr + (epoch * TRACE_STAGES + stage) * TRACE_WORDS_PER_STAGE;
    rec[0] = c.g.pos;
    rec[1] = s[0];
    rec[2] = s[1];
    rec[3] = s[2];
    rec[4] = s[3];
    rec[5] = (static_cast<u64>(c.n_active) << 32) | (stage + 1);
  }
}
#endif

// Model-internal-state record of one epoch (Simulation::dump_stats, reference:
// simulation.cpp:995-1056; logged after extrude and before release_lefs, :969-975).  Compiled in
// only with MODLE_STATE_LOG (`make statelog`): the front end loads that build when
// --log-model-internal-state is given.  Runs BEFORE the fused extrusion / release pass (which
// consumes the collision words), on positions + moves = the positions after extrusion.
#ifdef MODLE_STATE_LOG
MODLE_DEV_NOINLINE void log_internal_state(Cell& c, u64 epoch, bool burnin) {
  u64* log = c.lds.state_log;
  if (log == nullptr || epoch >= c.lds.state_log_cap) return;
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  u32* flag = ws.tmp[2];  // per LEF: its rev unit is stalled
  u32 st_rev = 0, st_fwd = 0, st_both = 0, n_bar = 0, n_prim = 0, n_sec = 0;
  u64 part = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 rc = wave::ld_sel(ws.r_coll, k, act, 0);
    const u32 P = wave::ld_sel(ws.r_pos, k, act, UNBOUND);
    if (act) flag[ws.r_id[k]] = cw_occurred(rc) ? 1u : 0u;
    st_rev += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred(rc))));
    n_bar += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_BAR))));
    n_prim += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_LEF_PRIMARY))));
    n_sec += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(rc, EV_LEF_LEF_SECONDARY))));
    if (act && P != UNBOUND) part -= static_cast<u64>(P - ws.r_move[k]);
  }
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    const bool act = k < n;
    const u32 fc = wave::ld_sel(ws.f_coll, k, act, 0);
    const u32 P = wave::ld_sel(ws.f_pos, k, act, UNBOUND);
    const bool both = act && cw_occurred(fc) && flag[ws.f_id[k]] != 0;
    st_fwd += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred(fc))));
    st_both += static_cast<u32>(wave::popc64(wave::ballot(both)));
    n_bar += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_BAR))));
    n_prim += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_LEF_PRIMARY))));
    n_sec += static_cast<u32>(wave::popc64(wave::ballot(act && cw_occurred_as(fc, EV_LEF_LEF_SECONDARY))));
    if (act && P != UNBOUND) part += static_cast<u64>(P + ws.f_move[k]);
  }
#pragma unroll
  for (u32 sft = 1; sft < 64; sft <<= 1) {
    const u64 o = wave::shfl_down(part, sft);
    if (lane + sft < 64) part += o;
  }
  const u64 loop_sum = wave::bcast(part, 0);
  u32 n_occ = 0;
  const u32 nb = wave::uniform(c.iv->n_barriers);
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    n_occ += static_cast<u32>(wave::popc64(wave::ballot(i < nb && ws.bar_active[i] != 0)));
  }
  if (lane == 0) {
    u64* rec = log + epoch * STATE_LOG_WORDS;
    rec[0] = epoch | (burnin ? (u64(1) << 63) : 0);
    rec[1] = n_occ;
    rec[2] = n;
    rec[3] = st_rev;
    rec[4] = st_fwd;
    rec[5] = st_both;
    rec[6] = n_bar;
    rec[7] = n_prim;
    rec[8] = n_sec;
    rec[9] = loop_sum;
  }
  wave::sync_mem();
}
#else
MODLE_DEV void log_internal_state(Cell&, u64, bool) {}
#endif

// Simulates one (interval, cell) task on the calling wave.  Returns 0 or a non-zero status when
// an internal capacity was exceeded (the host turns that into an error).
MODLE_DEV u32 simulate_cell(const Params& p, const Interval& iv, const Task& task,
                            const Workspace& ws, const WaveLds& lds, CellResult& res) {
  Cell c;
#ifdef MODLE_PHASE_TIMERS
  const u64 t_cell = wave::clock();
#endif
  const Interval ivg = interval_in_device_memory(iv);
  init_cell(c, p, ivg, ws, lds, task.num_lefs, task.prng);
  c.pair_interval = task.interval;
  reset_cell_buffers(c);

  u64 epoch = 0, num_burnin_epochs = 0, num_contacts = 0;
  u64 sum_active = 0, events_done = 0, sim_epochs = 0;
  bool burnin_completed = false;
  u32 status = 0;
  const f64 lef_binding_rate_burnin =
      static_cast<f64>(task.num_lefs) / static_cast<f64>(p.burnin_target_epochs_for_lef_activation);

  barriers_init_states(c);
  c.rel_valid = true;  // nothing released yet: every LEF to bind is a newly activated one
  if (p.skip_burnin) {
    activate_lefs(c, 0, c.n_lefs);
    burnin_completed = true;
  }
  for (;; ++epoch) {
    if (p.target_contact_density >= 0) {
      if (num_contacts >= task.num_target_contacts) break;
    } else if (epoch - num_burnin_epochs >= task.num_target_epochs) {
      break;
    }
    // cancellation (the reference polls `_ctx` once per epoch, simulation.cpp:933): the abort word
    // is in host memory (see modle_hip_cancel), one round trip over the fabric: it is read every
    // sixteenth epoch, a few milliseconds apart
    if (lds.abort_flag != nullptr && (epoch & 15u) == 0 &&
        wave::uniform(wave::load_system_u32(lds.abort_flag)) != 0) {
      status = ERR_CANCELLED;
      break;
    }
    // run_burnin (reference: simulation.cpp:866-894).  The loop-size statistics and the stability
    // test of an epoch run BEHIND the bind phase: they draw nothing, a LEF bound in between counts
    // exactly as it did while it was unbound (both units at one position: loop size 0), and behind the
    // bind phase the LDS sort buffer no longer holds the list of released LEFs, so the statistics can
    // restore the LEF-id order there instead of scattering to device memory (sim_burnin.h).
    bool stats_due = false;
    c.by_id_valid = false;  // (last epoch's extrusion moved every unit)
    if (!burnin_completed) {
      do {
        ++num_burnin_epochs;
        if (c.n_active != c.n_lefs) {
          PHASE(c, MODLE_PH_ACTIVATION, const u64 k = poisson_exact(c.g, lef_binding_rate_burnin);
                const u64 na = static_cast<u64>(c.n_active) + k;
                activate_lefs(c, c.n_active, na < c.n_lefs ? static_cast<u32>(na) : c.n_lefs));
        } else {
          stats_due = true;  // (every LEF is active: the loop ends here)
        }
      } while (c.n_active == 0);
    }
    PHASE(c, 1, if (c.rel_valid) phase_bind_listed(c, static_cast<u32>(epoch));
          else {
            phase_bind(c, static_cast<u32>(epoch));
            c.n_bound = c.n_active;
          });
    if (stats_due) {
      PHASE(c, 0, compute_loop_size_stats(c); burnin_completed = evaluate_burnin(c));
      burnin_completed = burnin_completed && epoch > p.min_burnin_epochs;
      if (!burnin_completed && epoch >= p.max_burnin_epochs) {
        burnin_completed = true;
        activate_lefs(c, c.n_active, c.n_lefs);
      }
    }
    trace_stage(c, epoch, 0);
    // helper-wave mode (sim_pair.h), burn-in epochs (nothing draws between the bind phase and the
    // moves): the helper draws the moves and the barrier states while this wave ranks the units
    c.pair_on = pair_helper_present(lds);
    const bool offload = c.pair_on && !burnin_completed;
    if (offload) pair_request(c, burnin_completed, task.interval);
    PHASE(c, 2, rank_update<false>(c, false));
    PHASE(c, 3, rank_update<true>(c, false));
    trace_stage(c, epoch, 1);
    if (c.error != 0) {
      const u32 first_error = c.error;
      if (offload) (void)pair_take_back(c);
      status = first_error;
      break;
    }

    if (burnin_completed) {
      PHASE(c, 4, num_contacts += phase_sample_contacts(c, task.contacts_per_epoch,
                                                        task.num_target_contacts, num_contacts,
                                                        events_done));
      trace_stage(c, epoch, 5);
      if (task.num_target_contacts != 0 && num_contacts >= task.num_target_contacts) break;
    }

    sum_active += c.n_active;
    ++sim_epochs;
    if (offload) {
      // (a wait that fails -- the host raised the abort word, or the helper reported an error -- ends
      // the cell: what the helper was to deliver is not there)
      c.max_fwd_move = 0xFFFFFFFFu;
      bool handed = false;
      PHASE(c, 5, handed = pair_wait(c, PAIR_MOVES));
      if (!handed) {
        status = c.error;
        break;
      }
      phase_adjust_moves_by_id(c);
      trace_stage(c, epoch, 2);
      PHASE(c, 7, handed = pair_take_back(c));
      if (!handed) {
        status = c.error;
        break;
      }
    } else {
      phase_generate_moves(c, burnin_completed);
      trace_stage(c, epoch, 2);
      PHASE(c, 7, barriers_next_state(c));
    }
    if (NARROW_MOVES && c.error != 0) {  // (ERR_MOVE_RANGE: a move the NARROW class cannot hold)
      status = c.error;
      break;
    }
    const bool coll_ok = phase_process_collisions(c);
    trace_stage(c, epoch, 3);
    if (!coll_ok) {
      status = c.error;
      break;
    }
    log_internal_state(c, epoch, !burnin_completed);
    PHASE(c, 13, phase_extrude_and_release(c, burnin_completed));
    trace_stage(c, epoch, 4);
  }

  trace_stage(c, epoch, 6);
#ifdef MODLE_PHASE_TIMERS
#ifndef MODLE_SUBTIMER
  c.ph[15] = wave::clock() - t_cell;  // (the whole cell: what the phases do not add up to is the glue between them)
#endif
  if (lds.phase_ticks != nullptr && wave::lane() == 0) {
    for (int i = 0; i < 16; ++i) wave::atomic_add_u64(lds.phase_ticks + i, c.ph[i]);
  }
  wave::lockstep();
#endif
  res.epochs = epoch;
  res.burnin_epochs = num_burnin_epochs;
  res.num_contacts = num_contacts;
  res.raws_consumed = c.g.pos;
  rng_final_state(c.g, res.prng_final);
  res.sum_active_lefs = sum_active;
  res.sampling_events = events_done;
  res.sim_epochs = sim_epochs;
  return status;
}

}  // namespace modle_dev
