// sim_hooks.h -- part of sim_device.h (included by it, in this order): phase-level and unit-level test entry points (the reference's Simulation::test_* hooks).
#pragma once

namespace modle_dev {

// =============================================================================================
// Phase-level entry point (mirrors Simulation::test_* hooks, reference: simulation.hpp:413-567)
//
// The caller's arrays use the reference's layout: positions / binding epochs / moves / collision
// words indexed by LEF id plus two rank arrays (LEF id at every rank).  They arrive in scratch
// arrays (`TestImage`), are converted to the rank-ordered device layout, the requested passes
// run, and the result is converted back.
// =============================================================================================
constexpr u32 PH_RANK = 0x001, PH_RANK_INIT = 0x002, PH_ADJUST = 0x004, PH_CLAMP = 0x008,
              PH_BOUNDARIES = 0x010, PH_LEF_BAR = 0x020, PH_PRIMARY = 0x040,
              PH_CORRECT_LEF_BAR = 0x080, PH_CORRECT_PRIMARY = 0x100, PH_SECONDARY = 0x200,
              PH_FIX_SECONDARY = 0x400, PH_USE_BOUNDARY_COUNTS = 0x800,
              PH_BIND = 0x1000,       // select_and_bind_lefs: bind every released LEF, then rank
              PH_GEN_MOVES = 0x2000;  // generate_moves: draw, adjust, clamp
// bits 16..31 of the mask: the current epoch (binding epoch of the LEFs PH_BIND binds)

struct TestImage {  // all by LEF id except the two rank arrays; n entries each
  u32 *rev_pos, *fwd_pos, *epoch, *rev_rank, *fwd_rank, *rev_moves, *fwd_moves, *rev_coll,
      *fwd_coll;
};

// rank positions whose unit carries an "avoided secondary collision" mark, in the order
// process_secondary would have produced them
template <bool FWD>
MODLE_DEV u32 collect_avoided(Cell& c, u32* list, u32 cap) {
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32* coll = FWD ? c.ws.f_coll : c.ws.r_coll;
  u32 cnt = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 off = base + lane;
    const bool act = off < n;
    const u32 k = FWD ? (n - 1 - off) : off;
    const bool hit = act && cw_avoided_as(coll[act ? k : 0], EV_LEF_LEF_SECONDARY) &&
                     (FWD ? k + 1 < n : k >= 1);
    const u64 m = wave::ballot(hit);
    if (hit) {
      const u32 j = cnt + static_cast<u32>(wave::popc64(m & lanemask_lt(lane)));
      if (j < cap) list[j] = k;
    }
    cnt += static_cast<u32>(wave::popc64(m));
  }
  wave::sync_mem();
  return cnt < cap ? cnt : cap;
}

MODLE_DEV u32 run_test_phases(const Params& p, const Interval& iv, const Workspace& ws,
                              const WaveLds& lds, const TestImage& img, u32 mask, u32 n,
                              const u64 prng[4], u64& raws_consumed) {
  Cell c;
  const Interval ivg = interval_in_device_memory(iv);
  init_cell(c, p, ivg, ws, lds, n, prng);
  c.n_active = n;
  const u32 lane = wave::lane();
  // reference layout -> device layout
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const bool init = (mask & PH_RANK) && (mask & PH_RANK_INIT);
      const u32 rid = init ? k : img.rev_rank[k];
      const u32 fid = init ? k : img.fwd_rank[k];
      c.ws.r_id[k] = rid;
      c.ws.r_pos[k] = img.rev_pos[rid];
      c.ws.r_move[k] = img.rev_moves[rid];
      c.ws.r_coll[k] = img.rev_coll[rid];
      c.ws.r_rank[rid] = k;
      c.ws.f_id[k] = fid;
      c.ws.f_pos[k] = img.fwd_pos[fid];
      c.ws.f_move[k] = img.fwd_moves[fid];
      c.ws.f_coll[k] = img.fwd_coll[fid];
      c.ws.f_rank[fid] = k;
      c.ws.epoch[k] = img.epoch[k];
      c.ws.stall[k] = 0;
    }
  }
  wave::sync_mem();
  // barrier positions of LEF-BAR words that came with the image (detect_lef_bar writes them
  // itself when it runs)
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 rc = c.ws.r_coll[k], fc = c.ws.f_coll[k];
      if (cw_occurred_as(rc, EV_LEF_BAR)) stalling_barrier_positions<false>(c.ws)[k] = stalling_barrier_pos(ivg, rc);
      if (cw_occurred_as(fc, EV_LEF_BAR)) stalling_barrier_positions<true>(c.ws)[k] = stalling_barrier_pos(ivg, fc);
    }
  }
  wave::sync_mem();
  if (mask & PH_BIND) {
    // Simulation::select_and_bind_lefs (simulation.cpp:988-993): the released LEFs of the image
    // are the ones to bind; the ranking that follows is the partially sorted one
    phase_bind(c, mask >> 16);
    rank_update<false>(c, false);
    rank_update<true>(c, false);
  }
  if (mask & PH_RANK) {
    // positions only: move / collision arrays are not meaningful across a re-ranking
    rank_update<false>(c, true);
    rank_update<true>(c, true);
  }
  if (mask & PH_GEN_MOVES) {
    bool unbound = false;
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      unbound = wave::any(k < n && c.ws.epoch[k] == UNBOUND) || unbound;
    }
    phase_generate_moves(c, true, !unbound);
  }
  if (mask & (PH_ADJUST | PH_CLAMP)) {
    adjust_moves_rev(c, (mask & PH_ADJUST) != 0, (mask & PH_CLAMP) != 0);
    adjust_moves_fwd(c, (mask & PH_ADJUST) != 0, (mask & PH_CLAMP) != 0);
  }
  BoundaryCounts bc{0, 0};
  if (mask & PH_BOUNDARIES) {
    const BoundaryCounts got = detect_boundaries(c);
    if (mask & PH_USE_BOUNDARY_COUNTS) bc = got;
  }
  if (mask & PH_LEF_BAR) {
    compact_stalling_barriers(c);
    detect_lef_bar<false>(c, bc);
    detect_lef_bar<true>(c, bc);
  }
  // the reference's hook sequences run "correct LEF-BAR moves" before "correct primary moves";
  // with both requested the fused forms are equivalent, otherwise run them stand-alone
  const bool fuse = (mask & PH_CORRECT_PRIMARY) && (mask & PH_CORRECT_LEF_BAR) && (mask & PH_PRIMARY);
  if (mask & PH_PRIMARY) detect_primary(c, bc, fuse);
  u32* list_rev = c.ws.tmp[5];
  u32* list_fwd = c.ws.tmp[6];
  const u32 cap = c.ws.capacity_lefs;
  u32 nr = 0, nf = 0;
  bool overflow = false;
  if (!fuse && (mask & PH_CORRECT_LEF_BAR)) {
    (void)process_secondary<false>(c, bc, list_rev, cap, overflow, true, false);
    (void)process_secondary<true>(c, bc, list_fwd, cap, overflow, true, false);
  }
  if (!fuse && (mask & PH_CORRECT_PRIMARY)) correct_moves_primary_standalone(c);
  if (mask & PH_SECONDARY) {
    nr = process_secondary<false>(c, bc, list_rev, cap, overflow, fuse, true);
    nf = process_secondary<true>(c, bc, list_fwd, cap, overflow, fuse, true);
  } else if (fuse) {
    (void)process_secondary<false>(c, bc, list_rev, cap, overflow, true, false);
    (void)process_secondary<true>(c, bc, list_fwd, cap, overflow, true, false);
  }
  if (mask & PH_FIX_SECONDARY) {
    if (!(mask & PH_SECONDARY)) {
      nr = collect_avoided<false>(c, list_rev, cap);
      nf = collect_avoided<true>(c, list_fwd, cap);
    }
    if (nr != 0) fix_secondary_rev(c, list_rev, nr);
    if (nf != 0) fix_secondary_fwd(c, list_fwd, nf);
  }
  // device layout -> reference layout
  wave::sync_mem();
  for (u32 base = 0; base < n; base += 64) {
    const u32 k = base + lane;
    if (k < n) {
      const u32 rid = c.ws.r_id[k], fid = c.ws.f_id[k];
      img.rev_rank[k] = rid;
      img.fwd_rank[k] = fid;
      img.rev_pos[rid] = c.ws.r_pos[k];
      img.rev_moves[rid] = c.ws.r_move[k];
      img.rev_coll[rid] = c.ws.r_coll[k];
      img.fwd_pos[fid] = c.ws.f_pos[k];
      img.fwd_moves[fid] = c.ws.f_move[k];
      img.fwd_coll[fid] = c.ws.f_coll[k];
      img.epoch[k] = c.ws.epoch[k];  // binding epochs change under PH_BIND
    }
  }
  wave::sync_mem();
  raws_consumed = c.g.pos;
  return overflow ? ERR_LIST_OVERFLOW : c.error;
}

// =============================================================================================
// Unit-level entry point: the small pieces of the path the reference tests on their own
// (test/units/stats/descriptive_test.cpp, test/units/contact_matrix/*_test.cpp,
// test/units/simulation_cpu/collision_encoding_test.cpp), run by the device code itself.
// =============================================================================================
constexpr u32 UNIT_LOOP_STATS = 1, UNIT_MATRIX_INCREMENT = 2, UNIT_COLLISION_WORDS = 3,
              UNIT_MATH_LOG_EXP = 4, UNIT_MATH_POW_SQRT = 5, UNIT_PHILOX = 6;

// predicates of one collision word, packed: bit 0 collision_occurred(), bit 1 collision_avoided(),
// bits 2..5 collision_occurred(CHROM_BOUNDARY / LEF_BAR / LEF_LEF_PRIMARY / LEF_LEF_SECONDARY),
// bits 6..9 collision_avoided(the same four)
MODLE_DEV u32 cw_predicates(u32 w) {
  const u32 kinds[4] = {EV_CHROM_BOUNDARY, EV_LEF_BAR, EV_LEF_LEF_PRIMARY, EV_LEF_LEF_SECONDARY};
  u32 f = (cw_occurred(w) ? 1u : 0u) | ((!cw_occurred(w) && w != 0) ? 2u : 0u);
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    f |= cw_occurred_as(w, kinds[k]) ? (4u << k) : 0u;
    f |= cw_avoided_as(w, kinds[k]) ? (64u << k) : 0u;
  }
  return f;
}

// `in`: n pairs of 64-bit values; `out`: what the unit produces (see the cases); uniform
MODLE_DEV u32 run_test_units(const Params& p, const Interval& iv, const Workspace& ws,
                             const WaveLds& lds, u32 what, const u64* in, u32 n, u64* out) {
  const u32 lane = wave::lane();
  const Interval ivg = interval_in_device_memory(iv);
  const u64 zero[4] = {1, 2, 3, 4};
  Cell c;
  init_cell(c, p, ivg, ws, lds, n, zero);
  c.n_active = n;
  if (what == UNIT_LOOP_STATS) {
    // pairs (rev position, fwd position) of LEF i.  The units sit at scrambled ranks (rev: reversed;
    // fwd: a stride permutation), as they do in a cell: the statistics must restore the id order
    const u32 stride = n % 7919u != 0 ? 7919u : 1u;  // (7919 is prime: a bijection modulo n)
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        const u32 ir = n - 1 - k;
        const u32 jf = static_cast<u32>((static_cast<u64>(k) * stride + 13u) % n);
        c.ws.r_pos[k] = static_cast<u32>(in[2 * ir]);
        c.ws.r_id[k] = ir;
        c.ws.f_pos[k] = static_cast<u32>(in[2 * jf + 1]);
        c.ws.f_id[k] = jf;
      }
    }
    wave::sync_mem();
    const LoopStats st = loop_size_stats(c);
    if (lane == 0) {
      out[0] = static_cast<u64>(__builtin_bit_cast(i64, st.avg));
      out[1] = static_cast<u64>(__builtin_bit_cast(i64, st.std));
    }
  } else if (what == UNIT_MATRIX_INCREMENT) {
    // pairs (row, col): ContactMatrixDense::increment
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) matrix_increment(ivg, in[2 * k], in[2 * k + 1]);
    }
  } else if (what == UNIT_COLLISION_WORDS) {
    // pairs (index, event): out = (word, predicates)
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        const u32 w = cw_make(static_cast<u32>(in[2 * k]), static_cast<u32>(in[2 * k + 1]));
        out[2 * k] = (static_cast<u64>(cw_event(w)) << 56) | cw_index(w);
        out[2 * k + 1] = cw_predicates(w);
      }
    }
  } else if (what == UNIT_MATH_LOG_EXP || what == UNIT_MATH_POW_SQRT) {
    // pairs (bits of x, bits of y): out = (log x, exp y) or (pow(x, y), sqrt x), as bit images:
    // the floating-point library the path uses (wave::f_*), evaluated per lane
    for (u32 base = 0; base < n; base += 64) {
      const u32 k = base + lane;
      if (k < n) {
        const f64 x = __builtin_bit_cast(f64, in[2 * k]), y = __builtin_bit_cast(f64, in[2 * k + 1]);
        const f64 a = what == UNIT_MATH_LOG_EXP ? wave::f_log(x) : wave::f_pow(x, y);
        const f64 b = what == UNIT_MATH_LOG_EXP ? wave::f_exp(y) : wave::f_sqrt(x);
        out[2 * k] = __builtin_bit_cast(u64, a);
        out[2 * k + 1] = __builtin_bit_cast(u64, b);
      }
    }
  } else if (what == UNIT_PHILOX) {
    // pairs (counter words 0..1 | 2..3 as two 64-bit values) followed by (key words 0..1, unused):
    // two pairs per vector; out = the four output words as two 64-bit values, then zeros
    for (u32 base = 0; base < n / 2; base += 64) {
      const u32 v = base + lane;
      if (v < n / 2) {
        const u64 c_lo = in[4 * v], c_hi = in[4 * v + 1], key = in[4 * v + 2];
        u32 x[4];
        philox4x32_10(static_cast<u32>(c_lo), static_cast<u32>(c_lo >> 32), static_cast<u32>(c_hi),
                      static_cast<u32>(c_hi >> 32), static_cast<u32>(key), static_cast<u32>(key >> 32), x);
        out[4 * v] = (static_cast<u64>(x[1]) << 32) | x[0];
        out[4 * v + 1] = (static_cast<u64>(x[3]) << 32) | x[2];
        out[4 * v + 2] = 0;
        out[4 * v + 3] = 0;
      }
    }
  } else {
    return ERR_INTERNAL;
  }
  wave::sync_mem();
  return 0;
}

}  // namespace modle_dev
