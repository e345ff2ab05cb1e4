// launch_common.hpp -- host-side glue shared by the HIP launcher (modle_hip.hip) and the CPU
// lane-emulator harness (tests/wave_emu): digests modle_hip_config into the device-side Params
// and converts between the ABI's 64-bit arrays and the device's 32-bit fields.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "modle_hip.h"
#include "sim_types.h"

namespace modle_host {

inline modle_dev::Params make_params(const modle_hip_config& c) {
  modle_dev::Params p;
  std::memset(&p, 0, sizeof(p));
  p.rev_speed = static_cast<double>(c.rev_extrusion_speed);
  p.fwd_speed = static_cast<double>(c.fwd_extrusion_speed);
  p.rev_speed_burnin = static_cast<double>(c.rev_extrusion_speed_burnin);
  p.fwd_speed_burnin = static_cast<double>(c.fwd_extrusion_speed_burnin);
  p.rev_std = c.rev_extrusion_speed_std;
  p.fwd_std = c.fwd_extrusion_speed_std;
  p.p_release = c.prob_of_lef_release;
  p.p_release_burnin = c.prob_of_lef_release_burnin;
  p.hard_stall_mult = c.hard_stall_lef_stability_multiplier;
  p.soft_stall_mult = c.soft_stall_lef_stability_multiplier;
  p.p_bypass = c.probability_of_extrusion_unit_bypass;
  p.pblock_major = c.lef_bar_major_collision_pblock;
  p.pblock_minor = c.lef_bar_minor_collision_pblock;
  p.tad_to_loop_ratio = c.tad_to_loop_contact_ratio;
  p.gev_mu = c.genextreme_mu;
  p.gev_sigma = c.genextreme_sigma;
  p.gev_xi = c.genextreme_xi;
  p.target_contact_density = c.target_contact_density;
  p.min_burnin_epochs = c.min_burnin_epochs;
  p.max_burnin_epochs = c.max_burnin_epochs;
  p.burnin_target_epochs_for_lef_activation = c.burnin_target_epochs_for_lef_activation;
  p.bin_size = static_cast<uint32_t>(c.bin_size);
  p.sampling_strategy = static_cast<uint32_t>(c.contact_sampling_strategy);
  p.skip_burnin = c.skip_burnin ? 1u : 0u;
  p.hist_len = static_cast<uint32_t>(c.burnin_history_length);
  p.window = static_cast<uint32_t>(c.burnin_smoothing_window_size);
  p.track_1d = c.track_1d_lef_position ? 1u : 0u;
  return p;
}

// checks that an interval / config can be represented by the device layout
inline const char* check_limits(const modle_hip_config& c, uint64_t start, uint64_t end,
                                uint64_t max_lefs, size_t n_barriers) {
  if (end <= start) return "empty interval";
  if (end >= 0xFFFFFFF0ull) return "interval end must be below 2^32 - 16 bp";
  if (c.bin_size == 0 || c.bin_size > 0xFFFFFFFFull) return "invalid bin_size";
  if (max_lefs == 0 || max_lefs >= (1u << 24)) return "number of LEFs must be in [1, 2^24)";
  if (n_barriers >= (1u << 24)) return "number of barriers must be below 2^24";
  if (c.burnin_history_length < c.burnin_smoothing_window_size + 2)
    return "burnin_history_length must exceed burnin_smoothing_window_size + 1";
  if (c.burnin_target_epochs_for_lef_activation == 0 && !c.skip_burnin)
    return "burnin_target_epochs_for_lef_activation must be positive";
  return nullptr;
}

// Size class a set-up NEEDS (sim_types.h; modle_hip_size_class adds the environment's override).
// NARROW (0): LEF ids and moves fit 16 bits.  Ids: fewer than 65 536 LEFs.  Moves: a draw is speed + std * z;
// the move adjustment (reference: simulation.cpp:350-407) raises a move by at most one per link of a chain of
// consecutive units, i.e. by less than the number of LEFs; clamping and the collision passes only shorten
// moves.  z is bounded at 40 here -- the stream cannot produce it (the ziggurat's tail would need a uniform
// below 2^-1000) -- and the kernel ends a cell with ERR_MOVE_RANGE should a move ever exceed the limit, instead
// of truncating it.
inline int size_class_required(const modle_hip_config& c, uint64_t max_lefs) {
  if (max_lefs >= 65536) return 1;
  const double speed = static_cast<double>(std::max(std::max(c.rev_extrusion_speed, c.fwd_extrusion_speed),
                                                    std::max(c.rev_extrusion_speed_burnin, c.fwd_extrusion_speed_burnin)));
  const double sd = std::max(std::fabs(c.rev_extrusion_speed_std), std::fabs(c.fwd_extrusion_speed_std));
  const double bound = speed + 40.0 * sd + static_cast<double>(max_lefs) + 2.0;
  return bound <= 65533.0 ? 0 : 1;  // (MOVE_LIMIT of the NARROW class)
}

inline uint32_t pos_to_dev(uint64_t v) {
  return v == UINT64_MAX ? modle_dev::UNBOUND : static_cast<uint32_t>(v);
}
inline uint64_t pos_to_abi(uint32_t v) {
  return v == modle_dev::UNBOUND ? UINT64_MAX : static_cast<uint64_t>(v);
}
inline uint32_t coll_to_dev(uint64_t w) {
  return static_cast<uint32_t>(w & modle_dev::CW_INDEX_MASK) |
         (static_cast<uint32_t>(w >> 56) << modle_dev::CW_SHIFT);
}
inline uint64_t coll_to_abi(uint32_t w) {
  return static_cast<uint64_t>(w & modle_dev::CW_INDEX_MASK) |
         (static_cast<uint64_t>((w >> modle_dev::CW_SHIFT) & modle_dev::CW_EVENT_MASK) << 56);
}

inline uint32_t pow2_ceil(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// words of scratch one wave needs for `max_lefs` LEFs / `max_barriers` barriers
struct WorkspaceLayout {
  size_t u32_words;   // NUM_STATE_ARRAYS arrays of max_lefs
  size_t u64_words;   // sort keys
  size_t f64_words;   // burn-in history
  size_t u8_bytes;    // barrier states
  size_t hit_words;   // 4 lists of stalling barriers (u32)
  size_t total_bytes;
};

inline WorkspaceLayout workspace_layout(uint32_t max_lefs, uint32_t max_barriers,
                                        uint32_t hist_len) {
  WorkspaceLayout w;
  const size_t Lp = (static_cast<size_t>(max_lefs) + 63) & ~size_t(63);
  w.u32_words = modle_dev::NUM_STATE_ARRAYS * Lp;
  w.u64_words = pow2_ceil(max_lefs < 64 ? 64 : max_lefs);
  w.f64_words = 2 * static_cast<size_t>(hist_len);
  w.u8_bytes = (static_cast<size_t>(max_barriers) + 63) & ~size_t(63);
  w.hit_words = 4 * ((static_cast<size_t>(max_barriers) + 63) & ~size_t(63));
  w.total_bytes = w.u64_words * 8 + w.f64_words * 8 + w.u32_words * 4 + w.u8_bytes + w.hit_words * 4;
  w.total_bytes = (w.total_bytes + 255) & ~size_t(255);
  return w;
}

// carves a Workspace out of `base` (must be 8-byte aligned)
inline modle_dev::Workspace carve_workspace(void* base, uint32_t max_lefs, uint32_t max_barriers,
                                            uint32_t hist_len) {
  const WorkspaceLayout w = workspace_layout(max_lefs, max_barriers, hist_len);
  const size_t Lp = (static_cast<size_t>(max_lefs) + 63) & ~size_t(63);
  modle_dev::Workspace ws;
  char* p = static_cast<char*>(base);
  ws.sort_keys = reinterpret_cast<uint64_t*>(p);
  p += w.u64_words * 8;
  ws.hist = reinterpret_cast<double*>(p);
  p += w.f64_words * 8;
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  ws.r_pos = q + 0 * Lp;
  ws.r_id = reinterpret_cast<modle_dev::lefid_t*>(q + 1 * Lp);
  ws.r_move = reinterpret_cast<modle_dev::move_t*>(q + 2 * Lp);
  ws.r_coll = q + 3 * Lp;
  ws.f_pos = q + 4 * Lp;
  ws.f_id = reinterpret_cast<modle_dev::lefid_t*>(q + 5 * Lp);
  ws.f_move = reinterpret_cast<modle_dev::move_t*>(q + 6 * Lp);
  ws.f_coll = q + 7 * Lp;
  ws.epoch = q + 8 * Lp;
  ws.r_rank = q + 9 * Lp;
  ws.f_rank = q + 10 * Lp;
  ws.stall = q + 11 * Lp;
  for (uint32_t k = 0; k < modle_dev::NUM_TMP; ++k) ws.tmp[k] = q + (12 + static_cast<size_t>(k)) * Lp;
  for (uint32_t d = 0; d < 2; ++d) ws.by_id_pos[d] = q + (12 + modle_dev::NUM_TMP + static_cast<size_t>(d)) * Lp;
  p += w.u32_words * 4;
  ws.bar_active = reinterpret_cast<uint8_t*>(p);
  p += w.u8_bytes;
  uint32_t* hq = reinterpret_cast<uint32_t*>(p);
  const size_t Bp = w.hit_words / 4;
  ws.hit_pos[0] = hq;
  ws.hit_pos[1] = hq + Bp;
  ws.hit_idx[0] = hq + 2 * Bp;
  ws.hit_idx[1] = hq + 3 * Bp;
  ws.capacity_lefs = max_lefs;
  ws.capacity_barriers = max_barriers;
  return ws;
}

// Phase-level test entry point: the caller's arrays (reference layout, 64-bit) packed into nine
// consecutive u32 arrays of n entries (modle_dev::TestImage)
template <class Image>
inline void fill_test_image(uint32_t* base, size_t n, const uint64_t* rev_pos,
                            const uint64_t* fwd_pos, const uint64_t* epoch,
                            const uint64_t* rev_rank, const uint64_t* fwd_rank,
                            const uint64_t* rev_moves, const uint64_t* fwd_moves,
                            const uint64_t* rev_coll, const uint64_t* fwd_coll, Image& img) {
  img.rev_pos = base + 0 * n;
  img.fwd_pos = base + 1 * n;
  img.epoch = base + 2 * n;
  img.rev_rank = base + 3 * n;
  img.fwd_rank = base + 4 * n;
  img.rev_moves = base + 5 * n;
  img.fwd_moves = base + 6 * n;
  img.rev_coll = base + 7 * n;
  img.fwd_coll = base + 8 * n;
  for (size_t i = 0; i < n; ++i) {
    img.rev_pos[i] = pos_to_dev(rev_pos[i]);
    img.fwd_pos[i] = pos_to_dev(fwd_pos[i]);
    img.epoch[i] = pos_to_dev(epoch[i]);
    img.rev_rank[i] = static_cast<uint32_t>(rev_rank[i]);
    img.fwd_rank[i] = static_cast<uint32_t>(fwd_rank[i]);
    img.rev_moves[i] = static_cast<uint32_t>(rev_moves[i]);
    img.fwd_moves[i] = static_cast<uint32_t>(fwd_moves[i]);
    img.rev_coll[i] = coll_to_dev(rev_coll[i]);
    img.fwd_coll[i] = coll_to_dev(fwd_coll[i]);
  }
}

template <class Image>
inline void read_test_image(const Image& img, size_t n, uint64_t* rev_pos, uint64_t* fwd_pos,
                            uint64_t* epoch, uint64_t* rev_rank, uint64_t* fwd_rank,
                            uint64_t* rev_moves, uint64_t* fwd_moves, uint64_t* rev_coll,
                            uint64_t* fwd_coll) {
  for (size_t i = 0; i < n; ++i) {
    rev_pos[i] = pos_to_abi(img.rev_pos[i]);
    fwd_pos[i] = pos_to_abi(img.fwd_pos[i]);
    epoch[i] = pos_to_abi(img.epoch[i]);
    rev_rank[i] = img.rev_rank[i];
    fwd_rank[i] = img.fwd_rank[i];
    rev_moves[i] = img.rev_moves[i];
    fwd_moves[i] = img.fwd_moves[i];
    rev_coll[i] = coll_to_abi(img.rev_coll[i]);
    fwd_coll[i] = coll_to_abi(img.fwd_coll[i]);
  }
}

// bucket table for barrier lookups (see modle_dev::Interval::bar_bucket)
inline std::vector<uint32_t> build_barrier_buckets(uint64_t start, uint64_t end,
                                                   const std::vector<uint32_t>& bar_pos) {
  const uint32_t shift = modle_dev::BAR_BUCKET_SHIFT;
  const size_t n_buckets = static_cast<size_t>((end - start) >> shift) + 2;
  std::vector<uint32_t> table(n_buckets);
  size_t i = 0;
  for (size_t b = 0; b < n_buckets; ++b) {
    const uint64_t lo = start + (static_cast<uint64_t>(b) << shift);
    while (i < bar_pos.size() && bar_pos[i] < lo) ++i;
    table[b] = static_cast<uint32_t>(i);
  }
  return table;
}

}  // namespace modle_host
