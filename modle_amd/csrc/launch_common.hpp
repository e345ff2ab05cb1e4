// launch_common.hpp -- host-side glue shared by the HIP launcher (modle_hip.hip) and the CPU
// lane-emulator harness (tests/wave_emu): digests modle_hip_config into the device-side Params
// and converts between the ABI's 64-bit arrays and the device's 32-bit fields.
#pragma once
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
         (static_cast<uint64_t>(w >> modle_dev::CW_SHIFT) << 56);
}

inline uint32_t pow2_ceil(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// words of scratch one wave needs for `max_lefs` LEFs / `max_barriers` barriers
struct WorkspaceLayout {
  size_t u32_words;   // 13 arrays of max_lefs
  size_t u64_words;   // sort keys
  size_t f64_words;   // burn-in history
  size_t u8_bytes;    // barrier states
  size_t total_bytes;
};

inline WorkspaceLayout workspace_layout(uint32_t max_lefs, uint32_t max_barriers,
                                        uint32_t hist_len) {
  WorkspaceLayout w;
  const size_t Lp = (static_cast<size_t>(max_lefs) + 63) & ~size_t(63);
  w.u32_words = 13 * Lp;
  w.u64_words = pow2_ceil(max_lefs < 64 ? 64 : max_lefs);
  w.f64_words = 2 * static_cast<size_t>(hist_len);
  w.u8_bytes = (static_cast<size_t>(max_barriers) + 63) & ~size_t(63);
  w.total_bytes = w.u64_words * 8 + w.f64_words * 8 + w.u32_words * 4 + w.u8_bytes;
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
  ws.rev_pos = q + 0 * Lp;
  ws.fwd_pos = q + 1 * Lp;
  ws.epoch = q + 2 * Lp;
  ws.rev_rank = q + 3 * Lp;
  ws.fwd_rank = q + 4 * Lp;
  ws.rev_moves = q + 5 * Lp;
  ws.fwd_moves = q + 6 * Lp;
  ws.rev_coll = q + 7 * Lp;
  ws.fwd_coll = q + 8 * Lp;
  ws.tmp_a = q + 9 * Lp;
  ws.tmp_b = q + 10 * Lp;
  ws.tmp_c = q + 11 * Lp;
  ws.tmp_d = q + 12 * Lp;
  p += w.u32_words * 4;
  ws.bar_active = reinterpret_cast<uint8_t*>(p);
  ws.capacity_lefs = max_lefs;
  ws.capacity_barriers = max_barriers;
  return ws;
}

}  // namespace modle_host
