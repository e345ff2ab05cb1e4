// host_logic.cpp -- host-side counterpart of the reference's scheduler / CLI maths for the hot
// path: Config defaults and derived parameters, interval seeding (XXH3-64), xoshiro256++ state
// derivation and the per-cell task list.  Pure C++17, no GPU.
//
// Reference (paths relative to /root/reference):
//   src/common/include/modle/common/simulation_config.hpp:47-113   defaults
//   src/modle/cli.cpp:886-1016                                     Cli::transform_args
//   src/libmodle/internal/genome.cpp:201-224                       GenomicInterval::hash
//   src/common/include/modle/common/random.hpp:26-32               random::PRNG
//   src/libmodle/cpu/scheduler_simulate.cpp:104-160                task generation
//   src/libmodle/cpu/simulation.cpp:1076-1090                      compute_num_lefs & co.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "modle_hip.h"
#include "launch_common.hpp"
#include "host_prng.hpp"

namespace {

void set_err(char* err, size_t errlen, const char* msg) {
  if (err != nullptr && errlen != 0) std::snprintf(err, errlen, "%s", msg);
}

// ---- XXH3-64 with seed for inputs up to 240 bytes (xxHash 0.8.x specification) ---------------
// GenomicInterval::hash streams at most a chromosome name plus three u64, so only the short /
// mid-size code paths of XXH3 are ever exercised by the path.
struct Xxh3 {
  static constexpr uint64_t P32_1 = 0x9E3779B1ULL, P32_2 = 0x85EBCA77ULL, P32_3 = 0xC2B2AE3DULL;
  static constexpr uint64_t P64_1 = 0x9E3779B185EBCA87ULL, P64_2 = 0xC2B2AE3D27D4EB4FULL,
                            P64_3 = 0x165667B19E3779F9ULL;
  static constexpr uint64_t MX1 = 0x165667919E3779F9ULL, MX2 = 0x9FB21C651E98DF25ULL;

  static const uint8_t* secret() {
    static const uint8_t k[192] = {
        0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad,
        0x1c, 0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3,
        0x67, 0x1f, 0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc,
        0xff, 0x72, 0x21, 0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6,
        0x81, 0x3a, 0x26, 0x4c, 0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65,
        0x8b, 0x1b, 0x53, 0x2e, 0xa3, 0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19,
        0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8, 0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9,
        0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d, 0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31,
        0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64, 0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb,
        0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb, 0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0,
        0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e, 0x2b, 0x16, 0xbe, 0x58, 0x7d,
        0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce, 0x45, 0xcb, 0x3a, 0x8f,
        0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e};
    return k;
  }
  template <class T>
  static T load(const uint8_t* p) {
    T v;
    std::memcpy(&v, p, sizeof(T));
    return v;
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  static uint64_t fold(uint64_t a, uint64_t b) {
    const unsigned __int128 m = static_cast<unsigned __int128>(a) * b;
    return static_cast<uint64_t>(m) ^ static_cast<uint64_t>(m >> 64);
  }
  static uint64_t avalanche(uint64_t h) {
    h ^= h >> 37;
    h *= MX1;
    return h ^ (h >> 32);
  }
  static uint64_t avalanche64(uint64_t h) {
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    return h ^ (h >> 32);
  }
  static uint64_t mix16(const uint8_t* in, const uint8_t* sec, uint64_t seed) {
    return fold(load<uint64_t>(in) ^ (load<uint64_t>(sec) + seed),
                load<uint64_t>(in + 8) ^ (load<uint64_t>(sec + 8) - seed));
  }

  static uint64_t hash(const uint8_t* in, size_t len, uint64_t seed) {
    const uint8_t* s = secret();
    if (len == 0) return avalanche64(seed ^ (load<uint64_t>(s + 56) ^ load<uint64_t>(s + 64)));
    if (len < 4) {
      const uint32_t combined = (uint32_t(in[0]) << 16) | (uint32_t(in[len >> 1]) << 24) |
                                uint32_t(in[len - 1]) | (uint32_t(len) << 8);
      return avalanche64(combined ^ ((load<uint32_t>(s) ^ load<uint32_t>(s + 4)) + seed));
    }
    if (len <= 8) {
      seed ^= uint64_t(__builtin_bswap32(uint32_t(seed))) << 32;
      const uint64_t in64 = load<uint32_t>(in + len - 4) + (uint64_t(load<uint32_t>(in)) << 32);
      uint64_t h = in64 ^ ((load<uint64_t>(s + 8) ^ load<uint64_t>(s + 16)) - seed);
      h ^= rotl(h, 49) ^ rotl(h, 24);
      h *= MX2;
      h ^= (h >> 35) + len;
      h *= MX2;
      return h ^ (h >> 28);
    }
    if (len <= 16) {
      const uint64_t lo =
          load<uint64_t>(in) ^ ((load<uint64_t>(s + 24) ^ load<uint64_t>(s + 32)) + seed);
      const uint64_t hi = load<uint64_t>(in + len - 8) ^
                          ((load<uint64_t>(s + 40) ^ load<uint64_t>(s + 48)) - seed);
      return avalanche(len + __builtin_bswap64(lo) + hi + fold(lo, hi));
    }
    uint64_t acc = len * P64_1;
    if (len <= 128) {
      // pairs of 16-byte stripes taken from both ends, innermost first
      const size_t npairs = (len - 1) / 32;  // 0..3
      for (size_t k = npairs + 1; k-- > 0;) {
        acc += mix16(in + 16 * k, s + 32 * k, seed);
        acc += mix16(in + len - 16 * (k + 1), s + 32 * k + 16, seed);
      }
      return avalanche(acc);
    }
    if (len <= 240) {
      for (size_t i = 0; i < 8; ++i) acc += mix16(in + 16 * i, s + 16 * i, seed);
      acc = avalanche(acc);
      for (size_t i = 8; i < len / 16; ++i) acc += mix16(in + 16 * i, s + 16 * (i - 8) + 3, seed);
      acc += mix16(in + len - 16, s + 136 - 17, seed);
      return avalanche(acc);
    }
    return 0;  // not reachable from the path
  }
};

double clamp01(double x) { return std::min(1.0, std::max(0.0, x)); }

}  // namespace

extern "C" {

double modle_hip_stp_active_from_occupancy(double stp_inactive, double occupancy) {
  // extrusion_barriers_impl.hpp:106-116
  if (occupancy == 0) return 0.0;
  const double tp_inactive_to_active = 1.0 - stp_inactive;
  const double tp_active_to_inactive =
      (tp_inactive_to_active - (occupancy * tp_inactive_to_active)) / occupancy;
  return clamp01(1.0 - tp_active_to_inactive);
}

double modle_hip_occupancy_from_stp(double stp_active, double stp_inactive) {
  // extrusion_barriers_impl.hpp:118-128
  if (stp_active + stp_inactive == 0) return 0.0;
  const double a = 1.0 - stp_inactive;
  const double b = 1.0 - stp_active;
  return clamp01(a / (a + b));
}

void modle_hip_config_default(modle_hip_config* c) {
  // simulation_config.hpp:47-113
  std::memset(c, 0, sizeof(*c));
  c->bin_size = 5000;
  c->diagonal_width = 3000000;
  c->fwd_extrusion_speed = c->bin_size * 8 / 10;
  c->rev_extrusion_speed = c->fwd_extrusion_speed;
  c->fwd_extrusion_speed_std = 0.05;
  c->rev_extrusion_speed_std = 0.05;
  c->rev_extrusion_speed_burnin = c->rev_extrusion_speed;
  c->fwd_extrusion_speed_burnin = c->fwd_extrusion_speed;
  c->hard_stall_lef_stability_multiplier = 5.0;
  c->soft_stall_lef_stability_multiplier = 1.0;
  c->probability_of_extrusion_unit_bypass = 0.1;
  c->lef_bar_major_collision_pblock = 1.0;
  c->lef_bar_minor_collision_pblock = 0.0;
  c->contact_sampling_interval = 50000;
  c->contact_sampling_strategy = MODLE_HIP_CS_TAD | MODLE_HIP_CS_LOOP | MODLE_HIP_CS_NOISIFY;
  c->tad_to_loop_contact_ratio = 5.0;
  c->genextreme_mu = 0;
  c->genextreme_sigma = 5000;
  c->genextreme_xi = 0.001;
  c->target_contact_density = 1.0;
  c->target_simulation_epochs = 2000;
  c->skip_burnin = 0;
  c->burnin_history_length = 100;
  c->burnin_smoothing_window_size = 5;
  c->min_burnin_epochs = 0;
  c->max_burnin_epochs = std::numeric_limits<uint64_t>::max();
  c->burnin_target_epochs_for_lef_activation = 320;
  c->track_1d_lef_position = 1;
  c->number_of_lefs_per_mbp = 20;
  c->num_cells = 512;
  c->seed = 0;
  c->simulate_chromosomes_wo_barriers = 0;
  c->avg_lef_processivity = 300000;
  c->burnin_speed_coefficient = 1.0;
  c->extrusion_barrier_occupancy = 0.825;
  c->barrier_occupied_stp = 0.0;
  c->barrier_not_occupied_stp = 0.70;
  c->probability_normalization_factor = c->rev_extrusion_speed + c->fwd_extrusion_speed;
  c->normalize_probabilities = 1;
}

int modle_hip_config_transform(modle_hip_config* c, char* err, size_t errlen) {
  if (c == nullptr || c->bin_size == 0 || c->avg_lef_processivity == 0) {
    set_err(err, errlen, "invalid config");
    return MODLE_HIP_ERR_ARG;
  }
  // cli_update_extr_speed (cli.cpp:886-911)
  if (!c->rev_extrusion_speed_set) c->rev_extrusion_speed = c->bin_size * 8 / 10;
  if (!c->fwd_extrusion_speed_set) c->fwd_extrusion_speed = c->bin_size * 8 / 10;
  if (c->fwd_extrusion_speed_std > 0 && c->fwd_extrusion_speed_std < 1)
    c->fwd_extrusion_speed_std *= static_cast<double>(c->fwd_extrusion_speed);
  if (c->rev_extrusion_speed_std > 0 && c->rev_extrusion_speed_std < 1)
    c->rev_extrusion_speed_std *= static_cast<double>(c->rev_extrusion_speed);
  c->rev_extrusion_speed_burnin = static_cast<uint64_t>(
      std::round(c->burnin_speed_coefficient * static_cast<double>(c->rev_extrusion_speed)));
  c->fwd_extrusion_speed_burnin = static_cast<uint64_t>(
      std::round(c->burnin_speed_coefficient * static_cast<double>(c->fwd_extrusion_speed)));
  // cli_compute_prob_of_lef_release (cli.cpp:914-920)
  c->prob_of_lef_release = static_cast<double>(c->rev_extrusion_speed + c->fwd_extrusion_speed) /
                           static_cast<double>(c->avg_lef_processivity);
  c->prob_of_lef_release_burnin =
      static_cast<double>(c->rev_extrusion_speed_burnin + c->fwd_extrusion_speed_burnin) /
      static_cast<double>(c->avg_lef_processivity);
  // cli_update_barrier_stp_and_occupancy (cli.cpp:923-936)
  if (c->extrusion_barrier_occupancy_set) {
    c->barrier_occupied_stp = modle_hip_stp_active_from_occupancy(c->barrier_not_occupied_stp,
                                                                  c->extrusion_barrier_occupancy);
  } else {
    c->extrusion_barrier_occupancy =
        modle_hip_occupancy_from_stp(c->barrier_occupied_stp, c->barrier_not_occupied_stp);
  }
  // cli_update_tad_to_loop_contact_ratio (cli.cpp:970-983)
  const bool loop = (c->contact_sampling_strategy & MODLE_HIP_CS_LOOP) != 0;
  const bool tad = (c->contact_sampling_strategy & MODLE_HIP_CS_TAD) != 0;
  if (!loop && !tad) {
    set_err(err, errlen, "contact_sampling_strategy must include loop and/or tad sampling");
    return MODLE_HIP_ERR_ARG;
  }
  if (loop && !tad) c->tad_to_loop_contact_ratio = 0;
  if (!loop && tad) c->tad_to_loop_contact_ratio = std::numeric_limits<double>::infinity();
  // cli_update_burnin_params (cli.cpp:985-991)
  const uint64_t burnin_speed = c->rev_extrusion_speed_burnin + c->fwd_extrusion_speed_burnin;
  if (burnin_speed == 0) {
    set_err(err, errlen, "burn-in extrusion speed must be positive");
    return MODLE_HIP_ERR_ARG;
  }
  c->burnin_target_epochs_for_lef_activation =
      std::min<uint64_t>(c->max_burnin_epochs, 5 * c->avg_lef_processivity / burnin_speed);
  // cli_normalize_probabilities (cli.cpp:939-968)
  if (c->normalize_probabilities) {
    const double ratio = static_cast<double>(c->rev_extrusion_speed + c->fwd_extrusion_speed) /
                         static_cast<double>(c->probability_normalization_factor);
    if (ratio != 1.0) {
      auto stable_pow = [](double base, double exp) {
        if (base == 0.0) return 0.0;
        if (base == 1.0) return 1.0;
        return std::exp(std::log(base) * exp);
      };
      c->barrier_not_occupied_stp = stable_pow(c->barrier_not_occupied_stp, ratio);
      c->barrier_occupied_stp = modle_hip_stp_active_from_occupancy(
          c->barrier_not_occupied_stp, c->extrusion_barrier_occupancy);
      const double p = c->probability_of_extrusion_unit_bypass;
      if (p != 0.0 && p != 1.0) c->probability_of_extrusion_unit_bypass = std::min(p * ratio, 1.0);
      c->lef_bar_major_collision_pblock = stable_pow(c->lef_bar_major_collision_pblock, ratio);
      c->lef_bar_minor_collision_pblock = stable_pow(c->lef_bar_minor_collision_pblock, ratio);
    }
  }
  return MODLE_HIP_OK;
}

uint64_t modle_hip_interval_hash(const char* chrom_name, uint64_t chrom_size, uint64_t start,
                                 uint64_t end, uint64_t seed) {
  uint8_t buf[240];
  size_t n = std::strlen(chrom_name);
  n = std::min<size_t>(n, sizeof(buf) - 24);
  std::memcpy(buf, chrom_name, n);
  std::memcpy(buf + n, &chrom_size, 8);
  std::memcpy(buf + n + 8, &start, 8);
  std::memcpy(buf + n + 16, &end, 8);
  return Xxh3::hash(buf, n + 24, seed);
}

void modle_hip_prng_seed(uint64_t seed, uint64_t state[4]) { modle_host::splitmix_seed(seed, state); }
void modle_hip_prng_jump(uint64_t state[4]) { modle_host::xoshiro_jump(state); }

uint64_t modle_hip_compute_num_lefs(const modle_hip_config* c, uint64_t size_bp) {
  const double size_mbp = static_cast<double>(size_bp) / 1.0e6;
  return std::max<uint64_t>(1, static_cast<uint64_t>(std::round(c->number_of_lefs_per_mbp * size_mbp)));
}

uint64_t modle_hip_compute_contacts_per_epoch(const modle_hip_config* c, uint64_t nlefs) {
  const double speed = static_cast<double>(c->rev_extrusion_speed + c->fwd_extrusion_speed);
  const double prob = speed / static_cast<double>(c->contact_sampling_interval);
  return static_cast<uint64_t>(std::max(1.0, std::round(static_cast<double>(nlefs) * prob)));
}

void modle_hip_matrix_shape(const modle_hip_config* c, uint64_t size_bp, uint64_t* nrows,
                            uint64_t* ncols) {
  const uint64_t nr = (c->diagonal_width + c->bin_size - 1) / c->bin_size;
  const uint64_t nc = (size_bp + c->bin_size - 1) / c->bin_size;
  *nrows = std::min(nr, nc);
  *ncols = nc;
}

int modle_hip_make_tasks(const modle_hip_config* c, const char* chrom_name, uint64_t chrom_size,
                         uint64_t start, uint64_t end, uint64_t first_task_id,
                         modle_hip_task* tasks) {
  if (c == nullptr || chrom_name == nullptr || tasks == nullptr || end <= start ||
      c->num_cells == 0) {
    return MODLE_HIP_ERR_ARG;
  }
  uint64_t state[4];
  modle_host::splitmix_seed(modle_hip_interval_hash(chrom_name, chrom_size, start, end, c->seed),
                            state);
  const uint64_t nlefs = modle_hip_compute_num_lefs(c, end - start);
  uint64_t nrows = 0, ncols = 0;
  modle_hip_matrix_shape(c, end - start, &nrows, &ncols);
  const uint64_t npixels = nrows * ncols;
  const auto tot_target_contacts =
      static_cast<uint64_t>(std::round(static_cast<double>(npixels) * c->target_contact_density));
  const uint64_t per_cell = (tot_target_contacts + c->num_cells - 1) / c->num_cells;
  uint64_t rolling = 0;
  for (uint64_t cell = 0; cell < c->num_cells; ++cell) {
    const uint64_t n = std::min(per_cell, tot_target_contacts - rolling);
    rolling += n;
    modle_hip_task& t = tasks[cell];
    t.id = first_task_id + cell;
    t.cell_id = cell;
    t.num_target_epochs = c->target_simulation_epochs;
    t.num_target_contacts = n;
    t.num_lefs = nlefs;
    std::memcpy(t.prng, state, sizeof(state));
    modle_host::xoshiro_jump(state);
  }
  return MODLE_HIP_OK;
}

int modle_hip_size_class(const modle_hip_config* c, uint64_t max_lefs) {
  // (the rule: modle_host::size_class_required, launch_common.hpp)
  if (c == nullptr) return 1;
  if (const char* e = std::getenv("MODLE_HIP_SIZE_CLASS"); e != nullptr && e[0] == 'w') return 1;
  return modle_host::size_class_required(*c, max_lefs);
}

void modle_hip_sort_barriers(uint64_t* bar_pos, uint8_t* bar_dir, double* bar_stp_active,
                             double* bar_stp_inactive, size_t n_barriers) {
  if (n_barriers < 2 || bar_pos == nullptr) return;
  bool sorted = true;
  for (size_t i = 1; i < n_barriers && sorted; ++i) sorted = bar_pos[i - 1] <= bar_pos[i];
  if (sorted) return;
  std::vector<size_t> idx(n_barriers);
  for (size_t i = 0; i < n_barriers; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(),
                   [&](size_t a, size_t b) { return bar_pos[a] < bar_pos[b]; });
  const auto permute = [&](auto* v) {
    if (v == nullptr) return;
    std::vector<std::remove_reference_t<decltype(*v)>> tmp(n_barriers);
    for (size_t i = 0; i < n_barriers; ++i) tmp[i] = v[idx[i]];
    std::copy(tmp.begin(), tmp.end(), v);
  };
  permute(bar_pos);
  permute(bar_dir);
  permute(bar_stp_active);
  permute(bar_stp_inactive);
}

}  // extern "C"
