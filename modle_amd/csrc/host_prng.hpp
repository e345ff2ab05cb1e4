// host_prng.hpp -- host-side xoshiro256++ helpers (seeding, 2^128 jump, arbitrary-stride jump
// tables for the device-side block generator).
//
// Reference: random::PRNG (src/common/include/modle/common/random.hpp:26-32) wraps
// XoshiroCpp::Xoshiro256PlusPlus seeded with four SplitMix64 outputs; the scheduler separates
// cells with PRNG_t::jump() (src/libmodle/cpu/scheduler_simulate.cpp:158).  xoshiro-cpp is not
// vendored in the reference; the algorithms are Blackman & Vigna's public-domain generators.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace modle_host {

inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

inline void splitmix_seed(uint64_t seed, uint64_t s[4]) {
  for (int i = 0; i < 4; ++i) {
    uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    s[i] = z ^ (z >> 31);
  }
}

inline uint64_t xoshiro_next(uint64_t s[4]) {
  const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return result;
}

inline void xoshiro_jump(uint64_t s[4]) {
  static const uint64_t kJump[4] = {0x180ec6d33cfd0abaULL, 0xd5a61266f0c9392cULL,
                                    0xa9582618e03fc9aaULL, 0x39abdc4529b1661cULL};
  uint64_t acc[4] = {0, 0, 0, 0};
  for (uint64_t word : kJump) {
    for (int b = 0; b < 64; ++b) {
      if (word & (1ULL << b)) {
        for (int k = 0; k < 4; ++k) acc[k] ^= s[k];
      }
      xoshiro_next(s);
    }
  }
  std::memcpy(s, acc, sizeof(acc));
}

// The state transition of xoshiro256 is linear over GF(2), so "advance by `stride` steps" is a
// fixed 256x256 bit matrix.  The device applies it by table lookup: for each of the 64 nibbles of
// the state, table[nibble][value] is the image of the state whose only non-zero nibble is
// `value`; the new state is the XOR of the 64 looked-up rows.  Layout: [64][16][4] uint64 (32 KiB).
inline std::vector<uint64_t> build_jump_table(uint64_t stride) {
  std::vector<uint64_t> basis(256 * 4);  // image of each unit vector
  for (int bit = 0; bit < 256; ++bit) {
    uint64_t s[4] = {0, 0, 0, 0};
    s[bit / 64] = 1ULL << (bit % 64);
    for (uint64_t k = 0; k < stride; ++k) xoshiro_next(s);
    std::memcpy(&basis[bit * 4], s, sizeof(s));
  }
  std::vector<uint64_t> table(64 * 16 * 4, 0);
  for (int nib = 0; nib < 64; ++nib) {
    for (int v = 0; v < 16; ++v) {
      uint64_t* row = &table[(nib * 16 + v) * 4];
      for (int b = 0; b < 4; ++b) {
        if (v & (1 << b)) {
          for (int k = 0; k < 4; ++k) row[k] ^= basis[(nib * 4 + b) * 4 + k];
        }
      }
    }
  }
  return table;
}

}  // namespace modle_host
