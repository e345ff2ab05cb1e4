// boost_draw_vectors.cpp -- emits the draw vectors that pin the oracle's restatement of
// Boost.Random, for use WHERE BOOST BUILDS (it does not in this repository's image: no Boost
// headers, no network; nothing here stands in for them).
//
//   g++ -O2 -std=c++17 tools/boost_draw_vectors.cpp -o boost_draw_vectors      (Boost >= 1.75 on the include path;
//   ./boost_draw_vectors > tests/golden/boost_draws.json                        the reference pins Boost 1.88)
//
// The reference draws every random number of the path through Boost.Random distributions on an
// xoshiro256++ engine seeded with four SplitMix64 outputs (src/common/include/modle/common/
// random.hpp:26-53, conanfile.py: boost/1.88.0, xoshiro-cpp/1.1).  The reference's own tests hold
// no vector for them, so the oracle's restatement (oracle/modle_oracle.c: mo_normal, mo_poisson,
// mo_binomial, mo_uniform_int, mo_canonical, mo_bernoulli) is "parity unpinned" (DESIGN.md).  This
// program prints, for the two seeds the reference's seed-dependent tests use and for each
// distribution with the parameters of the reference's call sites, the first N values and the
// number of 64-bit engine outputs consumed; tests/test_oracle_distributions.py compares the oracle
// with the file when it is present.
//
// Plain standard C++ plus Boost: the engine below is Blackman & Vigna's public-domain xoshiro256++
// and SplitMix64 (what xoshiro-cpp implements), written out so that Boost is the only dependency.
#include <boost/random/bernoulli_distribution.hpp>
#include <boost/random/binomial_distribution.hpp>
#include <boost/random/generate_canonical.hpp>
#include <boost/random/normal_distribution.hpp>
#include <boost/random/poisson_distribution.hpp>
#include <boost/random/uniform_int_distribution.hpp>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>

struct Xoshiro256pp {
  using result_type = std::uint64_t;
  std::uint64_t s[4];
  std::uint64_t count = 0;  // engine outputs consumed
  explicit Xoshiro256pp(std::uint64_t seed) {
    for (auto& w : s) {  // SplitMix64, as XoshiroCpp::SplitMix64::generateSeedSequence<4>
      std::uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
      w = z ^ (z >> 31);
    }
  }
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return std::numeric_limits<result_type>::max(); }
  static std::uint64_t rotl(std::uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  result_type operator()() {
    ++count;
    const std::uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    const std::uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
  }
};

static std::uint64_t bits_of(double x) {
  std::uint64_t u;
  std::memcpy(&u, &x, sizeof(u));
  return u;
}

constexpr int N = 10000;
static bool first_entry = true;

// values are printed as unsigned 64-bit integers: integer draws as they are, doubles as their bit
// images (exact, locale-free)
template <class Draw>
static void emit(const char* name, const char* params, std::uint64_t seed, Draw draw) {
  Xoshiro256pp eng(seed);
  std::printf("%s\n  {\"distribution\": \"%s\", \"params\": %s, \"seed\": %llu, \"values\": [", first_entry ? "" : ",", name,
              params, static_cast<unsigned long long>(seed));
  first_entry = false;
  for (int i = 0; i < N; ++i) std::printf("%s%llu", i ? "," : "", static_cast<unsigned long long>(draw(eng)));
  std::printf("], \"engine_outputs_consumed\": %llu}", static_cast<unsigned long long>(eng.count));
}

int main() {
  const std::uint64_t seeds[2] = {752741483ULL, 10556020843759504871ULL};  // simulation_complex_unit_test.cpp ("Simulation 011/012")
  std::printf("{\"boost_version\": %d, \"n\": %d, \"entries\": [", BOOST_VERSION, N);
  for (const std::uint64_t seed : seeds) {
    // generate_moves: normal_distribution<double>{speed, std} (simulation.cpp:291; defaults 4000, 200)
    emit("normal", "{\"mean\": 4000.0, \"sigma\": 200.0, \"value_is\": \"f64 bits\"}", seed, [](Xoshiro256pp& e) {
      return bits_of(boost::random::normal_distribution<double>{4000.0, 200.0}(e));
    });
    // run_burnin: poisson_distribution<size_t>{lef_binding_rate_burnin} (simulation.cpp:871): chr1 26.6 (PTRD), chr21 ~ 5, tiny 0.43 (inversion)
    emit("poisson", "{\"mean\": 26.6}", seed,
         [](Xoshiro256pp& e) { return static_cast<std::uint64_t>(boost::random::poisson_distribution<std::size_t, double>{26.6}(e)); });
    emit("poisson", "{\"mean\": 0.43}", seed,
         [](Xoshiro256pp& e) { return static_cast<std::uint64_t>(boost::random::poisson_distribution<std::size_t, double>{0.43}(e)); });
    // sample_and_register_contacts: binomial_distribution<ptrdiff_t>{n, p} (register_contacts.cpp:89): BTRD and inversion regimes
    emit("binomial", "{\"t\": 797, \"p\": 0.16666666666666666}", seed, [](Xoshiro256pp& e) {
      return static_cast<std::uint64_t>(boost::random::binomial_distribution<std::ptrdiff_t, double>{797, 1.0 / 6.0}(e));
    });
    emit("binomial", "{\"t\": 13, \"p\": 0.16666666666666666}", seed, [](Xoshiro256pp& e) {
      return static_cast<std::uint64_t>(boost::random::binomial_distribution<std::ptrdiff_t, double>{13, 1.0 / 6.0}(e));
    });
    // bind: uniform_int_distribution<bp_t>{start, end - 1} on chr1 (simulation_impl.hpp:47); contact sampling picks a LEF
    // (register_contacts.cpp:68) from a small range
    emit("uniform_int", "{\"lo\": 0, \"hi\": 248956421}", seed, [](Xoshiro256pp& e) {
      return boost::random::uniform_int_distribution<std::uint64_t>{0, 248956421ULL}(e);
    });
    emit("uniform_int", "{\"lo\": 0, \"hi\": 4978}", seed, [](Xoshiro256pp& e) {
      return boost::random::uniform_int_distribution<std::uint64_t>{0, 4978ULL}(e);
    });
    // ExtrusionBarriers::init_states: generate_canonical<double, 53> (extrusion_barriers.cpp:146)
    emit("canonical", "{\"bits\": 53, \"value_is\": \"f64 bits\"}", seed, [](Xoshiro256pp& e) {
      return bits_of(boost::random::generate_canonical<double, std::numeric_limits<double>::digits>(e));
    });
    // bernoulli_trial{p}: LEF-LEF collisions 1 - bypass (simulation_impl.hpp:95), release (simulation.cpp:594)
    emit("bernoulli", "{\"p\": 0.75}", seed,
         [](Xoshiro256pp& e) { return static_cast<std::uint64_t>(boost::random::bernoulli_distribution<double>{0.75}(e)); });
    emit("bernoulli", "{\"p\": 0.026666666666666668}", seed, [](Xoshiro256pp& e) {
      return static_cast<std::uint64_t>(boost::random::bernoulli_distribution<double>{8000.0 / 300000.0}(e));
    });
  }
  std::printf("\n]}\n");
  return 0;
}
