// genome_io.cpp -- chrom.sizes / BED import by the reference's rules (include/modle_genome.h).
// Host only.  Reference: src/libmodle/internal/genome.cpp:299-469, src/libmodle_io/bed.cpp,
// src/libmodle_io/chrom_sizes.cpp.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <string_view>
#include <tuple>
#include <vector>

#include "modle_genome.h"

namespace {

struct ParseError {
  std::string msg;
};

void set_err(char* err, size_t errlen, const std::string& msg) {
  if (err != nullptr && errlen != 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

std::string_view strip_trailing_ws(std::string_view s) {
  while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n' ||
                        s.back() == '\v' || s.back() == '\f'))
    s.remove_suffix(1);
  return s;
}

// utils::strip_quote_pairs (reference: src/common/utils_impl.hpp:204-214)
std::string_view strip_quote_pairs(std::string_view s) {
  if (s.size() < 2) return s;
  const bool b = s.front() == '\'' || s.front() == '"';
  const bool e = s.back() == '\'' || s.back() == '"';
  return (b && e) ? s.substr(1, s.size() - 2) : s;
}

std::vector<std::string_view> lines_of(std::string_view text) {
  std::vector<std::string_view> out;
  size_t pos = 0;
  while (pos <= text.size()) {
    const size_t nl = text.find('\n', pos);
    if (nl == std::string_view::npos) {
      if (pos < text.size()) out.push_back(text.substr(pos));
      break;
    }
    out.push_back(text.substr(pos, nl - pos));
    pos = nl + 1;
  }
  for (auto& l : out)
    if (!l.empty() && l.back() == '\r') l.remove_suffix(1);
  return out;
}

uint64_t parse_u64(std::string_view tok, const char* what) {
  if (tok.empty()) throw ParseError{std::string("missing ") + what};
  uint64_t v = 0;
  for (char ch : tok) {
    if (ch < '0' || ch > '9')
      throw ParseError{std::string("unable to convert \"") + std::string(tok) + "\" to a number (" + what + ")"};
    const uint64_t nv = v * 10 + static_cast<uint64_t>(ch - '0');
    if (nv / 10 != v) throw ParseError{std::string(what) + " out of range"};
    v = nv;
  }
  return v;
}

double parse_f64(std::string_view tok, const char* what) {
  const std::string s(tok);
  char* end = nullptr;
  errno = 0;
  const double v = std::strtod(s.c_str(), &end);
  if (s.empty() || end != s.c_str() + s.size() || errno == ERANGE)
    throw ParseError{std::string("unable to convert \"") + s + "\" to a number (" + what + ")"};
  return v;
}

struct Bed {
  std::string chrom;
  uint64_t start = 0, end = 0;
  std::string name;
  double score = 0.0;
  char strand = '.';
  size_t line = 0;
};

// bed_strand_encoding (reference: src/libmodle_io/include/bed/modle/bed/bed.hpp:251-264)
char parse_strand(std::string_view tok) {
  tok = strip_quote_pairs(tok);
  for (const char* s : {"+", "plus", "fwd", "Fwd", "forward", "Forward", "FWD", "FORWARD"})
    if (tok == s) return '+';
  for (const char* s : {"-", "minus", "rev", "Rev", "reverse", "Reverse", "REV", "REVERSE"})
    if (tok == s) return '-';
  for (const char* s : {".", "", "none", "None", "NONE", "unknown", "Unknown", "unk", "Unk", "UNK"})
    if (tok == s) return '.';
  throw ParseError{"unrecognized strand \"" + std::string(tok) + "\""};
}

// bed::Parser with a fixed dialect (3 or 6 fields) and standard compliance enforced
// (reference: bed.cpp:245-320 record parsing, :428-524 duplicate detection, :567-586 header)
std::vector<Bed> parse_bed(std::string_view text, unsigned min_fields, const char* what) {
  std::vector<Bed> out;
  std::map<std::tuple<std::string, uint64_t, uint64_t>, size_t> seen;
  const auto lines = lines_of(text);
  size_t i = 0;
  // header: leading empty lines, comment lines and track / browser lines
  for (; i < lines.size(); ++i) {
    const std::string_view l = lines[i];
    if (l.empty()) continue;
    if (l.front() == '#' || l.find("track") != std::string_view::npos ||
        l.find("browser") != std::string_view::npos)
      continue;
    break;
  }
  for (; i < lines.size(); ++i) {
    if (lines[i].empty()) continue;  // "look for the next non-empty line"
    const std::string_view rec = strip_trailing_ws(lines[i]);
    std::vector<std::string_view> toks;
    size_t p = 0;
    while (p <= rec.size()) {
      const size_t q = rec.find_first_of("\t ", p);
      const std::string_view tok = rec.substr(p, q == std::string_view::npos ? std::string_view::npos : q - p);
      if (!tok.empty()) toks.push_back(tok);
      if (q == std::string_view::npos) break;
      p = q + 1;
    }
    try {
      if (toks.size() < 3)
        throw ParseError{"expected at least 3 fields, got " + std::to_string(toks.size())};
      if (toks.size() < min_fields)
        throw ParseError{"Invalid BED record detected: Expected BED record with at least " +
                         std::to_string(min_fields) + " fields, got " + std::to_string(toks.size())};
      Bed b;
      b.line = i + 1;
      b.chrom = std::string(strip_quote_pairs(toks[0]));
      b.start = parse_u64(toks[1], "chromStart");
      b.end = parse_u64(toks[2], "chromEnd");
      if (b.start > b.end)
        throw ParseError{"Invalid BED record detected: chrom_start > chrom_end: chrom=\"" + b.chrom +
                         "\"; start=" + std::to_string(b.start) + "; end=" + std::to_string(b.end)};
      if (min_fields >= 6) {
        b.name = std::string(strip_quote_pairs(toks[3]));
        b.score = parse_f64(toks[4], "score");
        if (b.score < 0 || b.score > 1000)
          throw ParseError{"Invalid BED record detected: score field should be between 0.0 and 1000.0"};
        b.strand = parse_strand(toks[5]);
      }
      const auto key = std::make_tuple(b.chrom, b.start, b.end);
      const auto [it, fresh] = seen.emplace(key, b.line);
      if (!fresh)
        throw ParseError{"Detected duplicate record. First occurrence was at line " +
                         std::to_string(it->second)};
      out.push_back(std::move(b));
    } catch (const ParseError& e) {
      throw ParseError{std::string(what) + ", line " + std::to_string(i + 1) + ": " + e.msg +
                       "\n  record: \"" + std::string(rec) + "\""};
    }
  }
  return out;
}

struct Chrom {
  std::string name;
  uint64_t size;
};

// chrom_sizes::Parser::parse_all (reference: chrom_sizes.cpp) + Genome::import_chromosomes
std::vector<Chrom> parse_chrom_sizes(std::string_view text) {
  std::vector<Chrom> out;
  std::set<std::string> names;
  const auto lines = lines_of(text);
  for (size_t i = 0; i < lines.size(); ++i) {
    const std::string_view buff = strip_trailing_ws(lines[i]);
    if (buff.empty()) continue;
    try {
      std::vector<std::string_view> toks;
      size_t p = 0;
      for (;;) {
        const size_t q = buff.find('\t', p);
        toks.push_back(buff.substr(p, q == std::string_view::npos ? std::string_view::npos : q - p));
        if (q == std::string_view::npos) break;
        p = q + 1;
      }
      if (toks.size() != 2)
        throw ParseError{"expected exactly 2 fields, found " + std::to_string(toks.size())};
      const std::string name(strip_quote_pairs(toks[0]));
      if (names.count(name)) throw ParseError{"found multiple records for chrom \"" + name + "\""};
      if (toks[1] == "0") throw ParseError{"chrom \"" + name + "\" has a length of 0bp"};
      const uint64_t size = parse_u64(toks[1], "chromosome size");
      names.insert(name);
      out.push_back(Chrom{name, size});
    } catch (const ParseError& e) {
      throw ParseError{"chrom.sizes: encountered a malformed record at line " + std::to_string(i + 1) +
                       ": " + e.msg + ".\n  line: \"" + std::string(buff) + "\""};
    }
  }
  if (out.empty()) throw ParseError{"Unable to import any chromosome"};
  return out;
}

struct Interval {
  uint64_t chrom_id, start, end;
  std::vector<uint64_t> pos;
  std::vector<uint8_t> dir;
  std::vector<double> stp_active, stp_inactive;
};

}  // namespace

struct modle_genome {
  std::vector<Chrom> chroms;
  std::vector<Interval> intervals;
  uint64_t imported = 0, dropped = 0;
};

extern "C" {

int modle_genome_import(const char* chrom_sizes, size_t chrom_sizes_len, const char* barriers_bed,
                        size_t barriers_bed_len, const char* intervals_bed,
                        size_t intervals_bed_len, const modle_hip_config* cfg,
                        int name_is_not_bound_stp, modle_genome** out, char* err, size_t errlen) {
  if (chrom_sizes == nullptr || barriers_bed == nullptr || cfg == nullptr || out == nullptr) {
    set_err(err, errlen, "modle_genome_import: null argument");
    return MODLE_GENOME_ERR_ARG;
  }
  try {
    auto g = std::make_unique<modle_genome>();
    g->chroms = parse_chrom_sizes(std::string_view(chrom_sizes, chrom_sizes_len));
    std::map<std::string, size_t> chrom_index;
    for (size_t i = 0; i < g->chroms.size(); ++i) chrom_index.emplace(g->chroms[i].name, i);
    // intervals (reference: Genome::import_genomic_intervals, genome.cpp:350-420)
    if (intervals_bed == nullptr || intervals_bed_len == 0) {
      for (size_t i = 0; i < g->chroms.size(); ++i)
        g->intervals.push_back(Interval{i, 0, g->chroms[i].size, {}, {}, {}, {}});
    } else {
      const auto recs = parse_bed(std::string_view(intervals_bed, intervals_bed_len), 3, "genomic intervals");
      for (size_t c = 0; c < g->chroms.size(); ++c) {
        std::vector<Interval> found;
        for (const Bed& b : recs) {
          if (b.chrom != g->chroms[c].name) continue;
          if (b.end > g->chroms[c].size)
            throw ParseError{"genomic intervals, line " + std::to_string(b.line) + ": interval ends beyond the end of " + b.chrom};
          if (b.end == b.start) continue;
          found.push_back(Interval{c, b.start, b.end, {}, {}, {}, {}});
        }
        std::sort(found.begin(), found.end(), [](const Interval& a, const Interval& b) {
          return std::tie(a.start, a.end) < std::tie(b.start, b.end);
        });
        for (auto& iv : found) g->intervals.push_back(std::move(iv));
      }
      if (g->intervals.empty()) throw ParseError{"unable to import any interval"};
    }
    // barriers (reference: generate_barriers_from_bed_records, genome.cpp:423-469)
    const auto bars = parse_bed(std::string_view(barriers_bed, barriers_bed_len), 6, "extrusion barriers");
    const double pbb = cfg->barrier_occupied_stp, puu = cfg->barrier_not_occupied_stp;
    for (const Bed& b : bars) {
      const auto it = chrom_index.find(b.chrom);
      if (it == chrom_index.end()) continue;  // not on a chromosome of the genome: no interval overlaps
      // The reference turns into barriers -- and therefore validates -- only the records its
      // interval tree returns for some simulated interval (map_barriers_to_intervals,
      // genome.cpp:470-489: find_overlaps(chrom, interval.start, interval.end)); a record with a
      // bad score on a stretch that --genomic-intervals leaves out is never looked at.
      bool overlaps_an_interval = false;
      for (const Interval& iv : g->intervals)
        overlaps_an_interval = overlaps_an_interval ||
                               (iv.chrom_id == it->second && b.start < iv.end && iv.start < std::max(b.end, b.start + 1));
      if (!overlaps_an_interval) continue;
      try {
        if (b.strand == '.') {
          ++g->dropped;
          continue;
        }
        if (b.score < 0 || b.score > 1)
          throw ParseError{"invalid score field: expected a score between 0 and 1, found " + std::to_string(b.score)};
        if (name_is_not_bound_stp) {
          double v = -1.0;
          try {
            v = parse_f64(b.name, "name");
          } catch (const ParseError&) {
            v = -1.0;
          }
          if (v < 0 || v > 1)
            throw ParseError{"invalid name field: expected name to be a number between 0 and 1, found " + b.name};
        }
        const uint64_t pos = (b.start + b.end + 1) / 2;
        // compute_barrier_stp (genome.cpp:260-271): score 0 => the default stp of a bound barrier
        const double sa = b.score != 0.0 ? modle_hip_stp_active_from_occupancy(puu, b.score) : pbb;
        for (Interval& iv : g->intervals) {
          if (iv.chrom_id != it->second) continue;
          // A barrier belongs to the interval its position (the record's midpoint) falls in.
          // DELIBERATE DIVERGENCE ("Known divergences"): a record that overlaps an
          // interval while its midpoint lies outside it -- a record straddling the edge of a
          // --genomic-intervals window, or a 1-bp record on the last base of a chromosome, whose
          // midpoint (s + e + 1) / 2 equals the chromosome's end -- is KEPT by the reference's release
          // builds: map_barriers_to_intervals hands every find_overlaps hit to
          // GenomicInterval::add_extrusion_barriers(vector), which only asserts the range (debug
          // builds; genome.cpp:288-297, 470-489).  Such a barrier can never stall a unit (units stay
          // inside the interval) but it draws a state per epoch, so against a release reference the
          // PRNG streams of that interval's cells differ.  Here it is left out: the device layout
          // takes barrier positions inside the interval only (modle_hip_add_interval).
          if (pos < iv.start || pos >= iv.end) continue;
          iv.pos.push_back(pos);
          iv.dir.push_back(b.strand == '+' ? MODLE_HIP_DIR_REV : MODLE_HIP_DIR_FWD);
          iv.stp_active.push_back(sa);
          iv.stp_inactive.push_back(puu);
          ++g->imported;
        }
      } catch (const ParseError& e) {
        throw ParseError{"found invalid extrusion barrier " + b.chrom + ":" + std::to_string(b.start) + "-" +
                         std::to_string(b.end) + " (line " + std::to_string(b.line) + "): " + e.msg};
      }
    }
    *out = g.release();
    return MODLE_GENOME_OK;
  } catch (const ParseError& e) {
    set_err(err, errlen, e.msg);
    return MODLE_GENOME_ERR_PARSE;
  } catch (const std::exception& e) {
    set_err(err, errlen, e.what());
    return MODLE_GENOME_ERR_ARG;
  }
}

void modle_genome_free(modle_genome* g) { delete g; }

size_t modle_genome_num_chromosomes(const modle_genome* g) { return g ? g->chroms.size() : 0; }

int modle_genome_chromosome(const modle_genome* g, size_t i, const char** name, uint64_t* size) {
  if (g == nullptr || i >= g->chroms.size()) return MODLE_GENOME_ERR_ARG;
  if (name) *name = g->chroms[i].name.c_str();
  if (size) *size = g->chroms[i].size;
  return MODLE_GENOME_OK;
}

size_t modle_genome_num_intervals(const modle_genome* g) { return g ? g->intervals.size() : 0; }

int modle_genome_interval_info(const modle_genome* g, size_t i, modle_genome_interval* out) {
  if (g == nullptr || out == nullptr || i >= g->intervals.size()) return MODLE_GENOME_ERR_ARG;
  const Interval& iv = g->intervals[i];
  out->id = i;
  out->chrom_id = iv.chrom_id;
  out->start = iv.start;
  out->end = iv.end;
  out->num_barriers = iv.pos.size();
  return MODLE_GENOME_OK;
}

int modle_genome_interval_barriers(const modle_genome* g, size_t i, uint64_t* pos, uint8_t* dir,
                                   double* stp_active, double* stp_inactive) {
  if (g == nullptr || i >= g->intervals.size()) return MODLE_GENOME_ERR_ARG;
  const Interval& iv = g->intervals[i];
  if (pos) std::copy(iv.pos.begin(), iv.pos.end(), pos);
  if (dir) std::copy(iv.dir.begin(), iv.dir.end(), dir);
  if (stp_active) std::copy(iv.stp_active.begin(), iv.stp_active.end(), stp_active);
  if (stp_inactive) std::copy(iv.stp_inactive.begin(), iv.stp_inactive.end(), stp_inactive);
  return MODLE_GENOME_OK;
}

void modle_genome_barrier_counts(const modle_genome* g, uint64_t* imported, uint64_t* dropped) {
  if (g == nullptr) return;
  if (imported) *imported = g->imported;
  if (dropped) *dropped = g->dropped;
}

}  // extern "C"
