// bigwig_writer.cpp -- bigWig (version 4) writer for the 1-D LEF occupancy track
// (include/modle_bigwig.h).  Reference: src/libmodle/cpu/simulation.cpp:130-141, 170-197;
// src/libmodle_io/bigwig_impl.hpp:127-158 (libBigWig's bwAddIntervalSpanSteps underneath).
// The layout follows the published format (Kent et al. 2010, supplementary tables).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "modle_bigwig.h"

namespace {

constexpr uint32_t BIGWIG_MAGIC = 0x888FFC26u;
constexpr uint32_t BPT_MAGIC = 0x78CA8C91u;
constexpr uint32_t CIRTREE_MAGIC = 0x2468ACE0u;
constexpr uint32_t ITEMS_PER_SECTION = 8186;  // (32768 - 24) / 4: libBigWig's 32 KiB section buffer
constexpr uint32_t RTREE_BLOCK = 256;

void set_err(char* err, size_t errlen, const std::string& msg) {
  if (err != nullptr && errlen != 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

struct Section {
  uint32_t chrom, start, end;
  uint64_t offset, size;
};

struct Buf {
  std::vector<unsigned char> b;
  template <class T>
  void put(T v) {
    unsigned char tmp[sizeof(T)];
    std::memcpy(tmp, &v, sizeof(T));
    b.insert(b.end(), tmp, tmp + sizeof(T));
  }
};

}  // namespace

struct modle_bw_file {
  std::FILE* fp = nullptr;
  std::vector<std::string> names;
  std::vector<uint32_t> sizes;
  std::vector<Section> sections;
  uint64_t data_offset = 0;
  uint32_t max_uncompressed = 0;
  // total summary
  uint64_t bases = 0;
  double vmin = std::numeric_limits<double>::infinity(), vmax = -std::numeric_limits<double>::infinity();
  double sum = 0.0, sumsq = 0.0;
  uint32_t last_chrom = 0, last_end = 0;
  bool failed = false;
};

namespace {

bool write_all(modle_bw_file* f, const void* p, size_t n) {
  if (n != 0 && std::fwrite(p, 1, n, f->fp) != n) f->failed = true;
  return !f->failed;
}

}  // namespace

extern "C" {

int modle_bw_create(const char* path, int force_overwrite, const char* const* chrom_names,
                    const uint32_t* chrom_sizes, size_t n_chroms, modle_bw_file** out, char* err,
                    size_t errlen) {
  if (path == nullptr || chrom_names == nullptr || chrom_sizes == nullptr || n_chroms == 0 ||
      out == nullptr) {
    set_err(err, errlen, "modle_bw_create: invalid argument");
    return MODLE_BW_ERR_ARG;
  }
  if (!force_overwrite) {
    if (std::FILE* probe = std::fopen(path, "rb")) {
      std::fclose(probe);
      set_err(err, errlen, std::string("refusing to overwrite \"") + path + "\"");
      return MODLE_BW_ERR_IO;
    }
  }
  auto* f = new modle_bw_file;
  f->fp = std::fopen(path, "wb");
  if (f->fp == nullptr) {
    set_err(err, errlen, std::string("unable to open \"") + path + "\" for writing");
    delete f;
    return MODLE_BW_ERR_IO;
  }
  size_t key_size = 1;
  for (size_t i = 0; i < n_chroms; ++i) {
    f->names.emplace_back(chrom_names[i]);
    f->sizes.push_back(chrom_sizes[i]);
    key_size = std::max(key_size, f->names.back().size());
  }
  // 64-byte header (patched at close) -- no zoom headers follow
  std::vector<unsigned char> zero(64, 0);
  write_all(f, zero.data(), zero.size());
  // total summary placeholder (40 bytes) right after the header
  write_all(f, zero.data(), 40);
  // chromosome B+ tree: one leaf block holding every chromosome, ids in genome order
  Buf t;
  t.put<uint32_t>(BPT_MAGIC);
  t.put<uint32_t>(static_cast<uint32_t>(n_chroms));  // block size
  t.put<uint32_t>(static_cast<uint32_t>(key_size));
  t.put<uint32_t>(8);  // value size: chromId + chromSize
  t.put<uint64_t>(n_chroms);
  t.put<uint64_t>(0);
  t.put<uint8_t>(1);  // leaf
  t.put<uint8_t>(0);
  t.put<uint16_t>(static_cast<uint16_t>(n_chroms));
  for (size_t i = 0; i < n_chroms; ++i) {
    std::string key = f->names[i];
    key.resize(key_size, '\0');
    t.b.insert(t.b.end(), key.begin(), key.end());
    t.put<uint32_t>(static_cast<uint32_t>(i));
    t.put<uint32_t>(f->sizes[i]);
  }
  write_all(f, t.b.data(), t.b.size());
  // data: section count (patched at close), then the sections
  f->data_offset = 64 + 40 + t.b.size();
  const uint64_t nsec = 0;
  write_all(f, &nsec, 8);
  if (f->failed) {
    set_err(err, errlen, "write error");
    std::fclose(f->fp);
    delete f;
    return MODLE_BW_ERR_IO;
  }
  *out = f;
  return MODLE_BW_OK;
}

int modle_bw_write_range(modle_bw_file* f, size_t chrom_id, const float* values, size_t n_values,
                         uint32_t span, uint32_t step, uint32_t offset, char* err, size_t errlen) {
  if (f == nullptr || chrom_id >= f->names.size() || (values == nullptr && n_values != 0) ||
      span == 0 || step == 0) {
    set_err(err, errlen, "modle_bw_write_range: invalid argument");
    return MODLE_BW_ERR_ARG;
  }
  if (n_values == 0) return MODLE_BW_OK;
  if (!f->sections.empty() &&
      (chrom_id < f->last_chrom || (chrom_id == f->last_chrom && offset < f->last_end))) {
    set_err(err, errlen, "modle_bw_write_range: ranges must be appended in genome order");
    return MODLE_BW_ERR_ARG;
  }
  const uint64_t last_start = static_cast<uint64_t>(offset) + (n_values - 1) * static_cast<uint64_t>(step);
  if (last_start >= f->sizes[chrom_id]) {
    set_err(err, errlen, "modle_bw_write_range: the range does not fit the chromosome");
    return MODLE_BW_ERR_ARG;
  }
  for (size_t first = 0; first < n_values; first += ITEMS_PER_SECTION) {
    const uint32_t cnt = static_cast<uint32_t>(std::min<size_t>(ITEMS_PER_SECTION, n_values - first));
    const uint32_t s0 = offset + static_cast<uint32_t>(first) * step;
    const uint32_t s1 = std::min<uint64_t>(static_cast<uint64_t>(s0) + static_cast<uint64_t>(cnt - 1) * step + span,
                                           f->sizes[chrom_id]);
    Buf sec;
    sec.put<uint32_t>(static_cast<uint32_t>(chrom_id));
    sec.put<uint32_t>(s0);
    sec.put<uint32_t>(s1);
    sec.put<uint32_t>(step);
    sec.put<uint32_t>(span);
    sec.put<uint8_t>(3);  // fixedStep
    sec.put<uint8_t>(0);
    sec.put<uint16_t>(static_cast<uint16_t>(cnt));
    for (uint32_t i = 0; i < cnt; ++i) {
      const float v = values[first + i];
      sec.put<float>(v);
      const uint64_t b0 = static_cast<uint64_t>(s0) + static_cast<uint64_t>(i) * step;
      const uint64_t cov = std::min<uint64_t>(span, f->sizes[chrom_id] - b0);
      f->bases += cov;
      f->vmin = std::min<double>(f->vmin, v);
      f->vmax = std::max<double>(f->vmax, v);
      f->sum += static_cast<double>(v) * static_cast<double>(cov);
      f->sumsq += static_cast<double>(v) * static_cast<double>(v) * static_cast<double>(cov);
    }
    f->max_uncompressed = std::max<uint32_t>(f->max_uncompressed, static_cast<uint32_t>(sec.b.size()));
    uLongf clen = compressBound(static_cast<uLong>(sec.b.size()));
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, sec.b.data(), static_cast<uLong>(sec.b.size()), Z_DEFAULT_COMPRESSION) != Z_OK) {
      set_err(err, errlen, "zlib failure");
      return MODLE_BW_ERR_IO;
    }
    const long pos = std::ftell(f->fp);
    if (pos < 0 || !write_all(f, comp.data(), clen)) {
      set_err(err, errlen, "write error");
      return MODLE_BW_ERR_IO;
    }
    f->sections.push_back(Section{static_cast<uint32_t>(chrom_id), s0, s1, static_cast<uint64_t>(pos), clen});
    f->last_chrom = static_cast<uint32_t>(chrom_id);
    f->last_end = s1;
  }
  return MODLE_BW_OK;
}

int modle_bw_write_occupancy(modle_bw_file* f, size_t chrom_id, const uint64_t* occupancy,
                             size_t n_bins, uint32_t bin_size, uint32_t offset_bp, char* err,
                             size_t errlen) {
  if (f == nullptr || (occupancy == nullptr && n_bins != 0)) {
    set_err(err, errlen, "modle_bw_write_occupancy: invalid argument");
    return MODLE_BW_ERR_ARG;
  }
  if (n_bins == 0) return MODLE_BW_OK;
  // simulation.cpp:180-189: value / max as double, then float
  const uint64_t mx = *std::max_element(occupancy, occupancy + n_bins);
  std::vector<float> vals(n_bins);
  for (size_t i = 0; i < n_bins; ++i)
    vals[i] = static_cast<float>(static_cast<double>(occupancy[i]) / static_cast<double>(mx));
  return modle_bw_write_range(f, chrom_id, vals.data(), n_bins, bin_size, bin_size, offset_bp, err, errlen);
}

int modle_bw_close(modle_bw_file* f, char* err, size_t errlen) {
  if (f == nullptr) {
    set_err(err, errlen, "modle_bw_close: invalid argument");
    return MODLE_BW_ERR_ARG;
  }
  // R-tree over the sections, bottom-up; nodes are written level by level, root first
  const long index_offset = std::ftell(f->fp);
  struct Node {
    uint32_t c0, s0, c1, s1;
    uint64_t first, count;  // children: [first, first + count) of the level below (or sections)
  };
  std::vector<std::vector<Node>> levels;  // levels[0] = leaves
  {
    std::vector<Node> cur;
    for (size_t i = 0; i < f->sections.size(); i += RTREE_BLOCK) {
      const size_t n = std::min<size_t>(RTREE_BLOCK, f->sections.size() - i);
      cur.push_back(Node{f->sections[i].chrom, f->sections[i].start, f->sections[i + n - 1].chrom,
                         f->sections[i + n - 1].end, i, n});
    }
    if (cur.empty()) cur.push_back(Node{0, 0, 0, 0, 0, 0});
    levels.push_back(cur);
    while (levels.back().size() > 1) {
      const auto& below = levels.back();
      std::vector<Node> up;
      for (size_t i = 0; i < below.size(); i += RTREE_BLOCK) {
        const size_t n = std::min<size_t>(RTREE_BLOCK, below.size() - i);
        up.push_back(Node{below[i].c0, below[i].s0, below[i + n - 1].c1, below[i + n - 1].s1, i, n});
      }
      levels.push_back(up);
    }
  }
  // file offsets of every node: header (48 bytes), then levels from the root down
  std::vector<std::vector<uint64_t>> off(levels.size());
  uint64_t pos = static_cast<uint64_t>(index_offset) + 48;
  for (size_t l = levels.size(); l-- > 0;) {
    for (const Node& n : levels[l]) {
      off[l].push_back(pos);
      pos += 4 + (l == 0 ? 32 : 24) * n.count;
    }
  }
  Buf ix;
  ix.put<uint32_t>(CIRTREE_MAGIC);
  ix.put<uint32_t>(RTREE_BLOCK);
  ix.put<uint64_t>(f->sections.size());
  const Node& root = levels.back()[0];
  ix.put<uint32_t>(root.c0);
  ix.put<uint32_t>(root.s0);
  ix.put<uint32_t>(root.c1);
  ix.put<uint32_t>(root.s1);
  ix.put<uint64_t>(static_cast<uint64_t>(index_offset));  // end of the data = start of the index
  ix.put<uint32_t>(1);                                    // items per slot
  ix.put<uint32_t>(0);
  for (size_t l = levels.size(); l-- > 0;) {
    for (const Node& n : levels[l]) {
      ix.put<uint8_t>(l == 0 ? 1 : 0);
      ix.put<uint8_t>(0);
      ix.put<uint16_t>(static_cast<uint16_t>(n.count));
      for (uint64_t k = 0; k < n.count; ++k) {
        if (l == 0) {
          const Section& s = f->sections[n.first + k];
          ix.put<uint32_t>(s.chrom);
          ix.put<uint32_t>(s.start);
          ix.put<uint32_t>(s.chrom);
          ix.put<uint32_t>(s.end);
          ix.put<uint64_t>(s.offset);
          ix.put<uint64_t>(s.size);
        } else {
          const Node& c = levels[l - 1][n.first + k];
          ix.put<uint32_t>(c.c0);
          ix.put<uint32_t>(c.s0);
          ix.put<uint32_t>(c.c1);
          ix.put<uint32_t>(c.s1);
          ix.put<uint64_t>(off[l - 1][n.first + k]);
        }
      }
    }
  }
  write_all(f, ix.b.data(), ix.b.size());
  const uint32_t magic = BIGWIG_MAGIC;
  write_all(f, &magic, 4);
  // patch: section count, total summary, header
  const uint64_t nsec = f->sections.size();
  std::fseek(f->fp, static_cast<long>(f->data_offset), SEEK_SET);
  write_all(f, &nsec, 8);
  Buf sm;
  sm.put<uint64_t>(f->bases);
  sm.put<double>(f->sections.empty() ? 0.0 : f->vmin);
  sm.put<double>(f->sections.empty() ? 0.0 : f->vmax);
  sm.put<double>(f->sum);
  sm.put<double>(f->sumsq);
  std::fseek(f->fp, 64, SEEK_SET);
  write_all(f, sm.b.data(), sm.b.size());
  Buf h;
  h.put<uint32_t>(BIGWIG_MAGIC);
  h.put<uint16_t>(4);   // version
  h.put<uint16_t>(0);   // zoom levels
  h.put<uint64_t>(64 + 40);                              // chromosome tree
  h.put<uint64_t>(f->data_offset);                       // full data
  h.put<uint64_t>(static_cast<uint64_t>(index_offset));  // full index
  h.put<uint16_t>(0);   // field count
  h.put<uint16_t>(0);   // defined field count
  h.put<uint64_t>(0);   // autoSql
  h.put<uint64_t>(64);  // total summary
  h.put<uint32_t>(f->max_uncompressed);
  h.put<uint64_t>(0);   // extension
  std::fseek(f->fp, 0, SEEK_SET);
  write_all(f, h.b.data(), h.b.size());
  const bool ok = !f->failed && std::fclose(f->fp) == 0;
  delete f;
  if (!ok) {
    set_err(err, errlen, "write error while closing the bigWig file");
    return MODLE_BW_ERR_IO;
  }
  return MODLE_BW_OK;
}

}  // extern "C"
