// cooler_writer.cpp -- cooler v3 writer behind include/modle_cooler.h (host side, HDF5 C API).
//
// Layout written (the mandatory groups / datasets of hictk 2.1.4, cooler/impl/file_write_impl.hpp:
// 225-290, with its types): chroms/{name (fixed-length string), length (int32)},
// bins/{chrom, start, end (int32)}, pixels/{bin1_id, bin2_id (int64), count (int32)},
// indexes/{bin1_offset, chrom_offset (int64)}; chunked, deflate level 6; attributes of
// cooler/cooler.hpp:50-73 / file_write_impl.hpp:297-330.
#include <hdf5.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <limits>
#include <string>
#include <vector>

#include "modle_cooler.h"

namespace {

void set_err(char* err, size_t errlen, const std::string& msg) {
  if (err != nullptr && errlen != 0) std::snprintf(err, errlen, "%s", msg.c_str());
}

constexpr hsize_t kChunkBytes = 64u << 10;  // hictk DEFAULT_HDF5_CHUNK_SIZE
constexpr unsigned kDeflate = 6;            // hictk DEFAULT_COMPRESSION_LEVEL

struct H5Id {  // closes on scope exit
  hid_t id = -1;
  int (*closer)(hid_t) = nullptr;
  H5Id(hid_t i, int (*c)(hid_t)) : id(i), closer(c) {}
  H5Id(const H5Id&) = delete;
  H5Id& operator=(const H5Id&) = delete;
  ~H5Id() {
    if (id >= 0 && closer != nullptr) closer(id);
  }
  operator hid_t() const { return id; }
};

// extendable 1-D dataset of `type`, chunked and compressed
hid_t create_dataset(hid_t file, const char* path, hid_t type) {
  const hsize_t dims[1] = {0}, maxdims[1] = {H5S_UNLIMITED};
  H5Id space(H5Screate_simple(1, dims, maxdims), H5Sclose);
  H5Id cprop(H5Pcreate(H5P_DATASET_CREATE), H5Pclose);
  const hsize_t chunk[1] = {std::max<hsize_t>(1, kChunkBytes / H5Tget_size(type))};
  if (space < 0 || cprop < 0 || H5Pset_chunk(cprop, 1, chunk) < 0 || H5Pset_deflate(cprop, kDeflate) < 0)
    return -1;
  return H5Dcreate2(file, path, type, space, H5P_DEFAULT, cprop, H5P_DEFAULT);
}

bool append(hid_t dset, hid_t memtype, const void* data, hsize_t n, hsize_t& size) {
  if (n == 0) return true;
  const hsize_t newsize[1] = {size + n};
  if (H5Dset_extent(dset, newsize) < 0) return false;
  H5Id fspace(H5Dget_space(dset), H5Sclose);
  const hsize_t start[1] = {size}, count[1] = {n};
  if (fspace < 0 || H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, nullptr, count, nullptr) < 0)
    return false;
  H5Id mspace(H5Screate_simple(1, count, nullptr), H5Sclose);
  if (mspace < 0 || H5Dwrite(dset, memtype, mspace, fspace, H5P_DEFAULT, data) < 0) return false;
  size += n;
  return true;
}

bool write_attr_scalar(hid_t loc, const char* name, hid_t filetype, hid_t memtype, const void* v) {
  H5Id space(H5Screate(H5S_SCALAR), H5Sclose);
  H5Id attr(H5Acreate2(loc, name, filetype, space, H5P_DEFAULT, H5P_DEFAULT), H5Aclose);
  return attr >= 0 && H5Awrite(attr, memtype, v) >= 0;
}

bool write_attr_string(hid_t loc, const char* name, const std::string& v) {
  H5Id type(H5Tcopy(H5T_C_S1), H5Tclose);
  if (type < 0 || H5Tset_size(type, H5T_VARIABLE) < 0 || H5Tset_cset(type, H5T_CSET_UTF8) < 0)
    return false;
  const char* p = v.c_str();
  return write_attr_scalar(loc, name, type, type, &p);
}

}  // namespace

struct modle_cool_file {
  hid_t file = -1;
  hid_t d_bin1 = -1, d_bin2 = -1, d_count = -1;
  hsize_t n_pixels = 0;
  uint32_t bin_size = 0;
  std::vector<uint32_t> chrom_sizes;
  std::vector<int64_t> chrom_offset;  // first bin id of every chromosome, + total
  std::vector<int64_t> bin1_offset;   // filled while pixels are appended; nbins + 1 entries
  int64_t next_bin1 = 0;              // bins below it have their offset
  size_t next_chrom = 0;
  int64_t sum = 0, cis = 0;
  std::string assembly, generated_by, metadata;
};

namespace {

void destroy(modle_cool_file* f) {
  if (f == nullptr) return;
  for (hid_t d : {f->d_bin1, f->d_bin2, f->d_count})
    if (d >= 0) H5Dclose(d);
  if (f->file >= 0) H5Fclose(f->file);
  delete f;
}

template <class T>
bool write_whole(hid_t file, const char* path, hid_t filetype, hid_t memtype, const std::vector<T>& v) {
  H5Id d(create_dataset(file, path, filetype), H5Dclose);
  hsize_t size = 0;
  return d >= 0 && append(d, memtype, v.data(), v.size(), size);
}

}  // namespace

extern "C" int modle_cool_create(const char* path, int force_overwrite,
                                 const char* const* chrom_names, const uint32_t* chrom_sizes,
                                 size_t n_chroms, uint32_t bin_size, const char* assembly,
                                 const char* generated_by, const char* metadata_json,
                                 modle_cool_file** out, char* err, size_t errlen) {
  if (path == nullptr || chrom_names == nullptr || chrom_sizes == nullptr || n_chroms == 0 ||
      bin_size == 0 || out == nullptr || assembly == nullptr || generated_by == nullptr ||
      assembly[0] == '\0' || generated_by[0] == '\0') {
    set_err(err, errlen, "modle_cool_create: invalid argument");
    return MODLE_COOL_ERR_ARG;
  }
  *out = nullptr;
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);  // errors are reported through return codes
  auto* f = new modle_cool_file;
  f->file = H5Fcreate(path, force_overwrite ? H5F_ACC_TRUNC : H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
  if (f->file < 0) {
    set_err(err, errlen, std::string("cannot create \"") + path + "\"" +
                             (force_overwrite ? "" : " (file exists? pass force_overwrite)"));
    destroy(f);
    return MODLE_COOL_ERR_IO;
  }
  f->bin_size = bin_size;
  f->assembly = assembly;
  f->generated_by = generated_by;
  f->metadata = (metadata_json != nullptr && metadata_json[0] != '\0') ? metadata_json : "{}";
  f->chrom_sizes.assign(chrom_sizes, chrom_sizes + n_chroms);

  bool ok = true;
  for (const char* g : {"chroms", "bins", "pixels", "indexes"}) {
    H5Id grp(H5Gcreate2(f->file, g, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), H5Gclose);
    ok = ok && grp >= 0;
  }
  // chroms
  size_t longest = 1;
  for (size_t i = 0; i < n_chroms; ++i) {
    if (chrom_names[i] == nullptr || chrom_names[i][0] == '\0' ||
        chrom_sizes[i] > static_cast<uint32_t>(std::numeric_limits<int32_t>::max())) {
      set_err(err, errlen, "modle_cool_create: invalid chromosome name or size");
      destroy(f);
      return MODLE_COOL_ERR_ARG;
    }
    longest = std::max(longest, std::strlen(chrom_names[i]));
  }
  {
    std::vector<char> names(n_chroms * longest, '\0');
    for (size_t i = 0; i < n_chroms; ++i) std::memcpy(&names[i * longest], chrom_names[i], std::strlen(chrom_names[i]));
    H5Id stype(H5Tcopy(H5T_C_S1), H5Tclose);
    ok = ok && stype >= 0 && H5Tset_size(stype, longest) >= 0 && H5Tset_strpad(stype, H5T_STR_NULLPAD) >= 0;
    H5Id d(ok ? create_dataset(f->file, "chroms/name", stype) : -1, H5Dclose);
    hsize_t size = 0;
    ok = ok && d >= 0 && append(d, stype, names.data(), n_chroms, size);
    std::vector<int32_t> lengths(f->chrom_sizes.begin(), f->chrom_sizes.end());
    ok = ok && write_whole(f->file, "chroms/length", H5T_STD_I32LE, H5T_NATIVE_INT32, lengths);
  }
  // bins (fixed size; the last bin of a chromosome is shorter)
  {
    std::vector<int32_t> chrom, start, end;
    f->chrom_offset.push_back(0);
    for (size_t i = 0; i < n_chroms; ++i) {
      for (uint64_t s = 0; s < chrom_sizes[i]; s += bin_size) {
        chrom.push_back(static_cast<int32_t>(i));
        start.push_back(static_cast<int32_t>(s));
        end.push_back(static_cast<int32_t>(std::min<uint64_t>(s + bin_size, chrom_sizes[i])));
      }
      f->chrom_offset.push_back(static_cast<int64_t>(chrom.size()));
    }
    ok = ok && write_whole(f->file, "bins/chrom", H5T_STD_I32LE, H5T_NATIVE_INT32, chrom);
    ok = ok && write_whole(f->file, "bins/start", H5T_STD_I32LE, H5T_NATIVE_INT32, start);
    ok = ok && write_whole(f->file, "bins/end", H5T_STD_I32LE, H5T_NATIVE_INT32, end);
    f->bin1_offset.assign(chrom.size() + 1, 0);
  }
  f->d_bin1 = create_dataset(f->file, "pixels/bin1_id", H5T_STD_I64LE);
  f->d_bin2 = create_dataset(f->file, "pixels/bin2_id", H5T_STD_I64LE);
  f->d_count = create_dataset(f->file, "pixels/count", H5T_STD_I32LE);
  ok = ok && f->d_bin1 >= 0 && f->d_bin2 >= 0 && f->d_count >= 0;
  if (!ok) {
    set_err(err, errlen, std::string("HDF5 error while initialising \"") + path + "\"");
    destroy(f);
    return MODLE_COOL_ERR_IO;
  }
  *out = f;
  return MODLE_COOL_OK;
}

extern "C" int modle_cool_append_matrix(modle_cool_file* f, size_t chrom_id, uint64_t offset_bp,
                                        const uint32_t* band, uint64_t nrows, uint64_t ncols,
                                        char* err, size_t errlen) {
  if (f == nullptr || (band == nullptr && nrows * ncols != 0) || chrom_id >= f->chrom_sizes.size()) {
    set_err(err, errlen, "modle_cool_append_matrix: invalid argument");
    return MODLE_COOL_ERR_ARG;
  }
  const int64_t chrom_first = f->chrom_offset[chrom_id], chrom_last = f->chrom_offset[chrom_id + 1];
  const int64_t bin_offset = chrom_first + static_cast<int64_t>(offset_bp / f->bin_size);
  // Intervals arrive in genome order; a chromosome may contribute several disjoint intervals
  // (--genomic-intervals, reference: genome.cpp import_genomic_intervals), each with its own
  // offset.  What keeps the pixel table sorted is the order of the bins, not of the chromosomes.
  if (chrom_id + 1 < f->next_chrom || bin_offset < f->next_bin1) {
    set_err(err, errlen, "modle_cool_append_matrix: intervals must be appended in genome order and must not overlap");
    return MODLE_COOL_ERR_ARG;
  }
  if (bin_offset + static_cast<int64_t>(ncols) > chrom_last) {
    set_err(err, errlen, "modle_cool_append_matrix: the matrix does not fit the chromosome's bins");
    return MODLE_COOL_ERR_RANGE;
  }
  std::vector<int64_t> b1, b2;
  std::vector<int32_t> cnt;
  // bins before this interval's first row have no pixels
  for (int64_t b = f->next_bin1; b <= bin_offset; ++b) f->bin1_offset[static_cast<size_t>(b)] = static_cast<int64_t>(f->n_pixels);
  for (uint64_t i = 0; i < ncols; ++i) {
    f->bin1_offset[static_cast<size_t>(bin_offset) + i] = static_cast<int64_t>(f->n_pixels + b1.size());
    for (uint64_t j = i; j < ncols && j - i < nrows; ++j) {
      const uint32_t n = band[j * nrows + (j - i)];
      if (n == 0) continue;
      if (n > static_cast<uint32_t>(std::numeric_limits<int32_t>::max())) {
        set_err(err, errlen, "modle_cool_append_matrix: a count does not fit the int32 pixel type");
        return MODLE_COOL_ERR_RANGE;
      }
      b1.push_back(bin_offset + static_cast<int64_t>(i));
      b2.push_back(bin_offset + static_cast<int64_t>(j));
      cnt.push_back(static_cast<int32_t>(n));
      f->sum += n;
    }
  }
  f->cis = f->sum;  // every pixel joins two bins of one chromosome
  f->next_bin1 = bin_offset + static_cast<int64_t>(ncols);
  f->next_chrom = chrom_id + 1;
  hsize_t s1 = f->n_pixels, s2 = f->n_pixels, s3 = f->n_pixels;
  if (!append(f->d_bin1, H5T_NATIVE_INT64, b1.data(), b1.size(), s1) ||
      !append(f->d_bin2, H5T_NATIVE_INT64, b2.data(), b2.size(), s2) ||
      !append(f->d_count, H5T_NATIVE_INT32, cnt.data(), cnt.size(), s3)) {
    set_err(err, errlen, "HDF5 error while appending pixels");
    return MODLE_COOL_ERR_IO;
  }
  f->n_pixels = s1;
  return MODLE_COOL_OK;
}

extern "C" int modle_cool_close(modle_cool_file* f, char* err, size_t errlen) {
  if (f == nullptr) {
    set_err(err, errlen, "modle_cool_close: invalid argument");
    return MODLE_COOL_ERR_ARG;
  }
  for (size_t b = static_cast<size_t>(f->next_bin1); b < f->bin1_offset.size(); ++b)
    f->bin1_offset[b] = static_cast<int64_t>(f->n_pixels);
  bool ok = write_whole(f->file, "indexes/bin1_offset", H5T_STD_I64LE, H5T_NATIVE_INT64, f->bin1_offset);
  ok = ok && write_whole(f->file, "indexes/chrom_offset", H5T_STD_I64LE, H5T_NATIVE_INT64, f->chrom_offset);
  char date[64];
  {
    const std::time_t t = std::time(nullptr);
    std::tm tm{};
    gmtime_r(&t, &tm);
    std::strftime(date, sizeof(date), "%Y-%m-%dT%H:%M:%S", &tm);
  }
  const uint32_t bin_size = f->bin_size;
  const uint8_t version = 3;
  const int64_t nbins = static_cast<int64_t>(f->bin1_offset.size()) - 1;
  const int32_t nchroms = static_cast<int32_t>(f->chrom_sizes.size());
  const int64_t nnz = static_cast<int64_t>(f->n_pixels);
  ok = ok && write_attr_string(f->file, "assembly", f->assembly);
  ok = ok && write_attr_scalar(f->file, "bin-size", H5T_STD_U32LE, H5T_NATIVE_UINT32, &bin_size);
  ok = ok && write_attr_string(f->file, "bin-type", "fixed");
  ok = ok && write_attr_string(f->file, "creation-date", date);
  ok = ok && write_attr_string(f->file, "format", "HDF5::Cooler");
  ok = ok && write_attr_string(f->file, "format-url", "https://github.com/open2c/cooler");
  ok = ok && write_attr_scalar(f->file, "format-version", H5T_STD_U8LE, H5T_NATIVE_UINT8, &version);
  ok = ok && write_attr_string(f->file, "generated-by", f->generated_by);
  ok = ok && write_attr_string(f->file, "metadata", f->metadata);
  ok = ok && write_attr_scalar(f->file, "nbins", H5T_STD_I64LE, H5T_NATIVE_INT64, &nbins);
  ok = ok && write_attr_scalar(f->file, "nchroms", H5T_STD_I32LE, H5T_NATIVE_INT32, &nchroms);
  ok = ok && write_attr_scalar(f->file, "nnz", H5T_STD_I64LE, H5T_NATIVE_INT64, &nnz);
  ok = ok && write_attr_string(f->file, "storage-mode", "symmetric-upper");
  ok = ok && write_attr_scalar(f->file, "sum", H5T_STD_I64LE, H5T_NATIVE_INT64, &f->sum);
  ok = ok && write_attr_scalar(f->file, "cis", H5T_STD_I64LE, H5T_NATIVE_INT64, &f->cis);
  ok = ok && H5Fflush(f->file, H5F_SCOPE_GLOBAL) >= 0;
  destroy(f);
  if (!ok) {
    set_err(err, errlen, "HDF5 error while finalising the file");
    return MODLE_COOL_ERR_IO;
  }
  return MODLE_COOL_OK;
}

// ---------------------------------------------------------------------------------------------
// Reading a chromosome's cis contacts back into the band layout (for the evaluator,
// modle_amd/evaluate.py: the counterpart of what modle_tools evaluate does with
// hictk::cooler::File::fetch, reference: src/modle_tools/eval.cpp).  Pixels are located through
// the file's own indexes (chrom_offset -> bin1_offset).
// ---------------------------------------------------------------------------------------------
namespace {
template <class T>
bool read_all(hid_t file, const char* name, hid_t memtype, std::vector<T>& out) {
  const hid_t d = H5Dopen2(file, name, H5P_DEFAULT);
  if (d < 0) return false;
  const hid_t sp = H5Dget_space(d);
  const hssize_t n = H5Sget_simple_extent_npoints(sp);
  out.resize(static_cast<size_t>(n > 0 ? n : 0));
  const bool ok = n <= 0 || H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.data()) >= 0;
  H5Sclose(sp);
  H5Dclose(d);
  return ok;
}
template <class T>
bool read_slice(hid_t file, const char* name, hid_t memtype, hsize_t first, hsize_t count, std::vector<T>& out) {
  out.resize(count);
  if (count == 0) return true;
  const hid_t d = H5Dopen2(file, name, H5P_DEFAULT);
  if (d < 0) return false;
  const hid_t fs = H5Dget_space(d);
  H5Sselect_hyperslab(fs, H5S_SELECT_SET, &first, nullptr, &count, nullptr);
  const hid_t ms = H5Screate_simple(1, &count, nullptr);
  const bool ok = H5Dread(d, memtype, ms, fs, H5P_DEFAULT, out.data()) >= 0;
  H5Sclose(ms);
  H5Sclose(fs);
  H5Dclose(d);
  return ok;
}
}  // namespace

extern "C" int modle_cool_read_band(const char* path, const char* chrom, uint64_t nrows,
                                    uint32_t* band, uint64_t band_words, uint64_t* ncols_out,
                                    uint32_t* bin_size_out, uint64_t* missed_out, char* err,
                                    size_t errlen) {
  if (path == nullptr || chrom == nullptr || nrows == 0) {
    set_err(err, errlen, "modle_cool_read_band: invalid argument");
    return MODLE_COOL_ERR_ARG;
  }
  const hid_t file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (file < 0) {
    set_err(err, errlen, std::string("unable to open \"") + path + "\"");
    return MODLE_COOL_ERR_IO;
  }
  int rc = MODLE_COOL_OK;
  do {
    // chromosome names: fixed-length strings
    const hid_t d = H5Dopen2(file, "chroms/name", H5P_DEFAULT);
    if (d < 0) {
      rc = MODLE_COOL_ERR_IO;
      break;
    }
    const hid_t ft = H5Dget_type(d);
    const size_t len = H5Tget_size(ft);
    const hid_t sp = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    const size_t mlen = len + 1;  // room for the terminator of the (null-terminated) memory type
    std::vector<char> names(static_cast<size_t>(n) * mlen + 1, 0);
    const hid_t mt = H5Tcopy(H5T_C_S1);
    H5Tset_size(mt, mlen);
    const bool ok = H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, names.data()) >= 0;
    H5Tclose(mt);
    H5Sclose(sp);
    H5Tclose(ft);
    H5Dclose(d);
    if (!ok) {
      rc = MODLE_COOL_ERR_IO;
      break;
    }
    int64_t cid = -1;
    for (hssize_t i = 0; i < n; ++i) {
      const std::string nm(names.data() + static_cast<size_t>(i) * mlen, strnlen(names.data() + static_cast<size_t>(i) * mlen, mlen));
      if (nm == chrom) cid = i;
    }
    if (cid < 0) {
      set_err(err, errlen, std::string("chromosome \"") + chrom + "\" is not in the file");
      rc = MODLE_COOL_ERR_ARG;
      break;
    }
    std::vector<int64_t> chrom_offset, bin1_offset;
    if (!read_all(file, "indexes/chrom_offset", H5T_NATIVE_INT64, chrom_offset) ||
        !read_all(file, "indexes/bin1_offset", H5T_NATIVE_INT64, bin1_offset)) {
      rc = MODLE_COOL_ERR_IO;
      break;
    }
    const int64_t b0 = chrom_offset[static_cast<size_t>(cid)], b1 = chrom_offset[static_cast<size_t>(cid) + 1];
    const uint64_t ncols = static_cast<uint64_t>(b1 - b0);
    if (ncols_out) *ncols_out = ncols;
    if (bin_size_out) {
      const hid_t a = H5Aopen(file, "bin-size", H5P_DEFAULT);
      uint32_t bs = 0;
      if (a >= 0) {
        H5Aread(a, H5T_NATIVE_UINT32, &bs);
        H5Aclose(a);
      }
      *bin_size_out = bs;
    }
    if (band == nullptr) break;  // shape query
    const uint64_t rows = std::min<uint64_t>(nrows, ncols);
    if (band_words < rows * ncols) {
      set_err(err, errlen, "modle_cool_read_band: band buffer too small");
      rc = MODLE_COOL_ERR_ARG;
      break;
    }
    std::fill(band, band + rows * ncols, 0u);
    const hsize_t p0 = static_cast<hsize_t>(bin1_offset[static_cast<size_t>(b0)]);
    const hsize_t p1 = static_cast<hsize_t>(bin1_offset[static_cast<size_t>(b1)]);
    std::vector<int64_t> x1, x2;
    std::vector<int32_t> cn;
    if (!read_slice(file, "pixels/bin1_id", H5T_NATIVE_INT64, p0, p1 - p0, x1) ||
        !read_slice(file, "pixels/bin2_id", H5T_NATIVE_INT64, p0, p1 - p0, x2) ||
        !read_slice(file, "pixels/count", H5T_NATIVE_INT32, p0, p1 - p0, cn)) {
      rc = MODLE_COOL_ERR_IO;
      break;
    }
    uint64_t missed = 0;
    for (size_t k = 0; k < cn.size(); ++k) {
      if (x2[k] >= b1) continue;  // trans pixel
      const uint64_t i = static_cast<uint64_t>(x2[k] - x1[k]), j = static_cast<uint64_t>(x2[k] - b0);
      if (i >= rows) {
        missed += static_cast<uint64_t>(cn[k]);
        continue;
      }
      band[j * rows + i] = static_cast<uint32_t>(cn[k]);
    }
    if (missed_out) *missed_out = missed;
  } while (false);
  H5Fclose(file);
  if (rc == MODLE_COOL_ERR_IO) set_err(err, errlen, std::string("HDF5 error while reading \"") + path + "\"");
  return rc;
}
