/* modle_cooler.h -- C ABI of the cooler (v3) writer for the simulated contact matrices.
 *
 * the data format on the output side of the hot path.  Replaces,
 * for a host that links this library instead of hictk,
 *   io::init_cooler_file<int32_t>(...)            reference: src/libmodle_io/contact_matrix_dense_io_impl.hpp:75-150
 *                                                 (called at src/libmodle/cpu/simulation.cpp:117-127)
 *   io::append_contact_matrix_to_cooler(...)      reference: contact_matrix_dense_io_impl.hpp:51-71, 152-166
 *                                                 (called at simulation.cpp:143-168)
 * and what hictk::cooler::File (hictk 2.1.4, vendored as external/hictk-v2.1.4.tar.xz; schema
 * in its cooler/impl/file_write_impl.hpp:246-330 and cooler/cooler.hpp:50-73) does when the
 * file is closed: the bin1 / chromosome offset indexes and the standard attributes.
 *
 * The file holds every chromosome passed to modle_cool_create (also those without contacts),
 * int32 counts, pixels sorted by (bin1_id, bin2_id), symmetric-upper storage.  Host side only: no
 * GPU code; links libhdf5.
 */
#ifndef MODLE_COOLER_H
#define MODLE_COOLER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MODLE_COOL_OK 0
#define MODLE_COOL_ERR_ARG (-1)    /* invalid argument (null pointer, chromosome out of order, ...) */
#define MODLE_COOL_ERR_IO (-2)     /* HDF5 / file system failure */
#define MODLE_COOL_ERR_RANGE (-3)  /* a count does not fit the int32 pixel type, a bin id is out of range */

typedef struct modle_cool_file modle_cool_file;

/* Creates the file with its chromosome and bin tables (init_cooler_file).  `metadata_json` may be
 * NULL or empty (the attribute then holds "{}", hictk's default).  Fails if the file exists and
 * force_overwrite is 0. */
int modle_cool_create(const char* path, int force_overwrite, const char* const* chrom_names,
                      const uint32_t* chrom_sizes, size_t n_chroms, uint32_t bin_size,
                      const char* assembly, const char* generated_by, const char* metadata_json,
                      modle_cool_file** out, char* err, size_t errlen);

/* Appends the non-zero pixels of one interval's band matrix (layout of
 * modle_hip_interval_outputs: cell (row, col), row <= col, col - row < nrows, at
 * band[col * nrows + (col - row)]), visited row by row like the reference.  `offset_bp` is the
 * interval's start within the chromosome.  Intervals must be appended in genome order (ascending
 * chrom_id, ascending non-overlapping bin ranges within a chromosome): a chromosome may
 * contribute several disjoint intervals, as with the reference's --genomic-intervals, and the
 * pixels stay sorted like the reference's genome-order writes. */
int modle_cool_append_matrix(modle_cool_file* f, size_t chrom_id, uint64_t offset_bp,
                             const uint32_t* band, uint64_t nrows, uint64_t ncols, char* err,
                             size_t errlen);

/* Writes the indexes and the attributes (nnz, sum, cis, ...) and closes the file.  The handle is
 * freed also when an error is returned. */
int modle_cool_close(modle_cool_file* f, char* err, size_t errlen);

/* Reads the cis contacts of one chromosome back into the band layout (for the evaluator; the
 * counterpart of hictk's File::fetch in the reference's modle_tools evaluate).  `nrows`: band
 * width in bins; the band holds min(nrows, ncols) * ncols words.  Pass band = NULL to query the
 * shape (*ncols_out, *bin_size_out) only.  Contacts beyond the band are summed in *missed_out. */
int modle_cool_read_band(const char* path, const char* chrom, uint64_t nrows, uint32_t* band,
                         uint64_t band_words, uint64_t* ncols_out, uint32_t* bin_size_out,
                         uint64_t* missed_out, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
