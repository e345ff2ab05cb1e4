/* modle_bigwig.h -- C ABI of the bigWig writer for the 1-D LEF occupancy track, the second file
 * the reference's IO thread writes next to the cooler (reference:
 * src/libmodle/cpu/simulation.cpp:130-141 init_bigwig_writer, :170-197
 * write_lef_occupancy_to_bwig; src/libmodle_io/bigwig_impl.hpp:127-158 write_range ->
 * libBigWig's bwAddIntervalSpanSteps).  Host only, lives in libmodle_cooler.so.
 *
 * File layout: bigWig version 4, fixedStep sections (zlib-compressed), chromosome B+ tree with
 * the chromosomes in the order given (ids = genome order, like libBigWig), R-tree index over the
 * sections, total summary, no zoom levels.
 */
#ifndef MODLE_BIGWIG_H
#define MODLE_BIGWIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODLE_BW_OK 0
#define MODLE_BW_ERR_ARG (-1)
#define MODLE_BW_ERR_IO (-2)

typedef struct modle_bw_file modle_bw_file;

/* init_bigwig_writer + write_chromosomes: every chromosome of the genome, in genome order */
int modle_bw_create(const char* path, int force_overwrite, const char* const* chrom_names,
                    const uint32_t* chrom_sizes, size_t n_chroms, modle_bw_file** out, char* err,
                    size_t errlen);
/* write_range(chrom, values, span, step, offset): value i covers
 * [offset + i * step, offset + i * step + span).  Ranges must be appended in genome order. */
int modle_bw_write_range(modle_bw_file* f, size_t chrom_id, const float* values, size_t n_values,
                         uint32_t span, uint32_t step, uint32_t offset, char* err, size_t errlen);
/* write_lef_occupancy_to_bwig: the occupancy counts of one interval divided by their maximum
 * (float32 of the double quotient), span = step = bin size, offset = interval start */
int modle_bw_write_occupancy(modle_bw_file* f, size_t chrom_id, const uint64_t* occupancy,
                             size_t n_bins, uint32_t bin_size, uint32_t offset_bp, char* err,
                             size_t errlen);
/* writes the index and the summary and closes the file; the handle is freed in every case */
int modle_bw_close(modle_bw_file* f, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
