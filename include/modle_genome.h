/* modle_genome.h -- C ABI of the genome import that sits in front of the simulation path:
 * chrom.sizes + extrusion-barrier BED6 (+ optional BED3 of genomic intervals) -> intervals with
 * their barriers, by the reference's rules.
 *
 * Replaces, for callers that do not link the reference's own host code:
 *   Genome::Genome / import_chromosomes / import_genomic_intervals / map_barriers_to_intervals /
 *   generate_barriers_from_bed_records   (reference: src/libmodle/internal/genome.cpp:299-469)
 *   chrom_sizes::Parser::parse_all        (reference: src/libmodle_io/chrom_sizes.cpp)
 *   bed::Parser / bed::BED                (reference: src/libmodle_io/bed.cpp:245-600)
 * The inputs are the TEXT of the files (the caller reads / decompresses them); errors come back
 * as a negative code plus a message naming the offending line, nothing is thrown.
 * Host only (lives in libmodle_hip.so next to the other host logic; no GPU needed).
 */
#ifndef MODLE_GENOME_H
#define MODLE_GENOME_H

#include <stddef.h>
#include <stdint.h>

#include "modle_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MODLE_GENOME_OK 0
#define MODLE_GENOME_ERR_ARG (-1)
#define MODLE_GENOME_ERR_PARSE (-2) /* malformed / duplicate record, invalid score or strand, ... */

typedef struct modle_genome modle_genome;

typedef struct modle_genome_interval {
  uint64_t id;       /* GenomicInterval::id(): position in genome order */
  uint64_t chrom_id; /* index into the chromosome table (chrom.sizes order) */
  uint64_t start;    /* [start, end) in bp */
  uint64_t end;
  uint64_t num_barriers;
} modle_genome_interval;

/* Parses the three texts.  `intervals_bed` may be NULL / empty: every chromosome is then one
 * interval (reference: genome.cpp:355-366).  `cfg` supplies the default barrier self-transition
 * probabilities (barrier_occupied_stp / barrier_not_occupied_stp, after
 * modle_hip_config_transform).  `name_is_not_bound_stp`: --interpret-extrusion-barrier-name-as-
 * not-bound-stp (the name must then parse as a probability; like the reference, the value is
 * validated and then NOT used: compute_barrier_stp receives the defaults, genome.cpp:438-459). */
int modle_genome_import(const char* chrom_sizes, size_t chrom_sizes_len, const char* barriers_bed,
                        size_t barriers_bed_len, const char* intervals_bed,
                        size_t intervals_bed_len, const modle_hip_config* cfg,
                        int name_is_not_bound_stp, modle_genome** out, char* err, size_t errlen);
void modle_genome_free(modle_genome* g);

size_t modle_genome_num_chromosomes(const modle_genome* g);
/* `name` points into the handle (valid until modle_genome_free) */
int modle_genome_chromosome(const modle_genome* g, size_t i, const char** name, uint64_t* size);
size_t modle_genome_num_intervals(const modle_genome* g);
int modle_genome_interval_info(const modle_genome* g, size_t i, modle_genome_interval* out);
/* Barriers of interval i in import order (NOT sorted: modle_hip_add_interval sorts, like
 * State::operator= does per task): position = (chromStart + chromEnd + 1) / 2, blocking direction
 * MODLE_HIP_DIR_REV for strand '+' and MODLE_HIP_DIR_FWD for '-' ('.' records are dropped),
 * stp_active from the BED score (= occupancy; 0 => the default), stp_inactive the default.
 * Every array holds num_barriers entries. */
int modle_genome_interval_barriers(const modle_genome* g, size_t i, uint64_t* pos, uint8_t* dir,
                                   double* stp_active, double* stp_inactive);
/* records of the barrier file that were imported / dropped because of strand '.' */
void modle_genome_barrier_counts(const modle_genome* g, uint64_t* imported, uint64_t* dropped);

#ifdef __cplusplus
}
#endif
#endif
