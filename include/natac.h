/* natac.h -- C-ABI of libnatac_hip.so: NucleoATAC's per-chunk occ + nuc signal path on MI355X (gfx950).
 *
 * The reference (GreenleafLab/NucleoATAC v0.3.4, Python 2.7 + Cython) has no FFI surface; its de-facto
 * operator boundary is
 *   (1) the per-chunk map functions handed to multiprocessing.Pool.map:
 *         _occHelper  nucleoatac/run_occ.py:23-39   -> natac_run_occ  (+ natac_batch_download)
 *         _nucHelper  nucleoatac/run_nuc.py:22-39   -> natac_run_nuc, natac_run_candidates
 *   (2) the two Cython extension modules:
 *         pyatac/fragments.pyx:17   makeFragmentMat               -> natac_make_fragment_mat
 *         pyatac/fragments.pyx:43   getInsertions                 -> natac_get_insertions / natac_run_ins
 *         pyatac/fragments.pyx:71   getStrandedInsertions         -> natac_get_stranded_insertions
 *         pyatac/fragments.pyx:123  getFragmentSizesFromChunkList -> natac_fragment_sizes
 *         nucleoatac/multinomial_cov.pyx:20 calculateCov          -> natac_calculate_cov
 * Every entry point below names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no exceptions, no C++/torch types.  Return value: 0 = ok, <0 = error class
 *     (NATAC_E_*); natac_last_error() gives the message (thread-local).
 *   - a context (natac_ctx) owns one HIP device + one stream; one host thread per context.
 *   - host pointers are caller-owned; the library copies.  Device outputs live in the batch until it is freed.
 *   - all float tracks are float64 like the reference (pyatac/fragments.pyx:9, chunkmat2d.py:20); insertion
 *     counts are int32.
 *   - fragments are (l, n) = (pos+4, |tlen|-8) of forward proper-pair reads (pyatac/fragments.pyx:25-31),
 *     packed per chunk as in nucleoatac_amd/packing.py and SORTED BY CENTRE l + (n-1)//2 inside a chunk.
 */
#ifndef NATAC_H
#define NATAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an entry point is added, removed or changes meaning (2: round 5 removed natac_run_nuc_occ, added natac_bg_tiling /
 * natac_store_set_budget / natac_store_declined, gave natac_store_adopt's n_hard == -1 a meaning; 3: round 6 added natac_batch_format_fetch_begin / _wait);
 * the binding refuses another version */
#define NATAC_ABI_VERSION 3

enum {
    NATAC_OK = 0,
    NATAC_E_ARG = -1,   /* bad argument / inconsistent sizes */
    NATAC_E_HIP = -2,   /* HIP runtime error (no device, launch failure, ...) */
    NATAC_E_STATE = -3, /* call order: constants not set, stage not run yet */
    NATAC_E_NOMEM = -4
};

/* per-base tracks of a batch (concatenated over chunks in chunk order; chunk i occupies
 * [out_off[i], out_off[i+1]) ).  float64 unless noted. */
enum {
    NATAC_T_NUC_COV = 0,    /* CoverageTrack rows [vlower,vupper)   NucleosomeCalling.py:257-260 */
    NATAC_T_NFR_COV = 1,    /* CoverageTrack rows [0,vlower)        NucleosomeCalling.py:271-273 */
    NATAC_T_RAW = 2,        /* SignalTrack.calculateSignal           NucleosomeCalling.py:29-36  (nucleoatac_raw) */
    NATAC_T_BACKGROUND = 3, /* BiasTrack.calculateBackgroundSignal   NucleosomeCalling.py:49-64  (nucleoatac_background; an output only with
                             * --write_all: after the FFT kernel it is formed on the first download / track_ptr / writer request) */
    NATAC_T_NORM = 4,       /* NormSignalTrack                       NucleosomeCalling.py:38-43  (nucleoatac_signal) */
    NATAC_T_SMOOTH = 5,     /* NucChunk.smoothSignal                 NucleosomeCalling.py:274-283 (nucleoatac_signal.smooth) */
    NATAC_T_OCC = 6,        /* OccupancyTrack.smoothed_vals AFTER call_peaks' in-place NaN fill (Occupancy.py:147-153, utils.py:86-91) */
    NATAC_T_OCC_LOWER = 7,  /* smoothed_lower */
    NATAC_T_OCC_UPPER = 8,  /* smoothed_upper */
    NATAC_T_OCC_COV = 9,    /* OccChunk.getCov                       Occupancy.py:221-224 */
    NATAC_T_INS = 10,       /* int32 insertion counts, getInsertions pyatac/fragments.pyx:43-67 */
    NATAC_T_OCC_PREFILL = 11, /* smoothed_vals BEFORE the NaN fill (formed on the first download / track_ptr request) */
    NATAC_T_COUNT = 12
};

/* per-grid-point arrays of the occupancy MLE (one value per `step` bases; chunk i occupies
 * [grid_off[i], grid_off[i+1]) with grid point k at base halfstep + k*step):
 * OccupancyTrack.vals / lower_bound / upper_bound before smoothing (Occupancy.py:128-146). */
enum { NATAC_G_OCC = 0, NATAC_G_LOWER = 1, NATAC_G_UPPER = 2 };

/* kernels, for natac_profile_get */
enum {
    NATAC_K_FRAG_GATHER = 0, /* nuc_cov / nfr_cov / raw (sparse V-plot gather) */
    NATAC_K_BACKGROUND = 1,  /* dense bias x VMat correlation (dominant) */
    NATAC_K_SMOOTH_NUC = 2,
    NATAC_K_OCC_MLE = 3,
    NATAC_K_OCC_SMOOTH = 4,
    NATAC_K_OCC_FILL = 5,
    NATAC_K_INS = 6,
    NATAC_K_CAND = 7,
    NATAC_K_SIZE_HIST = 8,   /* insert-size histogram of natac_fragment_sizes */
    NATAC_K_COUNT = 9
};

typedef struct natac_ctx natac_ctx;
typedef struct natac_batch natac_batch;

/* ---- library / context ---------------------------------------------------------------- */
int natac_abi_version(void);
const char *natac_last_error(void);
int natac_device_count(int *count);
int natac_ctx_create(int device_id, natac_ctx **out);
void natac_ctx_destroy(natac_ctx *ctx);
int natac_ctx_sync(natac_ctx *ctx);
/* device name, CU count, global memory bytes of the context's device */
int natac_ctx_device_info(natac_ctx *ctx, char *name, size_t name_len, int *n_cu, size_t *mem_bytes);
/* which physical GPU the context computes on: the HIP device ordinal and its PCI bus id ("0000:05:00.0"); the multi-rank drivers
 * print both per rank and check that ranks meant to own a GPU each really do */
int natac_ctx_device_ids(natac_ctx *ctx, int *hip_device, char *pci_bus_id, size_t pci_len);

/* ---- run-level constants --------------------------------------------------------------- */
/* VMat template (pyatac/VMat.py:24-37): mat[(upper-lower) x (2w+1)] row-major, insert sizes [lower,upper). */
int natac_set_vmat(natac_ctx *ctx, const double *mat, int lower, int upper, int w);
/* global insert-size distribution over [0, upper) used by BiasMat2D.normByInsertDist (chunkmat2d.py:154-156). */
int natac_set_sizes(natac_ctx *ctx, const double *sizes, int upper);
/* OccupancyCalcParams + OccupancyParameters (Occupancy.py:89-102, 175-193): nuc_probs / nfr_probs over
 * [0, upper) (already normalised), the alpha grid (np.linspace(0,1,101)), the chi2 cutoff, step (odd), flank. */
int natac_set_occ_model(natac_ctx *ctx, const double *nuc_probs, const double *nfr_probs, int upper,
                        const double *alphas, int n_alpha, double cutoff, int step, int flank);

/* ---- batches of packed chunks ---------------------------------------------------------- */
/* Upload one batch (layout: nucleoatac_amd/packing.py).  bias_off/bias_log may be NULL (no FASTA: bias
 * matrix of ones, NucleosomeCalling.py:243-250).  bias_log covers [start-bias_left, end+bias_right). */
int natac_batch_create(natac_ctx *ctx, int32_t n_chunks, const int32_t *chunk_len, const int64_t *frag_off,
                       const int32_t *frag_lpos, const int32_t *frag_ilen, const int64_t *bias_off,
                       const double *bias_log, int32_t bias_left, int32_t bias_right, natac_batch **out);
/* The same batch with the Tn5 bias scored ON THE DEVICE from the genome sequence (InsertionBiasTrack.computeBias, pyatac/bias.py:85-92, as
 * the reference calls it per chunk, Occupancy.py:212-214 / NucleosomeCalling.py:246-248): seq[seq_off[i] .. seq_off[i+1]) = the upper-case
 * bases of [start - bias_left - up, end + bias_right + down) of chunk i (up + down + 1 = K, the PWM width); log_pwm[nrow x K] = log(PWM.mat),
 * nucleotides[nrow] its row letters (natac_pwm_bias).  One byte per base crosses PCIe instead of a float64 score down and up again. */
int natac_batch_create_from_seq(natac_ctx *ctx, int32_t n_chunks, const int32_t *chunk_len, const int64_t *frag_off,
                                const int32_t *frag_lpos, const int32_t *frag_ilen, const int64_t *seq_off, const uint8_t *seq,
                                const double *log_pwm, const uint8_t *nucleotides, int nrow, int K, int32_t bias_left,
                                int32_t bias_right, natac_batch **out);
void natac_batch_free(natac_batch *b);
int natac_batch_info(natac_batch *b, int64_t *total_bp, int64_t *total_grid, int64_t *n_frags);
/* How the background stage (NucleosomeCalling.py:49-64, the dense correlation of the bias matrix with the V-plot) tiles a chunk of
 * chunk_len bases with the V-plot that is set: n_tiles 512-point transforms per row pair, of which the first `extended` yield 16 more
 * outputs on each side (finished by an edge pass; chosen where that saves whole tiles).  0 tiles = the FFT path does not apply to
 * this V-plot (direct summation).  A function of the chunk's length and the model alone. */
int natac_bg_tiling(natac_ctx *ctx, int32_t chunk_len, int32_t *n_tiles, int32_t *extended);
/* Drop every output of the batch (per-base tracks, grid arrays, candidate / peak arrays) but keep its packed inputs resident:
 * for workloads whose outputs do not fit in HBM all at once (BASELINE configs[3] on one GPU: 3 Gbp x ~160 B/bp); the blocks go
 * back to the library's pool and are reused by the next batch's stages. */
int natac_batch_release_outputs(natac_batch *b);

/* NucChunk.process up to smoothSignal (NucleosomeCalling.py:328-334): fills NUC_COV, NFR_COV, RAW,
 * BACKGROUND, NORM, SMOOTH.  smooth_sd = NucParameters.smooth_sd (cli default 10).  Asynchronous. */
int natac_run_nuc(natac_batch *b, double smooth_sd);
/* OccChunk.process up to getCov + the call_peaks NaN fill (Occupancy.py:241-247): fills the grid arrays,
 * OCC, OCC_LOWER, OCC_UPPER, OCC_COV (OCC_PREFILL, the smoothed occupancy before the fill, on request).  Asynchronous. */
int natac_run_occ(natac_batch *b);
/* InsertionTrack.calculateInsertions (pyatac/tracks.py:164-168) for every chunk: fills INS. */
int natac_run_ins(natac_batch *b, int lower, int upper);
/* Per-candidate statistics (needs natac_run_nuc first).  cand_chunk[k] = chunk index, cand_pos[k] = position
 * relative to the chunk start.  Outputs (host, length n_cand):
 *   lr   Nucleosome.getLR       NucleosomeCalling.py:110-122
 *   var  calculateCov closed form r*(sum p v^2 - (sum p v)^2), r = int(nuc_cov[pos])  multinomial_cov.pyx:20-31
 *   z    Nucleosome.getZScore   NucleosomeCalling.py:123-127   (norm_signal / sqrt(var)) */
int natac_run_candidates(natac_batch *b, int64_t n_cand, const int32_t *cand_chunk, const int32_t *cand_pos,
                         double *lr, double *var, double *z);
/* calculateCov (nucleoatac/multinomial_cov.pyx:20-31) at MANY candidates of the batch at once, in one of three arithmetic variants
 * (BASELINE configs[4]: "fp64 multinomial_cov path, tolerance sweep"; needs natac_run_nuc first).  Candidate k's probability
 * vector is SignalDistribution's p = B window / sum (NucleosomeCalling.py:70-76), v = the V-plot, r = int(nuc_cov[pos]) (:125).
 * mode 0 = closed form r (sum p v^2 - (sum p v)^2) in fp64 -- what natac_run_candidates reports; mode 1 = the .pyx's literal
 * O(N^2) pair sum in fp64; mode 2 = the closed form evaluated in fp32.  var[n_cand] on the host. */
int natac_run_candidates_cov(natac_batch *b, int64_t n_cand, const int32_t *cand_chunk, const int32_t *cand_pos, int mode,
                             double *var);
/* Candidate search + statistics entirely on the device (needs natac_run_nuc first):
 * utils.call_peaks(norm + smoothed, min_signal, sep, boundary, order) exactly as NucChunk.findAllNucs calls it
 * (nucleoatac/NucleosomeCalling.py:297-301, pyatac/utils.py:56-102), followed by LR / variance / z for every candidate.
 * jitter[n_jitter] is the reference's tie-break stream np.random.RandomState(25).uniform(0, 1e-12, n) generated by the
 * host (n_jitter >= longest chunk).  *n_cand receives the number of candidates (chunk order, ascending position);
 * fetch them with natac_download_peaks.  Batches whose longest chunk has at most 16,384 bases: a
 * chunk with more than 2,048 local maxima sets status bit 1 (list truncated; the drivers redo those chunks with utils.call_peaks on the
 * host).  Longer chunks (merged windows of tens of kb to Mb) keep their lists in device memory: no limit. */
int natac_run_peaks(natac_batch *b, double min_signal, int sep, int boundary, int order, const double *jitter,
                    int64_t n_jitter, int64_t *n_cand);
int natac_download_peaks(natac_batch *b, int64_t n_cand, int32_t *cand_chunk, int32_t *cand_pos, double *lr, double *var,
                         double *z);
/* The same peak search on ONE per-base track (no candidate statistics): utils.call_peaks(track, min_signal, sep, boundary,
 * order), e.g. OccChunk.callPeaks on NATAC_T_OCC (nucleoatac/Occupancy.py:225-231: sep = nuc_sep, min_signal = min_occ,
 * boundary = sep/2, order = 1).  Fetch positions with natac_download_peaks (lr / var / z may be NULL). */
int natac_run_track_peaks(natac_batch *b, int track, double min_signal, int sep, int boundary, int order,
                          const double *jitter, int64_t n_jitter, int64_t *n_peaks);
/* OccChunk.callPeaks + OccChunk.getNucDist for every chunk on the device (nucleoatac/Occupancy.py:225-240; needs natac_run_occ):
 * call_peaks(smoothed_vals, sep, min_signal = min_occ) as natac_run_track_peaks(NATAC_T_OCC, ...) does it, then per peak the
 * OccPeak values (occ, lower, upper, reads = cov) and keep = (lower > min_occ && reads > 0), and per chunk
 * nuc_dist[upper] = sum over its kept peaks of the window's insert-size histogram / its total, in peak order.  `jitter` as in
 * natac_run_peaks.  *n_peaks = number of call_peaks peaks (kept or not). */
int natac_run_occ_peaks(natac_batch *b, double min_occ, int sep, const double *jitter, int64_t n_jitter, int64_t *n_peaks);
int natac_download_occ_peaks(natac_batch *b, int64_t n_peaks, int32_t *chunk, int32_t *pos, double *occ, double *lower,
                             double *upper, double *reads, int32_t *keep);
/* float64[n_chunks x upper] (upper of natac_set_occ_model), row i = OccChunk.getNucDist() of chunk i */
int natac_download_nuc_dist(natac_batch *b, double *dst, size_t dst_bytes);
/* copy one per-base track to host (float64[total_bp], or int32[total_bp] for NATAC_T_INS). Synchronises. */
int natac_batch_download(natac_batch *b, int track, void *dst, size_t dst_bytes);
/* copy one per-grid-point array to host (float64[total_grid]). */
int natac_batch_download_grid(natac_batch *b, int which, double *dst, size_t dst_bytes);
/* overwrite one float64 per-base track of the batch with host values (total_bp doubles, chunk order): lets any track -- an
 * externally computed one, or a test pattern -- go through the device-side writer (natac_batch_format_track) */
int natac_batch_set_track(natac_batch *b, int track, const double *vals, size_t n);
/* per-chunk status flags (int32[n_chunks]; 0 = ok, bit0 = occupancy likelihood undefined at some grid point
 * -- the reference would raise ValueError at Occupancy.py:118). */
int natac_batch_status(natac_batch *b, int32_t *dst, size_t dst_bytes);
/* raw device pointer of a per-base track (for zero-copy consumers); NULL if the stage has not run. */
int natac_batch_track_ptr(natac_batch *b, int track, void **dptr);

/* ---- drop-in replacements of the Cython functions (host buffers in / out, synchronous) --- */
/* makeFragmentMat, pyatac/fragments.pyx:17-40: mat[(upper-lower) x (end-start)] float64, zeroed + filled. */
int natac_make_fragment_mat(natac_ctx *ctx, int64_t n_frags, const int64_t *l, const int32_t *n, int64_t start,
                            int64_t end, int lower, int upper, double *mat);
/* getInsertions, pyatac/fragments.pyx:43-67: out[end-start] float64. */
int natac_get_insertions(natac_ctx *ctx, int64_t n_frags, const int64_t *l, const int32_t *n, int64_t start,
                         int64_t end, int lower, int upper, double *out);
/* getStrandedInsertions, pyatac/fragments.pyx:71-97: plus[end-start] = insertions at left fragment ends, minus[end-start] = at
 * right fragment ends (plus + minus == natac_get_insertions). */
int natac_get_stranded_insertions(natac_ctx *ctx, int64_t n_frags, const int64_t *l, const int32_t *n, int64_t start,
                                  int64_t end, int lower, int upper, double *plus, double *minus);
/* getFragmentSizesFromChunkList, pyatac/fragments.pyx:123-145 (one chromosome's fragments, its chunks):
 * sizes[upper-lower] float64 counts (not normalised). */
int natac_fragment_sizes(natac_ctx *ctx, int64_t n_frags, const int64_t *l, const int32_t *n, int32_t n_chunks,
                         const int64_t *chunk_start, const int64_t *chunk_end, int lower, int upper, double *sizes);
/* calculateCov, nucleoatac/multinomial_cov.pyx:20-31.  mode 0 = closed form O(N), mode 1 = literal O(N^2)
 * pair sum (same terms as the .pyx, tree-reduced).  r is truncated to int like the .pyx signature. */
int natac_calculate_cov(natac_ctx *ctx, const double *p, const double *v, int64_t n, int r, int mode, double *out);

/* ---- operator-level entry points behind the Track / BiasMat2D / bias / occupancy classes ------ */
/* utils.smooth, pyatac/utils.py:23-52, on one host array.  w[M] is the window (ones for 'flat', scipy's
 * gaussian(M, sd) otherwise; M odd).  mode 0 = 'valid' (out[n-M+1]), 1 = 'same' (out[n], needs n >= M).
 * norm != 0: NaN-aware normalisation (0 weight -> NaN). */
int natac_smooth(natac_ctx *ctx, const double *x, int64_t n, const double *w, int M, int mode, int norm, double *out);
/* BiasMat2D.makeBiasMat, pyatac/chunkmat2d.py:140-153 (log-scale track): mat[(upper-lower) x (end-start)].
 * bias_log[nb] starts at genomic coordinate track_start; returns NATAC_E_ARG if the track does not cover
 * [start - upper//2, end + upper//2). */
int natac_make_bias_mat(natac_ctx *ctx, const double *bias_log, int64_t nb, int64_t track_start, int64_t start,
                        int64_t end, int lower, int upper, double *mat);
/* InsertionBiasTrack.computeBias, pyatac/bias.py:85-92: seq[n] (upper-case ASCII) covers [start-up, end+down);
 * log_pwm[nrow x K] = log(PWM.mat), nucleotides[nrow] the PWM's row letters; out[n-K+1]. */
int natac_pwm_bias(natac_ctx *ctx, const uint8_t *seq, int64_t n, const double *log_pwm, const uint8_t *nucleotides,
                   int nrow, int K, double *out);
/* signal.correlate(sub, vmat, mode='valid')[0] as used by SignalTrack.calculateSignal / BiasTrack
 * (nucleoatac/NucleosomeCalling.py:34-36, 60-63): sub[R x ncol], vmat[R x W] row-major, out[ncol-W+1]. */
int natac_correlate_valid(natac_ctx *ctx, const double *sub, int64_t ncol, const double *vmat, int R, int W, double *out);
/* calculateOccupancy, nucleoatac/Occupancy.py:104-120, for one window: inserts[upper], bias[upper] with the model
 * set by natac_set_occ_model; out[3] = {occ, lower, upper}.  Returns NATAC_E_ARG with the message
 * "no alpha passes the likelihood-ratio test" where the reference raises ValueError (Occupancy.py:118). */
int natac_calculate_occupancy(natac_ctx *ctx, const double *inserts, const double *bias, double *out);

/* ---- native track writer (host side, multi-threaded) ------------- */
/* Track.write_track, pyatac/tracks.py:37-74, for n_chunks tracks at once (chunk i: chroms[i], chunk_start[i], values
 * vals[out_off[i] .. out_off[i+1]) ), run-length bedGraph text with python-2 float formatting, NaN runs skipped, and -- as in the
 * reference, whose loop overwrites prev_value with the NaN before flushing (tracks.py:56-66) -- a run of values directly followed
 * by a NaN is NOT written.  write_zero: bit 0 = write runs of 0 (the reference's write_zero), bit 1 = also keep runs that precede a
 * NaN (deviation from the reference, off by default).
 * compress: 0 = plain text, 1..9 = BGZF at that deflate level (bgzip-compatible; run_occ.py:130-136).  append != 0
 * appends to `path`; finish != 0 terminates a BGZF file with the EOF marker block.  n_threads <= 0: automatic.
 * No GPU involved; usable on the arrays natac_batch_download returns. */
int natac_write_bedgraph(const char *path, int append, int compress, int finish, int32_t n_chunks, const char *const *chroms,
                         const int64_t *chunk_start, const int64_t *out_off, const double *vals, int write_zero,
                         int n_threads, int64_t *bytes_written);

/* ---- device-side track writer (SURVEY.md section 8f row 1 moved onto the GPU) ------------------------------------------------- */
/* Track.write_track, pyatac/tracks.py:37-74, for one per-base track of a batch ON THE DEVICE: run-length detection (including the
 * reference's rule that a run directly followed by a NaN is not written, tracks.py:56-66), python-2 `str(float)` ('%.12g' with exact
 * 192-bit integer arithmetic) and the line layout `chrom \t start \t end \t value \n`, chunk i on names[chrom_id[i]] at chunk_start[i].
 * compress = 0: the text itself; compress != 0: BGZF members of <= 0xff00 text bytes each (no EOF marker; the bgzip step of
 * run_occ.py:130-136) from a line-structured LZ77 + a Huffman code built for this text (csrc/natac_deflate.hpp).  write_zero as
 * in natac_write_bedgraph.  The result stays on the device until the next call; fetch it with natac_batch_format_fetch.
 * n_bytes = size of the result, n_text_bytes = size of the text, n_lines = lines written, n_hard = values whose 12th digit could
 * not be decided from the truncated power-of-ten table (only |v| >= 1e12 or < 1e-44 can; the caller then formats this track
 * with natac_write_bedgraph instead).  Byte-identical to natac_write_bedgraph's text when n_hard == 0. */
int natac_batch_format_track(natac_batch *b, int track, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                             const int64_t *chunk_start, int write_zero, int compress, int64_t *n_bytes, int64_t *n_text_bytes,
                             int64_t *n_lines, int32_t *n_hard);
int natac_batch_format_fetch(natac_batch *b, void *dst, size_t dst_bytes);
/* The same without waiting: _begin starts the copy of the last natac_batch_format_track result into `dst` (pinned host memory) on a
 * second stream and returns; the batch may format its next track at once (the reference's writer processes likewise take track after
 * track, run_occ.py:130-136).  `dst` holds the result after natac_batch_format_fetch_wait, which waits for every copy begun on the
 * batch; natac_batch_free and natac_batch_release_outputs wait too.  The tabix records of a result (natac_batch_format_index_*) must be
 * fetched before the next natac_batch_format_track, as with the blocking fetch. */
int natac_batch_format_fetch_begin(natac_batch *b, void *dst, size_t dst_bytes);
int natac_batch_format_fetch_wait(natac_batch *b);
/* The .tbi of a file assembled from natac_batch_format_track(compress) results, WITHOUT re-reading the file (the reference runs
 * pysam.tabix_index(..., preset="bed") over every finished file, run_occ.py:136-139).  The device reduces the lines of a result to runs
 * of records per 16-kb leaf bin (one per ~16 kb instead of one per base): natac_batch_format_index_size / _fetch return the runs
 * of the LAST result -- chromosome id, [beg, end), record count, text offsets [t0, t1) -- and the member start offsets
 * member_pos[n_members + 1] inside the result (n_text = bytes of text in it: the last member is partial).  natac_tbi_push enters them into an incremental index once the caller knows at
 * which byte `file_offset` of the .gz the result is written (results may be produced out of order by several contexts and
 * written in order by one writer); natac_tbi_write serialises the same index natac_tabix_index builds from the finished file. */
typedef struct natac_tbi natac_tbi;
int natac_tbi_create(natac_tbi **out);
void natac_tbi_free(natac_tbi *t);
int natac_batch_format_index_size(natac_batch *b, int64_t *n_groups, int64_t *n_members, int64_t *n_text);
int natac_batch_format_index_fetch(natac_batch *b, int32_t *cid, int64_t *beg, int64_t *end, int64_t *count, uint64_t *t0, uint64_t *t1,
                                   uint64_t *member_pos);
int natac_tbi_push(natac_tbi *t, int64_t n, const char *const *names, int32_t n_names, const int32_t *cid, const int64_t *beg,
                   const int64_t *end, const int64_t *count, const uint64_t *t0, const uint64_t *t1, const uint64_t *member_pos,
                   int64_t n_members, int64_t n_text, int64_t file_offset);
int natac_tbi_write(natac_tbi *t, const char *tbi_path, int64_t *n_records);
/* the device formatter on arbitrary doubles (validation): out_off[i] .. out_off[i+1] = python-2 str(vals[i]); out_cap >= 24 n */
int natac_format_doubles(natac_ctx *ctx, const double *vals, int64_t n, char *out, size_t out_cap, int64_t *out_off, int32_t *n_hard);
/* host restatement of the device BGZF encoder (no GPU): text + line start offsets -> the members natac_batch_format_track(compress)
 * produces for the same text, byte for byte */
int natac_bgzf_lines_host(const char *text, int64_t n, const int64_t *line_off, int64_t n_lines, void *out, size_t out_cap,
                          int64_t *n_bytes);

/* BED-like rows with python-2 float columns, written natively: row r = names[chrom_id[r]] \t start[r] \t end[r] (\t vals[r][c])* --
 * the text of OccPeak.asBed / Nucleosome.asBed (nucleoatac/Occupancy.py:166-171, NucleosomeCalling.py:195-199) for millions of rows.
 * vals is row-major [n_rows x n_cols] (n_cols <= 32), NaN prints as "nan".  append != 0 appends to `path`. */
int natac_write_bed_rows(const char *path, int append, int64_t n_rows, const int32_t *chrom_id, const char *const *names,
                         int32_t n_names, const int64_t *start, const int64_t *end, const double *vals, int32_t n_cols);

/* the same with one more text column at the end of every row, labels[label_id[r]] -- the rows of nucmap_combined.bed
 * (MergedNuc.asBed, nucleoatac/merge.py:23-30: chrom start end occ occ_lower occ_upper reads source) */
int natac_write_bed_rows_labeled(const char *path, int append, int64_t n_rows, const int32_t *chrom_id, const char *const *names,
                                 int32_t n_names, const int64_t *start, const int64_t *end, const double *vals, int32_t n_cols,
                                 const int32_t *label_id, const char *const *labels, int32_t n_labels);

/* bgzip: compress a text file into BGZF members of <= 0xff00 input bytes + the EOF marker (the reference's
 * pysam.tabix_compress, pyatac/utils.py:135-141 / run_nuc.py:204-214). */
int natac_bgzip_file(const char *src, const char *dst, int level, int n_threads);
/* tabix: write the .tbi index ("bed" preset: columns 1/2/3, 0-based half-open, '#' comments) of a BGZF-compressed,
 * position-sorted BED / bedGraph file (the reference's pysam.tabix_index(..., preset="bed"), same call sites).
 * tbi_path NULL: `path` + ".tbi".  n_records (may be NULL) receives the number of indexed lines. */
int natac_tabix_index(const char *path, const char *tbi_path, int n_threads, int64_t *n_records);

/* region reads through the index (pysam.TabixFile.fetch as used by Track.read_track, pyatac/tracks.py:75-87): out[x - start] =
 * the value column (1-based, 4 for bedGraph) of the records of `chrom` overlapping [start, end); bases without a record keep
 * `empty`.  One handle per file and thread. */
typedef struct natac_tbx natac_tbx;
int natac_tbx_open(const char *path, natac_tbx **out);
void natac_tbx_close(natac_tbx *t);
int natac_tbx_read_values(natac_tbx *t, const char *chrom, int64_t start, int64_t end, int value_col, double empty, double *out,
                          int64_t *n_records);
/* the same for n regions in one call (the three occupancy tracks of every chunk of a batch: NucChunk.getOcc,
 * nucleoatac/NucleosomeCalling.py:284-293, NFRChunk.getOcc, NFRCalling.py:63-68): region i = names[chrom_id[i]]:[start[i], end[i]),
 * written to out[out_off[i] ...].  n_threads cursors of the handle take contiguous runs of the list (0 = up to 64); a cursor keeps
 * the members it inflated last, so position-sorted lists inflate and parse every member once. */
int natac_tbx_read_regions(natac_tbx *t, int64_t n, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                           const int64_t *start, const int64_t *end, int value_col, double empty, double *out, const int64_t *out_off,
                           int n_threads, int64_t *n_records);

/* ---- host-side packing of a chunk list (what replaces the per-chunk bamHandle.fetch of pyatac/fragments.pyx:21-36) ---- */
/* pos[c] / tlen[c]: the forward proper-pair reads of chromosome c (natac_bam_ref_reads), pos ascending.  chrom_id[i] < 0: a
 * chunk on a chromosome without reads.  Attached to chunk i are the reads with pos in [start - margin - shift, end + margin)
 * (shift = 4 when atac), as l = pos + shift - start, n = |tlen| - (8 when atac), ordered by centre l + (n-1)//2 (stable).
 * Call once with lpos == NULL: fills frag_off[0..n_chunks] and first[0..n_chunks) (count pass); allocate
 * frag_off[n_chunks] entries and call again with lpos / ilen to fill them (multi-threaded). */
int natac_pack_chunks(int32_t n_chunks, const int64_t *chunk_start, const int64_t *chunk_end, const int32_t *chrom_id, int32_t n_chroms,
                      const int64_t *const *pos, const int64_t *const *tlen, const int64_t *n_per_chrom, int64_t margin, int atac,
                      int64_t *frag_off, int64_t *first, int32_t *lpos, int32_t *ilen, int n_threads);

/* ---- native BAM -> fragment arrays extractor (host side) -------------- */
/* Decode a BAM once (streaming: bounded windows of compressed data, parallel BGZF inflate per window, records may straddle
 * windows; peak memory is independent of the file size) into per-reference arrays of the reads pyatac/fragments.pyx:25 keeps
 * (`is_proper_pair and not is_reverse`): pos = leftmost 0-based coordinate, tlen = |template length|, in file order. */
typedef struct natac_bam natac_bam;
int natac_bam_open(const char *path, int n_threads, natac_bam **out);
void natac_bam_close(natac_bam *bam);
int natac_bam_counts(natac_bam *bam, int32_t *n_refs, int64_t *n_records, int64_t *n_kept);
int natac_bam_ref_info(natac_bam *bam, int32_t ref, char *name, size_t name_len, int64_t *length, int64_t *n_reads);
int natac_bam_ref_reads(natac_bam *bam, int32_t ref, int64_t *pos, int64_t *tlen, int64_t n);
/* The same extraction with the BGZF members inflated and the records walked ON THE DEVICE (csrc/natac_bam_dev.hpp: one lane per
 * member; the record chain through the members is confirmed link by link from the end of the header, so the result is exactly
 * natac_bam_open's).  A file whose chain cannot be confirmed -- or a HIP failure such as no memory for a window -- sends the file
 * through the host decoder inside this call; *on_device (may be NULL) tells which one answered.  Damaged files fail with the host
 * decoder's messages. */
int natac_bam_open_device(natac_ctx *ctx, const char *path, natac_bam **out, int *on_device);
/* test entry: the device's raw-deflate decoder run on the host (one BGZF member payload -> isize bytes); returns its error code */
int natac_inflate_raw_host(const void *src, size_t csize, void *out, size_t isize);

/* ---- objective + finite-difference gradient of K Nucleosome.getFuzz fits (host side; reference NucleosomeCalling.py:92-97, 176-184 as
 * scipy's L-BFGS-B sees them through its numerical differentiation).  nucleoatac/fuzzfit.py advances many fits in lockstep and
 * calls this once per round.  X0, lb, ub: [K][n] (n = 3, 6 or 9); sig, xs: [K][M] padded; lens[K]; exp_loop / exp_data: numpy's own
 * inner loop of float64 exp (the values must be numpy's to the last bit: csrc/natac_fuzzfit.hpp); work: (2 n + n / 3 + 1) K M doubles. */
int natac_fuzz_evaluate(int32_t K, int32_t n, int32_t M, const double *X0, const double *lb, const double *ub, const double *sig,
                        const double *xs, const int64_t *lens, void *exp_loop, void *exp_data, double *work, double *f, double *g);

/* ---- a (gzipped) BED-like table read into columns (host side): what `nucleoatac merge` parses row by row (nucleoatac/merge.py:36-64).
 * cols: 0-based indices of the columns parsed as float64 (correctly rounded, like python's float()); column 0 = chromosome (names in
 * order of first appearance), 1 / 2 = start / end.  BGZF, plain gzip and plain text alike; empty lines are skipped. */
typedef struct natac_bedtab natac_bedtab;
int natac_bedtab_open(const char *path, const int32_t *cols, int32_t n_cols, natac_bedtab **out);
void natac_bedtab_close(natac_bedtab *t);
int natac_bedtab_dims(natac_bedtab *t, int64_t *n_rows, int32_t *n_names);
int natac_bedtab_name(natac_bedtab *t, int32_t i, char *name, size_t name_len);
int natac_bedtab_fetch(natac_bedtab *t, int32_t *chrom_id, int64_t *start, int64_t *end, double *vals);   /* vals: [n_rows][n_cols] */

/* ---- native FASTA loader (host side): the genome as one upper-case byte array per record, what pyatac/seq.py:11-22 /
 * pyatac/bias.py:85-92 fetch region by region through pysam.FastaFile.  Plain-text FASTA; record names end at the first blank. */
typedef struct natac_fasta natac_fasta;
int natac_fasta_open(const char *path, int n_threads, natac_fasta **out);
void natac_fasta_close(natac_fasta *fa);
int natac_fasta_count(natac_fasta *fa, int32_t *n_records);
int natac_fasta_info(natac_fasta *fa, int32_t record, char *name, size_t name_len, int64_t *length);
int natac_fasta_read(natac_fasta *fa, int32_t record, void *out, int64_t n);     /* n must equal the record's length */

/* ---- pinned host memory + device memory pool ---------------------------------------------------------- */
/* Page-locked host buffers (hipHostMalloc): uploads from / downloads into them run at full PCIe rate and asynchronously to
 * the host.  Any caller-owned host pointer of this ABI may be pinned or pageable; results are identical. */
int natac_host_alloc(size_t bytes, void **out);
int natac_host_free(void *p);
/* The library keeps freed device blocks in a per-device cache (hipMalloc / hipFree synchronise the device, which would
 * stall other contexts' overlapping copies); this returns every cached block to the driver. */
int natac_pool_trim(void);

/* ---- profiling (HIP events on the context's stream) ------------------------------------ */
int natac_profile_enable(natac_ctx *ctx, int on);
/* total milliseconds and launch count of kernel class `k` since the last reset */
int natac_profile_get(natac_ctx *ctx, int k, double *ms_total, int64_t *launches);
int natac_profile_reset(natac_ctx *ctx);
/* Resident tracks between the stages of ONE process (`nucleoatac run` = occ -> vprocess -> nuc -> merge -> nfr, cli.py:34-64 of the
 * reference).  The reference hands the occupancy tracks from `occ` to `nuc` / `nfr` through files: Track.write_track + bgzip, then
 * tabix reads per chunk (NucChunk.getOcc, NucleosomeCalling.py:284-293; NFRChunk.getOcc, NFRCalling.py:64-67).  Inside one process the
 * values can stay in HBM instead -- exactly the values a reader of the file would get:
 *   natac_store_adopt   a COPY of the batch's tracks as the file shows them: every run rounded to its twelve printed digits
 *                       (float('%.12g' % v)), NaN where Track.write_track writes no line (write_zero as in natac_batch_format_track);
 *                       returns the segment id, or -1 with *n_hard > 0 when a value cannot be rounded exactly on the device
 *                       (|v| < 1e-11, >= 1e34): nothing is kept and the caller reads the file for these regions; -1 with
                       *n_hard == -1 when the store's HBM budget declines the segment (below) -- same consequence;
 *   natac_store_set_budget  the store never fills the device: a segment is adopted only while the store stays below max_bytes
 *                       (< 0: no cap; env NATAC_STORE_MAX_BYTES) AND the device keeps min_free_bytes for the pipeline's batches
 *                       (< 0: a quarter of the device's memory; env NATAC_STORE_MIN_FREE_BYTES), counting the blocks the library
 *                       caches as free.  The first refusal closes the store: later sub-batches go through the files unasked;
 *   natac_store_read    ranges [offset, offset + length) of slot `slot` (the i-th adopted track) of the named segments, concatenated
 *                       into `out` (host): whole chunks for nfr's gap statistics, single positions for nuc's calls.
 *   natac_store_declined  how many segments the budget declined.
 * Offsets are positions in the batch's flat per-base layout (chunk k starts at out_off[k]).  Thread-safe. */
typedef struct natac_store natac_store;
int natac_store_create(natac_store **out);
void natac_store_free(natac_store *s);
int natac_store_adopt(natac_store *s, natac_batch *b, int32_t n_tracks, const int32_t *tracks, int write_zero, int64_t *segment,
                      int32_t *n_hard);
int natac_store_read(natac_store *s, natac_ctx *ctx, int64_t n, const int64_t *segment, const int64_t *offset, const int64_t *length,
                     int32_t slot, double *out, size_t out_values);
int natac_store_info(natac_store *s, int64_t *n_segments, int64_t *bytes);
int natac_store_set_budget(natac_store *s, int64_t max_bytes, int64_t min_free_bytes);
int natac_store_declined(natac_store *s, int64_t *n_declined);

/* Shader-clock trace (measurement aid, SURVEY.md section 8d asks for achieved rates against peaks that assume a clock): a one-wave
 * sampler kernel on its own stream notes (wall time, shader cycle counter) every interval_us while other launches run; between
 * two samples cycles / time = the clock the chip ran at.  _stop ends it and returns the number of samples and of profiled launches
 * (natac_profile_enable) that fell into the trace; _fetch returns the samples (ms since the trace start, cycle counter) and for every
 * such launch its NATAC_K_* class and its start / end on the same time axis. */
int natac_clock_trace_start(natac_ctx *ctx, int max_samples, int interval_us);
int natac_clock_trace_stop(natac_ctx *ctx, int64_t *n_samples, int64_t *n_intervals);
int natac_clock_trace_fetch(natac_ctx *ctx, int64_t n_samples, double *t_ms, double *cycles, int64_t n_intervals, int32_t *iv_kernel,
                            double *iv_t0_ms, double *iv_t1_ms);
/* stream-ordered timer: natac_timer_start records an event, natac_timer_stop records + syncs + returns ms */
int natac_timer_start(natac_ctx *ctx);
int natac_timer_stop(natac_ctx *ctx, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* NATAC_H */
