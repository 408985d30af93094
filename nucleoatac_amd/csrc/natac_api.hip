// natac_api.hip -- C-ABI (include/natac.h) over the kernels in natac_kernels.hpp.
// Host-side runtime: contexts, batches of packed chunks resident in HBM, tile tables, launches, profiling.
#include "../../include/natac.h"
#ifndef NATAC_GNBL
#define NATAC_GNBL 2
#endif
#include "natac_kernels.hpp"
#include "natac_fft_bg.hpp"
#include "natac_occ_fast.hpp"
#include "natac_cand.hpp"
#include "natac_covsweep.hpp"
#include "natac_textz.hpp"
#include "natac_cores.hpp"
#include "natac_writer.hpp"
#include "natac_tabix.hpp"
#include "natac_pack.hpp"
#include "natac_bam.hpp"
#include "natac_bam_dev.hpp"
#include "natac_fasta.hpp"
#include "natac_fuzzfit.hpp"
#include "natac_bedtab.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

using namespace natac;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? NATAC_E_NOMEM : NATAC_E_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

struct natac_ctx {
    int device = 0;
    hipStream_t stream = nullptr;    // every stage, uploads, drop-ins: one stream
    hipStream_t copy_stream = nullptr;   // natac_batch_format_fetch_begin: a finished result leaves the device while the next track is formatted
    hipDeviceProp_t prop;
    // constants
    double *d_vmat = nullptr, *d_vmat_pad = nullptr, *d_srow = nullptr, *d_sizes = nullptr;   // d_vmat_pad: VMatDev::matp
    double *d_lrt = nullptr;         // VMatDev::lrt (natac_lr_table), formed for model generation lrt_gen
    long long lrt_gen = -1;
    size_t lrt_cap = 0;
    int vlower = 0, vupper = 0, vw = 0, R = 0, W = 0, sizes_upper = 0;
    bool have_vmat = false, have_sizes = false, srow_dirty = true;
    bool vmat_zero = false, srow_zero = false;
    long long model_gen = 0;         // bumped by natac_set_vmat / natac_set_sizes
    // FFT background path: twiddles (once) and template spectra (per V-plot)
    double *d_fft_tw = nullptr, *d_fft_k = nullptr;
    double *d_fft_mtab = nullptr, *d_fft_swt = nullptr;   // natac_fft_edge_table_mfma (the edge pass of extended tiles)
    bool bg_ext = true;              // NATAC_BG_EXT=0: no extended FFT tiles (A-B timing / validation of the edge pass)
    bool fft_dirty = true, bg_direct = false, occ_ordered = true, occ_zero_nfr = false;
    std::vector<double> h_sizes;
    double *d_nucp = nullptr, *d_nfrp = nullptr, *d_alphas = nullptr;
    int occ_upper = 0, n_alpha = 0, step = 0, halfstep = 0, flank = 0;
    double cutoff = 0;
    bool have_occ = false;
    int occ_zero_flags = 0;          // zero pattern of nuc_probs / nfr_probs (bits as in the MLE kernel)
    // fast occupancy path (natac_occ_fast.hpp): model tables + eligibility
    double *d_occ_q4 = nullptr, *d_occ_rho = nullptr;
    int occ_nm = 0;
    bool occ_fast_ok = false, occ_force_general = false, occ_rn16 = false;
    double occ_b_floor = 0;          // see OccModelDev::b_floor
    // gaussian windows (cached by (M, sd))
    double *d_win_nuc = nullptr, *d_win_occ = nullptr;
    double *d_wb_occ = nullptr;      // block weights of natac_occ_smooth_blk for (win_occ_M, wb_step)
    int wb_M = 0, wb_step = 0;
    int win_nuc_M = 0, win_occ_M = 0;
    double win_nuc_sd = -1, win_occ_sd = -1;
    double win_nuc_sum = 0, win_occ_sum = 0;   // sequential sums of the window values (the all-valid denominator)
    // profiling
    bool profiling = false;
    struct Ev { int k; hipEvent_t a, b; hipStream_t st; };
    std::vector<Ev> pending;
    double prof_ms[NATAC_K_COUNT] = {0};
    int64_t prof_n[NATAC_K_COUNT] = {0};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // shader-clock trace (natac_clock_trace_*): a one-wave sampler kernel on its own stream + where the profiled launches fall on its time axis
    hipStream_t ck_stream = nullptr;
    hipEvent_t ck_start = nullptr;
    long long *d_ck = nullptr;       // [0] = samples written, [1] = stop flag, then (wall ticks, shader cycles) pairs
    int ck_cap = 0;
    bool ck_active = false;
    struct Iv { int k; float t0, t1; };
    std::vector<Iv> ck_iv;
    // device-side track writer (natac_textz.hpp): power-of-ten table of the '%.12g' formatter, CRC-32 tables
    natac_text::P10 *d_p10 = nullptr;
    natac_deflate::CrcTables *d_crc = nullptr;
};

struct natac_bam {
    natac_bamio::Bam *impl = nullptr;
};

struct natac_batch {
    natac_ctx *ctx = nullptr;
    int nc = 0;
    long long nf = 0, nb = 0, total_bp = 0, total_grid = 0;
    int bias_left = 0, bias_right = 0;
    std::vector<int> h_len;
    std::vector<long long> h_out_off, h_grid_off;
    int *d_len = nullptr, *d_lpos = nullptr, *d_ilen = nullptr, *d_centre = nullptr, *d_status = nullptr;
    long long *d_frag_off = nullptr, *d_bias_off = nullptr, *d_out_off = nullptr, *d_grid_off = nullptr;
    double *d_bias = nullptr, *d_ebias = nullptr;
    unsigned long long *d_occ_minkey = nullptr;   // natac_occ_smooth_blk: per-chunk minimum finite smoothed occupancy (double_key)
    int *d_occ_nan = nullptr, n_tiles_os = 0, os_width = 0;
    int2 *d_tiles_os = nullptr, *d_tiles1k = nullptr;
    int n_tiles1k = 0;
    bool prefill_valid = false;                   // OCC_PREFILL holds this run's values (written by the generic path or on demand)
    bool bg_valid = false;                        // BACKGROUND holds this run's values (the FFT path leaves it to materialise_bg)
    int2 *d_tiles256 = nullptr, *d_tiles_bg = nullptr, *d_tiles_occ = nullptr, *d_ranges_occ = nullptr, *d_ranges256 = nullptr;
    long long *d_tile256_first = nullptr;   // [nc + 1] first 256-base tile of every chunk (the candidates' way into d_ranges256)
    int *d_order_occ = nullptr;      // natac_tile_heavy: {count, claims, list[HEAVY_CAP], flag bytes[n_tiles_occ]} of the occupancy tiles
    int ranges256_w = -1;
    int ranges_occ_key[3] = {-1, -1, -1};   // (step, halfstep, flank) the occupancy tiles' fragment ranges were formed for
    int n_tiles256 = 0, n_tiles_bg = 0, n_tiles_occ = 0, bgG = 0;   // bgG: lanes' output count of the direct kernel, -1 = FFT tiles
    int grid_step = 0, grid_half = 0;
    // fast occupancy path: per-block sums of g_n / g_f, their tile table and the list of tiles left to natac_occ_mle
    long long *d_blk_off = nullptr;
    long long total_blocks = 0;
    double *d_gsum = nullptr;
    int2 *d_tiles_gs = nullptr;
    int n_tiles_gs = 0, gs_Q = -1;
    int *d_defer = nullptr;          // [0] = count, [1 ..] = tile indices
    // occupancy peaks (natac_run_occ_peaks): OccPeak values + keep flags of the last search, per-chunk nuc_dist
    double *d_opk_vals = nullptr, *d_nuc_dist = nullptr;
    int *d_opk_keep = nullptr;
    long long opk_cap = 0, opk_n = -1;
    int nd_upper = 0;
    double *d_track[NATAC_T_COUNT] = {nullptr};
    double *d_grid[3] = {nullptr, nullptr, nullptr};
    bool nuc_done = false, occ_done = false, ins_done = false, cov_from_nuc = false, ebias_fresh = false;
    // device-side candidate search
    double *d_jitter = nullptr, *d_pk_out = nullptr;
    long long *d_cap_off = nullptr, *d_pk_offs = nullptr;
    int *d_slot = nullptr, *d_pk_count = nullptr, *d_pk_chunk = nullptr, *d_pk_pos = nullptr;
    double *d_pk_big = nullptr;       // natac_peaks_chunk: global (sig, pos, state) lists of chunks with more maxima than fit in LDS
    long long pk_big_slots = 0;
    long long n_jitter = 0, pk_cap = 0, pk_n = -1, slot_total = 0;
    int pk_order = -1;
    bool pk_has_stats = false;
    int nuc_w = -1, nuc_upper = -1;   // V-plot geometry natac_run_nuc ran with (coverage tracks depend on it)
    double *d_bnum = nullptr, *d_bcov = nullptr;   // per-base sum B V / sum B of the background kernel (candidate statistics)
    unsigned char *d_fmt_out = nullptr;            // result of the last natac_batch_format_track (text or BGZF members)
    std::vector<std::pair<unsigned char *, hipEvent_t>> fmt_pending;   // results on their way to the host (natac_batch_format_fetch_begin)
    long long fmt_bytes = -1;
    // tabix records of that result (compress mode): runs of lines per leaf bin, member offsets, chromosome names
    std::vector<natac_textz::GroupRec> fmt_groups;
    std::vector<unsigned long long> fmt_member_pos;
    std::vector<std::string> fmt_names;
    long long fmt_text_bytes = 0;
    long long nuc_gen = -1;                        // model generation natac_run_nuc ran with
};

static hipError_t sync_all(natac_ctx *c) { return hipStreamSynchronize(c->stream); }

static int track_ready(natac_batch *b, int t);
static int fmt_pending_wait(natac_batch *b);

static void prof_begin(natac_ctx *c, int k, natac_ctx::Ev &ev, hipStream_t st = nullptr) {
    ev.k = k;
    ev.a = ev.b = nullptr;
    ev.st = nullptr;
    if (!c->profiling) return;
    (void)hipEventCreate(&ev.a);
    (void)hipEventCreate(&ev.b);
    ev.st = st ? st : c->stream;
    (void)hipEventRecord(ev.a, ev.st);
}
static void prof_end(natac_ctx *c, natac_ctx::Ev &ev) {
    if (!c->profiling) return;
    (void)hipEventRecord(ev.b, ev.st);
    c->pending.push_back(ev);
}
static void prof_collect(natac_ctx *c) {
    for (auto &e : c->pending) {
        float ms = 0;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.k] += ms;
            c->prof_n[e.k] += 1;
            float t0 = 0, t1 = 0;
            if (c->ck_active && hipEventElapsedTime(&t0, c->ck_start, e.a) == hipSuccess &&
                hipEventElapsedTime(&t1, c->ck_start, e.b) == hipSuccess)
                c->ck_iv.push_back({e.k, t0, t1});
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->pending.clear();
}

// Device memory goes through a small caching pool (per device, shared by the contexts of a process, thread-safe): a freed block
// is kept and handed to the next request of (nearly) the same size.  Batches of similar shape follow each other in every
// driver (run_occ / run_nuc / bench), and hipMalloc / hipFree synchronise the whole device -- which would also stall the
// streams of OTHER contexts that overlap their copies with this one's kernels.  Every use of a block is ordered on its
// context's stream and a batch is only freed after that stream has drained (natac_batch_free), so reuse is safe.
// NATAC_POOL=0 disables the cache; natac_pool_trim() returns the cached blocks to the driver.
#include <mutex>
#include <map>
#include <unordered_map>
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks[16];          // per device
    std::unordered_map<void *, std::pair<int, size_t>> live;  // ptr -> (device, bytes)
    bool enabled = true;
    DevPool() { const char *e = getenv("NATAC_POOL"); enabled = !(e && e[0] == '0'); }
    void trim_locked(int dev) {
        for (auto &kv : free_blocks[dev]) (void)hipFree(kv.second);
        free_blocks[dev].clear();
    }
};
static DevPool g_pool;

static hipError_t pool_alloc(void **p, size_t bytes) {
    *p = nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (g_pool.enabled) {
        auto it = g_pool.free_blocks[dev].lower_bound(bytes);
        if (it != g_pool.free_blocks[dev].end() && it->first <= bytes + bytes / 8 + 4096) {
            *p = it->second;
            g_pool.live[*p] = {dev, it->first};
            g_pool.free_blocks[dev].erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && !g_pool.free_blocks[dev].empty()) {      // give the cached blocks back and retry
        (void)hipGetLastError();
        g_pool.trim_locked(dev);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) g_pool.live[*p] = {dev, bytes};
    else (void)hipGetLastError();      // reported through the return value; a later hipGetLastError() on this thread must not see it again
    return e;
}

// bytes the device could still hand out: what the driver reports as free + the blocks this process caches for reuse
static hipError_t pool_mem_info(int dev, size_t *avail, size_t *total) {
    size_t fr = 0, tot = 0;
    hipError_t e = hipMemGetInfo(&fr, &tot);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto &kv : g_pool.free_blocks[dev & 15]) fr += kv.first;
    *avail = fr;
    *total = tot;
    return hipSuccess;
}

static void dev_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) { (void)hipFree(p); return; }
    const int dev = it->second.first;
    const size_t bytes = it->second.second;
    g_pool.live.erase(it);
    if (g_pool.enabled) g_pool.free_blocks[dev].insert({bytes, p});
    else (void)hipFree(p);
}

template <class T>
static int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(pool_alloc((void **)p, n * sizeof(T)));
    return NATAC_OK;
}
template <class T>
static int dev_upload(natac_ctx *c, T **p, const T *src, size_t n) {
    int rc = dev_alloc(p, n);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpyAsync(*p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return NATAC_OK;
}

// pick the per-lane output count G of the background kernel: minimise idle lanes (sum of tile widths) with a small
// penalty for the halo work of narrow tiles.
static int choose_bg_G(const natac_batch *b, int W) {
    const int cand[4] = {7, 9, 13, 17};
    double best = 1e300;
    int bestG = 9;
    // histogram-free: sample up to 65536 chunks evenly
    const int stride = std::max(1, b->nc / 65536);
    for (int ci = 0; ci < 4; ++ci) {
        const int G = cand[ci], TW = 64 * G;
        double cost = 0;
        for (int i = 0; i < b->nc; i += stride) {
            const int nt = (b->h_len[i] + TW - 1) / TW;
            cost += (double)nt * (TW + (W - 1) * 0.15 * 8);
        }
        if (cost < best) { best = cost; bestG = G; }
    }
    return bestG;
}

static void launch_candidates(natac_ctx *c, const ChunkTable &ct, const VMatDev &vm, const int *d_cc, const int *d_cp, long long n,
                              const double *nuc_cov, const double *norm, const double *bnum, const double *bcov, double *lr,
                              double *var, double *z, const long long *tile_first = nullptr, const int2 *ranges256 = nullptr) {
    const int EW = c->W + ((c->vupper - 2) >> 1) + ((c->vupper - 1) >> 1);
    const int ZN = (((c->vupper - 2) >> 1) + ((c->vupper - 1) >> 1) + 5 + 32) & ~1, ON = (c->W + 1) & ~1;
    const size_t lds4 = ((size_t)4 * CAND_PER_WAVE * ((EW + 1) & ~1) + ZN + ON) * sizeof(double);
    // paired-row kernel (natac_cand.hpp): needs the background kernel's window sums, no single-cell row, whole row pairs, a template
    // at least 64 columns wide (every lane's first column exists) and
    // a model without exact zeros (those take the per-cell zero test of natac_candidates4).  NATAC_CAND_OLD=1: validation.
    const size_t ldsp = (size_t)4 * CAND_PER_WAVE * CANDP_STRIDE * sizeof(double);
    bool paired = bnum && bcov && c->vlower >= 2 && (c->R & 1) == 0 && c->W >= 64 && !vm.has_zero && EW <= CANDP_STRIDE &&
                  !getenv("NATAC_CAND_FULL") && !getenv("NATAC_CAND_OLD");
    if (paired && (c->lrt_gen != c->model_gen || !c->d_lrt)) {      // log(V / s) of the current model, on the launch stream (natac_lr_table)
        if (!c->d_lrt || c->lrt_cap < (size_t)c->R * c->W) {
            dev_free(c->d_lrt);
            c->d_lrt = nullptr;
            c->lrt_cap = 0;
            if (dev_alloc(&c->d_lrt, (size_t)c->R * c->W) == NATAC_OK) c->lrt_cap = (size_t)c->R * c->W;
            else paired = false;                                     // out of memory: the per-cell kernel below needs no table
        }
        if (paired) {
            hipLaunchKernelGGL(natac_lr_table, dim3((unsigned)((c->R * c->W + 255) / 256)), dim3(256), 0, c->stream, c->d_vmat, c->d_srow, c->R, c->W,
                               c->d_lrt);
            c->lrt_gen = c->model_gen;
        }
    }
    if (paired) {
        VMatDev vml = vm;
        vml.lrt = c->d_lrt;
        const long long per_block = 4 * CAND_PER_WAVE;
        const dim3 grid((unsigned)((n + per_block - 1) / per_block));
        if (c->vlower & 1)
            hipLaunchKernelGGL((natac_candidates_paired<true>), grid, dim3(256), ldsp, c->stream, ct, vml, d_cc, d_cp, (int)n, nuc_cov, norm,
                               bnum, bcov, tile_first, ranges256, lr, var, z);
        else
            hipLaunchKernelGGL((natac_candidates_paired<false>), grid, dim3(256), ldsp, c->stream, ct, vml, d_cc, d_cp, (int)n, nuc_cov, norm,
                               bnum, bcov, tile_first, ranges256, lr, var, z);
        return;
    }
    if (lds4 <= 64 * 1024) {
        const long long per_block = 4 * CAND_PER_WAVE;
        if (bnum && bcov && !getenv("NATAC_CAND_FULL"))   // NATAC_CAND_FULL=1: all four window sums in the kernel (validation)
            hipLaunchKernelGGL((natac_candidates4<true>), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), lds4, c->stream,
                               ct, vm, d_cc, d_cp, (int)n, nuc_cov, norm, bnum, bcov, lr, var, z);
        else
            hipLaunchKernelGGL((natac_candidates4<false>), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), lds4, c->stream,
                               ct, vm, d_cc, d_cp, (int)n, nuc_cov, norm, bnum, bcov, lr, var, z);
    } else {   // very wide templates: one workgroup per candidate
        hipLaunchKernelGGL(natac_candidates, dim3((unsigned)n), dim3(256), (size_t)(EW + 2) * sizeof(double), c->stream, ct, vm, d_cc,
                           d_cp, nuc_cov, norm, lr, var, z);
    }
}

template <int G>
static void launch_bg(natac_batch *b, const ChunkTable &ct, const VMatDev &vm) {
    natac_ctx *c = b->ctx;
    constexpr int W = 121;
    const int PW = 64 * G + W - 1;
    const int EW = PW + ((vm.upper - 2) >> 1) + ((vm.upper - 1) >> 1);
    const size_t lds = ((size_t)((EW + 1) & ~1) + PW) * sizeof(double);
    hipLaunchKernelGGL((natac_background<G, W>), dim3(b->n_tiles_bg), dim3(64), lds, c->stream, ct, b->d_tiles_bg, vm,
                       b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_RAW], b->d_track[NATAC_T_BACKGROUND],
                       b->d_track[NATAC_T_NORM], b->d_bnum, b->d_bcov);
}

/* ---------------- device-side track writer (natac_textfmt.hpp, natac_deflate.hpp, natac_textz.hpp) ---------------- */

static int ensure_text_tables(natac_ctx *c) {
    if (c->d_p10 && c->d_crc) return NATAC_OK;
    int rc;
    if (!c->d_p10 && (rc = dev_upload(c, &c->d_p10, natac_text::H_P10, sizeof(natac_text::H_P10) / sizeof(natac_text::P10)))) return rc;
    if (!c->d_crc) {
        natac_deflate::CrcTables t;
        natac_deflate::crc_init(t);
        if ((rc = dev_upload(c, &c->d_crc, &t, 1))) return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return NATAC_OK;
}

struct TmpFree {               // frees device temporaries at scope exit
    std::vector<void *> v;
    ~TmpFree() { for (void *p : v) dev_free(p); }
    template <class T> T *keep(T *p) { v.push_back((void *)p); return p; }
};

// exclusive scan of in[0..n) into out[0..n], out[n] = total (device arrays; n > 0).  The block sums are a temporary of the CALLER's scope
// (`tmp`, freed after the caller's last synchronisation): freeing them here needed a stream synchronisation per scan -- six per formatted
// track -- because the pool may hand a freed block to another context at once.
template <class T>
static int dev_scan(natac_ctx *c, const T *in, long long n, unsigned long long *out, TmpFree &tmp) {
    using namespace natac_textz;
    const long long nblk = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    unsigned long long *sums = nullptr;
    int rc = dev_alloc(&sums, (size_t)nblk + 1);
    if (rc) return rc;
    tmp.keep(sums);
    hipLaunchKernelGGL((tz_scan_block_sums<T>), dim3((unsigned)nblk), dim3(256), 0, c->stream, in, n, sums);
    hipLaunchKernelGGL(tz_scan_sums, dim3(1), dim3(1024), 0, c->stream, sums, nblk);
    hipLaunchKernelGGL((tz_scan_final<T>), dim3((unsigned)nblk), dim3(256), 0, c->stream, in, n, sums, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NATAC_E_HIP, "scan: %s", hipGetErrorString(e));
    return NATAC_OK;
}

// the two scans over a track's runs (byte offsets and indices of its lines) in one pass over the length array (tz_scan2_*)
static int dev_scan2(natac_ctx *c, const unsigned char *in, long long n, unsigned long long *out_sum, unsigned long long *out_cnt, TmpFree &tmp) {
    using namespace natac_textz;
    const long long nblk = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    unsigned long long *sums = nullptr;
    int rc = dev_alloc(&sums, 2 * ((size_t)nblk + 1));
    if (rc) return rc;
    tmp.keep(sums);
    hipLaunchKernelGGL(tz_scan2_block_sums, dim3((unsigned)nblk), dim3(256), 0, c->stream, in, n, nblk, sums);
    hipLaunchKernelGGL(tz_scan_sums, dim3(1), dim3(1024), 0, c->stream, sums, nblk);
    hipLaunchKernelGGL(tz_scan_sums, dim3(1), dim3(1024), 0, c->stream, sums + nblk + 1, nblk);
    hipLaunchKernelGGL(tz_scan2_final, dim3((unsigned)nblk), dim3(256), 0, c->stream, in, n, nblk, sums, out_sum, out_cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NATAC_E_HIP, "scan: %s", hipGetErrorString(e));
    return NATAC_OK;
}

// text / BGZF of a device array of per-base values laid out like the batch's tracks.  Result stays in b->d_fmt_out.
static int format_values(natac_batch *b, const double *d_vals, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                         const int64_t *chunk_start, int write_zero, int compress, int64_t *n_bytes, int64_t *n_text_bytes, int64_t *n_lines,
                         int32_t *n_hard) {
    using namespace natac_textz;
    natac_ctx *c = b->ctx;
    int rc = ensure_text_tables(c);
    if (rc) return rc;
    dev_free(b->d_fmt_out);
    b->d_fmt_out = nullptr;
    b->fmt_bytes = -1;
    b->fmt_groups.clear();
    b->fmt_member_pos.clear();
    b->fmt_names.assign(names, names + n_names);
    // name table
    std::vector<char> cat;
    std::vector<int> noff(1, 0);
    for (int i = 0; i < n_names; ++i) {
        const size_t l = names[i] ? strlen(names[i]) : 0;
        if (l == 0 || l > 64) return fail(NATAC_E_ARG, "chromosome name %d must have 1..64 characters", i);
        cat.insert(cat.end(), names[i], names[i] + l);
        noff.push_back((int)cat.size());
    }
    // longest line this call can form: name + two coordinates + value text + 4 separators.  tz_write_lines stages the 256 lines of a
    // workgroup in LDS; sized for THIS bound (typically ~52 bytes per line) instead of the format's 160 it keeps 8 waves per SIMD
    // resident instead of 3.
    size_t max_name = 0;
    long long max_coord = 0, min_coord = 0;
    for (int i = 0; i < b->nc; ++i) {
        if (chrom_id[i] < 0 || chrom_id[i] >= n_names) return fail(NATAC_E_ARG, "chunk %d: chromosome id %d out of range", i, chrom_id[i]);
        max_name = std::max(max_name, (size_t)(noff[chrom_id[i] + 1] - noff[chrom_id[i]]));
        max_coord = std::max<long long>(max_coord, (long long)chunk_start[i] + b->h_len[i]);
        min_coord = std::min<long long>(min_coord, (long long)chunk_start[i]);
    }
    const int line_cap = std::min<int>(MAX_LINE, (int)max_name + 2 * std::max(natac_text::digits_i64(max_coord), natac_text::digits_i64(min_coord)) + VTXT + 4);
    if (b->total_bp >= 0xffffffffLL) return fail(NATAC_E_ARG, "batch too long for the device writer (%lld bases)", b->total_bp);
    TmpFree tmp;
    char *d_names = nullptr;
    int *d_noff = nullptr, *d_cid = nullptr, *d_tc = nullptr, *d_C = nullptr, *d_hard = nullptr;
    long long *d_cs = nullptr, *d_line_off = nullptr;
    unsigned long long *d_tb = nullptr, *d_boff = nullptr, *d_lidx = nullptr;
    unsigned int *d_R = nullptr;
    unsigned char *d_len8 = nullptr, *d_text = nullptr;
#define TRYF(x) do { if ((rc = (x)) != NATAC_OK) return rc; } while (0)
    TRYF(dev_upload(c, &d_names, cat.data(), cat.size())); tmp.keep(d_names);
    TRYF(dev_upload(c, &d_noff, noff.data(), noff.size())); tmp.keep(d_noff);
    TRYF(dev_upload(c, &d_cid, chrom_id, (size_t)b->nc)); tmp.keep(d_cid);
    TRYF(dev_upload(c, (long long **)&d_cs, (const long long *)chunk_start, (size_t)b->nc)); tmp.keep(d_cs);
    TRYF(dev_alloc(&d_hard, 1)); tmp.keep(d_hard);
    HIPCHK(hipMemsetAsync(d_hard, 0, sizeof(int), c->stream));
    TextJob job;
    job.vals = d_vals; job.out_off = b->d_out_off; job.chunk_len = b->d_len; job.tiles = b->d_tiles256; job.ntiles = b->n_tiles256;
    job.chrom_id = d_cid; job.chunk_start = d_cs; job.names = d_names; job.name_off = d_noff; job.p10 = c->d_p10;
    job.write_zero = write_zero & 1; job.keep_before_nan = (write_zero >> 1) & 1;
    const int nt = b->n_tiles256;
    TRYF(dev_alloc(&d_tc, (size_t)nt)); tmp.keep(d_tc);
    TRYF(dev_alloc(&d_tb, (size_t)nt + 1)); tmp.keep(d_tb);
    hipLaunchKernelGGL(tz_flags_count, dim3(nt), dim3(256), 0, c->stream, job, d_tc);
    TRYF(dev_scan(c, d_tc, (long long)nt, d_tb, tmp));
    unsigned long long nruns = 0;
    HIPCHK(hipMemcpyAsync(&nruns, d_tb + nt, sizeof nruns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    TRYF(dev_alloc(&d_R, (size_t)nruns)); tmp.keep(d_R);
    TRYF(dev_alloc(&d_C, (size_t)nruns)); tmp.keep(d_C);
    hipLaunchKernelGGL(tz_scatter_runs, dim3(nt), dim3(256), 0, c->stream, job, d_tb, d_R, d_C);
    TRYF(dev_alloc(&d_len8, (size_t)nruns)); tmp.keep(d_len8);
    const unsigned rb = (unsigned)((nruns + 255) / 256);
    unsigned long long *d_vtxt = nullptr;             // text of every run's value (tz_line_len -> tz_write_lines)
    TRYF(dev_alloc(&d_vtxt, (size_t)nruns * (natac_textz::VTXT / 8))); tmp.keep(d_vtxt);
    hipLaunchKernelGGL(tz_line_len, dim3(rb), dim3(256), 0, c->stream, job, (long long)nruns, d_R, d_C, d_len8, d_hard, d_vtxt);
    TRYF(dev_alloc(&d_boff, (size_t)nruns + 1)); tmp.keep(d_boff);
    TRYF(dev_alloc(&d_lidx, (size_t)nruns + 1)); tmp.keep(d_lidx);
    TRYF(dev_scan2(c, d_len8, (long long)nruns, d_boff, d_lidx, tmp));
    unsigned long long n_text = 0, nlines = 0;
    int hard = 0;
    HIPCHK(hipMemcpyAsync(&n_text, d_boff + nruns, sizeof n_text, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&nlines, d_lidx + nruns, sizeof nlines, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&hard, d_hard, sizeof hard, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    b->fmt_text_bytes = (long long)n_text;
    if (n_text_bytes) *n_text_bytes = (int64_t)n_text;
    if (n_lines) *n_lines = (int64_t)nlines;
    if (n_hard) *n_hard = hard;
    if (n_text == 0) { b->fmt_bytes = 0; if (n_bytes) *n_bytes = 0; return NATAC_OK; }
    TRYF(dev_alloc(&d_text, (size_t)n_text + 64));
    TRYF(dev_alloc(&d_line_off, (size_t)nlines + 1)); tmp.keep(d_line_off);
    int *d_lcid = nullptr;
    long long *d_lbeg = nullptr, *d_lend = nullptr;
    if (compress) {
        TRYF(dev_alloc(&d_lcid, (size_t)nlines)); tmp.keep(d_lcid);
        TRYF(dev_alloc(&d_lbeg, (size_t)nlines)); tmp.keep(d_lbeg);
        TRYF(dev_alloc(&d_lend, (size_t)nlines)); tmp.keep(d_lend);
    }
    hipLaunchKernelGGL(tz_write_lines, dim3(rb), dim3(256), (size_t)256 * line_cap + 32, c->stream, job, (long long)nruns, d_R, d_C, d_len8, d_boff, d_lidx, d_vtxt, d_text,
                       d_line_off, d_lcid, d_lbeg, d_lend);
    if (!compress) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { dev_free(d_text); return fail(NATAC_E_HIP, "format_track: %s", hipGetErrorString(e)); }
        b->d_fmt_out = d_text;
        b->fmt_bytes = (long long)n_text;
        if (n_bytes) *n_bytes = (int64_t)n_text;
        return NATAC_OK;
    }
    tmp.keep(d_text);
    namespace nd = natac_deflate;
    const long long nblk = ((long long)n_text + nd::BLK - 1) / nd::BLK;
    unsigned int *d_hist = nullptr, *d_sizes = nullptr;
    nd::Codes *d_codes = nullptr;
    unsigned char *d_regions = nullptr, *d_out = nullptr;
    unsigned long long *d_pos = nullptr;
    TRYF(dev_alloc(&d_hist, (size_t)nd::NLL + nd::ND)); tmp.keep(d_hist);
    HIPCHK(hipMemsetAsync(d_hist, 0, (nd::NLL + nd::ND) * sizeof(unsigned int), c->stream));
    // line segments per member -> offsets of the per-segment records pass A of the emit kernel leaves for its pass B
    unsigned int *d_nseg = nullptr;
    unsigned long long *d_segbase = nullptr;
    unsigned int *d_stage = nullptr;
    unsigned short *d_segbits = nullptr;
    long long *d_k0 = nullptr;
    TRYF(dev_alloc(&d_nseg, (size_t)nblk)); tmp.keep(d_nseg);
    TRYF(dev_alloc(&d_segbase, (size_t)nblk + 1)); tmp.keep(d_segbase);
    TRYF(dev_alloc(&d_k0, (size_t)nblk)); tmp.keep(d_k0);
    hipLaunchKernelGGL(tz_member_nseg, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, c->stream, d_line_off, (long long)nlines,
                       (long long)n_text, nblk, d_nseg, d_k0);
    TRYF(dev_scan(c, d_nseg, nblk, d_segbase, tmp));
    TRYF(dev_alloc(&d_segbits, (size_t)nlines + (size_t)nblk + 64)); tmp.keep(d_segbits);  // every line once + one more per straddled member border
    TRYF(dev_alloc(&d_stage, (size_t)nblk * STAGE_WORDS)); tmp.keep(d_stage);             // the lines' finished bits between the emit kernel's two passes
    // token histogram of a sample of the members (every member of a small batch): natac_deflate.hpp, sample_stride
    const int stride = nd::sample_stride(nblk);
    const long long counted = (nblk + stride - 1) / stride;
    const size_t lds_count = 65536 + (nd::NLL + nd::ND) * sizeof(unsigned int);
    hipLaunchKernelGGL(tz_count_tokens, dim3((unsigned)counted), dim3(TZ_THREADS), lds_count, c->stream, d_text, (long long)n_text, d_line_off,
                       (long long)nlines, stride, d_nseg, d_k0, d_hist);
    unsigned int hist[nd::NLL + nd::ND];
    HIPCHK(hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    nd::finish_hist(hist, hist + nd::NLL, counted, stride);      // one end-of-block per counted member; a code for every symbol when sampled
    nd::Codes codes;
    if (!nd::build_codes(hist, hist + nd::NLL, codes)) return fail(NATAC_E_ARG, "format_track: Huffman table description too long");
    TRYF(dev_upload(c, &d_codes, &codes, 1)); tmp.keep(d_codes);
    TRYF(dev_alloc(&d_regions, (size_t)nblk * nd::REGION)); tmp.keep(d_regions);
    HIPCHK(hipMemsetAsync(d_regions, 0, (size_t)nblk * nd::REGION, c->stream));
    TRYF(dev_alloc(&d_sizes, (size_t)nblk)); tmp.keep(d_sizes);
    TRYF(dev_alloc(&d_pos, (size_t)nblk + 1)); tmp.keep(d_pos);
    const size_t lds_emit = 65536 + ((sizeof(nd::Codes) + 15) & ~(size_t)15);
    hipLaunchKernelGGL(tz_emit_members, dim3((unsigned)nblk), dim3(TZ_THREADS), lds_emit, c->stream, d_text, (long long)n_text, d_line_off,
                       (long long)nlines, d_nseg, d_k0, d_segbase, d_stage, d_segbits, d_codes, c->d_crc, d_regions, d_sizes);
    HIPCHK(hipGetLastError());
    TRYF(dev_scan(c, d_sizes, nblk, d_pos, tmp));
    b->fmt_member_pos.resize((size_t)nblk + 1);
    HIPCHK(hipMemcpyAsync(b->fmt_member_pos.data(), d_pos, ((size_t)nblk + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    {   // tabix records: runs of lines per leaf bin (tz_group_*)
        unsigned char *d_gf = nullptr;
        unsigned long long *d_gidx = nullptr;
        long long *d_first = nullptr;
        GroupRec *d_rec = nullptr;
        const unsigned lb = (unsigned)((nlines + 255) / 256);
        TRYF(dev_alloc(&d_gf, (size_t)nlines)); tmp.keep(d_gf);
        TRYF(dev_alloc(&d_gidx, (size_t)nlines + 1)); tmp.keep(d_gidx);
        hipLaunchKernelGGL(tz_group_flags, dim3(lb), dim3(256), 0, c->stream, (long long)nlines, d_lcid, d_lbeg, d_lend, d_gf);
        TRYF(dev_scan(c, d_gf, (long long)nlines, d_gidx, tmp));
        unsigned long long ngroups = 0;
        HIPCHK(hipMemcpyAsync(&ngroups, d_gidx + nlines, sizeof ngroups, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        TRYF(dev_alloc(&d_first, (size_t)ngroups)); tmp.keep(d_first);
        TRYF(dev_alloc(&d_rec, (size_t)ngroups)); tmp.keep(d_rec);
        hipLaunchKernelGGL(tz_group_starts, dim3(lb), dim3(256), 0, c->stream, (long long)nlines, d_gf, d_gidx, d_first);
        hipLaunchKernelGGL(tz_group_records, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, c->stream, (long long)ngroups, d_first,
                           (long long)nlines, d_lcid, d_lbeg, d_lend, d_line_off, (long long)n_text, d_rec);
        b->fmt_groups.resize((size_t)ngroups);
        HIPCHK(hipMemcpyAsync(b->fmt_groups.data(), d_rec, (size_t)ngroups * sizeof(GroupRec), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    const unsigned long long total = b->fmt_member_pos[(size_t)nblk];
    TRYF(dev_alloc(&d_out, (size_t)total + 64));
    hipLaunchKernelGGL(tz_compact, dim3((unsigned)nblk), dim3(256), 0, c->stream, d_regions, d_sizes, d_pos, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { dev_free(d_out); return fail(NATAC_E_HIP, "format_track: %s", hipGetErrorString(e)); }
#undef TRYF
    b->d_fmt_out = d_out;
    b->fmt_bytes = (long long)total;
    if (n_bytes) *n_bytes = (int64_t)total;
    return NATAC_OK;
}

// natac_occ_gsum<STEP, REM> + natac_occ_decide<STEP, RN> for the model's step and window remainder
struct OccFastLaunch { natac_batch *b; const ChunkTable &ct; const OccFastDev &of; size_t lds_gs, lds_od; int rem; bool rn16; };
template <int STEP, int REM = 0>
static void occ_gsum_launch(const OccFastLaunch &f) {
    if constexpr (REM < STEP) {
        if (f.rem == REM)
            hipLaunchKernelGGL((natac_occ_gsum<STEP, REM>), dim3(f.b->n_tiles_gs), dim3(256), f.lds_gs, f.b->ctx->stream, f.ct, f.b->d_tiles_gs, f.of,
                               f.b->d_blk_off, f.b->total_blocks, f.b->d_gsum);
        else
            occ_gsum_launch<STEP, REM + 1>(f);
    }
}
template <int STEP>
static void occ_fast_launch(const OccFastLaunch &f) {
    natac_batch *b = f.b;
    occ_gsum_launch<STEP>(f);
    auto kd = (f.of.flags & 2) ? (f.rn16 ? natac_occ_decide<STEP, 16, true> : natac_occ_decide<STEP, 4, true>)
                               : (f.rn16 ? natac_occ_decide<STEP, 16, false> : natac_occ_decide<STEP, 4, false>);
    const bool heavy_first = b->ctx->occ_ordered;
    hipLaunchKernelGGL(kd, dim3((b->n_tiles_occ + (heavy_first ? HEAVY_CAP : 0) + 3) / 4), dim3(256), f.lds_od, b->ctx->stream, f.ct, b->d_tiles_occ,
                       b->n_tiles_occ, b->d_ranges_occ, heavy_first ? b->d_order_occ : nullptr, f.of, b->d_blk_off, b->total_blocks, b->d_gsum, b->d_grid[0], b->d_grid[1], b->d_grid[2], b->d_defer,
                       b->d_defer + 1);
}

extern "C" {

int natac_abi_version(void) { return NATAC_ABI_VERSION; }
const char *natac_last_error(void) { return g_err.c_str(); }

int natac_device_count(int *count) {
    if (!count) return fail(NATAC_E_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(NATAC_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return NATAC_OK;
}

int natac_ctx_create(int device_id, natac_ctx **out) {
    if (!out) return fail(NATAC_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(NATAC_E_ARG, "device %d out of range (%d devices)", device_id, n);
    HIPCHK(hipSetDevice(device_id));
    natac_ctx *c = new natac_ctx();
    c->device = device_id;
    HIPCHK(hipGetDeviceProperties(&c->prop, device_id));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // one stream for both stages.  Measured on MI355X: a second stream for the occ stage does not overlap with the nuc stage on its
    // own (the background kernel's workgroups keep every CU's LDS full); co-scheduling them on purpose (a persistent half-occupancy
    // background launch next to the occupancy kernels, round 4) overlapped exactly as designed and was not faster -- the fp64
    // kernels are clock-limited.  The experiment lives in DESIGN.md section 3.3c and profiles/r4/, not in the library.
    HIPCHK(hipEventCreate(&c->t0));
    HIPCHK(hipEventCreate(&c->t1));
    {   // NATAC_BG_DIRECT=1 selects the direct-summation background kernel (validation / A-B timing of the FFT path)
        const char *e = getenv("NATAC_BG_DIRECT");
        c->bg_direct = e && e[0] == '1';
        e = getenv("NATAC_BG_EXT");
        c->bg_ext = !(e && e[0] == '0');
        e = getenv("NATAC_OCC_GENERAL");   // NATAC_OCC_GENERAL=1: every tile through natac_occ_mle (validation / A-B timing)
        c->occ_force_general = e && e[0] == '1';
        e = getenv("NATAC_OCC_ORDER");     // NATAC_OCC_ORDER=0: natac_occ_decide visits its tiles in chunk order (A-B timing of the ordering)
        c->occ_ordered = !(e && e[0] == '0');
    }
    *out = c;
    return NATAC_OK;
}

void natac_ctx_destroy(natac_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    prof_collect(c);
    dev_free(c->d_vmat); dev_free(c->d_vmat_pad); dev_free(c->d_srow); dev_free(c->d_sizes); dev_free(c->d_lrt);
    dev_free(c->d_nucp); dev_free(c->d_nfrp); dev_free(c->d_alphas);
    dev_free(c->d_win_nuc); dev_free(c->d_win_occ); dev_free(c->d_wb_occ);
    dev_free(c->d_fft_tw); dev_free(c->d_fft_k); dev_free(c->d_fft_mtab); dev_free(c->d_fft_swt);
    dev_free(c->d_occ_q4); dev_free(c->d_occ_rho);
    dev_free(c->d_p10); dev_free(c->d_crc);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->ck_stream) (void)hipStreamDestroy(c->ck_stream);
    if (c->ck_start) (void)hipEventDestroy(c->ck_start);
    if (c->d_ck) (void)hipFree(c->d_ck);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int natac_ctx_sync(natac_ctx *c) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    prof_collect(c);
    return NATAC_OK;
}

int natac_ctx_device_info(natac_ctx *c, char *name, size_t name_len, int *n_cu, size_t *mem_bytes) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    }
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = c->prop.totalGlobalMem;
    return NATAC_OK;
}

int natac_ctx_device_ids(natac_ctx *c, int *hip_device, char *pci_bus_id, size_t pci_len) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    if (hip_device) *hip_device = c->device;
    if (pci_bus_id && pci_len) {
        pci_bus_id[0] = 0;
        HIPCHK(hipDeviceGetPCIBusId(pci_bus_id, (int)pci_len, c->device));
    }
    return NATAC_OK;
}

int natac_set_vmat(natac_ctx *c, const double *mat, int lower, int upper, int w) {
    if (!c || !mat) return fail(NATAC_E_ARG, "null argument");
    if (lower < 0 || upper <= lower || w < 0) return fail(NATAC_E_ARG, "bad vmat geometry lower=%d upper=%d w=%d", lower, upper, w);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    dev_free(c->d_vmat);
    c->d_vmat = nullptr;
    c->vlower = lower; c->vupper = upper; c->vw = w; c->R = upper - lower; c->W = 2 * w + 1;
    int rc = dev_upload(c, &c->d_vmat, mat, (size_t)c->R * c->W);
    if (rc) return rc;
    {   // the same rows between VPAD zero columns (natac_frag_gather)
        const int WP = c->W + 2 * VPAD;
        std::vector<double> padded((size_t)c->R * WP, 0.0);
        for (int r = 0; r < c->R; ++r) std::copy(mat + (size_t)r * c->W, mat + (size_t)(r + 1) * c->W, padded.begin() + (size_t)r * WP + VPAD);
        dev_free(c->d_vmat_pad);
        c->d_vmat_pad = nullptr;
        if ((rc = dev_upload(c, &c->d_vmat_pad, padded.data(), padded.size()))) return rc;
        HIPCHK(sync_all(c));   // `padded` is a local
    }
    c->vmat_zero = false;
    for (size_t i = 0; i < (size_t)c->R * c->W; ++i) if (mat[i] == 0.0) { c->vmat_zero = true; break; }
    HIPCHK(sync_all(c));
    c->have_vmat = true;
    c->srow_dirty = true;
    c->fft_dirty = true;
    ++c->model_gen;
    return NATAC_OK;
}

int natac_set_sizes(natac_ctx *c, const double *sizes, int upper) {
    if (!c || !sizes || upper <= 0) return fail(NATAC_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    dev_free(c->d_sizes);
    c->d_sizes = nullptr;
    int rc = dev_upload(c, &c->d_sizes, sizes, (size_t)upper);
    if (rc) return rc;
    c->h_sizes.assign(sizes, sizes + upper);
    HIPCHK(sync_all(c));
    c->sizes_upper = upper;
    c->have_sizes = true;
    c->srow_dirty = true;
    c->fft_dirty = true;             // the template spectra carry the size weights
    ++c->model_gen;
    return NATAC_OK;
}

int natac_set_occ_model(natac_ctx *c, const double *nuc_probs, const double *nfr_probs, int upper, const double *alphas,
                        int n_alpha, double cutoff, int step, int flank) {
    if (!c || !nuc_probs || !nfr_probs || !alphas) return fail(NATAC_E_ARG, "null argument");
    if (upper < 2 || n_alpha < 1 || n_alpha > 128) return fail(NATAC_E_ARG, "need upper >= 2 and 1 <= n_alpha <= 128");
    if (upper > 256) return fail(NATAC_E_ARG, "occupancy kernel supports upper <= 256 (got %d)", upper);
    if (step < 1 || flank < 0) return fail(NATAC_E_ARG, "bad step/flank");
    if (step % 2 == 0) step -= 1; /* Occupancy.py:190-191 */
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    dev_free(c->d_nucp); dev_free(c->d_nfrp); dev_free(c->d_alphas);
    c->d_nucp = c->d_nfrp = c->d_alphas = nullptr;
    int rc;
    if ((rc = dev_upload(c, &c->d_nucp, nuc_probs, (size_t)upper))) return rc;
    if ((rc = dev_upload(c, &c->d_nfrp, nfr_probs, (size_t)upper))) return rc;
    if ((rc = dev_upload(c, &c->d_alphas, alphas, (size_t)n_alpha))) return rc;
    HIPCHK(sync_all(c));
    {
        int zf = 0;
        double pmin = 1.0;
        bool odd = false;   // negative / non-finite probabilities: always take the exact per-element path
        for (int j = 0; j < upper; ++j) {
            const double a = nuc_probs[j], f = nfr_probs[j];
            if (a == 0.0) zf |= 1;
            if (f == 0.0) zf |= 2;
            if (a == 0.0 && f == 0.0) zf |= 4;
            if (!(a >= 0.0) || !(f >= 0.0) || std::isinf(a) || std::isinf(f)) odd = true;
            if (a > 0.0 && a < pmin) pmin = a;
            if (f > 0.0 && f < pmin) pmin = f;
        }
        c->occ_zero_flags = zf;
        // p * b >= DBL_MIN needs b >= DBL_MIN / pmin; 2^-1000 leaves room for any positive double pmin >= 2^-22
        c->occ_b_floor = odd ? std::numeric_limits<double>::infinity() : std::ldexp(1.0, -1000) / pmin;
    }
    c->occ_upper = upper; c->n_alpha = n_alpha; c->cutoff = cutoff; c->step = step;
    c->halfstep = (step - 1) / 2; c->flank = flank;
    {   // fast path (natac_occ_fast.hpp): up to 101 increasing alphas in [0, 1], no insert size with probability 0 under both models, an odd
        // step up to 9 (kernel instantiations; the CLI's --step), a window of at least one step (any --flank: a whole number of
        // step-blocks + the first (2 flank + 1) % step bases of the next one), and a probability range that keeps four likelihood
        // factors inside the fp64 range
        dev_free(c->d_occ_q4); dev_free(c->d_occ_rho);
        c->d_occ_q4 = c->d_occ_rho = nullptr;
        bool ok = n_alpha <= OD_NA && step <= 9 && 2 * flank + 1 >= step;
        for (int a = 0; ok && a < n_alpha; ++a) ok = alphas[a] >= 0.0 && alphas[a] <= 1.0 && (a == 0 || alphas[a] > alphas[a - 1]);
        double rmin = std::numeric_limits<double>::infinity(), rmax = 0.0, pmin = 1.0;
        bool anynuc = false;
        std::vector<double> rho((size_t)upper, 0.0);
        bool zero_nfr = false;
        for (int j = 0; ok && j < upper; ++j) {
            const double a = nuc_probs[j], f = nfr_probs[j];
            ok = f >= 0.0 && std::isfinite(f) && a >= 0.0 && std::isfinite(a) && (f > 0.0 || a > 0.0);   // 0 under both: natac_occ_mle
            if (!ok) break;
            if (f == 0.0) {       // a fragment of this size can only be nucleosomal: factor alpha (occ_eval, ZF)
                rho[j] = std::numeric_limits<double>::infinity();
                zero_nfr = true;
                anynuc = true;
                pmin = std::min(pmin, a);
                continue;
            }
            rho[j] = a / f;
            ok = std::isfinite(rho[j]);
            if (a > 0.0) { anynuc = true; rmin = std::min(rmin, rho[j]); rmax = std::max(rmax, rho[j]); pmin = std::min(pmin, a); }
            pmin = std::min(pmin, f);
        }
        if (rmax == 0.0) { rmin = 1.0; rmax = 1.0; }      // every nucleosomal size has nfr_prob 0: no finite ratio to bound
        ok = ok && anynuc && rmax <= rmin * std::ldexp(1.0, 200) && pmin >= std::ldexp(1.0, -200);
        // a zero probability excludes alpha = 1 (nuc) or alpha = 0 (nfr) for every window (Occupancy.py:112-114); with an alpha strictly
        // between them the likelihood is positive somewhere, so the all-(-inf) case (the reference raises) cannot reach these kernels
        if ((zero_nfr || (c->occ_zero_flags & 1)) && n_alpha < 3) ok = false;
        c->occ_zero_nfr = ok && zero_nfr;
        c->occ_fast_ok = ok;
        // a likelihood factor 1 + alpha (rho kappa - 1) lies in [1 - alpha, 1 + rmax / rmin] with 1 - alpha >= the grid's
        // smallest positive value (or exactly 0 at alpha = 1): 16 of them between two renormalisations stay far inside the
        // fp64 range when rmax / rmin < 2^50 and that smallest step is > 2^-50
        {
            double amin = 1.0;
            for (int a = 0; ok && a < n_alpha; ++a) {
                if (1 - alphas[a] > 0) amin = std::min(amin, 1 - alphas[a]);
                if (zero_nfr && alphas[a] > 0) amin = std::min(amin, alphas[a]);       // the factor alpha of a zero-nfr size
            }
            c->occ_rn16 = ok && rmax <= rmin * std::ldexp(1.0, 50) && amin >= std::ldexp(1.0, -50);
        }
        if (ok) {
            c->occ_nm = (upper + 1) / 2;
            std::vector<double> q4((size_t)4 * c->occ_nm, 0.0);
            for (int j = 0; j < upper; ++j) {
                q4[(size_t)4 * (j >> 1) + (j & 1)] = nuc_probs[j];
                q4[(size_t)4 * (j >> 1) + 2 + (j & 1)] = nfr_probs[j];
            }
            if ((rc = dev_upload(c, &c->d_occ_q4, q4.data(), q4.size()))) return rc;
            if ((rc = dev_upload(c, &c->d_occ_rho, rho.data(), rho.size()))) return rc;
            HIPCHK(sync_all(c));
        }
    }
    c->have_occ = true;
    return NATAC_OK;
}

static int ensure_srow(natac_ctx *c) {
    if (!c->have_vmat) return fail(NATAC_E_STATE, "natac_set_vmat has not been called");
    if (!c->have_sizes) return fail(NATAC_E_STATE, "natac_set_sizes has not been called");
    if (c->sizes_upper < c->vupper) return fail(NATAC_E_ARG, "sizes cover [0,%d) but vmat.upper is %d", c->sizes_upper, c->vupper);
    if (!c->srow_dirty) return NATAC_OK;
    dev_free(c->d_srow);
    c->d_srow = nullptr;
    int rc = dev_alloc(&c->d_srow, (size_t)c->R);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->d_srow, c->d_sizes + c->vlower, (size_t)c->R * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->srow_zero = false;
    for (int r = 0; r < c->R; ++r) if (c->h_sizes[(size_t)c->vlower + r] == 0.0) c->srow_zero = true;
    c->srow_dirty = false;
    return NATAC_OK;
}

// FFT background path applies when the valid part of a 512-point tile is still most of it and the product row has no
// single-cell special case (i == 1, handled by the generic kernel)
static bool fft_bg_applicable(const natac_ctx *c) {
    if (c->bg_direct || c->W > 192 || c->vlower < 2) return false;
    return natac::bg_fft_lds_bytes(c->vupper) <= 64 * 1024;
}

// Extended tiles (natac_fft_bg.hpp): FFT_EXT more outputs on each side of a tile, finished by an edge pass at the end of the tile's wave.  An extended
// tile costs ~10 % more than a plain one (edge pass 6-7 %, its own longer epilogue 3 %: profiles/r5; the rule prices it at 11), so a chunk gets the cheapest
// mix of n tiles of which the first k are extended, 100 n + 11 k smallest with 392 n + 32 k >= L: 2,120 bases take 5 extended tiles
// instead of 6 plain ones, 2,000 bases 5 tiles of which 2 are extended, 10,120 bases stay at 26 plain tiles (25 would need 10 extended
// ones).  The choice depends on the chunk's length alone -- results do not depend on the batch a chunk is in.
static bool bg_ext_possible(const natac_ctx *c) {
    // the edge pass's scratch (partial outputs, reduction scratch, the rows' weights) lives in the transposes' LDS region
    return c->bg_ext && c->W >= 2 * natac::FFT_EXT &&
           4 * natac::FFT_EXT + natac::EDGE_SCRATCH + 4 * ((c->R + 3) / 4) <= 2 * natac::FFT_LA;
}
static void bg_chunk_tiling(int L, int TV, bool ext_ok, int *n_tiles, int *n_extended) {
    const int E2 = 2 * natac::FFT_EXT;
    const int n_std = (L + TV - 1) / TV;
    int best_n = n_std, best_k = 0;
    long long best = 100LL * n_std;
    for (int n = n_std - 1; ext_ok && n >= 1 && (long long)n * (TV + E2) >= L; --n) {
        const int k = (int)((L - (long long)n * TV + E2 - 1) / E2);
        const long long cost = 100LL * n + 11LL * k;
        if (cost < best) { best = cost; best_n = n; best_k = k; }
    }
    *n_tiles = best_n;
    *n_extended = best_k;
}
int natac_bg_tiling(natac_ctx *c, int32_t chunk_len, int32_t *n_tiles, int32_t *extended) {
    if (!c || chunk_len <= 0) return fail(NATAC_E_ARG, "natac_bg_tiling: context and a positive chunk length");
    if (!c->d_vmat) return fail(NATAC_E_STATE, "natac_bg_tiling: set the V-plot first");
    int nt = 0, ex = 0;
    if (fft_bg_applicable(c)) bg_chunk_tiling(chunk_len, natac::FFT_N - c->W + 1, bg_ext_possible(c), &nt, &ex);
    if (n_tiles) *n_tiles = nt;
    if (extended) *extended = ex;
    return NATAC_OK;
}

static int build_tiles_bg(natac_batch *b, int TV, bool ext_ok) {
    std::vector<int2> tiles;
    tiles.reserve((size_t)(b->total_bp / TV) + b->nc);
    const int TVX = TV + 2 * natac::FFT_EXT;
    for (int i = 0; i < b->nc; ++i) {
        int nt, k;
        bg_chunk_tiling(b->h_len[i], TV, ext_ok, &nt, &k);
        int x = 0;
        for (int t = 0; t < nt; ++t) {
            if (t < k) {       // outputs [x, x + TVX): the transform's exact ones start at x + FFT_EXT
                tiles.push_back(make_int2(i, (x + natac::FFT_EXT) | natac::FFT_EXT_BIT));
                x += TVX;
            } else {
                tiles.push_back(make_int2(i, x));
                x += TV;
            }
        }
    }
    dev_free(b->d_tiles_bg);
    b->d_tiles_bg = nullptr;
    b->n_tiles_bg = (int)tiles.size();
    int rc = dev_upload(b->ctx, &b->d_tiles_bg, tiles.data(), tiles.size());
    if (rc) return rc;
    HIPCHK(sync_all(b->ctx));  // `tiles` is a local
    return NATAC_OK;
}

static int ensure_fft(natac_ctx *c) {
    int rc;
    if (!c->d_fft_tw) {
        std::vector<double> tw(2 * (size_t)natac::FFT_N);
        for (int k = 0; k < natac::FFT_N; ++k) {
            const double a = 2.0 * M_PI * k / natac::FFT_N;
            tw[2 * k] = std::cos(a);
            tw[2 * k + 1] = -std::sin(a);
        }
        // fault injection for the test suite's own sensitivity check (tests/test_gpu_tight_tier.py): NATAC_FAULT_TWIDDLE="k:delta" adds delta
        // to cos(2 pi k / 512).  The parity tests must FAIL with k = 37, delta = 1e-9 -- a tolerance that lets this through is not a test.
        if (const char *e = getenv("NATAC_FAULT_TWIDDLE")) {
            int k = 0; double dlt = 0.0;
            if (sscanf(e, "%d:%lf", &k, &dlt) == 2 && k >= 0 && k < natac::FFT_N) tw[2 * (size_t)k] += dlt;
        }
        if ((rc = dev_upload(c, &c->d_fft_tw, tw.data(), tw.size()))) return rc;
        HIPCHK(sync_all(c));
    }
    if (!c->fft_dirty) return NATAC_OK;
    HIPCHK(sync_all(c));
    dev_free(c->d_fft_k);
    c->d_fft_k = nullptr;
    const int npair = (c->R + 1) / 2;
    if ((rc = dev_alloc(&c->d_fft_k, (size_t)npair * 2 * natac::FFT_N))) return rc;
    hipLaunchKernelGGL(natac_fft_template, dim3(npair), dim3(64), 0, c->stream, c->d_vmat, c->d_srow, c->R, c->W, c->d_fft_tw, c->d_fft_k);
    if (c->W >= natac::FFT_EXT) {
        dev_free(c->d_fft_mtab); dev_free(c->d_fft_swt);
        c->d_fft_mtab = c->d_fft_swt = nullptr;
        const int NJ = (c->R + 3) / 4;
        if ((rc = dev_alloc(&c->d_fft_mtab, (size_t)2 * NJ * 64)) || (rc = dev_alloc(&c->d_fft_swt, (size_t)4 * NJ))) return rc;
        hipLaunchKernelGGL(natac_fft_edge_table_mfma, dim3((2 * NJ * 64 + 255) / 256), dim3(256), 0, c->stream, c->d_vmat, c->d_srow, c->R, c->W, NJ,
                           c->d_fft_mtab, c->d_fft_swt);
    }
    HIPCHK(hipGetLastError());
    c->fft_dirty = false;
    return NATAC_OK;
}

static int ensure_window(natac_ctx *c, double **slot, int *slotM, double *slotsd, int M, double sd, double *slotsum = nullptr) {
    if (*slot && *slotM == M && *slotsd == sd) return NATAC_OK;
    HIPCHK(sync_all(c));
    dev_free(*slot);
    *slot = nullptr;
    // scipy.signal.gaussian(M, sd): exp(-0.5 (n/sd)^2), n = arange(M) - (M-1)/2   (pyatac/utils.py:40)
    std::vector<double> w((size_t)M);
    for (int i = 0; i < M; ++i) {
        double n = (double)i - (M - 1) / 2.0;
        double q = n / sd;
        w[i] = std::exp(-0.5 * (q * q));
    }
    int rc = dev_upload(c, slot, w.data(), (size_t)M);
    if (rc) return rc;
    HIPCHK(sync_all(c));
    if (slotsum) {
        double acc = 0.0;
        for (int i = 0; i < M; ++i) acc = acc + w[i];     // same order as the kernel's fma(w, 1, den) chain
        *slotsum = acc;
    }
    *slotM = M;
    *slotsd = sd;
    return NATAC_OK;
}

// E = exp(bias) of the batch on `st`: part of the work of the stage that needs it.  natac_run_nuc always computes it;
// natac_run_occ reuses the array natac_run_nuc left since the last natac_run_occ (once per nuc + occ pass), else computes it.
static int run_exp_bias(natac_batch *b, hipStream_t st, bool from_occ) {
    if (!b->d_bias) return NATAC_OK;
    if (from_occ && b->ebias_fresh) { b->ebias_fresh = false; return NATAC_OK; }
    b->ebias_fresh = !from_occ;
    int rc;
    if (!b->d_ebias && (rc = dev_alloc(&b->d_ebias, (size_t)b->nb))) return rc;
    const int blocks = (int)std::min<long long>((b->nb + 255) / 256, 16384);
    hipLaunchKernelGGL(natac_exp_bias, dim3(blocks), dim3(256), 0, st, b->d_bias, b->d_ebias, b->nb);
    return NATAC_OK;
}

static ChunkTable make_table(natac_batch *b) {
    ChunkTable t;
    t.nc = b->nc; t.chunk_len = b->d_len; t.frag_off = b->d_frag_off; t.lpos = b->d_lpos; t.ilen = b->d_ilen;
    t.centre = b->d_centre; t.bias_off = b->d_bias_off; t.bias = b->d_bias; t.ebias = b->d_ebias; t.bias_left = b->bias_left;
    t.bias_right = b->bias_right; t.out_off = b->d_out_off; t.grid_off = b->d_grid_off;
    return t;
}
static VMatDev make_vmat(natac_ctx *c) {
    VMatDev v;
    v.mat = c->d_vmat; v.matp = c->d_vmat_pad; v.srow = c->d_srow; v.lower = c->vlower; v.upper = c->vupper; v.w = c->vw; v.R = c->R; v.W = c->W;
    v.has_zero = (c->vmat_zero || c->srow_zero) ? 1 : 0;
    v.lrt = nullptr;
    return v;
}
static OccModelDev make_occ(natac_ctx *c) {
    OccModelDev o;
    o.nuc_probs = c->d_nucp; o.nfr_probs = c->d_nfrp; o.alphas = c->d_alphas; o.upper = c->occ_upper;
    o.n_alpha = c->n_alpha; o.step = c->step; o.halfstep = c->halfstep; o.flank = c->flank; o.cutoff = c->cutoff;
    o.ci_factor = std::exp(-0.5 * c->cutoff);
    o.zero_flags = c->occ_zero_flags;
    o.b_floor = c->occ_b_floor;
    return o;
}

static int build_tiles(natac_batch *b, int width, int2 **d_out, int *n_out, bool grid_units = false, int step = 1, int half = 0) {
    std::vector<int2> tiles;
    tiles.reserve((size_t)(b->total_bp / std::max(1, width)) + b->nc);
    for (int i = 0; i < b->nc; ++i) {
        int n = b->h_len[i];
        if (grid_units) n = (b->h_len[i] > half) ? (b->h_len[i] - half + step - 1) / step : 0;
        for (int x = 0; x < n; x += width) tiles.push_back(make_int2(i, x));
    }
    dev_free(*d_out);
    *d_out = nullptr;
    *n_out = (int)tiles.size();
    int rc = dev_upload(b->ctx, d_out, tiles.data(), tiles.size());
    if (rc) return rc;
    HIPCHK(sync_all(b->ctx));  // `tiles` is a local
    return NATAC_OK;
}

int natac_batch_create(natac_ctx *c, int32_t nc, const int32_t *chunk_len, const int64_t *frag_off, const int32_t *frag_lpos,
                       const int32_t *frag_ilen, const int64_t *bias_off, const double *bias_log, int32_t bias_left,
                       int32_t bias_right, natac_batch **out) {
    if (!c || !out || !chunk_len || !frag_off) return fail(NATAC_E_ARG, "null argument");
    *out = nullptr;
    if (nc <= 0) return fail(NATAC_E_ARG, "n_chunks must be positive");
    if (bias_log && !bias_off) return fail(NATAC_E_ARG, "bias_log without bias_off");
    if (frag_off[0] != 0) return fail(NATAC_E_ARG, "frag_off[0] must be 0");
    const long long nf = frag_off[nc];
    if (nf > 0 && (!frag_lpos || !frag_ilen)) return fail(NATAC_E_ARG, "fragment arrays are NULL");
    HIPCHK(hipSetDevice(c->device));
    natac_batch *b = new natac_batch();
    b->ctx = c; b->nc = nc; b->nf = nf; b->bias_left = bias_left; b->bias_right = bias_right;
    b->h_len.assign(chunk_len, chunk_len + nc);
    b->h_out_off.resize((size_t)nc + 1);
    b->h_out_off[0] = 0;
    for (int i = 0; i < nc; ++i) {
        if (chunk_len[i] <= 0) { delete b; return fail(NATAC_E_ARG, "chunk %d has length %d", i, chunk_len[i]); }
        if (frag_off[i + 1] < frag_off[i] || frag_off[i + 1] - frag_off[i] > 0x7fffffffLL) {
            delete b; return fail(NATAC_E_ARG, "frag_off not monotone (chunk %d)", i);
        }
        if (bias_off && bias_off[i + 1] - bias_off[i] != (long long)chunk_len[i] + bias_left + bias_right) {
            delete b; return fail(NATAC_E_ARG, "bias slice of chunk %d must cover [start-%d, end+%d)", i, bias_left, bias_right);
        }
        b->h_out_off[i + 1] = b->h_out_off[i] + chunk_len[i];
    }
    b->total_bp = b->h_out_off[nc];
    b->nb = bias_off ? bias_off[nc] : 0;
    int rc = NATAC_OK;
#define TRY(x) do { if ((rc = (x)) != NATAC_OK) { natac_batch_free(b); return rc; } } while (0)
    TRY(dev_upload(c, &b->d_len, chunk_len, (size_t)nc));
    TRY(dev_upload(c, (long long **)&b->d_frag_off, (const long long *)frag_off, (size_t)nc + 1));
    TRY(dev_upload(c, &b->d_lpos, frag_lpos, (size_t)nf));
    TRY(dev_upload(c, &b->d_ilen, frag_ilen, (size_t)nf));
    TRY(dev_alloc(&b->d_centre, (size_t)nf));
    TRY(dev_upload(c, &b->d_out_off, b->h_out_off.data(), (size_t)nc + 1));
    if (bias_off) {
        TRY(dev_upload(c, (long long **)&b->d_bias_off, (const long long *)bias_off, (size_t)nc + 1));
        if (bias_log) TRY(dev_upload(c, &b->d_bias, bias_log, (size_t)b->nb));
        else TRY(dev_alloc(&b->d_bias, (size_t)b->nb));           // filled on the device (natac_batch_create_from_seq)
    }
    TRY(dev_alloc(&b->d_status, (size_t)nc));
    hipError_t e = hipMemsetAsync(b->d_status, 0, (size_t)nc * sizeof(int), c->stream);
    if (e != hipSuccess) { natac_batch_free(b); return fail(NATAC_E_HIP, "memset: %s", hipGetErrorString(e)); }
    if (nf > 0) {
        int blocks = (int)std::min<long long>((nf + 255) / 256, 4096);
        hipLaunchKernelGGL(natac_frag_centres, dim3(blocks), dim3(256), 0, c->stream, b->d_lpos, b->d_ilen, b->d_centre, nf);
    }
    TRY(build_tiles(b, 256, &b->d_tiles256, &b->n_tiles256));
    {
        std::vector<long long> first((size_t)nc + 1, 0);
        for (int i = 0; i < nc; ++i) first[(size_t)i + 1] = first[i] + (chunk_len[i] + 255) / 256;
        TRY(dev_upload(c, &b->d_tile256_first, first.data(), first.size()));
    }
    e = hipStreamSynchronize(c->stream);  // host buffers may be released by the caller after return
    if (e != hipSuccess) { natac_batch_free(b); return fail(NATAC_E_HIP, "upload: %s", hipGetErrorString(e)); }
#undef TRY
    *out = b;
    return NATAC_OK;
}

int natac_batch_create_from_seq(natac_ctx *c, int32_t nc, const int32_t *chunk_len, const int64_t *frag_off, const int32_t *frag_lpos,
                                const int32_t *frag_ilen, const int64_t *seq_off, const uint8_t *seq, const double *log_pwm,
                                const uint8_t *nucleotides, int nrow, int K, int32_t bias_left, int32_t bias_right, natac_batch **out) {
    if (!c || !out || !chunk_len || !seq_off || !seq || !log_pwm || !nucleotides) return fail(NATAC_E_ARG, "null argument");
    if (nc <= 0 || nrow < 1 || K < 1) return fail(NATAC_E_ARG, "bad argument");
    std::vector<int64_t> boff((size_t)nc + 1, 0);
    for (int i = 0; i < nc; ++i) {
        const int64_t nb = (int64_t)chunk_len[i] + bias_left + bias_right;
        if (seq_off[i + 1] - seq_off[i] != nb + K - 1)
            return fail(NATAC_E_ARG, "sequence window of chunk %d must hold %lld bases ([start-%d-up, end+%d+down))", i, (long long)(nb + K - 1),
                        bias_left, bias_right);
        boff[(size_t)i + 1] = boff[(size_t)i] + nb;
    }
    int rc = natac_batch_create(c, nc, chunk_len, frag_off, frag_lpos, frag_ilen, boff.data(), nullptr, bias_left, bias_right, out);
    if (rc) return rc;
    natac_batch *b = *out;
    unsigned char *d_s = nullptr, *d_n = nullptr;
    long long *d_so = nullptr;
    double *d_p = nullptr;
    const size_t nseq = (size_t)seq_off[nc];
    if ((rc = dev_upload(c, &d_s, (const unsigned char *)seq, nseq)) == NATAC_OK &&
        (rc = dev_upload(c, (long long **)&d_so, (const long long *)seq_off, (size_t)nc + 1)) == NATAC_OK &&
        (rc = dev_upload(c, &d_n, (const unsigned char *)nucleotides, (size_t)nrow)) == NATAC_OK &&
        (rc = dev_upload(c, &d_p, log_pwm, (size_t)nrow * K)) == NATAC_OK) {
        int maxlen = 0;
        for (int i = 0; i < nc; ++i) maxlen = std::max(maxlen, chunk_len[i] + bias_left + bias_right);
        const unsigned gx = (unsigned)std::min(16, (maxlen + 1023) / 1024);
        hipLaunchKernelGGL(natac_pwm_score_chunks, dim3(gx, (unsigned)nc), dim3(256), 0, c->stream, d_s, d_so, b->d_bias_off, d_p, d_n, nrow, K,
                           b->d_bias);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(NATAC_E_HIP, "pwm score: %s", hipGetErrorString(e));
    }
    dev_free(d_s); dev_free(d_so); dev_free(d_n); dev_free(d_p);
    if (rc) { natac_batch_free(b); *out = nullptr; }
    return rc;
}

void natac_batch_free(natac_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)sync_all(b->ctx);
    prof_collect(b->ctx);
    dev_free(b->d_len); dev_free(b->d_lpos); dev_free(b->d_ilen); dev_free(b->d_centre); dev_free(b->d_status);
    dev_free(b->d_frag_off); dev_free(b->d_bias_off); dev_free(b->d_out_off); dev_free(b->d_grid_off); dev_free(b->d_bias);
    dev_free(b->d_ebias);
    dev_free(b->d_occ_minkey); dev_free(b->d_occ_nan); dev_free(b->d_tiles_os); dev_free(b->d_tiles1k);
    dev_free(b->d_tiles256); dev_free(b->d_tiles_bg); dev_free(b->d_tiles_occ); dev_free(b->d_ranges_occ); dev_free(b->d_ranges256);
    dev_free(b->d_tile256_first);
    dev_free(b->d_order_occ);
    dev_free(b->d_jitter); dev_free(b->d_pk_out); dev_free(b->d_cap_off);
    dev_free(b->d_pk_offs); dev_free(b->d_slot); dev_free(b->d_pk_count); dev_free(b->d_pk_chunk); dev_free(b->d_pk_pos);
    dev_free(b->d_pk_big);
    for (int i = 0; i < NATAC_T_COUNT; ++i) dev_free(b->d_track[i]);
    dev_free(b->d_bnum); dev_free(b->d_bcov); dev_free(b->d_fmt_out);
    (void)fmt_pending_wait(b);
    dev_free(b->d_blk_off); dev_free(b->d_gsum); dev_free(b->d_tiles_gs); dev_free(b->d_defer);
    dev_free(b->d_opk_vals); dev_free(b->d_nuc_dist); dev_free(b->d_opk_keep);
    for (int i = 0; i < 3; ++i) dev_free(b->d_grid[i]);
    delete b;
}

int natac_batch_release_outputs(natac_batch *b) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    prof_collect(c);
    for (int i = 0; i < NATAC_T_COUNT; ++i) { dev_free(b->d_track[i]); b->d_track[i] = nullptr; }
    for (int i = 0; i < 3; ++i) { dev_free(b->d_grid[i]); b->d_grid[i] = nullptr; }
    dev_free(b->d_bnum); dev_free(b->d_bcov); dev_free(b->d_gsum); dev_free(b->d_pk_out);
    dev_free(b->d_pk_chunk); dev_free(b->d_pk_pos); dev_free(b->d_opk_vals); dev_free(b->d_opk_keep); dev_free(b->d_nuc_dist);
    b->d_bnum = b->d_bcov = b->d_gsum = b->d_pk_out = b->d_opk_vals = b->d_nuc_dist = nullptr;
    b->d_pk_chunk = b->d_pk_pos = b->d_opk_keep = nullptr;
    dev_free(b->d_fmt_out); b->d_fmt_out = nullptr; b->fmt_bytes = -1;
    (void)fmt_pending_wait(b);
    b->pk_cap = 0; b->pk_n = -1; b->opk_cap = 0; b->opk_n = -1;
    b->nuc_done = b->occ_done = b->ins_done = b->cov_from_nuc = b->prefill_valid = b->bg_valid = false;
    // the per-chunk status words describe the outputs that were just dropped
    HIPCHK(hipMemsetAsync(b->d_status, 0, (size_t)b->nc * sizeof(int), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NATAC_OK;     // offset / tile tables stay: the next natac_run_* only re-allocates the arrays
}

int natac_batch_info(natac_batch *b, int64_t *total_bp, int64_t *total_grid, int64_t *n_frags) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    if (total_bp) *total_bp = b->total_bp;
    if (total_grid) *total_grid = b->total_grid;
    if (n_frags) *n_frags = b->nf;
    return NATAC_OK;
}

static int ensure_track(natac_batch *b, int t) {
    if (b->d_track[t]) return NATAC_OK;
    return dev_alloc(&b->d_track[t], (size_t)b->total_bp);  // INS uses the first half of a double slot (int32)
}

int natac_run_nuc(natac_batch *b, double smooth_sd) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_srow(c);
    if (rc) return rc;
    if (!(smooth_sd > 0)) return fail(NATAC_E_ARG, "smooth_sd must be positive");
    const int M = 6 * (int)smooth_sd + 1;  /* NucleosomeCalling.py:276 (integer sd from the cli) */
    const int need_l = c->vw + ((c->vupper - 2) >> 1), need_r = c->vw + ((c->vupper - 1) >> 1) + 1;
    if (b->d_bias && (b->bias_left < need_l || b->bias_right < need_r))
        return fail(NATAC_E_ARG, "bias track must extend >= %d left / %d right of the chunk (has %d / %d)", need_l, need_r,
                    b->bias_left, b->bias_right);
    for (int i = 0; i < b->nc; ++i)
        if (b->h_len[i] < M) return fail(NATAC_E_ARG, "chunk %d shorter (%d) than the smoothing window (%d)", i, b->h_len[i], M);
    if ((rc = ensure_window(c, &c->d_win_nuc, &c->win_nuc_M, &c->win_nuc_sd, M, smooth_sd, &c->win_nuc_sum))) return rc;
    const bool use_fft = fft_bg_applicable(c);
    // The background track itself is an output only with --write_all (run_nuc.py:22-39: _nucHelper returns it, run_nuc writes it only
    // then); the FFT kernel leaves its two factors per base (bnum, bcov: the candidates read them) and T_BACKGROUND is formed from them
    // on the first request (materialise_bg) -- the same expression, the same bits, 8 bytes per base less written per step.
    for (int t : {NATAC_T_NUC_COV, NATAC_T_NFR_COV, NATAC_T_RAW, NATAC_T_NORM, NATAC_T_SMOOTH})
        if ((rc = ensure_track(b, t))) return rc;
    if (!use_fft && (rc = ensure_track(b, NATAC_T_BACKGROUND))) return rc;
    if (!b->d_bnum && (rc = dev_alloc(&b->d_bnum, (size_t)b->total_bp))) return rc;
    if (!b->d_bcov && (rc = dev_alloc(&b->d_bcov, (size_t)b->total_bp))) return rc;
    const bool fast = !use_fft && (c->W == 121 && c->vlower >= 2);
    if (use_fft) {
        if ((rc = ensure_fft(c))) return rc;
        const int TV = FFT_N - c->W + 1;
        const bool ext_ok = bg_ext_possible(c);
        const int key = -(TV + (ext_ok ? 4096 : 0));   // the tiling this batch's table was built for
        if (b->bgG != key) {
            if ((rc = build_tiles_bg(b, TV, ext_ok))) return rc;
            b->bgG = key;
        }
    } else if (fast) {
        const int G = choose_bg_G(b, c->W);
        if (G != b->bgG) {
            if ((rc = build_tiles(b, 64 * G, &b->d_tiles_bg, &b->n_tiles_bg))) return rc;
            b->bgG = G;
        }
    }
    if (!b->d_ebias && b->d_bias && (rc = dev_alloc(&b->d_ebias, (size_t)b->nb))) return rc;
    const ChunkTable ct = make_table(b);
    const VMatDev vm = make_vmat(c);
    natac_ctx::Ev ev;
    if (!b->d_ranges256 && (rc = dev_alloc(&b->d_ranges256, (size_t)b->n_tiles256))) return rc;
    if ((M - 1) / 2 == 30 && !b->d_tiles1k && (rc = build_tiles(b, 1024, &b->d_tiles1k, &b->n_tiles1k))) return rc;
    prof_begin(c, NATAC_K_FRAG_GATHER, ev);
    if (b->ranges256_w != c->vw) {
        hipLaunchKernelGGL(natac_tile_ranges256, dim3((b->n_tiles256 + 255) / 256), dim3(256), 0, c->stream, ct, b->d_tiles256,
                           b->n_tiles256, c->vw, b->d_ranges256);
        b->ranges256_w = c->vw;
    }
    // OccChunk.getCov = nuc_cov + nfr_cov when the occupancy model's window / size range are the V-plot's: written here too
    const bool cov_too = c->have_occ && c->flank == c->vw && c->occ_upper == c->vupper;
    if (cov_too && (rc = ensure_track(b, NATAC_T_OCC_COV))) return rc;
    constexpr int GNBL = NATAC_GNBL;      // adjacent bases per lane: 2.  Round 5, lanes outside a window reading one shared zero line:
                                          // 0.63 ms per 20 k chunks against 0.79 with 4 and 0.85 with 1 (round 2: 3.5 / 4.5 / 3.8 ms per step)
    hipLaunchKernelGGL((natac_frag_gather<GNBL>), dim3(b->n_tiles256), dim3(256 / GNBL), 0, c->stream, ct, b->d_tiles256, b->d_ranges256, vm,
                       b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_NFR_COV], b->d_track[NATAC_T_RAW],
                       cov_too ? b->d_track[NATAC_T_OCC_COV] : nullptr);
    b->cov_from_nuc = cov_too;
    prof_end(c, ev);
    prof_begin(c, NATAC_K_BACKGROUND, ev);
    if ((rc = run_exp_bias(b, c->stream, false))) return rc;
    if (use_fft) {
        const size_t lds = bg_fft_lds_bytes(vm.upper);
        hipLaunchKernelGGL(natac_background_fft, dim3(b->n_tiles_bg), dim3(64), lds, c->stream, ct, b->d_tiles_bg, vm, c->d_fft_tw,
                           c->d_fft_k, b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_RAW], (double *)nullptr,
                           b->d_track[NATAC_T_NORM], b->d_bnum, b->d_bcov, (unsigned)b->n_tiles_bg, c->d_fft_mtab,
                           c->d_fft_swt, (c->R + 3) / 4);
        b->bg_valid = false;
    } else if (fast) {
        switch (b->bgG) {
            case 7: launch_bg<7>(b, ct, vm); break;
            case 9: launch_bg<9>(b, ct, vm); break;
            case 13: launch_bg<13>(b, ct, vm); break;
            default: launch_bg<17>(b, ct, vm); break;
        }
        b->bg_valid = true;
    } else {
        hipLaunchKernelGGL(natac_background_generic, dim3(b->n_tiles256), dim3(256), 0, c->stream, ct, b->d_tiles256, vm,
                           b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_RAW], b->d_track[NATAC_T_BACKGROUND],
                           b->d_track[NATAC_T_NORM], b->d_bnum, b->d_bcov);
        b->bg_valid = true;
    }
    prof_end(c, ev);
    prof_begin(c, NATAC_K_SMOOTH_NUC, ev);
    {
        const int h = (M - 1) / 2;
        if (h == 30) {      // the cli's smooth_sd = 10: four bases per lane
            hipLaunchKernelGGL((natac_smooth_same4<true, 30>), dim3(b->n_tiles1k), dim3(256), (size_t)8 * SM4_S(30) * sizeof(double),
                               c->stream, ct, b->d_tiles1k, c->d_win_nuc, c->win_nuc_sum, b->d_track[NATAC_T_NORM],
                               b->d_track[NATAC_T_SMOOTH]);
        } else {
            const size_t lds = ((size_t)2 * (256 + 2 * h)) * sizeof(double);
            hipLaunchKernelGGL((natac_smooth_same<true>), dim3(b->n_tiles256), dim3(256), lds, c->stream, ct, b->d_tiles256,
                               c->d_win_nuc, M, c->win_nuc_sum, b->d_track[NATAC_T_NORM], b->d_track[NATAC_T_SMOOTH]);
        }
    }
    prof_end(c, ev);
    HIPCHK(hipGetLastError());
    b->nuc_done = true;
    b->nuc_gen = c->model_gen;
    b->nuc_w = c->vw;
    b->nuc_upper = c->vupper;
    return NATAC_OK;
}

// natac_occ_smooth (one base per lane, any step / window): smoothed occupancy -> dst_occ (un-filled), bounds -> dst_lo / dst_hi (or null)
static void launch_occ_smooth_generic(natac_batch *b, const ChunkTable &ct, const OccModelDev &om, int M, double *dst_occ, double *dst_lo,
                                      double *dst_hi) {
    natac_ctx *c = b->ctx;
    const int h = (M - 1) / 2;
    const int NB = 2 * ((h + c->step - 1) / c->step) + 2, NG = (255 + 2 * h) / c->step + 3;
    const size_t lds = ((size_t)((M + 1) & ~1) + (size_t)c->step * NB + 3 * (size_t)NG) * sizeof(double);
    hipLaunchKernelGGL(natac_occ_smooth, dim3(b->n_tiles256), dim3(256), lds, c->stream, ct, b->d_tiles256, om, c->d_win_occ, M,
                       b->d_grid[0], b->d_grid[1], b->d_grid[2], dst_occ, dst_lo, dst_hi);
}

// blocks the STEP bases of a step-block see in natac_occ_smooth_blk: ceil(h / step) to the left + their own + (h + step - 1) / step to
// the right; for h a multiple of the step that is 2 h / step + 1, launched with the spare (zero-weight) block round 2 measured with
static int occ_smooth_blocks(int h, int step) {
    if (h % step == 0) return 2 * (h / step) + 2;
    return (h + step - 1) / step + (h + step - 1) / step + 1;
}

// block weights of natac_occ_smooth_blk: wb[bi][j] = sum of the window taps that fall on block bmin + bi for a base at offset j
// of its block -- the table natac_occ_smooth forms per workgroup, same summation order
static int ensure_block_weights(natac_ctx *c, int M, double sd, int NB) {
    if (c->d_wb_occ && c->wb_M == M && c->wb_step == c->step) return NATAC_OK;
    HIPCHK(sync_all(c));
    dev_free(c->d_wb_occ);
    c->d_wb_occ = nullptr;
    const int h = (M - 1) / 2, step = c->step;
    std::vector<double> w((size_t)M), wb((size_t)step * NB + step);     // + the full denominators wb[NB][j] of the clean sweep
    for (int i = 0; i < M; ++i) {           // as ensure_window
        const double n = (double)i - (M - 1) / 2.0, q = n / sd;
        w[i] = std::exp(-0.5 * (q * q));
    }
    for (int j = 0; j < step; ++j)
        for (int bi = 0; bi < NB; ++bi) {
            const int bmin = -((h + step - 1) / step);    // leftmost block any base of a block reaches
            double sacc = 0.0;
            for (int d = 0; d < step; ++d) {
                const int n = j + h - (bmin + bi) * step - d;
                if (n >= 0 && n < M) sacc += w[n];
            }
            wb[(size_t)bi * step + j] = sacc;
        }
    for (int j = 0; j < step; ++j) {        // den[j] of a base with every block present: the kernel's own sequence fma(w, 1, den)
        double den = 0.0;
        for (int bi = 0; bi < NB; ++bi) den = std::fma(wb[(size_t)bi * step + j], 1.0, den);
        wb[(size_t)NB * step + j] = den;
    }
    int rc = dev_upload(c, &c->d_wb_occ, wb.data(), wb.size());
    if (rc) return rc;
    HIPCHK(sync_all(c));
    c->wb_M = M;
    c->wb_step = step;
    return NATAC_OK;
}

// BACKGROUND after the FFT kernel: (bnum * nuc_cov) / bcov per base, the expression of the kernel's own epilogue (natac_fft_bg.hpp)
static int materialise_bg(natac_batch *b) {
    if (b->bg_valid) return NATAC_OK;
    natac_ctx *c = b->ctx;
    int rc = ensure_track(b, NATAC_T_BACKGROUND);
    if (rc) return rc;
    const int blocks = (int)std::min<long long>((b->total_bp + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(natac_bg_from_factors, dim3(blocks), dim3(256), 0, c->stream, b->d_bnum, b->d_track[NATAC_T_NUC_COV], b->d_bcov,
                       b->d_track[NATAC_T_BACKGROUND], b->total_bp);
    HIPCHK(hipGetLastError());
    b->bg_valid = true;
    return NATAC_OK;
}

// OCC_PREFILL (smoothed_vals before call_peaks' NaN fill) is not part of the default pass: written on the first request
static int materialise_prefill(natac_batch *b) {
    if (b->prefill_valid) return NATAC_OK;
    natac_ctx *c = b->ctx;
    int rc = ensure_track(b, NATAC_T_OCC_PREFILL);
    if (rc) return rc;
    const ChunkTable ct = make_table(b);
    const OccModelDev om = make_occ(c);
    launch_occ_smooth_generic(b, ct, om, 2 * c->flank + 1, b->d_track[NATAC_T_OCC_PREFILL], nullptr, nullptr);
    HIPCHK(hipGetLastError());
    b->prefill_valid = true;
    return NATAC_OK;
}

// tracks that are formed on the first request
static int materialise_track(natac_batch *b, int track) {
    if (track == NATAC_T_OCC_PREFILL && b->occ_done) return materialise_prefill(b);
    if (track == NATAC_T_BACKGROUND && b->nuc_done) return materialise_bg(b);
    return NATAC_OK;
}

// natac_run_occ, part 1: every allocation, table upload and host synchronisation of the stage (nothing is launched that
// depends on another stream), so that part 2 can be enqueued behind a running kernel without stalling the host
static int occ_prepare(natac_batch *b) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!c->have_occ) return fail(NATAC_E_STATE, "natac_set_occ_model has not been called");
    const int M = 2 * c->flank + 1;              /* OccupancyParameters.window, Occupancy.py:184 */
    const double sd = c->flank / 3.0;            /* Occupancy.py:220 */
    const int need_l = c->flank + ((c->occ_upper - 2) >> 1), need_r = c->flank + ((c->occ_upper - 1) >> 1) + 1;
    if (b->d_bias && (b->bias_left < need_l || b->bias_right < need_r))
        return fail(NATAC_E_ARG, "bias track must extend >= %d left / %d right of the chunk (has %d / %d)", need_l, need_r,
                    b->bias_left, b->bias_right);
    for (int i = 0; i < b->nc; ++i)
        if (b->h_len[i] < M) return fail(NATAC_E_ARG, "chunk %d shorter (%d) than the occupancy window (%d)", i, b->h_len[i], M);
    int rc;
    if ((rc = ensure_window(c, &c->d_win_occ, &c->win_occ_M, &c->win_occ_sd, M, sd))) return rc;
    if (!b->d_grid_off || b->grid_step != c->step) {
        b->h_grid_off.assign((size_t)b->nc + 1, 0);
        for (int i = 0; i < b->nc; ++i) {
            const int nk = (b->h_len[i] > c->halfstep) ? (b->h_len[i] - c->halfstep + c->step - 1) / c->step : 0;
            b->h_grid_off[i + 1] = b->h_grid_off[i] + nk;
        }
        b->total_grid = b->h_grid_off[b->nc];
        HIPCHK(sync_all(c));
        dev_free(b->d_grid_off);
        b->d_grid_off = nullptr;
        if ((rc = dev_upload(c, &b->d_grid_off, b->h_grid_off.data(), (size_t)b->nc + 1))) return rc;
        for (int i = 0; i < 3; ++i) {
            dev_free(b->d_grid[i]);
            b->d_grid[i] = nullptr;
            if ((rc = dev_alloc(&b->d_grid[i], (size_t)b->total_grid))) return rc;
        }
        if ((rc = build_tiles(b, OCC_T * OCC_NP, &b->d_tiles_occ, &b->n_tiles_occ, true, c->step, c->halfstep))) return rc;
        dev_free(b->d_ranges_occ); dev_free(b->d_order_occ);
        b->d_ranges_occ = nullptr; b->d_order_occ = nullptr;
        if ((rc = dev_alloc(&b->d_ranges_occ, (size_t)b->n_tiles_occ))) return rc;
        if ((rc = dev_alloc(&b->d_order_occ, (size_t)2 + HEAVY_CAP + ((size_t)b->n_tiles_occ + 3) / 4))) return rc;
        b->ranges_occ_key[0] = -1;       // new tile table: ranges not formed yet
        b->gs_Q = -1;                    // ... and the block tables of natac_occ_gsum follow the grid
        b->grid_step = c->step;
        b->grid_half = c->halfstep;
    }
    for (int i = 0; i < 3; ++i)      // released by natac_batch_release_outputs
        if (!b->d_grid[i] && (rc = dev_alloc(&b->d_grid[i], (size_t)b->total_grid))) return rc;
    for (int t : {NATAC_T_OCC, NATAC_T_OCC_LOWER, NATAC_T_OCC_UPPER, NATAC_T_OCC_COV})
        if ((rc = ensure_track(b, t))) return rc;
    b->prefill_valid = false;
    if (!b->d_ebias && b->d_bias && (rc = dev_alloc(&b->d_ebias, (size_t)b->nb))) return rc;
    const bool fast = c->occ_fast_ok && !c->occ_force_general;
    if (fast) {   // per-block sum buffers + tile table of natac_occ_gsum (geometry: step / flank of the model)
        const int Q = (2 * c->flank + 1) / c->step;      // whole step-blocks of a window; the rest is a prefix of the next block
        if (b->gs_Q != Q) {
            HIPCHK(sync_all(c));
            std::vector<long long> bo((size_t)b->nc + 1);
            std::vector<int2> tiles;
            for (int i = 0; i <= b->nc; ++i) bo[i] = b->h_grid_off[i] + (long long)Q * i;
            for (int i = 0; i < b->nc; ++i) {
                const int nblk = (int)(b->h_grid_off[i + 1] - b->h_grid_off[i]) + Q;
                for (int x = 0; x < nblk; x += GS_BLOCKS) tiles.push_back(make_int2(i, x));
            }
            b->total_blocks = bo[b->nc];
            dev_free(b->d_blk_off); dev_free(b->d_gsum); dev_free(b->d_tiles_gs); dev_free(b->d_defer);
            b->d_blk_off = nullptr; b->d_gsum = nullptr; b->d_tiles_gs = nullptr; b->d_defer = nullptr;
            if ((rc = dev_upload(c, &b->d_blk_off, bo.data(), bo.size()))) return rc;
            if ((rc = dev_upload(c, &b->d_tiles_gs, tiles.data(), tiles.size()))) return rc;
            HIPCHK(sync_all(c));
            b->n_tiles_gs = (int)tiles.size();
            if ((rc = dev_alloc(&b->d_defer, (size_t)b->n_tiles_occ + 1))) return rc;
            b->gs_Q = Q;
        }
        if (!b->d_gsum && (rc = dev_alloc(&b->d_gsum, (size_t)4 * b->total_blocks))) return rc;
    }
    // geometry checks + tables of the smoothing pass
    {
        const int U = c->occ_upper, UP = (U + 1) & ~1;
        const int span = (OCC_T * OCC_NP - 1) * c->step + M + c->step;
        const int EW = span + ((U - 2) >> 1) + ((U - 1) >> 1) + 2;
        const int n_ones = ((OCC_T - 1) * c->step + M + c->step + 3) & ~1;
        const size_t lds = ((size_t)((EW + 1) & ~1) + (size_t)OCC_T * UP + 2 * (size_t)UP + OCC_ACL + n_ones) * sizeof(double) +
                           (size_t)2 * OCC_FMAX * sizeof(int);
        if (lds > 64 * 1024)
            return fail(NATAC_E_ARG, "occupancy window / step too large for the device tile (step=%d flank=%d upper=%d)", c->step, c->flank, U);
        if (fast) {
            const int R = c->occ_nm - 1, Q = b->gs_Q;
            const size_t lds_gs = ((size_t)((GS_BLOCKS * c->step + 2 * R + 2 * GS_MG + 1) & ~1) + 16 * 64) * sizeof(double);
            const int NGP = (64 + Q + 1) & ~1;
            const size_t lds_od = (size_t)4 * ((OD_FM + 4) + 4 * NGP + OD_FM / 2) * sizeof(double);
            if (lds_gs > 64 * 1024 || lds_od > 64 * 1024)
                return fail(NATAC_E_ARG, "occupancy window too large for the device tile (flank=%d upper=%d)", c->flank, U);
        }
    }
    const bool blk = c->step <= 9;     // natac_occ_smooth_blk<1, 3, 5, 7, 9> (the step is odd)
    if (blk) {
        const int NB = occ_smooth_blocks((M - 1) / 2, c->step);
        if ((rc = ensure_block_weights(c, M, sd, NB))) return rc;
        if (b->os_width != 256 * c->step) {
            if ((rc = build_tiles(b, 256 * c->step, &b->d_tiles_os, &b->n_tiles_os))) return rc;
            b->os_width = 256 * c->step;
        }
        if (!b->d_occ_minkey && (rc = dev_alloc(&b->d_occ_minkey, (size_t)b->nc))) return rc;
        if (!b->d_occ_nan && (rc = dev_alloc(&b->d_occ_nan, (size_t)b->nc))) return rc;
    } else {
        if ((rc = ensure_track(b, NATAC_T_OCC_PREFILL))) return rc;
    }
    return NATAC_OK;
}

// natac_run_occ, part 2: the launches
static int occ_launch(natac_batch *b) {
    natac_ctx *c = b->ctx;
    const int M = 2 * c->flank + 1;
    int rc;
    b->prefill_valid = false;
    const ChunkTable ct = make_table(b);
    const OccModelDev om = make_occ(c);
    natac_ctx::Ev ev;
    const bool fast = c->occ_fast_ok && !c->occ_force_general;
    prof_begin(c, NATAC_K_OCC_MLE, ev, c->stream);
    {
        const int U = c->occ_upper, UP = (U + 1) & ~1;
        const int span = (OCC_T * OCC_NP - 1) * c->step + M + c->step;
        const int EW = span + ((U - 2) >> 1) + ((U - 1) >> 1) + 2;
        const int n_ones = ((OCC_T - 1) * c->step + M + c->step + 3) & ~1;
        const size_t lds = ((size_t)((EW + 1) & ~1) + (size_t)OCC_T * UP + 2 * (size_t)UP + OCC_ACL + n_ones) * sizeof(double) +
                           (size_t)2 * OCC_FMAX * sizeof(int);
        if (b->ranges_occ_key[0] != c->step || b->ranges_occ_key[1] != c->halfstep || b->ranges_occ_key[2] != c->flank) {
            // an index over the (immutable) fragment list, like the 256-base tiles' ranges: formed once per batch and geometry
            hipLaunchKernelGGL(natac_occ_tile_ranges, dim3((b->n_tiles_occ + 255) / 256), dim3(256), 0, c->stream, ct, b->d_tiles_occ,
                               b->n_tiles_occ, c->step, c->halfstep, c->flank, b->d_ranges_occ);
            // ... and the heavy tiles natac_occ_decide visits first: more than four times the batch's mean fragment count (and > 256)
            const double span = (double)(OCC_T * OCC_NP - 1) * c->step + 2.0 * c->flank + 1.0;
            const int thr = (int)std::max(256.0, 4.0 * (double)b->nf * span / (double)std::max<long long>(1, b->total_bp));
            HIPCHK(hipMemsetAsync(b->d_order_occ, 0, 2 * sizeof(int), c->stream));
            hipLaunchKernelGGL(natac_tile_heavy, dim3((b->n_tiles_occ + 255) / 256), dim3(256), 0, c->stream, b->d_ranges_occ, b->n_tiles_occ, thr,
                               b->d_order_occ, b->d_order_occ + 2, (unsigned char *)(b->d_order_occ + 2 + HEAVY_CAP));
            b->ranges_occ_key[0] = c->step; b->ranges_occ_key[1] = c->halfstep; b->ranges_occ_key[2] = c->flank;
        }
        const int *d_list = nullptr, *d_count = nullptr;
        unsigned grid_general = (unsigned)b->n_tiles_occ;
        if (fast) {
            OccFastDev of;
            of.q4 = c->d_occ_q4; of.rho = c->d_occ_rho; of.alphas = c->d_alphas; of.na = c->n_alpha; of.nm = c->occ_nm; of.upper = U; of.step = c->step;
            of.halfstep = c->halfstep; of.flank = c->flank; of.Q = b->gs_Q; of.flags = (c->occ_zero_flags & 1) | (c->occ_zero_nfr ? 2 : 0);
            of.ci_factor = om.ci_factor; of.e_lo = std::ldexp(1.0, -190); of.e_hi = std::ldexp(1.0, 190);
            const int R = of.nm - 1;
            const size_t lds_gs = ((size_t)((GS_BLOCKS * c->step + 2 * R + 2 * GS_MG + 1) & ~1) + 16 * 64) * sizeof(double);
            const int NGP = (64 + of.Q + 1) & ~1;
            const size_t lds_od = (size_t)4 * ((OD_FM + 4) + 4 * NGP + OD_FM / 2) * sizeof(double);
            HIPCHK(hipMemsetAsync(b->d_defer, 0, sizeof(int), c->stream));
            if ((rc = run_exp_bias(b, c->stream, true))) return rc;
            const OccFastLaunch fl{b, ct, of, lds_gs, lds_od, (2 * c->flank + 1) % c->step, c->occ_rn16};
            switch (c->step) {
                case 1: occ_fast_launch<1>(fl); break;
                case 3: occ_fast_launch<3>(fl); break;
                case 5: occ_fast_launch<5>(fl); break;
                case 7: occ_fast_launch<7>(fl); break;
                default: occ_fast_launch<9>(fl); break;
            }
            d_count = b->d_defer;
            d_list = b->d_defer + 1;
            grid_general = (unsigned)std::min(b->n_tiles_occ, 2048);   // walks the deferred list (normally empty)
        }
        if (c->step == 5 && c->flank == 60 && c->n_alpha <= 16 * OCC_RA)
            hipLaunchKernelGGL((natac_occ_mle<5, 60, 0, 1>), dim3(grid_general), dim3(256), lds, c->stream, ct, b->d_tiles_occ,
                               b->d_ranges_occ, om, b->d_grid[0], b->d_grid[1], b->d_grid[2], b->d_status, d_list, d_count);
        else if (c->step == 5 && c->flank == 60)
            hipLaunchKernelGGL((natac_occ_mle<5, 60, 0>), dim3(grid_general), dim3(256), lds, c->stream, ct, b->d_tiles_occ,
                               b->d_ranges_occ, om, b->d_grid[0], b->d_grid[1], b->d_grid[2], b->d_status, d_list, d_count);
        else
            hipLaunchKernelGGL((natac_occ_mle<0, 0, 0>), dim3(grid_general), dim3(256), lds, c->stream, ct, b->d_tiles_occ,
                               b->d_ranges_occ, om, b->d_grid[0], b->d_grid[1], b->d_grid[2], b->d_status, d_list, d_count);
    }
    prof_end(c, ev);
    prof_begin(c, NATAC_K_OCC_SMOOTH, ev, c->stream);
    const bool blk = c->step <= 9;     // natac_occ_smooth_blk<1, 3, 5, 7, 9> (the step is odd)
    if (blk) {
        const int NB = occ_smooth_blocks((M - 1) / 2, c->step);
        HIPCHK(hipMemsetAsync(b->d_occ_minkey, 0xff, (size_t)b->nc * sizeof(unsigned long long), c->stream));
        HIPCHK(hipMemsetAsync(b->d_occ_nan, 0, (size_t)b->nc * sizeof(int), c->stream));
        const size_t lds_os = ((size_t)3 * (256 + NB) + 256 * c->step) * sizeof(double);
        auto ks = c->step == 1 ? natac_occ_smooth_blk<1> : c->step == 3 ? natac_occ_smooth_blk<3> : c->step == 5 ? natac_occ_smooth_blk<5>
                : c->step == 7 ? natac_occ_smooth_blk<7> : natac_occ_smooth_blk<9>;
        hipLaunchKernelGGL(ks, dim3(b->n_tiles_os), dim3(256), lds_os, c->stream, ct,
                           b->d_tiles_os, om, c->d_win_occ, M, c->d_wb_occ, NB, b->d_grid[0], b->d_grid[1], b->d_grid[2],
                           b->d_track[NATAC_T_OCC], b->d_track[NATAC_T_OCC_LOWER], b->d_track[NATAC_T_OCC_UPPER], b->d_occ_minkey,
                           b->d_occ_nan);
    } else {
        launch_occ_smooth_generic(b, ct, om, M, b->d_track[NATAC_T_OCC_PREFILL], b->d_track[NATAC_T_OCC_LOWER],
                                  b->d_track[NATAC_T_OCC_UPPER]);
        b->prefill_valid = true;
    }
    {
        if (b->nuc_done && b->cov_from_nuc && c->flank == b->nuc_w && c->occ_upper == b->nuc_upper) {
            // natac_frag_gather of natac_run_nuc already wrote OCC_COV = nuc_cov + nfr_cov for this geometry
        } else if (b->nuc_done && c->flank == b->nuc_w && c->occ_upper == b->nuc_upper)
            hipLaunchKernelGGL(natac_add_tracks, dim3(4096), dim3(256), 0, c->stream, b->d_track[NATAC_T_NUC_COV],
                               b->d_track[NATAC_T_NFR_COV], b->d_track[NATAC_T_OCC_COV], b->total_bp);
        else
            hipLaunchKernelGGL(natac_occ_cov, dim3(b->n_tiles256), dim3(256), 0, c->stream, ct, b->d_tiles256, c->occ_upper, c->flank,
                               b->d_track[NATAC_T_OCC_COV]);
    }
    prof_end(c, ev);
    prof_begin(c, NATAC_K_OCC_FILL, ev, c->stream);
    if (blk)     // in place, and only the chunks that hold a NaN
        hipLaunchKernelGGL(natac_fill_nan_chunks, dim3(b->nc), dim3(256), 0, c->stream, ct, b->d_occ_minkey, b->d_occ_nan,
                           b->d_track[NATAC_T_OCC]);
    else
        hipLaunchKernelGGL(natac_fill_nan_min, dim3(b->nc), dim3(256), 0, c->stream, ct, b->d_track[NATAC_T_OCC_PREFILL],
                           b->d_track[NATAC_T_OCC]);
    prof_end(c, ev);
    HIPCHK(hipGetLastError());
    b->occ_done = true;
    return NATAC_OK;
}

int natac_run_occ(natac_batch *b) {
    int rc = occ_prepare(b);
    if (rc) return rc;
    return occ_launch(b);
}

static int ins_launch(natac_batch *b, int lower, int upper);

int natac_run_ins(natac_batch *b, int lower, int upper) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_track(b, NATAC_T_INS);
    if (rc) return rc;
    return ins_launch(b, lower, upper);
}

static int ins_launch(natac_batch *b, int lower, int upper) {
    natac_ctx *c = b->ctx;
    const ChunkTable ct = make_table(b);
    natac_ctx::Ev ev;
    prof_begin(c, NATAC_K_INS, ev, c->stream);
    int maxL = 0;
    for (int i = 0; i < b->nc; ++i) maxL = std::max(maxL, b->h_len[i]);
    if ((size_t)maxL * sizeof(int) <= 60 * 1024) {
        hipLaunchKernelGGL(natac_insertions_lds, dim3(b->nc), dim3(256), (size_t)maxL * sizeof(int), c->stream, ct, lower, upper,
                           (int *)b->d_track[NATAC_T_INS]);
    } else {
        HIPCHK(hipMemsetAsync(b->d_track[NATAC_T_INS], 0, (size_t)b->total_bp * sizeof(int), c->stream));
        hipLaunchKernelGGL(natac_insertions, dim3(b->nc), dim3(256), 0, c->stream, ct, lower, upper, (int *)b->d_track[NATAC_T_INS]);
    }
    prof_end(c, ev);
    HIPCHK(hipGetLastError());
    b->ins_done = true;
    return NATAC_OK;
}

int natac_run_candidates(natac_batch *b, int64_t n_cand, const int32_t *cand_chunk, const int32_t *cand_pos, double *lr,
                         double *var, double *z) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    if (!b->nuc_done) return fail(NATAC_E_STATE, "natac_run_nuc must run before natac_run_candidates");
    if (n_cand < 0 || (n_cand > 0 && (!cand_chunk || !cand_pos || !lr || !var || !z))) return fail(NATAC_E_ARG, "null argument");
    if (n_cand == 0) return NATAC_OK;
    if (n_cand > 0x7fffffffLL) return fail(NATAC_E_ARG, "too many candidates");
    HIPCHK(hipSetDevice(c->device));
    for (int64_t k = 0; k < n_cand; ++k) {
        const int ci = cand_chunk[k];
        if (ci < 0 || ci >= b->nc || cand_pos[k] < 0 || cand_pos[k] >= b->h_len[ci])
            return fail(NATAC_E_ARG, "candidate %lld out of range (chunk %d pos %d)", (long long)k, ci, cand_pos[k]);
    }
    int *d_cc = nullptr, *d_cp = nullptr;
    double *d_out = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_cc, cand_chunk, (size_t)n_cand))) return rc;
    if ((rc = dev_upload(c, &d_cp, cand_pos, (size_t)n_cand))) { dev_free(d_cc); return rc; }
    if ((rc = dev_alloc(&d_out, (size_t)3 * n_cand))) { dev_free(d_cc); dev_free(d_cp); return rc; }
    const ChunkTable ct = make_table(b);
    const VMatDev vm = make_vmat(c);
    natac_ctx::Ev ev;
    prof_begin(c, NATAC_K_CAND, ev);
    launch_candidates(c, ct, vm, d_cc, d_cp, n_cand, b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_NORM],
                      b->nuc_gen == c->model_gen ? b->d_bnum : nullptr, b->nuc_gen == c->model_gen ? b->d_bcov : nullptr, d_out,
                      d_out + n_cand,
                      d_out + 2 * n_cand, b->ranges256_w == c->vw ? b->d_tile256_first : nullptr, b->ranges256_w == c->vw ? b->d_ranges256 : nullptr);
    prof_end(c, ev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(lr, d_out, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(var, d_out + n_cand, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(z, d_out + 2 * n_cand, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_cc); dev_free(d_cp); dev_free(d_out);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "candidates: %s", hipGetErrorString(e));
    prof_collect(c);
    return NATAC_OK;
}

int natac_run_candidates_cov(natac_batch *b, int64_t n_cand, const int32_t *cand_chunk, const int32_t *cand_pos, int mode,
                             double *var) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    natac_ctx *c = b->ctx;
    if (!b->nuc_done) return fail(NATAC_E_STATE, "natac_run_nuc must run before natac_run_candidates_cov");
    if (mode < 0 || mode > 2) return fail(NATAC_E_ARG, "mode must be 0 (closed form), 1 (literal) or 2 (closed form in fp32)");
    if (n_cand < 0 || (n_cand > 0 && (!cand_chunk || !cand_pos || !var))) return fail(NATAC_E_ARG, "null argument");
    if (n_cand == 0) return NATAC_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_srow(c);
    if (rc) return rc;
    for (int64_t k = 0; k < n_cand; ++k) {
        const int ci = cand_chunk[k];
        if (ci < 0 || ci >= b->nc || cand_pos[k] < 0 || cand_pos[k] >= b->h_len[ci])
            return fail(NATAC_E_ARG, "candidate %lld out of range (chunk %d pos %d)", (long long)k, ci, cand_pos[k]);
    }
    const int N = c->R * c->W;
    const int NBLK = 64;                       // row groups of the literal pair sum per candidate
    const int64_t SLAB = 2048;                 // candidates per pass: 2048 x N doubles = 289 MB for the default V-plot
    const ChunkTable ct = make_table(b);
    const VMatDev vm = make_vmat(c);
    const int EW = c->W + ((c->vupper - 2) >> 1) + ((c->vupper - 1) >> 1);
    int *d_cc = nullptr, *d_cp = nullptr;
    double *d_p = nullptr, *d_out = nullptr;
    const int64_t slab = std::min<int64_t>(SLAB, n_cand);
    if ((rc = dev_alloc(&d_cc, (size_t)slab)) || (rc = dev_alloc(&d_cp, (size_t)slab)) || (rc = dev_alloc(&d_p, (size_t)slab * N)) ||
        (rc = dev_alloc(&d_out, (size_t)slab * NBLK))) {
        dev_free(d_cc); dev_free(d_cp); dev_free(d_p); dev_free(d_out);
        return rc;
    }
    std::vector<double> part((size_t)slab * NBLK);
    hipError_t e = hipSuccess;
    for (int64_t k0 = 0; k0 < n_cand && e == hipSuccess; k0 += slab) {
        const int64_t m = std::min<int64_t>(slab, n_cand - k0);
        e = hipMemcpyAsync(d_cc, cand_chunk + k0, (size_t)m * sizeof(int), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_cp, cand_pos + k0, (size_t)m * sizeof(int), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(natac_cand_window_probs, dim3((unsigned)m), dim3(256), (size_t)(EW + 2) * sizeof(double), c->stream, ct, vm,
                           d_cc, d_cp, d_p);
        const int per = mode == 1 ? NBLK : 1;
        if (mode == 1)
            hipLaunchKernelGGL(natac_cov_literal_many, dim3(NBLK, (unsigned)m), dim3(256), 0, c->stream, d_p, c->d_vmat, N, d_out);
        else if (mode == 0)
            hipLaunchKernelGGL((natac_cov_closed_many<double>), dim3((unsigned)m), dim3(256), 0, c->stream, d_p, c->d_vmat, N, d_out);
        else
            hipLaunchKernelGGL((natac_cov_closed_many<float>), dim3((unsigned)m), dim3(256), 0, c->stream, d_p, c->d_vmat, N, d_out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_out, (size_t)m * per * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) break;
        for (int64_t k = 0; k < m; ++k) {
            double s = 0;
            for (int j = 0; j < per; ++j) s += part[(size_t)k * per + j];
            var[k0 + k] = s;
        }
    }
    dev_free(d_cc); dev_free(d_cp); dev_free(d_p); dev_free(d_out);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "candidates_cov: %s", hipGetErrorString(e));
    // r = int(nuc_cov[pos]) like the .pyx's `int r` (NucleosomeCalling.py:125): one gather of the coverage values
    std::vector<double> reads((size_t)n_cand);
    {
        std::vector<long long> idx((size_t)n_cand);
        for (int64_t k = 0; k < n_cand; ++k) idx[(size_t)k] = b->h_out_off[cand_chunk[k]] + cand_pos[k];
        long long *d_idx = nullptr;
        double *d_r = nullptr;
        if ((rc = dev_upload(c, &d_idx, idx.data(), (size_t)n_cand))) return rc;
        if ((rc = dev_alloc(&d_r, (size_t)n_cand))) { dev_free(d_idx); return rc; }
        hipLaunchKernelGGL(natac_gather_f64, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, c->stream,
                           b->d_track[NATAC_T_NUC_COV], d_idx, (long long)n_cand, d_r);
        e = hipMemcpyAsync(reads.data(), d_r, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        dev_free(d_idx); dev_free(d_r);
        if (e != hipSuccess) return fail(NATAC_E_HIP, "candidates_cov: %s", hipGetErrorString(e));
    }
    for (int64_t k = 0; k < n_cand; ++k) var[k] = var[k] * (double)(int)reads[(size_t)k];
    return NATAC_OK;
}

static int run_peaks_impl(natac_batch *b, const double *sig_a, const double *sig_b, bool with_stats, double min_signal, int sep,
                          int boundary, int order, const double *jitter, int64_t n_jitter, int64_t *n_cand) {
    if (!b || !jitter || !n_cand) return fail(NATAC_E_ARG, "null argument");
    natac_ctx *c = b->ctx;
    if (order < 1 || order > 255 || sep < 1 || boundary < 0) return fail(NATAC_E_ARG, "bad peak parameters");
    int maxL = 0;
    for (int i = 0; i < b->nc; ++i) maxL = std::max(maxL, b->h_len[i]);
    if (n_jitter < maxL) return fail(NATAC_E_ARG, "jitter stream (%lld) shorter than the longest chunk (%d)", (long long)n_jitter, maxL);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    int rc;
    if (!b->d_jitter || b->n_jitter < maxL) {
        dev_free(b->d_jitter);
        b->d_jitter = nullptr;
        if ((rc = dev_upload(c, &b->d_jitter, jitter, (size_t)maxL))) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
        b->n_jitter = maxL;
    } else {
        HIPCHK(hipMemcpyAsync(b->d_jitter, jitter, (size_t)maxL * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (b->pk_order != order) {   // slot regions: a chunk of L bases holds at most L/(order+1) + 1 maxima
        std::vector<long long> cap((size_t)b->nc + 1, 0);
        for (int i = 0; i < b->nc; ++i) cap[i + 1] = cap[i] + b->h_len[i] / (order + 1) + 2;
        b->slot_total = cap[b->nc];
        dev_free(b->d_cap_off); dev_free(b->d_slot);
        b->d_cap_off = nullptr; b->d_slot = nullptr;
        if ((rc = dev_upload(c, &b->d_cap_off, cap.data(), (size_t)b->nc + 1))) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
        if ((rc = dev_alloc(&b->d_slot, (size_t)b->slot_total))) return rc;
        b->pk_order = order;
    }
    if (!b->d_pk_count && (rc = dev_alloc(&b->d_pk_count, (size_t)b->nc))) return rc;
    if (!b->d_pk_offs && (rc = dev_alloc(&b->d_pk_offs, (size_t)b->nc + 1))) return rc;
    const ChunkTable ct = make_table(b);
    const double *norm = sig_a, *sm = sig_b;
    natac_ctx::Ev ev;
    prof_begin(c, NATAC_K_CAND, ev);
    {   // one workgroup per chunk: LDS = the jittered signal (or a segment of it) + the chunk's list of maxima
        const int pk_cap = (std::min(PEAK_MAX, maxL / (order + 1) + 2) + 7) & ~7;
        const size_t lds_lists = (size_t)pk_cap * (sizeof(double) + sizeof(int) + 1);
        const int bt = maxL <= 4096 ? 256 : 1024;               // threads per workgroup of the register variant
        const int nj = (maxL + bt - 1) / bt;
        const size_t lds_reg = (size_t)((bt * nj + 2 * order + 1) & ~1) * sizeof(double) + lds_lists;
#define NATAC_PEAKS_REG(NJ, BT)                                                                                                    \
    case NJ: {                                                                                                                     \
        auto kfn = natac_peaks_chunk_reg<NJ, BT>;                                                                                  \
        if (lds_reg > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg)); \
        hipLaunchKernelGGL(kfn, dim3(b->nc), dim3(BT), lds_reg, c->stream, ct, norm, sm, b->d_jitter, min_signal, boundary, order, sep, \
                           pk_cap, b->d_cap_off, b->d_slot, b->d_pk_count, b->d_status);                                          \
        break;                                                                                                                     \
    }
        if (bt == 256) {
            switch (nj) {
                NATAC_PEAKS_REG(1, 256) NATAC_PEAKS_REG(2, 256) NATAC_PEAKS_REG(3, 256) NATAC_PEAKS_REG(4, 256) NATAC_PEAKS_REG(5, 256)
                NATAC_PEAKS_REG(6, 256) NATAC_PEAKS_REG(7, 256) NATAC_PEAKS_REG(8, 256) NATAC_PEAKS_REG(9, 256) NATAC_PEAKS_REG(10, 256)
                NATAC_PEAKS_REG(11, 256) NATAC_PEAKS_REG(12, 256) NATAC_PEAKS_REG(13, 256) NATAC_PEAKS_REG(14, 256)
                NATAC_PEAKS_REG(15, 256) NATAC_PEAKS_REG(16, 256)
            }
        } else if (nj <= 16 && lds_reg <= 150 * 1024) {        // chunks up to 16,384 bases: the whole chunk in LDS, 16 waves
            switch (nj) {
                NATAC_PEAKS_REG(5, 1024) NATAC_PEAKS_REG(6, 1024) NATAC_PEAKS_REG(7, 1024) NATAC_PEAKS_REG(8, 1024) NATAC_PEAKS_REG(9, 1024)
                NATAC_PEAKS_REG(10, 1024) NATAC_PEAKS_REG(11, 1024) NATAC_PEAKS_REG(12, 1024) NATAC_PEAKS_REG(13, 1024)
                NATAC_PEAKS_REG(14, 1024) NATAC_PEAKS_REG(15, 1024) NATAC_PEAKS_REG(16, 1024)
            }
        } else {   // longer still: segments of 4,096 bases, signal read per segment
            const int seg = 4096;
            const size_t lds = (size_t)((seg + 2 * order + 1) & ~1) * sizeof(double) + lds_lists;
            double *big_sig = nullptr;
            int *big_pos = nullptr;
            unsigned char *big_state = nullptr;
            if (maxL / (order + 1) + 2 > pk_cap) {      // some chunk can hold more maxima than the LDS lists: global lists for those
                if (b->pk_big_slots < b->slot_total) {
                    dev_free(b->d_pk_big);
                    b->d_pk_big = nullptr;
                    b->pk_big_slots = 0;
                    // per slot: sig (8 bytes) + pos (4) + state (1), laid out as three arrays in one block
                    if ((rc = dev_alloc(&b->d_pk_big, (size_t)b->slot_total * 2 + 2))) return rc;
                    b->pk_big_slots = b->slot_total;
                }
                big_sig = b->d_pk_big;
                big_pos = (int *)(b->d_pk_big + b->slot_total);
                big_state = (unsigned char *)(big_pos + b->slot_total);
            }
            hipLaunchKernelGGL(natac_peaks_chunk, dim3(b->nc), dim3(256), lds, c->stream, ct, norm, sm, b->d_jitter, min_signal,
                               boundary, order, sep, seg, pk_cap, b->d_cap_off, b->d_slot, b->d_pk_count, b->d_status, big_sig, big_pos,
                               big_state);
        }
#undef NATAC_PEAKS_REG
    }
    hipLaunchKernelGGL(natac_scan_counts, dim3(1), dim3(1024), 0, c->stream, b->d_pk_count, b->nc, b->d_pk_offs);
    long long total = 0;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&total, b->d_pk_offs + b->nc, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (total > b->pk_cap) {
        dev_free(b->d_pk_chunk); dev_free(b->d_pk_pos); dev_free(b->d_pk_out);
        b->d_pk_chunk = b->d_pk_pos = nullptr; b->d_pk_out = nullptr;
        b->pk_cap = total + total / 4 + 16;
        if ((rc = dev_alloc(&b->d_pk_chunk, (size_t)b->pk_cap))) return rc;
        if ((rc = dev_alloc(&b->d_pk_pos, (size_t)b->pk_cap))) return rc;
        if ((rc = dev_alloc(&b->d_pk_out, (size_t)3 * b->pk_cap))) return rc;
    }
    if (total > 0) {
        hipLaunchKernelGGL(natac_compact_candidates, dim3((b->nc + 3) / 4), dim3(256), 0, c->stream, b->nc, b->d_pk_count, b->d_pk_offs,
                           b->d_cap_off, b->d_slot, b->d_pk_chunk, b->d_pk_pos);
        if (with_stats) {
            const VMatDev vm = make_vmat(c);
            launch_candidates(c, ct, vm, b->d_pk_chunk, b->d_pk_pos, total, b->d_track[NATAC_T_NUC_COV], b->d_track[NATAC_T_NORM],
                              b->nuc_gen == c->model_gen ? b->d_bnum : nullptr, b->nuc_gen == c->model_gen ? b->d_bcov : nullptr,
                              b->d_pk_out, b->d_pk_out + b->pk_cap, b->d_pk_out + 2 * b->pk_cap,
                              b->ranges256_w == c->vw ? b->d_tile256_first : nullptr, b->ranges256_w == c->vw ? b->d_ranges256 : nullptr);
        }
    }
    b->pk_has_stats = with_stats;
    prof_end(c, ev);
    HIPCHK(hipGetLastError());
    b->pk_n = total;
    *n_cand = total;
    return NATAC_OK;
}

int natac_run_peaks(natac_batch *b, double min_signal, int sep, int boundary, int order, const double *jitter, int64_t n_jitter,
                    int64_t *n_cand) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    if (!b->nuc_done) return fail(NATAC_E_STATE, "natac_run_nuc must run before natac_run_peaks");
    return run_peaks_impl(b, b->d_track[NATAC_T_NORM], b->d_track[NATAC_T_SMOOTH], true, min_signal, sep, boundary, order, jitter,
                          n_jitter, n_cand);
}

int natac_run_track_peaks(natac_batch *b, int track, double min_signal, int sep, int boundary, int order, const double *jitter,
                          int64_t n_jitter, int64_t *n_peaks) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    if (track == NATAC_T_INS) return fail(NATAC_E_ARG, "peak search needs a float64 track");
    int rc = track_ready(b, track);
    if (rc) return rc;
    HIPCHK(hipSetDevice(b->ctx->device));
    if ((rc = materialise_track(b, track))) return rc;
    return run_peaks_impl(b, b->d_track[track], nullptr, false, min_signal, sep, boundary, order, jitter, n_jitter, n_peaks);
}

int natac_run_occ_peaks(natac_batch *b, double min_occ, int sep, const double *jitter, int64_t n_jitter, int64_t *n_peaks) {
    if (!b || !n_peaks) return fail(NATAC_E_ARG, "null argument");
    if (!b->occ_done) return fail(NATAC_E_STATE, "natac_run_occ must run before natac_run_occ_peaks");
    natac_ctx *c = b->ctx;
    const int U = c->occ_upper;
    if (U > 1024) return fail(NATAC_E_ARG, "upper > 1024");
    // OccChunk.callPeaks: call_peaks(smoothed_vals, sep, min_signal = min_occ), boundary = sep / 2, order = 1 (Occupancy.py:227)
    int rc = run_peaks_impl(b, b->d_track[NATAC_T_OCC], nullptr, false, min_occ, sep, sep / 2, 1, jitter, n_jitter, n_peaks);
    if (rc) return rc;
    const long long n = *n_peaks;
    if (n > b->opk_cap) {
        HIPCHK(sync_all(c));
        dev_free(b->d_opk_vals); dev_free(b->d_opk_keep);
        b->d_opk_vals = nullptr; b->d_opk_keep = nullptr;
        b->opk_cap = n + n / 4 + 16;
        if ((rc = dev_alloc(&b->d_opk_vals, (size_t)4 * b->opk_cap))) return rc;
        if ((rc = dev_alloc(&b->d_opk_keep, (size_t)b->opk_cap))) return rc;
    }
    if (!b->d_opk_vals) {
        b->opk_cap = 16;
        if ((rc = dev_alloc(&b->d_opk_vals, (size_t)4 * b->opk_cap))) return rc;
        if ((rc = dev_alloc(&b->d_opk_keep, (size_t)b->opk_cap))) return rc;
    }
    if (!b->d_nuc_dist || b->nd_upper != U) {
        HIPCHK(sync_all(c));
        dev_free(b->d_nuc_dist);
        b->d_nuc_dist = nullptr;
        if ((rc = dev_alloc(&b->d_nuc_dist, (size_t)b->nc * U))) return rc;
        b->nd_upper = U;
    }
    const ChunkTable ct = make_table(b);
    natac_ctx::Ev ev;
    prof_begin(c, NATAC_K_CAND, ev);
    hipLaunchKernelGGL(natac_occ_peak_dist, dim3(b->nc), dim3(256), (size_t)U * sizeof(int), c->stream, ct, b->d_pk_offs, b->d_pk_pos,
                       b->d_track[NATAC_T_OCC], b->d_track[NATAC_T_OCC_LOWER], b->d_track[NATAC_T_OCC_UPPER],
                       b->d_track[NATAC_T_OCC_COV], min_occ, c->flank, U, b->opk_cap, b->d_opk_vals, b->d_opk_keep, b->d_nuc_dist);
    prof_end(c, ev);
    HIPCHK(hipGetLastError());
    b->opk_n = n;
    return NATAC_OK;
}

int natac_download_occ_peaks(natac_batch *b, int64_t n, int32_t *chunk, int32_t *pos, double *occ, double *lower, double *upper,
                             double *reads, int32_t *keep) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    if (b->opk_n < 0 || b->opk_n != b->pk_n) return fail(NATAC_E_STATE, "natac_run_occ_peaks has not run (or another peak search ran since)");
    if (n != b->opk_n) return fail(NATAC_E_ARG, "expected %lld peaks, got buffers for %lld", b->opk_n, (long long)n);
    if (n == 0) return NATAC_OK;
    if (!chunk || !pos || !occ || !lower || !upper || !reads || !keep) return fail(NATAC_E_ARG, "null argument");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t nb = (size_t)n * sizeof(double);
    HIPCHK(hipMemcpyAsync(chunk, b->d_pk_chunk, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pos, b->d_pk_pos, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(occ, b->d_opk_vals, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(lower, b->d_opk_vals + b->opk_cap, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(upper, b->d_opk_vals + 2 * b->opk_cap, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(reads, b->d_opk_vals + 3 * b->opk_cap, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(keep, b->d_opk_keep, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    return NATAC_OK;
}

int natac_download_nuc_dist(natac_batch *b, double *dst, size_t dst_bytes) {
    if (!b || !dst) return fail(NATAC_E_ARG, "null argument");
    if (b->opk_n < 0 || !b->d_nuc_dist) return fail(NATAC_E_STATE, "natac_run_occ_peaks has not run");
    const size_t need = (size_t)b->nc * b->nd_upper * sizeof(double);
    if (dst_bytes != need) return fail(NATAC_E_ARG, "destination holds %zu bytes, nuc_dist needs %zu", dst_bytes, need);
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipMemcpyAsync(dst, b->d_nuc_dist, need, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    return NATAC_OK;
}

int natac_download_peaks(natac_batch *b, int64_t n, int32_t *cand_chunk, int32_t *cand_pos, double *lr, double *var, double *z) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    if (b->pk_n < 0) return fail(NATAC_E_STATE, "natac_run_peaks has not run");
    if (n != b->pk_n) return fail(NATAC_E_ARG, "expected %lld candidates, got buffers for %lld", b->pk_n, (long long)n);
    if (n == 0) return NATAC_OK;
    if (!cand_chunk || !cand_pos) return fail(NATAC_E_ARG, "null argument");
    if ((lr || var || z) && !b->pk_has_stats) return fail(NATAC_E_STATE, "the last peak search computed no candidate statistics");
    if (b->pk_has_stats && (!lr || !var || !z)) return fail(NATAC_E_ARG, "null argument");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(cand_chunk, b->d_pk_chunk, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(cand_pos, b->d_pk_pos, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (b->pk_has_stats) {
        HIPCHK(hipMemcpyAsync(lr, b->d_pk_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(var, b->d_pk_out + b->pk_cap, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(z, b->d_pk_out + 2 * b->pk_cap, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    return NATAC_OK;
}

static int track_ready(natac_batch *b, int t) {
    if (t < 0 || t >= NATAC_T_COUNT) return fail(NATAC_E_ARG, "bad track id %d", t);
    const bool nuc = (t <= NATAC_T_SMOOTH), occ = (t >= NATAC_T_OCC && t <= NATAC_T_OCC_COV) || t == NATAC_T_OCC_PREFILL;
    if ((nuc && !b->nuc_done) || (occ && !b->occ_done) || (t == NATAC_T_INS && !b->ins_done))
        return fail(NATAC_E_STATE, "track %d has not been computed yet", t);
    return NATAC_OK;
}

int natac_batch_download(natac_batch *b, int track, void *dst, size_t dst_bytes) {
    if (!b || !dst) return fail(NATAC_E_ARG, "null argument");
    int rc = track_ready(b, track);
    if (rc) return rc;
    const size_t need = (size_t)b->total_bp * (track == NATAC_T_INS ? sizeof(int) : sizeof(double));
    if (dst_bytes != need) return fail(NATAC_E_ARG, "destination holds %zu bytes, track needs %zu", dst_bytes, need);
    HIPCHK(hipSetDevice(b->ctx->device));
    if ((rc = materialise_track(b, track))) return rc;
    HIPCHK(hipMemcpyAsync(dst, b->d_track[track], need, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(sync_all(b->ctx));
    prof_collect(b->ctx);
    return NATAC_OK;
}

int natac_batch_download_grid(natac_batch *b, int which, double *dst, size_t dst_bytes) {
    if (!b || !dst) return fail(NATAC_E_ARG, "null argument");
    if (which < 0 || which > 2) return fail(NATAC_E_ARG, "bad grid id");
    if (!b->occ_done) return fail(NATAC_E_STATE, "natac_run_occ has not run");
    const size_t need = (size_t)b->total_grid * sizeof(double);
    if (dst_bytes != need) return fail(NATAC_E_ARG, "destination holds %zu bytes, grid needs %zu", dst_bytes, need);
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipMemcpyAsync(dst, b->d_grid[which], need, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(sync_all(b->ctx));
    return NATAC_OK;
}

int natac_batch_set_track(natac_batch *b, int track, const double *vals, size_t n) {
    if (!b || !vals) return fail(NATAC_E_ARG, "null argument");
    if (track < 0 || track >= NATAC_T_COUNT || track == NATAC_T_INS) return fail(NATAC_E_ARG, "bad track id %d (float64 tracks only)", track);
    if (n != (size_t)b->total_bp) return fail(NATAC_E_ARG, "track needs %lld values, got %zu", b->total_bp, n);
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    int rc = ensure_track(b, track);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(b->d_track[track], vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (track <= NATAC_T_SMOOTH) { b->nuc_done = true; if (track == NATAC_T_BACKGROUND) b->bg_valid = true; }   // the stage flags only gate downloads / the writer
    else { b->occ_done = true; if (track == NATAC_T_OCC_PREFILL) b->prefill_valid = true; }
    return NATAC_OK;
}

int natac_batch_status(natac_batch *b, int32_t *dst, size_t dst_bytes) {
    if (!b || !dst) return fail(NATAC_E_ARG, "null argument");
    if (dst_bytes != (size_t)b->nc * sizeof(int)) return fail(NATAC_E_ARG, "status buffer must hold n_chunks int32");
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipMemcpyAsync(dst, b->d_status, dst_bytes, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(sync_all(b->ctx));
    return NATAC_OK;
}

int natac_batch_track_ptr(natac_batch *b, int track, void **dptr) {
    if (!b || !dptr) return fail(NATAC_E_ARG, "null argument");
    if (track < 0 || track >= NATAC_T_COUNT) return fail(NATAC_E_ARG, "bad track id");
    if ((track == NATAC_T_OCC_PREFILL && b->occ_done) || (track == NATAC_T_BACKGROUND && b->nuc_done)) {
        HIPCHK(hipSetDevice(b->ctx->device));
        int rc = materialise_track(b, track);
        if (rc) return rc;
    }
    *dptr = b->d_track[track];
    return NATAC_OK;
}

/* ---------------- Cython-function drop-ins ---------------- */

int natac_make_fragment_mat(natac_ctx *c, int64_t nf, const int64_t *l, const int32_t *n, int64_t start, int64_t end, int lower,
                            int upper, double *mat) {
    if (!c || !mat || (nf > 0 && (!l || !n))) return fail(NATAC_E_ARG, "null argument");
    if (end <= start || upper <= lower) return fail(NATAC_E_ARG, "empty matrix");
    const long long ncol = end - start, nrow = upper - lower;
    if (ncol > 0x7fffffffLL) return fail(NATAC_E_ARG, "region too long");
    HIPCHK(hipSetDevice(c->device));
    long long *d_l = nullptr; int *d_n = nullptr; double *d_m = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_l, (const long long *)l, (size_t)nf))) return rc;
    if ((rc = dev_upload(c, &d_n, n, (size_t)nf))) { dev_free(d_l); return rc; }
    if ((rc = dev_alloc(&d_m, (size_t)(nrow * ncol)))) { dev_free(d_l); dev_free(d_n); return rc; }
    hipError_t e = hipMemsetAsync(d_m, 0, (size_t)(nrow * ncol) * sizeof(double), c->stream);
    if (e == hipSuccess && nf > 0) {
        int blocks = (int)std::min<long long>((nf + 255) / 256, 4096);
        hipLaunchKernelGGL(natac_fragment_mat, dim3(blocks), dim3(256), 0, c->stream, d_l, d_n, (long long)nf, (long long)start,
                           (int)ncol, lower, (int)nrow, d_m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mat, d_m, (size_t)(nrow * ncol) * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_l); dev_free(d_n); dev_free(d_m);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "make_fragment_mat: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_get_insertions(natac_ctx *c, int64_t nf, const int64_t *l, const int32_t *n, int64_t start, int64_t end, int lower,
                         int upper, double *out) {
    if (!c || !out || (nf > 0 && (!l || !n))) return fail(NATAC_E_ARG, "null argument");
    if (end <= start) return fail(NATAC_E_ARG, "empty region");
    const long long npos = end - start;
    if (npos > 0x7fffffffLL) return fail(NATAC_E_ARG, "region too long");
    HIPCHK(hipSetDevice(c->device));
    long long *d_l = nullptr; int *d_n = nullptr, *d_i = nullptr; double *d_o = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_l, (const long long *)l, (size_t)nf))) return rc;
    if ((rc = dev_upload(c, &d_n, n, (size_t)nf))) { dev_free(d_l); return rc; }
    if ((rc = dev_alloc(&d_i, (size_t)npos))) { dev_free(d_l); dev_free(d_n); return rc; }
    if ((rc = dev_alloc(&d_o, (size_t)npos))) { dev_free(d_l); dev_free(d_n); dev_free(d_i); return rc; }
    hipError_t e = hipMemsetAsync(d_i, 0, (size_t)npos * sizeof(int), c->stream);
    if (e == hipSuccess) {
        if (nf > 0) {
            int blocks = (int)std::min<long long>((nf + 255) / 256, 4096);
            hipLaunchKernelGGL(natac_insertions_region, dim3(blocks), dim3(256), 0, c->stream, d_l, d_n, (long long)nf,
                               (long long)start, (int)npos, lower, upper, d_i);
        }
        int blocks = (int)std::min<long long>((npos + 255) / 256, 4096);
        hipLaunchKernelGGL(natac_i32_to_f64, dim3(blocks), dim3(256), 0, c->stream, d_i, d_o, npos);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)npos * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_l); dev_free(d_n); dev_free(d_i); dev_free(d_o);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "get_insertions: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_get_stranded_insertions(natac_ctx *c, int64_t nf, const int64_t *l, const int32_t *n, int64_t start, int64_t end, int lower,
                                  int upper, double *plus, double *minus) {
    if (!c || !plus || !minus || (nf > 0 && (!l || !n))) return fail(NATAC_E_ARG, "null argument");
    if (end <= start) return fail(NATAC_E_ARG, "empty region");
    const long long npos = end - start;
    if (npos > 0x3fffffffLL) return fail(NATAC_E_ARG, "region too long");
    HIPCHK(hipSetDevice(c->device));
    long long *d_l = nullptr; int *d_n = nullptr, *d_i = nullptr; double *d_o = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_l, (const long long *)l, (size_t)nf))) return rc;
    if ((rc = dev_upload(c, &d_n, n, (size_t)nf))) { dev_free(d_l); return rc; }
    if ((rc = dev_alloc(&d_i, (size_t)2 * npos))) { dev_free(d_l); dev_free(d_n); return rc; }
    if ((rc = dev_alloc(&d_o, (size_t)2 * npos))) { dev_free(d_l); dev_free(d_n); dev_free(d_i); return rc; }
    hipError_t e = hipMemsetAsync(d_i, 0, (size_t)2 * npos * sizeof(int), c->stream);
    if (e == hipSuccess) {
        if (nf > 0) {
            int blocks = (int)std::min<long long>((nf + 255) / 256, 4096);
            hipLaunchKernelGGL(natac_stranded_insertions_region, dim3(blocks), dim3(256), 0, c->stream, d_l, d_n, (long long)nf,
                               (long long)start, (int)npos, lower, upper, d_i, d_i + npos);
        }
        int blocks = (int)std::min<long long>((2 * npos + 255) / 256, 4096);
        hipLaunchKernelGGL(natac_i32_to_f64, dim3(blocks), dim3(256), 0, c->stream, d_i, d_o, 2 * npos);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(plus, d_o, (size_t)npos * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(minus, d_o + npos, (size_t)npos * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_l); dev_free(d_n); dev_free(d_i); dev_free(d_o);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "get_stranded_insertions: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_fragment_sizes(natac_ctx *c, int64_t nf, const int64_t *l, const int32_t *n, int32_t nchunks, const int64_t *cs,
                         const int64_t *ce, int lower, int upper, double *sizes) {
    if (!c || !sizes || (nf > 0 && (!l || !n)) || (nchunks > 0 && (!cs || !ce))) return fail(NATAC_E_ARG, "null argument");
    if (upper <= lower) return fail(NATAC_E_ARG, "upper <= lower");
    if (nf < 0 || nchunks < 0) return fail(NATAC_E_ARG, "negative size");
    const int nb = upper - lower;
    if (nb > (1 << 20)) return fail(NATAC_E_ARG, "size range too wide");
    HIPCHK(hipSetDevice(c->device));
    // chunk starts and ends sorted independently: #{cs <= x} - #{ce <= x} chunks contain x (see natac_size_hist);
    // an inverted interval (end < start) contains nothing, like the reference's `center >= start and center < end`
    std::vector<long long> hs((size_t)nchunks), he((size_t)nchunks);
    for (int k = 0; k < nchunks; ++k) { hs[k] = cs[k]; he[k] = std::max<long long>(ce[k], cs[k]); }
    std::sort(hs.begin(), hs.end());
    std::sort(he.begin(), he.end());
    long long *d_l = nullptr, *d_cs = nullptr, *d_ce = nullptr; int *d_n = nullptr; unsigned long long *d_h = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_l, (const long long *)l, (size_t)nf))) return rc;
    if ((rc = dev_upload(c, &d_n, n, (size_t)nf))) { dev_free(d_l); return rc; }
    if ((rc = dev_upload(c, &d_cs, hs.data(), (size_t)nchunks))) { dev_free(d_l); dev_free(d_n); return rc; }
    if ((rc = dev_upload(c, &d_ce, he.data(), (size_t)nchunks))) { dev_free(d_l); dev_free(d_n); dev_free(d_cs); return rc; }
    if ((rc = dev_alloc(&d_h, (size_t)nb))) { dev_free(d_l); dev_free(d_n); dev_free(d_cs); dev_free(d_ce); return rc; }
    std::vector<unsigned long long> h((size_t)nb, 0);
    hipError_t e = hipMemsetAsync(d_h, 0, (size_t)nb * sizeof(unsigned long long), c->stream);
    natac_ctx::Ev ev;
    prof_begin(c, NATAC_K_SIZE_HIST, ev);
    if (e == hipSuccess && nf > 0 && nchunks > 0) {
        const long long nseg = (nf + SIZE_SEG - 1) / SIZE_SEG;
        const int blocks = (int)std::min<long long>(nseg, 8192);
        const int use_lds = nb <= 8192 ? 1 : 0;
        const size_t lds = (size_t)2 * SIZE_STAGE * sizeof(long long) + (use_lds ? (size_t)nb * sizeof(unsigned) : 0);
        hipLaunchKernelGGL(natac_size_hist, dim3(blocks), dim3(256), lds, c->stream, d_l, d_n, (long long)nf, d_cs, d_ce, (int)nchunks,
                           lower, upper, use_lds, d_h);
        e = hipGetLastError();
    }
    prof_end(c, ev);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_h, (size_t)nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_l); dev_free(d_n); dev_free(d_cs); dev_free(d_ce); dev_free(d_h);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "fragment_sizes: %s", hipGetErrorString(e));
    prof_collect(c);
    for (int i = 0; i < nb; ++i) sizes[i] = (double)h[i];
    return NATAC_OK;
}

int natac_calculate_cov(natac_ctx *c, const double *p, const double *v, int64_t n, int r, int mode, double *out) {
    if (!c || !p || !v || !out) return fail(NATAC_E_ARG, "null argument");
    if (n <= 0) return fail(NATAC_E_ARG, "p and v must be non-empty");
    if (mode != 0 && mode != 1) return fail(NATAC_E_ARG, "mode must be 0 (closed form) or 1 (literal)");
    HIPCHK(hipSetDevice(c->device));
    double *d_p = nullptr, *d_v = nullptr, *d_part = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_p, p, (size_t)n))) return rc;
    if ((rc = dev_upload(c, &d_v, v, (size_t)n))) { dev_free(d_p); return rc; }
    const int blocks = mode == 0 ? (int)std::min<int64_t>((n + 255) / 256, 1024) : (int)std::min<int64_t>(n, 4096);
    if ((rc = dev_alloc(&d_part, (size_t)2 * blocks))) { dev_free(d_p); dev_free(d_v); return rc; }
    std::vector<double> part((size_t)2 * blocks);
    if (mode == 0)
        hipLaunchKernelGGL(natac_cov_closed, dim3(blocks), dim3(256), 0, c->stream, d_p, d_v, (long long)n, d_part);
    else
        hipLaunchKernelGGL(natac_cov_literal, dim3(blocks), dim3(256), 0, c->stream, d_p, d_v, (long long)n, d_part);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_part, (size_t)(mode == 0 ? 2 : 1) * blocks * sizeof(double),
                                            hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_p); dev_free(d_v); dev_free(d_part);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "calculate_cov: %s", hipGetErrorString(e));
    if (mode == 0) {
        double s1 = 0, s2 = 0;
        for (int i = 0; i < blocks; ++i) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
        *out = (s1 - s2 * s2) * r;
    } else {
        double s = 0;
        for (int i = 0; i < blocks; ++i) s += part[i];
        *out = s * r;
    }
    return NATAC_OK;
}

/* ---------------- operator-level entry points ---------------- */

int natac_smooth(natac_ctx *c, const double *x, int64_t n, const double *w, int M, int mode, int norm, double *out) {
    if (!c || !x || !w || !out) return fail(NATAC_E_ARG, "null argument");
    if (M < 1 || M % 2 != 1) return fail(NATAC_E_ARG, "window length must be odd (got %d)", M);
    if (mode != 0 && mode != 1) return fail(NATAC_E_ARG, "mode must be 0 (valid) or 1 (same)");
    if (n < M) return fail(NATAC_E_ARG, "signal (%lld) shorter than the window (%d)", (long long)n, M);
    const long long nout = mode == 0 ? n - M + 1 : n;
    HIPCHK(hipSetDevice(c->device));
    double *d_x = nullptr, *d_w = nullptr, *d_y = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_x, x, (size_t)n))) return rc;
    if ((rc = dev_upload(c, &d_w, w, (size_t)M))) { dev_free(d_x); return rc; }
    if ((rc = dev_alloc(&d_y, (size_t)nout))) { dev_free(d_x); dev_free(d_w); return rc; }
    hipLaunchKernelGGL(natac_smooth1d, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, c->stream, d_x, (long long)n, d_w, M, mode,
                       norm, d_y, nout);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_y, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_x); dev_free(d_w); dev_free(d_y);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "smooth: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_make_bias_mat(natac_ctx *c, const double *bias_log, int64_t nb, int64_t track_start, int64_t start, int64_t end,
                        int lower, int upper, double *mat) {
    if (!c || !bias_log || !mat) return fail(NATAC_E_ARG, "null argument");
    if (end <= start || upper <= lower || lower < 0) return fail(NATAC_E_ARG, "empty matrix");
    const long long ncol = end - start, nrow = upper - lower;
    if (ncol > 0x7fffffffLL) return fail(NATAC_E_ARG, "region too long");
    HIPCHK(hipSetDevice(c->device));
    double *d_b = nullptr, *d_m = nullptr;
    int *d_oob = nullptr, oob = 0;
    int rc;
    if ((rc = dev_upload(c, &d_b, bias_log, (size_t)nb))) return rc;
    if ((rc = dev_alloc(&d_m, (size_t)(nrow * ncol)))) { dev_free(d_b); return rc; }
    if ((rc = dev_alloc(&d_oob, 1))) { dev_free(d_b); dev_free(d_m); return rc; }
    hipError_t e = hipMemsetAsync(d_oob, 0, sizeof(int), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(natac_bias_mat_dense, dim3((unsigned)((nrow * ncol + 255) / 256)), dim3(256), 0, c->stream, d_b, (long long)nb,
                           (long long)(start - track_start), (int)ncol, lower, (int)nrow, d_m, d_oob);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mat, d_m, (size_t)(nrow * ncol) * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&oob, d_oob, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_b); dev_free(d_m); dev_free(d_oob);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "make_bias_mat: %s", hipGetErrorString(e));
    if (oob) return fail(NATAC_E_ARG, "bias track does not cover [start - upper//2, end + upper//2)");
    return NATAC_OK;
}

int natac_pwm_bias(natac_ctx *c, const uint8_t *seq, int64_t n, const double *log_pwm, const uint8_t *nucleotides, int nrow, int K,
                   double *out) {
    if (!c || !seq || !log_pwm || !nucleotides || !out) return fail(NATAC_E_ARG, "null argument");
    if (nrow < 1 || K < 1 || n < K) return fail(NATAC_E_ARG, "sequence shorter than the PWM");
    HIPCHK(hipSetDevice(c->device));
    unsigned char *d_s = nullptr, *d_n = nullptr;
    double *d_p = nullptr, *d_o = nullptr;
    const long long nout = n - K + 1;
    int rc;
    if ((rc = dev_upload(c, &d_s, (const unsigned char *)seq, (size_t)n))) return rc;
    if ((rc = dev_upload(c, &d_n, (const unsigned char *)nucleotides, (size_t)nrow))) { dev_free(d_s); return rc; }
    if ((rc = dev_upload(c, &d_p, log_pwm, (size_t)nrow * K))) { dev_free(d_s); dev_free(d_n); return rc; }
    if ((rc = dev_alloc(&d_o, (size_t)nout))) { dev_free(d_s); dev_free(d_n); dev_free(d_p); return rc; }
    hipLaunchKernelGGL(natac_pwm_score, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, c->stream, d_s, (long long)n, d_p, d_n, nrow, K,
                       d_o);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_s); dev_free(d_n); dev_free(d_p); dev_free(d_o);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "pwm_bias: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_correlate_valid(natac_ctx *c, const double *sub, int64_t ncol, const double *vmat, int R, int W, double *out) {
    if (!c || !sub || !vmat || !out) return fail(NATAC_E_ARG, "null argument");
    if (R < 1 || W < 1 || ncol < W) return fail(NATAC_E_ARG, "matrix narrower than the template");
    const long long nout = ncol - W + 1;
    HIPCHK(hipSetDevice(c->device));
    double *d_s = nullptr, *d_v = nullptr, *d_o = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_s, sub, (size_t)R * ncol))) return rc;
    if ((rc = dev_upload(c, &d_v, vmat, (size_t)R * W))) { dev_free(d_s); return rc; }
    if ((rc = dev_alloc(&d_o, (size_t)nout))) { dev_free(d_s); dev_free(d_v); return rc; }
    hipLaunchKernelGGL(natac_correlate_dense, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, c->stream, d_s, (long long)ncol, d_v, R, W,
                       d_o, nout);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_s); dev_free(d_v); dev_free(d_o);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "correlate_valid: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_calculate_occupancy(natac_ctx *c, const double *inserts, const double *bias, double *out) {
    if (!c || !inserts || !bias || !out) return fail(NATAC_E_ARG, "null argument");
    if (!c->have_occ) return fail(NATAC_E_STATE, "natac_set_occ_model has not been called");
    HIPCHK(hipSetDevice(c->device));
    double *d_i = nullptr, *d_b = nullptr, *d_o = nullptr;
    int *d_s = nullptr, st = 0;
    int rc;
    if ((rc = dev_upload(c, &d_i, inserts, (size_t)c->occ_upper))) return rc;
    if ((rc = dev_upload(c, &d_b, bias, (size_t)c->occ_upper))) { dev_free(d_i); return rc; }
    if ((rc = dev_alloc(&d_o, 3))) { dev_free(d_i); dev_free(d_b); return rc; }
    if ((rc = dev_alloc(&d_s, 1))) { dev_free(d_i); dev_free(d_b); dev_free(d_o); return rc; }
    hipLaunchKernelGGL(natac_occupancy_single, dim3(1), dim3(128), 0, c->stream, d_i, d_b, make_occ(c), d_o, d_s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&st, d_s, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_i); dev_free(d_b); dev_free(d_o); dev_free(d_s);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "calculate_occupancy: %s", hipGetErrorString(e));
    if (st) return fail(NATAC_E_ARG, "no alpha passes the likelihood-ratio test");
    return NATAC_OK;
}

/* ---------------- native track writer ---------------- */

/* ---------------- device-side track writer: entry points (helpers above extern "C") ---------------- */

int natac_batch_format_track(natac_batch *b, int track, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                             const int64_t *chunk_start, int write_zero, int compress, int64_t *n_bytes, int64_t *n_text_bytes,
                             int64_t *n_lines, int32_t *n_hard) {
    if (!b || !chrom_id || !names || !chunk_start || n_names <= 0) return fail(NATAC_E_ARG, "null argument");
    int rc = track_ready(b, track);
    if (rc) return rc;
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    if ((rc = materialise_track(b, track))) return rc;
    if (track != NATAC_T_INS)
        return format_values(b, b->d_track[track], chrom_id, names, n_names, chunk_start, write_zero, compress, n_bytes, n_text_bytes, n_lines,
                             n_hard);
    double *d_tmp = nullptr;                   // insertion counts are int32 on the device; the writer takes float64 like the reference's track
    if ((rc = dev_alloc(&d_tmp, (size_t)b->total_bp))) return rc;
    const int blocks = (int)std::min<long long>((b->total_bp + 255) / 256, 65536);
    hipLaunchKernelGGL(natac_i32_to_f64, dim3(blocks), dim3(256), 0, c->stream, (const int *)b->d_track[NATAC_T_INS], d_tmp, b->total_bp);
    rc = format_values(b, d_tmp, chrom_id, names, n_names, chunk_start, write_zero, compress, n_bytes, n_text_bytes, n_lines, n_hard);
    (void)hipStreamSynchronize(c->stream);
    dev_free(d_tmp);
    return rc;
}

// results handed to natac_batch_format_fetch_begin: wait for their copies, give the device buffers back
static int fmt_pending_wait(natac_batch *b) {
    hipError_t first = hipSuccess;
    for (auto &p : b->fmt_pending) {
        hipError_t e = hipEventSynchronize(p.second);
        if (e != hipSuccess && first == hipSuccess) first = e;
        (void)hipEventDestroy(p.second);
        dev_free(p.first);
    }
    b->fmt_pending.clear();
    if (first != hipSuccess) return fail(NATAC_E_HIP, "format_fetch: %s", hipGetErrorString(first));
    return NATAC_OK;
}

int natac_batch_format_fetch_begin(natac_batch *b, void *dst, size_t dst_bytes) {
    if (!b || (!dst && dst_bytes)) return fail(NATAC_E_ARG, "null argument");
    if (b->fmt_bytes < 0) return fail(NATAC_E_STATE, "natac_batch_format_track has not run");
    if ((long long)dst_bytes < b->fmt_bytes) return fail(NATAC_E_ARG, "destination holds %zu bytes, result has %lld", dst_bytes, b->fmt_bytes);
    if (b->fmt_bytes == 0) return NATAC_OK;
    if (!b->d_fmt_out) return fail(NATAC_E_STATE, "the result has already been handed to natac_batch_format_fetch_begin");
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // natac_batch_format_track returns with its stream drained: the result is complete, the copy needs no event from it
    hipEvent_t ev = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipMemcpyAsync(dst, b->d_fmt_out, (size_t)b->fmt_bytes, hipMemcpyDeviceToHost, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(ev, c->copy_stream);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); return fail(NATAC_E_HIP, "format_fetch_begin: %s", hipGetErrorString(e)); }
    b->fmt_pending.emplace_back(b->d_fmt_out, ev);       // the buffer is the copy's until natac_batch_format_fetch_wait
    b->d_fmt_out = nullptr;          // fmt_bytes stays: the result's tabix records are still to be fetched; a second fetch of the bytes is refused below
    return NATAC_OK;
}

int natac_batch_format_fetch_wait(natac_batch *b) {
    if (!b) return fail(NATAC_E_ARG, "batch is NULL");
    HIPCHK(hipSetDevice(b->ctx->device));
    return fmt_pending_wait(b);
}

int natac_batch_format_fetch(natac_batch *b, void *dst, size_t dst_bytes) {
    if (!b || (!dst && dst_bytes)) return fail(NATAC_E_ARG, "null argument");
    if (b->fmt_bytes < 0) return fail(NATAC_E_STATE, "natac_batch_format_track has not run");
    if ((long long)dst_bytes < b->fmt_bytes) return fail(NATAC_E_ARG, "destination holds %zu bytes, result has %lld", dst_bytes, b->fmt_bytes);
    if (b->fmt_bytes == 0) return NATAC_OK;
    if (!b->d_fmt_out) return fail(NATAC_E_STATE, "the result has already been handed to natac_batch_format_fetch_begin");
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipMemcpyAsync(dst, b->d_fmt_out, (size_t)b->fmt_bytes, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    return NATAC_OK;
}

struct natac_tbi { natac_tabix::Builder bld; };

int natac_tbi_create(natac_tbi **out) {
    if (!out) return fail(NATAC_E_ARG, "null argument");
    *out = new natac_tbi();
    return NATAC_OK;
}
void natac_tbi_free(natac_tbi *t) { delete t; }

static int tbi_push_groups(natac_tbi *t, int64_t n, const char *const *names, int32_t n_names, const int32_t *cid, const int64_t *beg,
                           const int64_t *end, const int64_t *count, const uint64_t *t0, const uint64_t *t1, const uint64_t *member_pos,
                           int64_t n_members, int64_t n_text, int64_t file_offset) {
    const size_t nblk = (size_t)n_members;
    // virtual offset of a text position: member start in the file << 16 | offset inside the member; a position at a member's end
    // belongs to the start of the next member (for the last one: whatever the caller writes next -- more members or the EOF block)
    auto voff = [&](uint64_t toff) -> uint64_t {
        if (toff >= (uint64_t)n_text) return ((uint64_t)file_offset + member_pos[nblk]) << 16;      // the end of the last (partial) member
        const size_t m = (size_t)(toff / natac_deflate::BLK);
        return (((uint64_t)file_offset + member_pos[m]) << 16) | (toff - (uint64_t)m * natac_deflate::BLK);
    };
    for (int64_t i = 0; i < n; ++i) {
        if (cid[i] < 0 || cid[i] >= n_names) return fail(NATAC_E_ARG, "record with chromosome id %d", cid[i]);
        const char *nm = names[cid[i]];
        if (!t->bld.push(nm, strlen(nm), beg[i], end[i], voff(t0[i]), voff(t1[i]), count[i])) return fail(NATAC_E_ARG, "tabix index: %s", t->bld.err.c_str());
    }
    return NATAC_OK;
}

int natac_batch_format_index_size(natac_batch *b, int64_t *n_groups, int64_t *n_members, int64_t *n_text) {
    if (!b || !n_groups || !n_members || !n_text) return fail(NATAC_E_ARG, "null argument");
    *n_text = b->fmt_text_bytes;
    if (b->fmt_bytes < 0) return fail(NATAC_E_STATE, "natac_batch_format_track has not run");
    *n_groups = (int64_t)b->fmt_groups.size();
    *n_members = b->fmt_member_pos.empty() ? 0 : (int64_t)b->fmt_member_pos.size() - 1;
    return NATAC_OK;
}

int natac_batch_format_index_fetch(natac_batch *b, int32_t *cid, int64_t *beg, int64_t *end, int64_t *count, uint64_t *t0, uint64_t *t1,
                                   uint64_t *member_pos) {
    if (!b) return fail(NATAC_E_ARG, "null argument");
    if (b->fmt_bytes < 0) return fail(NATAC_E_STATE, "natac_batch_format_track has not run");
    const size_t n = b->fmt_groups.size();
    if (n && (!cid || !beg || !end || !count || !t0 || !t1)) return fail(NATAC_E_ARG, "null argument");
    for (size_t i = 0; i < n; ++i) {
        const auto &g = b->fmt_groups[i];
        cid[i] = g.cid; beg[i] = g.beg; end[i] = g.end; count[i] = g.count; t0[i] = g.t0; t1[i] = g.t1;
    }
    if (!b->fmt_member_pos.empty()) {
        if (!member_pos) return fail(NATAC_E_ARG, "null argument");
        memcpy(member_pos, b->fmt_member_pos.data(), b->fmt_member_pos.size() * sizeof(uint64_t));
    }
    return NATAC_OK;
}

int natac_tbi_push(natac_tbi *t, int64_t n, const char *const *names, int32_t n_names, const int32_t *cid, const int64_t *beg,
                   const int64_t *end, const int64_t *count, const uint64_t *t0, const uint64_t *t1, const uint64_t *member_pos,
                   int64_t n_members, int64_t n_text, int64_t file_offset) {
    if (!t || n < 0 || n_members < 0 || file_offset < 0 || n_text < 0) return fail(NATAC_E_ARG, "bad argument");
    if (n == 0) return NATAC_OK;
    if (!names || !cid || !beg || !end || !count || !t0 || !t1 || !member_pos) return fail(NATAC_E_ARG, "null argument");
    return tbi_push_groups(t, n, names, n_names, cid, beg, end, count, t0, t1, member_pos, n_members, n_text, file_offset);
}

int natac_tbi_write(natac_tbi *t, const char *tbi_path, int64_t *n_records) {
    if (!t || !tbi_path) return fail(NATAC_E_ARG, "null argument");
    const int rc = t->bld.write(tbi_path);
    if (rc) return fail(NATAC_E_ARG, "cannot write %s (code %d)", tbi_path, rc);
    if (n_records) *n_records = t->bld.nrec;
    return NATAC_OK;
}

int natac_format_doubles(natac_ctx *c, const double *vals, int64_t n, char *out, size_t out_cap, int64_t *out_off, int32_t *n_hard) {
    if (!c || !vals || !out || !out_off || n <= 0) return fail(NATAC_E_ARG, "bad argument");
    if (out_cap < (size_t)n * natac_text::MAX_VALUE_CHARS) return fail(NATAC_E_ARG, "out must hold %d bytes per value", natac_text::MAX_VALUE_CHARS);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_text_tables(c);
    if (rc) return rc;
    double *d_v = nullptr;
    char *d_o = nullptr;
    int *d_len = nullptr, *d_hard = nullptr;
    if ((rc = dev_upload(c, &d_v, vals, (size_t)n))) return rc;
    TmpFree tmp;
    tmp.keep(d_v);
    if ((rc = dev_alloc(&d_o, (size_t)n * natac_text::MAX_VALUE_CHARS))) return rc;
    tmp.keep(d_o);
    if ((rc = dev_alloc(&d_len, (size_t)n + 1))) return rc;
    tmp.keep(d_len);
    HIPCHK(hipMemsetAsync(d_len + n, 0, sizeof(int), c->stream));
    d_hard = d_len + n;
    hipLaunchKernelGGL(natac_textz::tz_format_values, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_v, (long long)n, c->d_p10, d_o,
                       d_len, d_hard);
    std::vector<int> len((size_t)n + 1);
    std::vector<char> raw((size_t)n * natac_text::MAX_VALUE_CHARS);
    HIPCHK(hipMemcpyAsync(len.data(), d_len, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(raw.data(), d_o, raw.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long o = 0;
    for (int64_t i = 0; i < n; ++i) {
        out_off[i] = o;
        memcpy(out + o, raw.data() + (size_t)i * natac_text::MAX_VALUE_CHARS, (size_t)len[(size_t)i]);
        o += len[(size_t)i];
    }
    out_off[n] = o;
    if (n_hard) *n_hard = len[(size_t)n];
    return NATAC_OK;
}

int natac_bgzf_lines_host(const char *text, int64_t n, const int64_t *line_off, int64_t n_lines, void *out, size_t out_cap, int64_t *n_bytes) {
    if (!text || !line_off || !n_bytes || n < 0 || n_lines < 0) return fail(NATAC_E_ARG, "bad argument");
    std::string res;
    if (!natac_deflate::bgzf_lines_host((const unsigned char *)text, n, (const long long *)line_off, n_lines, res))
        return fail(NATAC_E_ARG, "Huffman table description too long");
    *n_bytes = (int64_t)res.size();
    if (res.size() > out_cap) return fail(NATAC_E_ARG, "destination holds %zu bytes, result has %zu", out_cap, res.size());
    if (out && !res.empty()) memcpy(out, res.data(), res.size());
    return NATAC_OK;
}

/* ---------------- native track writer + bgzip ---------------- */

int natac_write_bedgraph(const char *path, int append, int compress, int finish, int32_t n_chunks, const char *const *chroms,
                         const int64_t *chunk_start, const int64_t *out_off, const double *vals, int write_zero, int n_threads,
                         int64_t *bytes_written) {
    if (!path || n_chunks < 0 || (n_chunks > 0 && (!chroms || !chunk_start || !out_off || !vals)))
        return fail(NATAC_E_ARG, "null argument");
    if (compress < 0 || compress > 9) return fail(NATAC_E_ARG, "compress must be 0 (text) or a deflate level 1..9");
    for (int i = 0; i < n_chunks; ++i) {
        if (!chroms[i] || std::strlen(chroms[i]) > 100) return fail(NATAC_E_ARG, "bad chromosome name for chunk %d", i);
        if (out_off[i + 1] < out_off[i]) return fail(NATAC_E_ARG, "out_off must be non-decreasing");
    }
    const int rc = natac_writer::write_bedgraph(path, append != 0, compress, finish != 0, n_chunks, chroms, chunk_start, out_off, vals,
                                                (write_zero & 1) != 0, n_threads, bytes_written, (write_zero & 2) != 0);
    if (rc == 1) return fail(NATAC_E_ARG, "cannot open %s", path);
    if (rc == 2) return fail(NATAC_E_ARG, "write to %s failed", path);
    if (rc == 3) return fail(NATAC_E_NOMEM, "deflate failed");
    return NATAC_OK;
}

int natac_write_bed_rows(const char *path, int append, int64_t n_rows, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                         const int64_t *start, const int64_t *end, const double *vals, int32_t n_cols) {
    if (!path || n_rows < 0 || n_cols < 0 || n_cols > 32) return fail(NATAC_E_ARG, "bad argument");
    if (n_rows > 0 && (!chrom_id || !names || !start || !end || (n_cols > 0 && !vals))) return fail(NATAC_E_ARG, "null argument");
    for (int64_t r = 0; r < n_rows; ++r)
        if (chrom_id[r] < 0 || chrom_id[r] >= n_names) return fail(NATAC_E_ARG, "row %lld: chromosome id %d out of range", (long long)r, chrom_id[r]);
    const int rc = natac_writer::write_bed_rows(path, append != 0, n_rows, chrom_id, names, start, end, vals, n_cols);
    if (rc == 1) return fail(NATAC_E_ARG, "cannot open %s", path);
    if (rc) return fail(NATAC_E_ARG, "write error on %s", path);
    return NATAC_OK;
}

int natac_write_bed_rows_labeled(const char *path, int append, int64_t n_rows, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                                 const int64_t *start, const int64_t *end, const double *vals, int32_t n_cols, const int32_t *label_id,
                                 const char *const *labels, int32_t n_labels) {
    if (!path || n_rows < 0 || n_cols < 0 || n_cols > 32) return fail(NATAC_E_ARG, "bad argument");
    if (n_rows > 0 && (!chrom_id || !names || !start || !end || (n_cols > 0 && !vals) || !label_id || !labels)) return fail(NATAC_E_ARG, "null argument");
    for (int64_t r = 0; r < n_rows; ++r) {
        if (chrom_id[r] < 0 || chrom_id[r] >= n_names) return fail(NATAC_E_ARG, "row %lld: chromosome id %d out of range", (long long)r, chrom_id[r]);
        if (label_id[r] < 0 || label_id[r] >= n_labels) return fail(NATAC_E_ARG, "row %lld: label id %d out of range", (long long)r, label_id[r]);
    }
    const int rc = natac_writer::write_bed_rows(path, append != 0, n_rows, chrom_id, names, start, end, vals, n_cols, label_id, labels);
    if (rc == 1) return fail(NATAC_E_ARG, "cannot open %s", path);
    if (rc) return fail(NATAC_E_ARG, "write error on %s", path);
    return NATAC_OK;
}

int natac_bgzip_file(const char *src, const char *dst, int level, int n_threads) {
    if (!src || !dst) return fail(NATAC_E_ARG, "null argument");
    if (level < 1 || level > 9) return fail(NATAC_E_ARG, "level must be 1..9");
    std::string text;
    {
        FILE *f = std::fopen(src, "rb");
        if (!f) return fail(NATAC_E_ARG, "cannot open %s", src);
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        text.resize((size_t)sz);
        const bool ok = sz == 0 || std::fread(&text[0], 1, (size_t)sz, f) == (size_t)sz;
        std::fclose(f);
        if (!ok) return fail(NATAC_E_ARG, "cannot read %s", src);
    }
    const size_t BLK = 0xff00, nblk = (text.size() + BLK - 1) / BLK;
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, (nblk + 15) / 16));
    std::vector<std::string> parts(n_threads);
    std::vector<int> bad(n_threads, 0);
    auto work = [&](int t) {   // contiguous block ranges so that the parts concatenate in file order
        const size_t b0 = nblk * t / n_threads, b1 = nblk * (t + 1) / n_threads;
        natac_writer::Deflater df;                // one z_stream per thread, deflateReset per member
        parts[t].reserve((b1 - b0) * 24000);
        for (size_t b = b0; b < b1; ++b)
            if (!natac_writer::bgzf_block(parts[t], (const unsigned char *)text.data() + b * BLK, std::min(BLK, text.size() - b * BLK), level, &df)) {
                bad[t] = 1;
                return;
            }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int b : bad) if (b) return fail(NATAC_E_NOMEM, "deflate failed");
    FILE *f = std::fopen(dst, "wb");
    if (!f) return fail(NATAC_E_ARG, "cannot open %s", dst);
    bool ok = true;
    for (auto &p : parts) ok = ok && (p.empty() || std::fwrite(p.data(), 1, p.size(), f) == p.size());
    ok = ok && std::fwrite(natac_writer::BGZF_EOF, 1, 28, f) == 28;
    if (std::fclose(f) != 0 || !ok) return fail(NATAC_E_ARG, "write to %s failed", dst);
    return NATAC_OK;
}

int natac_tabix_index(const char *path, const char *tbi_path, int n_threads, int64_t *n_records) {
    if (!path) return fail(NATAC_E_ARG, "path is NULL");
    std::string msg;
    const int rc = natac_tabix::index_bed(path, tbi_path, n_threads, n_records, &msg);
    switch (rc) {
        case 0: return NATAC_OK;
        case 1: return fail(NATAC_E_ARG, "cannot read %s", path);
        case 2: return fail(NATAC_E_ARG, "%s is not a BGZF file", path);
        case 3: return fail(NATAC_E_NOMEM, "inflate / deflate failed on %s", path);
        case 4: return fail(NATAC_E_ARG, "%s: %s", path, msg.c_str());
        default: return fail(NATAC_E_ARG, "cannot write the index of %s", path);
    }
}

int natac_pack_chunks(int32_t n_chunks, const int64_t *chunk_start, const int64_t *chunk_end, const int32_t *chrom_id, int32_t n_chroms,
                      const int64_t *const *pos, const int64_t *const *tlen, const int64_t *n_per_chrom, int64_t margin, int atac,
                      int64_t *frag_off, int64_t *first, int32_t *lpos, int32_t *ilen, int n_threads) {
    if (n_chunks < 0 || n_chroms < 0 || margin < 0) return fail(NATAC_E_ARG, "bad sizes");
    if (n_chunks > 0 && (!chunk_start || !chunk_end || !chrom_id || !frag_off || !first)) return fail(NATAC_E_ARG, "null argument");
    if (n_chroms > 0 && (!pos || !tlen || !n_per_chrom)) return fail(NATAC_E_ARG, "null argument");
    for (int32_t i = 0; i < n_chunks; ++i) {
        if (chrom_id[i] >= n_chroms) return fail(NATAC_E_ARG, "chunk %d: chromosome id %d out of range", i, chrom_id[i]);
        if (chunk_end[i] < chunk_start[i]) return fail(NATAC_E_ARG, "chunk %d: end < start", i);
    }
    const int shift = atac ? 4 : 0, trim = atac ? 8 : 0;
    if (!lpos || !ilen) {
        if (!frag_off) return fail(NATAC_E_ARG, "null argument");
        if (n_chunks == 0) { if (frag_off) frag_off[0] = 0; return NATAC_OK; }
        natac_pack::count(n_chunks, chunk_start, chunk_end, chrom_id, pos, n_per_chrom, margin, shift, frag_off, first);
        return NATAC_OK;
    }
    if (n_chunks == 0) return NATAC_OK;
    natac_pack::fill(n_chunks, chunk_start, chrom_id, pos, tlen, frag_off, first, shift, trim, lpos, ilen, n_threads);
    return NATAC_OK;
}

struct natac_tbx { natac_tabix::Reader *impl = nullptr; };

int natac_tbx_open(const char *path, natac_tbx **out) {
    if (!path || !out) return fail(NATAC_E_ARG, "null argument");
    *out = nullptr;
    natac_tabix::Reader *r = nullptr;
    const int rc = natac_tabix::open_reader(path, &r);
    if (rc == 1) return fail(NATAC_E_ARG, "cannot open %s (or its .tbi)", path);
    if (rc) return fail(NATAC_E_ARG, "%s.tbi is not a tabix index", path);
    natac_tbx *t = new natac_tbx();
    t->impl = r;
    *out = t;
    return NATAC_OK;
}

void natac_tbx_close(natac_tbx *t) {
    if (!t) return;
    natac_tabix::close_reader(t->impl);
    delete t;
}

int natac_tbx_read_values(natac_tbx *t, const char *chrom, int64_t start, int64_t end, int value_col, double empty, double *out,
                          int64_t *n_records) {
    if (!t || !chrom || (end > start && !out)) return fail(NATAC_E_ARG, "null argument");
    if (value_col < 1 || value_col > 8) return fail(NATAC_E_ARG, "value_col must be 1..8");
    const int64_t n = natac_tabix::read_values(t->impl, chrom, start, end, value_col, empty, out);
    if (n < 0) return fail(NATAC_E_ARG, "read error in the indexed file");
    if (n_records) *n_records = n;
    return NATAC_OK;
}

int natac_tbx_read_regions(natac_tbx *t, int64_t n, const int32_t *chrom_id, const char *const *names, int32_t n_names,
                           const int64_t *start, const int64_t *end, int value_col, double empty, double *out, const int64_t *out_off,
                           int n_threads, int64_t *n_records) {
    if (!t || n < 0 || (n > 0 && (!chrom_id || !names || !start || !end || !out_off))) return fail(NATAC_E_ARG, "null argument");
    if (value_col < 1 || value_col > 8) return fail(NATAC_E_ARG, "value_col must be 1..8");
    for (int64_t i = 0; i < n; ++i) {
        if (chrom_id[i] < 0 || chrom_id[i] >= n_names) return fail(NATAC_E_ARG, "region %lld: chromosome index out of range", (long long)i);
        if (end[i] > start[i] && !out) return fail(NATAC_E_ARG, "null output");
    }
    const int64_t u = natac_tabix::read_regions(t->impl, n, chrom_id, names, n_names, start, end, value_col, empty, out, out_off, n_threads);
    if (u < 0) return fail(NATAC_E_ARG, "read error in the indexed file");
    if (n_records) *n_records = u;
    return NATAC_OK;
}

int natac_fuzz_evaluate(int32_t K, int32_t n, int32_t M, const double *X0, const double *lb, const double *ub, const double *sig,
                        const double *xs, const int64_t *lens, void *exp_loop, void *exp_data, double *work, double *f, double *g) {
    if (K < 0 || M < 0 || (n != 3 && n != 6 && n != 9)) return fail(NATAC_E_ARG, "n must be 3, 6 or 9");
    if (K == 0) return NATAC_OK;
    if (!X0 || !lb || !ub || !sig || !xs || !lens || !exp_loop || !work || !f || !g) return fail(NATAC_E_ARG, "null argument");
    for (int k = 0; k < K; ++k)
        if (lens[k] < 0 || lens[k] > M) return fail(NATAC_E_ARG, "fit %d: window length out of range", k);
    natac_fuzzfit::evaluate(K, n, M, X0, lb, ub, sig, xs, lens, (natac_fuzzfit::ufunc_loop)exp_loop, exp_data, work, f, g);
    return NATAC_OK;
}

/* ---------------- BED-like table reader ---------------- */

struct natac_bedtab { natac_bedtabio::Table impl; };

int natac_bedtab_open(const char *path, const int32_t *cols, int32_t n_cols, natac_bedtab **out) {
    if (!path || !out || n_cols < 0 || n_cols > 32 || (n_cols > 0 && !cols)) return fail(NATAC_E_ARG, "bad argument");
    *out = nullptr;
    for (int c = 0; c < n_cols; ++c)
        if (cols[c] < 0 || cols[c] > 4095) return fail(NATAC_E_ARG, "column index out of range");
    natac_bedtab *t = new natac_bedtab();
    std::string err;
    const int rc = natac_bedtabio::load(path, cols, n_cols, &t->impl, err);
    if (rc) { delete t; return fail(NATAC_E_ARG, "%s: %s", path, err.c_str()); }
    *out = t;
    return NATAC_OK;
}

void natac_bedtab_close(natac_bedtab *t) { delete t; }

int natac_bedtab_dims(natac_bedtab *t, int64_t *n_rows, int32_t *n_names) {
    if (!t) return fail(NATAC_E_ARG, "table is NULL");
    if (n_rows) *n_rows = (int64_t)t->impl.chrom_id.size();
    if (n_names) *n_names = (int32_t)t->impl.names.size();
    return NATAC_OK;
}

int natac_bedtab_name(natac_bedtab *t, int32_t i, char *name, size_t name_len) {
    if (!t || !name) return fail(NATAC_E_ARG, "null argument");
    if (i < 0 || (size_t)i >= t->impl.names.size()) return fail(NATAC_E_ARG, "name index out of range");
    const std::string &s = t->impl.names[(size_t)i];
    if (s.size() + 1 > name_len) return fail(NATAC_E_ARG, "name longer than the buffer");
    std::memcpy(name, s.c_str(), s.size() + 1);
    return NATAC_OK;
}

int natac_bedtab_fetch(natac_bedtab *t, int32_t *chrom_id, int64_t *start, int64_t *end, double *vals) {
    if (!t) return fail(NATAC_E_ARG, "table is NULL");
    const size_t n = t->impl.chrom_id.size();
    if (n && (!chrom_id || !start || !end || (t->impl.n_cols && !vals))) return fail(NATAC_E_ARG, "null argument");
    if (n) {
        std::memcpy(chrom_id, t->impl.chrom_id.data(), n * sizeof(int32_t));
        std::memcpy(start, t->impl.start.data(), n * sizeof(int64_t));
        std::memcpy(end, t->impl.end.data(), n * sizeof(int64_t));
        if (t->impl.n_cols) std::memcpy(vals, t->impl.vals.data(), n * (size_t)t->impl.n_cols * sizeof(double));
    }
    return NATAC_OK;
}

/* ---------------- native FASTA loader ---------------- */

struct natac_fasta { natac_fastaio::Fasta *impl = nullptr; };

int natac_fasta_open(const char *path, int n_threads, natac_fasta **out) {
    if (!path || !out) return fail(NATAC_E_ARG, "null argument");
    *out = nullptr;
    std::string err;
    natac_fastaio::Fasta *impl = natac_fastaio::load(path, n_threads, err);
    if (!impl) return fail(NATAC_E_ARG, "%s: %s", path, err.c_str());
    natac_fasta *h = new natac_fasta();
    h->impl = impl;
    *out = h;
    return NATAC_OK;
}

void natac_fasta_close(natac_fasta *fa) {
    if (!fa) return;
    delete fa->impl;
    delete fa;
}

int natac_fasta_count(natac_fasta *fa, int32_t *n_records) {
    if (!fa || !n_records) return fail(NATAC_E_ARG, "null argument");
    *n_records = (int32_t)fa->impl->recs.size();
    return NATAC_OK;
}

int natac_fasta_info(natac_fasta *fa, int32_t record, char *name, size_t name_len, int64_t *length) {
    if (!fa) return fail(NATAC_E_ARG, "fasta is NULL");
    if (record < 0 || (size_t)record >= fa->impl->recs.size()) return fail(NATAC_E_ARG, "record out of range");
    const natac_fastaio::Record &r = fa->impl->recs[(size_t)record];
    if (name && name_len) {
        if (r.name.size() + 1 > name_len) return fail(NATAC_E_ARG, "record name longer than the buffer");
        std::memcpy(name, r.name.c_str(), r.name.size() + 1);
    }
    if (length) *length = r.length;
    return NATAC_OK;
}

int natac_fasta_read(natac_fasta *fa, int32_t record, void *out, int64_t n) {
    if (!fa) return fail(NATAC_E_ARG, "fasta is NULL");
    if (record < 0 || (size_t)record >= fa->impl->recs.size()) return fail(NATAC_E_ARG, "record out of range");
    if (n != fa->impl->recs[(size_t)record].length || (n > 0 && !out)) return fail(NATAC_E_ARG, "buffer does not match the record's length");
    natac_fastaio::read_record(fa->impl, (size_t)record, (unsigned char *)out);
    return NATAC_OK;
}

/* ---------------- native BAM extractor ---------------- */

int natac_bam_open(const char *path, int n_threads, natac_bam **out) {
    if (!path || !out) return fail(NATAC_E_ARG, "null argument");
    *out = nullptr;
    std::string err;
    size_t window = (size_t)48 << 20;                 // compressed bytes per streaming window; NATAC_BAM_WINDOW overrides (tests)
    if (const char *e = getenv("NATAC_BAM_WINDOW")) { const long long v = atoll(e); if (v > 0) window = (size_t)v; }
    natac_bamio::Bam *impl = natac_bamio::decode(path, n_threads, err, window);
    if (!impl) return fail(NATAC_E_ARG, "%s: %s", path, err.c_str());
    natac_bam *h = new natac_bam();
    h->impl = impl;
    *out = h;
    return NATAC_OK;
}

int natac_bam_open_device(natac_ctx *c, const char *path, natac_bam **out, int *on_device) {
    if (!c || !path || !out) return fail(NATAC_E_ARG, "null argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    std::string err;
    size_t window = (size_t)8 << 30;                  // compressed bytes per device window; NATAC_BAM_DEV_WINDOW overrides (tests)
    if (const char *e = getenv("NATAC_BAM_DEV_WINDOW")) { const long long v = atoll(e); if (v > 0) window = (size_t)v; }
    bool undecided = false;
    natac_bamio::Bam *impl = natac_bamdev::decode_device(path, c->stream, err, &undecided, window);
    if (on_device) *on_device = impl ? 1 : 0;
    if (!impl && undecided) {                         // the chain of record starts was not confirmed: the host decoder answers
        size_t hw = (size_t)48 << 20;
        if (const char *e = getenv("NATAC_BAM_WINDOW")) { const long long v = atoll(e); if (v > 0) hw = (size_t)v; }
        impl = natac_bamio::decode(path, 0, err, hw);
    }
    if (!impl) return fail(NATAC_E_ARG, "%s: %s", path, err.c_str());
    natac_bam *h = new natac_bam();
    h->impl = impl;
    *out = h;
    return NATAC_OK;
}

int natac_inflate_raw_host(const void *src, size_t csize, void *out, size_t isize) {
    if ((!src && csize) || (!out && isize) || csize > 0xffffffffull || isize > 0xffffffffull) return -1;
    std::vector<unsigned char> padded(csize + 8, 0);         // the bit reader takes whole words: 8 readable bytes behind the payload
    if (csize) std::memcpy(padded.data(), src, csize);
    return natac_bamdev::inflate_member_host(padded.data(), (unsigned int)csize, (unsigned char *)out, (unsigned int)isize);
}

void natac_bam_close(natac_bam *bam) {
    if (!bam) return;
    delete bam->impl;
    delete bam;
}

int natac_bam_counts(natac_bam *bam, int32_t *n_refs, int64_t *n_records, int64_t *n_kept) {
    if (!bam) return fail(NATAC_E_ARG, "bam is NULL");
    if (n_refs) *n_refs = (int32_t)bam->impl->refs.size();
    if (n_records) *n_records = bam->impl->n_records;
    if (n_kept) *n_kept = bam->impl->n_kept;
    return NATAC_OK;
}

int natac_bam_ref_info(natac_bam *bam, int32_t ref, char *name, size_t name_len, int64_t *length, int64_t *n_reads) {
    if (!bam || ref < 0 || ref >= (int32_t)bam->impl->refs.size()) return fail(NATAC_E_ARG, "bad reference index");
    const natac_bamio::Ref &r = bam->impl->refs[ref];
    if (name && name_len) snprintf(name, name_len, "%s", r.name.c_str());
    if (length) *length = r.length;
    if (n_reads) *n_reads = (int64_t)r.pos.size();
    return NATAC_OK;
}

int natac_bam_ref_reads(natac_bam *bam, int32_t ref, int64_t *pos, int64_t *tlen, int64_t n) {
    if (!bam || ref < 0 || ref >= (int32_t)bam->impl->refs.size()) return fail(NATAC_E_ARG, "bad reference index");
    const natac_bamio::Ref &r = bam->impl->refs[ref];
    if (n != (int64_t)r.pos.size()) return fail(NATAC_E_ARG, "reference holds %zu reads, buffers are for %lld", r.pos.size(), (long long)n);
    if (n && (!pos || !tlen)) return fail(NATAC_E_ARG, "null argument");
    if (n) {
        std::memcpy(pos, r.pos.data(), (size_t)n * sizeof(int64_t));
        std::memcpy(tlen, r.tlen.data(), (size_t)n * sizeof(int64_t));
    }
    return NATAC_OK;
}

/* ---------------- pinned host memory + pool control ---------------- */

int natac_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(NATAC_E_ARG, "out is NULL");
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return NATAC_OK;
}

int natac_host_free(void *p) {
    if (!p) return NATAC_OK;
    HIPCHK(hipHostFree(p));
    return NATAC_OK;
}

int natac_pool_trim(void) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 16; ++d) {
        if (g_pool.free_blocks[d].empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        g_pool.trim_locked(d);
    }
    (void)hipSetDevice(cur);
    return NATAC_OK;
}

/* ---------------- profiling ---------------- */

int natac_profile_enable(natac_ctx *c, int on) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    c->profiling = on != 0;
    return NATAC_OK;
}
int natac_profile_get(natac_ctx *c, int k, double *ms_total, int64_t *launches) {
    if (!c || k < 0 || k >= NATAC_K_COUNT) return fail(NATAC_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    prof_collect(c);
    if (ms_total) *ms_total = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return NATAC_OK;
}
int natac_profile_reset(natac_ctx *c) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    HIPCHK(sync_all(c));
    prof_collect(c);
    for (int i = 0; i < NATAC_K_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return NATAC_OK;
}
int natac_timer_start(natac_ctx *c) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->t0, c->stream));
    return NATAC_OK;
}
int natac_timer_stop(natac_ctx *c, double *ms) {
    if (!c || !ms) return fail(NATAC_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->t1, c->stream));
    HIPCHK(hipEventSynchronize(c->t1));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, c->t0, c->t1));
    *ms = f;
    prof_collect(c);
    return NATAC_OK;
}

/* ---------------- resident tracks between the stages of one process ---------------- */

struct natac_store {
    struct Seg { double *p[4]; int n_tracks; long long n; int device; };
    std::mutex mu;
    std::vector<Seg> segs;
    // HBM budget: the store is a convenience next to the files, it must never be what fills the device.  A segment is adopted only
    // while the store stays below max_bytes AND the device keeps min_free bytes (default: a quarter of its memory) for the batches
    // of the pipeline; the first refusal closes the store (later sub-batches go through the files without asking again).
    long long bytes = 0, max_bytes = -1, min_free = -1, declined = 0;
    bool closed = false;
};

int natac_store_create(natac_store **out) {
    if (!out) return fail(NATAC_E_ARG, "out is NULL");
    natac_store *s = new natac_store();
    const char *e = getenv("NATAC_STORE_MAX_BYTES");
    if (e && *e) s->max_bytes = atoll(e);
    e = getenv("NATAC_STORE_MIN_FREE_BYTES");
    if (e && *e) s->min_free = atoll(e);
    *out = s;
    return NATAC_OK;
}

int natac_store_set_budget(natac_store *s, int64_t max_bytes, int64_t min_free_bytes) {
    if (!s) return fail(NATAC_E_ARG, "store is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    s->max_bytes = max_bytes;
    s->min_free = min_free_bytes;
    s->closed = false;
    return NATAC_OK;
}

void natac_store_free(natac_store *s) {
    if (!s) return;
    for (auto &g : s->segs)
        for (int i = 0; i < g.n_tracks; ++i) dev_free(g.p[i]);
    delete s;
}

int natac_store_adopt(natac_store *s, natac_batch *b, int32_t n_tracks, const int32_t *tracks, int write_zero, int64_t *segment,
                      int32_t *n_hard) {
    using namespace natac_textz;
    if (!s || !b || !tracks || !segment || n_tracks < 1 || n_tracks > 4) return fail(NATAC_E_ARG, "bad argument");
    *segment = -1;
    if (n_hard) *n_hard = 0;
    natac_ctx *c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    for (int i = 0; i < n_tracks; ++i) {
        if (tracks[i] == NATAC_T_INS) return fail(NATAC_E_ARG, "float64 tracks only");
        if ((rc = track_ready(b, tracks[i]))) return rc;
        if ((rc = materialise_track(b, tracks[i]))) return rc;
    }
    if (b->total_bp >= 0xffffffffLL) return fail(NATAC_E_ARG, "batch too long (%lld bases)", b->total_bp);
    {   // budget first: nothing is launched or allocated for a segment the store will not keep
        const long long need = (long long)n_tracks * b->total_bp * (long long)sizeof(double);
        size_t avail = 0, total = 0;
        HIPCHK(pool_mem_info(c->device, &avail, &total));
        std::lock_guard<std::mutex> lk(s->mu);
        const long long min_free = s->min_free >= 0 ? s->min_free : (long long)(total / 4);
        // run tables + one pass of scratch ride on top of the segment itself while it is formed
        const long long scratch = 2 * b->total_bp * (long long)sizeof(int) + (1 << 20);
        if (s->closed || (s->max_bytes >= 0 && s->bytes + need > s->max_bytes) || (long long)avail - need - scratch < min_free) {
            s->closed = true;
            ++s->declined;
            if (n_hard) *n_hard = -1;      // declined: budget (the caller reads the files, as for n_hard > 0)
            return NATAC_OK;
        }
    }
    if ((rc = ensure_text_tables(c))) return rc;
    HIPCHK(sync_all(c));
    TmpFree tmp;
    int *d_tc = nullptr, *d_C = nullptr, *d_hard = nullptr;
    unsigned long long *d_tb = nullptr;
    unsigned int *d_R = nullptr;
    const int nt = b->n_tiles256;
#define TRYS(x) do { if ((rc = (x)) != NATAC_OK) { for (int q_ = 0; q_ < 4; ++q_) dev_free(seg.p[q_]); dev_free(d_R); dev_free(d_C); return rc; } } while (0)
    natac_store::Seg seg{{nullptr, nullptr, nullptr, nullptr}, n_tracks, b->total_bp, c->device};
    TRYS(dev_alloc(&d_hard, 1)); tmp.keep(d_hard);
    TRYS(dev_alloc(&d_tc, (size_t)nt)); tmp.keep(d_tc);
    TRYS(dev_alloc(&d_tb, (size_t)nt + 1)); tmp.keep(d_tb);
    hipError_t e = hipMemsetAsync(d_hard, 0, sizeof(int), c->stream);
    int hard = 0;
    for (int i = 0; i < n_tracks && e == hipSuccess; ++i) {
        TextJob job;
        job.vals = b->d_track[tracks[i]]; job.out_off = b->d_out_off; job.chunk_len = b->d_len; job.tiles = b->d_tiles256; job.ntiles = nt;
        job.chrom_id = nullptr; job.chunk_start = nullptr; job.names = nullptr; job.name_off = nullptr; job.p10 = c->d_p10;
        job.write_zero = write_zero & 1; job.keep_before_nan = (write_zero >> 1) & 1;
        hipLaunchKernelGGL(tz_flags_count, dim3(nt), dim3(256), 0, c->stream, job, d_tc);
        TRYS(dev_scan(c, d_tc, (long long)nt, d_tb, tmp));
        unsigned long long nruns = 0;
        if ((e = hipMemcpyAsync(&nruns, d_tb + nt, sizeof nruns, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) break;
        dev_free(d_R); dev_free(d_C);
        d_R = nullptr; d_C = nullptr;
        TRYS(dev_alloc(&d_R, (size_t)nruns));
        TRYS(dev_alloc(&d_C, (size_t)nruns));
        hipLaunchKernelGGL(tz_scatter_runs, dim3(nt), dim3(256), 0, c->stream, job, d_tb, d_R, d_C);
        TRYS(dev_alloc(&seg.p[i], (size_t)b->total_bp));
        hipLaunchKernelGGL(tz_as_written, dim3((unsigned)((nruns + 255) / 256)), dim3(256), 0, c->stream, job, (long long)nruns, d_R, d_C,
                           seg.p[i], d_hard);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&hard, d_hard, sizeof hard, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_R); dev_free(d_C);
    if (e != hipSuccess) { for (int q = 0; q < 4; ++q) dev_free(seg.p[q]); return fail(NATAC_E_HIP, "store_adopt: %s", hipGetErrorString(e)); }
#undef TRYS
    if (n_hard) *n_hard = hard;
    if (hard) {      // a value the device cannot round like the text round trip would: nothing is adopted, the caller reads the file
        for (int q = 0; q < 4; ++q) dev_free(seg.p[q]);
        return NATAC_OK;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    s->segs.push_back(seg);
    s->bytes += (long long)n_tracks * b->total_bp * (long long)sizeof(double);
    *segment = (int64_t)s->segs.size() - 1;
    return NATAC_OK;
}

int natac_store_read(natac_store *s, natac_ctx *c, int64_t n, const int64_t *segment, const int64_t *offset, const int64_t *length,
                     int32_t slot, double *out, size_t out_values) {
    using namespace natac_textz;
    if (!s || !c || n < 0 || (n > 0 && (!segment || !offset || !length || !out))) return fail(NATAC_E_ARG, "null argument");
    if (n == 0) return NATAC_OK;
    HIPCHK(hipSetDevice(c->device));
    std::vector<StoreRegion> reg((size_t)n);
    long long total = 0;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        for (int64_t j = 0; j < n; ++j) {
            if (segment[j] < 0 || segment[j] >= (int64_t)s->segs.size()) return fail(NATAC_E_ARG, "region %lld: no segment %lld", (long long)j, (long long)segment[j]);
            const natac_store::Seg &g = s->segs[(size_t)segment[j]];
            if (slot < 0 || slot >= g.n_tracks) return fail(NATAC_E_ARG, "segment %lld holds %d tracks", (long long)segment[j], g.n_tracks);
            if (g.device != c->device) return fail(NATAC_E_ARG, "segment %lld lives on device %d", (long long)segment[j], g.device);
            if (offset[j] < 0 || length[j] < 0 || offset[j] + length[j] > g.n)
                return fail(NATAC_E_ARG, "region %lld: [%lld, +%lld) outside its segment (%lld values)", (long long)j, (long long)offset[j], (long long)length[j], g.n);
            reg[(size_t)j] = {g.p[slot] + offset[j], total, (long long)length[j]};
            total += length[j];
        }
    }
    if ((size_t)total != out_values) return fail(NATAC_E_ARG, "destination holds %zu values, the regions %lld", out_values, total);
    if (total == 0) return NATAC_OK;
    StoreRegion *d_reg = nullptr;
    double *d_out = nullptr;
    int rc;
    if ((rc = dev_upload(c, &d_reg, reg.data(), reg.size()))) return rc;
    if ((rc = dev_alloc(&d_out, (size_t)total))) { dev_free(d_reg); return rc; }
    hipLaunchKernelGGL(tz_store_gather, dim3((unsigned)n), dim3(256), 0, c->stream, d_reg, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_reg); dev_free(d_out);
    if (e != hipSuccess) return fail(NATAC_E_HIP, "store_read: %s", hipGetErrorString(e));
    return NATAC_OK;
}

int natac_store_info(natac_store *s, int64_t *n_segments, int64_t *bytes) {
    if (!s) return fail(NATAC_E_ARG, "store is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    long long by = 0;
    for (auto &g : s->segs) by += (long long)g.n_tracks * g.n * (long long)sizeof(double);
    if (n_segments) *n_segments = (int64_t)s->segs.size();
    if (bytes) *bytes = by;
    return NATAC_OK;
}

int natac_store_declined(natac_store *s, int64_t *n_declined) {
    if (!s || !n_declined) return fail(NATAC_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    *n_declined = s->declined;
    return NATAC_OK;
}

/* ---------------- shader-clock trace ---------------- */
}  // extern "C"

// one wave; lane 0 notes (constant-rate wall ticks since its start, shader cycle counter) every `interval` wall ticks and sleeps in
// between: the slope between two samples is the clock the CU ran at -- under whatever else the chip is executing meanwhile
__global__ void __launch_bounds__(64) natac_clock_sampler(long long *__restrict__ buf, int cap, long long interval) {
    if (threadIdx.x) return;
    const long long w0 = wall_clock64();
    long long next = w0;
    int i = 0;
    for (; i < cap; ++i) {
        long long w;
        do {
            __builtin_amdgcn_s_sleep(64);
            w = wall_clock64();
        } while (w < next);
        buf[2 + 2 * i] = w - w0;
        buf[3 + 2 * i] = clock64();
        next = w + interval;
        if (__hip_atomic_load(buf + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ++i; break; }
    }
    buf[0] = i;
}

extern "C" {

int natac_clock_trace_start(natac_ctx *c, int max_samples, int interval_us) {
    if (!c || max_samples < 2 || interval_us < 1) return fail(NATAC_E_ARG, "bad argument");
    if (c->ck_active) return fail(NATAC_E_STATE, "a clock trace is already running");
    HIPCHK(hipSetDevice(c->device));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) return fail(NATAC_E_HIP, "the device reports no wall clock rate");
    if (!c->ck_stream) HIPCHK(hipStreamCreateWithFlags(&c->ck_stream, hipStreamNonBlocking));
    if (!c->ck_start) HIPCHK(hipEventCreate(&c->ck_start));
    if (c->ck_cap < max_samples) {
        if (c->d_ck) (void)hipFree(c->d_ck);
        c->d_ck = nullptr;
        HIPCHK(hipMalloc((void **)&c->d_ck, (2 + 2 * (size_t)max_samples) * sizeof(long long)));
        c->ck_cap = max_samples;
    }
    HIPCHK(hipMemsetAsync(c->d_ck, 0, 2 * sizeof(long long), c->ck_stream));
    HIPCHK(hipEventRecord(c->ck_start, c->ck_stream));
    hipLaunchKernelGGL(natac_clock_sampler, dim3(1), dim3(64), 0, c->ck_stream, c->d_ck, max_samples,
                       (long long)khz * interval_us / 1000);
    HIPCHK(hipGetLastError());
    c->ck_iv.clear();
    c->ck_active = true;
    return NATAC_OK;
}

int natac_clock_trace_stop(natac_ctx *c, int64_t *n_samples, int64_t *n_intervals) {
    if (!c) return fail(NATAC_E_ARG, "ctx is NULL");
    if (!c->ck_active) return fail(NATAC_E_STATE, "no clock trace is running");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    prof_collect(c);                     // intervals of the launches profiled since the start
    const long long one = 1;
    HIPCHK(hipMemcpy(c->d_ck + 1, &one, sizeof one, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(c->ck_stream));
    long long n = 0;
    HIPCHK(hipMemcpy(&n, c->d_ck, sizeof n, hipMemcpyDeviceToHost));
    c->ck_active = false;
    if (n_samples) *n_samples = n;
    if (n_intervals) *n_intervals = (int64_t)c->ck_iv.size();
    return NATAC_OK;
}

int natac_clock_trace_fetch(natac_ctx *c, int64_t n_samples, double *t_ms, double *cycles, int64_t n_intervals, int32_t *iv_kernel,
                            double *iv_t0_ms, double *iv_t1_ms) {
    if (!c || (n_samples > 0 && (!t_ms || !cycles)) || (n_intervals > 0 && (!iv_kernel || !iv_t0_ms || !iv_t1_ms)))
        return fail(NATAC_E_ARG, "null argument");
    if (c->ck_active) return fail(NATAC_E_STATE, "stop the clock trace first");
    if (n_samples > c->ck_cap || n_intervals > (int64_t)c->ck_iv.size()) return fail(NATAC_E_ARG, "more than was recorded");
    HIPCHK(hipSetDevice(c->device));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    std::vector<long long> h(2 * (size_t)n_samples);
    if (n_samples) HIPCHK(hipMemcpy(h.data(), c->d_ck + 2, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n_samples; ++i) {
        t_ms[i] = (double)h[2 * i] / (double)khz;
        cycles[i] = (double)h[2 * i + 1];
    }
    for (int64_t i = 0; i < n_intervals; ++i) {
        iv_kernel[i] = c->ck_iv[(size_t)i].k;
        iv_t0_ms[i] = c->ck_iv[(size_t)i].t0;
        iv_t1_ms[i] = c->ck_iv[(size_t)i].t1;
    }
    return NATAC_OK;
}

}  // extern "C"
