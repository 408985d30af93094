// natac_pack.hpp -- host-side packing of a chunk list into the CSR fragment arrays natac_batch_create takes.
//
// For every chunk: the forward proper-pair reads of its chromosome with pos in [start - margin - shift, end + margin)
// (a superset of the reference's per-chunk bamHandle.fetch, pyatac/fragments.pyx:21-25), shifted like the reference
// (l = pos + 4, n = |tlen| - 8 when atac; fragments.pyx:26-34), relative to the chunk start and ordered by centre
// l + (n - 1) // 2 (stable), which is the order every device kernel expects (nucleoatac_amd/packing.py).
#pragma once
#include "natac_cores.hpp"
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace natac_pack {

inline int64_t lower_bound64(const int64_t *a, int64_t n, int64_t key) {
    return (int64_t)(std::lower_bound(a, a + n, key) - a);
}
inline int32_t fhalf(int32_t x) { return x >= 0 ? x / 2 : -((-x + 1) / 2); }

// count pass: frag_off[0..nc] and first[i] = index of the chunk's first read in its chromosome's arrays
inline void count(int32_t nc, const int64_t *cstart, const int64_t *cend, const int32_t *chrom_id, const int64_t *const *pos,
                  const int64_t *n_per_chrom, int64_t margin, int shift, int64_t *frag_off, int64_t *first) {
    frag_off[0] = 0;
    for (int32_t i = 0; i < nc; ++i) {
        const int c = chrom_id[i];
        int64_t a = 0, b = 0;
        if (c >= 0) {
            a = lower_bound64(pos[c], n_per_chrom[c], cstart[i] - margin - shift);
            b = lower_bound64(pos[c], n_per_chrom[c], cend[i] + margin);
            if (b < a) b = a;
        }
        first[i] = a;
        frag_off[i + 1] = frag_off[i] + (b - a);
    }
}

inline void fill(int32_t nc, const int64_t *cstart, const int32_t *chrom_id, const int64_t *const *pos, const int64_t *const *tlen,
                 const int64_t *frag_off, const int64_t *first, int shift, int trim, int32_t *lpos, int32_t *ilen, int n_threads) {
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    n_threads = std::max(1, std::min(n_threads, std::max(1, nc / 64)));
    auto work = [&](int t) {
        std::vector<std::pair<int32_t, int32_t>> key;      // (centre, rank) -> stable order
        std::vector<int32_t> tl, tn;
        const int32_t i0 = (int32_t)((int64_t)nc * t / n_threads), i1 = (int32_t)((int64_t)nc * (t + 1) / n_threads);
        for (int32_t i = i0; i < i1; ++i) {
            const int64_t m = frag_off[i + 1] - frag_off[i];
            if (m == 0) continue;
            const int c = chrom_id[i];
            const int64_t *p = pos[c] + first[i], *tt = tlen[c] + first[i];
            key.resize((size_t)m); tl.resize((size_t)m); tn.resize((size_t)m);
            bool sorted = true;
            for (int64_t k = 0; k < m; ++k) {
                const int32_t l = (int32_t)(p[k] + shift - cstart[i]);
                const int64_t a = tt[k] < 0 ? -tt[k] : tt[k];
                const int32_t n = (int32_t)(a - trim);
                tl[k] = l; tn[k] = n;
                key[k] = {l + fhalf(n - 1), (int32_t)k};
                if (k && key[k].first < key[k - 1].first) sorted = false;
            }
            if (!sorted) std::sort(key.begin(), key.end());   // pairs: ties keep the original (position) order
            int32_t *ol = lpos + frag_off[i], *on = ilen + frag_off[i];
            for (int64_t k = 0; k < m; ++k) { ol[k] = tl[key[k].second]; on[k] = tn[key[k].second]; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

}  // namespace natac_pack
