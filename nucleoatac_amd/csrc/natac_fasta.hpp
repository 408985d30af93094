// natac_fasta.hpp -- native FASTA -> per-record upper-case byte arrays (host C++17, multi-threaded).
//
// The reference fetches the sequence of every region through pysam.FastaFile (pyatac/bias.py:85-92, pyatac/seq.py:11-22).  The batched
// drivers score the Tn5 bias of whole sub-batches from sequence windows (pipeline._seq_windows), so they keep the genome as one byte
// array per record; a Python line loop over a 3-Gbp FASTA takes minutes, this loader runs at memory speed:
//   1. the file is read whole; T threads find the header lines ('>' at a line start) of their part;
//   2. per record, T threads count the sequence bytes (everything but '\n' / '\r') of their part of the record's text, a prefix sum
//      gives the output offsets, and the same threads copy the bytes upper-cased (blanks inside sequence lines are dropped too).
// Records keep the name up to the first white space, like pysam / faidx.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "natac_cores.hpp"

namespace natac_fastaio {

struct Record {
    std::string name;
    size_t text_begin = 0, text_end = 0;     // bytes of the file holding this record's sequence lines
    int64_t length = 0;                      // sequence bytes
};

struct Fasta {
    std::vector<unsigned char> data;
    std::vector<Record> recs;
    int n_threads = 1;
};

template <class F>
inline void parallel_parts(int T, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

inline int64_t count_seq_bytes(const unsigned char *p, const unsigned char *e) {
    int64_t n = 0;
    for (; p < e; ++p) n += (*p != '\n') & (*p != '\r') & (*p != ' ') & (*p != '\t');
    return n;
}

// returns nullptr + err on failure
inline Fasta *load(const char *path, int n_threads, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return nullptr; }
    Fasta *fa = new Fasta();
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (size < 0) { std::fclose(f); delete fa; err = "cannot size the file"; return nullptr; }
    fa->data.resize((size_t)size);
    if (size && std::fread(fa->data.data(), 1, (size_t)size, f) != (size_t)size) { std::fclose(f); delete fa; err = "short read"; return nullptr; }
    std::fclose(f);
    const unsigned char *d = fa->data.data();
    const size_t n = fa->data.size();
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, n / ((size_t)1 << 20) + 1));
    fa->n_threads = T;
    // ---- header lines
    std::vector<std::vector<size_t>> found((size_t)T);
    parallel_parts(T, [&](int t) {
        const size_t a = n * (size_t)t / (size_t)T, b = n * ((size_t)t + 1) / (size_t)T;
        const unsigned char *p = d + a, *e = d + b;
        while (p < e) {
            const unsigned char *q = (const unsigned char *)std::memchr(p, '>', (size_t)(e - p));
            if (!q) break;
            if (q == d || q[-1] == '\n') found[(size_t)t].push_back((size_t)(q - d));
            p = q + 1;
        }
    });
    std::vector<size_t> heads;
    for (auto &v : found) heads.insert(heads.end(), v.begin(), v.end());
    for (size_t i = 0; i < heads.size(); ++i) {
        Record r;
        const unsigned char *h = d + heads[i] + 1;
        const unsigned char *nl = (const unsigned char *)std::memchr(h, '\n', n - (heads[i] + 1));
        const unsigned char *he = nl ? nl : d + n;
        const unsigned char *w = h;
        while (w < he && *w != ' ' && *w != '\t' && *w != '\r') ++w;
        r.name.assign((const char *)h, (size_t)(w - h));
        r.text_begin = nl ? (size_t)(nl + 1 - d) : n;
        r.text_end = i + 1 < heads.size() ? heads[i + 1] : n;
        fa->recs.push_back(std::move(r));
    }
    // ---- sequence lengths: one count per (thread, record part); parts are cut by bytes over ALL records so that one big
    // chromosome and a thousand scaffolds load equally well
    for (auto &r : fa->recs) {
        const size_t len = r.text_end - r.text_begin;
        const int Tr = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, len / ((size_t)1 << 20) + 1));
        std::vector<int64_t> cnt((size_t)Tr, 0);
        parallel_parts(Tr, [&](int t) {
            cnt[(size_t)t] = count_seq_bytes(d + r.text_begin + len * (size_t)t / (size_t)Tr, d + r.text_begin + len * ((size_t)t + 1) / (size_t)Tr);
        });
        for (auto c : cnt) r.length += c;
    }
    return fa;
}

// out[0 .. length): the record's sequence, upper case
inline void read_record(const Fasta *fa, size_t i, unsigned char *out) {
    const Record &r = fa->recs[i];
    const unsigned char *d = fa->data.data();
    const size_t len = r.text_end - r.text_begin;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)fa->n_threads, len / ((size_t)1 << 20) + 1));
    std::vector<int64_t> off((size_t)T + 1, 0);
    parallel_parts(T, [&](int t) {
        off[(size_t)t + 1] = count_seq_bytes(d + r.text_begin + len * (size_t)t / (size_t)T, d + r.text_begin + len * ((size_t)t + 1) / (size_t)T);
    });
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] += off[(size_t)t];
    parallel_parts(T, [&](int t) {
        const unsigned char *p = d + r.text_begin + len * (size_t)t / (size_t)T, *e = d + r.text_begin + len * ((size_t)t + 1) / (size_t)T;
        unsigned char *o = out + off[(size_t)t];
        for (; p < e; ++p) {
            const unsigned char c = *p;
            if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
            *o++ = (unsigned char)((c >= 'a' && c <= 'z') ? c - 32 : c);
        }
    });
}

}  // namespace natac_fastaio
