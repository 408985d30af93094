// natac_bedtab.hpp -- a (gzipped) BED-like table read natively into columns (host C++): chromosome, start, end and chosen float columns.
//
// `nucleoatac merge` reads occpeaks.bed.gz and nucpos.bed.gz row by row (reference nucleoatac/merge.py:36-64: split, int(), float()):
// 10^5-10^6 rows per genome, one second of interpreter time per 300,000.  Here the file is inflated with zlib's gz* reader (BGZF,
// plain gzip and plain text alike), lines are split at tabs, coordinates parsed as integers and the requested columns as doubles
// with natac_tabix::parse_double (correctly rounded like float(): one exact operation on the fast path, strtod otherwise).
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "natac_tabix.hpp"

namespace natac_bedtabio {

struct Table {
    std::vector<std::string> names;          // chromosome names in order of first appearance
    std::vector<int32_t> chrom_id;
    std::vector<int64_t> start, end;
    std::vector<double> vals;                // [n_rows][n_cols]
    int n_cols = 0;
};

// 0 ok, 1 cannot open, 2 read error, 3 a row has too few columns (err names the line)
inline int load(const char *path, const int32_t *cols, int n_cols, Table *t, std::string &err) {
    gzFile f = gzopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return 1; }
    gzbuffer(f, 1 << 20);
    std::string text;
    std::vector<char> buf((size_t)4 << 20);
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(f); err = "read error"; return 2; }
        if (got == 0) break;
        text.append(buf.data(), (size_t)got);
    }
    gzclose(f);
    t->n_cols = n_cols;
    int max_col = 2;
    for (int c = 0; c < n_cols; ++c) max_col = cols[c] > max_col ? cols[c] : max_col;
    std::unordered_map<std::string, int32_t> ids;
    std::string last_name;
    int32_t last_id = -1;
    const char *p = text.data(), *e = p + text.size();
    int64_t lineno = 0;
    std::vector<const char *> fld((size_t)max_col + 2);
    while (p < e) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        ++lineno;
        if (le > p) {
            int nf = 0;
            fld[(size_t)nf++] = p;
            for (const char *q = p; q < le && nf <= max_col + 1; ++q)
                if (*q == '\t') fld[(size_t)nf++] = q + 1;
            if (nf <= max_col) { err = "line " + std::to_string(lineno) + " has fewer than " + std::to_string(max_col + 1) + " columns"; return 3; }
            const size_t nlen = (size_t)(fld[1] - 1 - fld[0]);
            if (last_id < 0 || last_name.size() != nlen || std::memcmp(last_name.data(), fld[0], nlen) != 0) {
                last_name.assign(fld[0], nlen);
                auto it = ids.find(last_name);
                if (it == ids.end()) { it = ids.emplace(last_name, (int32_t)t->names.size()).first; t->names.push_back(last_name); }
                last_id = it->second;
            }
            t->chrom_id.push_back(last_id);
            t->start.push_back(natac_tabix::parse_int(fld[1]));
            t->end.push_back(natac_tabix::parse_int(fld[2]));
            for (int c = 0; c < n_cols; ++c) t->vals.push_back(natac_tabix::parse_double(fld[(size_t)cols[c]]));
        }
        if (!nl) break;
        p = nl + 1;
    }
    return 0;
}

}  // namespace natac_bedtabio
