// natac_textfmt.hpp -- Track.write_track (pyatac/tracks.py:37-74) ON THE DEVICE: run-length detection, the reference's
// run-before-NaN rule, python-2 `str(float)` (= '%.12g', ".0" appended to integral text) and the bedGraph line layout as
// HIP kernels, so that a per-base float64 track leaves the GPU as the exact bytes the reference's writer processes would
// produce.  The host-side formatter (natac_writer.hpp) manages 30-60 Mbp/s per track on the box's
// 16 cores; the GPU formats a 212-Mbp track in milliseconds.
//
// '%.12g' on the device: |v| = m * 2^e2 exactly (m normalised to 64 bits); with s = 11 - floor(log10 |v|) the twelve significant
// digits are D = round_half_even(|v| * 10^s).  10^s comes from a table of 128-bit truncated mantissas (natac_pow10.inc, generated
// with exact integer arithmetic by tools/gen_pow10_table.py) and m * 10^s is formed exactly as a 192-bit product.  For
// 0 <= s <= 55 (1e-44 <= |v| < 1e12: every value a track holds in practice) the table entry is exact, so the product, its
// remainder and therefore ties are exact -- the result is the correctly rounded one that printf / to_chars give.  Outside that
// range the entry is a truncation (relative error < 2^-127): the rounding decision can only differ from the exact one when
// the 88+ bits below the half bit are all ones; that case (probability ~2^-88 per value) is COUNTED (`hard`), never guessed --
// the caller falls back to the host formatter for that track.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NATAC_HD __host__ __device__
#else
#define NATAC_HD
#endif

namespace natac_text {

struct P10 { uint64_t hi, lo; int e; int exact; };

static const P10 H_P10[] = {      // host copy; natac_api.hip uploads it once per context for the kernels
#include "natac_pow10.inc"
};

NATAC_HD inline uint64_t mul64(uint64_t a, uint64_t b, uint64_t *hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    *hi = __umul64hi(a, b);
    return a * b;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    *hi = (uint64_t)(p >> 64);
    return (uint64_t)p;
#endif
}

NATAC_HD inline int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

// decimal digits of v (1 for 0); genome coordinates fit 32 bits: that path has no 64-bit division
NATAC_HD inline int digits_u64(uint64_t v) {
    if (v < 4294967296ull) {
        const uint32_t u = (uint32_t)v;
        return u < 10u ? 1 : u < 100u ? 2 : u < 1000u ? 3 : u < 10000u ? 4 : u < 100000u ? 5 : u < 1000000u ? 6 : u < 10000000u ? 7
               : u < 100000000u ? 8 : u < 1000000000u ? 9 : 10;
    }
    int n = 0;
    do { ++n; v /= 10; } while (v);
    return n;
}
NATAC_HD inline char *put_u64(char *p, uint64_t v) {
    const int n = digits_u64(v);
    if (v < 4294967296ull) {
        uint32_t u = (uint32_t)v;
        for (int i = n - 1; i >= 0; --i) { const uint32_t q = u / 10u; p[i] = (char)('0' + (u - q * 10u)); u = q; }
    } else {
        for (int i = n - 1; i >= 0; --i) { const uint64_t q = v / 10; p[i] = (char)('0' + (v - q * 10)); v = q; }
    }
    return p + n;
}
NATAC_HD inline char *put_i64(char *p, long long v) {
    if (v < 0) { *p++ = '-'; return put_u64(p, (uint64_t)(-(v + 1)) + 1); }
    return put_u64(p, (uint64_t)v);
}
NATAC_HD inline int digits_i64(long long v) {
    return v < 0 ? 1 + digits_u64((uint64_t)(-(v + 1)) + 1) : digits_u64((uint64_t)v);
}

// Where the characters of a value go.  PtrSink: a byte buffer (host writer, tests).  RegSink: three 64-bit words held in registers --
// on the device a `char buf[24]` written at run-time positions lives in scratch memory (32 bytes per lane in tz_line_len and
// tz_format_values until round 6); here a character is OR-ed into the word its position selects.
struct PtrSink {
    char *p;
    NATAC_HD void put(char c) { *p++ = c; }
};
struct RegSink {
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    int n = 0;
    NATAC_HD void put(char c) {
        const uint64_t v = (uint64_t)(unsigned char)c << ((n & 7) * 8);
        const int k = n >> 3;
        w0 |= k == 0 ? v : 0ull;
        w1 |= k == 1 ? v : 0ull;
        w2 |= k == 2 ? v : 0ull;
        ++n;
    }
};

// python-2 str(float) of a non-NaN double: '%.12g' + ".0" for integral text, character by character into `out`.  *hard is incremented
// when the rounding could not be decided from the truncated table entry (see the header).  tab = the P10 table.
template <class Sink>
NATAC_HD inline void fmt_py2_float_to(Sink &out, double v, const P10 *tab, int *hard) {
    union { double d; uint64_t u; } cv;
    cv.d = v;
    const uint64_t bits = cv.u;
    const bool neg = (bits >> 63) != 0;
    const int ef = (int)((bits >> 52) & 0x7ff);
    const uint64_t frac = bits & 0xfffffffffffffull;
    if (ef == 0x7ff) {
        if (frac) { out.put('n'); out.put('a'); out.put('n'); return; }
        if (neg) out.put('-');
        out.put('i'); out.put('n'); out.put('f');
        return;
    }
    if (neg) out.put('-');
    if (ef == 0 && frac == 0) { out.put('0'); out.put('.'); out.put('0'); return; }
    uint64_t m = ef ? (frac | (1ull << 52)) : frac;
    int e2 = ef ? ef - 1075 : -1074;
    const int lz = clz64(m);
    m <<= lz;
    e2 -= lz;                                            // |v| = m 2^e2, 2^63 <= m < 2^64
    int e10 = ((e2 + 63) * 78913) >> 18;                 // floor((e2 + 63) log10 2): floor(log10 |v|) or one less
    uint64_t D = 0;
    for (int it = 0; it < 3; ++it) {
        const int s = 11 - e10;
        const P10 T = tab[s - NATAC_P10_SMIN];
        uint64_t l1, h1;
        const uint64_t l0 = mul64(m, T.lo, &l1);
        const uint64_t h0 = mul64(m, T.hi, &h1);
        const uint64_t p0 = l0;
        const uint64_t p1 = l1 + h0;
        const uint64_t p2 = h1 + (p1 < l1 ? 1 : 0);      // m * T = p2:p1:p0 (192 bits), |v| 10^s = that * 2^(e2 + T.e)
        const int sh = -(e2 + T.e) - 128;                // D = p2 >> sh, 22 <= sh <= 63 when e10 is right (or one less)
        const uint64_t D0 = p2 >> sh;
        if (D0 >= 1000000000000ull) { ++e10; continue; }  // the guess was one too small
        const uint64_t hb = (p2 >> (sh - 1)) & 1;
        const uint64_t mask = (1ull << (sh - 1)) - 1;
        const uint64_t rest = p2 & mask;
        bool up;
        if (T.exact) up = hb && ((rest | p1 | p0) != 0 || (D0 & 1));
        else {
            up = hb != 0;
            if (!hb && rest == mask && p1 == ~0ull) ++*hard;          // a carry out of the truncated tail would flip the decision
        }
        D = D0 + (up ? 1 : 0);
        if (D < 100000000000ull) { --e10; continue; }    // (cannot happen with the floor guess; kept as a guard)
        if (D >= 1000000000000ull) { D = 100000000000ull; ++e10; }   // 999999999999.5+ rounded up to 10^12
        break;
    }
    // the twelve digits as one packed word, digit i (0 = first) in bits [4 (11 - i), 4 (11 - i) + 4): no indexed array
    uint64_t bcd = 0;
    {
        const uint32_t hi6 = (uint32_t)(D / 1000000ull), lo6 = (uint32_t)(D - (uint64_t)hi6 * 1000000ull);     // two 6-digit halves: 32-bit divisions
        uint32_t u = lo6;
        for (int i = 0; i < 6; ++i) { const uint32_t q = u / 10u; bcd |= (uint64_t)(u - q * 10u) << (4 * i); u = q; }
        u = hi6;
        for (int i = 6; i < 12; ++i) { const uint32_t q = u / 10u; bcd |= (uint64_t)(u - q * 10u) << (4 * i); u = q; }
    }
    auto dg = [&](int i) -> char { return (char)('0' + (int)((bcd >> (4 * (11 - i))) & 0xf)); };
    int nd = 12;
    while (nd > 1 && ((bcd >> (4 * (12 - nd))) & 0xf) == 0) --nd;            // %g strips trailing zeros
    if (e10 < -4 || e10 >= 12) {                         // scientific: d[.ddd]e+XX (at least two exponent digits); has an 'e': no ".0"
        out.put(dg(0));
        if (nd > 1) { out.put('.'); for (int i = 1; i < nd; ++i) out.put(dg(i)); }
        out.put('e');
        int x = e10;
        if (x < 0) { out.put('-'); x = -x; } else out.put('+');
        if (x >= 100) { out.put((char)('0' + x / 100)); x %= 100; out.put((char)('0' + x / 10)); out.put((char)('0' + x % 10)); }
        else { out.put((char)('0' + x / 10)); out.put((char)('0' + x % 10)); }
    } else if (e10 >= 0) {
        const int ni = e10 + 1;                          // integer digits
        for (int i = 0; i < ni; ++i) out.put((i < nd) ? dg(i) : '0');
        out.put('.');
        if (nd > ni) { for (int i = ni; i < nd; ++i) out.put(dg(i)); }
        else out.put('0');                               // integral text gets ".0"
    } else {
        out.put('0'); out.put('.');
        for (int i = 0; i < -e10 - 1; ++i) out.put('0');
        for (int i = 0; i < nd; ++i) out.put(dg(i));
    }
}
// into a byte buffer; returns the end pointer
NATAC_HD inline char *fmt_py2_float(char *p, double v, const P10 *tab, int *hard) {
    PtrSink s{p};
    fmt_py2_float_to(s, v, tab, hard);
    return s.p;
}

constexpr int MAX_VALUE_CHARS = 24;     // "-1.23456789012e-308" is 19

// The double a reader gets back from fmt_py2_float's text: float('%.12g' % v) -- v rounded (half-even, exactly, as above) to twelve
// significant digits D x 10^(e10 - 11), then that decimal rounded to the nearest double.  For 0 <= |e10 - 11| <= 22 both D (< 10^12)
// and the power of ten are exact doubles, so ONE IEEE division / multiplication is the correctly rounded result (what strtod and
// the native tabix reader's fast path return).  Outside that range (|v| < 1e-11 or >= 1e34), or when the twelfth digit could not be
// decided, *hard is incremented and the value is returned unrounded: the caller must not use the array as "what the file holds".
NATAC_HD inline double round12(double v, const P10 *tab, int *hard) {
    union { double d; uint64_t u; } cv;
    cv.d = v;
    const uint64_t bits = cv.u;
    const int ef = (int)((bits >> 52) & 0x7ff);
    const uint64_t frac = bits & 0xfffffffffffffull;
    if (ef == 0x7ff || (ef == 0 && frac == 0)) return v;             // nan, inf, +-0
    uint64_t m = ef ? (frac | (1ull << 52)) : frac;
    int e2 = ef ? ef - 1075 : -1074;
    const int lz = clz64(m);
    m <<= lz;
    e2 -= lz;
    int e10 = ((e2 + 63) * 78913) >> 18;
    uint64_t D = 0;
    int h = 0;
    for (int it = 0; it < 3; ++it) {                                  // the digit loop of fmt_py2_float
        const int s = 11 - e10;
        if (s < NATAC_P10_SMIN || s > NATAC_P10_SMAX) { ++*hard; return v; }
        const P10 T = tab[s - NATAC_P10_SMIN];
        uint64_t l1, h1;
        const uint64_t l0 = mul64(m, T.lo, &l1);
        const uint64_t h0 = mul64(m, T.hi, &h1);
        const uint64_t p0 = l0;
        const uint64_t p1 = l1 + h0;
        const uint64_t p2 = h1 + (p1 < l1 ? 1 : 0);
        const int sh = -(e2 + T.e) - 128;
        const uint64_t D0 = p2 >> sh;
        if (D0 >= 1000000000000ull) { ++e10; continue; }
        const uint64_t hb = (p2 >> (sh - 1)) & 1;
        const uint64_t mask = (1ull << (sh - 1)) - 1;
        const uint64_t rest = p2 & mask;
        bool up;
        h = 0;
        if (T.exact) up = hb && ((rest | p1 | p0) != 0 || (D0 & 1));
        else {
            up = hb != 0;
            if (!hb && rest == mask && p1 == ~0ull) h = 1;
        }
        D = D0 + (up ? 1 : 0);
        if (D < 100000000000ull) { --e10; continue; }
        if (D >= 1000000000000ull) { D = 100000000000ull; ++e10; }
        break;
    }
    const int s = 11 - e10;                                           // value = D / 10^s
    if (h || s > 22 || s < -22) { ++*hard; return v; }
    const double P[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19,
                          1e20, 1e21, 1e22};
    const double d = (double)D;
    const double r = s >= 0 ? d / P[s] : d * P[-s];
    return (bits >> 63) ? -r : r;
}

}  // namespace natac_text
