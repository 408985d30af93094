// host-side sweep of the device formatter's arithmetic (natac_textfmt.hpp is __host__ __device__): fmt_py2_float against the
// native writer's formatter (natac_writer.hpp: exact 128-bit fast path + std::to_chars), which tests/test_writer.py pins to
// python's '%.12g'.   build: g++ -O2 -std=c++17 -I nucleoatac_amd/csrc tools/test_textfmt_host.cpp -o /tmp/test_textfmt -lz -lpthread
#include "natac_textfmt.hpp"
#include "natac_writer.hpp"
#include <cstdio>
#include <cstring>
#include <random>

int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 20000000;
    std::mt19937_64 rng(12345);
    long long bad = 0, hard_total = 0, n = 0;
    auto check = [&](double v) {
        if (v != v) return;
        char a[64], b[64];
        int hard = 0;
        char *ea = natac_text::fmt_py2_float(a, v, natac_text::H_P10, &hard);
        char *eb = natac_writer::fmt_py2_float(b, v);
        hard_total += hard;
        ++n;
        {   // the register sink the kernels use (three 64-bit words) holds the same bytes as the pointer sink
            natac_text::RegSink rs;
            int h2 = 0;
            natac_text::fmt_py2_float_to(rs, v, natac_text::H_P10, &h2);
            char c[24];
            std::memcpy(c, &rs.w0, 8); std::memcpy(c + 8, &rs.w1, 8); std::memcpy(c + 16, &rs.w2, 8);
            if (rs.n != ea - a || std::memcmp(c, a, rs.n)) { if (bad < 20) std::printf("REG-SINK MISMATCH %.17g\n", v); ++bad; }
        }
        if (hard) return;
        if (ea - a != eb - b || std::memcmp(a, b, ea - a)) {
            if (bad < 20) { *ea = 0; *eb = 0; std::printf("MISMATCH %.17g: device-fmt '%s' vs native '%s'\n", v, a, b); }
            ++bad;
        }
    };
    const double special[] = {0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e-4, 9.99999999999e-5, 0.0001, 0.00011, 1e11, 1e12, 999999999999.0, 999999999999.5,
                              999999999999.4999, 99999999999.95, 1e-5, 1e-7, 123456789012.5, 1234567890125.0, 12345678901250.0, 1e22, 1e23, 5e-324, 2.2250738585072014e-308,
                              1.7976931348623157e308, 1.0 / 0.0, -1.0 / 0.0, 0.3, 2.5, 0.125, 1e100, 1e-100, 9.5, 10.5, 100000000000.5, 0.000123456789012345,
                              123456.789012, 1e15, 1e16, 9007199254740993.0, 0.30000000000000004, 1e-310, 4.9e-324, 1e-44, 1e-45, 9.999999999995e-45};
    for (double v : special) { check(v); check(-v); }
    for (long long i = 0; i < N; ++i) {
        const uint64_t r = rng();
        double v;
        switch (i % 6) {
            case 0: { union { uint64_t u; double d; } c; c.u = r; v = c.d; break; }                         // any bit pattern
            case 1: v = (double)(r >> 11) * (1.0 / 9007199254740992.0); break;                                // [0, 1)
            case 2: v = std::ldexp((double)(r >> 11), (int)(r % 200) - 150); break;                           // wide dyadic
            case 3: { const double t = (double)((r >> 20) % 2000000000000ull); v = (t + 0.5) / std::pow(10.0, (double)(r % 14)); break; }  // near .5 boundaries
            case 4: v = std::round((double)(r % 4000000000000ull)) / 1000.0; break;                           // 3-decimals
            default: v = (double)(long long)(r % 2000000) * std::pow(10.0, (double)((int)(r >> 40) % 40 - 20)); break;   // short decimals, all scales
        }
        check(v);
    }
    std::printf("%lld values, %lld mismatches, %lld undecided (hard)\n", n, bad, hard_total);
    return bad ? 1 : 0;
}
