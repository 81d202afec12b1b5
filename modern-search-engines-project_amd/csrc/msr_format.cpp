// Host-side result formatting (host C++ only: compiled without an offload architecture): the batch lines of search_api.py:290,
//     f"{query_num}\t{rank}\t{url}\t{score:.3f}"
// for the final lists of a whole batch in one call, straight from the arrays the engine returned -- the reference builds a
// dict and two f-strings per result in Python (search_api.py:276-292); at hundreds of queries per batch that loop, not the
// GPU, is the batch's run time.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/msretr.h"

namespace {

// "%.3f" of a double, digit for digit what Python's format(x, ".3f") and glibc's printf give: both round the EXACT binary
// value half-to-even at the third decimal.  x = m 2^e with a 53-bit integer m, so 1000 x = (1000 m) / 2^-e is an exact
// 63-bit integer over a power of two: quotient, remainder and the half-way comparison are integer operations.
static inline int fmt3(double x, char* out) {
    const double ax = fabs(x);
    if (!(ax < 1.0e12)) return snprintf(out, 40, "%.3f", x);          // huge, inf, nan
    uint64_t v = 0;
    if (ax != 0.0) {
        int ex;
        const double fr = frexp(ax, &ex);                               // ax = fr 2^ex, fr in [0.5, 1)
        const uint64_t M = (uint64_t)ldexp(fr, 53) * 1000u;             // < 2^63
        const int k = 53 - ex;                                          // 1000 ax = M / 2^k
        if (k <= 0) {
            v = M << -k;                                                // (ax < 1e12: no overflow)
        } else if (k < 64) {
            v = M >> k;
            const uint64_t rem = M & ((1ull << k) - 1), half = 1ull << (k - 1);
            if (rem > half || (rem == half && (v & 1))) ++v;
        }                                                               // k >= 64: 1000 ax < 1/2 -> 0
    }
    char* p = out;
    if (signbit(x)) *p++ = '-';                                         // (Python prints -0.000 for a negative that rounds to 0)
    uint64_t ip = v / 1000;
    const unsigned fr3 = (unsigned)(v % 1000);
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    *p++ = (char)('0' + fr3 / 100); *p++ = (char)('0' + fr3 / 10 % 10); *p++ = (char)('0' + fr3 % 10);
    return (int)(p - out);
}

static inline int fmt_u(uint32_t v, char* out) {
    char tmp[12];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}

}  // namespace

extern "C" int64_t msr_format_lines(const char* qnum_blob, const int64_t* qnum_off, int32_t n_queries, const int32_t* doc,
                                    const double* score, const int32_t* n, int32_t stride, const char* url_blob,
                                    const int64_t* url_off, int64_t n_docs, char* out, int64_t capacity) {
    if (!qnum_blob || !qnum_off || !doc || !score || !n || !url_off || n_queries < 0 || stride < 0 || capacity < 0 || (capacity && !out))
        return INT64_MIN;
    // pass 1: an upper bound of the bytes (the formatted rank and score take at most 10 + 24 characters for ranks < 2^32 and
    // |score| < 1e12; snprintf's output for anything larger is bounded by 40)
    int64_t need = 0;
    for (int32_t q = 0; q < n_queries; ++q) {
        const int64_t ql = qnum_off[q + 1] - qnum_off[q];
        const int32_t cnt = n[q] < 0 ? 0 : (n[q] > stride ? stride : n[q]);
        for (int32_t r = 0; r < cnt; ++r) {
            const int32_t d = doc[(int64_t)q * stride + r];
            const int64_t ul = (d >= 0 && d < n_docs && url_blob) ? url_off[d + 1] - url_off[d] : 0;
            need += ql + ul + 56;
        }
    }
    if (need > capacity) return -need;
    char* p = out;
    for (int32_t q = 0; q < n_queries; ++q) {
        const char* qs = qnum_blob + qnum_off[q];
        const int64_t ql = qnum_off[q + 1] - qnum_off[q];
        const int32_t cnt = n[q] < 0 ? 0 : (n[q] > stride ? stride : n[q]);
        for (int32_t r = 0; r < cnt; ++r) {
            const int64_t i = (int64_t)q * stride + r;
            const int32_t d = doc[i];
            memcpy(p, qs, (size_t)ql); p += ql;
            *p++ = '\t';
            p += fmt_u((uint32_t)(r + 1), p);
            *p++ = '\t';
            if (d >= 0 && d < n_docs && url_blob) {
                const int64_t ul = url_off[d + 1] - url_off[d];
                memcpy(p, url_blob + url_off[d], (size_t)ul); p += ul;
            }
            *p++ = '\t';
            p += fmt3(score[i], p);
            *p++ = '\n';
        }
    }
    return (int64_t)(p - out);
}
