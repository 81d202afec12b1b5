// Host-side result formatting (host C++ only: compiled without an offload architecture): the batch lines of search_api.py:290,
//     f"{query_num}\t{rank}\t{url}\t{score:.3f}"
// for the final lists of a whole batch in one call, straight from the arrays the engine returned -- the reference builds a
// dict and two f-strings per result in Python (search_api.py:276-292); at hundreds of queries per batch that loop, not the
// GPU, is the batch's run time.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/msretr.h"

namespace {

// "%.3f" of a double, digit for digit what Python's format(x, ".3f") and glibc's printf give: both round the EXACT binary
// value half-to-even at the third decimal.  x = m / 2^k with a 53-bit integer m (read off the bits), so 1000 x = (1000 m) / 2^k
// is an exact 63-bit integer over a power of two: quotient, remainder and the half-way comparison are integer operations.
static inline int fmt3(double x, char* out) {
    const double ax = fabs(x);
    if (!(ax < 1.0e12)) return snprintf(out, 40, "%.3f", x);          // huge, inf, nan
    uint64_t v = 0;
    if (ax != 0.0) {
        uint64_t bits;
        memcpy(&bits, &ax, 8);
        const int ef = (int)(bits >> 52);                               // biased exponent (sign is clear)
        const uint64_t m = ef ? ((bits & ((1ull << 52) - 1)) | (1ull << 52)) : (bits & ((1ull << 52) - 1));
        const int k = ef ? 1075 - ef : 1074;                            // ax = m / 2^k
        const uint64_t M = m * 1000u;                                   // < 2^63: 1000 ax = M / 2^k
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

// format queries [q0, q1) into p (which has room for their upper bound); returns the end
static char* format_range(const char* qnum_blob, const int64_t* qnum_off, int32_t q0, int32_t q1, const int32_t* doc,
                          const double* score, const int32_t* n, int32_t stride, const char* url_blob, const int64_t* url_off,
                          int64_t n_docs, int64_t max_url, char* p) {
    for (int32_t q = q0; q < q1; ++q) {
        const char* qs = qnum_blob + qnum_off[q];
        const int64_t ql = qnum_off[q + 1] - qnum_off[q];
        const int32_t cnt = n[q] < 0 ? 0 : (n[q] > stride ? stride : n[q]);
        for (int32_t r = 0; r < cnt; ++r) {
            const int64_t i = (int64_t)q * stride + r;
            const int32_t d = doc[i];
            // the URLs of a result list are scattered over the blob (tens of MB): fetch the offsets 8 results and the bytes 4
            // results ahead, or every line costs two cache misses
            if (r + 8 < cnt) { const int32_t d8 = doc[i + 8]; if (d8 >= 0 && d8 < n_docs) __builtin_prefetch(url_off + d8); }
            if (r + 4 < cnt && url_blob) { const int32_t d4 = doc[i + 4]; if (d4 >= 0 && d4 < n_docs) __builtin_prefetch(url_blob + url_off[d4]); }
            memcpy(p, qs, (size_t)ql); p += ql;
            *p++ = '\t';
            p += fmt_u((uint32_t)(r + 1), p);
            *p++ = '\t';
            if (d >= 0 && d < n_docs && url_blob) {
                int64_t ul = url_off[d + 1] - url_off[d];
                if (ul > max_url) ul = max_url;                         // (a wrong max_url_len must not overrun the buffer)
                if (ul > 0) { memcpy(p, url_blob + url_off[d], (size_t)ul); p += ul; }
            }
            *p++ = '\t';
            p += fmt3(score[i], p);
            *p++ = '\n';
        }
    }
    return p;
}

extern "C" int64_t msr_format_lines(const char* qnum_blob, const int64_t* qnum_off, int32_t n_queries, const int32_t* doc,
                                    const double* score, const int32_t* n, int32_t stride, const char* url_blob,
                                    const int64_t* url_off, int64_t n_docs, int64_t max_url_len, char* out, int64_t capacity) {
    if (!qnum_blob || !qnum_off || !doc || !score || !n || !url_off || n_queries < 0 || stride < 0 || capacity < 0 || (capacity && !out))
        return INT64_MIN;
    // pass 1: an upper bound of the bytes per query, without touching the URL table (its entries are scattered: that would be
    // a cache miss per line): rank and score take at most 10 + 24 characters (ranks < 2^32, |score| < 1e12; snprintf's output
    // for anything larger is bounded by 40), a URL at most `max_url` bytes (the caller's max_url_len; a URL longer than
    // that is cut to it rather than written past its line's share of the buffer)
    int64_t max_url = max_url_len;
    if (max_url <= 0) {                                                 // (not given: one sequential pass over the offsets)
        max_url = 0;
        if (url_blob) for (int64_t d = 0; d < n_docs; ++d) max_url = std::max(max_url, url_off[d + 1] - url_off[d]);
    }
    std::vector<int64_t> bound((size_t)n_queries + 1, 0);
    for (int32_t q = 0; q < n_queries; ++q) {
        const int64_t ql = qnum_off[q + 1] - qnum_off[q];
        const int32_t cnt = n[q] < 0 ? 0 : (n[q] > stride ? stride : n[q]);
        bound[q + 1] = bound[q] + (int64_t)cnt * (ql + max_url + 56);
    }
    const int64_t need = bound[n_queries];
    if (need > capacity) return -need;
    // pass 2: contiguous ranges of queries on a few threads, each into the region its bound reserves, then closed up in order
    int n_thr = (int)std::min<int64_t>(4, std::min<int64_t>((int64_t)std::thread::hardware_concurrency() / 2, need / (384 * 1024) + 1));
    if (n_thr < 1) n_thr = 1;
    std::vector<int32_t> cut((size_t)n_thr + 1, 0);
    for (int t = 1; t < n_thr; ++t)                                     // equal shares of the bound
        cut[t] = (int32_t)(std::lower_bound(bound.begin(), bound.end(), need * t / n_thr) - bound.begin());
    cut[n_thr] = n_queries;
    for (int t = 1; t <= n_thr; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    std::vector<char*> end((size_t)n_thr, nullptr);
    auto work = [&](int t) {
        end[t] = format_range(qnum_blob, qnum_off, cut[t], cut[t + 1], doc, score, n, stride, url_blob, url_off, n_docs, max_url,
                              out + bound[cut[t]]);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    char* p = end[0];
    for (int t = 1; t < n_thr; ++t) {
        const char* src = out + bound[cut[t]];
        const int64_t len = end[t] - src;
        if (p != src) memmove(p, src, (size_t)len);
        p += len;
    }
    return (int64_t)(p - out);
}
