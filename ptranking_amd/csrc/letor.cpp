// Native LETOR / libsvm-style text parser (host code, multi-threaded) — the data format on the input side of the hot path.
//
// Replaces the pure-Python tokenizer ptranking/data/data_utils.py:276-387 (iter_lines / parse_letor), which walks every
// token of every line in the interpreter (minutes for MSLR-WEB30K's 3.7 M lines).  Same semantics:
//   <target> qid:<id> <fid>:<val> <fid>:<val> ... [# comment]
//   * feature ids are one-indexed unless told otherwise; absent features take `missing` (the reference's default 0.0);
//   * the feature count is max fid over the WHOLE file (parse_letor pads every row to the widest one, :372-376);
//   * values go text -> double -> float32, exactly the reference's float() followed by the FloatTensor cast
//     (data_utils.py:610), so the parsed matrix is bit-identical;
//   * consecutive lines with the same qid form a query (data_utils.py:420-549 groups the same way).
// Stateless two-call protocol with caller-owned HOST buffers: ptr_letor_scan() sizes, ptr_letor_load() fills.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptranking_amd.h"

namespace ptr {
void set_error(const char *fmt, ...);
}

namespace {

struct FileBuf {
    std::vector<char> data;
    std::vector<size_t> line_start;   // offsets of non-empty lines
};

int read_file(const char *path, FileBuf &fb) {
    FILE *f = fopen(path, "rb");
    if (!f) { ptr::set_error("ptr_letor: cannot open %s", path); return PTR_ERR_INVALID_ARG; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    fb.data.resize((size_t)n + 2);                       // '\n' sentinel + NUL terminator: strtod / strtoll never run off the buffer
    size_t got = n > 0 ? fread(fb.data.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if ((long)got != n) { ptr::set_error("ptr_letor: short read on %s", path); return PTR_ERR_INVALID_ARG; }
    fb.data[(size_t)n] = '\n';
    fb.data[(size_t)n + 1] = '\0';
    size_t pos = 0, end = (size_t)n;
    while (pos < end) {
        const char *nl = (const char *)memchr(fb.data.data() + pos, '\n', end - pos + 1);
        size_t e = nl ? (size_t)(nl - fb.data.data()) : end;
        size_t s = pos;
        while (s < e && (fb.data[s] == ' ' || fb.data[s] == '\t' || fb.data[s] == '\r')) ++s;
        if (s < e && fb.data[s] != '#') fb.line_start.push_back(s);
        pos = e + 1;
    }
    return 0;
}

inline const char *skip_ws(const char *p) {
    while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
    return p;
}

// Decimal text -> double, correctly rounded.  Clinger's fast path: a mantissa of <= 15 significant digits (exact in a double)
// times / divided by a power of ten <= 1e22 (also exact) is ONE correctly rounded IEEE operation, so the result equals
// strtod's / Python float()'s.  Anything else (long mantissas, big exponents, inf / nan / hex spellings) goes to strtod.
inline double parse_double(const char *p, const char **end) {
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                   1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *s = p;
    bool neg = false;
    if (*s == '-') { neg = true; ++s; } else if (*s == '+') { ++s; }
    uint64_t mant = 0;
    int nd = 0, exp10 = 0;
    bool any = false;
    while (*s >= '0' && *s <= '9') { any = true; if (mant || *s != '0') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); ++nd; } else { ++exp10; nd = 100; } } ++s; }
    if (*s == '.') {
        ++s;
        while (*s >= '0' && *s <= '9') { any = true; if (mant || *s != '0') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); ++nd; --exp10; } else { nd = 100; } } else { --exp10; } ++s; }
    }
    if (any && (*s == 'e' || *s == 'E')) {
        const char *t = s + 1;
        bool eneg = false;
        if (*t == '-') { eneg = true; ++t; } else if (*t == '+') { ++t; }
        if (*t >= '0' && *t <= '9') {
            int e = 0;
            while (*t >= '0' && *t <= '9') { if (e < 10000) e = e * 10 + (*t - '0'); ++t; }
            exp10 += eneg ? -e : e;
            s = t;
        }
    }
    if (any && nd <= 15 && exp10 >= -22 && exp10 <= 22) {
        double v = (double)mant;
        v = exp10 < 0 ? v / P10[-exp10] : v * P10[exp10];
        *end = s;
        return neg ? -v : v;
    }
    char *e2 = nullptr;
    const double v = strtod(p, &e2);
    *end = e2;
    return v;
}

struct LineInfo { int64_t qid; int32_t max_fid; bool bad; };

// Parses one line.  X_row == nullptr: only qid / max feature id are extracted.
template <class T>
LineInfo parse_line(const char *p, int one_indexed, T *X_row, int32_t n_features, float *y_out) {
    LineInfo li{0, -1, false};
    char *endp = nullptr;
    const double target = strtod(p, &endp);
    if (endp == p) { li.bad = true; return li; }
    if (y_out) *y_out = (float)target;
    p = skip_ws(endp);
    if (strncmp(p, "qid:", 4) != 0) { li.bad = true; return li; }
    p += 4;
    if (*p == '\n' || *p == '\r' || *p == '\0' || *p == '#' || *p == ' ' || *p == '\t') { li.bad = true; return li; }   // "qid:" with nothing
    // behind it: strtoll would skip the newline and swallow the next line's target
    li.qid = strtoll(p, &endp, 10);
    if (endp == p) {   // non-numeric query id: stable 63-bit FNV-1a hash of the token
        uint64_t h = 1469598103934665603ull;
        while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') { h = (h ^ (unsigned char)*p++) * 1099511628211ull; }
        li.qid = (int64_t)(h >> 1);
        endp = const_cast<char *>(p);
    }
    p = endp;
    for (;;) {
        p = skip_ws(p);
        if (*p == '\n' || *p == '#' || *p == '\0') break;
        long fid = strtol(p, &endp, 10);
        if (endp == p || *endp != ':') { li.bad = true; return li; }
        p = endp + 1;
        if (*p == '\n' || *p == '\r' || *p == '\0' || *p == '#' || *p == ' ' || *p == '\t') { li.bad = true; return li; }   // "fid:" without a
        // value (strtod skips leading whitespace, newlines included)
        const char *vend = p;
        double val = 0.0;
        if (X_row) {
            val = parse_double(p, &vend);
        } else {   // sizing pass: the value only has to be there
            while (*vend != ' ' && *vend != '\t' && *vend != '\n' && *vend != '\r' && *vend != '#' && *vend != '\0') ++vend;
        }
        if (vend == p) { li.bad = true; return li; }
        p = vend;
        if (one_indexed) fid -= 1;
        if (fid < 0) { li.bad = true; return li; }
        if ((int32_t)fid > li.max_fid) li.max_fid = (int32_t)fid;
        if (X_row && fid < n_features) X_row[fid] = (T)val;
    }
    return li;
}

template <class F> void parallel_for(size_t n, F &&fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : std::min(nt, 32u);
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    const size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

int scan(const FileBuf &fb, int one_indexed, std::vector<int64_t> &qid_of_line, int32_t &n_features) {
    const size_t n = fb.line_start.size();
    qid_of_line.resize(n);
    std::vector<int32_t> maxf(n, -1);
    std::vector<char> bad(n, 0);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            LineInfo li = parse_line<float>(fb.data.data() + fb.line_start[i], one_indexed, nullptr, 0, nullptr);
            qid_of_line[i] = li.qid; maxf[i] = li.max_fid; bad[i] = li.bad;
        }
    });
    n_features = 0;
    for (size_t i = 0; i < n; ++i) {
        if (bad[i]) { ptr::set_error("ptr_letor: malformed line %zu", i + 1); return PTR_ERR_INVALID_ARG; }
        n_features = std::max(n_features, maxf[i] + 1);
    }
    return 0;
}

}  // namespace

extern "C" int ptr_letor_scan(const char *path, int one_indexed, int64_t *n_docs, int32_t *n_features, int64_t *n_queries) {
    if (!path || !n_docs || !n_features || !n_queries) { ptr::set_error("ptr_letor_scan: NULL argument"); return PTR_ERR_INVALID_ARG; }
    FileBuf fb;
    if (int rc = read_file(path, fb)) return rc;
    std::vector<int64_t> qids;
    int32_t nf = 0;
    if (int rc = scan(fb, one_indexed, qids, nf)) return rc;
    int64_t nq = 0;
    for (size_t i = 0; i < qids.size(); ++i) nq += (i == 0 || qids[i] != qids[i - 1]) ? 1 : 0;
    *n_docs = (int64_t)qids.size(); *n_features = nf; *n_queries = nq;
    return 0;
}

extern "C" int ptr_letor_load(const char *path, int one_indexed, float missing, int64_t n_docs, int32_t n_features, int64_t n_queries,
                              void *X, int x_is_f64, float *y, int64_t *qids, int64_t *qoff) {
    if (!path || !X || !y || !qids || !qoff) { ptr::set_error("ptr_letor_load: NULL argument"); return PTR_ERR_INVALID_ARG; }
    FileBuf fb;
    if (int rc = read_file(path, fb)) return rc;
    if ((int64_t)fb.line_start.size() != n_docs) { ptr::set_error("ptr_letor_load: file has %zu rows, caller expects %lld", fb.line_start.size(), (long long)n_docs); return PTR_ERR_INVALID_ARG; }
    std::vector<int64_t> line_qid((size_t)n_docs);
    std::vector<char> bad((size_t)n_docs, 0);
    parallel_for((size_t)n_docs, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const char *line = fb.data.data() + fb.line_start[i];
            LineInfo li;
            if (x_is_f64) {
                double *row = (double *)X + i * (size_t)n_features;
                for (int32_t k = 0; k < n_features; ++k) row[k] = (double)missing;
                li = parse_line<double>(line, one_indexed, row, n_features, y + i);
            } else {
                float *row = (float *)X + i * (size_t)n_features;
                for (int32_t k = 0; k < n_features; ++k) row[k] = missing;
                li = parse_line<float>(line, one_indexed, row, n_features, y + i);
            }
            line_qid[i] = li.qid; bad[i] = li.bad || li.max_fid >= n_features;
        }
    });
    int64_t q = 0;
    for (int64_t i = 0; i < n_docs; ++i) {
        if (bad[(size_t)i]) { ptr::set_error("ptr_letor_load: malformed or too wide line %lld", (long long)(i + 1)); return PTR_ERR_INVALID_ARG; }
        if (i == 0 || line_qid[(size_t)i] != line_qid[(size_t)i - 1]) {
            if (q >= n_queries) { ptr::set_error("ptr_letor_load: more queries than the caller sized for"); return PTR_ERR_INVALID_ARG; }
            qids[q] = line_qid[(size_t)i]; qoff[q] = i; ++q;
        }
    }
    if (q != n_queries) { ptr::set_error("ptr_letor_load: %lld queries found, caller expects %lld", (long long)q, (long long)n_queries); return PTR_ERR_INVALID_ARG; }
    qoff[q] = n_docs;
    return 0;
}
