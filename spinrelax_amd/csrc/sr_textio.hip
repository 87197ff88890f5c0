// Host-side text I/O of the reference's xmgrace-style multi-set files (general_scripts.py:182-213, 275-290), native and threaded.
// No device code: the drop-in chain of run-all.bash exchanges C(t) between its scripts as text (512 residues x 2 048 lags x 3
// columns = 31 MB per file), and with the kernels at milliseconds the Python formatting / parsing of those files was
// 90 % of the chain's wall time (bench.py: cli_wall_s).  The bytes written and the doubles read are exactly those of the Python
// implementations in spinrelax_amd/general_scripts.py (tests/test_formats_and_hostlogic.py compares them); anything outside the
// regular case (non-finite values, three-digit exponents, ragged or malformed files) is reported back with a positive
// return code and handled by the Python code path.
#include "sr_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- numpy's array printer for a two-element float64 row (default print options), as general_scripts._numpy_str_pairs ----
// returns false when the row needs numpy itself
bool format_pair(double a, double b, std::string &out)
{
    if (!std::isfinite(a) || !std::isfinite(b)) return false;
    const double v[2] = {a, b};
    const double aa = std::fabs(a), ab = std::fabs(b);
    const bool nza = aa > 0, nzb = ab > 0;
    const double big = std::max(nza ? aa : 0.0, nzb ? ab : 0.0);
    const double small = std::min(nza ? aa : INFINITY, nzb ? ab : INFINITY);
    const bool expf = (nza || nzb) && ((big >= 1e8) || (small < 1e-4) || (big / small > 1000.0));
    if (expf && ((big >= 1e100) || (small < 1e-99))) return false;
    char s[2][48];
    int len[2], dot[2];
    if (!expf) {
        for (int k = 0; k < 2; ++k) {
            len[k] = snprintf(s[k], sizeof s[k], "%.8f", v[k]);
            while (len[k] > 0 && s[k][len[k] - 1] == '0') --len[k];          // '12.00000000' -> '12.'
            s[k][len[k]] = 0;
            dot[k] = (int)(strchr(s[k], '.') - s[k]);
        }
        const int frac[2] = {len[0] - dot[0] - 1, len[1] - dot[1] - 1};
        const int dmax = std::max(dot[0], dot[1]), fmax = std::max(frac[0], frac[1]);
        for (int k = 0; k < 2; ++k) {
            out.append((size_t)(dmax - dot[k]), ' ');
            out.append(s[k], (size_t)len[k]);
            out.append((size_t)(fmax - frac[k]), ' ');
            if (k == 0) out.push_back(' ');
        }
        return true;
    }
    // scientific: the digits both elements need (at most 8 after the point), common to the row
    int prec = 0;
    for (int k = 0; k < 2; ++k) {
        char t[48];
        snprintf(t, sizeof t, "%.8e", v[k]);
        char *e = strchr(t, 'e');
        int n = (int)(e - t);
        while (n > 0 && t[n - 1] == '0') --n;
        const int d = (int)(strchr(t, '.') - t);
        prec = std::max(prec, n - d - 1);
    }
    for (int k = 0; k < 2; ++k) {
        len[k] = snprintf(s[k], sizeof s[k], "%.*e", prec, v[k]);
        if (prec == 0) {                                                     // C prints '1e-05', numpy keeps the point: '1.e-05'
            char *e = strchr(s[k], 'e');
            memmove(e + 1, e, strlen(e) + 1);
            *e = '.';
            ++len[k];
        }
        dot[k] = (int)(strchr(s[k], '.') - s[k]);
    }
    const int dmax = std::max(dot[0], dot[1]);
    for (int k = 0; k < 2; ++k) {
        out.append((size_t)(dmax - dot[k]), ' ');
        out.append(s[k], (size_t)len[k]);
        if (k == 0) out.push_back(' ');
    }
    return true;
}

template <class F>
void parallel_for(int64_t n, int nthreads, F f)
{
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 32), n));
    if (nt == 1) { for (int64_t i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() { for (int64_t i = t; i < n; i += nt) f(i); });
    for (auto &x : th) x.join();
}

// the k-th string of a '\0'-separated list
std::vector<const char *> split0(const char *p, int64_t n)
{
    std::vector<const char *> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) { v[(size_t)i] = p; p += strlen(p) + 1; }
    return v;
}

struct SxySet {
    std::vector<double> x, y, dy;
};
struct SxyFile {
    std::vector<SxySet> sets;
    std::vector<std::string> legends;
};

inline bool plain_number(const char *b, const char *e)
{
    if (b == e) return false;
    for (const char *p = b; p < e; ++p) {
        const char c = *p;
        if (!((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E')) return false;
    }
    return true;
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// data lines of one set: [b, e) holds whole lines; false = something the Python reader should look at
bool parse_rows(const char *b, const char *e, SxySet &s)
{
    while (b < e) {
        const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl : e;
        if (le == b) { b = le + 1; continue; }                 // "\n": skipped by the reference too
        if (is_space(*b)) return false;                         // leading blanks: l[0] tests of the reference differ
        double val[3];
        int nt = 0;
        const char *p = b;
        while (p < le) {
            while (p < le && is_space(*p)) ++p;
            if (p >= le) break;
            const char *q = p;
            while (q < le && !is_space(*q)) ++q;
            if (nt == 3) { ++nt; break; }
            if (!plain_number(p, q)) return false;
            char tmp[64];
            const size_t n = (size_t)(q - p);
            if (n >= sizeof tmp) return false;
            memcpy(tmp, p, n);
            tmp[n] = 0;
            char *endp = nullptr;
            val[nt] = strtod(tmp, &endp);
            if (endp != tmp + n) return false;
            ++nt;
            p = q;
        }
        if (nt < 2 || nt > 3) return false;                     // (more columns: the reference reads the first three, we defer)
        s.x.push_back(val[0]);
        s.y.push_back(val[1]);
        if (nt == 3) s.dy.push_back(val[2]);
        b = le + 1;
    }
    return true;
}

}  // namespace

// print_sxylist (general_scripts.py:275-290) for sets of (C, dC) pairs: per set `legend line`, then `x <pair>` per point, then `&`.
//   header        text written first (complete lines, may be NULL)
//   legend_lines  nsets strings separated by '\0' (each the complete '@s<i> legend "<name>"' line)
//   xstr          npts strings separated by '\0' (str() of the x values, formatted by the caller)
//   ydy           (nsets, npts, 2) float64
// Returns 0, or 1 when some row is outside the regular formats (nothing is written then), negative on I/O errors.
int sr_text_write_sxydy_f64(const char *path, const char *header, int64_t nsets, int64_t npts, const char *legend_lines,
                            const char *xstr, const double *ydy, int nthreads)
{
    if (!path || !legend_lines || !xstr || !ydy || nsets < 0 || npts < 0) { sr_set_error("sr_text_write_sxydy_f64: bad arguments"); return -2; }
    const auto legs = split0(legend_lines, nsets);
    const auto xs = split0(xstr, npts);
    std::vector<std::string> blocks((size_t)nsets);
    std::vector<char> bad((size_t)nsets, 0);
    parallel_for(nsets, nthreads, [&](int64_t i) {
        std::string &o = blocks[(size_t)i];
        o.reserve((size_t)npts * 40 + 64);
        o.append(legs[(size_t)i]);
        o.push_back('\n');
        const double *p = ydy + (size_t)i * (size_t)npts * 2;
        for (int64_t j = 0; j < npts; ++j) {
            o.append(xs[(size_t)j]);
            o.push_back(' ');
            if (!format_pair(p[2 * j], p[2 * j + 1], o)) { bad[(size_t)i] = 1; return; }
            o.push_back('\n');
        }
        o.append("&\n");
    });
    for (char b : bad)
        if (b) return 1;
    FILE *fp = fopen(path, "w");
    if (!fp) { sr_set_error("sr_text_write_sxydy_f64: cannot open %s", path); return -4; }
    bool ok = true;
    if (header && *header) ok = fwrite(header, 1, strlen(header), fp) == strlen(header);
    for (const auto &b : blocks) ok = ok && fwrite(b.data(), 1, b.size(), fp) == b.size();
    ok = (fclose(fp) == 0) && ok;
    if (!ok) { sr_set_error("sr_text_write_sxydy_f64: write to %s failed", path); return -4; }
    return 0;
}

// "%8g %8g" rows (autoCorrelations.export, fitting_Ct_functions.py:107-126): n rows of two values into out (caller: 32 bytes
// per row are enough); returns the number of bytes written.  bounds (ascending row indices, nb of them, may be NULL): offsets[k]
// receives the byte position at which row bounds[k] starts (n: the end) -- the caller cuts the text into blocks there.
int64_t sr_text_format_g8_pairs(const double *a, const double *b, int64_t n, char *out, int64_t out_bytes, int nthreads,
                                const int64_t *bounds, int64_t nb, int64_t *offsets)
{
    if (nb > 0 && (!bounds || !offsets)) { sr_set_error("sr_text_format_g8_pairs: bounds without offsets"); return -2; }
    if (!a || !b || !out || n < 0 || out_bytes < n * 32) { sr_set_error("sr_text_format_g8_pairs: bad arguments"); return -2; }
    // fixed 32-byte cells in parallel, then compacted
    std::vector<int> len((size_t)n);
    const int64_t chunk = 4096, nch = (n + chunk - 1) / chunk;
    parallel_for(nch, nthreads, [&](int64_t c) {
        for (int64_t i = c * chunk; i < std::min(n, (c + 1) * chunk); ++i)
            len[(size_t)i] = snprintf(out + i * 32, 32, "%8g %8g\n", a[i], b[i]);
    });
    int64_t w = 0, ib = 0;
    for (int64_t i = 0; i < n; ++i) {
        while (bounds && ib < nb && bounds[ib] == i) offsets[ib++] = w;
        if (len[(size_t)i] <= 0 || len[(size_t)i] >= 32) { sr_set_error("sr_text_format_g8_pairs: row %lld does not fit", (long long)i); return -3; }
        if (w != i * 32) memmove(out + w, out + i * 32, (size_t)len[(size_t)i]);
        w += len[(size_t)i];
    }
    while (bounds && ib < nb) offsets[ib++] = w;
    return w;
}

// load_sxydylist (general_scripts.py:182-213).  sr_text_open_sxydy parses the file; the getters copy the result out.
// NULL + error text on I/O errors; a handle whose `regular` flag is 0 tells the caller to use the Python reader.
struct sr_sxydy {
    SxyFile f;
    int regular;
};

sr_sxydy *sr_text_open_sxydy(const char *path, const char *key, int nthreads)
{
    if (!path || !key) { sr_set_error("sr_text_open_sxydy: bad arguments"); return nullptr; }
    FILE *fp = fopen(path, "rb");
    if (!fp) { sr_set_error("sr_text_open_sxydy: cannot open %s", path); return nullptr; }
    std::string buf;
    {
        char tmp[1 << 16];
        size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.append(tmp, n);
    }
    fclose(fp);
    auto *h = new sr_sxydy();
    h->regular = 1;
    // pass 1 (serial, memchr-bound): classify lines, collect legends, find the data ranges of the sets
    struct Range { size_t b, e; };
    std::vector<Range> ranges;
    const char *base = buf.data(), *end = base + buf.size();
    const char *p = base;
    size_t cur_b = 0;
    bool in_data = false, any_data_since_close = false;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char c = le > p ? *p : '\n';
        if (c == '#' || c == '@') {
            if (in_data) { ranges.push_back({cur_b, (size_t)(p - base)}); in_data = false; h->regular = 0; }   // header inside a set: defer
            const std::string line(p, (size_t)(le - p));
            if (line.find(key) != std::string::npos) {
                // w[-1].strip('"') of the reference
                size_t e2 = line.size();
                while (e2 > 0 && (is_space(line[e2 - 1]))) --e2;
                size_t b2 = e2;
                while (b2 > 0 && !is_space(line[b2 - 1])) --b2;
                std::string tok = line.substr(b2, e2 - b2);
                while (!tok.empty() && tok.front() == '"') tok.erase(tok.begin());
                while (!tok.empty() && tok.back() == '"') tok.pop_back();
                h->f.legends.push_back(tok);
            }
        } else if (c == '&') {
            ranges.push_back({in_data ? cur_b : (size_t)(p - base), (size_t)(p - base)});
            in_data = false;
            any_data_since_close = false;
        } else if (le > p) {
            if (!in_data) { cur_b = (size_t)(p - base); in_data = true; }
            any_data_since_close = true;
        }
        p = le + 1;
    }
    if (in_data && any_data_since_close) { ranges.push_back({cur_b, buf.size()}); h->regular = 0; }   // no closing '&': the reference's dy quirk, defer
    h->f.sets.resize(ranges.size());
    std::vector<char> bad(ranges.size(), 0);
    parallel_for((int64_t)ranges.size(), nthreads, [&](int64_t i) {
        if (!parse_rows(base + ranges[(size_t)i].b, base + ranges[(size_t)i].e, h->f.sets[(size_t)i])) bad[(size_t)i] = 1;
    });
    for (char b : bad)
        if (b) h->regular = 0;
    // the Python reader returns rectangular arrays only when every set has the same length and dy is all-or-nothing per set
    for (const auto &s : h->f.sets) {
        if (s.x.size() != h->f.sets[0].x.size()) h->regular = 0;
        if (!(s.dy.empty() || s.dy.size() == s.x.size())) h->regular = 0;
        if (s.dy.empty() != h->f.sets[0].dy.empty()) h->regular = 0;
        if (s.x.empty()) h->regular = 0;
    }
    return h;
}

void sr_text_close_sxydy(sr_sxydy *h) { delete h; }

// regular, nsets, npts, has_dy, number of legends, bytes of the '\0'-joined legends
int sr_text_sxydy_info(const sr_sxydy *h, int64_t *info6)
{
    if (!h || !info6) { sr_set_error("sr_text_sxydy_info: bad arguments"); return -2; }
    info6[0] = h->regular;
    info6[1] = (int64_t)h->f.sets.size();
    info6[2] = h->f.sets.empty() ? 0 : (int64_t)h->f.sets[0].x.size();
    info6[3] = h->f.sets.empty() ? 0 : (h->f.sets[0].dy.empty() ? 0 : 1);
    info6[4] = (int64_t)h->f.legends.size();
    int64_t nb = 0;
    for (const auto &l : h->f.legends) nb += (int64_t)l.size() + 1;
    info6[5] = nb;
    return 0;
}

int sr_text_sxydy_get(const sr_sxydy *h, double *x, double *y, double *dy, char *legends)
{
    if (!h || !h->regular) { sr_set_error("sr_text_sxydy_get: not a regular file"); return -3; }
    const size_t n = h->f.sets.empty() ? 0 : h->f.sets[0].x.size();
    for (size_t i = 0; i < h->f.sets.size(); ++i) {
        const auto &s = h->f.sets[i];
        if (x) memcpy(x + i * n, s.x.data(), n * sizeof(double));
        if (y) memcpy(y + i * n, s.y.data(), n * sizeof(double));
        if (dy && !s.dy.empty()) memcpy(dy + i * n, s.dy.data(), n * sizeof(double));
    }
    if (legends) {
        char *q = legends;
        for (const auto &l : h->f.legends) { memcpy(q, l.c_str(), l.size() + 1); q += l.size() + 1; }
    }
    return 0;
}
