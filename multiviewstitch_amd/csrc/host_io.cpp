// host_io.cpp — the reference's file formats and the Processor::Deform call sequence (include/mvs_io.h).
// Host code only; the numeric work is done by the entries of include/mvs.h.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstdint>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mvs_io.h"
#include "engine.h"
#include "trace.h"

namespace {

struct File {
    FILE* f = nullptr;
    File(const char* path, const char* mode) { if (path) f = std::fopen(path, mode); }
    ~File() { if (f) std::fclose(f); }
    File(const File&) = delete;
    File& operator=(const File&) = delete;
};

int io_fail(const char* what, const char* path) {
    mvs_set_error("%s %s: %s", what, path ? path : "(null)", errno ? std::strerror(errno) : "bad content");
    return MVS_E_IO;
}

// ---- text <-> number, at the speed a 2 M-vertex scan needs (an OBJ of 324 MB took snprintf("%g") 10 s to write and strtof /
// sscanf 6.4 s to read: three orders of magnitude above everything the GPU does with it).  Both fast paths are EXACT: each
// knows when its double arithmetic cannot decide a rounding and hands that value to the C library.
const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// what `os << x` prints for a float (promoted) or a double with the default precision of 6: printf %g
inline void put_g_libc(std::string& out, double v) {
    char buf[40];
    const int n = std::snprintf(buf, sizeof buf, "%g", v);
    out.append(buf, (size_t)n);
}
// -> characters written to buf (at most 24)
inline int fmt_g(char* buf, double v) {
    const double a = std::fabs(v);
    if (!(a >= 1e-17 && a < 1e17)) return std::snprintf(buf, 40, "%g", v);   // 0, -0, inf, nan, the far exponents
    int e = (int)std::floor(std::log10(a));                                  // 10^e <= a < 10^(e+1), fixed up below
    if (e < -17) e = -17;
    if (e > 16) e = 16;
    auto pw = [](int k) { return k >= 0 ? P10[k] : 1.0 / P10[-k]; };         // (comparison only: a last-place error moves e by one step that the loops undo)
    while (e > -17 && a < pw(e)) --e;
    while (e < 16 && a >= pw(e + 1)) ++e;
    // six significant digits: m = round(a / 10^(e-5)); ONE rounding error (P10 is exact), |error| <= 1e6 * 2^-52
    const int k = e - 5;
    const double scaled = k >= 0 ? a / P10[k] : a * P10[-k];
    const double fl = std::floor(scaled), fr = scaled - fl;
    if (std::fabs(fr - 0.5) < 1e-6 || scaled < 99999.0 || scaled >= 1000001.0) return std::snprintf(buf, 40, "%g", v);   // a tie (or nearly): exact decimal arithmetic decides
    uint32_t m = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);
    if (m >= 1000000u) { m = 100000u; ++e; }
    if (m < 100000u) return std::snprintf(buf, 40, "%g", v);
    char d[6];
    for (int i = 5; i >= 0; --i) { d[i] = (char)('0' + m % 10u); m /= 10u; }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') --nd;                                 // %g strips trailing zeros
    int n = 0;
    if (v < 0) buf[n++] = '-';
    if (e < -4 || e >= 6) {                                                  // d.ddddde+XX
        buf[n++] = d[0];
        if (nd > 1) { buf[n++] = '.'; for (int i = 1; i < nd; ++i) buf[n++] = d[i]; }
        buf[n++] = 'e';
        int x = e;
        if (x < 0) { buf[n++] = '-'; x = -x; } else buf[n++] = '+';
        buf[n++] = (char)('0' + x / 10); buf[n++] = (char)('0' + x % 10);    // (|e| <= 17: two digits, as printf pads)
    } else if (e >= 0) {
        for (int i = 0; i <= e; ++i) buf[n++] = i < nd ? d[i] : '0';
        if (nd > e + 1) { buf[n++] = '.'; for (int i = e + 1; i < nd; ++i) buf[n++] = d[i]; }
    } else {
        buf[n++] = '0'; buf[n++] = '.';
        for (int i = 0; i < -e - 1; ++i) buf[n++] = '0';
        for (int i = 0; i < nd; ++i) buf[n++] = d[i];
    }
    return n;
}
inline void put_g(std::string& out, double v) {
    char buf[40];
    out.append(buf, (size_t)fmt_g(buf, v));
}
inline int fmt_int(char* buf, int v) {
    char b[16];
    int n = 0, k = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { b[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    if (v < 0) buf[k++] = '-';
    while (n) buf[k++] = b[--n];
    return k;
}
inline double as_f32(double v) { return (double)(float)v; }

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// `stream >> float` / strtof on [p, end): the next whitespace-delimited number rounded to float32; false when none starts there.
// Fast path (Clinger): up to 15 significant digits and a decimal exponent within +-22 give the correctly rounded DOUBLE with one
// operation; its rounding to float32 is then right unless the double sits exactly on a float32 midpoint — those, the subnormal
// and overflowing results, long mantissas and everything that is not plain decimal go to strtof on a copy of the token.
bool next_f32(const char*& p, const char* end, double* out) {
    const char* q = p;
    while (q < end && is_ws(*q)) ++q;
    if (q >= end) return false;
    const char* tok = q;
    bool neg = false;
    if (*q == '+' || *q == '-') { neg = *q == '-'; ++q; }
    uint64_t mant = 0;
    int nd = 0, dropped = 0, frac = 0;
    bool any = false, plain = q < end && ((*q >= '0' && *q <= '9') || *q == '.');
    while (q < end && *q >= '0' && *q <= '9') { any = true; if (mant || *q != '0') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*q - '0'); ++nd; } else ++dropped; } ++q; }
    if (q < end && *q == '.') {
        ++q;
        while (q < end && *q >= '0' && *q <= '9') { any = true; if (mant || *q != '0') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*q - '0'); ++nd; ++frac; } } else ++frac; ++q; }
    }
    int ex = 0;
    if (any && q < end && (*q == 'e' || *q == 'E')) {
        const char* r = q + 1;
        bool eneg = false;
        if (r < end && (*r == '+' || *r == '-')) { eneg = *r == '-'; ++r; }
        if (r < end && *r >= '0' && *r <= '9') {
            int v = 0;
            while (r < end && *r >= '0' && *r <= '9') { if (v < 100000) v = v * 10 + (*r - '0'); ++r; }
            ex = eneg ? -v : v;
            q = r;
        }
    }
    if (plain && any) {
        const int e10 = ex - frac + dropped;
        if (mant == 0) { *out = neg ? -0.0 : 0.0; p = q; return true; }
        if (nd <= 15 && dropped == 0 && e10 >= -22 && e10 <= 22) {
            const double d = e10 >= 0 ? (double)mant * P10[e10] : (double)mant / P10[-e10];
            uint64_t bits;
            std::memcpy(&bits, &d, 8);
            if (d > 1e-30 && d < 1e30 && (bits & 0x1fffffffu) != 0x10000000u) { const float f = (float)d; *out = neg ? -(double)f : (double)f; p = q; return true; }
        }
    }
    // the C library decides (on a terminated copy: the text goes on behind the token)
    const char* t = tok;
    while (t < end && !is_ws(*t)) ++t;
    char buf[128];
    const size_t len = std::min<size_t>((size_t)(t - tok), sizeof buf - 1);
    std::memcpy(buf, tok, len);
    buf[len] = 0;
    char* e = nullptr;
    const float v = std::strtof(buf, &e);
    if (e == buf) return false;
    p = tok + (e - buf);
    *out = (double)v;
    return true;
}

// next decimal integer on [p, end) (sscanf's %d: leading blanks, a sign); false when none
inline bool next_int(const char*& p, const char* end, int* out) {
    const char* q = p;
    while (q < end && is_ws(*q)) ++q;
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-')) { neg = *q == '-'; ++q; }
    if (q >= end || *q < '0' || *q > '9') return false;
    long long v = 0;
    while (q < end && *q >= '0' && *q <= '9') { if (v < (1ll << 40)) v = v * 10 + (*q - '0'); ++q; }
    *out = (int)(neg ? -v : v);
    p = q;
    return true;
}

// fn(part, begin, end) on `parts` ranges of [0, n) — threads for the large files only
int io_threads(size_t bytes) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw, 16u), bytes >> 20));
}
void parallel_parts(int parts, const std::function<void(int)>& fn) {
    if (parts <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < parts; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto& x : th) x.join();
}

bool read_whole(const char* path, std::string* s) {
    File fp(path, "rb");
    if (!fp.f) return false;
    if (std::fseek(fp.f, 0, SEEK_END) == 0) {                    // a regular file: one allocation, one read
        const long sz = std::ftell(fp.f);
        if (sz > 0 && std::fseek(fp.f, 0, SEEK_SET) == 0) {
            s->resize((size_t)sz);
            const size_t got = std::fread(&(*s)[0], 1, (size_t)sz, fp.f);
            s->resize(got);
            if (got == (size_t)sz) return true;
        } else {
            std::rewind(fp.f);
        }
    }
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp.f)) > 0) s->append(buf, n);
    return true;
}

bool write_whole(const char* path, const std::string& s) {
    File fp(path, "wb");
    if (!fp.f) return false;
    return std::fwrite(s.data(), 1, s.size(), fp.f) == s.size();
}

// Eigen's default operator<< for a dense matrix: every coefficient formatted with the stream precision (6), columns
// right-aligned to the widest coefficient of the whole matrix, " " between columns, "\n" between rows
void put_eigen(std::string& out, const double* M, int rows, int cols) {
    std::vector<std::string> cell((size_t)rows * cols);
    size_t width = 0;
    for (int i = 0; i < rows * cols; ++i) { put_g(cell[i], M[i]); width = std::max(width, cell[i].size()); }
    for (int r = 0; r < rows; ++r) {
        if (r) out += '\n';
        for (int c = 0; c < cols; ++c) {
            if (c) out += ' ';
            const std::string& s = cell[(size_t)r * cols + c];
            out.append(width - s.size(), ' ');
            out += s;
        }
    }
}

const char* const PART_NAMES[16] = {"Head", "Neck", "LeftUpperArm", "LeftLowerArm", "LeftHand", "RightUpperArm", "RightLowerArm",
                                    "RightHand", "LeftThigh", "LeftShank", "LeftFoot", "RightThigh", "RightShank", "RightFoot",
                                    "Truncus", "Hip"};   // enum PART order, PartRecognition.h:13-30

}  // namespace

extern "C" {

// One line of an OBJ file (the reference's loop body, PlyObj.cpp:38-72).  kind: 0 = skipped, 1 = vertex, 2 = normal, 3 = facet,
// -1 = a vertex / normal / facet line that does not parse.  `nn_before` = normals read so far (a facet line is read as
// "f a//b ..." only once a normal has been seen, :61-66).
struct ObjLine { int kind; double v[3]; int f[3]; };
static void obj_line(const char* b, const char* e, int64_t nn_before, bool values, ObjLine* o) {
    if (e > b && e[-1] == '\r') --e;
    o->kind = 0;
    if (e == b || *b == '#') return;
    if (*b == 'v') {
        const bool is_n = e - b > 1 && b[1] == 'n';
        if (!is_n && e - b > 1 && b[1] != ' ' && b[1] != '\t') return;            // vt / vp: the reference would push garbage
        o->kind = is_n ? 2 : 1;
        if (!values) return;
        const char* p = b + (is_n ? 2 : 1);
        if (!next_f32(p, e, o->v) || !next_f32(p, e, o->v + 1) || !next_f32(p, e, o->v + 2)) o->kind = -1;
    } else if (*b == 'f') {
        o->kind = 3;
        if (!values) return;
        const char* p = b + 1;
        const bool slashes = nn_before > 0 && std::memchr(b, '/', (size_t)(e - b)) != nullptr;
        bool ok = p < e && is_ws(*p);                                             // ("f %d": the blank after f)
        for (int c = 0; ok && c < 3; ++c) {
            ok = next_int(p, e, &o->f[c]);
            if (ok && slashes) { int nidx; ok = e - p >= 2 && p[0] == '/' && p[1] == '/' && (p += 2, next_int(p, e, &nidx)); }
        }
        if (!ok) {                                                                // anything unusual: sscanf decides, as before
            const std::string line(b, e);
            int a[3], bb[3];
            ok = slashes ? std::sscanf(line.c_str(), "f %d//%d %d//%d %d//%d", &a[0], &bb[0], &a[1], &bb[1], &a[2], &bb[2]) == 6
                         : std::sscanf(line.c_str(), "f %d %d %d", &a[0], &a[1], &a[2]) == 3;
            for (int c = 0; c < 3; ++c) o->f[c] = a[c];
        }
        if (!ok) o->kind = -1;
    }
}

// the text of an OBJ file -> counts (always) and arrays (those that are not NULL)
static int obj_parse(const std::string& text, const char* path, int64_t* n_vertices, int64_t* n_normals, int64_t* n_faces, double* points,
                     double* normals, int32_t* faces) {
    // line-aligned parts; pass 1 counts the lines of each kind per part (and finds the first line of more than 511 characters:
    // getline(line, 512) fails there and the reference's loop ends, PlyObj.cpp:40-41), pass 2 parses every part at its offsets
    const char* T = text.data();
    const size_t N = text.size();
    const int parts = io_threads(N);
    std::vector<size_t> cut((size_t)parts + 1, N);
    cut[0] = 0;
    for (int t = 1; t < parts; ++t) {
        size_t c = N / parts * t;
        const void* nl = c < N ? std::memchr(T + c, '\n', N - c) : nullptr;
        cut[t] = nl ? (size_t)((const char*)nl - T) + 1 : N;
    }
    for (int t = 1; t <= parts; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    struct Cnt { int64_t nv = 0, nn = 0, nf = 0; size_t stop = (size_t)-1; };
    std::vector<Cnt> cnt((size_t)parts);
    auto walk = [&](int t, size_t stop, const std::function<void(const char*, const char*)>& line) {
        size_t pos = cut[t];
        const size_t lim = std::min(cut[t + 1], stop);
        while (pos < lim) {
            const void* nl = std::memchr(T + pos, '\n', N - pos);
            const size_t eol = nl ? (size_t)((const char*)nl - T) : N;
            if (eol - pos > 511) { cnt[t].stop = std::min(cnt[t].stop, pos); return; }
            line(T + pos, T + eol);
            pos = eol + 1;
        }
    };
    parallel_parts(parts, [&](int t) {                           // (pass 1: first characters only)
        size_t pos = cut[t];
        Cnt c;
        while (pos < cut[t + 1]) {
            const void* nl = std::memchr(T + pos, '\n', N - pos);
            const size_t eol = nl ? (size_t)((const char*)nl - T) : N;
            if (eol - pos > 511) { c.stop = pos; break; }
            const char* b = T + pos;
            size_t len = eol - pos;
            if (len && b[len - 1] == '\r') --len;
            if (len) {
                if (b[0] == 'f') ++c.nf;
                else if (b[0] == 'v') { if (len > 1 && b[1] == 'n') ++c.nn; else if (len == 1 || b[1] == ' ' || b[1] == '\t') ++c.nv; }
            }
            pos = eol + 1;
        }
        cnt[t] = c;
    });
    size_t stop = (size_t)-1;
    for (int t = 0; t < parts; ++t) stop = std::min(stop, cnt[t].stop);
    if (stop != (size_t)-1) {                                    // recount the part that holds the long line, drop the parts behind it
        for (int t = 0; t < parts; ++t) {
            if (cut[t] >= stop) { cnt[t] = Cnt(); continue; }
            if (cut[t + 1] <= stop) continue;
            cnt[t] = Cnt();
            ObjLine o;
            walk(t, stop, [&](const char* b, const char* e) {
                obj_line(b, e, 0, false, &o);
                if (o.kind == 1) ++cnt[t].nv; else if (o.kind == 2) ++cnt[t].nn; else if (o.kind == 3) ++cnt[t].nf;
            });
            cnt[t].stop = (size_t)-1;
        }
    }
    std::vector<int64_t> ov((size_t)parts + 1, 0), on((size_t)parts + 1, 0), of((size_t)parts + 1, 0);
    for (int t = 0; t < parts; ++t) { ov[t + 1] = ov[t] + cnt[t].nv; on[t + 1] = on[t] + cnt[t].nn; of[t + 1] = of[t] + cnt[t].nf; }
    *n_vertices = ov[parts]; *n_normals = on[parts]; *n_faces = of[parts];
    if (!points && !normals && !faces) return MVS_OK;            // (the counting call: a line that does not parse is reported by the filling call)
    std::vector<size_t> bad((size_t)parts, (size_t)-1);
    std::vector<int> bad_kind((size_t)parts, 0);
    parallel_parts(parts, [&](int t) {
        int64_t nv = ov[t], nn = on[t], nf = of[t];
        ObjLine o;
        walk(t, stop, [&](const char* b, const char* e) {
            if (bad[t] != (size_t)-1) return;
            obj_line(b, e, nn, true, &o);
            if (o.kind == 0) return;
            if (o.kind == -1) { bad[t] = (size_t)(b - T); bad_kind[t] = *b == 'f' ? 3 : 1; return; }
            if (o.kind == 1) { if (points) for (int c = 0; c < 3; ++c) points[3 * nv + c] = o.v[c]; ++nv; }
            else if (o.kind == 2) {
                if (normals) {
                    const double len = std::sqrt(o.v[0] * o.v[0] + o.v[1] * o.v[1] + o.v[2] * o.v[2]);   // Vector3d::normalize after the cast (:50-51)
                    for (int c = 0; c < 3; ++c) normals[3 * nn + c] = o.v[c] / len;
                }
                ++nn;
            } else if (o.kind == 3) { if (faces) for (int c = 0; c < 3; ++c) faces[3 * nf + c] = o.f[c] - 1; ++nf; }
        });
    });
    for (int t = 0; t < parts; ++t)
        if (bad[t] != (size_t)-1) { errno = 0; return io_fail(bad_kind[t] == 3 ? "bad face line in" : "bad vertex line in", path); }
    return MVS_OK;
}

int mvs_obj_read(const char* path, int64_t* n_vertices, int64_t* n_normals, int64_t* n_faces, double* points, double* normals,
                 int32_t* faces) {
    MVS_TRACE();
    if (!path || !n_vertices || !n_normals || !n_faces) { mvs_set_error("mvs_obj_read: null argument"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    return obj_parse(text, path, n_vertices, n_normals, n_faces, points, normals, faces);
}

int mvs_obj_write(const char* path, int64_t n_vertices, const double* points, const double* normals, int64_t n_faces,
                  const int32_t* faces) {
    MVS_TRACE();
    if (!path || n_vertices < 0 || n_faces < 0 || (n_vertices && !points) || (n_faces && !faces)) {
        mvs_set_error("mvs_obj_write: bad arguments"); return MVS_E_INVALID_ARG;
    }
    std::string head;
    head += "####\n#\n# OBJ File Generated by MultiviewStitch Program\n#\n####\n# Object ";
    head += path;
    head += "\n#\n# Vertices: " + std::to_string(n_vertices) + "\n# Faces: " + std::to_string(n_faces) + "\n#\n####\n";
    // the vertex block, then the facet block: rounds of `parts` slices of BLOCK items formatted side by side into buffers that are
    // reused, each round written in order (a buffer per part for the WHOLE file was 324 MB of first-touched memory: 2 s)
    const int parts = io_threads((size_t)n_vertices * (normals ? 60 : 30) + (size_t)n_faces * 30);
    const int64_t BLOCK = 1 << 16;
    std::vector<std::string> txt((size_t)parts);
    errno = 0;
    File fp(path, "wb");
    bool ok = fp.f != nullptr;
    auto put = [&](const std::string& x) { if (ok && !x.empty()) ok = std::fwrite(x.data(), 1, x.size(), fp.f) == x.size(); };
    put(head);
    auto rounds = [&](int64_t count, const std::function<void(std::string&, int64_t, int64_t)>& fmt) {
        for (int64_t base = 0; ok && base < count; base += BLOCK * parts) {
            const int live = (int)std::min<int64_t>(parts, (count - base + BLOCK - 1) / BLOCK);
            parallel_parts(live, [&](int t) {
                txt[t].clear();
                fmt(txt[t], base + BLOCK * t, std::min(count, base + BLOCK * (t + 1)));
            });
            for (int t = 0; t < live; ++t) put(txt[t]);
        }
    };
    rounds(n_vertices, [&](std::string& o, int64_t v0, int64_t v1) {
        char ln[320];
        for (int64_t i = v0; i < v1; ++i) {
            int n = 0;
            if (normals) {
                ln[n++] = 'v'; ln[n++] = 'n'; ln[n++] = ' ';
                for (int c = 0; c < 3; ++c) { n += fmt_g(ln + n, as_f32(normals[3 * i + c])); ln[n++] = c < 2 ? ' ' : '\n'; }
            }
            ln[n++] = 'v'; ln[n++] = ' ';
            for (int c = 0; c < 3; ++c) { n += fmt_g(ln + n, as_f32(points[3 * i + c])); ln[n++] = c < 2 ? ' ' : '\n'; }
            o.append(ln, (size_t)n);
        }
    });
    put("# " + std::to_string(n_vertices) + " vertices, " + std::to_string(normals ? n_vertices : 0) + " vertices normals\n\n");
    rounds(n_faces, [&](std::string& q, int64_t f0, int64_t f1) {
        char ln[160];
        for (int64_t i = f0; i < f1; ++i) {
            int n = 0;
            ln[n++] = 'f'; ln[n++] = ' ';
            for (int c = 0; c < 3; ++c) {
                const int id = faces[3 * i + c] + 1;
                const int w = fmt_int(ln + n, id);
                n += w;
                if (normals) { ln[n++] = '/'; ln[n++] = '/'; std::memcpy(ln + n, ln + n - 2 - w, (size_t)w); n += w; }
                ln[n++] = c < 2 ? ' ' : '\n';
            }
            q.append(ln, (size_t)n);
        }
    });
    if (!ok) return io_fail("cannot write", path);
    return MVS_OK;
}

// the whitespace-delimited tokens of [p, end) are counted (pass 1 of the parallel .npts reader); *plain = every one of them is a
// plain decimal number [+-]digits[.digits][e[+-]digits] from its first to its last character
static int64_t count_tokens(const char* p, const char* end, bool* plain) {
    int64_t n = 0;
    bool ok = true;
    while (p < end) {
        while (p < end && is_ws(*p)) ++p;
        if (p >= end) break;
        ++n;
        if (p < end && (*p == '+' || *p == '-')) ++p;
        bool digits = false;
        while (p < end && *p >= '0' && *p <= '9') { digits = true; ++p; }
        if (p < end && *p == '.') { ++p; while (p < end && *p >= '0' && *p <= '9') { digits = true; ++p; } }
        if (digits && p < end && (*p == 'e' || *p == 'E')) {
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            bool ed = false;
            while (p < end && *p >= '0' && *p <= '9') { ed = true; ++p; }
            if (!ed) ok = false;
        }
        if (!digits || (p < end && !is_ws(*p))) ok = false;
        while (p < end && !is_ws(*p)) ++p;
    }
    *plain = ok;
    return n;
}

int mvs_npts_read(const char* path, int64_t* n, double* points, double* normals) {
    MVS_TRACE();
    if (!path || !n) { mvs_set_error("mvs_npts_read: null argument"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    // a stream of numbers, six per record whatever the line structure: parts cut at blanks, pass 1 counts each part's tokens,
    // pass 2 parses them into their places; the first token that is no number ends the file there (as `stream >> float` does)
    const char* T = text.data();
    const size_t N = text.size();
    const int parts = io_threads(N);
    std::vector<size_t> cut((size_t)parts + 1, N);
    cut[0] = 0;
    for (int t = 1; t < parts; ++t) {
        size_t c = N / parts * t;
        while (c < N && !is_ws(T[c])) ++c;
        cut[t] = c;
    }
    for (int t = 1; t <= parts; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    std::vector<int64_t> off((size_t)parts + 1, 0);
    std::vector<char> plain((size_t)parts, 1);
    parallel_parts(parts, [&](int t) { bool ok; off[t + 1] = count_tokens(T + cut[t], T + cut[t + 1], &ok); plain[t] = ok ? 1 : 0; });
    for (int t = 0; t < parts; ++t) off[t + 1] += off[t];
    bool all_plain = true;
    for (int t = 0; t < parts; ++t) all_plain = all_plain && plain[t];
    if (!all_plain) {
        // something that is no plain number: `stream >> float` stops AT it — read as the reference does, one value after the other
        const char* p = T;
        const char* e = T + N;
        int64_t k = 0;
        for (;;) {
            double v[6];
            int got = 0;
            while (got < 6 && next_f32(p, e, v + got)) ++got;
            if (got == 0) break;
            if (got != 6) { errno = 0; return io_fail("truncated record in", path); }
            if (points) for (int c = 0; c < 3; ++c) points[3 * k + c] = v[c];
            if (normals) for (int c = 0; c < 3; ++c) normals[3 * k + c] = v[3 + c];
            ++k;
        }
        *n = k;
        return MVS_OK;
    }
    const int64_t total = off[parts];
    if (total % 6 != 0) { errno = 0; return io_fail("truncated record in", path); }
    *n = total / 6;
    if (!points && !normals) return MVS_OK;
    parallel_parts(parts, [&](int t) {
        const char* p = T + cut[t];
        const char* e = T + cut[t + 1];
        int64_t k = off[t];
        double v;
        while (k < off[t + 1] && next_f32(p, e, &v)) {
            const int64_t rec = k / 6, c = k % 6;
            if (c < 3) { if (points) points[3 * rec + c] = v; } else if (normals) normals[3 * rec + c - 3] = v;
            ++k;
        }
    });
    return MVS_OK;
}

int mvs_npts_write(const char* path, int64_t n, const double* points, const double* normals) {
    MVS_TRACE();
    if (!path || n < 0 || (n && (!points || !normals))) { mvs_set_error("mvs_npts_write: bad arguments"); return MVS_E_INVALID_ARG; }
    const int parts = io_threads((size_t)n * 60);
    std::vector<std::string> txt((size_t)parts);
    parallel_parts(parts, [&](int t) {
        std::string& o = txt[t];
        const int64_t i0 = n * t / parts, i1 = n * (t + 1) / parts;
        o.reserve((size_t)(i1 - i0) * 72);
        char ln[320];
        for (int64_t i = i0; i < i1; ++i) {
            int k = 0;
            for (int c = 0; c < 3; ++c) { k += fmt_g(ln + k, points[3 * i + c]); ln[k++] = ' '; }
            for (int c = 0; c < 3; ++c) { k += fmt_g(ln + k, normals[3 * i + c]); ln[k++] = c < 2 ? ' ' : '\n'; }
            o.append(ln, (size_t)k);
        }
    });
    errno = 0;
    File fp(path, "wb");
    bool ok = fp.f != nullptr;
    for (int t = 0; ok && t < parts; ++t) if (!txt[t].empty()) ok = std::fwrite(txt[t].data(), 1, txt[t].size(), fp.f) == txt[t].size();
    if (!ok) return io_fail("cannot write", path);
    return MVS_OK;
}

int mvs_srt_txt_read(const char* path, int64_t n_seq, double* scales, double* R, double* t) {
    MVS_TRACE();
    if (!path || n_seq < 0 || (n_seq && (!scales || !R || !t))) { mvs_set_error("mvs_srt_txt_read: bad arguments"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    const char* p = text.data();
    const char* end = p + text.size();
    for (int64_t k = 0; k < n_seq; ++k) {
        bool ok = next_f32(p, end, scales + k);
        for (int i = 0; ok && i < 9; ++i) ok = next_f32(p, end, R + 9 * k + i);
        for (int i = 0; ok && i < 3; ++i) ok = next_f32(p, end, t + 3 * k + i);
        if (!ok) { errno = 0; return io_fail("fewer sequences than requested in", path); }
    }
    return MVS_OK;
}

int mvs_srt_txt_write(const char* path, int64_t n_seq, const double* scales, const double* R, const double* t) {
    MVS_TRACE();
    if (!path || n_seq < 0 || (n_seq && (!scales || !R || !t))) { mvs_set_error("mvs_srt_txt_write: bad arguments"); return MVS_E_INVALID_ARG; }
    std::string o;
    for (int64_t k = 0; k < n_seq; ++k) {
        put_g(o, scales[k]); o += '\n';
        put_eigen(o, R + 9 * k, 3, 3); o += '\n';
        put_eigen(o, t + 3 * k, 1, 3); o += '\n';
    }
    errno = 0;
    if (!write_whole(path, o)) return io_fail("cannot write", path);
    return MVS_OK;
}

int mvs_depth_raw_read(const char* path, int32_t w, int32_t h, float* raster) {
    MVS_TRACE();
    if (!path || w <= 0 || h <= 0 || !raster) { mvs_set_error("mvs_depth_raw_read: bad arguments"); return MVS_E_INVALID_ARG; }
    errno = 0;
    File fp(path, "rb");
    if (!fp.f) return io_fail("cannot open", path);
    const size_t n = (size_t)w * h;
    if (std::fread(raster, sizeof(float), n, fp.f) != n) { errno = 0; return io_fail("short raster in", path); }
    return MVS_OK;
}

int mvs_depth_raw_write(const char* path, int64_t n, const double* raster) {
    MVS_TRACE();
    if (!path || n < 0 || (n && !raster)) { mvs_set_error("mvs_depth_raw_write: bad arguments"); return MVS_E_INVALID_ARG; }
    std::vector<float> f((size_t)n);
    for (int64_t i = 0; i < n; ++i) f[i] = (float)raster[i];
    errno = 0;
    File fp(path, "wb");
    if (!fp.f || std::fwrite(f.data(), sizeof(float), f.size(), fp.f) != f.size()) return io_fail("cannot write", path);
    return MVS_OK;
}

int mvs_parts_read(const char* path, int64_t n_vertices, int32_t* labels) {
    MVS_TRACE();
    if (!path || n_vertices < 0 || (n_vertices && !labels)) { mvs_set_error("mvs_parts_read: bad arguments"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    std::fill(labels, labels + n_vertices, 0);
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string name = line.substr(0, eq);
        int part = 0;                                                     // std::map::operator[] of an unknown name yields 0
        for (int k = 0; k < 16; ++k) if (name == PART_NAMES[k]) part = k;
        size_t q = eq + 1;
        while (q < line.size()) {
            size_t sc = line.find(';', q);
            if (sc == std::string::npos) sc = line.size();
            if (sc > q) {
                const long v = std::atol(line.substr(q, sc - q).c_str());
                if (v < 0 || v >= n_vertices) { mvs_set_error("mvs_parts_read: vertex %ld outside [0, %lld)", v, (long long)n_vertices); return MVS_E_INVALID_ARG; }
                labels[v] = part;
            }
            q = sc + 1;
        }
    }
    return MVS_OK;
}

int mvs_processor_deform(const char* model_obj, const char* template_obj, const char* parts_path, const double cam_R[9],
                         double dist_thres, const mvs_deform_params* params, const char* out_obj, mvs_deform_stats* stats) {
    MVS_TRACE();
    if (!model_obj || !template_obj || !parts_path || !cam_R || !out_obj) { mvs_set_error("mvs_processor_deform: null argument"); return MVS_E_INVALID_ARG; }
    int rc;
    int64_t nt, ntn, ntf, ns, nsn, nsf;
    errno = 0;
    std::string model_text, template_text;                       // (each file is read ONCE: counted, then parsed into arrays of that size)
    if (!read_whole(model_obj, &model_text)) return io_fail("cannot open", model_obj);
    if (!read_whole(template_obj, &template_text)) return io_fail("cannot open", template_obj);
    if ((rc = obj_parse(model_text, model_obj, &nt, &ntn, &ntf, nullptr, nullptr, nullptr))) return rc;             // Processor.cpp:1121-1123
    if ((rc = obj_parse(template_text, template_obj, &ns, &nsn, &nsf, nullptr, nullptr, nullptr))) return rc;       // :1125-1127
    if (ntn != nt || nsn != ns) { mvs_set_error("mvs_processor_deform: both meshes need one normal per vertex (model %lld/%lld, template %lld/%lld)",
                                                (long long)ntn, (long long)nt, (long long)nsn, (long long)ns); return MVS_E_INVALID_ARG; }
    std::vector<double> tgt((size_t)nt * 3), tnrm((size_t)nt * 3), src((size_t)ns * 3), snrm((size_t)ns * 3);
    std::vector<int32_t> tf((size_t)ntf * 3), sf((size_t)nsf * 3), s_labels((size_t)ns), t_labels((size_t)nt);
    if ((rc = obj_parse(model_text, model_obj, &nt, &ntn, &ntf, tgt.data(), tnrm.data(), tf.data()))) return rc;
    if ((rc = obj_parse(template_text, template_obj, &ns, &nsn, &nsf, src.data(), snrm.data(), sf.data()))) return rc;
    std::string().swap(model_text); std::string().swap(template_text);
    if ((rc = mvs_parts_read(parts_path, ns, s_labels.data()))) return rc;                                  // Alignment.cpp:38-41
    const double view_ray[3] = {cam_R[6], cam_R[7], cam_R[8]};                                             // R.transpose().col(2), :1133
    double ground[3];
    if ((rc = mvs_align(src.data(), snrm.data(), ns, s_labels.data(), tgt.data(), tnrm.data(), &nt, tf.data(), &ntf, view_ray,
                        dist_thres, t_labels.data(), ground))) return rc;
    mvs_deform_params prm;
    if (params) prm = *params; else mvs_deform_default_params(&prm);
    prm.proj_len_err = 100.0; prm.proj_dist_err = 100.0;                                                   // :1137
    mvs_deform_t h = nullptr;
    if ((rc = mvs_deform_create(ns, src.data(), snrm.data(), nsf, sf.data(), &h))) return rc;             // :1136
    int64_t K = 0;
    std::vector<double> out_n((size_t)ns * 3);
    int warn = MVS_OK;                                   // MVS_W_UNCONVERGED: the result is still written, the caller is told
    if ((rc = mvs_deform_set_target(h, nt, tgt.data(), tnrm.data(), 0)) ||
        (rc = mvs_deform_sample_nodes(h, 16, &K)) ||                                                       // Deformation.cpp:248-250
        ((rc = mvs_deform_iterate(h, &prm, 1, stats)) < 0) ||
        ((warn = rc), (rc = mvs_deform_get_vertices(h, src.data()))) ||
        (rc = mvs_deform_compute_normals(h, out_n.data()))) { mvs_deform_destroy(h); return rc; }          // exportOBJ, Deformation.h:174-221
    mvs_deform_destroy(h);
    rc = mvs_obj_write(out_obj, ns, src.data(), out_n.data(), nsf, sf.data());
    return rc ? rc : warn;
}

}  // extern "C"
