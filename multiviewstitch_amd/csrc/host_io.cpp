// host_io.cpp — the reference's file formats and the Processor::Deform call sequence (include/mvs_io.h).
// Host code only; the numeric work is done by the entries of include/mvs.h.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// what `os << x` prints for a float (promoted) or a double with the default precision of 6: printf %g
inline void put_g(std::string& out, double v) {
    char buf[40];
    const int n = std::snprintf(buf, sizeof buf, "%g", v);
    out.append(buf, (size_t)n);
}
inline double as_f32(double v) { return (double)(float)v; }

// `stream >> float`: next whitespace-delimited number, rounded to float32
bool next_f32(const char*& p, double* out) {
    char* e = nullptr;
    const float v = std::strtof(p, &e);
    if (e == p) return false;
    p = e;
    *out = (double)v;
    return true;
}

bool read_whole(const char* path, std::string* s) {
    File fp(path, "rb");
    if (!fp.f) return false;
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

int mvs_obj_read(const char* path, int64_t* n_vertices, int64_t* n_normals, int64_t* n_faces, double* points, double* normals,
                 int32_t* faces) {
    MVS_TRACE();
    if (!path || !n_vertices || !n_normals || !n_faces) { mvs_set_error("mvs_obj_read: null argument"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    int64_t nv = 0, nn = 0, nf = 0;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        if (eol - pos > 511) break;                    // getline(line, 512) fails on a longer line and the loop ends (PlyObj.cpp:40-41)
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        if (line[0] == 'v') {
            const bool is_n = line.size() > 1 && line[1] == 'n';
            if (!is_n && line.size() > 1 && line[1] != ' ' && line[1] != '\t') continue;   // vt / vp: the reference would push garbage
            const char* p = line.c_str() + (is_n ? 2 : 1);
            double v[3];
            if (!next_f32(p, v) || !next_f32(p, v + 1) || !next_f32(p, v + 2)) { errno = 0; return io_fail("bad vertex line in", path); }
            if (is_n) {
                if (normals) {
                    const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);   // Vector3d::normalize after the cast (:50-51)
                    for (int c = 0; c < 3; ++c) normals[3 * nn + c] = v[c] / len;
                }
                ++nn;
            } else {
                if (points) for (int c = 0; c < 3; ++c) points[3 * nv + c] = v[c];
                ++nv;
            }
        } else if (line[0] == 'f') {
            int a[3], b[3];
            bool ok;
            if (nn > 0 && line.find('/') != std::string::npos)                                 // :61-66
                ok = std::sscanf(line.c_str(), "f %d//%d %d//%d %d//%d", &a[0], &b[0], &a[1], &b[1], &a[2], &b[2]) == 6;
            else
                ok = std::sscanf(line.c_str(), "f %d %d %d", &a[0], &a[1], &a[2]) == 3;
            if (!ok) { errno = 0; return io_fail("bad face line in", path); }
            if (faces) for (int c = 0; c < 3; ++c) faces[3 * nf + c] = a[c] - 1;
            ++nf;
        }
    }
    *n_vertices = nv; *n_normals = nn; *n_faces = nf;
    return MVS_OK;
}

int mvs_obj_write(const char* path, int64_t n_vertices, const double* points, const double* normals, int64_t n_faces,
                  const int32_t* faces) {
    MVS_TRACE();
    if (!path || n_vertices < 0 || n_faces < 0 || (n_vertices && !points) || (n_faces && !faces)) {
        mvs_set_error("mvs_obj_write: bad arguments"); return MVS_E_INVALID_ARG;
    }
    std::string o;
    o.reserve((size_t)n_vertices * (normals ? 80 : 40) + (size_t)n_faces * 40 + 256);
    o += "####\n#\n# OBJ File Generated by MultiviewStitch Program\n#\n####\n# Object ";
    o += path;
    o += "\n#\n# Vertices: " + std::to_string(n_vertices) + "\n# Faces: " + std::to_string(n_faces) + "\n#\n####\n";
    for (int64_t i = 0; i < n_vertices; ++i) {
        if (normals) {
            o += "vn ";
            put_g(o, as_f32(normals[3 * i])); o += ' '; put_g(o, as_f32(normals[3 * i + 1])); o += ' '; put_g(o, as_f32(normals[3 * i + 2]));
            o += '\n';
        }
        o += "v ";
        put_g(o, as_f32(points[3 * i])); o += ' '; put_g(o, as_f32(points[3 * i + 1])); o += ' '; put_g(o, as_f32(points[3 * i + 2]));
        o += '\n';
    }
    o += "# " + std::to_string(n_vertices) + " vertices, " + std::to_string(normals ? n_vertices : 0) + " vertices normals\n\n";
    for (int64_t i = 0; i < n_faces; ++i) {
        o += "f ";
        for (int c = 0; c < 3; ++c) {
            const std::string id = std::to_string(faces[3 * i + c] + 1);
            o += id;
            if (normals) { o += "//"; o += id; }
            o += c < 2 ? ' ' : '\n';
        }
    }
    errno = 0;
    if (!write_whole(path, o)) return io_fail("cannot write", path);
    return MVS_OK;
}

int mvs_npts_read(const char* path, int64_t* n, double* points, double* normals) {
    MVS_TRACE();
    if (!path || !n) { mvs_set_error("mvs_npts_read: null argument"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    const char* p = text.c_str();
    int64_t k = 0;
    for (;;) {
        double v[6];
        int got = 0;
        while (got < 6 && next_f32(p, v + got)) ++got;
        if (got == 0) break;
        if (got != 6) { errno = 0; return io_fail("truncated record in", path); }
        if (points) for (int c = 0; c < 3; ++c) points[3 * k + c] = v[c];
        if (normals) for (int c = 0; c < 3; ++c) normals[3 * k + c] = v[3 + c];
        ++k;
    }
    *n = k;
    return MVS_OK;
}

int mvs_npts_write(const char* path, int64_t n, const double* points, const double* normals) {
    MVS_TRACE();
    if (!path || n < 0 || (n && (!points || !normals))) { mvs_set_error("mvs_npts_write: bad arguments"); return MVS_E_INVALID_ARG; }
    std::string o;
    o.reserve((size_t)n * 72);
    for (int64_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) { put_g(o, points[3 * i + c]); o += ' '; }
        for (int c = 0; c < 3; ++c) { put_g(o, normals[3 * i + c]); o += c < 2 ? ' ' : '\n'; }
    }
    errno = 0;
    if (!write_whole(path, o)) return io_fail("cannot write", path);
    return MVS_OK;
}

int mvs_srt_txt_read(const char* path, int64_t n_seq, double* scales, double* R, double* t) {
    MVS_TRACE();
    if (!path || n_seq < 0 || (n_seq && (!scales || !R || !t))) { mvs_set_error("mvs_srt_txt_read: bad arguments"); return MVS_E_INVALID_ARG; }
    errno = 0;
    std::string text;
    if (!read_whole(path, &text)) return io_fail("cannot open", path);
    const char* p = text.c_str();
    for (int64_t k = 0; k < n_seq; ++k) {
        bool ok = next_f32(p, scales + k);
        for (int i = 0; ok && i < 9; ++i) ok = next_f32(p, R + 9 * k + i);
        for (int i = 0; ok && i < 3; ++i) ok = next_f32(p, t + 3 * k + i);
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
    if ((rc = mvs_obj_read(model_obj, &nt, &ntn, &ntf, nullptr, nullptr, nullptr))) return rc;             // Processor.cpp:1121-1123
    if ((rc = mvs_obj_read(template_obj, &ns, &nsn, &nsf, nullptr, nullptr, nullptr))) return rc;          // :1125-1127
    if (ntn != nt || nsn != ns) { mvs_set_error("mvs_processor_deform: both meshes need one normal per vertex (model %lld/%lld, template %lld/%lld)",
                                                (long long)ntn, (long long)nt, (long long)nsn, (long long)ns); return MVS_E_INVALID_ARG; }
    std::vector<double> tgt((size_t)nt * 3), tnrm((size_t)nt * 3), src((size_t)ns * 3), snrm((size_t)ns * 3);
    std::vector<int32_t> tf((size_t)ntf * 3), sf((size_t)nsf * 3), s_labels((size_t)ns), t_labels((size_t)nt);
    if ((rc = mvs_obj_read(model_obj, &nt, &ntn, &ntf, tgt.data(), tnrm.data(), tf.data()))) return rc;
    if ((rc = mvs_obj_read(template_obj, &ns, &nsn, &nsf, src.data(), snrm.data(), sf.data()))) return rc;
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
