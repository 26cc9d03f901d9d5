// orc_match.cpp — TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  CPU restatement of the match-filter cascade of
// Processor::AlignmentSeq (R/Processor/Processor.cpp:644-735) and of SSD() (R/Common/Utils.h:221-241).
// PARITY UNPINNED: the reference has no fixture for it and the grey conversion lives in OpenCV (not vendored); the
// 8-bit formula below is a recollection of cv::cvtColor(COLOR_RGB2GRAY).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <set>
#include <vector>

namespace {
inline int grey8(const uint8_t* px) { return (4899 * px[0] + 9617 * px[1] + 1868 * px[2] + 8192) >> 14; }
}

extern "C" int orc_match_filter(const int32_t* raw, int64_t n, const int32_t* tex1, const uint8_t* valid1, const int32_t* tex2,
                                const uint8_t* valid2, const uint8_t* img1, const uint8_t* img2, int w, int h, int view_count, int ssd_win,
                                double ssd_err, int sample_interval, int32_t* out, int64_t* n_out, int64_t* stage_counts) {
    const int64_t npx = (int64_t)w * h;
    std::set<std::array<int32_t, 4>> uniq;                                                   // :650-667
    for (int64_t k = 0; k < n; ++k) {
        const int32_t* r = raw + 6 * k;
        const int u1 = r[1], v1 = r[2], u2 = r[4], v2 = r[5];
        if (!(u1 >= 0 && u1 < w && v1 >= 0 && v1 < h && u2 >= 0 && u2 < w && v2 >= 0 && v2 < h)) continue;
        const int idx1 = tex1[r[0] * npx + (int64_t)v1 * w + u1], idx2 = tex2[r[3] * npx + (int64_t)v2 * w + u2];
        if (idx1 != -1 && idx2 != -1 && valid1[(int64_t)v1 * w + u1] && valid2[(int64_t)v2 * w + u2])
            uniq.insert({idx1 % w, idx1 / w, idx2 % w, idx2 / w});
    }
    std::vector<std::array<int32_t, 4>> m(uniq.begin(), uniq.end());                         // :671-680
    const int64_t n1 = (int64_t)m.size();
    size_t keep = 0;                                                                          // :683-707
    for (size_t k = 0; k < m.size(); ++k) {
        const int u1 = m[k][0], v1 = m[k][1], u2 = m[k][2], v2 = m[k][3];
        if (u1 >= ssd_win && v1 >= ssd_win && u2 >= ssd_win && v2 >= ssd_win && u1 < w - ssd_win && v1 < h - ssd_win && u2 < w - ssd_win &&
            v2 < h - ssd_win) {
            double sum = 0.0;
            const int len = 2 * ssd_win + 1;
            for (int i = 0; i < len; ++i)
                for (int j = 0; j < len; ++j) {
                    const int g1 = grey8(img1 + 3 * ((int64_t)(v1 - ssd_win + i) * w + (u1 - ssd_win + j)));
                    const int g2 = grey8(img2 + 3 * ((int64_t)(v2 - ssd_win + i) * w + (u2 - ssd_win + j)));
                    sum += double(g1 - g2) * double(g1 - g2);
                }
            if (std::sqrt(sum / (len * len)) <= ssd_err) m[keep++] = m[k];
        }
    }
    m.resize(keep);
    const int64_t n2 = (int64_t)keep;
    const double gap = (double)sample_interval * (double)sample_interval;                    // :711-731
    size_t ns = 0;
    for (size_t k = 0; k < m.size(); ++k) {
        bool flag = false;
        for (size_t k0 = 0; k0 < ns; ++k0) {
            const int x0 = m[k0][0] - m[k][0], x1 = m[k0][1] - m[k][1], y0 = m[k0][2] - m[k][2], y1 = m[k0][3] - m[k][3];
            if ((x0 * x0 + x1 * x1) <= gap || (y0 * y0 + y1 * y1) <= gap) { flag = true; break; }
        }
        if (!flag) m[ns++] = m[k];
    }
    for (size_t k = 0; k < ns; ++k) std::copy(m[k].begin(), m[k].end(), out + 4 * k);
    *n_out = (int64_t)ns;
    if (stage_counts) { stage_counts[0] = n1; stage_counts[1] = n2; stage_counts[2] = (int64_t)ns; }
    return 0;
}
