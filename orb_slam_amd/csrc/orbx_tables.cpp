// orbx_tables.cpp -- host-side tables: constructor tables of the extractor,
// per-level geometry, and the fixed-point coefficient tables of OpenCV's
// 8-bit bilinear resize (generic path of OpenCV 3.4.1, which the reference
// pins at CMakeLists.txt:107-109).  Compiled with -ffp-contract=off.
#include "orbx_tables.hpp"

#include <cmath>
#include <cstring>
#include <algorithm>

#include "../../include/orbx.h"

namespace orbx {

// OpenCV cvRound: round half to even
static inline int round_half_even(float v) { return (int)lrintf(v); }
static inline int round_half_even(double v) { return (int)lrint(v); }
static inline int floor_int(float v) { int i = (int)v; return i - (i > v); }
static inline int ceil_int(float v) { int i = (int)v; return i + (i < v); }
static inline int16_t sat16(int v) { return (int16_t)std::min(32767, std::max(-32768, v)); }

bool build_params(Params& p, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh)
{
    if (nlevels < 1 || nlevels > kMaxLevels || nfeatures < 0 || !(scaleFactor > 1.0f)) return false;
    p.nfeatures = nfeatures;
    p.nlevels = nlevels;
    p.iniTh = iniTh;
    p.minTh = minTh;
    p.scaleFactor = scaleFactor;

    // src/ORBextractor.cc:971-987
    p.scale[0] = 1.0f;
    p.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        p.scale[i] = (float)(p.scale[i - 1] * p.scaleFactor);
        p.sigma2[i] = p.scale[i] * p.scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        p.inv_scale[i] = 1.0f / p.scale[i];
        p.inv_sigma2[i] = 1.0f / p.sigma2[i];
    }
    // :991-1002 geometric split of the feature budget
    float factor = (float)(1.0f / p.scaleFactor);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        p.nfeat[l] = round_half_even(desired);
        sum += p.nfeat[l];
        desired *= factor;
    }
    p.nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);

    // :1010-1025 row extents of the circular patch
    int vmax = floor_int(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = ceil_int(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) p.umax[v] = round_half_even(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (p.umax[v0] == p.umax[v0 + 1]) ++v0;
        p.umax[v] = v0;
        ++v0;
    }
    int n = 0;
    for (int dv = -kHalfPatch; dv <= kHalfPatch; dv++) {
        int d = p.umax[dv < 0 ? -dv : dv];
        for (int du = -d; du <= d; du++) {
            if (n >= kPatchPixels) return false;
            p.patch_du[n] = (int8_t)du;
            p.patch_dv[n] = (int8_t)dv;
            n++;
        }
    }
    return n == kPatchPixels;
}

void level_size(const Params& p, int w, int h, int level, int* lw, int* lh)
{
    float s = p.inv_scale[level];  // src/ORBextractor.cc:1196-1197
    *lw = round_half_even((float)w * s);
    *lh = round_half_even((float)h * s);
}

// cv::resize INTER_LINEAR coefficient tables for one axis: destination index d
// reads source s and s+1 with weights (a0, a1) in 1/2048 units.
static void axis_table(int src, int dst, std::vector<int>& ofs, std::vector<int16_t>& c0, std::vector<int16_t>& c1,
                       bool clamp_in_table)
{
    double inv_scale = (double)dst / src;
    double scale = 1.0 / inv_scale;
    ofs.resize(dst); c0.resize(dst); c1.resize(dst);
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = floor_int(f);
        f -= s;
        if (clamp_in_table) {           // horizontal: the table itself is clamped
            if (s < 0) { f = 0; s = 0; }
            if (s >= src - 1) { f = 0; s = src - 1; }
        }
        ofs[d] = s;
        c0[d] = sat16(round_half_even((1.f - f) * 2048));
        c1[d] = sat16(round_half_even(f * 2048));
    }
}

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

int build_geometry(const Params& p, int w, int h, FrameGeom& g, ResizeTables& tabs, const char** msg)
{
    std::vector<XTab>& xtab = tabs.xtab;
    std::vector<YTab>& ytab = tabs.ytab;
    std::vector<XTab4>& xtab4 = tabs.xtab4;
    std::vector<YTab2>& ytab2 = tabs.ytab2;
    xtab4.clear();
    ytab2.clear();
    bool slide_ok = true;
    static const char* ok = "";
    *msg = ok;
    std::memset(&g, 0, sizeof(g));
    if (w < 1 || h < 1) { *msg = "empty image"; return ORBX_ERR_INVALID; }
    if (w > 16384 || h > 16384) { *msg = "image larger than 16384 pixels on a side"; return ORBX_ERR_UNSUPPORTED; }
    g.nlevels = p.nlevels;
    g.iniTh = std::min(std::max(p.iniTh, 0), 255);   // cv::FAST clamps its threshold
    g.minTh = std::min(std::max(p.minTh, 0), 255);
    xtab.clear();
    ytab.clear();
    uint64_t pyr = 0, cand = 0;
    int cells = 0, kps = 0, max_nodes = 0;
    for (int l = 0; l < p.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        level_size(p, w, h, l, &L.w, &L.h);
        if (L.w < 1 || L.h < 1) { *msg = "a pyramid level is empty"; return ORBX_ERR_UNSUPPORTED; }
        L.pitch = (L.w + 2 * kPadX + 63) & ~63;
        L.rows = L.h + 2 * kPadY;
        L.plane_off = (uint32_t)pyr;
        pyr += (uint64_t)L.pitch * L.rows;
        pyr = (pyr + 255) & ~(uint64_t)255;
        L.scale = p.scale[l];
        L.patch_size = (float)(int)(kPatch * p.scale[l]);   // src/ORBextractor.cc:1109

        // cell grid, src/ORBextractor.cc:1036-1050
        int width = L.w - 2 * kDetBorder, height = L.h - 2 * kDetBorder;
        L.nfeat = p.nfeat[l];
        L.cell_base = cells;
        L.cand_base = (uint32_t)cand;
        L.key_base = (uint32_t)cand;
        L.kp_base = kps;
        if (width >= 30 && height >= 30) {
            const float W = 30;
            L.width = width;
            L.height = height;
            L.nCols = (int)((float)width / W);
            L.nRows = (int)((float)height / W);
            L.rcp_ncols = L.nCols > 1 ? (uint32_t)(((1ull << 32) + L.nCols - 1) / L.nCols) : 0u;
            L.wCell = (int)std::ceil((float)width / L.nCols);
            L.hCell = (int)std::ceil((float)height / L.nRows);
            if (L.wCell > 64 || L.hCell > 64) { *msg = "FAST cell larger than 64 pixels"; return ORBX_ERR_UNSUPPORTED; }
            // strict 8-neighbour maxima cannot touch: at most one per 2x2 block
            L.cell_cap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
            int ncell = L.nCols * L.nRows;
            L.key_cap = (uint32_t)ncell * L.cell_cap;
            cells += ncell;
            cand += L.key_cap;
            // quadtree roots, src/ORBextractor.cc:492-494
            L.nIni = (int)std::round((float)width / (float)height);
            if (L.nIni < 1) { *msg = "image taller than twice its width (the reference divides by zero)"; return ORBX_ERR_UNSUPPORTED; }
            L.hX = (float)width / L.nIni;
            L.kp_cap = std::max(L.nfeat + 3, 4 * L.nIni);
            if (L.kp_cap + 4 > kMaxNodesBig) { *msg = "more than 16376 features on one level"; return ORBX_ERR_UNSUPPORTED; }
            max_nodes = std::max(max_nodes, L.kp_cap + 4);
            kps += (L.kp_cap + 7) & ~7;    // whole groups of 8 slots per level: a descriptor workgroup (8 keypoints) never straddles levels
        }
        // A level whose search region is smaller than one 30-pixel cell yields no
        // keypoints (the reference would divide by zero there).

        if (l > 0) {
            const LevelGeom& P = g.lv[l - 1];
            std::vector<int> ofs;
            std::vector<int16_t> c0, c1;
            axis_table(P.w, L.w, ofs, c0, c1, true);
            L.xtab_off = (uint32_t)xtab.size();
            for (int d = 0; d < L.w; d++) xtab.push_back(XTab{(uint16_t)ofs[d], 0, c0[d], c1[d]});
            // the same per plane dword column, border columns resolved through REFLECT_101
            L.xtab4_off = (uint32_t)xtab4.size();
            for (int dwc = 0; dwc < L.pitch / 4; dwc++) {
                int sx[4], base = 1 << 30;
                for (int k = 0; k < 4; k++) {
                    sx[k] = ofs[reflect101(4 * dwc + k - kPadX, L.w)];
                    base = std::min(base, sx[k]);
                }
                XTab4 e;
                e.base = (uint32_t)base;
                for (int k = 0; k < 4; k++) {
                    int dx = reflect101(4 * dwc + k - kPadX, L.w);
                    int ia = sx[k] - base, ib = ia + 1;
                    if (ib > 7) slide_ok = false;
                    e.sel[k] = (uint32_t)(ia & 7) | (0x0cu << 8) | ((uint32_t)(ib & 7) << 16) | (0x0cu << 24);
                    e.a01[k] = (uint32_t)(uint16_t)c0[dx] | ((uint32_t)(uint16_t)c1[dx] << 16);
                }
                xtab4.push_back(e);
            }
            axis_table(P.h, L.h, ofs, c0, c1, false);
            L.ytab_off = (uint32_t)ytab.size();
            for (int d = 0; d < L.h; d++) {
                // vertical: rows s and s+1, each clipped into the source
                int y0 = std::min(std::max(ofs[d], 0), P.h - 1);
                int y1 = std::min(std::max(ofs[d] + 1, 0), P.h - 1);
                ytab.push_back(YTab{(uint16_t)y0, (uint16_t)y1, c0[d], c1[d]});
            }
            L.ytab2_off = (uint32_t)ytab2.size();
            for (int r = 0; r < L.rows; r++) {
                int d = reflect101(r - kPadY, L.h);
                int y0 = std::min(std::max(ofs[d], 0), P.h - 1);
                int y1 = std::min(std::max(ofs[d] + 1, 0), P.h - 1);
                if (c0[d] < 0 || c1[d] < 0) slide_ok = false;
                ytab2.push_back(YTab2{(uint32_t)((y0 + kPadY) * P.pitch + kPadX), (uint32_t)((y1 + kPadY) * P.pitch + kPadX),
                                      c0[d], c1[d]});
            }
        }
    }
    if (pyr > 0xFFFFFFFFull || cand > 0x7FFFFFFFull) { *msg = "frame too large"; return ORBX_ERR_UNSUPPORTED; }
    g.ncells = cells;
    // per-cell constants of the FAST kernel (tested areas tile [3, width-3) x [3, height-3))
    tabs.fast_cells.assign((size_t)cells, FastCell{});
    for (int l = 0; l < p.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        if (L.width <= 0) continue;
        for (int cl = 0; cl < L.nCols * L.nRows; cl++) {
            FastCell& fc = tabs.fast_cells[(size_t)L.cell_base + cl];
            const int ci = cl / L.nCols, cj = cl - ci * L.nCols;
            const int tx0 = cj * L.wCell + 3, tx1 = std::min((cj + 1) * L.wCell + 3, L.width - 3);
            const int ty0 = ci * L.hCell + 3, ty1 = std::min((ci + 1) * L.hCell + 3, L.height - 3);
            const int cw = std::max(tx1 - tx0, 0), ch = std::max(ty1 - ty0, 0);
            const int pcol0 = kPadX + kDetBorder + tx0 - 3, prow0 = kPadY + kDetBorder + ty0 - 3;
            const int acol0 = pcol0 & ~3, ox = pcol0 - acol0;
            const int ndw = (ox + cw + 6 + 3) >> 2;
            fc.cand_off = L.cand_base + (uint32_t)cl * (uint32_t)L.cell_cap;
            fc.cell_cap = (uint16_t)L.cell_cap;
            fc.pitch = (uint16_t)L.pitch;
            if (cw == 0 || ch == 0) continue;      // cw = ch = 0: the kernel only writes a zero count
            fc.src_off = L.plane_off + (uint32_t)prow0 * (uint32_t)L.pitch + (uint32_t)acol0;
            fc.rcp_cw = ((1u << 18) + cw - 1) / cw;
            fc.cw = (uint8_t)cw;
            fc.ch = (uint8_t)ch;
            fc.ox = (uint8_t)ox;
            fc.ndw = (uint8_t)ndw;
            // pretest lane layout: four pixels per lane, whole rows per 64-lane round
            const int lpr = (cw + 3) / 4;
            fc.lpr = (uint8_t)lpr;
            fc.fast_rows = (uint8_t)(64 / lpr);
            fc.rcp_lpr = (uint32_t)((65536 + lpr - 1) / lpr);
        }
    }
    g.pyr_frame_bytes = (uint32_t)pyr;
    g.cand_per_frame = (uint32_t)cand;
    g.keys_per_frame = (uint32_t)cand;
    g.kp_per_frame = kps;
    g.max_nodes = std::max(max_nodes, 64);
    g.resize_slide_ok = slide_ok ? 1 : 0;
    if (xtab4.empty()) xtab4.push_back(XTab4{0, {0, 0, 0, 0}, {0, 0, 0, 0}});
    if (ytab2.empty()) ytab2.push_back(YTab2{0, 0, 0, 0});
    if (xtab.empty()) xtab.push_back(XTab{0, 0, 0, 0});
    if (ytab.empty()) ytab.push_back(YTab{0, 0, 0, 0});
    return ORBX_OK;
}

void gauss7_taps(int taps[7])
{
    // getGaussianKernel(7, sigma=2) scaled by 256 and rounded (not renormalised)
    double g[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; g[i] = std::exp(-0.5 * x * x / 4.0); sum += g[i]; }
    for (int i = 0; i < 7; i++) taps[i] = (int)lrint(256.0 * g[i] / sum);
}

} // namespace orbx
