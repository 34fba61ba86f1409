/*
 * orb_oracle.c -- CPU restatement of the GF-ORB-SLAM2 ORB front-end (see orb_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the reference has no golden
 * vectors for this path and OpenCV 3.4.1 cannot be run here).
 *
 * Build with -ffp-contract=off: every float expression below is meant to be
 * evaluated as written (one IEEE-754 rounding per operation), which is also
 * what the GPU kernels do.
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime for the budgeted search */
#include "orb_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* OpenCV 3.4.1 semantics: scalar helpers                                     */
/* ------------------------------------------------------------------------- */

/* cvRound(float/double): round-half-to-even (SSE cvtss2si / lrint). */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static const int g_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
const int *oo_pattern(void) { return g_pattern; }

/* ------------------------------------------------------------------------- */
/* Extractor object and constructor tables                                    */
/* ------------------------------------------------------------------------- */

struct oo_extractor {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor; /* the reference keeps this member as double (include/ORBextractor.h:163) */
    float scale[OO_MAX_LEVELS], inv_scale[OO_MAX_LEVELS];
    float sigma2[OO_MAX_LEVELS], inv_sigma2[OO_MAX_LEVELS];
    int nfeat[OO_MAX_LEVELS];
    int umax[OO_HALF_PATCH + 2];
    /* pyramid of the last ComputePyramid call */
    uint8_t *plane[OO_MAX_LEVELS];
    int lw[OO_MAX_LEVELS], lh[OO_MAX_LEVELS];
    int level_counts[OO_MAX_LEVELS];
    int trig_mode; /* OO_TRIG_LIBM (the reference's calls) or OO_TRIG_ROUNDED */
};

/* src/ORBextractor.cc:967-1026 */
oo_extractor *oo_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    if (nlevels < 1 || nlevels > OO_MAX_LEVELS) return NULL;
    oo_extractor *ex = (oo_extractor *)calloc(1, sizeof(*ex));
    ex->nfeatures = nfeatures;
    ex->scaleFactor = scaleFactor;
    ex->nlevels = nlevels;
    ex->iniTh = iniThFAST;
    ex->minTh = minThFAST;
    ex->trig_mode = OO_TRIG_LIBM;

    ex->scale[0] = 1.0f;
    ex->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        ex->scale[i] = (float)(ex->scale[i - 1] * ex->scaleFactor); /* :977 float*double */
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        ex->inv_scale[i] = 1.0f / ex->scale[i];
        ex->inv_sigma2[i] = 1.0f / ex->sigma2[i];
    }

    float factor = (float)(1.0f / ex->scaleFactor); /* :992 */
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        ex->nfeat[level] = cv_round_f(nDesired);
        sum += ex->nfeat[level];
        nDesired *= factor;
    }
    ex->nfeat[nlevels - 1] = imax(nfeatures - sum, 0);

    /* :1010-1025 end of each row of the circular patch */
    int v, v0;
    int vmax = cv_floor_f(OO_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil_f(OO_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = OO_HALF_PATCH * OO_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) ex->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = OO_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
        ex->umax[v] = v0;
        ++v0;
    }
    return ex;
}

void oo_destroy(oo_extractor *ex)
{
    if (!ex) return;
    for (int i = 0; i < OO_MAX_LEVELS; i++) free(ex->plane[i]);
    free(ex);
}

void oo_get_tables(const oo_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                   float *inv_sigma2, int *nfeat_per_level, int *umax16)
{
    for (int i = 0; i < ex->nlevels; i++) {
        if (scale) scale[i] = ex->scale[i];
        if (inv_scale) inv_scale[i] = ex->inv_scale[i];
        if (sigma2) sigma2[i] = ex->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->inv_sigma2[i];
        if (nfeat_per_level) nfeat_per_level[i] = ex->nfeat[i];
    }
    if (umax16) for (int i = 0; i <= OO_HALF_PATCH; i++) umax16[i] = ex->umax[i];
}

void oo_level_size(const oo_extractor *ex, int w, int h, int level, int *lw, int *lh)
{
    float scale = ex->inv_scale[level]; /* :1196-1197 */
    *lw = cv_round_f((float)w * scale);
    *lh = cv_round_f((float)h * scale);
}

/* ------------------------------------------------------------------------- */
/* OpenCV 3.4.1 semantics: cv::resize(INTER_LINEAR) on 8UC1                   */
/* generic path: resizeGeneric_ + HResizeLinear + VResizeLinear<uchar,int,    */
/* short,FixedPtCast<int,uchar,22>>; INTER_RESIZE_COEF_BITS = 11.             */
/* ------------------------------------------------------------------------- */
void oo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                         uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *hrow0 = (int *)malloc(sizeof(int) * dw);
    int *hrow1 = (int *)malloc(sizeof(int) * dw);

    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short(cv_round_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat_short(cv_round_f(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        short b0 = sat_short(cv_round_f((1.f - fy) * 2048));
        short b1 = sat_short(cv_round_f(fy * 2048));
        /* rows sy and sy+1, each clipped into [0, sh-1] */
        int y0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)y0 * sstride, *S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx; /* weight is 0 there */
            hrow0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            hrow1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (hrow0[dx] >> 4)) >> 16) + ((b1 * (hrow1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(hrow0); free(hrow1);
}

/* copyMakeBorder(REFLECT_101) in place around the level image of a bordered plane */
static void fill_border_reflect101(uint8_t *plane, int w, int h)
{
    const int B = OO_EDGE_THRESHOLD, bw = w + 2 * B, bh = h + 2 * B;
    for (int y = 0; y < bh; y++) {
        int sy = reflect101(y - B, h);
        uint8_t *row = plane + (size_t)y * bw;
        const uint8_t *srow = plane + (size_t)(sy + B) * bw;
        for (int x = 0; x < bw; x++) {
            if (y >= B && y < B + h && x >= B && x < B + w) continue;
            row[x] = srow[reflect101(x - B, w) + B];
        }
    }
}

/* src/ORBextractor.cc:1192-1216 */
int oo_compute_pyramid(oo_extractor *ex, const uint8_t *img, int w, int h, int stride)
{
    const int B = OO_EDGE_THRESHOLD;
    for (int level = 0; level < ex->nlevels; level++) {
        int lw, lh;
        oo_level_size(ex, w, h, level, &lw, &lh);
        if (lw < 1 || lh < 1) return -1;
        ex->lw[level] = lw;
        ex->lh[level] = lh;
        free(ex->plane[level]);
        int bw = lw + 2 * B, bh = lh + 2 * B;
        ex->plane[level] = (uint8_t *)calloc((size_t)bw * bh, 1);
        uint8_t *roi = ex->plane[level] + (size_t)B * bw + B;
        if (level != 0) {
            int pbw = ex->lw[level - 1] + 2 * B;
            const uint8_t *proi = ex->plane[level - 1] + (size_t)B * pbw + B;
            oo_resize_linear_u8(proi, ex->lw[level - 1], ex->lh[level - 1], pbw, roi, lw, lh, bw);
        } else {
            for (int y = 0; y < h; y++) memcpy(roi + (size_t)y * bw, img + (size_t)y * stride, w);
        }
        fill_border_reflect101(ex->plane[level], lw, lh);
    }
    return 0;
}

const uint8_t *oo_pyramid_plane(const oo_extractor *ex, int level, int *bw, int *bh)
{
    if (bw) *bw = ex->lw[level] + 2 * OO_EDGE_THRESHOLD;
    if (bh) *bh = ex->lh[level] + 2 * OO_EDGE_THRESHOLD;
    return ex->plane[level];
}

/* ------------------------------------------------------------------------- */
/* OpenCV 3.4.1 semantics: cv::FAST TYPE_9_16 with non-max suppression        */
/* (features2d/src/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>;       */
/* in-repo restatement of the same structure: src/FAST_NEON.cc:91-287)        */
/* ------------------------------------------------------------------------- */
static const int g_ring[16][2] = { /* src/FAST_NEON.cc:3-7: Bresenham circle r=3 */
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static void make_offsets(int pixel[25], int stride)
{
    int k = 0;
    for (; k < 16; k++) pixel[k] = g_ring[k][0] + g_ring[k][1] * stride;
    for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

/* cornerScore<16>: largest threshold for which the pixel is still a corner */
static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int N = 25;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = imin((int)d[k + 1], (int)d[k + 2]);
        a = imin(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, (int)d[k + 4]);
        a = imin(a, (int)d[k + 5]);
        a = imin(a, (int)d[k + 6]);
        a = imin(a, (int)d[k + 7]);
        a = imin(a, (int)d[k + 8]);
        a0 = imax(a0, imin(a, (int)d[k]));
        a0 = imax(a0, imin(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = imax((int)d[k + 1], (int)d[k + 2]);
        b = imax(b, (int)d[k + 3]);
        b = imax(b, (int)d[k + 4]);
        b = imax(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, (int)d[k + 6]);
        b = imax(b, (int)d[k + 7]);
        b = imax(b, (int)d[k + 8]);
        b0 = imin(b0, imax(b, (int)d[k]));
        b0 = imin(b0, imax(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

int oo_fast9_nms(const uint8_t *img, int w, int h, int stride, int threshold, oo_cand *out, int cap)
{
    const int K = 8, N = 25;
    int i, j, k, pixel[25], nout = 0;
    make_offsets(pixel, stride);
    threshold = imin(imax(threshold, 0), 255);

    uint8_t threshold_tab[512];
    for (i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    if (w < 1 || h < 1) return 0;
    uint8_t *bufmem = (uint8_t *)calloc((size_t)w * 3, 1);
    int *cpmem = (int *)calloc((size_t)(w + 1) * 3, sizeof(int));
    uint8_t *buf[3] = {bufmem, bufmem + w, bufmem + 2 * w};
    int *cpbuf[3] = {cpmem + 1, cpmem + 1 + (w + 1), cpmem + 1 + 2 * (w + 1)};

    for (i = 3; i < h - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, w);
        int ncorners = 0;

        if (i < h - 3) {
            for (j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];

                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;

        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                if (nout < cap) {
                    out[nout].x = j;
                    out[nout].y = i - 1;
                    out[nout].response = score;
                }
                nout++;
            }
        }
    }
    free(bufmem);
    free(cpmem);
    return nout;
}

/* Independent derivation: S(p) = max over the 16 arcs A of 9 contiguous ring
 * pixels of max(min_{q in A}(I(p)-I(q)), min_{q in A}(I(q)-I(p))) - 1.
 * "p is a corner at threshold t" <=> S(p) >= t, and the stored score is S(p). */
int oo_fast_score_bruteforce(const uint8_t *center, int stride)
{
    int v = center[0], best = -1000;
    for (int s = 0; s < 16; s++) {
        int mn_dark = 1000, mn_bright = 1000;
        for (int j = 0; j < 9; j++) {
            int k = (s + j) & 15;
            int p = center[g_ring[k][0] + g_ring[k][1] * stride];
            mn_dark = imin(mn_dark, v - p);
            mn_bright = imin(mn_bright, p - v);
        }
        best = imax(best, imax(mn_dark, mn_bright));
    }
    return best - 1;
}

/* ------------------------------------------------------------------------- */
/* Cell loop of ComputeKeyPointsOctTree, src/ORBextractor.cc:1036-1101        */
/* ------------------------------------------------------------------------- */
static int detect_level_literal(const oo_extractor *ex, int level, oo_cand *out, int cap)
{
    const int B = OO_EDGE_THRESHOLD;
    const int cols = ex->lw[level], rows = ex->lh[level];
    const int pitch = cols + 2 * B;
    const uint8_t *im = ex->plane[level] + (size_t)B * pitch + B;
    const float W = 30;
    const int minBorderX = OO_EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = cols - OO_EDGE_THRESHOLD + 3, maxBorderY = rows - OO_EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return 0; /* reference would divide by zero */
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int nout = 0;
    int cellcap = (wCell + 6) * (hCell + 6);
    oo_cand *cell = (oo_cand *)malloc(sizeof(oo_cand) * cellcap);

    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            const uint8_t *sub = im + (size_t)y0 * pitch + x0;
            int n = oo_fast9_nms(sub, x1 - x0, y1 - y0, pitch, ex->iniTh, cell, cellcap);
            if (n == 0) n = oo_fast9_nms(sub, x1 - x0, y1 - y0, pitch, ex->minTh, cell, cellcap);
            for (int k = 0; k < n; k++) {
                if (nout < cap) {
                    out[nout].x = cell[k].x + j * wCell;
                    out[nout].y = cell[k].y + i * hCell;
                    out[nout].response = cell[k].response;
                }
                nout++;
            }
        }
    }
    free(cell);
    return nout;
}

/* Single-pass equivalent: one threshold-free score image; in-cell strict
 * local maxima with S >= minTh; a cell that owns a maximum with S >= iniTh
 * keeps only those. The tested areas of the cells tile
 * [3,width-3) x [3,height-3) (coordinates relative to (16,16)). */
static int detect_level_scoremap(const oo_extractor *ex, int level, oo_cand *out, int cap)
{
    const int B = OO_EDGE_THRESHOLD;
    const int cols = ex->lw[level], rows = ex->lh[level];
    const int pitch = cols + 2 * B;
    const uint8_t *im = ex->plane[level] + (size_t)B * pitch + B;
    const int width = cols - 2 * 16, height = rows - 2 * 16;
    const int nCols = (int)((float)width / 30.f), nRows = (int)((float)height / 30.f);
    if (nCols < 1 || nRows < 1) return 0;
    const int wCell = (int)ceilf((float)width / nCols), hCell = (int)ceilf((float)height / nRows);
    int thr_min = imin(imax(ex->minTh, 0), 255), thr_ini = imin(imax(ex->iniTh, 0), 255);
    int nout = 0;
    if (width < 7 || height < 7) return 0;

    /* score image over relative coordinates, zero outside [3,width-3)x[3,height-3) */
    uint8_t *S = (uint8_t *)calloc((size_t)width * height, 1);
    for (int y = 3; y < height - 3; y++)
        for (int x = 3; x < width - 3; x++) {
            int s = oo_fast_score_bruteforce(im + (size_t)(y + 16) * pitch + (x + 16), pitch);
            S[(size_t)y * width + x] = (uint8_t)((s >= thr_min && s > 0) ? s : 0);
        }
    /* a corner with S == 0 (only possible for threshold 0) is stored as 0 and can
     * never be a strict maximum over zero neighbours -- same as the literal path */

    for (int ci = 0; ci < nRows; ci++) {
        int ty0 = ci * hCell + 3, ty1 = imin((ci + 1) * hCell + 3, height - 3);
        for (int cj = 0; cj < nCols; cj++) {
            int tx0 = cj * wCell + 3, tx1 = imin((cj + 1) * wCell + 3, width - 3);
            if (ty0 >= ty1 || tx0 >= tx1) continue;
            int any_ini = 0;
            for (int pass = 0; pass < 2; pass++) {
                for (int y = ty0; y < ty1; y++)
                    for (int x = tx0; x < tx1; x++) {
                        int s = S[(size_t)y * width + x];
                        if (s == 0) continue;
                        int ok = 1;
                        for (int dy = -1; dy <= 1 && ok; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (!dx && !dy) continue;
                                int xx = x + dx, yy = y + dy;
                                int q = (xx >= tx0 && xx < tx1 && yy >= ty0 && yy < ty1)
                                            ? S[(size_t)yy * width + xx] : 0;
                                if (!(s > q)) { ok = 0; break; }
                            }
                        if (!ok) continue;
                        if (pass == 0) {
                            if (s >= thr_ini) any_ini = 1;
                        } else if (!any_ini || s >= thr_ini) {
                            if (nout < cap) {
                                out[nout].x = x;
                                out[nout].y = y;
                                out[nout].response = s;
                            }
                            nout++;
                        }
                    }
            }
        }
    }
    free(S);
    return nout;
}

int oo_detect_level(const oo_extractor *ex, int level, int literal, oo_cand *out, int cap)
{
    return literal ? detect_level_literal(ex, level, out, cap) : detect_level_scoremap(ex, level, out, cap);
}

/* ------------------------------------------------------------------------- */
/* DistributeOctTree -- literal std::list emulation                           */
/* src/ORBextractor.cc:488-711, ExtractorNode::DivideNode :431-486            */
/* ------------------------------------------------------------------------- */
typedef struct lnode {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int *keys;
    int nkeys;
    int noMore;
    long addr; /* creation sequence number: stand-in for the heap address */
    struct lnode *prev, *next;
} lnode;

typedef struct { lnode *head, *tail; int size; long next_addr; } llist;

static lnode *node_new(int cap)
{
    lnode *n = (lnode *)calloc(1, sizeof(lnode));
    n->keys = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    return n;
}
static void node_free(lnode *n) { free(n->keys); free(n); }
static void list_push_front(llist *l, lnode *n)
{
    n->addr = l->next_addr++;
    n->prev = NULL; n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n; l->size++;
}
static void list_push_back(llist *l, lnode *n)
{
    n->addr = l->next_addr++;
    n->next = NULL; n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n; l->size++;
}
static lnode *list_erase(llist *l, lnode *n) /* returns next */
{
    lnode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    node_free(n);
    return nx;
}

static void divide_node(const lnode *p, const oo_cand *c, lnode *ch[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    lnode *n1 = ch[0] = node_new(p->nkeys), *n2 = ch[1] = node_new(p->nkeys);
    lnode *n3 = ch[2] = node_new(p->nkeys), *n4 = ch[3] = node_new(p->nkeys);
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;

    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;

    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;

    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;

    for (int i = 0; i < p->nkeys; i++) {
        const oo_cand *kp = &c[p->keys[i]];
        lnode *dst;
        if ((float)kp->x < (float)n1->URx) dst = ((float)kp->y < (float)n1->BRy) ? n1 : n3;
        else dst = ((float)kp->y < (float)n1->BRy) ? n2 : n4;
        dst->keys[dst->nkeys++] = p->keys[i];
    }
    for (int q = 0; q < 4; q++)
        if (ch[q]->nkeys == 1) ch[q]->noMore = 1;
}

typedef struct { int size; lnode *node; } sizeptr;
static int sizeptr_cmp(const void *a, const void *b)
{
    const sizeptr *x = (const sizeptr *)a, *y = (const sizeptr *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    if (x->node->addr != y->node->addr) return x->node->addr < y->node->addr ? -1 : 1;
    return 0;
}

static int octree_literal(const oo_cand *c, int n, int width, int height, int N, int *out_idx, int cap)
{
    const int nIni = (int)roundf((float)width / (float)height);
    if (nIni < 1) return -2;
    const float hX = (float)width / nIni;
    llist L = {0};
    lnode **ini = (lnode **)malloc(sizeof(lnode *) * nIni);
    for (int i = 0; i < nIni; i++) {
        lnode *ni = node_new(n);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = height;
        ni->BRx = ni->URx; ni->BRy = height;
        list_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)c[i].x / hX);
        if (r < 0 || r >= nIni) { free(ini); return -3; }
        ini[r]->keys[ini[r]->nkeys++] = i;
    }
    free(ini);
    for (lnode *lit = L.head; lit;) {
        if (lit->nkeys == 1) { lit->noMore = 1; lit = lit->next; }
        else if (lit->nkeys == 0) lit = list_erase(&L, lit);
        else lit = lit->next;
    }

    int bFinish = 0;
    sizeptr *vSP = (sizeptr *)malloc(sizeof(sizeptr) * (4 * (size_t)n + 16));
    sizeptr *vPrev = (sizeptr *)malloc(sizeof(sizeptr) * (4 * (size_t)n + 16));
    int nSP = 0;

    while (!bFinish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nSP = 0;
        lnode *lit = L.head;
        while (lit) {
            if (lit->noMore) { lit = lit->next; continue; }
            lnode *ch[4];
            divide_node(lit, c, ch);
            for (int q = 0; q < 4; q++) {
                if (ch[q]->nkeys > 0) {
                    list_push_front(&L, ch[q]);
                    if (ch[q]->nkeys > 1) {
                        nToExpand++;
                        vSP[nSP].size = ch[q]->nkeys; vSP[nSP].node = ch[q]; nSP++;
                    }
                } else
                    node_free(ch[q]);
            }
            lit = list_erase(&L, lit);
        }

        if (L.size >= N || L.size == prevSize) {
            bFinish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = L.size;
                int nPrev = nSP;
                memcpy(vPrev, vSP, sizeof(sizeptr) * nSP);
                nSP = 0;
                qsort(vPrev, nPrev, sizeof(sizeptr), sizeptr_cmp);
                for (int j = nPrev - 1; j >= 0; j--) {
                    lnode *ch[4];
                    divide_node(vPrev[j].node, c, ch);
                    for (int q = 0; q < 4; q++) {
                        if (ch[q]->nkeys > 0) {
                            list_push_front(&L, ch[q]);
                            if (ch[q]->nkeys > 1) {
                                vSP[nSP].size = ch[q]->nkeys; vSP[nSP].node = ch[q]; nSP++;
                            }
                        } else
                            node_free(ch[q]);
                    }
                    list_erase(&L, vPrev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vSP);
    free(vPrev);

    int nout = 0;
    for (lnode *lit = L.head; lit; lit = lit->next) {
        int best = lit->keys[0];
        float maxResponse = (float)c[best].response;
        for (int k = 1; k < lit->nkeys; k++)
            if ((float)c[lit->keys[k]].response > maxResponse) {
                best = lit->keys[k];
                maxResponse = (float)c[best].response;
            }
        if (nout < cap) out_idx[nout] = best;
        nout++;
    }
    while (L.head) list_erase(&L, L.head);
    return nout;
}

/* ------------------------------------------------------------------------- */
/* DistributeOctTree -- array formulation (what the GPU kernel implements)    */
/*                                                                            */
/* The list is an array in list order; a pass computes every child partition  */
/* independently and rebuilds the array with the list discipline              */
/* "children of the parent processed last come first, each as n4,n3,n2,n1,    */
/* then the surviving old nodes in their old order".  The size-ordered sweep  */
/* stops at the first index, in (size, creation) descending order, where      */
/* |L| + inclusive_scan(children-1) >= N.                                     */
/* ------------------------------------------------------------------------- */
typedef struct { int x0, x1, y0, y1, cnt; } anode;

static int quadrant_of(const anode *nd, int x, int y)
{
    const int hx = (int)ceilf((float)(nd->x1 - nd->x0) / 2);
    const int hy = (int)ceilf((float)(nd->y1 - nd->y0) / 2);
    int right = !((float)x < (float)(nd->x0 + hx));
    int down = !((float)y < (float)(nd->y0 + hy));
    return right + 2 * down; /* 0:n1 1:n2 2:n3 3:n4 */
}
static void child_rect(const anode *p, int q, anode *c)
{
    const int hx = (int)ceilf((float)(p->x1 - p->x0) / 2);
    const int hy = (int)ceilf((float)(p->y1 - p->y0) / 2);
    c->x0 = (q & 1) ? p->x0 + hx : p->x0;
    c->x1 = (q & 1) ? p->x1 : p->x0 + hx;
    c->y0 = (q & 2) ? p->y0 + hy : p->y0;
    c->y1 = (q & 2) ? p->y1 : p->y0 + hy;
}

typedef struct { int cnt, cidx, pos; } sortent;
static int sortent_cmp_desc(const void *a, const void *b)
{
    const sortent *x = (const sortent *)a, *y = (const sortent *)b;
    if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;
    if (x->cidx != y->cidx) return x->cidx > y->cidx ? -1 : 1;
    return 0;
}

static int octree_arrays(const oo_cand *c, int n, int width, int height, int N, int *out_idx, int cap)
{
    const int nIni = (int)roundf((float)width / (float)height);
    if (nIni < 1) return -2;
    const float hX = (float)width / nIni;
    int maxNodes = imax(N + 4, 4 * nIni + 4) + 4;
    anode *cur = (anode *)calloc(maxNodes, sizeof(anode)), *nxt = (anode *)calloc(maxNodes, sizeof(anode));
    int *knode = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int *kq = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int *cc = (int *)calloc((size_t)maxNodes * 4, sizeof(int));
    int *childpos = (int *)calloc((size_t)maxNodes * 4, sizeof(int));
    int *newpos = (int *)calloc(maxNodes, sizeof(int));
    int *split = (int *)calloc(maxNodes, sizeof(int));
    int *cidx_cur = (int *)calloc(maxNodes, sizeof(int)); /* creation index among expandable nodes, -1 if not one */
    int *cidx_nxt = (int *)calloc(maxNodes, sizeof(int));
    sortent *se = (sortent *)malloc(sizeof(sortent) * maxNodes);
    int size = 0, result = 0;

    /* roots :492-534 */
    {
        int *rcount = (int *)calloc(nIni, sizeof(int));
        int *rpos = (int *)malloc(sizeof(int) * nIni);
        for (int i = 0; i < n; i++) {
            int r = (int)((float)c[i].x / hX);
            if (r < 0 || r >= nIni) { result = -3; free(rcount); free(rpos); goto done; }
            kq[i] = r;
            rcount[r]++;
        }
        for (int r = 0; r < nIni; r++) {
            rpos[r] = -1;
            if (rcount[r] == 0) continue;
            rpos[r] = size;
            cur[size].x0 = (int)(hX * (float)r);
            cur[size].x1 = (int)(hX * (float)(r + 1));
            cur[size].y0 = 0;
            cur[size].y1 = height;
            cur[size].cnt = rcount[r];
            size++;
        }
        for (int i = 0; i < n; i++) knode[i] = rpos[kq[i]];
        free(rcount); free(rpos);
    }

    int phase2 = 0, finish = 0;
    while (!finish) {
        int prevSize = size;
        /* which nodes are split in this step */
        int nsplit = 0;
        if (!phase2) {
            for (int p = 0; p < size; p++) { split[p] = cur[p].cnt > 1; nsplit += split[p]; }
        }
        /* child occupancy of every node that might be split */
        for (int p = 0; p < size; p++) cc[4 * p] = cc[4 * p + 1] = cc[4 * p + 2] = cc[4 * p + 3] = 0;
        for (int i = 0; i < n; i++) {
            int p = knode[i];
            if (phase2 ? cidx_cur[p] >= 0 : split[p]) {
                kq[i] = quadrant_of(&cur[p], c[i].x, c[i].y);
                cc[4 * p + kq[i]]++;
            }
        }
        /* processing order of the split parents */
        int norder = 0;
        if (!phase2) {
            for (int p = 0; p < size; p++)
                if (split[p]) { se[norder].pos = p; se[norder].cnt = cur[p].cnt; se[norder].cidx = norder; norder++; }
        } else {
            for (int p = 0; p < size; p++) {
                split[p] = 0;
                if (cidx_cur[p] >= 0) { se[norder].pos = p; se[norder].cnt = cur[p].cnt; se[norder].cidx = cidx_cur[p]; norder++; }
            }
            qsort(se, norder, sizeof(sortent), sortent_cmp_desc);
            /* break position: first s with size + sum_{s'<=s} (children-1) >= N */
            int run = size, cut = norder;
            for (int s = 0; s < norder; s++) {
                int p = se[s].pos;
                int nch = (cc[4 * p] > 0) + (cc[4 * p + 1] > 0) + (cc[4 * p + 2] > 0) + (cc[4 * p + 3] > 0);
                run += nch - 1;
                if (run >= N) { cut = s + 1; break; }
            }
            norder = cut;
            for (int s = 0; s < norder; s++) split[se[s].pos] = 1;
        }
        /* new list: children of the parent processed last first (n4..n1), ... */
        int np = 0, ncreated = 0;
        for (int s = norder - 1; s >= 0; s--) {
            int p = se[s].pos;
            for (int q = 3; q >= 0; q--)
                if (cc[4 * p + q] > 0) {
                    childpos[4 * p + q] = np;
                    child_rect(&cur[p], q, &nxt[np]);
                    nxt[np].cnt = cc[4 * p + q];
                    np++;
                }
        }
        /* creation order (for the next size sort): processing order, then n1..n4 */
        for (int k = 0; k < np; k++) cidx_nxt[k] = -1;
        for (int s = 0; s < norder; s++) {
            int p = se[s].pos;
            for (int q = 0; q < 4; q++)
                if (cc[4 * p + q] > 1) cidx_nxt[childpos[4 * p + q]] = ncreated++;
        }
        for (int p = 0; p < size; p++)
            if (!split[p]) {
                newpos[p] = np;
                nxt[np] = cur[p];
                cidx_nxt[np] = -1; /* old nodes are never revisited by the size-ordered sweep */
                np++;
            }
        if (!phase2) {
            /* phase 1 revisits every multi-key node, old or new: none are old */
        }
        for (int i = 0; i < n; i++) {
            int p = knode[i];
            knode[i] = split[p] ? childpos[4 * p + kq[i]] : newpos[p];
        }
        { anode *t = cur; cur = nxt; nxt = t; }
        { int *t = cidx_cur; cidx_cur = cidx_nxt; cidx_nxt = t; }
        size = np;
        if (size > maxNodes - 4) { result = -4; goto done; }

        if (size >= N || size == prevSize) finish = 1;
        else if (!phase2 && size + ncreated * 3 > N) phase2 = 1;
    }

    /* best key of every node: max response, first in key order on ties (:689-710) */
    {
        int *best = (int *)malloc(sizeof(int) * (size > 0 ? size : 1));
        for (int p = 0; p < size; p++) best[p] = -1;
        for (int i = 0; i < n; i++) {
            int p = knode[i];
            if (best[p] < 0 || c[i].response > c[best[p]].response) best[p] = i;
        }
        for (int p = 0; p < size; p++) {
            if (p < cap) out_idx[p] = best[p];
        }
        result = size;
        free(best);
    }
done:
    free(cur); free(nxt); free(knode); free(kq); free(cc); free(childpos); free(newpos);
    free(split); free(cidx_cur); free(cidx_nxt); free(se);
    return result;
}

int oo_distribute_octree(const oo_cand *cands, int n, int width, int height, int N,
                         int literal, int *out_idx, int cap)
{
    return literal ? octree_literal(cands, n, width, height, N, out_idx, cap)
                   : octree_arrays(cands, n, width, height, N, out_idx, cap);
}

/* ------------------------------------------------------------------------- */
/* Orientation: IC_Angle src/ORBextractor.cc:74-101 + cv::fastAtan2           */
/* ------------------------------------------------------------------------- */
void oo_ic_moments(const uint8_t *center, int stride, const int *umax, int *m01_out, int *m10_out)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -OO_HALF_PATCH; u <= OO_HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= OO_HALF_PATCH; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    *m01_out = m_01;
    *m10_out = m_10;
}

/* OpenCV 3.4.1 semantics: cv::fastAtan2 (core/src/mathfuncs_core.simd.hpp atan_f32) */
float oo_fast_atan2(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------------- */
/* OpenCV 3.4.1 semantics: GaussianBlur 7x7, sigma 2, 8-bit                   */
/* separable kernel round(256*g) = [18,34,49,55,49,34,18] (sum 257, not        */
/* renormalised), exact integer sums, dst = sat_u8((acc + 32768) >> 16).       */
/* ------------------------------------------------------------------------- */
static void gauss7_kernel(int k[7])
{
    /* getGaussianKernel(7, 2): exp(-x^2/(2 sigma^2)) normalised, then fixed point 8.8 */
    double g[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; g[i] = exp(-0.5 * x * x / 4.0); sum += g[i]; }
    for (int i = 0; i < 7; i++) k[i] = (int)lrint(256.0 * g[i] / sum);
}

void oo_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    int k[7];
    gauss7_kernel(k);
    int *rows = (int *)malloc(sizeof(int) * (size_t)w * (h + 6));
    for (int y = -3; y < h + 3; y++) {
        const uint8_t *s = src + (ptrdiff_t)y * sstride;
        int *r = rows + (size_t)(y + 3) * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int t = 0; t < 7; t++) acc += k[t] * s[x + t - 3];
            r[x] = acc;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int t = 0; t < 7; t++) acc += k[t] * rows[(size_t)(y + t) * w + x];
            int v = (acc + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    free(rows);
}

/* ------------------------------------------------------------------------- */
/* computeOrbDescriptor src/ORBextractor.cc:103-150                           */
/* ------------------------------------------------------------------------- */
void oo_set_trig_mode(oo_extractor *ex, int mode) { ex->trig_mode = mode; }

/* The reference writes cos(angle) / sin(angle) on a float under `using namespace std`
 * (src/ORBextractor.cc:64-65, :108-109): the float overloads, i.e. libm's cosf / sinf, whose
 * last bit depends on the libm in use (they are not correctly rounded).  OO_TRIG_LIBM calls this
 * host's cosf / sinf as the reference does; OO_TRIG_ROUNDED takes the correctly rounded values
 * (double cos / sin rounded to float), the platform-independent formulation the GPU kernel
 * evaluates.  See the count of differing angles in orb_oracle.h. */
void oo_orb_descriptor_mode(const uint8_t *center, int stride, float angle_deg, uint8_t desc[32], int trig_mode)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a, b;
    if (trig_mode == OO_TRIG_LIBM) { a = cosf(angle); b = sinf(angle); }
    else { a = (float)cos((double)angle); b = (float)sin((double)angle); }
    const int *pat = g_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int x0 = pat[4 * k], y0 = pat[4 * k + 1], x1 = pat[4 * k + 2], y1 = pat[4 * k + 3];
            float r0 = x0 * b + y0 * a, c0 = x0 * a - y0 * b;
            float r1 = x1 * b + y1 * a, c1 = x1 * a - y1 * b;
            int t0 = center[cv_round_f(r0) * stride + cv_round_f(c0)];
            int t1 = center[cv_round_f(r1) * stride + cv_round_f(c1)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

void oo_orb_descriptor(const uint8_t *center, int stride, float angle_deg, uint8_t desc[32])
{
    oo_orb_descriptor_mode(center, stride, angle_deg, desc, OO_TRIG_LIBM);
}

/* number of float angles in [0, 360) degrees (every value cv::fastAtan2 can return lies there)
 * whose libm cosf / sinf differ from the correctly rounded value; counts[0] = cos, counts[1] = sin.
 * Walks the floats first, first + step, ... below 360.0f (first = 0, step = 1: all 1 135 869 952). */
void oo_count_trig_differences(uint32_t first, uint32_t step, uint64_t counts[2], uint64_t *tested)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const uint32_t end = 0x43B40000u; /* bit pattern of 360.0f */
    uint64_t c = 0, s = 0, n = 0;
    if (step == 0) step = 1;
    for (uint64_t u = first; u < end; u += step) {
        uint32_t bits = (uint32_t)u;
        float deg;
        memcpy(&deg, &bits, 4);
        volatile float x = deg * factorPI;
        const float xa = x;
        c += cosf(xa) != (float)cos((double)xa);
        s += sinf(xa) != (float)sin((double)xa);
        n++;
    }
    counts[0] = c; counts[1] = s; *tested = n;
}

/* ------------------------------------------------------------------------- */
/* ORBextractor::operator() src/ORBextractor.cc:1127-1190 and                 */
/* ComputeKeyPointsOctTree :1028-1125                                         */
/* ------------------------------------------------------------------------- */
int oo_extract(oo_extractor *ex, const uint8_t *img, int w, int h, int stride,
               oo_keypoint *kps, uint8_t *desc, int cap, int literal)
{
    const int B = OO_EDGE_THRESHOLD;
    if (!img || w <= 0 || h <= 0) return 0;
    if (oo_compute_pyramid(ex, img, w, h, stride) != 0) return -1;
    int total = 0;
    for (int level = 0; level < ex->nlevels; level++) {
        ex->level_counts[level] = 0;
        const int cols = ex->lw[level], rows = ex->lh[level];
        const int pitch = cols + 2 * B;
        const int width = cols - 32, height = rows - 32;
        if (width < 30 || height < 30) continue; /* reference would divide by zero (nCols/nRows == 0) */
        int ccap = (width * height) / 4 + 64;
        oo_cand *cand = (oo_cand *)malloc(sizeof(oo_cand) * ccap);
        int ncand = oo_detect_level(ex, level, literal, cand, ccap);
        if (ncand > ccap) { free(cand); return -1; }
        int ocap = imax(ex->nfeat[level] + 8, 4 * (width / height + 2) + 8);
        int *sel = (int *)malloc(sizeof(int) * ocap);
        int nsel = oo_distribute_octree(cand, ncand, width, height, ex->nfeat[level], literal, sel, ocap);
        if (nsel < 0 || nsel > ocap) { free(cand); free(sel); return -1; }
        if (total + nsel > cap) { free(cand); free(sel); return -1; }

        const uint8_t *im = ex->plane[level] + (size_t)B * pitch + B;
        const int scaledPatchSize = (int)(OO_PATCH * ex->scale[level]); /* :1109 */
        uint8_t *blur = NULL;
        if (nsel > 0) {
            blur = (uint8_t *)malloc((size_t)cols * rows);
            oo_gaussian_blur7(im, cols, rows, pitch, blur, cols);
        }
        for (int k = 0; k < nsel; k++) {
            oo_keypoint *kp = &kps[total + k];
            kp->x = (float)(cand[sel[k]].x + 16); /* :1115-1118 */
            kp->y = (float)(cand[sel[k]].y + 16);
            kp->octave = level;
            kp->size = (float)scaledPatchSize;
            kp->response = (float)cand[sel[k]].response;
            kp->class_id = -1;
            int cx = cv_round_f(kp->x), cy = cv_round_f(kp->y);
            int m01, m10;
            oo_ic_moments(im + (size_t)cy * pitch + cx, pitch, ex->umax, &m01, &m10);
            kp->angle = oo_fast_atan2((float)m01, (float)m10);
            oo_orb_descriptor_mode(blur + (size_t)cy * cols + cx, cols, kp->angle, desc + (size_t)(total + k) * 32, ex->trig_mode);
            if (level != 0) { /* :1179-1186 */
                float scale = ex->scale[level];
                kp->x *= scale;
                kp->y *= scale;
            }
        }
        free(blur);
        free(cand);
        free(sel);
        ex->level_counts[level] = nsel;
        total += nsel;
    }
    return total;
}

void oo_last_level_counts(const oo_extractor *ex, int *counts)
{
    for (int i = 0; i < ex->nlevels; i++) counts[i] = ex->level_counts[i];
}

/* ------------------------------------------------------------------------- */
/* Matching                                                                   */
/* ------------------------------------------------------------------------- */
/* src/ORBmatcher.cc:1768-1784 (SWAR popcount) */
int oo_hamming(const uint32_t pa[8], const uint32_t pb[8])
{
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void oo_match_bruteforce(const uint32_t *q, int Q, const uint32_t *t, int T,
                         int32_t *best_idx, int32_t *best_dist, int32_t *second_dist)
{
    for (int i = 0; i < Q; i++) {
        int bestDist = 256, bestDist2 = 256, bestIdx = -1;
        for (int j = 0; j < T; j++) {
            int dist = oo_hamming(q + 8 * (size_t)i, t + 8 * (size_t)j);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2;
    }
}

void oo_match_windowed(const uint32_t *q, int Q, const uint32_t *t, int T,
                       const int32_t *cand_off, const int32_t *cand_idx, const int32_t *t_octave,
                       int32_t *best_idx, int32_t *best_dist, int32_t *second_dist,
                       int32_t *best_lvl, int32_t *second_lvl)
{
    (void)T;
    for (int i = 0; i < Q; i++) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = cand_off[i]; k < cand_off[i + 1]; k++) {
            int idx = cand_idx[k];
            int dist = oo_hamming(q + 8 * (size_t)i, t + 8 * (size_t)idx);
            int lvl = t_octave ? t_octave[idx] : 0;
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = lvl; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = lvl; bestDist2 = dist;
            }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2;
        if (best_lvl) best_lvl[i] = bestLevel;
        if (second_lvl) second_lvl[i] = bestLevel2;
    }
}

void oo_hamming_csr(const uint32_t *q, int Q, const uint32_t *t, int T,
                    const int32_t *cand_off, const int32_t *cand_idx, uint16_t *dist_out)
{
    (void)T;
    for (int i = 0; i < Q; i++)
        for (int k = cand_off[i]; k < cand_off[i + 1]; k++)
            dist_out[k] = (uint16_t)oo_hamming(q + 8 * (size_t)i, t + 8 * (size_t)cand_idx[k]);
}

/* include/Frame.h:217-251 + src/Frame.cc:1306-1375 */
void oo_stereo_search(const oo_keypoint *kl, const uint32_t *dl, int nl,
                      const oo_keypoint *kr, const uint32_t *dr, int nr,
                      const float *scale_factors, int nrows, float minD, float maxD, float minX,
                      int32_t *best_idx, int32_t *best_dist)
{
    /* row table (CSR built in two passes; order inside a row = ascending iR) */
    int *cnt = (int *)calloc((size_t)nrows + 1, sizeof(int));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)fminf((float)(nrows - 1), ceilf(kpY + r));
        const int minr = (int)fmaxf(0.0f, floorf(kpY - r));
        for (int yi = minr; yi <= maxr; yi++) cnt[yi + 1]++;
    }
    for (int y = 0; y < nrows; y++) cnt[y + 1] += cnt[y];
    int *fill = (int *)malloc(sizeof(int) * ((size_t)nrows + 1));
    memcpy(fill, cnt, sizeof(int) * ((size_t)nrows + 1));
    int *rowidx = (int *)malloc(sizeof(int) * (cnt[nrows] > 0 ? cnt[nrows] : 1));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)fminf((float)(nrows - 1), ceilf(kpY + r));
        const int minr = (int)fmaxf(0.0f, floorf(kpY - r));
        for (int yi = minr; yi <= maxr; yi++) rowidx[fill[yi]++] = iR;
    }
    for (int iL = 0; iL < nl; iL++) {
        best_idx[iL] = -1;
        best_dist[iL] = 100; /* ORBmatcher::TH_HIGH */
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        if (vL < 0 || vL > nrows - 1) continue;
        int row = (int)vL;
        if (cnt[row + 1] == cnt[row]) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < minX) continue;
        int bestDist = 100, bestIdxR = 0;
        for (int k = cnt[row]; k < cnt[row + 1]; k++) {
            int iR = rowidx[k];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                int dist = oo_hamming(dl + 8 * (size_t)iL, dr + 8 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        best_idx[iL] = bestIdxR;
        best_dist[iL] = bestDist;
    }
    free(cnt); free(fill); free(rowidx);
}

typedef struct { int dist, idx; } distidx;
static int distidx_cmp(const void *a, const void *b)
{
    const distidx *x = (const distidx *)a, *y = (const distidx *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* src/Frame.cc:1380-1419 */
int oo_stereo_finish(const oo_keypoint *kl, int nl, const oo_keypoint *kr,
                     const int32_t *best_idx, const int32_t *best_dist,
                     float mbf, float minD, float maxD, float *u_right, float *depth)
{
    const int thOrbDist = (100 + 50) / 2;
    distidx *v = (distidx *)malloc(sizeof(distidx) * (nl > 0 ? nl : 1));
    int nv = 0, nmatched = 0;
    for (int iL = 0; iL < nl; iL++) {
        u_right[iL] = -1.0f;
        depth[iL] = -1.0f;
        if (best_idx[iL] < 0) continue;
        if (best_dist[iL] < thOrbDist) {
            float bestuR = kr[best_idx[iL]].x;
            float disparity = kl[iL].x - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = (float)(kl[iL].x - 0.01);
                }
                depth[iL] = mbf / disparity;
                u_right[iL] = bestuR;
                v[nv].dist = best_dist[iL]; v[nv].idx = iL; nv++;
            }
        }
        nmatched++;
    }
    if (nv > 0) {
        qsort(v, nv, sizeof(distidx), distidx_cmp);
        const float median = (float)v[nv / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; i--) {
            if ((float)v[i].dist < thDist) break;
            u_right[v[i].idx] = -1;
            depth[v[i].idx] = -1;
            nmatched--;
        }
    }
    free(v);
    return nmatched;
}

/* ------------------------------------------------------------------------- */
/* Delayed stereo matching of the default build (DELAYED_STEREO_MATCHING,       */
/* include/Frame.h:40): the Frame members it reads and writes are the arrays    */
/* of oo_stereo_state.                                                         */
/* ------------------------------------------------------------------------- */
typedef struct { int *start, *idx; } oo_rowtab;

/* mvRowIndices, include/Frame.h:228-246 (order inside a row = ascending iR) */
static void rowtab_build(oo_rowtab *rt, const oo_keypoint *kr, int nr, const float *scale_factors, int nrows)
{
    rt->start = (int *)calloc((size_t)nrows + 1, sizeof(int));
    for (int pass = 0; pass < 2; pass++) {
        int *fill = NULL;
        if (pass == 1) {
            for (int y = 0; y < nrows; y++) rt->start[y + 1] += rt->start[y];
            rt->idx = (int *)malloc(sizeof(int) * (rt->start[nrows] > 0 ? rt->start[nrows] : 1));
            fill = (int *)malloc(sizeof(int) * ((size_t)nrows + 1));
            memcpy(fill, rt->start, sizeof(int) * ((size_t)nrows + 1));
        }
        for (int iR = 0; iR < nr; iR++) {
            const float kpY = kr[iR].y;
            const float r = 2.0f * scale_factors[kr[iR].octave];
            const int maxr = (int)fminf((float)(nrows - 1), ceilf(kpY + r));
            const int minr = (int)fmaxf(0.0f, floorf(kpY - r));
            for (int yi = minr; yi <= maxr; yi++) {
                if (pass == 0) rt->start[yi + 1]++;
                else rt->idx[fill[yi]++] = iR;
            }
        }
        free(fill);
    }
}
static void rowtab_free(oo_rowtab *rt) { free(rt->start); free(rt->idx); }

/* Frame::PrepareStereoCandidates, include/Frame.h:217-251: the state a Frame starts from */
void oo_stereo_prepare(oo_stereo_state *st, int nl)
{
    for (int i = 0; i < nl; i++) {
        st->u_right[i] = -1.0f;
        st->depth[i] = -1.0f;
        st->stereo_matched[i] = 0;
    }
    st->n_dist_idx = 0;
}

/* The part the two reference functions share word for word: src/Frame.cc:1189-1257
 * (ComputeStereoMatch_OnePoint) and :1306-1395 (loop body of ComputeStereoMatches_Undistorted).
 * min_u_limit is 0 in the former (:1207) and mnMinX in the latter (:1344).
 * Returns 0 if the keypoint leaves before the scan (return false / continue), 1 if it was scanned
 * without an accepted disparity, 2 if a disparity was stored. */
static int stereo_one_left(const oo_keypoint *kl, const uint32_t *dl, int iL, const oo_keypoint *kr, const uint32_t *dr,
                           const oo_rowtab *rt, int nrows, int thOrbDist, float minD, float maxD, float mbf,
                           float min_u_limit, oo_stereo_state *st, int32_t *best_idx, int32_t *best_dist)
{
    const int levelL = kl[iL].octave;
    const float vL = kl[iL].y, uL = kl[iL].x;
    if (vL < 0 || vL > nrows - 1) return 0;
    const int row = (int)vL;
    if (rt->start[row + 1] == rt->start[row]) return 0;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < min_u_limit) return 0;
    int bestDist = 100; /* ORBmatcher::TH_HIGH */
    int bestIdxR = 0;
    for (int k = rt->start[row]; k < rt->start[row + 1]; k++) {
        const int iR = rt->idx[k];
        if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
        const float uR = kr[iR].x;
        if (uR >= minU && uR <= maxU) {
            const int dist = oo_hamming(dl + 8 * (size_t)iL, dr + 8 * (size_t)iR);
            if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
        }
    }
    if (best_idx) best_idx[iL] = bestIdxR;
    if (best_dist) best_dist[iL] = bestDist;
    if (bestDist < thOrbDist) {
        float bestuR = kr[bestIdxR].x;
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01f;
                bestuR = (float)(uL - 0.01);
            }
            st->depth[iL] = mbf / disparity;
            st->u_right[iL] = bestuR;
            st->dist_idx_dist[st->n_dist_idx] = bestDist;
            st->dist_idx_il[st->n_dist_idx] = iL;
            st->n_dist_idx++;
            return 2;
        }
    }
    return 1;
}

/* Frame::ComputeStereoMatches_Undistorted(isOnline), src/Frame.cc:1262-1422, as compiled with
 * DELAYED_STEREO_MATCHING.  has_mappoint[iL] = (mvpMapPoints[iL] != NULL), NULL = no map points;
 * prior_disp[iL] = mbf / Pc.z of a map point that is not bad and lies in front of the camera
 * (:1325-1334), <= 0 where there is none; disparity_thres = DISPARITY_THRES (include/Frame.h:47).
 * best_idx / best_dist (may be NULL) are written for scanned keypoints only.
 * With an empty mvDistIdx the reference reads element 0 of an empty vector (:1403); the cut is
 * skipped here. */
int oo_stereo_undistorted(const oo_keypoint *kl, const uint32_t *dl, int nl,
                          const oo_keypoint *kr, const uint32_t *dr, int nr,
                          const float *scale_factors, int nrows, float mbf, float mb, float minX,
                          float disparity_thres, const uint8_t *has_mappoint, const float *prior_disp,
                          int is_online, oo_stereo_state *st, int32_t *best_idx, int32_t *best_dist)
{
    const int thOrbDist = (100 + 50) / 2;
    oo_rowtab rt;
    rowtab_build(&rt, kr, nr, scale_factors, nrows);
    const float minZ = mb;
    int nmatched = 0;
    for (int iL = 0; iL < nl; iL++) {
        const int mp = has_mappoint ? has_mappoint[iL] : 0;
        if (is_online) {
            if (!mp || st->stereo_matched[iL]) continue;
        } else {
            if (st->stereo_matched[iL]) continue;
        }
        float minD = 0;
        float maxD = mbf / minZ;
        st->stereo_matched[iL] = 1;
        if (mp && prior_disp && prior_disp[iL] > 0) {
            const float disp = prior_disp[iL];
            minD = fmaxf(disp - disparity_thres, 0.0f);
            maxD = fminf(disp + disparity_thres, mbf / mb);
        }
        if (stereo_one_left(kl, dl, iL, kr, dr, &rt, nrows, thOrbDist, minD, maxD, mbf, minX, st, best_idx, best_dist) == 0)
            continue;
        nmatched++;
    }
    if (!is_online && st->n_dist_idx > 0) {
        const int nv = st->n_dist_idx;
        distidx *v = (distidx *)malloc(sizeof(distidx) * nv);
        for (int i = 0; i < nv; i++) { v[i].dist = st->dist_idx_dist[i]; v[i].idx = st->dist_idx_il[i]; }
        qsort(v, nv, sizeof(distidx), distidx_cmp);
        for (int i = 0; i < nv; i++) { st->dist_idx_dist[i] = v[i].dist; st->dist_idx_il[i] = v[i].idx; }
        const float median = (float)v[nv / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; i--) {
            if ((float)v[i].dist < thDist) break;
            st->u_right[v[i].idx] = -1;
            st->depth[v[i].idx] = -1;
            nmatched--;
        }
        free(v);
    }
    rowtab_free(&rt);
    return nmatched;
}

/* Frame::ComputeStereoMatch_OnePoint, src/Frame.cc:1175-1260. has_mappoint as above. */
int oo_stereo_one_point(const oo_keypoint *kl, const uint32_t *dl, int nl,
                        const oo_keypoint *kr, const uint32_t *dr, int nr,
                        const float *scale_factors, int nrows, float mbf,
                        const uint8_t *has_mappoint, int iL, int thOrbDist, float minD, float maxD,
                        oo_stereo_state *st, int32_t *best_idx, int32_t *best_dist)
{
    if (iL < 0 || iL >= nl) return 0;
    if (!has_mappoint || !has_mappoint[iL] || st->stereo_matched[iL]) return 0;
    st->stereo_matched[iL] = 1;
    oo_rowtab rt;
    rowtab_build(&rt, kr, nr, scale_factors, nrows);
    const int r = stereo_one_left(kl, dl, iL, kr, dr, &rt, nrows, thOrbDist, minD, maxD, mbf, 0.0f, st, best_idx, best_dist);
    rowtab_free(&rt);
    return r == 2;
}

/* src/Frame.cc:984-1173 */
int oo_stereo_classic(const oo_extractor *exl, const oo_extractor *exr,
                      const oo_keypoint *kl, const uint32_t *dl, int nl,
                      const oo_keypoint *kr, const uint32_t *dr, int nr,
                      float mbf, float mb,
                      int32_t *best_idx, int32_t *best_dist, float *u_right, float *depth)
{
    const int B = OO_EDGE_THRESHOLD;
    const int thOrbDist = (100 + 50) / 2;
    const int nRows = exl->lh[0];
    int *cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    for (int pass = 0; pass < 1; pass++)
        for (int iR = 0; iR < nr; iR++) {
            const float kpY = kr[iR].y;
            const float r = 2.0f * exl->scale[kr[iR].octave];
            int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
            if (minr < 0) minr = 0;
            if (maxr > nRows - 1) maxr = nRows - 1;
            for (int yi = minr; yi <= maxr; yi++) cnt[yi + 1]++;
        }
    for (int y = 0; y < nRows; y++) cnt[y + 1] += cnt[y];
    int *fill = (int *)malloc(sizeof(int) * ((size_t)nRows + 1));
    memcpy(fill, cnt, sizeof(int) * ((size_t)nRows + 1));
    int *rowidx = (int *)malloc(sizeof(int) * (cnt[nRows] > 0 ? cnt[nRows] : 1));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * exl->scale[kr[iR].octave];
        int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        if (minr < 0) minr = 0;
        if (maxr > nRows - 1) maxr = nRows - 1;
        for (int yi = minr; yi <= maxr; yi++) rowidx[fill[yi]++] = iR;
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    distidx *v = (distidx *)malloc(sizeof(distidx) * (nl > 0 ? nl : 1));
    int nv = 0;
    for (int iL = 0; iL < nl; iL++) {
        best_idx[iL] = -1;
        best_dist[iL] = 100;
        u_right[iL] = -1.0f;
        depth[iL] = -1.0f;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        if (cnt[row + 1] == cnt[row]) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100, bestIdxR = 0;
        for (int k = cnt[row]; k < cnt[row + 1]; k++) {
            int iR = rowidx[k];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                int dist = oo_hamming(dl + 8 * (size_t)iL, dr + 8 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        best_idx[iL] = bestIdxR;
        best_dist[iL] = bestDist;
        if (bestDist < thOrbDist) {
            /* coordinates in the image pyramid at the left keypoint's scale (:1076-1081) */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = exl->inv_scale[levelL];
            const float scaleduL = roundf(kl[iL].x * scaleFactor);
            const float scaledvL = roundf(kl[iL].y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int pitchL = exl->lw[levelL] + 2 * B, pitchR = exr->lw[levelL] + 2 * B;
            const uint8_t *imL = exl->plane[levelL] + (size_t)B * pitchL + B;
            const uint8_t *imR = exr->plane[levelL] + (size_t)B * pitchR + B;
            const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= exr->lw[levelL]) continue;
            int bestDistS = 2147483647, bestincR = 0;
            float vDists[11];
            const float cL = (float)imL[(ptrdiff_t)cy * pitchL + cxL];
            for (int incR = -L; incR <= L; incR++) {
                const float cR = (float)imR[(ptrdiff_t)cy * pitchR + cxR0 + incR];
                float dist = 0;
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        float a = (float)imL[(ptrdiff_t)(cy + dy) * pitchL + cxL + dx] - cL;
                        float b = (float)imR[(ptrdiff_t)(cy + dy) * pitchR + cxR0 + incR + dx] - cR;
                        dist += fabsf(a - b);
                    }
                if (dist < (float)bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = exl->scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = (float)(uL - 0.01);
                }
                depth[iL] = mbf / disparity;
                u_right[iL] = bestuR;
                v[nv].dist = bestDistS; v[nv].idx = iL; nv++;
            }
        }
    }
    int nkept = nv;
    if (nv > 0) {
        qsort(v, nv, sizeof(distidx), distidx_cmp);
        const float median = (float)v[nv / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; i--) {
            if ((float)v[i].dist < thDist) break;
            u_right[v[i].idx] = -1;
            depth[v[i].idx] = -1;
            nkept--;
        }
    }
    free(v); free(cnt); free(fill); free(rowidx);
    return nkept;
}

/* src/Frame.cc:536-551, :669-734 */
#define OO_GRID_COLS 64
#define OO_GRID_ROWS 48
int oo_features_in_area(const oo_keypoint *kps, int n, float minX, float minY, float invW, float invH,
                        const float *qx, const float *qy, const float *qr, const int32_t *qmin,
                        const int32_t *qmax, int Q, int32_t *cand_off, int32_t *cand_idx, int cap)
{
    /* mGrid[col][row] as CSR, keypoints pushed in index order */
    int *start = (int *)calloc(OO_GRID_COLS * OO_GRID_ROWS + 1, sizeof(int));
    int *cellof = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int posX = (int)roundf((kps[i].x - minX) * invW);
        int posY = (int)roundf((kps[i].y - minY) * invH);
        cellof[i] = -1;
        if (posX < 0 || posX >= OO_GRID_COLS || posY < 0 || posY >= OO_GRID_ROWS) continue;
        cellof[i] = posX * OO_GRID_ROWS + posY;
        start[cellof[i] + 1]++;
    }
    for (int c = 0; c < OO_GRID_COLS * OO_GRID_ROWS; c++) start[c + 1] += start[c];
    int *fill = (int *)malloc(sizeof(int) * (OO_GRID_COLS * OO_GRID_ROWS + 1));
    memcpy(fill, start, sizeof(int) * (OO_GRID_COLS * OO_GRID_ROWS + 1));
    int *list = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        if (cellof[i] >= 0) list[fill[cellof[i]]++] = i;

    int total = 0;
    for (int q = 0; q < Q; q++) {
        cand_off[q] = total;
        const float x = qx[q], y = qy[q], r = qr[q];
        const int minLevel = qmin[q], maxLevel = qmax[q];
        const int nMinCellX = imax(0, (int)floorf((x - minX - r) * invW));
        if (nMinCellX >= OO_GRID_COLS) continue;
        const int nMaxCellX = imin(OO_GRID_COLS - 1, (int)ceilf((x - minX + r) * invW));
        if (nMaxCellX < 0) continue;
        const int nMinCellY = imax(0, (int)floorf((y - minY - r) * invH));
        if (nMinCellY >= OO_GRID_ROWS) continue;
        const int nMaxCellY = imin(OO_GRID_ROWS - 1, (int)ceilf((y - minY + r) * invH));
        if (nMaxCellY < 0) continue;
        const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c = ix * OO_GRID_ROWS + iy;
                for (int j = start[c]; j < start[c + 1]; j++) {
                    const oo_keypoint *kp = &kps[list[j]];
                    if (bCheckLevels) {
                        if (kp->octave < minLevel) continue;
                        if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                    }
                    const float distx = kp->x - x, disty = kp->y - y;
                    if (fabsf(distx) < r && fabsf(disty) < r) {
                        if (cand_idx && total < cap) cand_idx[total] = list[j];
                        total++;
                    }
                }
            }
    }
    cand_off[Q] = total;
    free(start); free(cellof); free(fill); free(list);
    return total;
}

/* ------------------------------------------------------------------------- *
 * Keypoint undistortion: Frame::UndistortKeyPoints / UndistortKeyPointsStereo,
 * src/Frame.cc:745-852.  The reference calls cv::undistortPoints (or, with
 * USE_FISHEYE_DISTORTION, cv::fisheye::undistortPoints) on the keypoint
 * coordinates and copies every other keypoint field.
 *
 * OpenCV 3.4.1 semantics (restated, parity unpinned like the rest of this file):
 *  - all arithmetic in double, inputs and outputs are float;
 *  - RR = P(:, 0:3) * R (identity where a matrix is absent);
 *  - cv::undistortPoints: x = (u - cx) / fx via ifx = 1/fx; 5 fixed-point iterations
 *    (default TermCriteria(COUNT, 5)) of
 *      icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2)
 *      dX = 2 k2 x y + k3 (r2 + 2 x x) + k8 r2 + k9 r2 r2
 *      dY = k2 (r2 + 2 y y) + 2 k3 x y + k10 r2 + k11 r2 r2
 *      x = (x0 - dX) icdist, y = (y0 - dY) icdist
 *    with k = (k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4); the tilt model
 *    (coefficients 13, 14) is not supported;
 *  - cv::fisheye::undistortPoints: pw = (p - c) / f, theta_d = |pw| clamped to pi/2,
 *    10 iterations theta = theta_d / (1 + k0 t^2 + k1 t^4 + k2 t^6 + k3 t^8),
 *    scale = tan(theta) / theta_d (1 when theta_d <= 1e-8);
 *  - projection x' = (RR0 . (x, y, 1)) / (RR2 . (x, y, 1)).
 * ------------------------------------------------------------------------- */
void oo_undistort_rr(const double *R, const double *P, int p_cols, double RR[9])
{
    double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) memcpy(r, R, sizeof(r));
    if (!P) { memcpy(RR, r, sizeof(r)); return; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[i * p_cols + k] * r[k * 3 + j];
            RR[i * 3 + j] = s;
        }
}

void oo_undistort_points(const float *xy_in, int n, const double K[9], const double *dist, int ndist,
                         const double *R, const double *P, int p_cols, int fisheye, float *xy_out)
{
    double k[12] = {0}, RR[9];
    for (int i = 0; i < ndist && i < 12; i++) k[i] = dist[i];
    oo_undistort_rr(R, P, p_cols, RR);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        double x, y;
        if (!fisheye) {
            x = ((double)xy_in[2 * i] - cx) * ifx;
            y = ((double)xy_in[2 * i + 1] - cy) * ify;
            if (ndist > 0) {
                const double x0 = x, y0 = y;
                for (int j = 0; j < 5; j++) {
                    const double r2 = x * x + y * y;
                    const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                    const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
                    const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
                    x = (x0 - deltaX) * icdist;
                    y = (y0 - deltaY) * icdist;
                }
            }
        } else {
            const double pwx = ((double)xy_in[2 * i] - cx) / fx, pwy = ((double)xy_in[2 * i + 1] - cy) / fy;
            double scale = 1.0;
            double theta_d = sqrt(pwx * pwx + pwy * pwy);
            const double half_pi = 3.1415926535897932384626433832795 / 2.;
            theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
            if (theta_d > 1e-8) {
                double theta = theta_d;
                for (int j = 0; j < 10; j++) {
                    const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
                    theta = theta_d / (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
                }
                scale = tan(theta) / theta_d;
            }
            x = pwx * scale;
            y = pwy * scale;
        }
        if (!fisheye) {
            const double xx = RR[0] * x + RR[1] * y + RR[2];
            const double yy = RR[3] * x + RR[4] * y + RR[5];
            const double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
            xy_out[2 * i] = (float)(xx * ww);
            xy_out[2 * i + 1] = (float)(yy * ww);
        } else {
            const double px = RR[0] * x + RR[1] * y + RR[2];
            const double py = RR[3] * x + RR[4] * y + RR[5];
            const double pz = RR[6] * x + RR[7] * y + RR[8];
            xy_out[2 * i] = (float)(px / pz);
            xy_out[2 * i + 1] = (float)(py / pz);
        }
    }
}

/* ------------------------------------------------------------------------- *
 * Search routines with sequential exclusion: the result of query k depends on
 * what queries 0..k-1 claimed.  Restated on plain arrays; the candidate list of
 * a query (Frame::GetFeaturesInArea) is a pure function of the frame, so all
 * lists are formed first and the loops below walk them exactly as written.
 * ------------------------------------------------------------------------- */
/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1723-1764 (histo[i].size() given as sizes[i]) */
void oo_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s; *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

#define OO_TH_LOW 50          /* src/ORBmatcher.cc:37 */
#define OO_TH_HIGH 100        /* :38 */
#define OO_HISTO_LENGTH 30    /* :39 */

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:520-633 */
int oo_search_for_initialization(const oo_keypoint *k1, const uint32_t *d1, int n1,
                                 const oo_keypoint *k2, const uint32_t *d2, int n2,
                                 float minX, float minY, float invW, float invH,
                                 float *prev_matched_xy, int windowSize, float nnratio, int check_orientation,
                                 int32_t *matches12)
{
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                      /* :523 */
    int *rotHist[OO_HISTO_LENGTH], rotN[OO_HISTO_LENGTH];                 /* :525-527 */
    for (int i = 0; i < OO_HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1)); rotN[i] = 0; }
    const float factor = 1.0f / OO_HISTO_LENGTH;                          /* :528 */
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) { vMatchedDistance[i] = INT32_MAX; vnMatches21[i] = -1; }   /* :530-531 */

    /* F2.GetFeaturesInArea(vbPrevMatched[i1].x, .y, windowSize, level1, level1) for every i1 (:540) */
    float *qx = (float *)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(float)), *qy = (float *)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(float));
    float *qr = (float *)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(float));
    int32_t *ql = (int32_t *)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(int32_t));
    int32_t *off = (int32_t *)calloc(((size_t)n1 + 1), sizeof(int32_t));
    for (int i = 0; i < n1; i++) {
        qx[i] = prev_matched_xy[2 * i]; qy[i] = prev_matched_xy[2 * i + 1];
        qr[i] = (float)windowSize; ql[i] = k1[i].octave;
    }
    const int total = oo_features_in_area(k2, n2, minX, minY, invW, invH, qx, qy, qr, ql, ql, n1, off, NULL, 0);
    int32_t *idx = (int32_t *)calloc((size_t)(total > 0 ? total : 1), sizeof(int32_t));
    oo_features_in_area(k2, n2, minX, minY, invW, invH, qx, qy, qr, ql, ql, n1, off, idx, total);

    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;                                          /* :537-538 */
        if (off[i1 + 1] == off[i1]) continue;                              /* :542-543 */
        int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
        for (int k = off[i1]; k < off[i1 + 1]; k++) {
            const int i2 = idx[k];
            const int dist = oo_hamming(d1 + 8 * (size_t)i1, d2 + 8 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;                    /* :559-560 */
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= OO_TH_LOW) {                                       /* :574 */
            if (bestDist < (float)bestDist2 * nnratio) {                   /* :576 */
                if (vnMatches21[bestIdx2] >= 0) {                          /* :578-582 */
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {                                   /* :588-598 */
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == OO_HISTO_LENGTH) bin = 0;
                    rotHist[bin][rotN[bin]++] = i1;
                }
            }
        }
    }
    if (check_orientation) {                                               /* :603-625 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(rotN, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                const int idx1 = rotHist[i][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                                        /* :628-630 */
        if (matches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = k2[matches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    for (int i = 0; i < OO_HISTO_LENGTH; i++) free(rotHist[i]);
    free(vMatchedDistance); free(vnMatches21); free(qx); free(qy); free(qr); free(ql); free(off); free(idx);
    return nmatches;
}

/* ORBmatcher::RadiusByViewingCos, src/ORBmatcher.cc:243-249 (float compared with the double 0.998) */
static float radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th),
 * src/ORBmatcher.cc:155-241 */
int oo_search_by_projection(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                            const float *scale_factors, float minX, float minY, float invW, float invH,
                            const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                            const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                            const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                            const int32_t *mp_observations, int M, float th, float nnratio,
                            int32_t *frame_obs, int32_t *frame_match, int32_t *match_score)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;                                         /* :159 */
    const size_t m1 = (size_t)(M > 0 ? M : 1);
    float *qx = (float *)calloc(m1, sizeof(float)), *qy = (float *)calloc(m1, sizeof(float)), *qr = (float *)calloc(m1, sizeof(float));
    float *rr = (float *)calloc(m1, sizeof(float));
    int32_t *qmin = (int32_t *)calloc(m1, sizeof(int32_t)), *qmax = (int32_t *)calloc(m1, sizeof(int32_t));
    int32_t *off = (int32_t *)calloc(((size_t)M + 1), sizeof(int32_t));
    for (int i = 0; i < M; i++) {
        float r = radius_by_viewing_cos(mp_view_cos[i]);                   /* :173 */
        if (bFactor) r *= th;                                              /* :175-176 */
        rr[i] = r;
        const int lvl = mp_level[i];
        /* :178-179; a map point the loop skips (:164-168) asks for nothing */
        const int skip = !mp_track_in_view[i] || mp_is_bad[i];
        qx[i] = mp_proj_x[i]; qy[i] = mp_proj_y[i];
        qr[i] = skip ? -1.0f : r * scale_factors[lvl];
        qmin[i] = lvl - 1; qmax[i] = lvl;
    }
    const int total = oo_features_in_area(kps, N, minX, minY, invW, invH, qx, qy, qr, qmin, qmax, M, off, NULL, 0);
    int32_t *idxs = (int32_t *)calloc((size_t)(total > 0 ? total : 1), sizeof(int32_t));
    oo_features_in_area(kps, N, minX, minY, invW, invH, qx, qy, qr, qmin, qmax, M, off, idxs, total);

    for (int iMP = 0; iMP < M; iMP++) {
        if (!mp_track_in_view[iMP]) continue;                              /* :164-165 */
        if (mp_is_bad[iMP]) continue;                                      /* :167-168 */
        const int nPredictedLevel = mp_level[iMP];
        const float r = rr[iMP];
        if (off[iMP + 1] == off[iMP]) continue;                            /* :181-182 */
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = off[iMP]; k < off[iMP + 1]; k++) {
            const int idx = idxs[k];
            if (frame_obs[idx] > 0) continue;                              /* :197-199: a map point with observations */
            if (u_right && u_right[idx] > 0) {                             /* :201-206 */
                const float er = fabsf(mp_proj_xr[iMP] - u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = oo_hamming(mp_desc + 8 * (size_t)iMP, desc + 8 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= OO_TH_HIGH) {                                      /* :228 */
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;   /* :230-231 */
            frame_match[bestIdx] = iMP;                                    /* :233 */
            frame_obs[bestIdx] = mp_observations[iMP];
            match_score[bestIdx] = bestDist;                               /* :235 */
            nmatches++;
        }
    }
    free(qx); free(qy); free(qr); free(rr); free(qmin); free(qmax); free(off); free(idxs);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono,
 * double &numVisibleMpt), src/ORBmatcher.cc:1440-1593, from the point where a map point of the last frame has been
 * projected into the current one (:1496 on; the projection itself, :1451-1493, is the caller's geometry).
 * lp_visible[i]: LastFrame.mvpMapPoints[i] exists, is no outlier, projects with invzc >= 0 inside the image bounds. */
int oo_search_by_projection_last(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                 const float *scale_factors, float minX, float minY, float invW, float invH,
                                 const uint8_t *lp_visible, const float *lp_u, const float *lp_v, const float *lp_invzc,
                                 const int32_t *lp_octave, const float *lp_angle, const uint32_t *lp_desc,
                                 const int32_t *lp_observations, int M, float th, float mbf, int bForward, int bBackward,
                                 int check_orientation, int32_t *frame_obs, int32_t *frame_match)
{
    int nmatches = 0;
    const size_t m1 = (size_t)(M > 0 ? M : 1);
    int *rotHist[OO_HISTO_LENGTH], rotN[OO_HISTO_LENGTH];
    for (int i = 0; i < OO_HISTO_LENGTH; i++) { rotHist[i] = (int *)calloc(m1, sizeof(int)); rotN[i] = 0; }
    const float factor = 1.0f / OO_HISTO_LENGTH;
    float *qx = (float *)calloc(m1, sizeof(float)), *qy = (float *)calloc(m1, sizeof(float)), *qr = (float *)calloc(m1, sizeof(float));
    int32_t *qmin = (int32_t *)calloc(m1, sizeof(int32_t)), *qmax = (int32_t *)calloc(m1, sizeof(int32_t));
    int32_t *off = (int32_t *)calloc((size_t)M + 1, sizeof(int32_t));
    for (int i = 0; i < M; i++) {
        const int nLastOctave = lp_octave[i];
        qx[i] = lp_u[i]; qy[i] = lp_v[i];
        qr[i] = lp_visible[i] ? th * scale_factors[nLastOctave] : -1.0f;        /* :1499 */
        if (bForward) { qmin[i] = nLastOctave; qmax[i] = -1; }                   /* :1503-1504 */
        else if (bBackward) { qmin[i] = 0; qmax[i] = nLastOctave; }             /* :1505-1506 */
        else { qmin[i] = nLastOctave - 1; qmax[i] = nLastOctave + 1; }          /* :1507-1508 */
    }
    const int total = oo_features_in_area(kps, N, minX, minY, invW, invH, qx, qy, qr, qmin, qmax, M, off, NULL, 0);
    int32_t *idxs = (int32_t *)calloc((size_t)(total > 0 ? total : 1), sizeof(int32_t));
    oo_features_in_area(kps, N, minX, minY, invW, invH, qx, qy, qr, qmin, qmax, M, off, idxs, total);
    for (int i = 0; i < M; i++) {
        if (!lp_visible[i]) continue;
        const float u = lp_u[i], invzc = lp_invzc[i];
        const float radius = qr[i];
        if (off[i + 1] == off[i]) continue;                                    /* :1510-1511 */
        int bestDist = 256, bestIdx2 = -1;
        for (int k = off[i]; k < off[i + 1]; k++) {
            const int i2 = idxs[k];
            if (frame_obs[i2] > 0) continue;                                   /* :1521-1523 */
            if (u_right && u_right[i2] > 0) {                                  /* :1525-1531 */
                const float ur = u - mbf * invzc;
                const float er = fabsf(ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = oo_hamming(lp_desc + 8 * (size_t)i, desc + 8 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= OO_TH_HIGH) {                                          /* :1544 */
            frame_match[bestIdx2] = i;
            frame_obs[bestIdx2] = lp_observations[i];
            nmatches++;
            if (check_orientation) {                                           /* :1554-1564 */
                float rot = lp_angle[i] - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == OO_HISTO_LENGTH) bin = 0;
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) {                                                   /* :1570-1590 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(rotN, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) {
                    frame_match[rotHist[i][j]] = -1;                           /* mvpMapPoints[...] = NULL */
                    frame_obs[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    for (int i = 0; i < OO_HISTO_LENGTH; i++) free(rotHist[i]);
    free(qx); free(qy); free(qr); free(qmin); free(qmax); free(off); free(idxs);
    return nmatches;
}

/* ------------------------------------------------------------------------- *
 * The good-feature forms of the projection search: one map point per call
 * (ORBmatcher::SearchByProjection_OnePoint, include/ORBmatcher.h:68-149;
 * GetCandidates :151-171 + MatchCandidates :173-252, which split the same body
 * in two) and the loop with a time budget (SearchByProjection_Budget,
 * src/ORBmatcher.cc:45-153).  Frame state as in oo_search_by_projection:
 * frame_obs[i] = Observations() of F.mvpMapPoints[i] or -1, frame_match[i] =
 * the map point written into F.mvpMapPoints[i], match_score = F.mvpMatchScore.
 * ------------------------------------------------------------------------- */
/* GetCandidates, include/ORBmatcher.h:151-171: returns the list length (0 when the point is skipped) */
int oo_get_candidates(const oo_keypoint *kps, int N, const float *scale_factors, float minX, float minY,
                      float invW, float invH, int track_in_view, int is_bad, float proj_x, float proj_y,
                      int level, float view_cos, float th, int32_t *cand_idx, int cap)
{
    const int bFactor = th != 1.0;                                         /* :153 */
    if (!track_in_view) return 0;                                          /* :155-156 */
    if (is_bad) return 0;                                                  /* :158-159 */
    float r = radius_by_viewing_cos(view_cos);                             /* :164 */
    if (bFactor) r *= th;                                                  /* :166-167 */
    const float qr = r * scale_factors[level];
    const int32_t qmin = level - 1, qmax = level;                          /* :169-170 */
    int32_t off[2];
    return oo_features_in_area(kps, N, minX, minY, invW, invH, &proj_x, &proj_y, &qr, &qmin, &qmax, 1, off, cand_idx, cap);
}

/* MatchCandidates, include/ORBmatcher.h:173-252, on the list GetCandidates stored in the map point */
int oo_match_candidates(const oo_keypoint *kps, const uint32_t *desc, const float *u_right,
                        const float *scale_factors, const int32_t *cand_idx, int ncand,
                        int track_in_view, int is_bad, float proj_xr, int level, float view_cos,
                        const uint32_t *mp_desc, int mp_observations, int mp_index, float th, float nnratio,
                        int32_t *frame_obs, int32_t *frame_match, int32_t *match_score)
{
    if (!track_in_view) return -1;                                         /* :176-177 */
    if (is_bad) return -1;                                                 /* :179-180 */
    if (ncand == 0) return -1;                                             /* :182-183 */
    float r = radius_by_viewing_cos(view_cos);                             /* :188 */
    const int bFactor = th != 1.0;
    if (bFactor) r *= th;                                                  /* :190-192 */
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < ncand; k++) {                                      /* :203-235 */
        const int idx = cand_idx[k];
        if (frame_obs[idx] > 0) continue;
        if (u_right && u_right[idx] > 0) {
            const float er = fabsf(proj_xr - u_right[idx]);
            if (er > r * scale_factors[level]) continue;
        }
        const int dist = oo_hamming(mp_desc, desc + 8 * (size_t)idx);
        if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx;
        } else if (dist < bestDist2) {
            bestLevel2 = kps[idx].octave; bestDist2 = dist;
        }
    }
    if (bestDist <= OO_TH_HIGH) {                                          /* :238-249 */
        if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) return -1;
        frame_match[bestIdx] = mp_index;
        frame_obs[bestIdx] = mp_observations;
        match_score[bestIdx] = bestDist;
        return bestIdx;
    }
    return -1;
}

/* SearchByProjection_OnePoint, include/ORBmatcher.h:68-149: the two halves above in one call */
int oo_search_by_projection_one_point(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                      const float *scale_factors, float minX, float minY, float invW, float invH,
                                      int track_in_view, int is_bad, float proj_x, float proj_y, float proj_xr,
                                      int level, float view_cos, const uint32_t *mp_desc, int mp_observations,
                                      int mp_index, float th, float nnratio,
                                      int32_t *frame_obs, int32_t *frame_match, int32_t *match_score)
{
    if (!track_in_view || is_bad) return -1;                               /* :72-76 */
    int32_t *list = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    const int n = oo_get_candidates(kps, N, scale_factors, minX, minY, invW, invH, 1, 0, proj_x, proj_y, level,
                                    view_cos, th, list, N);                /* :86-90 */
    const int best = oo_match_candidates(kps, desc, u_right, scale_factors, list, n, 1, 0, proj_xr, level, view_cos,
                                         mp_desc, mp_observations, mp_index, th, nnratio, frame_obs, frame_match,
                                         match_score);                     /* :92-148 */
    free(list);
    return best;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

/* SearchByProjection_Budget, src/ORBmatcher.cc:45-153: the loop of oo_search_by_projection with a wall clock read
 * after every map point that reaches the end of the loop body (:140-146; the `continue`s before it skip the clock).
 * The clock makes the result depend on the machine; it is deterministic at the two ends: a budget never reached
 * (the result of SearchByProjection) and a budget of 0 (the loop ends after the first map point that reaches :140).
 * *stopped_at = index of the map point after which the loop broke, M when it ran to the end. */
int oo_search_by_projection_budget(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                   const float *scale_factors, float minX, float minY, float invW, float invH,
                                   const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                                   const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                                   const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                                   const int32_t *mp_observations, int M, float th, float nnratio, double time_constr,
                                   int32_t *frame_obs, int32_t *frame_match, int32_t *match_score, int *stopped_at)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;                                         /* :51 */
    const double t0 = now_ms();                                            /* :54-55 */
    int32_t *list = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    *stopped_at = M;
    for (int iMP = 0; iMP < M; iMP++) {
        if (!mp_track_in_view[iMP]) continue;                              /* :60-61 */
        if (mp_is_bad[iMP]) continue;                                      /* :63-64 */
        const int nPredictedLevel = mp_level[iMP];
        float r = radius_by_viewing_cos(mp_view_cos[iMP]);                 /* :69 */
        if (bFactor) r *= th;                                              /* :71-72 */
        const float qr = r * scale_factors[nPredictedLevel];
        const int32_t qmin = nPredictedLevel - 1, qmax = nPredictedLevel;
        int32_t off[2];
        const int n = oo_features_in_area(kps, N, minX, minY, invW, invH, &mp_proj_x[iMP], &mp_proj_y[iMP], &qr, &qmin,
                                          &qmax, 1, off, list, N);         /* :74-75 */
        if (n == 0) continue;                                              /* :77-78 */
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < n; k++) {                                      /* :89-121 */
            const int idx = list[k];
            if (frame_obs[idx] > 0) continue;
            if (u_right && u_right[idx] > 0) {
                const float er = fabsf(mp_proj_xr[iMP] - u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = oo_hamming(mp_desc + 8 * (size_t)iMP, desc + 8 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= OO_TH_HIGH) {                                      /* :124-138 */
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            frame_match[bestIdx] = iMP;
            frame_obs[bestIdx] = mp_observations[iMP];
            match_score[bestIdx] = bestDist;
            nmatches++;
        }
        if (now_ms() - t0 >= time_constr) { *stopped_at = iMP; break; }    /* :141-146 */
    }
    free(list);
    return nmatches;
}

/* ------------------------------------------------------------------------- *
 * ORBmatcher::SearchByBoW.  A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) is passed as its
 * node ids in ascending order (the map's iteration order), offsets and feature indices.
 * ------------------------------------------------------------------------- */
static int fv_lower_bound(const int32_t *node, int n, int from, int32_t key)
{
    int i = from;                                      /* lower_bound over the whole map = over the part not yet passed */
    while (i < n && node[i] < key) i++;
    return i;
}

static int rot_bin(float a1, float a2)
{
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / OO_HISTO_LENGTH));
    if (bin == OO_HISTO_LENGTH) bin = 0;
    return bin;
}

/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:270-404.
 * kf_has_point[i]: vpMapPointsKF[i] exists and is not bad.  matches_f[n_f]: the keypoint of the keyframe whose map
 * point vpMapPointMatches[i] holds, -1 for NULL.  Returns nmatches. */
int oo_search_by_bow_frame(const oo_keypoint *kf_kps, const uint32_t *kf_desc, const uint8_t *kf_has_point,
                           const int32_t *kf_node, const int32_t *kf_off, const int32_t *kf_idx, int kf_nodes,
                           const oo_keypoint *f_kps, const uint32_t *f_desc, int n_f,
                           const int32_t *f_node, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                           float nnratio, int check_orientation, int32_t *matches_f)
{
    int nmatches = 0;
    for (int i = 0; i < n_f; i++) matches_f[i] = -1;                       /* :274 */
    int *hist = (int *)calloc((size_t)OO_HISTO_LENGTH * (size_t)(n_f > 0 ? n_f : 1), sizeof(int));
    int hn[OO_HISTO_LENGTH] = {0};
    int KFit = 0, Fit = 0;
    while (KFit != kf_nodes && Fit != f_nodes) {                           /* :291 */
        if (kf_node[KFit] == f_node[Fit]) {
            for (int iKF = kf_off[KFit]; iKF < kf_off[KFit + 1]; iKF++) {
                const int realIdxKF = kf_idx[iKF];
                if (!kf_has_point[realIdxKF]) continue;                    /* :304-308 */
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = f_off[Fit]; iF < f_off[Fit + 1]; iF++) {
                    const int realIdxF = f_idx[iF];
                    if (matches_f[realIdxF] >= 0) continue;                /* :320-321 */
                    const int dist = oo_hamming(kf_desc + 8 * (size_t)realIdxKF, f_desc + 8 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= OO_TH_LOW) {                              /* :339 */
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {   /* :341 */
                        matches_f[bestIdxF] = realIdxKF;
                        if (check_orientation) {                           /* :347-357 */
                            const int bin = rot_bin(kf_kps[realIdxKF].angle, f_kps[bestIdxF].angle);
                            hist[bin * n_f + hn[bin]++] = bestIdxF;
                        }
                        nmatches++;
                    }
                }
            }
            KFit++; Fit++;
        } else if (kf_node[KFit] < f_node[Fit]) {
            KFit = fv_lower_bound(kf_node, kf_nodes, KFit, f_node[Fit]);   /* :375 */
        } else {
            Fit = fv_lower_bound(f_node, f_nodes, Fit, kf_node[KFit]);     /* :379 */
        }
    }
    if (check_orientation) {                                               /* :383-401 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(hn, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { matches_f[hist[i * n_f + j]] = -1; nmatches--; }
        }
    }
    free(hist);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:635-768.
 * has_point1 / has_point2: the keypoint's map point exists and is not bad.  matches12[n1]: the keypoint of pKF2
 * whose map point vpMatches12[i] holds, -1 for NULL.  Returns nmatches. */
int oo_search_by_bow_keyframes(const oo_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1, int n1,
                               const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                               const oo_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2, int n2,
                               const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                               float nnratio, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                        /* :647 */
    uint8_t *vbMatched2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1); /* :648 */
    int *hist = (int *)calloc((size_t)OO_HISTO_LENGTH * (size_t)(n1 > 0 ? n1 : 1), sizeof(int));
    int hn[OO_HISTO_LENGTH] = {0};
    int f1it = 0, f2it = 0;
    while (f1it != nodes1 && f2it != nodes2) {                             /* :663 */
        if (node1[f1it] == node2[f2it]) {
            for (int i1 = off1[f1it]; i1 < off1[f1it + 1]; i1++) {
                const int i1x = idx1[i1];
                if (!has_point1[i1x]) continue;                            /* :671-675 */
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = off2[f2it]; i2 < off2[f2it + 1]; i2++) {
                    const int i2x = idx2[i2];
                    if (vbMatched2[i2x] || !has_point2[i2x]) continue;     /* :689-693 */
                    const int dist = oo_hamming(d1 + 8 * (size_t)i1x, d2 + 8 * (size_t)i2x);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2x; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < OO_TH_LOW) {                               /* :711: strictly below here */
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        matches12[i1x] = bestIdx2;                         /* :715 */
                        vbMatched2[bestIdx2] = 1;
                        if (check_orientation) {
                            const int bin = rot_bin(k1[i1x].angle, k2[bestIdx2].angle);
                            hist[bin * n1 + hn[bin]++] = i1x;
                        }
                        nmatches++;
                    }
                }
            }
            f1it++; f2it++;
        } else if (node1[f1it] < node2[f2it]) {
            f1it = fv_lower_bound(node1, nodes1, f1it, node2[f2it]);
        } else {
            f2it = fv_lower_bound(node2, nodes2, f2it, node1[f1it]);
        }
    }
    if (check_orientation) {                                               /* :747-765 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(hn, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { matches12[hist[i * n1 + j]] = -1; nmatches--; }
        }
    }
    free(vbMatched2); free(hist);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
 * const int ORBdist), src/ORBmatcher.cc:1595-1721, from the point where a map point of the keyframe has passed the
 * tests of :1615-1644 (kp_visible) and been projected (kp_u, kp_v) with its predicted level (kp_level).  frame_obs[i2]
 * >= 0 stands for CurrentFrame.mvpMapPoints[i2] != NULL. */
int oo_search_by_projection_keyframe(const oo_keypoint *kps, const uint32_t *desc, int N, const float *scale_factors,
                                     float minX, float minY, float invW, float invH,
                                     const uint8_t *kp_visible, const float *kp_u, const float *kp_v,
                                     const int32_t *kp_level, const float *kp_angle, const uint32_t *kp_desc,
                                     const int32_t *kp_observations, int M, float th, int ORBdist, int check_orientation,
                                     int32_t *frame_obs, int32_t *frame_match)
{
    int nmatches = 0;
    int *hist = (int *)calloc((size_t)OO_HISTO_LENGTH * (size_t)(N > 0 ? N : 1), sizeof(int));
    int hn[OO_HISTO_LENGTH] = {0};
    int32_t *list = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    for (int i = 0; i < M; i++) {
        if (!kp_visible[i]) continue;                                      /* :1615-1644 */
        const int nPredictedLevel = kp_level[i];
        const float radius = th * scale_factors[nPredictedLevel];          /* :1649 */
        const int32_t qmin = nPredictedLevel - 1, qmax = nPredictedLevel + 1;
        int32_t off[2];
        const int n = oo_features_in_area(kps, N, minX, minY, invW, invH, &kp_u[i], &kp_v[i], &radius, &qmin, &qmax, 1,
                                          off, list, N);                   /* :1651 */
        if (n == 0) continue;                                              /* :1653-1654 */
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < n; k++) {                                      /* :1661-1676 */
            const int i2 = list[k];
            if (frame_obs[i2] >= 0) continue;                              /* :1664-1665 */
            const int dist = oo_hamming(kp_desc + 8 * (size_t)i, desc + 8 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {                                         /* :1678 */
            frame_match[bestIdx2] = i;                                     /* :1680 */
            frame_obs[bestIdx2] = kp_observations[i];
            nmatches++;
            if (check_orientation) {                                       /* :1683-1693 */
                const int bin = rot_bin(kp_angle[i], kps[bestIdx2].angle);
                hist[bin * N + hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) {                                               /* :1699-1718 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(hn, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hn[i]; j++) {
                    frame_match[hist[i * N + j]] = -1;
                    frame_obs[hist[i * N + j]] = -1;
                    nmatches--;
                }
    }
    free(hist); free(list);
    return nmatches;
}

/* ------------------------------------------------------------------------- *
 * The searches of the mapping and loop-closing threads, each from the point where the reference has projected
 * its map points (visible[i] = every `continue` before the window search passed).
 * ------------------------------------------------------------------------- */
/* best keypoint of levels [level - 1, level] inside the window (the inner loops of :486-508, :1172-1191, :1311-1331):
 * `gate` adds the reprojection test of Fuse, :1026-1050.  Returns bestIdx, *best = bestDist. */
static int window_best(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                       const float *scale_factors, const float *inv_sigma2, float minX, float minY, float invW, float invH,
                       float u, float v, float ur, int level, const uint32_t *mp_desc, float th, int gate,
                       const int32_t *blocked, int32_t *list, int *best)
{
    const float radius = th * scale_factors[level];
    const int32_t qmin = -1, qmax = -1;                                    /* GetFeaturesInArea(u, v, radius): every level */
    int32_t off[2];
    const int n = oo_features_in_area(kps, N, minX, minY, invW, invH, &u, &v, &radius, &qmin, &qmax, 1, off, list, N);
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < n; k++) {
        const int idx = list[k];
        if (blocked && blocked[idx] >= 0) continue;                        /* :491-492 */
        const int kpLevel = kps[idx].octave;
        if (kpLevel < level - 1 || kpLevel > level) continue;              /* :496-497, :1023-1024, :1179, :1319 */
        if (gate) {
            if (u_right && u_right[idx] >= 0) {                            /* :1026-1039 */
                const float ex = u - kps[idx].x, ey = v - kps[idx].y, er = ur - u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
            } else {                                                       /* :1040-1050 */
                const float ex = u - kps[idx].x, ey = v - kps[idx].y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
            }
        }
        const int dist = oo_hamming(mp_desc, desc + 8 * (size_t)idx);
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    *best = bestDist;
    return bestIdx;
}

/* ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched,
 * int th), src/ORBmatcher.cc:406-518, from :473 on.  kf_matched[idx] >= 0 = vpMatched[idx] != NULL (in/out: an accepted
 * point stores 1), kf_match[idx] = the point this call stored there.  Returns nmatches. */
int oo_search_by_projection_sim3(const oo_keypoint *kps, const uint32_t *desc, int N, const float *scale_factors,
                                 float minX, float minY, float invW, float invH, const uint8_t *mp_visible,
                                 const float *mp_u, const float *mp_v, const int32_t *mp_level, const uint32_t *mp_desc,
                                 int M, int th, int32_t *kf_matched, int32_t *kf_match)
{
    int nmatches = 0;
    int32_t *list = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    for (int iMP = 0; iMP < M; iMP++) {
        if (!mp_visible[iMP]) continue;
        int bestDist;
        const int bestIdx = window_best(kps, desc, NULL, N, scale_factors, NULL, minX, minY, invW, invH, mp_u[iMP], mp_v[iMP],
                                        0.0f, mp_level[iMP], mp_desc + 8 * (size_t)iMP, (float)th, 0, kf_matched, list, &bestDist);
        if (bestDist <= OO_TH_LOW) {                                       /* :510-514 */
            kf_matched[bestIdx] = 1;
            kf_match[bestIdx] = iMP;
            nmatches++;
        }
    }
    free(list);
    return nmatches;
}

/* The search of ORBmatcher::Fuse: gate != 0 the form of :937-1087 (reprojection test; u_right = pKF->mvuRight or NULL for a
 * monocular keyframe, mp_ur = u - bf * invz), gate == 0 the Sim3 form of :1089-1212.  best_idx[i] = bestIdx if bestDist <=
 * TH_LOW (:1064, :1194) else -1, best_dist[i] = bestDist (256 when nothing was compared).  What happens to a fused point
 * (Replace / AddObservation, :1066-1081) is the caller's map bookkeeping.  Returns nFused. */
int oo_fuse_search(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N, const float *scale_factors,
                   const float *inv_sigma2, float minX, float minY, float invW, float invH, const uint8_t *mp_visible,
                   const float *mp_u, const float *mp_v, const float *mp_ur, const int32_t *mp_level,
                   const uint32_t *mp_desc, int M, float th, int gate, int32_t *best_idx, int32_t *best_dist)
{
    int nFused = 0;
    int32_t *list = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    for (int i = 0; i < M; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!mp_visible[i]) continue;
        int bestDist;
        const int bestIdx = window_best(kps, desc, u_right, N, scale_factors, inv_sigma2, minX, minY, invW, invH, mp_u[i], mp_v[i],
                                        mp_ur ? mp_ur[i] : 0.0f, mp_level[i], mp_desc + 8 * (size_t)i, th, gate, NULL, list,
                                        &bestDist);
        best_dist[i] = bestDist;
        if (bestDist <= OO_TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    free(list);
    return nFused;
}

/* ORBmatcher::SearchBySim3, src/ORBmatcher.cc:1214-1438, from the projected points on: p1_* = the map points of pKF1 seen
 * from pKF2 (visible = exists, not matched yet, not bad, :1264-1295 passed), searched among pKF2's keypoints (k2, d2), and
 * the other way round.  matches12[i1] = idx2 for the pairs that chose each other (:1419-1435), -1 elsewhere. */
int oo_search_by_sim3(const oo_keypoint *k1, const uint32_t *d1, int n1, const oo_keypoint *k2, const uint32_t *d2, int n2,
                      const float *scale_factors, float minX, float minY, float invW, float invH,
                      const uint8_t *p1_visible, const float *p1_u, const float *p1_v, const int32_t *p1_level,
                      const uint32_t *p1_desc, const uint8_t *p2_visible, const float *p2_u, const float *p2_v,
                      const int32_t *p2_level, const uint32_t *p2_desc, float th, int32_t *matches12)
{
    int32_t *vnMatch1 = (int32_t *)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(int32_t));
    int32_t *vnMatch2 = (int32_t *)calloc((size_t)(n2 > 0 ? n2 : 1), sizeof(int32_t));
    int32_t *list = (int32_t *)calloc((size_t)((n1 > n2 ? n1 : n2) + 1), sizeof(int32_t));
    for (int i1 = 0; i1 < n1; i1++) {
        vnMatch1[i1] = -1;
        if (!p1_visible[i1]) continue;
        int bestDist;
        const int bestIdx = window_best(k2, d2, NULL, n2, scale_factors, NULL, minX, minY, invW, invH, p1_u[i1], p1_v[i1], 0.0f,
                                        p1_level[i1], p1_desc + 8 * (size_t)i1, th, 0, NULL, list, &bestDist);
        if (bestDist <= OO_TH_HIGH) vnMatch1[i1] = bestIdx;                /* :1333-1336 */
    }
    for (int i2 = 0; i2 < n2; i2++) {
        vnMatch2[i2] = -1;
        if (!p2_visible[i2]) continue;
        int bestDist;
        const int bestIdx = window_best(k1, d1, NULL, n1, scale_factors, NULL, minX, minY, invW, invH, p2_u[i2], p2_v[i2], 0.0f,
                                        p2_level[i2], p2_desc + 8 * (size_t)i2, th, 0, NULL, list, &bestDist);
        if (bestDist <= OO_TH_HIGH) vnMatch2[i2] = bestIdx;                /* :1413-1416 */
    }
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {                                      /* :1419-1435 */
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    free(vnMatch1); free(vnMatch2); free(list);
    return nFound;
}

/* ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:251-268 (F12 row-major, float) */
static int check_dist_epipolar_line(const oo_keypoint *kp1, const oo_keypoint *kp2, const float *F12, const float *sigma2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2[kp2->octave];
}

/* ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:770-935.  has_point = the keypoint already holds a map point
 * (:815, :838), stereo = mvuRight >= 0 (:818, :841), (ex, ey) = the epipole in the second image (:776-783), feature vectors
 * as in oo_search_by_bow_frame.  The reference declares vbMatched2 and never sets it: a keypoint of the second keyframe can
 * be the partner of several of the first.  matches12[n1] = vMatches12.  Returns nmatches. */
int oo_search_for_triangulation(const oo_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1, const uint8_t *stereo1,
                                int n1, const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                const oo_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2, const uint8_t *stereo2,
                                const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                const float *scale_factors, const float *sigma2, const float *F12, float ex, float ey,
                                int bOnlyStereo, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                        /* :791 */
    int *hist = (int *)calloc((size_t)OO_HISTO_LENGTH * (size_t)(n1 > 0 ? n1 : 1), sizeof(int));
    int hn[OO_HISTO_LENGTH] = {0};
    int f1it = 0, f2it = 0;
    while (f1it != nodes1 && f2it != nodes2) {
        if (node1[f1it] == node2[f2it]) {
            for (int i1 = off1[f1it]; i1 < off1[f1it + 1]; i1++) {
                const int i1x = idx1[i1];
                if (has_point1[i1x]) continue;                             /* :812-816 */
                const int bStereo1 = stereo1[i1x];
                if (bOnlyStereo && !bStereo1) continue;                    /* :820-822 */
                int bestDist = OO_TH_LOW, bestIdx2 = -1;                   /* :828-829 */
                for (int i2 = off2[f2it]; i2 < off2[f2it + 1]; i2++) {
                    const int i2x = idx2[i2];
                    if (has_point2[i2x]) continue;                         /* :838 (vbMatched2 is never set) */
                    const int bStereo2 = stereo2[i2x];
                    if (bOnlyStereo && !bStereo2) continue;                /* :843-845 */
                    const int dist = oo_hamming(d1 + 8 * (size_t)i1x, d2 + 8 * (size_t)i2x);
                    if (dist > OO_TH_LOW || dist > bestDist) continue;     /* :851-852 */
                    if (!bStereo1 && !bStereo2) {                          /* :856-862 */
                        const float distex = ex - k2[i2x].x, distey = ey - k2[i2x].y;
                        if (distex * distex + distey * distey < 100 * scale_factors[k2[i2x].octave]) continue;
                    }
                    if (check_dist_epipolar_line(&k1[i1x], &k2[i2x], F12, sigma2)) { bestIdx2 = i2x; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {                                       /* :871-888 */
                    matches12[i1x] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        const int bin = rot_bin(k1[i1x].angle, k2[bestIdx2].angle);
                        hist[bin * n1 + hn[bin]++] = i1x;
                    }
                }
            }
            f1it++; f2it++;
        } else if (node1[f1it] < node2[f2it]) {
            f1it = fv_lower_bound(node1, nodes1, f1it, node2[f2it]);
        } else {
            f2it = fv_lower_bound(node2, nodes2, f2it, node1[f1it]);
        }
    }
    if (check_orientation) {                                               /* :904-922 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        oo_three_maxima(hn, OO_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < OO_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { matches12[hist[i * n1 + j]] = -1; nmatches--; }
        }
    }
    free(hist);
    return nmatches;
}
