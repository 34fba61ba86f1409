// orbx_kernels.hip -- gfx950 kernels of the ORB extraction path.
//
// Stage order of one batch (every kernel covers all frames of the batch):
//   k_pyr_level0       copyMakeBorder of the input into level 0        (src/ORBextractor.cc:1212-1213)
//   k_resize_slide     level l from level l-1 + REFLECT_101 border     (:1205-1208)
//                      (k_pyr_resize: same result, for scale factors whose 4-pixel source window exceeds 8 bytes)
//   k_fast_blur        per-cell FAST-9/16 + NMS + threshold fallback   (:1052-1101) and, as further
//                      workgroups of the same launch, the 7x7 Gaussian of every level (:1169-1170)
//                      (k_fast_cells_wave and k_blur_slide: the two on their own)
//   k_quadtree         DistributeOctTree                               (:488-711)
//                      (k_quadtree_wide / _mid: 1024- / 512-thread shapes for fewer frames; k_qt_bins before it when a
//                      level has very many candidates: such a level runs on node counts, see quadtree_counts_body)
//   k_orient_desc      IC_Angle + rBRIEF + output assembly             (:74-150, :1109-1124, :1173-1188)
//
// All arithmetic that decides a result is integer, or float written with
// explicit round-to-nearest intrinsics (no contraction), so that results are
// bit-identical to a scalar evaluation of the same expressions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <mutex>
#include <cstdlib>

#include "../../include/orbx.h"
#include "orbx_common.hpp"
#include "orbx_kernels.hpp"

namespace orbx {

// ---------------------------------------------------------------------------
// constant tables
// ---------------------------------------------------------------------------
__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16];

// the blur kernel has the 8.8 fixed-point taps of cv::getGaussianKernel(7, 2) compiled in
const int kGaussTaps[7] = {18, 34, 49, 55, 49, 34, 18};

int upload_constant_tables(const int* gauss7, const int* umax16, hipStream_t s)
{
    for (int i = 0; i < 7; i++)
        if (gauss7[i] != kGaussTaps[i]) return -1;      // the host derivation and the kernel must agree
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int), 0, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(s);
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------
// pyramid
// ---------------------------------------------------------------------------
// One thread = 16 plane bytes (pitch is a multiple of 64, planes are 64-byte aligned).  Pieces
// that lie inside the image come in as four dwords when the source is dword-addressable
// (aligned != 0), else (a width like 1241: every row starts at another byte of a dword) as the five
// aligned dwords around them, shifted into place -- a misaligned dword load is split into several
// cache accesses, and a byte gather is sixteen loads; pieces touching the left/right border are
// gathered byte by byte.  The index space lists all interior pieces first and the border pieces
// after them, so that (all but one) wavefronts run only one of the two paths.
// [lo, hi): the bytes of the whole batch, which an aligned load around a piece must not leave.
__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ in, int stride, size_t frame_stride,
                                                    uint8_t* __restrict__ pyr, FrameGeom g, int aligned,
                                                    const uint8_t* lo, const uint8_t* hi)
{
    const LevelGeom& L = g.lv[0];
    const int per_row = L.pitch >> 4;
    const int n_int = L.w >> 4, n_brd = per_row - n_int;
    const int n_interior = n_int * L.rows;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_row * L.rows) return;
    int row, piece;
    const bool interior = idx < n_interior;
    if (interior) {
        row = idx / n_int;
        piece = (kPadX >> 4) + idx - row * n_int;
    } else {
        const int b = idx - n_interior;
        row = b / n_brd;
        const int jb = b - row * n_brd;
        piece = jb < (kPadX >> 4) ? jb : jb + n_int;
    }
    const int sy = reflect101(row - kPadY, L.h);
    const uint8_t* src = in + (size_t)blockIdx.y * frame_stride + (size_t)sy * stride;
    const int sx0 = (piece << 4) - kPadX;
    uint32_t v[4];
    const uint8_t* p = src + sx0;
    const unsigned a = aligned ? 0u : (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(p - a);
    if (interior && a == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = s4[j];
    } else if (interior && reinterpret_cast<const uint8_t*>(s4) >= lo && reinterpret_cast<const uint8_t*>(s4 + 5) <= hi) {
        uint32_t w[5];
#pragma unroll
        for (int j = 0; j < 5; j++) w[j] = s4[j];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], a);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v[j] |= (uint32_t)src[reflect101(sx0 + 4 * j + k, L.w)] << (8 * k);
        }
    }
    uint8_t* dst = pyr + (size_t)blockIdx.y * g.pyr_frame_bytes + L.plane_off + (size_t)row * L.pitch + (piece << 4);
    *reinterpret_cast<uint4*>(dst) = make_uint4(v[0], v[1], v[2], v[3]);
}

// OpenCV 8-bit bilinear: horizontal pass in 1/2048 units without rounding,
// vertical pass ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.
__global__ __launch_bounds__(256) void k_pyr_resize(uint8_t* __restrict__ pyr, FrameGeom g, int level,
                                                    const XTab* __restrict__ xtab, const YTab* __restrict__ ytab)
{
    const LevelGeom& L = g.lv[level];
    const LevelGeom& P = g.lv[level - 1];
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int row = blockIdx.y;
    if (x4 >= L.pitch) return;
    uint8_t* frame = pyr + (size_t)blockIdx.z * g.pyr_frame_bytes;
    const int dy = reflect101(row - kPadY, L.h);
    const YTab yt = ytab[L.ytab_off + dy];
    const uint8_t* S0 = frame + P.plane_off + (size_t)(yt.y0 + kPadY) * P.pitch + kPadX;
    const uint8_t* S1 = frame + P.plane_off + (size_t)(yt.y1 + kPadY) * P.pitch + kPadX;
    const int b0 = yt.b0, b1 = yt.b1;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int dx = reflect101(x4 + k - kPadX, L.w);
        const XTab xt = xtab[L.xtab_off + dx];
        int h0 = S0[xt.xofs] * xt.a0 + S0[xt.xofs + 1] * xt.a1;   // a1 == 0 where xofs+1 leaves the image
        int h1 = S1[xt.xofs] * xt.a0 + S1[xt.xofs + 1] * xt.a1;
        int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v |= (uint32_t)(o & 0xff) << (8 * k);
    }
    uint8_t* dst = frame + L.plane_off + (size_t)row * L.pitch + x4;
    *reinterpret_cast<uint32_t*>(dst) = v;
}

// The same resize with a lane owning 4 adjacent destination pixels (one plane
// dword, border columns included) and walking down a band of plane rows.  Per
// source row one unaligned 8-byte load holds every pixel the 4 columns need;
// v_perm_b32 pulls each column's (left, right) pair out as two u16 and
// v_dot2_u32_u16 applies the two horizontal weights.  No LDS, no barriers.
struct __attribute__((packed)) Unaligned8 { uint32_t lo, hi; };
struct Aligned12 { uint32_t a, b, c; };
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hpair(const Unaligned8 s, uint32_t sel, uint32_t a01)
{
    const uint32_t pq = __builtin_amdgcn_perm(s.hi, s.lo, sel);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, pq), __builtin_bit_cast(ushort2_t, a01), 0u, false) >> 4;
}

// rows [r0, r1) of destination dword column dwc of level L of one frame.  src + yt[row].off0 / off1 = column 0 of the two
// source rows: the previous level's plane in the same pyramid block, or (RAW, level 1 only) the caller's image itself,
// read through a buffer descriptor that covers exactly the frame (the 12-byte windows of the last columns run a few
// bytes past their row, which at the end of the last row would be past the caller's buffer: such bytes read as zero
// and carry no weight).
// ANY (RAW only; src_w = width of the image): an image whose rows start at any byte.  The lane then loads its 8 bytes
// where they lie; a window that would run past its row starts 8 bytes before the row's end instead, with the byte
// selectors moved along: a dword that straddles the end of the buffer descriptor reads as zero as a whole, valid bytes
// included, and nothing may be read past the caller's last row.
template <bool RAW, bool ANY = false>
__device__ __forceinline__ void resize_band(uint8_t* frame, const uint8_t* src, uint32_t src_bytes, const LevelGeom& L,
                                            const XTab4* __restrict__ xtab, const YTab2* __restrict__ yt,
                                            int dwc, int r0, int r1, int mirror, int src_w = 0)
{
    XTab4 x = xtab[L.xtab4_off + dwc];
    if (ANY) {
        const uint32_t start = min(x.base, (uint32_t)(src_w - 8));
        const uint32_t sh = x.base - start;                          // 0 .. 7
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t ia = min((x.sel[j] & 7u) + sh, 7u), ib = min(((x.sel[j] >> 16) & 7u) + sh, 7u);   // (a byte past the row has no weight)
            x.sel[j] = ia | (0x0cu << 8) | (ib << 16) | (0x0cu << 24);
        }
        x.base = start;
    }
    // The 8 source bytes of this lane start at an arbitrary byte; a misaligned vector load is split
    // into many cache accesses, so fetch the 3 aligned dwords around them and shift (row offsets,
    // plane offsets and pitches are multiples of 4: the misalignment is the lane's constant).
    const uint32_t shx = ANY ? 0u : x.base & 3u;
    // addresses = wave-uniform row pointer + the lane's 32-bit offset: the form a global load / store
    // takes as scalar base + vector offset, with no 64-bit vector arithmetic per access
    uint8_t* dst = frame + L.plane_off;
    const uint32_t xoff = ANY ? x.base : x.base & ~3u, doff = 4u * (uint32_t)dwc;
    const int rlast = L.rows - 1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, RAW ? (int)src_bytes : 0, 0x00020000);
    auto load12 = [&](uint32_t row_off) -> Aligned12 {
        if (RAW && ANY) {
            typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
            const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)xoff, (int)row_off, 0);
            return Aligned12{v.x, v.y, 0u};
        }
        if (RAW) {
            typedef unsigned int v3u_t __attribute__((ext_vector_type(3)));
            const v3u_t v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)xoff, (int)row_off, 0);
            return Aligned12{v.x, v.y, v.z};
        }
        return *reinterpret_cast<const Aligned12*>(src + (size_t)(uint32_t)(row_off + xoff));
    };
    // Groups of 4 rows, all loads issued before the arithmetic; a group that runs past the plane
    // repeats its last row (same bytes written again).  Successive destination rows mostly share
    // a source row (the second one of row r is the first one of row r+1 five times out of six at
    // scale 1.2): its horizontal sums are kept instead of being loaded and computed again.  The
    // row numbers are wave-uniform, so these decisions are scalar branches.
    uint32_t hp[4] = {0, 0, 0, 0};                  // horizontal sums of the last second source row
    uint32_t prev_off = 0xffffffffu;                // its offset
    for (int rg = r0; rg < r1; rg += 4) {
        Aligned12 t0[4], t1[4];
        YTab2 y[4];
        int rr[4];
        bool fresh[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            rr[k] = min(rg + k, rlast);
            y[k] = yt[rr[k]];
            fresh[k] = y[k].off0 != (k == 0 ? prev_off : y[k - 1].off1);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (fresh[k]) t0[k] = load12(y[k].off0);
            t1[k] = load12(y[k].off1);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t h0[4], h1[4];
            Unaligned8 s1;
            s1.lo = __builtin_amdgcn_alignbyte(t1[k].b, t1[k].a, shx);
            s1.hi = __builtin_amdgcn_alignbyte(t1[k].c, t1[k].b, shx);
#pragma unroll
            for (int j = 0; j < 4; j++) h1[j] = hpair(s1, x.sel[j], x.a01[j]);
            if (fresh[k]) {
                Unaligned8 s0;
                s0.lo = __builtin_amdgcn_alignbyte(t0[k].b, t0[k].a, shx);
                s0.hi = __builtin_amdgcn_alignbyte(t0[k].c, t0[k].b, shx);
#pragma unroll
                for (int j = 0; j < 4; j++) h0[j] = hpair(s0, x.sel[j], x.a01[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) h0[j] = hp[j];
            }
            const uint32_t b0 = (uint32_t)y[k].b0, b1 = (uint32_t)y[k].b1;
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // 24-bit multiplies (weights <= 2048, sums < 2^15): full rate, a 32-bit v_mul_lo is not
                const uint32_t v = ((__umul24(b0, h0[j]) >> 16) + (__umul24(b1, h1[j]) >> 16) + 2u) >> 2;
                o |= (v & 0xffu) << (8 * j);
                hp[j] = h1[j];
            }
            *reinterpret_cast<uint32_t*>(dst + (size_t)((uint32_t)(rr[k] * (int)L.pitch) + doff)) = o;
            if (mirror) {
                // REFLECT_101 rows: image row d is also border row -d (d = 1 .. kPadY) and row 2 (h - 1) - d below the image
                const int d = rr[k] - kPadY;
                if (d >= 1 && d <= kPadY)
                    *reinterpret_cast<uint32_t*>(dst + (size_t)((uint32_t)((kPadY - d) * (int)L.pitch) + doff)) = o;
                if (d >= L.h - 1 - kPadY && d <= L.h - 2)
                    *reinterpret_cast<uint32_t*>(dst + (size_t)((uint32_t)((kPadY + 2 * (L.h - 1) - d) * (int)L.pitch) + doff)) = o;
            }
        }
        prev_off = y[3].off1;
    }
}

__global__ __launch_bounds__(256) void k_resize_slide(uint8_t* __restrict__ pyr, FrameGeom g, int level,
                                                      const XTab4* __restrict__ xtab, const YTab2* __restrict__ ytab,
                                                      int rows_per_band, int ncb, int mirror, RawL0 raw)
{
    const LevelGeom& L = g.lv[level];
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rb = item / ncb, cb = item - rb * ncb;
    const int dwc = cb * 64 + lane;
    // With `mirror` the bands cover the image rows only and every row near the top / bottom edge is stored a second time
    // as the border row it reflects into (the border rows are a fifth of a small level: not computed twice).
    const int row_lo = mirror ? kPadY : 0, row_hi = mirror ? kPadY + L.h : L.rows;
    // the band is the same for the whole wavefront: keep its row numbers in scalar registers so
    // that the row table comes in through scalar loads instead of one vector load per lane
    const int r0 = __builtin_amdgcn_readfirstlane(row_lo + rb * rows_per_band), r1 = min(r0 + rows_per_band, row_hi);
    if (dwc * 4 >= L.pitch || r0 >= row_hi) return;
    uint8_t* frame = pyr + (size_t)blockIdx.y * g.pyr_frame_bytes;
    if (raw.on && level == 1 && raw.any_alignment)
        resize_band<true, true>(frame, raw.base + (size_t)blockIdx.y * raw.frame_stride, raw.frame_bytes, L, xtab, raw.ytab2_l1, dwc, r0, r1, mirror, g.lv[0].w);
    else if (raw.on && level == 1)
        resize_band<true>(frame, raw.base + (size_t)blockIdx.y * raw.frame_stride, raw.frame_bytes, L, xtab, raw.ytab2_l1, dwc, r0, r1, mirror);
    else
        resize_band<false>(frame, frame + g.lv[level - 1].plane_off, 0u, L, xtab, ytab + L.ytab2_off, dwc, r0, r1, mirror);
}

// ---------------------------------------------------------------------------
// 7x7 Gaussian blur, sigma 2, 8.8 fixed-point taps, exact integer sums,
// dst = min(255, (acc + 32768) >> 16).  Reads the REFLECT_101 border that is
// stored in the plane.
// Sliding-window form: a lane owns 4 adjacent pixels (one
// dword) of a column block and walks down a band of rows, keeping the last 7
// horizontal sums of its pixels in registers.  No LDS, no barriers: one
// 12-byte load, 14 byte-alignment / dot4 operations for the 4 horizontal sums,
// 7 multiply-adds per pixel for the column, one dword store per row.
// Work item = (level, 64-dword column block, row band); 4 items per workgroup.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void blur_hsum4w(uint32_t Lw, uint32_t Cw, uint32_t Rw, uint32_t out[4])
{
    // (Lw, Cw, Rw) = 12 bytes: pixels x-4 .. x+7 of one row; pixel j's window = bytes j+1 .. j+7.  The three dwords stay as
    // they are and the taps are shifted instead: 10 four-term dot products, no byte alignment.
    constexpr uint32_t w0 = 18u, w1 = 34u, w2 = 49u, w3 = 55u, w4 = 49u, w5 = 34u, w6 = 18u;
    out[0] = __builtin_amdgcn_udot4(Cw, w3 | (w4 << 8) | (w5 << 16) | (w6 << 24),
                                    __builtin_amdgcn_udot4(Lw, (w0 << 8) | (w1 << 16) | (w2 << 24), 0u, false), false);
    out[1] = __builtin_amdgcn_udot4(Rw, w6,
                                    __builtin_amdgcn_udot4(Cw, w2 | (w3 << 8) | (w4 << 16) | (w5 << 24),
                                                           __builtin_amdgcn_udot4(Lw, (w0 << 16) | (w1 << 24), 0u, false), false), false);
    out[2] = __builtin_amdgcn_udot4(Rw, w5 | (w6 << 8),
                                    __builtin_amdgcn_udot4(Cw, w1 | (w2 << 8) | (w3 << 16) | (w4 << 24),
                                                           __builtin_amdgcn_udot4(Lw, w0 << 24, 0u, false), false), false);
    out[3] = __builtin_amdgcn_udot4(Rw, w4 | (w5 << 8) | (w6 << 16),
                                    __builtin_amdgcn_udot4(Cw, w0 | (w1 << 8) | (w2 << 16) | (w3 << 24), 0u, false), false);
}

// two signed 16-bit integers -> two bytes with unsigned saturation, in the low half of the result
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t pair)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(pair));
    return r;
}

__device__ __forceinline__ void blur_hsum4(const uint8_t* p, uint32_t out[4])
{
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    blur_hsum4w(q[0], q[1], q[2], out);
}

__device__ __forceinline__ uint32_t blur_vsum(uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3, uint32_t h4,
                                              uint32_t h5, uint32_t h6)
{
    uint32_t acc = 18u * (h0 + h6) + 34u * (h1 + h5) + 49u * (h2 + h4) + 55u * h3 + 32768u;
    acc >>= 16;
    return acc > 255u ? 255u : acc;
}

// The rows of one lane's band.  Source row nb + p (row 0 is image row y0 - 3; nb is the same for the six rows of a
// group, p a compile-time constant) of the lane's three dwords comes through `load3`; the destination is the blurred plane.
// Branch-free schedule: 6 warm-up rows, then groups of 6 rows whose loads are all issued
// before the arithmetic.  A band's last group may run past the band (at most 5 rows): those
// rows are read from the stored border (or reflected, when the source is the caller's image) and written with the
// same values the next band writes (or into the unused border of the blurred plane).
// Column pass: the horizontal sums of rows t-1 and t are kept as one packed pair
// P[t % 6] = h(t-1) | h(t) << 16, so that output row n-6 (taps = rows n-6 .. n) is three
// 2-term dot products over the pairs of times n-5, n-3, n-1 plus the tap of row n.
template <class Load3>
__device__ __forceinline__ void blur_band(Load3 load3, uint8_t* __restrict__ dst, uint32_t doff, uint32_t dpitch, int nrows)
{
    const ushort2_t W01 = {18, 34}, W23 = {49, 55}, W45 = {49, 34};
    uint32_t P[6][4], hprev[4];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        uint32_t hc[4], a, b, c;
        load3(0, p, a, b, c);
        blur_hsum4w(a, b, c, hc);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (p > 0) P[p][j] = hprev[j] | (hc[j] << 16);
            hprev[j] = hc[j];
        }
    }
    for (int r0 = 0; r0 < nrows; r0 += 6) {
        const uint32_t dofs = doff + __umul24((uint32_t)r0, dpitch);   // rows, pitch < 2^24
        uint32_t raw[6][3];
#pragma unroll
        for (int p = 0; p < 6; p++) load3(r0 + 6, p, raw[p][0], raw[p][1], raw[p][2]);
#pragma unroll
        for (int p = 0; p < 6; p++) {
            // row n = r0+6+p enters; r0 is a multiple of 6, so time n lives in slot p
            uint32_t hc[4], acc[4];
            blur_hsum4w(raw[p][0], raw[p][1], raw[p][2], hc);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t s = __umul24(18u, hc[j]) + 32768u;      // 24-bit multiply-add (full rate)
                s = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, P[(p + 5) % 6][j]), W45, s, false);
                s = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, P[(p + 3) % 6][j]), W23, s, false);
                s = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, P[(p + 1) % 6][j]), W01, s, false);
                acc[j] = s;
                P[p][j] = hprev[j] | (hc[j] << 16);
                hprev[j] = hc[j];
            }
            // s >> 16 is at most 257: the high halves of two sums as a pair of 16-bit integers, saturated to two bytes
            const uint32_t lo = sat_pk_u8(__builtin_amdgcn_perm(acc[1], acc[0], 0x07060302u));
            const uint32_t hi = sat_pk_u8(__builtin_amdgcn_perm(acc[3], acc[2], 0x07060302u));
            *reinterpret_cast<uint32_t*>(dst + (size_t)(dofs + __umul24((uint32_t)p, dpitch))) = lo | (hi << 16);
        }
    }
}

// one work item (level, 64-dword column block, row band) on one wavefront; item and frame are wave-uniform
__device__ __forceinline__ void blur_item(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                          const FrameGeom& g, const SlideMap& sm, int item, int lane, int frame,
                                          const RawL0& raw)
{
    if (item >= sm.base[kMaxLevels]) return;
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++)
        if (l < g.nlevels && item >= sm.base[l]) level = l;
    const LevelGeom& L = g.lv[level];
    // the level's (band, dword column) pairs are dealt to lanes in one flat sequence, so only the
    // last wavefront of a level has idle lanes
    const int flat = (item - sm.base[level]) * 64 + lane;
    const int ndw = sm.ndw[level];
    if (flat >= ndw * sm.nrb[level]) return;
    const int rb = ndw == 1 ? flat : (int)__umulhi((uint32_t)flat, sm.rcp_ndw[level]);   // band
    const int dw = flat - rb * ndw;                      // dword column inside the level image
    const int y0 = min(rb * sm.rows[level], L.h - sm.rows[level]), y1 = y0 + sm.rows[level];
    const size_t fbase = (size_t)frame * g.pyr_frame_bytes + L.plane_off;
    // addresses = wave-uniform plane pointer + the lane's 32-bit offset (scalar base + vector offset
    // in the load / store, one 32-bit add per row instead of 64-bit vector arithmetic)
    uint8_t* dst = blur + fbase;
    const uint32_t pitch = (uint32_t)L.pitch;
    const uint32_t doff = __umul24((uint32_t)(kPadY + y0), pitch) + (uint32_t)(kPadX + 4 * dw);
    const int nrows = y1 - y0;
    if (raw.on && level == 0) {
        // Level 0 from the caller's image (rows `stride` apart, no border around them): a buffer descriptor over the
        // frame makes the few reads outside it (the dword left of column 0 in row 0, right of the last column in the
        // last row) return zero instead of touching memory that is not the caller's.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(raw.base + (size_t)frame * raw.frame_stride), 0, (int)raw.frame_bytes, 0x00020000);
        typedef unsigned int v3u_t __attribute__((ext_vector_type(3)));
        const uint32_t spitch = (uint32_t)raw.stride;
        const int last_n = ((nrows + 5) / 6) * 6 + 5;                    // last source row counter a band loads
        const bool edge = dw == 0 || 4 * dw + 7 >= L.w || y0 < 3 || y0 - 3 + last_n > L.h - 1;
        if (__ballot(edge) == 0ull) {                                     // wave-uniform: no lane needs a pixel outside the image
            const uint32_t soff = __umul24((uint32_t)(y0 - 3), spitch) + (uint32_t)(4 * dw - 4);
            blur_band([&](int nb, int p, uint32_t& a, uint32_t& b, uint32_t& c) {
                const v3u_t v = __builtin_amdgcn_raw_buffer_load_b96(
                    rsrc, (int)(soff + __umul24((uint32_t)nb, spitch) + __umul24((uint32_t)p, spitch)), 0, 0);
                a = v.x; b = v.y; c = v.z;
            }, dst, doff, pitch, nrows);
        } else {
            // REFLECT_101 formed here: rows by index (row -r is row r, row h - 1 + r is row h - 1 - r), columns by byte
            // permutes.  A lane loads 12 bytes that lie inside its row (the first 12 for the leftmost lane, the last 12 for
            // the lanes whose taps reach past column w - 1, its own 12 otherwise) and cuts its three dwords (pixels
            // 4 dw - 4 .. 4 dw + 7) out of them: each is four bytes of one pair of adjacent loaded dwords, through a
            // selector computed here once per band.  Any width: the last dword column holds 1 to 4 pixels.
            const int X = 4 * dw, w = L.w;
            const bool left = dw == 0, rightish = X + 7 >= w;         // (a 12-byte load must not cross the end of the last row)
            const int h2 = 2 * L.h - 2;
            const int coff = left ? 0 : (rightish ? w - 12 : X - 4);
            uint32_t sel[3];
            bool upper[3];                                                  // the dword's bytes come from (v.y, v.z), not (v.x, v.y)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                int idx[4], lo = 12;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int col = X - 4 + 4 * q + j;
                    col = col < 0 ? -col : (col >= w ? 2 * (w - 1) - col : col);
                    idx[j] = min(max(col - coff, 0), 11);                   // (the unused bytes 0 and 11 may fall outside)
                    lo = min(lo, idx[j]);
                }
                upper[q] = lo >= 4;
                const int m4 = upper[q] ? 4 : 0;
                sel[q] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) sel[q] |= (uint32_t)min(idx[j] - m4, 7) << (8 * j);
            }
            blur_band([&](int nb, int p, uint32_t& a, uint32_t& b, uint32_t& c) {
                int r = y0 - 3 + nb + p;
                r = max(r, -r);
                r = min(r, h2 - r);
                const v3u_t v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(__umul24((uint32_t)r, spitch) + (uint32_t)coff), 0, 0);
                a = __builtin_amdgcn_perm(upper[0] ? v.z : v.y, upper[0] ? v.y : v.x, sel[0]);
                b = __builtin_amdgcn_perm(upper[1] ? v.z : v.y, upper[1] ? v.y : v.x, sel[1]);
                c = __builtin_amdgcn_perm(upper[2] ? v.z : v.y, upper[2] ? v.y : v.x, sel[2]);
            }, dst, doff, pitch, nrows);
        }
        return;
    }
    const uint8_t* src = pyr + fbase;
    const uint32_t soff = __umul24((uint32_t)(kPadY + y0 - 3), pitch) + (uint32_t)(kPadX + 4 * dw - 4);
    blur_band([&](int nb, int p, uint32_t& a, uint32_t& b, uint32_t& c) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + (size_t)(soff + __umul24((uint32_t)nb, pitch) + __umul24((uint32_t)p, pitch)));
        a = q[0]; b = q[1]; c = q[2];
    }, dst, doff, pitch, nrows);
}

__global__ __launch_bounds__(256) void k_blur_slide(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                    FrameGeom g, SlideMap sm, RawL0 raw)
{
    // wave-uniform: level, band and column block live in scalar registers
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    blur_item(pyr, blur, g, sm, item, threadIdx.x & 63, blockIdx.y, raw);
}

// ---------------------------------------------------------------------------
// FAST-9/16
// ---------------------------------------------------------------------------
// Threshold-free corner score: S = max over the 16 arcs of 9 contiguous ring
// pixels of max(min(v - p), min(p - v)) - 1.  "corner at threshold t" <=> S >= t,
// and S is the score OpenCV stores for a detected corner.  With
// maxarc(k) = max of the 9 ring pixels starting at k (and minarc likewise):
// S = max(v - min_k maxarc(k), max_k minarc(k) - v) - 1.
__device__ __forceinline__ int fast_score_lds(const uint8_t* c, int pitch)
{
    int p[16];
    p[0] = c[3 * pitch];      p[1] = c[3 * pitch + 1];  p[2] = c[2 * pitch + 2];   p[3] = c[pitch + 3];
    p[4] = c[3];              p[5] = c[-pitch + 3];     p[6] = c[-2 * pitch + 2];  p[7] = c[-3 * pitch + 1];
    p[8] = c[-3 * pitch];     p[9] = c[-3 * pitch - 1]; p[10] = c[-2 * pitch - 2]; p[11] = c[-pitch - 3];
    p[12] = c[-3];            p[13] = c[pitch - 3];     p[14] = c[2 * pitch - 2];  p[15] = c[3 * pitch - 1];
    const int v = c[0];
    int mx3[16], mn3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx3[k] = max(max(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
        mn3[k] = min(min(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
    }
    int mnmx = 255, mxmn = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mx9 = max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]);
        int mn9 = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
        mnmx = min(mnmx, mx9);
        mxmn = max(mxmn, mn9);
    }
    return max(v - mnmx, mxmn - v) - 1;
}

// The same score for two pixels at once: ring pixel k of pixel A in the low and of pixel B in the high
// 16 bits of one register, and the 3-input packed minimum / maximum of gfx950.  Those are half-float
// instructions; on the bit patterns 0x0000 .. 0x00ff (positive denormals, kept as they are: the kernels
// run with 16-bit denormals enabled) float order is integer order, and a minimum / maximum returns one of
// its inputs unchanged.  Returns max(S, 0) per half (a negative S is never a corner).
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2_t pk_max3_u8(us2_t a, us2_t b, us2_t c)
{
    us2_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ us2_t pk_min3_u8(us2_t a, us2_t b, us2_t c)
{
    us2_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ us2_t fast_score2_lds(const uint8_t* ca, const uint8_t* cb, int pitch)
{
    const int off[16] = {3 * pitch, 3 * pitch + 1, 2 * pitch + 2, pitch + 3, 3, -pitch + 3, -2 * pitch + 2, -3 * pitch + 1,
                         -3 * pitch, -3 * pitch - 1, -2 * pitch - 2, -pitch - 3, -3, pitch - 3, 2 * pitch - 2, 3 * pitch - 1};
    us2_t p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = us2_t{ca[off[k]], cb[off[k]]};
    const us2_t v = {ca[0], cb[0]};
    us2_t mx3[16], mn3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx3[k] = pk_max3_u8(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        mn3[k] = pk_min3_u8(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    us2_t mx9[16], mn9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx9[k] = pk_max3_u8(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
        mn9[k] = pk_min3_u8(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
    }
    // min over the 16 arc maxima, max over the 16 arc minima: 16 -> 6 -> 2 -> 1
    us2_t a[6], b[6];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a[k] = pk_min3_u8(mx9[3 * k], mx9[3 * k + 1], mx9[3 * k + 2]);
        b[k] = pk_max3_u8(mn9[3 * k], mn9[3 * k + 1], mn9[3 * k + 2]);
    }
    a[5] = mx9[15];
    b[5] = mn9[15];
    const us2_t mnmx = pk_min3_u8(pk_min3_u8(a[0], a[1], a[2]), pk_min3_u8(a[3], a[4], a[5]), a[5]);
    const us2_t mxmn = pk_max3_u8(pk_max3_u8(b[0], b[1], b[2]), pk_max3_u8(b[3], b[4], b[5]), b[5]);
    const us2_t one = {1, 1};
    const us2_t sc = __builtin_elementwise_max(__builtin_elementwise_sub_sat(v, mnmx), __builtin_elementwise_sub_sat(mxmn, v));
    return __builtin_elementwise_sub_sat(sc, one);
}

// inclusive scan over the 64 lanes with DPP moves (no LDS traffic): Kogge-Stone inside each row of
// 16 lanes (row_shr 1, 2, 4, 8 shifting zeros in), then the last lane of rows 0 and 2 is added to
// rows 1 and 3 (row_bcast15), then lane 31 to rows 2 and 3 (row_bcast31)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast31 into rows 2, 3
    return v;
}

// four pixels at a time for the FAST pretest: a dword of pixels split into its
// even and odd bytes as pairs of 16-bit lanes, so that packed 16-bit min / max / saturating subtract
// work on two pixels per instruction
__device__ __forceinline__ us2_t even_bytes(uint32_t x) { return __builtin_bit_cast(us2_t, x & 0x00ff00ffu); }
__device__ __forceinline__ us2_t odd_bytes(uint32_t x) { return __builtin_bit_cast(us2_t, __builtin_amdgcn_perm(0u, x, 0x0c030c01u)); }
// the compiler turns min(x, 1) into compare + select + repack; keep it one instruction (b: both halves in an SGPR)
__device__ __forceinline__ us2_t pk_min_u16(us2_t a, uint32_t b)
{
    us2_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// a * b + c on two 16-bit lanes (b: both halves in an SGPR); the compiler splits it into shift and add
__device__ __forceinline__ us2_t pk_mad_u16(us2_t a, uint32_t b, uint32_t c)
{
    us2_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// max(br, dk) of the compass test, clamped at 0, for two pixels
__device__ __forceinline__ us2_t compass2(us2_t v, us2_t p0, us2_t p8, us2_t p4, us2_t p12)
{
    const us2_t hi = __builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12));
    const us2_t lo = __builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12));
    return __builtin_elementwise_max(__builtin_elementwise_sub_sat(hi, v), __builtin_elementwise_sub_sat(v, lo));
}

constexpr int kFastTileFixed = 48, kFastScoreFixed = 40;   // pitches of the specialised wave-per-cell kernel

// ---------------------------------------------------------------------------
// Per-cell FAST, one wavefront per cell.  Most pixels fail a 4-point necessary
// test (an arc of 9 ring pixels always contains two adjacent compass points,
// so a corner needs two adjacent compass pixels beyond the threshold on the
// same side); the survivors are queued in pixel order and only they get the
// full score.  Everything after the tile load is wave-synchronous: ballots and
// lane prefix counts keep the queue, and therefore the output, in row-major
// order.  The cells' tested areas tile [3,width-3) x [3,height-3) of the detection region; a
// non-maximum test sees score 0 outside its own cell's area (each cell is a separate cv::FAST
// call in the reference).  Output: the cell's keypoints in row-major order, packed
// (y_local << 16 | x_local << 8 | score), and their count.
// ---------------------------------------------------------------------------
// Optional per-section cycle counts of the per-cell FAST code (diagnostic builds only:
// make EXTRA=-DORBX_FAST_PROFILE): summed over all cells into g_fast_prof, read with orbx_debug_fast_profile.
#ifdef ORBX_FAST_PROFILE
__device__ unsigned long long g_fast_prof[16];
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[6] = {0, 0, 0, 0, 0, 0}
#define PROF_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[k] += now_ - prof_t; prof_t = now_; } while (0)
#define PROF_COUNT(k, v) do { if (threadIdx.x == 0) atomicAdd(&g_fast_prof[k], (unsigned long long)(v)); } while (0)
#define PROF_FLUSH() do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_fast_prof[k_], prof_acc[k_]); atomicAdd(&g_fast_prof[6], 1ull); } } while (0)
#else
#define PROF_DECL
#define PROF_MARK(k)
#define PROF_COUNT(k, v)
#define PROF_FLUSH()
#endif
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// TPC / SPC: compile-time tile and score pitches (0 = take them from fl): with constants every
// ring pixel is an immediate offset from one LDS address.
template <int TPC, int SPC>
__device__ __forceinline__ void fast_one_cell(const uint8_t* __restrict__ pyr, const FrameGeom& g, const FastLds& fl,
                                              uint32_t* __restrict__ cell_cand, int* __restrict__ cell_count,
                                              const FastCell* __restrict__ cells, int cell, int f, unsigned char* smem,
                                              const RawL0& raw)
{
    // one wavefront = one workgroup: wave-level fences replace workgroup barriers
    uint32_t* s_tile = reinterpret_cast<uint32_t*>(smem);
    uint8_t* s_score = smem + fl.score_off;
    uint16_t* s_queue = reinterpret_cast<uint16_t*>(smem + fl.queue_off);
    const int TP = TPC ? TPC : fl.tile_pitch, SP = SPC ? SPC : fl.score_pitch;
    const int lane = threadIdx.x;
    // the cells of level 0 come first; with raw.on their tiles are cut out of the caller's image (a detection area
    // stays 16 pixels inside the image, so a tile with its 3-pixel ring never touches the border)
    const bool from_image = raw.on && cell < raw.ncells0;
    const FastCell fc = from_image ? raw.cells0[cell] : cells[cell];   // wave-uniform: one scalar load
    const int cw = fc.cw, ch = fc.ch, ox = fc.ox, ndw = fc.ndw, pitch = fc.pitch, cell_cap = fc.cell_cap;
    int* count_out = cell_count + (size_t)f * g.ncells + cell;
    if (cw == 0) {
        if (lane == 0) *count_out = 0;
        return;
    }
    const uint8_t* src = (from_image ? raw.base + (size_t)f * raw.frame_stride : pyr + (size_t)f * g.pyr_frame_bytes) + fc.src_off;
    const uint32_t src_left = (from_image ? raw.frame_bytes : g.pyr_frame_bytes) - fc.src_off;   // bytes from src to the end of the frame
    const int th = ch + 6;
    PROF_DECL;
    // Tile load: 16 lanes per tile row (32 when the pitch is not the compiled-in one), whole rows per
    // step, so that every address is the previous one plus a constant.  A denser packing of the
    // lanes needs fewer loads but a division per dword, and this kernel is bound by instruction issue.
    if (TPC && th <= 48) {
        // Compiled-in pitch: unrolled steps of 4 rows.  Buffer loads take the lane's constant 32-bit offset plus a
        // scalar row offset, the stores a constant LDS offset: no vector arithmetic per step, and every load is in
        // flight before the first store.  The buffer ends with the frame's pyramid, so that no address can fault.
        // The last step may write up to 3 rows past the tile's rows: unused tile rows or the head of the score
        // tile, which is cleared right after.
        const int rr = lane >> 4, cc = lane & 15;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(src), 0, (int)src_left, 0x00020000);
        if (cc < ndw) {
            const int so = __mul24(rr, pitch) + 4 * cc;
            uint32_t* d = s_tile + rr * (TPC >> 2) + cc;
            uint32_t v[12];
#pragma unroll
            for (int k = 0; k < 12; k++)
                if (4 * k < th) v[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, so, 4 * k * pitch, 0);
#pragma unroll
            for (int k = 0; k < 12; k++)
                if (4 * k < th) d[k * 4 * (TPC >> 2)] = v[k];
        }
    } else {
        constexpr int LPR = TPC ? 16 : 32, RPS = 64 / LPR;
        const int rr = lane / LPR, cc = lane & (LPR - 1);
        if (cc < ndw) {
            uint32_t so = (uint32_t)(__mul24(rr, pitch) + 4 * cc);
            uint32_t* d = s_tile + rr * (TP >> 2) + cc;
            const uint32_t sstep = (uint32_t)(RPS * pitch);
            for (int r = rr; r < th; r += RPS) {
                *d = *reinterpret_cast<const uint32_t*>(src + (size_t)so);
                so += sstep;
                d += RPS * (TP >> 2);
            }
        }
    }
    {
        // the score tile ends on a 16-byte boundary of its own (the queue starts at the next one)
        uint4* sc128 = reinterpret_cast<uint4*>(s_score);
        for (int i = lane; i < ((ch + 2) * SP + 15) >> 4; i += 64) sc128[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    WAVE_SYNC();
    PROF_MARK(0);

    const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(s_tile);
    const int npix = cw * ch;
    const uint32_t rcp_cw = fc.rcp_cw;                           // i / cw for i < 2^12, cw <= 64
    const int t = g.minTh, ts = max(g.iniTh, t);
    int nq = 0;
    // The queue holds qcap entries, enough for any natural image; when a round might not fit
    // (noise: most pixels pass), the queued pixels are scored at once and the queue restarts.
    // `flushed` then sends the maxima search over all pixels instead of over the queue.
    // Queue entry: pixel index (12 bits) | 0x8000 if the pretest value exceeds iniThFAST as well.
    const int qcap = fl.queue_cap;
    bool flushed = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    // Scores of the queue entries [lo, hi) into the score tile.  More than 64 entries go two per lane through
    // the packed form (entries q and q + 64 of a round of 128).  With `compact`, the entries whose score is a
    // corner score (>= minThFAST) are rewritten, in order, to the front of the queue and their number is
    // returned: the maxima search then visits only those (most pretest survivors score nothing).
    auto score_entries = [&](int lo, int hi, bool compact) -> int {
        int kept = 0;
        if (hi - lo > 64) {
            for (int q0 = lo; q0 < hi; q0 += 128) {
                const int qa = q0 + lane, qb = qa + 64;
                const bool va = qa < hi, vb = qb < hi;
                const int ia = s_queue[va ? qa : lo] & 0x0fff;
                const int ib = vb ? (s_queue[qb] & 0x0fff) : ia;
                const int pya = (int)(__umul24((uint32_t)ia, rcp_cw) >> 18), pxa = ia - __mul24(pya, cw);
                const int pyb = (int)(__umul24((uint32_t)ib, rcp_cw) >> 18), pxb = ib - __mul24(pyb, cw);
                const us2_t sc = fast_score2_lds(tile8 + (pya + 3) * TP + pxa + 3 + ox, tile8 + (pyb + 3) * TP + pxb + 3 + ox, TP);
                const int sa = sc.x, sb = sc.y;
                const bool ka = va && sa >= t && sa > 0, kb = vb && sb >= t && sb > 0;
                if (va) s_score[(pya + 1) * SP + pxa + 1] = (uint8_t)(ka ? sa : 0);
                if (vb) s_score[(pyb + 1) * SP + pxb + 1] = (uint8_t)(kb ? sb : 0);
                if (compact) {
                    const unsigned long long ba = __ballot(ka), bb = __ballot(kb);
                    if (ka) s_queue[kept + __popcll(ba & below)] = (uint16_t)ia;
                    kept += __popcll(ba);
                    if (kb) s_queue[kept + __popcll(bb & below)] = (uint16_t)ib;
                    kept += __popcll(bb);
                }
            }
            return kept;
        }
        for (int q0 = lo; q0 < hi; q0 += 64) {
            const int q = q0 + lane;
            const bool valid = q < hi;
            int i = 0;
            bool keep = false;
            if (valid) {
                i = s_queue[q] & 0x0fff;
                const int py = (int)(__umul24((uint32_t)i, rcp_cw) >> 18), px = i - __mul24(py, cw);
                const int s = fast_score_lds(tile8 + (py + 3) * TP + px + 3 + ox, TP);
                keep = s >= t && s > 0;
                s_score[(py + 1) * SP + px + 1] = (uint8_t)(keep ? s : 0);
            }
            if (compact) {
                const unsigned long long bal = __ballot(keep);
                if (keep) s_queue[kept + __popcll(bal & below)] = (uint16_t)i;
                kept += __popcll(bal);
            }
        }
        return kept;
    };
    auto score_queue = [&]() {
        WAVE_SYNC();
        score_entries(0, nq, false);
        WAVE_SYNC();
    };
    {
        // Pretest, four horizontally adjacent pixels per lane: lane = (row r of the round, group c of
        // 4 columns), lpr = ceil(cw / 4) lanes per row, R = 64 / lpr rows per round.  Five dword reads
        // (rows y-3 and y+3 at x, row y at x-3, x, x+3) feed packed 16-bit arithmetic; the survivors of
        // a round are counted per lane, placed by a wave scan and written in pixel order.
        const int lpr = fc.lpr, R = fc.fast_rows;
        const int r = (int)(__umul24((uint32_t)lane, fc.rcp_lpr) >> 16), c = lane - __mul24(r, lpr), c4 = 4 * c;
        // Misaligned LDS dwords are slow, so each of the five is cut out of an aligned pair: the
        // byte offsets (ox, ox + 3, ox + 6 from the lane's aligned column group) are wave-uniform.
        const int TPW = TP >> 2, TPW3 = 3 * TPW;
        const int jv = ox + 3, j4 = ox + 6;
        int w0 = __mul24(r, TPW) + c;                      // dword index of row y-3 at the lane's column group
        int row = r < R ? r : ch;                           // lanes past the last whole row never become valid
        const uint32_t i0 = (uint32_t)(__mul24(r, cw) + c4);
        uint32_t idxE = i0 | ((i0 + 2) << 16), idxO = (i0 + 1) | ((i0 + 3) << 16);
        const uint32_t stepi = (uint32_t)__mul24(R, cw), step2 = stepi | (stepi << 16);
        const int nv = cw - c4;                             // valid pixels of this lane's group: columns < cw
        const uint32_t colE = (nv > 0 ? 0x0000ffffu : 0u) | (nv > 2 ? 0xffff0000u : 0u);
        const uint32_t colO = (nv > 1 ? 0x0000ffffu : 0u) | (nv > 3 ? 0xffff0000u : 0u);
        const us2_t ts2 = {(unsigned short)(0x7fff - ts), (unsigned short)(0x7fff - ts)};      // ts <= 255
        const uint32_t t32 = (uint32_t)t | ((uint32_t)t << 16);
        // per-lane thresholds, 0xffff where the column is invalid; lanes past the last whole row of a round never take part
        const uint32_t thrE = r < R ? (t32 | ~colE) : 0xffffffffu, thrO = r < R ? (t32 | ~colO) : 0xffffffffu;
        const uint32_t one2 = 0x00010001u;
        const int step_w = __mul24(R, TPW);
        // one group of four pixels per lane: counts (1 per survivor) and queue entries of its even / odd pixels
        auto group = [&](int w, int rw, uint32_t iE, uint32_t iO, us2_t& cE, us2_t& cO, us2_t& eE, us2_t& eO) {
            // rows past the cell still lie inside the LDS block (the score tile follows): read, then discard
            const uint32_t* qa = s_tile + w;
            const uint32_t* qv = qa + (jv >> 2);
            const uint32_t* q4 = qa + (j4 >> 2);
            const uint32_t a8 = __builtin_amdgcn_alignbyte(qv[1], qv[0], jv & 3);
            const uint32_t av = __builtin_amdgcn_alignbyte(qv[TPW3 + 1], qv[TPW3], jv & 3);
            const uint32_t a0 = __builtin_amdgcn_alignbyte(qv[2 * TPW3 + 1], qv[2 * TPW3], jv & 3);
            const uint32_t a12 = __builtin_amdgcn_alignbyte(qa[TPW3 + 1], qa[TPW3], ox);
            const uint32_t a4 = __builtin_amdgcn_alignbyte(q4[TPW3 + 1], q4[TPW3], j4 & 3);
            const us2_t mE = compass2(even_bytes(av), even_bytes(a0), even_bytes(a8), even_bytes(a4), even_bytes(a12));
            const us2_t mO = compass2(odd_bytes(av), odd_bytes(a0), odd_bytes(a8), odd_bytes(a4), odd_bytes(a12));
            // invalid pixels (columns past the cell, rows past it) get the threshold 0xffff: their saturated difference is 0
            const bool row_ok = rw < ch;
            const uint32_t tEw = row_ok ? thrE : 0xffffffffu, tOw = row_ok ? thrO : 0xffffffffu;
            const us2_t tE = __builtin_bit_cast(us2_t, tEw), tO = __builtin_bit_cast(us2_t, tOw);
            cE = pk_min_u16(__builtin_elementwise_sub_sat(mE, tE), one2);          // 1 per survivor
            cO = pk_min_u16(__builtin_elementwise_sub_sat(mO, tO), one2);
            // entry = index | 0x8000 if the pretest value exceeds iniThFAST: m + (0x7fff - ts) carries into bit 15
            // exactly then (m <= 255), and the index stays below 0x1000
            eE = __builtin_bit_cast(us2_t, (__builtin_bit_cast(uint32_t, mE + ts2) & 0x80008000u) | iE);
            eO = __builtin_bit_cast(us2_t, (__builtin_bit_cast(uint32_t, mO + ts2) & 0x80008000u) | iO);
        };
        auto put = [&](int pos, us2_t cE, us2_t cO, us2_t eE, us2_t eO) {
            if (cE.x) s_queue[pos] = eE.x;
            pos += cE.x;
            if (cO.x) s_queue[pos] = eO.x;
            pos += cO.x;
            if (cE.y) s_queue[pos] = eE.y;
            pos += cE.y;
            if (cO.y) s_queue[pos] = eO.y;
        };
        // Two rounds of R rows per iteration (rows y0 .. y0 + R - 1, then the next R): their counts share one
        // wave scan, packed as two 16-bit fields; all entries of the first round precede those of the second.
        // The queue holds qcap >= 512 entries, enough for any natural image.  When a pair of rounds does not fit
        // (noise: most pixels pass), the loop is left, the queued pixels are scored at once, the queue restarts and
        // the loop resumes with the same pair (leaving the loop keeps the rare path out of its register budget).
        int y0 = 0;
        while (true) {
            bool full = false;
            for (; y0 < ch; y0 += 2 * R) {
                us2_t cEa, cOa, eEa, eOa, cEb, cOb, eEb, eOb;
                group(w0, row, idxE, idxO, cEa, cOa, eEa, eOa);
                group(w0 + step_w, row + R, idxE + step2, idxO + step2, cEb, cOb, eEb, eOb);
                const uint32_t ba = __builtin_bit_cast(uint32_t, cEa) + __builtin_bit_cast(uint32_t, cOa);
                const uint32_t bb = __builtin_bit_cast(uint32_t, cEb) + __builtin_bit_cast(uint32_t, cOb);
                const int cnta = (int)((ba & 0xffffu) + (ba >> 16)), cntb = (int)((bb & 0xffffu) + (bb >> 16));
                const int incl = wave_incl_scan(cnta | (cntb << 16));
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                const int tota = tot & 0xffff, totb = tot >> 16;
                if (nq + tota + totb > qcap) { full = true; break; }
                put(nq + (incl & 0xffff) - cnta, cEa, cOa, eEa, eOa);
                put(nq + tota + (incl >> 16) - cntb, cEb, cOb, eEb, eOb);
                nq += tota + totb;
                w0 += 2 * step_w;
                row += 2 * R;
                idxE += 2 * step2;
                idxO += 2 * step2;
            }
            if (!full) break;
            score_queue();
            nq = 0;
            flushed = true;
        }
    }
    uint32_t* out = cell_cand + (size_t)f * g.cand_per_frame + fc.cand_off;
    int base = 0;
    PROF_MARK(1);
    PROF_COUNT(8, nq);
    if (!flushed && nq > 64) {       // up to 64 survivors are one round of scores anyway
        // Only pixels whose pretest value exceeds iniThFAST (the score is below the pretest value) can be
        // corners of the reference's first cv::FAST call, and a neighbour that is not among them cannot
        // out-score one that is.  They are copied, in order, behind the queue, scored, and their strict
        // maxima at iniThFAST emitted: when there is at least one, that is the cell's result and the
        // other survivors (natural images: 2/3 of them) are never scored.
        int ns = 0;
        WAVE_SYNC();
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int q = q0 + lane;
            const int e = q < nq ? s_queue[q] : 0;
            const bool strong = (e & 0x8000) != 0;
            const unsigned long long bal = __ballot(strong);
            const int pos = nq + ns + __popcll(bal & below);
            if (strong && pos < qcap) s_queue[pos] = (uint16_t)e;
            ns += __popcll(bal);
        }
        if (ns > 0 && nq + ns <= qcap) {
            WAVE_SYNC();
            score_entries(nq, nq + ns, false);
            WAVE_SYNC();
            for (int q0 = nq; q0 < nq + ns; q0 += 64) {
                const int q = q0 + lane;
                bool hit = false;
                int py = 0, px = 0, s = 0;
                if (q < nq + ns) {
                    const int i = s_queue[q] & 0x0fff;
                    py = (int)(__umul24((uint32_t)i, rcp_cw) >> 18);
                    px = i - __mul24(py, cw);
                    const uint8_t* sc = s_score + (py + 1) * SP + px + 1;
                    s = sc[0];
                    const int m0 = max(max((int)sc[-SP - 1], (int)sc[-SP]), (int)sc[-SP + 1]);
                    const int m1 = max(max((int)sc[SP - 1], (int)sc[SP]), (int)sc[SP + 1]);
                    const int m2 = max(max((int)sc[-1], (int)sc[1]), 0);
                    hit = s > max(max(m0, m1), m2) && s >= g.iniTh;
                }
                const unsigned long long bal = __ballot(hit);
                if (hit) {
                    const int pos = base + __popcll(bal & below);
                    if (pos < cell_cap) out[pos] = ((uint32_t)py << 16) | ((uint32_t)px << 8) | (uint32_t)s;
                }
                base += __popcll(bal);
            }
            PROF_COUNT(9, ns);
            if (base > 0) {
                if (lane == 0) *count_out = min(base, cell_cap);
                PROF_MARK(2);
                PROF_COUNT(7, 1);
                PROF_FLUSH();
                return;
            }
        }
    }
    PROF_MARK(2);
    if (flushed) score_queue();
    else {
        // the entries that scored replace the queue; positions written are never ahead of positions still to be read
        WAVE_SYNC();
        nq = score_entries(0, nq, true);
        WAVE_SYNC();
    }
    PROF_MARK(3);
    // Strict 8-neighbour maxima.  Maxima that reach the initial threshold are emitted right
    // here, in queue (= row-major) order: if the cell has any, they are its whole result
    // (the first cv::FAST call of the reference found something).  Only a cell without one
    // needs the second sweep, which emits every flagged maximum (the minThFAST call).
    if (flushed) {
        // rare path: the queue no longer lists every candidate, so visit every pixel (row-major);
        // first for maxima at the initial threshold, then, if there is none, for all maxima
        for (int sweep = 0; sweep < 2 && base == 0; sweep++) {
            for (int i0 = 0; i0 < npix; i0 += 64) {
                const int i = i0 + lane;
                bool hit = false;
                int py = 0, px = 0, s = 0;
                if (i < npix) {
                    py = (int)(__umul24((uint32_t)i, rcp_cw) >> 18);
                    px = i - __mul24(py, cw);
                    const uint8_t* sc = s_score + (py + 1) * SP + px + 1;
                    s = sc[0];
                    const int m0 = max(max((int)sc[-SP - 1], (int)sc[-SP]), (int)sc[-SP + 1]);
                    const int m1 = max(max((int)sc[SP - 1], (int)sc[SP]), (int)sc[SP + 1]);
                    const int m2 = max(max((int)sc[-1], (int)sc[1]), 0);
                    hit = s > max(max(m0, m1), m2) && (sweep == 1 || s >= g.iniTh);
                }
                const unsigned long long bal = __ballot(hit);
                if (hit) {
                    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < cell_cap) out[pos] = ((uint32_t)py << 16) | ((uint32_t)px << 8) | (uint32_t)s;
                }
                base += __popcll(bal);
            }
        }
        if (lane == 0) *count_out = min(base, cell_cap);
        return;
    }
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int q = q0 + lane;
        bool hit = false;
        int py = 0, px = 0, s = 0;
        if (q < nq) {
            const int i = s_queue[q] & 0x0fff;
            py = (int)(__umul24((uint32_t)i, rcp_cw) >> 18);
            px = i - __mul24(py, cw);
            const uint8_t* sc = s_score + (py + 1) * SP + px + 1;
            s = sc[0];
            const int m0 = max(max((int)sc[-SP - 1], (int)sc[-SP]), (int)sc[-SP + 1]);
            const int m1 = max(max((int)sc[SP - 1], (int)sc[SP]), (int)sc[SP + 1]);
            const int m2 = max(max((int)sc[-1], (int)sc[1]), 0);
            if (s > max(max(m0, m1), m2)) {            // strict maximum of its 8 neighbours, and non-zero
                s_queue[q] = (uint16_t)(i | 0x4000);
                hit = s >= g.iniTh;
            }
        }
        const unsigned long long bal = __ballot(hit);
        if (hit) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < cell_cap) out[pos] = ((uint32_t)py << 16) | ((uint32_t)px << 8) | (uint32_t)s;
        }
        base += __popcll(bal);
    }
    if (base == 0) {
        WAVE_SYNC();
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int q = q0 + lane;
            bool keep = false;
            int py = 0, px = 0, s = 0;
            if (q < nq) {
                const int e = s_queue[q];
                if (e & 0x4000) {
                    const int i = e & 0x0fff;
                    py = (int)(__umul24((uint32_t)i, rcp_cw) >> 18);
                    px = i - __mul24(py, cw);
                    s = s_score[(py + 1) * SP + px + 1];
                    keep = true;
                }
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < cell_cap) out[pos] = ((uint32_t)py << 16) | ((uint32_t)px << 8) | (uint32_t)s;
            }
            base += __popcll(bal);
        }
    }
    if (lane == 0) *count_out = min(base, cell_cap);
    PROF_MARK(4);
    PROF_FLUSH();
}

#ifdef ORBX_FAST_PROFILE
}  // namespace orbx
extern "C" int orbx_debug_fast_profile(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(orbx::g_fast_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(orbx::g_fast_prof), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
namespace orbx {
#endif

// One wavefront = one workgroup = one cell.  (Several cells per wavefront to save workgroup dispatches
// measured slower: the dispatch of one cell overlaps the arithmetic of the others.)
// One wavefront = one workgroup = one cell.  (Several cells per wavefront, or several one-cell
// wavefronts per workgroup, measured slower: the kernel is bound by vector instruction issue, and
// the dispatch and tile load of one cell overlap the arithmetic of the others.)
template <int TPC = 0, int SPC = 0>
__global__ __launch_bounds__(64) void k_fast_cells_wave(const uint8_t* __restrict__ pyr, FrameGeom g, FastLds fl,
                                                        uint32_t* __restrict__ cell_cand, int* __restrict__ cell_count,
                                                        const FastCell* __restrict__ cells, int frame_per_xcd,
                                                        unsigned long long rcp_ncells, int nframes, RawL0 raw)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Neighbouring
    // tiles share their 3-pixel rings and cache lines, so a large batch gives every frame to one
    // XCD (1-D grid, block b -> XCD b % 8 -> frame 8 * (b / 8 / ncells) + b % 8): 2.8x less
    // fetch than cells scattered over all L2s.  Small batches use the plain (cell, frame) grid
    // so that one frame still spreads over the whole device.
    int cell, f;
    if (frame_per_xcd) {
        const uint32_t seq = blockIdx.x >> 3;
        const uint32_t fq = (uint32_t)(((unsigned long long)seq * rcp_ncells) >> 40);    // seq / ncells
        cell = (int)(seq - fq * (uint32_t)g.ncells);
        f = (int)(fq * 8 + (blockIdx.x & 7));
        if (f >= nframes) return;
    } else {
        cell = blockIdx.x;
        f = blockIdx.y;
    }
    fast_one_cell<TPC, SPC>(pyr, g, fl, cell_cand, cell_count, cells, cell, f, smem, raw);
}

// FAST and the blur in one launch: both only need the finished pyramid, so the blur's work items ride
// along as extra one-wavefront workgroups behind the cells.  A lone frame saves one launch of its
// chain of dependent launches; in a batch the two overlap because one is bound by vector issue and
// the other by memory (measured: the fused launch takes what the two took, and the descriptor kernel
// that follows finds more of the blurred planes in cache).
template <int TPC = 0, int SPC = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, FrameGeom g,
                                                  FastLds fl, SlideMap sm, uint32_t* __restrict__ cell_cand,
                                                  int* __restrict__ cell_count, const FastCell* __restrict__ cells,
                                                  int frame_per_xcd, unsigned long long rcp_nitems, int nitems, int nframes,
                                                  RawL0 raw)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // large batches: every frame on one XCD, as in k_fast_cells_wave (nitems = cells + blur items per frame)
    int b, f;
    if (frame_per_xcd) {
        const uint32_t seq = blockIdx.x >> 3;
        const uint32_t fq = (uint32_t)(((unsigned long long)seq * rcp_nitems) >> 40);    // seq / nitems
        b = (int)(seq - fq * (uint32_t)nitems);
        f = (int)(fq * 8 + (blockIdx.x & 7));
        if (f >= nframes) return;
    } else {
        b = blockIdx.x;
        f = blockIdx.y;
    }
    if (b < g.ncells) fast_one_cell<TPC, SPC>(pyr, g, fl, cell_cand, cell_count, cells, b, f, smem, raw);
    else blur_item(pyr, blur, g, sm, b - g.ncells, threadIdx.x, f, raw);
}

#undef WAVE_SYNC

// ---------------------------------------------------------------------------
// block-wide scans (NT threads = NT / 64 wavefronts per workgroup)
// ---------------------------------------------------------------------------
// exclusive scan of one value per thread; tmp = NT / 64 ints of LDS
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int* tmp, int* total)
{
    const int incl = wave_incl_scan(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        int t = tmp[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

// The same with one barrier: tmp must not be the array the workgroup's previous scan used (the callers alternate
// between two), so that nobody still reads what this call writes.
template <int NT>
__device__ __forceinline__ int block_excl_scan1(int v, int* tmp, int* total)
{
    const int incl = wave_incl_scan(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const int t = tmp[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}
// two values per thread, one barrier; tmp = 2 * NT / 64 ints
template <int NT>
__device__ __forceinline__ void block_excl_scan1_pair(int va, int vb, int* tmp, int* base_a, int* base_b, int* tot_a, int* tot_b)
{
    const int ia = wave_incl_scan(va), ib = wave_incl_scan(vb);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) { tmp[w] = ia; tmp[NT / 64 + w] = ib; }
    __syncthreads();
    int ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const int x = tmp[i], y = tmp[NT / 64 + i];
        if (i < w) { ba += x; bb += y; }
        ta += x;
        tb += y;
    }
    *tot_a = ta;
    *tot_b = tb;
    *base_a = ba + ia - va;
    *base_b = bb + ib - vb;
}

// in-place exclusive scan of an LDS array; returns the total to every thread
template <int NT>
__device__ __forceinline__ int lds_excl_scan(int* a, int n, int* tmp)
{
    const int chunk = (n + NT - 1) / NT;
    const int b = min((int)threadIdx.x * chunk, n), e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    int tot;
    int base = block_excl_scan<NT>(s, tmp, &tot);
    for (int i = b; i < e; i++) {
        int t = a[i];
        a[i] = base;
        base += t;
    }
    __syncthreads();
    return tot;
}

// two independent in-place exclusive scans sharing their barriers; tmp = 2 * NT / 64 ints
template <int NT>
__device__ __forceinline__ void lds_excl_scan2(int* a, int na, int* b, int nb, int* tmp, int* tot_a, int* tot_b)
{
    const int ca = (na + NT - 1) / NT, cb = (nb + NT - 1) / NT;
    const int a0 = min((int)threadIdx.x * ca, na), a1 = min(a0 + ca, na);
    const int b0 = min((int)threadIdx.x * cb, nb), b1 = min(b0 + cb, nb);
    int sa = 0, sb = 0;
    for (int i = a0; i < a1; i++) sa += a[i];
    for (int i = b0; i < b1; i++) sb += b[i];
    const int ia = wave_incl_scan(sa), ib = wave_incl_scan(sb);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 63) { tmp[w] = ia; tmp[NT / 64 + w] = ib; }
    __syncthreads();
    int base_a = 0, base_b = 0, ta = 0, tb = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const int x = tmp[i], y = tmp[NT / 64 + i];
        if (i < w) { base_a += x; base_b += y; }
        ta += x;
        tb += y;
    }
    *tot_a = ta;
    *tot_b = tb;
    base_a += ia - sa;
    base_b += ib - sb;
    for (int i = a0; i < a1; i++) { const int t = a[i]; a[i] = base_a; base_a += t; }
    for (int i = b0; i < b1; i++) { const int t = b[i]; b[i] = base_b; base_b += t; }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// DistributeOctTree (src/ORBextractor.cc:488-711) as array passes.
//
// The std::list of the reference is an array in list order.  One step splits a
// set of nodes; the new list is "children of the node processed last first,
// each as n4,n3,n2,n1, then the untouched nodes in their old order".  While
// |L| + 3*(expandable nodes) <= N every multi-key node is split (processing
// order = list order); afterwards the expandable nodes created by the previous
// step are split in (size, creation) descending order up to and including the
// first one that brings |L| to N -- found with a scan, because a split grows
// |L| by (non-empty children - 1) independently of the other splits.  Among
// nodes of equal size the one created later goes first (the reference orders
// equal sizes by heap address, :633).
// One workgroup per (level, frame).
// ---------------------------------------------------------------------------
// Number of keys among key[0..m) larger than ki, the share of lane `sub` of a group of G lanes: four keys per LDS read
// (key is 16-byte aligned); the keys are distinct, so the ranks are a permutation.
__device__ __forceinline__ int rank_packed(const uint32_t* key, int m, uint32_t ki, int sub, int G)
{
    int r = 0;
    const int m4 = m >> 2;
    const uint4* k4 = reinterpret_cast<const uint4*>(key);
#pragma unroll 2
    for (int j = sub; j < m4; j += G) {
        const uint4 k = k4[j];
        r += (k.x > ki) + (k.y > ki) + (k.z > ki) + (k.w > ki);
    }
    if (sub == 0)
        for (int j = m4 << 2; j < m; j++) r += key[j] > ki;
    return r;
}

struct QNode { short x0, x1, y0, y1; };
// ctr[idx] += 1 from every calling lane.  Keys are listed cell by cell, so the lanes of a wavefront mostly sit in the
// same node and quadrant, and LDS atomics on one address run one lane at a time: the lanes that share the first
// lane's counter add once, together (up to three rounds), the rest add on their own.
__device__ __forceinline__ void wave_count(int* ctr, int idx)
{
    const int lane = (int)(threadIdx.x & 63);
    bool todo = true;
    for (int it = 0; it < 3; it++) {
        const unsigned long long act = __builtin_amdgcn_ballot_w64(todo);
        if (!act) return;
        const int leader = (int)__ffsll((long long)act) - 1;
        const int lidx = __builtin_amdgcn_readlane(idx, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(todo && idx == lidx);
        if (lane == leader) atomicAdd(&ctr[lidx], (int)__popcll(same));
        todo = todo && idx != lidx;
    }
    if (todo) atomicAdd(&ctr[idx], 1);
}
// bytes of the node arrays in LDS for M nodes (the cell-offset array follows them)
__host__ __device__ inline size_t quadtree_fixed_lds(int M_)
{
    const size_t M = (size_t)M_;
    return (2 * M * 8 + 4 * M * 4 + 2 * M * 4 + 3 * M * 4 + 96 * 4 + 2 * M * 2 + 4 * M * 2 + 3 * M * 2 + M + 64 + 15) & ~(size_t)15;
}
constexpr int kKeyRegsWide = 3;   // register keys per thread of the 1024-thread shape (3072 keys per level)
constexpr int kKeyRegs = 10;   // keys per thread held in registers (10 * 256 = 2560 keys per level); 91 VGPRs = 5 wavefronts per SIMD. 16 keys cost 210 VGPRs (2 per SIMD) and 40 % of the time

__device__ __forceinline__ int ceil_half(int d)
{
    // ceil(static_cast<float>(d) / 2) for integer d (src/ORBextractor.cc:433-434); node extents are
    // non-negative and far below 2^24, where the float expression equals (d + 1) >> 1 exactly
    return (d + 1) >> 1;
}

// ---------------------------------------------------------------------------
// DistributeOctTree on node counts alone, for levels with very many candidates (large frames).
//
// The rectangles of the tree do not depend on the data: a node is its root and the sequence of quadrants below it,
// and DivideNode (:431-476) sends a key to the quadrant its coordinates select.  So a key's whole path is known in
// advance, and the number of keys of any node is a sum over the leaves below it.  k_qt_bins (all frames and blocks of
// leaves in parallel) counts every leaf's keys at depth D, keeps the leaf's best key (largest response, first in the
// reference's append order on ties) and adds up the depths inside a block; quadtree_counts_body (the prologue of the
// quadtree kernels for such a level) builds the sums and maxima of the few depths above the blocks and then runs the
// same list steps as quadtree_body, reading a node's child occupancy from the sums instead of passing over the keys
// (80 000 keys x 10 steps on one compute unit otherwise).  A node is (depth, path), path = root * 4^depth +
// quadrant digits.  If a step has to split a node at depth D, the workgroup goes on with the passes over the keys.
// ---------------------------------------------------------------------------
// entries of the depths below d (every depth padded to a multiple of 4 entries, so that a node's four children are one aligned load)
__host__ __device__ inline int qt_depth_off(int nIni, int d)
{
    int o = 0;
    for (int k = 0; k < d; k++) o += ((nIni << (2 * k)) + 3) & ~3;
    return o;
}
// bytes of one (frame, level) slice: counts (int) of every depth, then best keys (64 bit) of every depth
__host__ __device__ inline size_t qt_counts_bytes(int nIni, int D) { return ((size_t)qt_depth_off(nIni, D + 1) * 12 + 255) & ~(size_t)255; }
__host__ __device__ inline size_t qt_counts_best_off(int nIni, int D) { return ((size_t)qt_depth_off(nIni, D + 1) * 4 + 7) & ~(size_t)7; }

// depth of the nodes one workgroup of k_qt_bins fills with everything below them: 5 depths (32 x 32 leaves) per workgroup
constexpr int kQtBlockDepths = 5;
__host__ __device__ inline int qt_block_depth(int D) { return D > kQtBlockDepths ? D - kQtBlockDepths : 0; }

// One workgroup per node at depth Db = qt_block_depth(D) ("block"): it visits the FAST cells that can hold keys of the
// block (the block's rectangle widened by a cell: a key's root comes from a float division, :515-519, so membership is
// decided by the key's path, not by the rectangle), counts the block's keys per leaf and keeps every leaf's best key
// with LDS atomics, adds the depths between the block and its leaves, and writes all of them with plain stores: no
// global atomics and nothing to clear beforehand.  The depths above the blocks are summed by the level's workgroup.
#ifdef ORBX_QT_TIMING
// diagnostic build only: wall-clock marks (100 MHz) of every (frame, level) workgroup of the quadtree kernels
__device__ unsigned long long g_qt_timing[64 * 8 * 8];
#define QT_MARK(sec) do { if (threadIdx.x == 0) g_qt_timing[((blockIdx.x & 63) * 8 + (blockIdx.y & 7)) * 8 + (sec)] = wall_clock64(); } while (0)
#define QT_NOTE(sec, v) do { if (threadIdx.x == 0) g_qt_timing[((blockIdx.x & 63) * 8 + (blockIdx.y & 7)) * 8 + (sec)] = (unsigned long long)(v); } while (0)
__device__ unsigned long long g_qt_step[64 * 8 * 8];     // time per stretch of a step, summed over the steps of the workgroup
#define QT_STEP_DECL unsigned long long qt_last_ = 0, qt_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int qt_steps_ = 0; const unsigned long long qt_c0_ = clock64(), qt_w0_ = wall_clock64(); if (threadIdx.x == 0) qt_last_ = wall_clock64()
#define QT_STEP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); qt_acc_[k] += now_ - qt_last_; qt_last_ = now_; } } while (0)
#define QT_STEP_END() do { qt_steps_++; } while (0)
#define QT_STEP_FLUSH() do { qt_acc_[6] = clock64() - qt_c0_; qt_acc_[7] = wall_clock64() - qt_w0_; if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) g_qt_step[((blockIdx.x & 63) * 8 + (blockIdx.y & 7)) * 8 + k_] = qt_acc_[k_]; \
                                                     g_qt_timing[((blockIdx.x & 63) * 8 + (blockIdx.y & 7)) * 8 + 7] = (unsigned long long)qt_steps_; } } while (0)
extern "C" int orbx_debug_qt_timing(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_timing), sizeof(g_qt_timing)); }
extern "C" int orbx_debug_qt_steps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_step), sizeof(g_qt_step)); }
__device__ unsigned long long g_qt_bins_t[2048 * 8];
#define QTB_MARK(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z < 8) g_qt_bins_t[((blockIdx.z * 128 + blockIdx.x) & 2047) * 8 + (k)] = wall_clock64(); } while (0)
extern "C" int orbx_debug_qt_bins(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_bins_t), sizeof(g_qt_bins_t)); }
#else
#define QTB_MARK(k)
#define QT_MARK(sec)
#define QT_NOTE(sec, v)
#define QT_STEP_DECL
#define QT_STEP(k)
#define QT_STEP_END()
#define QT_STEP_FLUSH()
#endif

// Candidates per (frame, level) of the levels that may run on node counts: the sum both the blocks of k_qt_bins and the
// level's quadtree workgroup decide on (once here instead of once per block)
__global__ __launch_bounds__(1024) void k_qt_totals(FrameGeom g, CountLevels cl, const int* __restrict__ cell_count,
                                                    int* __restrict__ level_ncand)
{
    __shared__ int s_red[16];
    const int level = cl.level[blockIdx.x], f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const int* counts = cell_count + (size_t)f * g.ncells + L.cell_base;
    const int ncells = L.nCols * L.nRows;
    int part = 0;
    for (int c0 = tid; c0 < ncells; c0 += 1024 * 4) {
        int v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = c0 + 1024 * q < ncells ? counts[c0 + 1024 * q] : 0;
#pragma unroll
        for (int q = 0; q < 4; q++) part += v[q];
    }
    part = wave_incl_scan(part);
    if ((tid & 63) == 63) s_red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        int total = 0;
        for (int w = 0; w < 16; w++) total += s_red[w];
        level_ncand[f * kMaxLevels + level] = total;
    }
}

constexpr int kQtBinsThreads = 256;
__global__ __launch_bounds__(kQtBinsThreads) void k_qt_bins(FrameGeom g, CountLevels cl, const uint32_t* __restrict__ cell_cand,
                                                 const int* __restrict__ cell_count, const int* __restrict__ level_ncand,
                                                 unsigned char* __restrict__ scratch)
{
    constexpr int kNodes = 1 + 4 + 16 + 64 + 256 + 1024;      // nodes of a block over its kQtBlockDepths + 1 depths
    constexpr int kTab = 1024;                                // pixels per axis the leaf tables below can cover
    __shared__ int s_H[kNodes];
    __shared__ unsigned long long s_Bk[kNodes];
    __shared__ unsigned short s_tx[kTab], s_ty[kTab];
    const int level = cl.level[blockIdx.y], f = blockIdx.z, tid = threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const int nIni = L.nIni, D = cl.depth[blockIdx.y], Db = qt_block_depth(D), LB = D - Db;
    const int pb = blockIdx.x;                          // path of the block: root * 4^Db + quadrant digits
    if (pb >= (nIni << (2 * Db))) return;
    QTB_MARK(0);
    const int* counts = cell_count + (size_t)f * g.ncells + L.cell_base;
    // the level runs on node counts only with min_ratio x nfeat candidates (the test its quadtree workgroup makes too)
    if (level_ncand[f * kMaxLevels + level] < cl.min_ratio * L.nfeat) return;
    const float hX = L.hX;
    // the block's rectangle
    const int root = pb >> (2 * Db);
    int bx0 = (short)(int)__fmul_rn(hX, (float)root), bx1 = (short)(int)__fmul_rn(hX, (float)(root + 1)), by0 = 0, by1 = (short)L.height;
    int bxp = 0, byp = 0;                               // the block's quadrant digits, x and y bits apart
    for (int d = 0; d < Db; d++) {
        const int dig = (pb >> (2 * (Db - 1 - d))) & 3;
        const int hx = ceil_half(bx1 - bx0), hy = ceil_half(by1 - by0);
        if (dig & 1) bx0 = (short)(bx0 + hx); else bx1 = (short)(bx0 + hx);
        if (dig & 2) by0 = (short)(by0 + hy); else by1 = (short)(by0 + hy);
        bxp = 2 * bxp + (dig & 1);
        byp = 2 * byp + (dig >> 1);
    }
    for (int i = tid; i < kNodes; i += kQtBinsThreads) { s_H[i] = 0; s_Bk[i] = 0ull; }
    // cells whose keys (x in [cj * wCell + 3, cj * wCell + 3 + wCell), likewise y) can lie in the rectangle widened by a pixel
    const int cj0 = max(0, (bx0 - 4) / L.wCell), cj1 = min(L.nCols - 1, max(bx1 - 3, 0) / L.wCell);
    const int ci0 = max(0, (by0 - 4) / L.hCell), ci1 = min(L.nRows - 1, max(by1 - 3, 0) / L.hCell);
    const int ncj = cj1 - cj0 + 1, nci = ci1 - ci0 + 1;
    const int leaf_base = ((1 << (2 * LB)) - 1) / 3;     // first leaf entry of the block's arrays
    // A key's quadrant digits are an x bit and a y bit per depth, and each axis is halved on its own: the x bits follow
    // from x alone (root included), the y bits from y alone.  So the descent is made once per pixel column and pixel row
    // the visited cells cover instead of once per key (seven dependent halvings): s_tx / s_ty hold the column's / row's
    // bits inside the block, spread to the even / odd bit positions of the leaf number, or 0xffff when the column / row
    // leaves the block.  Keys outside the tables (blocks wider than kTab pixels) make the descent themselves.
    const int xlo = cj0 * L.wCell + 3, ylo = ci0 * L.hCell + 3;
    const int nx = (ncj > 0 && nci > 0) ? min(kTab, ncj * L.wCell) : 0, ny = (ncj > 0 && nci > 0) ? min(kTab, nci * L.hCell) : 0;
    auto spread = [](int v) { return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2) | ((v & 8) << 3) | ((v & 16) << 4); };
    for (int i = tid; i < nx + ny; i += kQtBinsThreads) {
        const bool isx = i < nx;
        const int p = isx ? xlo + i : ylo + (i - nx);
        int r = 0, p0 = 0, p1 = (short)L.height;
        if (isx) {
            r = (int)__fdiv_rn((float)p, hX);                      // :515-519
            r = min(max(r, 0), nIni - 1);
            p0 = (short)(int)__fmul_rn(hX, (float)r);
            p1 = (short)(int)__fmul_rn(hX, (float)(r + 1));
        }
        int bits = 0;
        for (int d = 0; d < D; d++) {
            const int h = ceil_half(p1 - p0);
            const int far = p < p0 + h ? 0 : 1;
            if (far) p0 = (short)(p0 + h); else p1 = (short)(p0 + h);
            bits = 2 * bits + far;
        }
        const bool inside = isx ? (r == root && (bits >> LB) == bxp) : (bits >> LB) == byp;
        const int sp = spread(bits & ((1 << LB) - 1));
        if (isx) s_tx[i] = (unsigned short)(inside ? sp : 0xffff);
        else s_ty[i - nx] = (unsigned short)(inside ? 2 * sp : 0xffff);
    }
    __syncthreads();
    QTB_MARK(1);
    const int sub = tid & 7;
    const uint32_t* cands = cell_cand + (size_t)f * g.cand_per_frame + L.cand_base;
    auto add_key = [&](uint32_t u, int xc, int yc, uint32_t slot) {
        const int x = xc + (int)((u >> 8) & 0xff), y = yc + (int)(u >> 16);
        if ((unsigned)(x - xlo) < (unsigned)nx && (unsigned)(y - ylo) < (unsigned)ny) {
            const uint32_t t = (uint32_t)s_tx[x - xlo] | (uint32_t)s_ty[y - ylo];
            if (t & 0x8000u) return;
            atomicAdd(&s_H[leaf_base + (int)t], 1);
            atomicMax(&s_Bk[leaf_base + (int)t], ((unsigned long long)(u & 0xff) << 32) | (unsigned long long)(0xFFFFFFFFu - slot));
            return;
        }
        int r = (int)__fdiv_rn((float)x, hX);                      // :515-519
        r = min(max(r, 0), nIni - 1);
        int x0 = (short)(int)__fmul_rn(hX, (float)r), x1 = (short)(int)__fmul_rn(hX, (float)(r + 1)), y0 = 0, y1 = (short)L.height;
        int path = r;
        for (int d = 0; d < D; d++) {
            const int hx = ceil_half(x1 - x0), hy = ceil_half(y1 - y0);
            const int right = x < x0 + hx ? 0 : 1, low = y < y0 + hy ? 0 : 1;
            if (right) x0 = (short)(x0 + hx); else x1 = (short)(x0 + hx);
            if (low) y0 = (short)(y0 + hy); else y1 = (short)(y0 + hy);
            path = path * 4 + right + 2 * low;
        }
        if ((path >> (2 * LB)) != pb) return;
        const int leaf = leaf_base + (path & ((1 << (2 * LB)) - 1));
        atomicAdd(&s_H[leaf], 1);
        atomicMax(&s_Bk[leaf], ((unsigned long long)(u & 0xff) << 32) | (unsigned long long)(0xFFFFFFFFu - slot));
    };
    // 8 lanes per cell, four cells of a lane group in flight: their counts are loaded together, then their first two keys
    // per lane (a cell rarely holds more than 16), so a workgroup waits for memory twice per 512 cells
    const int ncr = (ncj > 0 && nci > 0) ? ncj * nci : 0;
    for (int k0 = tid >> 3; k0 < ncr; k0 += kQtBinsThreads / 2) {
        int cnt[4], xc[4], yc[4];
        uint32_t slot0[4], u0[4], u1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = k0 + (kQtBinsThreads / 8) * q;
            const int kk = k < ncr ? k : k0;
            const int ci = ci0 + kk / ncj, cj = cj0 + kk % ncj;
            const int c = __mul24(ci, L.nCols) + cj;
            slot0[q] = (uint32_t)__mul24(c, L.cell_cap);
            xc[q] = __mul24(cj, L.wCell) + 3;
            yc[q] = __mul24(ci, L.hCell) + 3;
            cnt[q] = k < ncr ? counts[c] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u0[q] = sub < cnt[q] ? cands[slot0[q] + sub] : 0u;
            u1[q] = sub + 8 < cnt[q] ? cands[slot0[q] + sub + 8] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (sub < cnt[q]) add_key(u0[q], xc[q], yc[q], slot0[q] + (uint32_t)sub);
            if (sub + 8 < cnt[q]) add_key(u1[q], xc[q], yc[q], slot0[q] + (uint32_t)sub + 8u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            for (int j = sub + 16; j < cnt[q]; j += 8) add_key(cands[slot0[q] + j], xc[q], yc[q], slot0[q] + (uint32_t)j);
    }
    __syncthreads();
    QTB_MARK(2);
    // the depths between the leaves and the block
    for (int k = LB - 1; k >= 0; k--) {
        const int o = ((1 << (2 * k)) - 1) / 3, oc = ((1 << (2 * (k + 1))) - 1) / 3, nk = 1 << (2 * k);
        for (int i = tid; i < nk; i += kQtBinsThreads) {
            s_H[o + i] = s_H[oc + 4 * i] + s_H[oc + 4 * i + 1] + s_H[oc + 4 * i + 2] + s_H[oc + 4 * i + 3];
            s_Bk[o + i] = max(max(s_Bk[oc + 4 * i], s_Bk[oc + 4 * i + 1]), max(s_Bk[oc + 4 * i + 2], s_Bk[oc + 4 * i + 3]));
        }
        __syncthreads();
    }
    QTB_MARK(3);
    unsigned char* slice = scratch + ((size_t)f * cl.n + blockIdx.y) * cl.stride;
    int* H = reinterpret_cast<int*>(slice);
    unsigned long long* B = reinterpret_cast<unsigned long long*>(slice + qt_counts_best_off(nIni, D));
    for (int k = 0; k <= LB; k++) {
        const int o = ((1 << (2 * k)) - 1) / 3, nk = 1 << (2 * k);
        const int go = qt_depth_off(nIni, Db + k) + pb * nk;
        for (int i = tid; i < nk; i += kQtBinsThreads) {
            H[go + i] = s_H[o + i];
            B[go + i] = s_Bk[o + i];
        }
    }
    QTB_MARK(4);
}

// LDS of quadtree_counts_body for M nodes (less than the node arrays of the key passes, whose allocation it uses)
__host__ __device__ inline size_t qt_counts_lds(int M_)
{
    const size_t M = (size_t)M_;
    return 4 * M * 4 + 2 * M * 4 + 3 * M * 4 + 96 * 4 + 2 * M * 4 + 2 * M * 2 + 2 * M * 2 + 2 * M + M + 64;
}

// Returns true when the level is done; false (nothing written) when a node at the leaves' depth would have to be
// split, and the caller then runs the passes over the keys.  Uniform over the workgroup.  smem: qt_counts_lds(M) bytes.

template <int NT>
__device__ __forceinline__ bool quadtree_counts_body(const FrameGeom& g, const CountLevels& cl, int slot, int level, int f, int total,
                                                     const uint32_t* __restrict__ cell_cand, uint2* __restrict__ kp_stage,
                                                     int* __restrict__ level_count, int* __restrict__ level_ncand,
                                                     unsigned char* __restrict__ scratch, unsigned char* smem)
{
    const int tid = threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const int M = g.max_nodes, N = L.nfeat, nIni = L.nIni, D = cl.depth[slot];
    int* s_cc = reinterpret_cast<int*>(smem);                          // 4*M child occupancy
    int* s_cnt = s_cc + 4 * M;                                         // 2*M keys per node (two lists)
    int* s_A = s_cnt + 2 * M;                                          // M
    int* s_B = s_A + M;                                                // M
    int* s_C = s_B + M;                                                // M
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_C);                // M (only while ranking)
    int* s_var = s_C + M;                                              // 96
    int* s_path = s_var + 96;                                          // 2*M
    short* s_cidx = reinterpret_cast<short*>(s_path + 2 * M);          // 2*M
    unsigned short* s_order = reinterpret_cast<unsigned short*>(s_cidx + 2 * M);   // M
    unsigned short* s_cand = s_order + M;                              // M
    unsigned char* s_depth = reinterpret_cast<unsigned char*>(s_cand + M);         // 2*M
    unsigned char* s_fl = s_depth + 2 * M;                             // M
    int* s_tmp = s_var + 16;
    // s_var[4]=size [5]=phase2 [6]=finish [7]=cntlt [8]=a node at depth D would have to be split

    unsigned char* slice = scratch + ((size_t)f * cl.n + slot) * cl.stride;
    int* H = reinterpret_cast<int*>(slice);
    unsigned long long* B = reinterpret_cast<unsigned long long*>(slice + qt_counts_best_off(nIni, D));
    // sums and maxima of the depths above the blocks k_qt_bins filled
    for (int d = qt_block_depth(D) - 1; d >= 0; d--) {
        const int o = qt_depth_off(nIni, d), oc = qt_depth_off(nIni, d + 1), nd_ = nIni << (2 * d);
        for (int p = tid; p < nd_; p += NT) {
            const int4 h = *reinterpret_cast<const int4*>(H + oc + 4 * p);
            H[o + p] = h.x + h.y + h.z + h.w;
            const unsigned long long* b = B + oc + 4 * p;
            B[o + p] = max(max(b[0], b[1]), max(b[2], b[3]));
        }
        __syncthreads();
    }
    QT_MARK(2);
    // ---- root nodes (:492-534)
    for (int r = tid; r < nIni; r += NT) { s_A[r] = H[r]; s_B[r] = H[r] > 0; }
    if (tid == 0) {
        int n = 0;
        for (int r = 0; r < nIni; r++) n += H[r];
        level_ncand[f * kMaxLevels + level] = n;
    }
    __syncthreads();
    int size = lds_excl_scan<NT>(s_B, nIni, s_tmp);
    int cur = 0;
    for (int r = tid; r < nIni; r += NT)
        if (s_A[r] > 0) {
            const int pos = s_B[r];
            s_cnt[pos] = s_A[r];
            s_cidx[pos] = -1;
            s_path[pos] = r;
            s_depth[pos] = 0;
        }
    if (tid == 0) { s_var[4] = size; s_var[5] = 0; s_var[6] = 0; s_var[8] = 0; }
    __syncthreads();

    // ---- splitting steps: as in quadtree_body (a thread owns runs of list positions and of the processing order),
    // the child occupancy read from the sums
    int* const s_tmpA = s_var + 16;     // scan scratch, two arrays used in turn (32 ints each)
    int* const s_tmpB = s_var + 48;
    const bool packed_rank = total < (1 << 19);
    QT_STEP_DECL;
    while (true) {
        size = s_var[4];
        const int phase2 = s_var[5];
        int* cc_cur = s_cnt + cur * M;
        int* cn_nxt = s_cnt + (cur ^ 1) * M;
        short* ci_cur = s_cidx + cur * M;
        short* ci_nxt = s_cidx + (cur ^ 1) * M;
        int* pa_cur = s_path + cur * M;
        int* pa_nxt = s_path + (cur ^ 1) * M;
        unsigned char* de_cur = s_depth + cur * M;
        unsigned char* de_nxt = s_depth + (cur ^ 1) * M;
        const int cp = (size + NT - 1) / NT;
        const int p0 = min(tid * cp, size), p1 = min(p0 + cp, size);

        int nf = 0;
        for (int p = p0; p < p1; p++) {
            const int e = phase2 ? (ci_cur[p] >= 0) : (cc_cur[p] > 1);
            s_fl[p] = (unsigned char)e;
            nf += e;
            int4 occ = make_int4(0, 0, 0, 0);
            if (e) {
                const int d = de_cur[p];
                if (d >= D) s_var[8] = 1;
                else occ = *reinterpret_cast<const int4*>(H + qt_depth_off(nIni, d + 1) + 4 * pa_cur[p]);
            }
            *reinterpret_cast<int4*>(&s_cc[4 * p]) = occ;
        }
        if (tid == 0) s_var[7] = 0;
        int m;
        int fpos = block_excl_scan1<NT>(nf, s_tmpA, &m);
        QT_STEP(0);
        if (s_var[8]) {                                    // deeper than the leaves: the passes over the keys take this level
            __syncthreads();
            return false;
        }
        for (int p = p0; p < p1; p++)
            if (s_fl[p]) {
                if (!phase2) {
                    s_order[fpos] = (unsigned short)p;
                } else {
                    s_cand[fpos] = (unsigned short)p;
                    // (size, creation index) as one key when the sizes fit beside the 13 bits of a creation index
                    s_key[fpos] = packed_rank ? (((uint32_t)cc_cur[p] << 13) | (uint32_t)ci_cur[p]) : (uint32_t)cc_cur[p];
                    s_B[fpos] = ci_cur[p];
                }
                fpos++;
            }
        __syncthreads();
        QT_STEP(1);
        int norder = m;
        if (phase2) {
            // rank in (size, creation) descending order: G lanes share one candidate's m comparisons
            int G = 1;
            while (G < 64 && m * (2 * G) <= NT) G *= 2;
            for (int i0 = 0; i0 < m; i0 += NT / G) {
                const int i = i0 + tid / G, sub = tid & (G - 1);
                int r = 0;
                if (i < m) {
                    const uint32_t ki = s_key[i];
                    if (packed_rank) {
                        r = rank_packed(s_key, m, ki, sub, G);
                    } else {
                        const int ci = s_B[i];
                        for (int j = sub; j < m; j += G) {
                            const uint32_t kj = s_key[j];
                            r += (kj > ki) || (kj == ki && s_B[j] > ci);
                        }
                    }
                }
                for (int d = 1; d < G; d <<= 1) r += __shfl_xor(r, d, 64);
                if (i < m && sub == 0) s_order[r] = s_cand[i];
            }
            __syncthreads();
            // A level with fewer keys than features to keep can never hold N nodes (no node is empty), so every
            // sweep splits all its candidates: no search for the split that reaches N.
            if (total >= N) {
            const int cs = (m + NT - 1) / NT;
            const int s0 = min(tid * cs, m), s1 = min(s0 + cs, m);
            int gsum = 0;
            for (int s_ = s0; s_ < s1; s_++) {
                const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
                gsum += (occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0) - 1;
            }
            int gtot;
            int gpos = block_excl_scan1<NT>(gsum, s_tmpB, &gtot);
            int below = 0;
            for (int s_ = s0; s_ < s1; s_++) {
                const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
                const int gr = (occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0) - 1;
                below += (size + gpos + gr < N);
                gpos += gr;
            }
            if (below) atomicAdd(&s_var[7], below);
            __syncthreads();
            norder = min(s_var[7] + 1, m);
            for (int s_ = s0; s_ < s1; s_++) s_fl[s_order[s_]] = (unsigned char)(s_ < norder);
            __syncthreads();
            }
            QT_STEP(2);
        }
        const int co = (norder + NT - 1) / NT;
        const int o0 = min(tid * co, norder), o1 = min(o0 + co, norder);
        int sum_par = 0, nu = 0;
        for (int s_ = o0; s_ < o1; s_++) {
            const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
            sum_par += ((occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0)) |
                       (((occ.x > 1) + (occ.y > 1) + (occ.z > 1) + (occ.w > 1)) << 16);
        }
        for (int p = p0; p < p1; p++) nu += !s_fl[p];
        int par_pos, old_pos, both, nold;
        block_excl_scan1_pair<NT>(sum_par, nu, phase2 ? s_tmpA : s_tmpB, &par_pos, &old_pos, &both, &nold);
        QT_STEP(3);
        const int totalch = both & 0xffff, ncreated = both >> 16;
        const int newsize = totalch + nold;
        for (int s_ = o0; s_ < o1; s_++) {
            const int p = s_order[s_];
            const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * p]);
            const int oc[4] = {occ.x, occ.y, occ.z, occ.w};
            const int nch = (oc[0] > 0) + (oc[1] > 0) + (oc[2] > 0) + (oc[3] > 0);
            int pos = totalch - ((par_pos & 0xffff) + nch);   // children of later parents come first
            int cb = par_pos >> 16;
            int cpos[4] = {0, 0, 0, 0};
#pragma unroll
            for (int qd = 3; qd >= 0; qd--)
                if (oc[qd] > 0) {
                    cn_nxt[pos] = oc[qd];
                    ci_nxt[pos] = -1;
                    pa_nxt[pos] = 4 * pa_cur[p] + qd;
                    de_nxt[pos] = (unsigned char)(de_cur[p] + 1);
                    cpos[qd] = pos;
                    pos++;
                }
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                if (oc[qd] > 1) ci_nxt[cpos[qd]] = (short)(cb++);
            par_pos += nch | (((oc[0] > 1) + (oc[1] > 1) + (oc[2] > 1) + (oc[3] > 1)) << 16);
        }
        for (int p = p0; p < p1; p++)
            if (!s_fl[p]) {
                const int pos = totalch + old_pos++;
                cn_nxt[pos] = cc_cur[p];
                ci_nxt[pos] = -1;
                pa_nxt[pos] = pa_cur[p];
                de_nxt[pos] = de_cur[p];
            }
        if (tid == 0) {
            s_var[4] = newsize;
            if (newsize >= N || newsize == size) s_var[6] = 1;
            else if (!phase2 && newsize + 3 * ncreated > N) s_var[5] = 1;
        }
        cur ^= 1;
        __syncthreads();
        QT_STEP(4);
        QT_STEP_END();
        if (s_var[6]) break;
    }

    QT_STEP_FLUSH();
    QT_MARK(3);
    QT_NOTE(5, 0);
    // ---- best key of every node (:689-710): the maximum over its leaves, decoded from the candidate slot
    size = s_var[4];
    const uint32_t* cands = cell_cand + (size_t)f * g.cand_per_frame + L.cand_base;
    uint2* stage = kp_stage + (size_t)f * g.kp_per_frame + L.kp_base;
    for (int p = tid; p < size && p < L.kp_cap; p += NT) {
        const unsigned long long b = B[qt_depth_off(nIni, s_depth[cur * M + p]) + s_path[cur * M + p]];
        const uint32_t slot = 0xFFFFFFFFu - (uint32_t)(b & 0xFFFFFFFFull);
        const uint32_t u = cands[slot];
        const int c = (int)(slot / (uint32_t)L.cell_cap);
        const int ci = c / L.nCols, cj = c - ci * L.nCols;
        const int x = cj * L.wCell + 3 + (int)((u >> 8) & 0xff), y = ci * L.hCell + 3 + (int)(u >> 16);
        stage[p] = make_uint2((uint32_t)x | ((uint32_t)y << 16), (uint32_t)(b >> 32));
    }
    if (tid == 0) level_count[f * kMaxLevels + level] = min(size, L.kp_cap);
    QT_MARK(4);
    return true;
}

// NB = bits of a node index in a key's node word (node | quadrant << NB, 16 bits in all).  NB = 11: the node
// arrays of up to 2048 nodes live in LDS.  NB = 14: levels asked for more features than that keep the same arrays
// in a per-workgroup slice of global memory (gscratch): far slower, but the result is the same and no setting is refused
// up to 16376 features on one level.
// UF = keys beyond the register keys that a thread loads before it works on them (levels of large frames have far more
// candidates than registers hold: a 3840x2160 frame brings 80 000 to level 0, and a loop of one dependent load per key
// and pass is bound by the load latency).
template <int NT, int KR, int NB, int UF = 1>
__device__ __forceinline__ void quadtree_body(const FrameGeom& g, const uint32_t* __restrict__ cell_cand,
                                                  const int* __restrict__ cell_count, int* __restrict__ cell_off,
                                                  uint32_t* __restrict__ key_xy, uint8_t* __restrict__ key_resp,
                                                  uint16_t* __restrict__ key_node, uint2* __restrict__ kp_stage,
                                                  int* __restrict__ level_count, int* __restrict__ level_ncand,
                                                  int off_in_lds, unsigned char* __restrict__ gscratch, size_t gstride,
                                                  const CountLevels& cl, unsigned char* __restrict__ count_scratch)
{
    extern __shared__ __align__(16) unsigned char smem_lds[];
    unsigned char* const smem = NB > 11 ? gscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * gstride : smem_lds;
    constexpr uint32_t kNodeMask = (1u << NB) - 1u;
    const int level = blockIdx.y, f = blockIdx.x;   // x = frame: the large low levels are dispatched first
    const LevelGeom& L = g.lv[level];
    const int tid = threadIdx.x;
    if (level >= g.nlevels || L.width == 0) {
        if (tid == 0) { level_count[f * kMaxLevels + level] = 0; level_ncand[f * kMaxLevels + level] = 0; }
        return;
    }
    QT_MARK(0);
    // A level with many times more candidates than features to keep runs on node counts (k_qt_bins has filled the
    // leaves): its tree stays shallow.  The passes over the keys below serve it only if the tree grows deeper than the
    // leaves after all; a level that keeps most of its candidates splits down to single keys and goes there directly.
    if (NB == 11 && cl.n > 0 && cl.slot[level] >= 0) {
        const int total = level_ncand[f * kMaxLevels + level];      // k_qt_totals has written it
        QT_MARK(1);
        QT_NOTE(6, total);
        if (total >= cl.min_ratio * L.nfeat &&
            quadtree_counts_body<NT>(g, cl, cl.slot[level], level, f, total, cell_cand, kp_stage, level_count, level_ncand, count_scratch, smem_lds))
            return;
    }
    const int M = g.max_nodes;
    // LDS carve-up (8-byte items first; s_best shares s_cc, s_key shares s_C)
    QNode* s_rect = reinterpret_cast<QNode*>(smem);                                     // 2*M
    int* s_cc = reinterpret_cast<int*>(s_rect + 2 * M);                                 // 4*M
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_cc);           // M (after the last step)
    int* s_cnt = s_cc + 4 * M;                                                          // 2*M
    int* s_A = s_cnt + 2 * M;                                                           // M
    int* s_B = s_A + M;                                                                 // M
    int* s_C = s_B + M;                                                                 // M
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_C);                                 // M (only while ranking)
    int* s_var = s_C + M;                                                               // 96
    short* s_cidx = reinterpret_cast<short*>(s_var + 96);                               // 2*M
    unsigned short* s_childpos = reinterpret_cast<unsigned short*>(s_cidx + 2 * M);     // 4*M
    unsigned short* s_order = s_childpos + 5 * M;                                       // M (M entries before it are unused)
    unsigned short* s_cand = s_order + M;                                               // M
    unsigned char* s_fl = reinterpret_cast<unsigned char*>(s_cand + M);                 // M
    int* s_tmp = s_var + 16;     // scan scratch, one int per wavefront (two for the paired scan)
    // s_var[4]=size [5]=phase2 [6]=finish [7]=cntlt

    const int N = L.nfeat;
    const int ncell = L.nCols * L.nRows;
    const int* ccount = cell_count + (size_t)f * g.ncells + L.cell_base;
    (void)cell_off;
    const uint32_t* cands = cell_cand + (size_t)f * g.cand_per_frame + L.cand_base;
    uint32_t* kxy = key_xy + (size_t)f * g.keys_per_frame + L.key_base;
    uint8_t* kresp = key_resp + (size_t)f * g.keys_per_frame + L.key_base;
    uint16_t* knode = key_node + (size_t)f * g.keys_per_frame + L.key_base;

    // ---- gather the cells' lists into one list in the reference's append order:
    // exclusive scan of the cell counts, then one thread per key finds its cell by
    // binary search over the offsets when they fit in LDS; otherwise the owner of a cell copies its list.
    int n;
    int* s_off = reinterpret_cast<int*>(smem + quadtree_fixed_lds(M));    // ncell + 1 entries, if off_in_lds
    {
        const int chunk = (ncell + NT - 1) / NT;
        const int b = min(tid * chunk, ncell), e = min(b + chunk, ncell);
        int s = 0;
        for (int c = b; c < e; c++) s += ccount[c];
        int base = block_excl_scan<NT>(s, s_tmp, &n);
        for (int c = b; c < e; c++) {
            const int cnt = ccount[c];
            if (off_in_lds) {
                s_off[c] = base;
            } else {
                // Too many cells for their offsets to fit in LDS (large frames): the thread that owns the cell copies
                // its list straight to the key arrays, instead of one binary search through global memory per key.
                const int ci = L.nCols > 1 ? (int)__umulhi((uint32_t)c, L.rcp_ncols) : c, cj = c - __mul24(ci, L.nCols);
                const int x0 = __mul24(cj, L.wCell) + 3, y0 = __mul24(ci, L.hCell) + 3;
                const uint32_t* src = cands + (uint32_t)__mul24(c, L.cell_cap);
                for (int j = 0; j < cnt; j++) {
                    const uint32_t u = src[j];
                    kxy[base + j] = (uint32_t)(x0 + (int)((u >> 8) & 0xff)) | ((uint32_t)(y0 + (int)(u >> 16)) << 16);
                    kresp[base + j] = (uint8_t)(u & 0xff);
                }
            }
            base += cnt;
        }
        if (tid == NT - 1) { if (off_in_lds) s_off[ncell] = n; }
        __syncthreads();
    }
    if (tid == 0) level_ncand[f * kMaxLevels + level] = n;
    QT_MARK(1);
    QT_NOTE(6, n);
    QT_NOTE(5, 1);

    // The first KR*NT keys live in registers (coordinates, response and current node);
    // only the rare overflow beyond that walks the global arrays.  FOR_KEYS(body) runs body
    // for every key of this thread with `i` (key index), `xy` (packed coordinates) and `nd`
    // (node word: node | quadrant << NB | response << 16, written back afterwards).
    uint32_t rxy[KR], rnode[KR];
    auto fetch_key = [&](int i, uint32_t& xy, uint32_t& resp) {
        int lo = 0, hi = ncell;                 // largest c with off[c] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= i) lo = mid; else hi = mid;
        }
        const int c = lo;
        const int o = s_off[c];
        // small operands: 24-bit multiplies and a reciprocal instead of 32-bit multiplies and a division
        const uint32_t u = cands[(uint32_t)(__mul24(c, L.cell_cap) + (i - o))];
        const int ci = L.nCols > 1 ? (int)__umulhi((uint32_t)c, L.rcp_ncols) : c, cj = c - __mul24(ci, L.nCols);
        const int x = __mul24(cj, L.wCell) + 3 + (int)((u >> 8) & 0xff);
        const int y = __mul24(ci, L.hCell) + 3 + (int)(u >> 16);
        xy = (uint32_t)x | ((uint32_t)y << 16);
        resp = u & 0xff;
        kxy[i] = xy;
        kresp[i] = (uint8_t)resp;
    };
#pragma unroll
    for (int k = 0; k < KR; k++) {
        const int i = tid + NT * k;
        rxy[k] = 0u;
        rnode[k] = 0u;
        if (i < n) {
            if (off_in_lds) {
                uint32_t resp;
                fetch_key(i, rxy[k], resp);
                rnode[k] = resp << 16;
            } else {
                rxy[k] = kxy[i];
                rnode[k] = (uint32_t)kresp[i] << 16;
            }
        }
    }
    for (int i = tid + NT * KR; i < n; i += NT) {
        if (off_in_lds) {
            uint32_t xy, resp;
            fetch_key(i, xy, resp);
        }
        knode[i] = 0;
    }
    __syncthreads();
#define FOR_KEYS(...)                                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < KR; k_++) {                   \
        const int i = tid + NT * k_;                                           \
        if (i < n) {                                                            \
            const uint32_t xy = rxy[k_];                                        \
            uint32_t nd = rnode[k_] & 0xffffu;                                  \
            const uint32_t resp = rnode[k_] >> 16;                              \
            (void)xy; (void)resp;                                               \
            __VA_ARGS__;                                                        \
            rnode[k_] = (nd & 0xffffu) | (resp << 16);                          \
        }                                                                       \
    }                                                                           \
    for (int i0_ = tid + NT * KR; i0_ < n; i0_ += NT * UF) {                    \
        uint32_t oxy_[UF], ond_[UF], ors_[UF];                                  \
        _Pragma("unroll") for (int u_ = 0; u_ < UF; u_++) {                   \
            const int i = i0_ + NT * u_;                                        \
            oxy_[u_] = 0u; ond_[u_] = 0u; ors_[u_] = 0u;                        \
            if (i < n) { oxy_[u_] = kxy[i]; ond_[u_] = knode[i]; ors_[u_] = kresp[i]; } \
        }                                                                       \
        _Pragma("unroll") for (int u_ = 0; u_ < UF; u_++) {                   \
            const int i = i0_ + NT * u_;                                        \
            if (i < n) {                                                        \
                const uint32_t xy = oxy_[u_];                                   \
                uint32_t nd = ond_[u_];                                         \
                const uint32_t resp = ors_[u_];                                 \
                (void)xy; (void)resp;                                           \
                __VA_ARGS__;                                                    \
                knode[i] = (uint16_t)nd;                                        \
            }                                                                   \
        }                                                                       \
    }

    // ---- root nodes (:492-534)
    const int nIni = L.nIni;
    const float hX = L.hX;
    for (int r = tid; r < nIni; r += NT) s_A[r] = 0;
    __syncthreads();
    // every key adds to one of nIni (1 to 4) counters: count per wavefront with ballots and add once per
    // wavefront and counter instead of queueing thousands of atomics on a handful of addresses
    FOR_KEYS({
        int x = (int)(xy & 0xffff);
        int r = (int)__fdiv_rn((float)x, hX);
        r = min(max(r, 0), nIni - 1);
        if (nIni <= 4) {
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);      // lanes that hold a key
            const int leader = (int)__ffsll((long long)act) - 1;
            for (int rr = 0; rr < nIni; rr++) {
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(r == rr);
                if ((int)(threadIdx.x & 63) == leader && mk) atomicAdd(&s_A[rr], __popcll(mk));
            }
        } else {
            atomicAdd(&s_A[r], 1);
        }
        nd = (uint32_t)r;
    })
    __syncthreads();
    for (int r = tid; r < nIni; r += NT) s_B[r] = s_A[r] > 0;
    __syncthreads();
    int size = lds_excl_scan<NT>(s_B, nIni, s_tmp);
    int cur = 0;
    for (int r = tid; r < nIni; r += NT) {
        s_childpos[4 * r] = (unsigned short)s_B[r];      // root counter -> list position, applied by the first key pass
        if (s_A[r] > 0) {
            int pos = s_B[r];
            QNode q;
            q.x0 = (short)(int)__fmul_rn(hX, (float)r);
            q.x1 = (short)(int)__fmul_rn(hX, (float)(r + 1));
            q.y0 = 0;
            q.y1 = (short)L.height;
            s_rect[pos] = q;
            s_cnt[pos] = s_A[r];
            s_cidx[pos] = -1;
        }
    }
    if (tid == 0) { s_var[4] = size; s_var[5] = 0; s_var[6] = 0; }
    __syncthreads();

    QT_MARK(2);
    const bool packed_rank = NB == 11 && n < (1 << 19);   // the packed key holds creation indices below 4 * 2048
    // ---- splitting steps.  A thread owns a run of consecutive list positions and a run of consecutive places in the
    // processing order, so that the scans of a step need only the other threads' totals: five barriers per step while
    // every multi-key node is split, nine in the size-ordered sweeps (each stretch between two barriers costs about
    // as much as the others, whatever the number of nodes).
    int* const s_tmpA = s_var + 16;     // scan scratch, two arrays used in turn (32 ints each)
    int* const s_tmpB = s_var + 48;
    QT_STEP_DECL;
    while (true) {
        size = s_var[4];
        const int phase2 = s_var[5];
        QNode* rc = s_rect + cur * M;
        QNode* rn = s_rect + (cur ^ 1) * M;
        int* cc_cur = s_cnt + cur * M;
        int* cn_nxt = s_cnt + (cur ^ 1) * M;
        short* ci_cur = s_cidx + cur * M;
        short* ci_nxt = s_cidx + (cur ^ 1) * M;
        const int cp = (size + NT - 1) / NT;
        const int p0 = min(tid * cp, size), p1 = min(p0 + cp, size);

        // a. nodes that may be split in this step; clear child counts; their index among the flagged nodes in list order
        int nf = 0;
        for (int p = p0; p < p1; p++) {
            const int e = phase2 ? (ci_cur[p] >= 0) : (cc_cur[p] > 1);
            s_fl[p] = (unsigned char)e;
            nf += e;
            *reinterpret_cast<int4*>(&s_cc[4 * p]) = make_int4(0, 0, 0, 0);
        }
        if (tid == 0) s_var[7] = 0;
        int m;
        int fpos = block_excl_scan1<NT>(nf, s_tmpA, &m);
        QT_STEP(0);
        // b. processing order of the parents: list order while every multi-key node is split; the sweeps rank them below
        for (int p = p0; p < p1; p++)
            if (s_fl[p]) {
                if (!phase2) {
                    s_order[fpos] = (unsigned short)p;
                } else {
                    s_cand[fpos] = (unsigned short)p;
                    // (size, creation index) as one key: creation indices are distinct and below 4 * 2048;
                    // a level with 2^19 keys or more (sizes that do not fit) compares the two fields instead
                    s_key[fpos] = packed_rank ? (((uint32_t)cc_cur[p] << 13) | (uint32_t)ci_cur[p]) : (uint32_t)cc_cur[p];
                    s_B[fpos] = ci_cur[p];
                }
                fpos++;
            }
        // c. child occupancy (DivideNode :431-476)
        //    A key first moves to its node's position in this step's list: s_childpos still holds the previous step's
        //    map (child position per quadrant of a split node, the new position in all four entries of any other node).
        FOR_KEYS({
            int p = (int)s_childpos[4 * (int)(nd & kNodeMask) + (int)(nd >> NB)];
            nd = (uint32_t)p;
            if (s_fl[p]) {
                QNode q = rc[p];
                int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
                int hx = ceil_half(q.x1 - q.x0), hy = ceil_half(q.y1 - q.y0);
                int qd = (x < q.x0 + hx ? 0 : 1) + (y < q.y0 + hy ? 0 : 2);
                // levels with very many keys (large frames) are bound by same-address LDS atomics; ordinary levels are not
                if (UF > 1 && n > 8192) wave_count(&s_cc[0], 4 * p + qd);
                else atomicAdd(&s_cc[4 * p + qd], 1);
                nd = (uint32_t)(p | (qd << NB));
            }
        })
        __syncthreads();
        QT_STEP(1);
        int norder = m;
        if (phase2) {
            // rank in (size, creation) descending order: G lanes share one candidate's m comparisons
            int G = 1;
            while (G < 64 && m * (2 * G) <= NT) G *= 2;
            for (int i0 = 0; i0 < m; i0 += NT / G) {
                const int i = i0 + tid / G, sub = tid & (G - 1);
                int r = 0;
                if (i < m) {
                    const uint32_t ki = s_key[i];
                    if (packed_rank) {
                        r = rank_packed(s_key, m, ki, sub, G);
                    } else {
                        const int ci = s_B[i];
                        for (int j = sub; j < m; j += G) {
                            const uint32_t kj = s_key[j];
                            r += (kj > ki) || (kj == ki && s_B[j] > ci);
                        }
                    }
                }
                for (int d = 1; d < G; d <<= 1) r += __shfl_xor(r, d, 64);
                if (i < m && sub == 0) s_order[r] = s_cand[i];
            }
            __syncthreads();
            // A level with fewer keys than features to keep can never hold N nodes (no node is empty), so every
            // sweep splits all its candidates: no search for the split that reaches N.
            if (n >= N) {
            // growth of |L| per split, in processing order; the splits up to and including the first that reaches N
            const int cs = (m + NT - 1) / NT;
            const int s0 = min(tid * cs, m), s1 = min(s0 + cs, m);
            int gsum = 0;
            for (int s_ = s0; s_ < s1; s_++) {
                const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
                gsum += (occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0) - 1;
            }
            int gtot;
            int gpos = block_excl_scan1<NT>(gsum, s_tmpB, &gtot);
            int below = 0;
            for (int s_ = s0; s_ < s1; s_++) {
                const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
                const int gr = (occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0) - 1;
                below += (size + gpos + gr < N);
                gpos += gr;
            }
            if (below) atomicAdd(&s_var[7], below);
            __syncthreads();
            norder = min(s_var[7] + 1, m);
            for (int s_ = s0; s_ < s1; s_++) s_fl[s_order[s_]] = (unsigned char)(s_ < norder);
            __syncthreads();
            }
            QT_STEP(2);
        }
        // d. positions: children and expandable children of the parents before this thread's, untouched nodes before its own
        const int co = (norder + NT - 1) / NT;
        const int o0 = min(tid * co, norder), o1 = min(o0 + co, norder);
        int sum_par = 0, nu = 0;
        for (int s_ = o0; s_ < o1; s_++) {
            const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * (int)s_order[s_]]);
            // both counts in one word (each sum stays below 2^16: at most 4 * 2048), one scan for the two
            sum_par += ((occ.x > 0) + (occ.y > 0) + (occ.z > 0) + (occ.w > 0)) |
                       (((occ.x > 1) + (occ.y > 1) + (occ.z > 1) + (occ.w > 1)) << 16);
        }
        for (int p = p0; p < p1; p++) nu += !s_fl[p];
        int par_pos, old_pos, both, nold;
        block_excl_scan1_pair<NT>(sum_par, nu, phase2 ? s_tmpA : s_tmpB, &par_pos, &old_pos, &both, &nold);
        QT_STEP(3);
        const int totalch = both & 0xffff, ncreated = both >> 16;
        const int newsize = totalch + nold;
        // e. build the next list
        for (int s_ = o0; s_ < o1; s_++) {
            const int p = s_order[s_];
            const int4 occ = *reinterpret_cast<const int4*>(&s_cc[4 * p]);
            const int oc[4] = {occ.x, occ.y, occ.z, occ.w};
            QNode q = rc[p];
            int hx = ceil_half(q.x1 - q.x0), hy = ceil_half(q.y1 - q.y0);
            int nch = (oc[0] > 0) + (oc[1] > 0) + (oc[2] > 0) + (oc[3] > 0);
            int pos = totalch - ((par_pos & 0xffff) + nch);   // children of later parents come first
            int cpos[4] = {0, 0, 0, 0};
#pragma unroll
            for (int qd = 3; qd >= 0; qd--)
                if (oc[qd] > 0) {
                    QNode c;
                    c.x0 = (qd & 1) ? (short)(q.x0 + hx) : q.x0;
                    c.x1 = (qd & 1) ? q.x1 : (short)(q.x0 + hx);
                    c.y0 = (qd & 2) ? (short)(q.y0 + hy) : q.y0;
                    c.y1 = (qd & 2) ? q.y1 : (short)(q.y0 + hy);
                    rn[pos] = c;
                    cn_nxt[pos] = oc[qd];
                    ci_nxt[pos] = -1;
                    s_childpos[4 * p + qd] = (unsigned short)pos;
                    cpos[qd] = pos;
                    pos++;
                }
            int cb = par_pos >> 16;
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                if (oc[qd] > 1) ci_nxt[cpos[qd]] = (short)(cb++);
            par_pos += nch | (((oc[0] > 1) + (oc[1] > 1) + (oc[2] > 1) + (oc[3] > 1)) << 16);
        }
        for (int p = p0; p < p1; p++)
            if (!s_fl[p]) {
                const int pos = totalch + old_pos++;
                s_childpos[4 * p] = s_childpos[4 * p + 1] = s_childpos[4 * p + 2] = s_childpos[4 * p + 3] = (unsigned short)pos;
                rn[pos] = rc[p];
                cn_nxt[pos] = cc_cur[p];
                ci_nxt[pos] = -1;
            }
        // f. the keys move at the start of the next step's pass over them (or of the final one)
        // g. loop control (:618-622, :683-684)
        if (tid == 0) {
            s_var[4] = newsize;
            if (newsize >= N || newsize == size) s_var[6] = 1;
            else if (!phase2 && newsize + 3 * ncreated > N) s_var[5] = 1;
        }
        cur ^= 1;
        __syncthreads();
        QT_STEP(4);
        QT_STEP_END();
        if (s_var[6]) break;
    }

    QT_STEP_FLUSH();
    QT_MARK(3);
    // ---- best key of every node (:689-710): largest response, first key on ties
    size = s_var[4];
    for (int p = tid; p < size; p += NT) s_best[p] = 0ull;
    __syncthreads();
    FOR_KEYS({
        nd = (uint32_t)s_childpos[4 * (int)(nd & kNodeMask) + (int)(nd >> NB)];
        unsigned long long v = ((unsigned long long)resp << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
        atomicMax(&s_best[nd], v);
    })
#undef FOR_KEYS
    __syncthreads();
    uint2* stage = kp_stage + (size_t)f * g.kp_per_frame + L.kp_base;
    for (int p = tid; p < size && p < L.kp_cap; p += NT) {
        uint32_t i = 0xFFFFFFFFu - (uint32_t)(s_best[p] & 0xFFFFFFFFull);
        stage[p] = make_uint2(kxy[i], (uint32_t)(s_best[p] >> 32));
    }
    if (tid == 0) level_count[f * kMaxLevels + level] = min(size, L.kp_cap);
    QT_MARK(4);
}

#define QUADTREE_PARAMS FrameGeom g, const uint32_t* __restrict__ cell_cand, const int* __restrict__ cell_count,          \
                        int* __restrict__ cell_off, uint32_t* __restrict__ key_xy, uint8_t* __restrict__ key_resp,      \
                        uint16_t* __restrict__ key_node, uint2* __restrict__ kp_stage, int* __restrict__ level_count,   \
                        int* __restrict__ level_ncand, int off_in_lds, unsigned char* __restrict__ gscratch, size_t gstride, \
                        CountLevels cl, unsigned char* __restrict__ count_scratch
#define QUADTREE_ARGS g, cell_cand, cell_count, cell_off, key_xy, key_resp, key_node, kp_stage, level_count, level_ncand, off_in_lds, gscratch, gstride, cl, count_scratch
// Two shapes of the same body.  256 threads x 10 register keys (91 VGPRs, 5 wavefronts per SIMD) is
// the throughput shape for batches.  1024 threads x 3 register keys shortens the per-key passes of
// a lone workgroup: the latency shape for a handful of frames, where the device is mostly idle.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_quadtree(QUADTREE_PARAMS) { quadtree_body<256, kKeyRegs, 11>(QUADTREE_ARGS); }
__global__ __launch_bounds__(1024) void k_quadtree_wide(QUADTREE_PARAMS) { quadtree_body<1024, kKeyRegsWide, 11, 8>(QUADTREE_ARGS); }
// 512 threads x 5 register keys: between the two, for batches that fill the device only once
__global__ __launch_bounds__(512) void k_quadtree_mid(QUADTREE_PARAMS) { quadtree_body<512, 5, 11>(QUADTREE_ARGS); }
// more than 2048 nodes on a level: node arrays in global memory
__global__ __launch_bounds__(1024) void k_quadtree_big(QUADTREE_PARAMS) { quadtree_body<1024, kKeyRegsWide, 14, 8>(QUADTREE_ARGS); }
#undef QUADTREE_ARGS
#undef QUADTREE_PARAMS

void make_fast_lds(const FrameGeom& g, FastLds& fl)
{
    int wc = 1, hc = 1;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].width > 0) { wc = std::max(wc, g.lv[l].wCell); hc = std::max(hc, g.lv[l].hCell); }
    fl.tile_pitch = (wc + 6 + 3 + 3) & ~3;                 // + up to 3 bytes of dword misalignment
    fl.score_pitch = (wc + 2 + 3) & ~3;
    // the usual 30-pixel cells fit the pitches the specialised kernel is compiled for
    fl.fixed_pitch = fl.tile_pitch <= kFastTileFixed && fl.score_pitch <= kFastScoreFixed && !getenv("ORBX_FAST_VAR_PITCH");
    if (fl.fixed_pitch) { fl.tile_pitch = kFastTileFixed; fl.score_pitch = kFastScoreFixed; }
    const int tile_bytes = fl.tile_pitch * (hc + 6);
    const int score_bytes = fl.score_pitch * (hc + 2);
    fl.score_off = (tile_bytes + 15) & ~15;
    fl.queue_off = (fl.score_off + score_bytes + 15) & ~15;
    // a queue for 640 pretest survivors (natural images: about 200 of a cell's ~1000 pixels) instead of one
    // for every pixel keeps a wavefront's LDS under 5 KB, i.e. 8 wavefronts per SIMD
    // (never below 512: a pair of pretest rounds can queue eight pixels per lane)
    fl.queue_cap = std::max(512, std::min((wc * hc + 63) & ~63, getenv("ORBX_FAST_QCAP") ? atoi(getenv("ORBX_FAST_QCAP")) : 640));
    fl.total = fl.queue_off + 2 * fl.queue_cap + 16;
    fl.total = (fl.total + 15) & ~15;
}

void launch_fast_cells_wave(const uint8_t* pyr, const FrameGeom& g, const FastLds& fl, const FastCell* cells,
                            uint32_t* cell_cand, int* cell_count, int nframes, hipStream_t s, const RawL0& raw)
{
    if (g.ncells == 0) return;
    static const int env_map = getenv("ORBX_FAST_XCD_MAP") ? atoi(getenv("ORBX_FAST_XCD_MAP")) : -1;
    // one XCD per frame needs enough frames to keep all 8 busy
    const int per_xcd = env_map >= 0 ? env_map : (nframes >= 32);
    const unsigned long long rcp = ((1ull << 40) + (unsigned long long)g.ncells - 1) / (unsigned long long)g.ncells;
    dim3 grid = per_xcd ? dim3((unsigned)(8 * g.ncells * ((nframes + 7) / 8))) : dim3(g.ncells, nframes);
    if (fl.fixed_pitch)
        hipLaunchKernelGGL((k_fast_cells_wave<kFastTileFixed, kFastScoreFixed>), grid, dim3(64), (size_t)fl.total, s, pyr,
                           g, fl, cell_cand, cell_count, cells, per_xcd, rcp, nframes, raw);
    else
        hipLaunchKernelGGL(k_fast_cells_wave<>, grid, dim3(64), (size_t)fl.total, s, pyr, g, fl, cell_cand, cell_count,
                           cells, per_xcd, rcp, nframes, raw);
}

size_t quadtree_lds_bytes(int max_nodes) { return quadtree_fixed_lds(max_nodes); }

// ---------------------------------------------------------------------------
// orientation + descriptor + output assembly
// ---------------------------------------------------------------------------
// cv::fastAtan2 (OpenCV 3.4.1 atan_f32), every operation rounded separately
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cvRound(float): round half to even
__device__ __forceinline__ int round_rn(float v) { return __float2int_rn(v); }

// sin and cos of a double in [0, 2*pi] (the orientation in radians): quadrant reduction with a
// two-part pi/2 and the classic degree-13/14 minimax polynomials on [-pi/4, pi/4]; the error is
// below one unit in the last place of a double, far inside what rounding to float keeps.
__device__ __forceinline__ void sincos_turn(double x, double* sn, double* cs)
{
    const double kd = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kd, 1.57079632679489655800e+00, x);
    r = fma(-kd, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int k = (int)kd;
    const double s1 = (k & 1) ? c : s, c1 = (k & 1) ? s : c;
    *sn = (k & 2) ? -s1 : s1;
    *cs = ((k + 1) & 2) ? -c1 : c1;
}

// 32 lanes per keypoint, 8 keypoints per workgroup.  Workgroups are dealt round-robin over the
// 8 XCDs, and the patches of one frame's keypoints overlap heavily, so a large batch runs on a 1-D
// grid of 8 * nbx * ceil(frames/8) workgroups that gives every frame to one XCD (frame f -> XCD
// f % 8): its planes are then fetched into one L2 once instead of into all eight.  Small batches
// use the plain (block, frame) grid.  Either way the frames are taken LAST FIRST: the planes of a large batch (0.7 GB
// for 256 frames) do not fit the 256 MB memory-side cache, the launches before this one wrote them in ascending frame
// order, and so the last frames are the ones still cached (ascending order found none of them: 0.180 -> 0.166 ms).
// Descriptor sampling window: the rotated test points stay within 18 pixels of the keypoint
// (pattern coordinates are within +-13, 13 * sqrt(2) = 18.4), so a 37-row window of 10 aligned
// dwords per row (37 columns + up to 3 bytes of misalignment) holds every sample.
constexpr int kDescRadius = 18, kWinRows = 2 * kDescRadius + 1, kWinDw = 10, kWinItems = kWinRows * kWinDw;
static_assert(kDescRadius <= kPadY && kDescRadius + 3 <= kPadX, "window must stay inside the stored border");

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__global__ __launch_bounds__(256) void k_orient_desc(FrameGeom g, const uint8_t* __restrict__ pyr,
                                                     const uint8_t* __restrict__ blur,
                                                     const uint2* __restrict__ kp_stage,
                                                     const int* __restrict__ level_count,
                                                     orbx_keypoint* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                     int cap, int* __restrict__ n_out, int nbx, int nframes, int frame_per_xcd,
                                                     RawL0 raw)
{
    // the 256 test pairs as floats, s_pat[j * 32 + byte] = (x0, y0, x1, y1) of pair 8 * byte + j
    __shared__ float4 s_pat[256];
    __shared__ uint32_t s_win[8][kWinItems];
    const int tid = threadIdx.x, lane = tid & 31;
    {
        const int8_t* pp = c_pattern + 4 * tid;
        s_pat[(tid & 7) * 32 + (tid >> 3)] = make_float4((float)pp[0], (float)pp[1], (float)pp[2], (float)pp[3]);
    }
    __syncthreads();

    int f, bx;
    if (frame_per_xcd) {
        const int seq = blockIdx.x >> 3;              // position in this XCD's sequence
        const int fq = seq / nbx;
        bx = seq - fq * nbx;
        f = (((nframes + 7) >> 3) - 1 - fq) * 8 + (int)(blockIdx.x & 7);
        if (f >= nframes) return;
    } else {                                          // small batch: plain (block, frame) grid
        bx = blockIdx.x;
        f = nframes - 1 - (int)blockIdx.y;
    }
    // every level owns whole groups of 8 slots, so the level of this workgroup's 8 keypoints is
    // uniform: the search runs on the scalar unit
    const int slot0 = bx * 8;
    const int* lc = level_count + f * kMaxLevels;
    int level = -1, prefix = 0, total = 0;
#pragma unroll
    for (int l = 0; l < kMaxLevels; l++) {
        if (l < g.nlevels) {
            const int c = lc[l];
            if (g.lv[l].kp_cap > 0 && slot0 >= g.lv[l].kp_base && slot0 < g.lv[l].kp_base + g.lv[l].kp_cap) {
                level = l;
                prefix = total;
            }
            total += c;
        }
    }
    const int k = level >= 0 ? slot0 - g.lv[level].kp_base + (tid >> 5) : 0;
    if (bx == 0 && tid == 0) n_out[f] = total;
    if (level < 0 || k >= lc[level]) return;
    const int oi = prefix + k;
    if (oi >= cap) return;
    const LevelGeom& L = g.lv[level];
    const uint2 st = kp_stage[(size_t)f * g.kp_per_frame + L.kp_base + k];
    const int x = (int)(st.x & 0xffff) + kDetBorder, y = (int)(st.x >> 16) + kDetBorder;   // :1115-1116
    const int pitch = L.pitch;
    const int ctr = (int)L.plane_off + (kPadY + y) * pitch + kPadX + x;   // keypoint, byte offset inside the frame

    // Stage the blurred window of this keypoint in LDS with coalesced dword loads (issued first,
    // consumed after the orientation): the 16 scattered samples per lane then cost LDS reads
    // instead of one cache-line lookup each.
    const uint8_t* bf = blur + (size_t)f * g.pyr_frame_bytes;
    const int sh = (kPadX + x - kDescRadius) & 3;            // bytes between the aligned window start and column x-18
    const int win0 = ctr - kDescRadius * pitch - kDescRadius - sh;
    // 16 lanes per window row (10 of them load), two rows per step: more loads than a dense 32-lane
    // packing, but every address is the previous one plus two pitches
    constexpr int kWinSteps = (kWinRows + 1) / 2;
    const int wrow = lane >> 4, wq = lane & 15;
    uint32_t wv[kWinSteps];
    {
        uint32_t wo = (uint32_t)(win0 + __mul24(wrow, pitch) + 4 * min(wq, kWinDw - 1));
#pragma unroll
        for (int t = 0; t < kWinSteps; t++) {
            // the row after the last one (odd row count) is read like any other and dropped at the store
            wv[t] = *reinterpret_cast<const uint32_t*>(bf + wo);
            wo += 2u * (uint32_t)pitch;
        }
    }

    // IC_Angle :74-101 on the un-blurred level: lane = patch column u = lane-15, walking the
    // rows |v| <= umax[|u|] of its column (the disc is symmetric in u and v); 31 independent,
    // row-coalesced byte loads per lane, m10 = u * (column sum).
    // (level 0 from the caller's image when raw.on: a keypoint lies at least 19 pixels inside, its patch 15)
    const bool from_image = raw.on && level == 0;              // uniform over the workgroup
    const uint8_t* pf = from_image ? raw.base + (size_t)f * raw.frame_stride : pyr + (size_t)f * g.pyr_frame_bytes;
    const int mpitch = from_image ? raw.stride : pitch;
    const int mctr = from_image ? y * raw.stride + x : ctr;
    int m10 = 0, m01 = 0;
    {
        const int u = lane - kHalfPatch;
        const int d = lane < 31 ? c_umax[u < 0 ? -u : u] : -1;
        int colsum = 0;
#pragma unroll
        for (int v = -kHalfPatch; v <= kHalfPatch; v++) {
            // unconditional load (rows +-15 of a keypoint always lie inside the plane) so that
            // all 31 loads are in flight together; rows outside the disc contribute zero
            const int av = v < 0 ? -v : v;
            const int px = pf[(uint32_t)(mctr + v * mpitch + u)];
            const int val = av <= d ? px : 0;
            colsum += val;
            m01 += v * val;
        }
        m10 = u * colsum;
    }
#pragma unroll
    for (int dd = 16; dd >= 1; dd >>= 1) {
        m10 += __shfl_xor(m10, dd, 32);
        m01 += __shfl_xor(m01, dd, 32);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    uint32_t* win = s_win[tid >> 5];
    if (wq < kWinDw) {
#pragma unroll
        for (int t = 0; t < kWinSteps; t++)
            if (2 * t + 1 < kWinRows || wrow == 0) win[(2 * t + wrow) * kWinDw + wq] = wv[t];
    }
    WAVE_SYNC();

    // computeOrbDescriptor :104-150 on the blurred level; lane = descriptor byte
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    double sn, cs;
    sincos_turn((double)rad, &sn, &cs);
    const float a = (float)cs, b = (float)sn;
    const uint8_t* win8 = reinterpret_cast<const uint8_t*>(win) + kDescRadius * (4 * kWinDw) + kDescRadius + sh;
    // Rotation and rounding of a test point: (row, col) = (x b + y a, x a - y b), every product and
    // sum rounded on its own as in the reference (the compiler pairs them into two-float
    // instructions), then cvRound by adding 1.5 * 2^23, which leaves
    // 0x4B400000 + round-half-even(value) in the bits.  The window offset row * 40 + col comes
    // straight from those bits: a 24-bit multiply sees 0x400000 + row, the rest is one constant.
    constexpr uint32_t kUnbias = 0x400000u * (uint32_t)(4 * kWinDw) + 0x4B400000u;
    auto sample = [&](float px, float py) -> int {
        const float rr = __fadd_rn(__fadd_rn(__fmul_rn(px, b), __fmul_rn(py, a)), 12582912.f);
        const float cc = __fadd_rn(__fsub_rn(__fmul_rn(px, a), __fmul_rn(py, b)), 12582912.f);
        const uint32_t off = (uint32_t)__umul24(__builtin_bit_cast(uint32_t, rr), (uint32_t)(4 * kWinDw)) + __builtin_bit_cast(uint32_t, cc) - kUnbias;
        return win8[(int)off];
    };
    int val = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float4 pt = s_pat[j * 32 + lane];
        const int t0 = sample(pt.x, pt.y);
        const int t1 = sample(pt.z, pt.w);
        val |= (t0 < t1) << j;
    }
    out_desc[((size_t)f * cap + oi) * 32 + lane] = (uint8_t)val;
    if (lane == 0) {
        orbx_keypoint kp;
        kp.x = (float)x;
        kp.y = (float)y;
        if (level != 0) {   // :1179-1186
            kp.x = __fmul_rn(kp.x, L.scale);
            kp.y = __fmul_rn(kp.y, L.scale);
        }
        kp.size = L.patch_size;
        kp.angle = angle;
        kp.response = (float)st.y;
        kp.octave = level;
        kp.class_id = -1;
        out_kps[(size_t)f * cap + oi] = kp;
    }
}
#undef WAVE_SYNC

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_pyr_level0(const uint8_t* in, int stride, size_t frame_stride, uint8_t* pyr, const FrameGeom& g,
                       int nframes, hipStream_t s)
{
    const LevelGeom& L = g.lv[0];
    const int aligned = (((uintptr_t)in | (uintptr_t)frame_stride | (uintptr_t)stride) & 3u) == 0;
    dim3 grid(((L.pitch >> 4) * L.rows + 255) / 256, nframes);
    // the bytes the batch covers, rounded inwards to whole dwords
    const uintptr_t b0 = (uintptr_t)in, b1 = b0 + (size_t)(nframes - 1) * frame_stride + (size_t)(L.h - 1) * stride + L.w;
    const uint8_t* lo = reinterpret_cast<const uint8_t*>((b0 + 3) & ~(uintptr_t)3);
    const uint8_t* hi = reinterpret_cast<const uint8_t*>(b1 & ~(uintptr_t)3);
    hipLaunchKernelGGL(k_pyr_level0, grid, dim3(256), 0, s, in, stride, frame_stride, pyr, g, aligned, lo, hi);
}

void launch_pyr_resize(uint8_t* pyr, const FrameGeom& g, int level, const XTab* xtab, const YTab* ytab,
                       int nframes, hipStream_t s)
{
    const LevelGeom& L = g.lv[level];
    dim3 grid((L.pitch / 4 + 255) / 256, L.rows, nframes);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), 0, s, pyr, g, level, xtab, ytab);
}

void launch_resize_slide(uint8_t* pyr, const FrameGeom& g, int level, const XTab4* xtab, const YTab2* ytab,
                         int nframes, hipStream_t s, const RawL0& raw)
{
    const LevelGeom& L = g.lv[level];
    static const int env_rows = getenv("ORBX_RESIZE_ROWS") ? atoi(getenv("ORBX_RESIZE_ROWS")) : 0;
    const int rows_per_band = env_rows > 0 ? env_rows : 8;     // measured: 8 beats 4, 16 and 32 (short wavefronts hide the load latency, long ones reuse more rows)
    const int ncb = (L.pitch / 4 + 63) / 64;
    // Levels tall enough for a single reflection (all but the last levels of tiny images) compute image rows only, in
    // batches: a lone frame is bound by the latency of its seven dependent launches, where fewer, longer wavefronts
    // cost more than the border rows save (36 against 32 us).
    static const bool no_mirror = getenv("ORBX_RESIZE_NO_MIRROR") != nullptr;
    const int mirror = (!no_mirror && nframes >= 8 && L.h >= 2 * kPadY + 2) ? 1 : 0;
    const int nrows = mirror ? L.h : L.rows;
    const int nrb = (nrows + rows_per_band - 1) / rows_per_band;
    dim3 grid((ncb * nrb + 3) / 4, nframes);
    hipLaunchKernelGGL(k_resize_slide, grid, dim3(256), 0, s, pyr, g, level, xtab, ytab, rows_per_band, ncb, mirror, raw);
}

void make_slide_map(const FrameGeom& g, int target_rows, SlideMap& sm)
{
    int base = 0;
    for (int l = 0; l < kMaxLevels; l++) {
        sm.base[l] = base;
        sm.ndw[l] = 1;
        sm.rcp_ndw[l] = 0;
        sm.nrb[l] = 0;
        sm.rows[l] = 1;
        if (l < g.nlevels && g.lv[l].w > 0 && g.lv[l].h > 0) {
            const int nrb = (g.lv[l].h + target_rows - 1) / target_rows;
            sm.ndw[l] = (g.lv[l].w + 3) / 4;
            // exact for flat < 2^32 / ndw, far above bands * ndw (ndw = 1 is handled in the kernel)
            sm.rcp_ndw[l] = sm.ndw[l] > 1 ? (uint32_t)(((1ull << 32) + sm.ndw[l] - 1) / sm.ndw[l]) : 0u;
            sm.nrb[l] = nrb;
            sm.rows[l] = (g.lv[l].h + nrb - 1) / nrb;      // <= h
            base += (sm.ndw[l] * nrb + 63) / 64;
        }
    }
    sm.base[kMaxLevels] = base;
}

void launch_fast_blur(const uint8_t* pyr, uint8_t* blur, const FrameGeom& g, const FastLds& fl, const SlideMap& sm,
                      const FastCell* cells, uint32_t* cell_cand, int* cell_count, int nframes, hipStream_t s,
                      const RawL0& raw)
{
    const int nitems = g.ncells + sm.base[kMaxLevels];
    static const int env_map = getenv("ORBX_FAST_XCD_MAP") ? atoi(getenv("ORBX_FAST_XCD_MAP")) : -1;
    const int per_xcd = env_map >= 0 ? env_map : (nframes >= 32);
    const unsigned long long rcp = ((1ull << 40) + (unsigned long long)nitems - 1) / (unsigned long long)nitems;
    dim3 grid = per_xcd ? dim3((unsigned)(8 * nitems * ((nframes + 7) / 8))) : dim3(nitems, nframes);
    if (fl.fixed_pitch)
        hipLaunchKernelGGL((k_fast_blur<kFastTileFixed, kFastScoreFixed>), grid, dim3(64), (size_t)fl.total, s, pyr, blur, g,
                           fl, sm, cell_cand, cell_count, cells, per_xcd, rcp, nitems, nframes, raw);
    else
        hipLaunchKernelGGL(k_fast_blur<>, grid, dim3(64), (size_t)fl.total, s, pyr, blur, g, fl, sm, cell_cand, cell_count,
                           cells, per_xcd, rcp, nitems, nframes, raw);
}

void launch_blur_slide(const uint8_t* pyr, uint8_t* blur, const FrameGeom& g, const SlideMap& sm, int nframes,
                       hipStream_t s, const RawL0& raw)
{
    dim3 grid((sm.base[kMaxLevels] + 3) / 4, nframes);
    hipLaunchKernelGGL(k_blur_slide, grid, dim3(256), 0, s, pyr, blur, g, sm, raw);
}

static int max_level_cells(const FrameGeom& g)
{
    int m = 0;
    for (int l = 0; l < g.nlevels; l++) m = std::max(m, g.lv[l].nCols * g.lv[l].nRows);
    return m;
}

// dynamic LDS of the quadtree kernel; *off_in_lds says whether the cell offsets fit beside the nodes
static size_t quadtree_total_lds(const FrameGeom& g, int* off_in_lds)
{
    const size_t fixed = quadtree_fixed_lds(g.max_nodes);
    const size_t offs = ((size_t)max_level_cells(g) + 1) * 4;
    *off_in_lds = (fixed + offs <= 64 * 1024) ? 1 : 0;
    return *off_in_lds ? fixed + offs : fixed;
}

// bytes of global scratch one (frame, level) workgroup of k_quadtree_big uses; 0 when the LDS shapes serve the geometry
size_t quadtree_scratch_stride(const FrameGeom& g)
{
    if (g.max_nodes <= kMaxNodes) return 0;
    return (quadtree_fixed_lds(g.max_nodes) + ((size_t)max_level_cells(g) + 1) * 4 + 255) & ~(size_t)255;
}

// levels whose quadtree runs on node counts: many cells (so, as a rule, many thousands of candidates) and node arrays
// that fit LDS; depth of the leaves: as deep as kQtLeafBins leaves per level allow
constexpr int kQtCountCells = 800, kQtLeafBins = 32768;
void quadtree_count_levels(const FrameGeom& g, CountLevels& cl)
{
    cl = CountLevels{};
    for (int l = 0; l < kMaxLevels; l++) cl.slot[l] = -1;
    // read per geometry (tests switch them inside one process)
    const bool off = getenv("ORBX_QT_NO_COUNTS") != nullptr;
    const int min_cells = getenv("ORBX_QT_COUNT_CELLS") ? atoi(getenv("ORBX_QT_COUNT_CELLS")) : kQtCountCells;
    if (off || g.max_nodes > kMaxNodes) return;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        if (L.width == 0 || L.nCols * L.nRows < min_cells) continue;
        int D = 0;
        const int leaf_bins = getenv("ORBX_QT_LEAF_BINS") ? atoi(getenv("ORBX_QT_LEAF_BINS")) : kQtLeafBins;
        while (D < 8 && (L.nIni << (2 * (D + 1))) <= leaf_bins) D++;
        if (D < 3) continue;
        cl.level[cl.n] = l;
        cl.min_ratio = getenv("ORBX_QT_COUNT_RATIO") ? atoi(getenv("ORBX_QT_COUNT_RATIO")) : 4;
        cl.slot[l] = cl.n;
        cl.depth[cl.n] = D;
        cl.max_cells = std::max(cl.max_cells, L.nCols * L.nRows);
        cl.max_blocks = std::max(cl.max_blocks, L.nIni << (2 * qt_block_depth(D)));
        cl.stride = std::max(cl.stride, qt_counts_bytes(L.nIni, D));
        cl.n++;
    }
}

void launch_quadtree(const FrameGeom& g, const uint32_t* cell_cand, const int* cell_count, int* cell_off,
                     uint32_t* key_xy, uint8_t* key_resp, uint16_t* key_node, void* kp_stage, int* level_count,
                     int* level_ncand, int nframes, unsigned char* gscratch, const CountLevels* cl_in, unsigned char* count_scratch,
                     hipStream_t s)
{
    dim3 grid(nframes, kMaxLevels);
    int off_in_lds = 0;
    CountLevels cl{};
    for (int l = 0; l < kMaxLevels; l++) cl.slot[l] = -1;
    if (cl_in && cl_in->n > 0 && count_scratch && g.max_nodes <= kMaxNodes) {
        // the node counts of the levels that run on them: every block of leaves by a workgroup of its own
        cl = *cl_in;
        hipLaunchKernelGGL(k_qt_totals, dim3(cl.n, nframes), dim3(1024), 0, s, g, cl, cell_count, level_ncand);
        hipLaunchKernelGGL(k_qt_bins, dim3(cl.max_blocks, cl.n, nframes), dim3(kQtBinsThreads), 0, s, g, cl, cell_cand,
                           cell_count, level_ncand, count_scratch);
    }
    if (g.max_nodes > kMaxNodes) {      // node arrays (and the cell offsets behind them) in global memory
        hipLaunchKernelGGL(k_quadtree_big, grid, dim3(1024), 0, s, g, cell_cand, cell_count, cell_off, key_xy, key_resp,
                           key_node, reinterpret_cast<uint2*>(kp_stage), level_count, level_ncand, 1, gscratch,
                           quadtree_scratch_stride(g), cl, count_scratch);
        return;
    }
    size_t lds = quadtree_total_lds(g, &off_in_lds);
    // up to one workgroup per CU (32 frames x 8 levels) the wide shape is fastest, up to about 112 frames the
    // 512-thread one, beyond that the narrow one (measured: 0.040 against 0.052 ms at 64 frames, 0.106 against 0.061 at 256)
    // read per launch (a test switches it inside one process)
    const char* e_wide = getenv("ORBX_QT_WIDE_MAX");
    const char* e_mid = getenv("ORBX_QT_MID_MAX");
    const int wide_max = e_wide ? atoi(e_wide) : 32, mid_max = e_mid ? atoi(e_mid) : 112;
    if (nframes <= wide_max)
        hipLaunchKernelGGL(k_quadtree_wide, grid, dim3(1024), lds, s, g, cell_cand, cell_count, cell_off, key_xy, key_resp,
                           key_node, reinterpret_cast<uint2*>(kp_stage), level_count, level_ncand, off_in_lds, nullptr, (size_t)0,
                           cl, count_scratch);
    else if (nframes <= mid_max)
        hipLaunchKernelGGL(k_quadtree_mid, grid, dim3(512), lds, s, g, cell_cand, cell_count, cell_off, key_xy, key_resp,
                           key_node, reinterpret_cast<uint2*>(kp_stage), level_count, level_ncand, off_in_lds, nullptr, (size_t)0,
                           cl, count_scratch);
    else
        hipLaunchKernelGGL(k_quadtree, grid, dim3(256), lds, s, g, cell_cand, cell_count, cell_off, key_xy, key_resp,
                           key_node, reinterpret_cast<uint2*>(kp_stage), level_count, level_ncand, off_in_lds, nullptr, (size_t)0,
                           cl, count_scratch);
}

int quadtree_prepare(int max_nodes)
{
    if (max_nodes > kMaxNodes) return 0;     // k_quadtree_big keeps its arrays in global memory
    size_t lds = quadtree_lds_bytes(max_nodes);
    if (lds > 160 * 1024) return -1;
    if (lds > 64 * 1024) {
        // The opt-in for more than 64 KB of dynamic LDS belongs to the kernel on a device, not to a context: it is only
        // ever raised, so that a second context with a smaller quadtree cannot take away what the first one was granted.
        static std::mutex mu;
        static size_t granted[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 64 && lds <= granted[dev]) return 0;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree_wide),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree_mid),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) granted[dev] = lds;
    }
    return 0;
}

void launch_orient_desc(const FrameGeom& g, const uint8_t* pyr, const uint8_t* blur, const void* kp_stage,
                        const int* level_count, orbx_keypoint* out_kps, uint8_t* out_desc, int cap, int* n_out,
                        int nframes, hipStream_t s, const RawL0& raw)
{
    const int nbx = g.kp_per_frame > 0 ? (g.kp_per_frame + 7) / 8 : 1;
    const int per_xcd = nframes >= 32;                // one XCD per frame needs enough frames to keep all 8 busy
    dim3 grid = per_xcd ? dim3((unsigned)(8 * nbx * ((nframes + 7) / 8))) : dim3(nbx, nframes);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, s, g, pyr, blur, reinterpret_cast<const uint2*>(kp_stage),
                       level_count, out_kps, out_desc, cap, n_out, nbx, nframes, per_xcd, raw);
}

} // namespace orbx
