// orbx_match.hip -- gfx950 kernels of the Hamming searches.
//
// A search result is "the first two candidates in (distance, position)
// order", which is exactly what the reference's sequential scan
//     if (dist < bestDist) {second = best; best = dist} else if (dist < bestDist2) second = dist
// (src/ORBmatcher.cc:212-224) leaves behind.  Packing (distance << 22 | position)
// into one 32-bit key turns the scan into key = min / median-of-three updates
// that are associative, so the train set can be split over workgroups.
// Distances are popcounts of XORed 32-bit words (v_bcnt_u32_b32 accumulates).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <mutex>

#include "../../include/orbx.h"
#include "orbx_common.hpp"
#include "orbx_kernels.hpp"

namespace orbx {

constexpr uint32_t kKeyNone = (256u << 22) | 0x3FFFFFu;   // distance 256, no index (a distance of 256 never wins: 'dist < 256')

__device__ __forceinline__ int hamming256(const uint32_t q[8], const uint4 a, const uint4 b)
{
    return __popc(q[0] ^ a.x) + __popc(q[1] ^ a.y) + __popc(q[2] ^ a.z) + __popc(q[3] ^ a.w) +
           __popc(q[4] ^ b.x) + __popc(q[5] ^ b.y) + __popc(q[6] ^ b.z) + __popc(q[7] ^ b.w);
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    return max(min(a, b), min(max(a, b), c));
}

// (the minimum is written b-and-c first: min(a, b) would be shared with the median of the same three values, and the
// two would then take three instructions instead of v_min3_u32 + v_med3_u32.  Not as inline assembly: the compiler
// does not count the wait states between a matrix instruction and an assembly statement that reads its result.)
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }

// keep the two smallest keys seen so far (best <= second)
__device__ __forceinline__ void push_key(uint32_t& best, uint32_t& second, uint32_t key)
{
    second = umed3(best, second, key);
    best = min(best, key);
}

// grid (ceil(q_stride/256), tsplit, nbatch); thread = one query; the train
// descriptor address is wave-uniform, so its loads are scalar.
__global__ __launch_bounds__(256) void k_match_bruteforce(const uint32_t* __restrict__ q, int Q, int q_stride,
                                                          const int32_t* __restrict__ nq,
                                                          const uint32_t* __restrict__ t, int T, int t_stride,
                                                          const int32_t* __restrict__ nt, int tsplit,
                                                          uint32_t* __restrict__ partial,
                                                          int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                          int32_t* __restrict__ second_dist)
{
    const int b = blockIdx.z;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int nQ = nq ? min(nq[b], q_stride) : Q;
    const int nT = nt ? min(nt[b], T) : T;     // T is the per-batch maximum
    const int chunk = (nT + tsplit - 1) / tsplit;
    const int j0 = min((int)blockIdx.y * chunk, nT), j1 = min(j0 + chunk, nT);
    if (qi >= nQ) return;
    uint32_t qd[8];
    {
        const uint4* qp = reinterpret_cast<const uint4*>(q + ((size_t)b * q_stride + qi) * 8);
        uint4 a = qp[0], c = qp[1];
        qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w;
        qd[4] = c.x; qd[5] = c.y; qd[6] = c.z; qd[7] = c.w;
    }
    const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)b * t_stride * 8);
    uint32_t best = kKeyNone, second = kKeyNone;
#pragma unroll 4
    for (int j = j0; j < j1; j++) {
        const uint4 a = tp[2 * j], c = tp[2 * j + 1];
        const int d = hamming256(qd, a, c);
        push_key(best, second, ((uint32_t)d << 22) | (uint32_t)j);
    }
    const size_t o = (size_t)b * q_stride + qi;
    if (tsplit == 1) {
        best_idx[o] = (best >> 22) >= 256u ? -1 : (int32_t)(best & 0x3FFFFFu);
        best_dist[o] = (int32_t)(best >> 22);
        second_dist[o] = (int32_t)(second >> 22);
    } else {
        uint32_t* p = partial + (((size_t)b * tsplit + blockIdx.y) * q_stride + qi) * 2;
        p[0] = best;
        p[1] = second;
    }
}

__global__ __launch_bounds__(256) void k_match_merge(const uint32_t* __restrict__ partial, int Q, int q_stride,
                                                     const int32_t* __restrict__ nq, int tsplit,
                                                     int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                     int32_t* __restrict__ second_dist)
{
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int nQ = nq ? min(nq[b], q_stride) : Q;
    if (qi >= nQ) return;
    uint32_t best = kKeyNone, second = kKeyNone;
    for (int s = 0; s < tsplit; s++) {
        const uint32_t* p = partial + (((size_t)b * tsplit + s) * q_stride + qi) * 2;
        push_key(best, second, p[0]);
        push_key(best, second, p[1]);
    }
    const size_t o = (size_t)b * q_stride + qi;
    best_idx[o] = (best >> 22) >= 256u ? -1 : (int32_t)(best & 0x3FFFFFu);
    best_dist[o] = (int32_t)(best >> 22);
    second_dist[o] = (int32_t)(second >> 22);
}

// The same for many splits (few queries against a long train set: a thousand queries split 390 ways left four
// workgroups walking 390 dependent loads each, 99 us; this form 5 us).  Grid (ceil(q_stride/16), nbatch): sixteen
// queries per workgroup, sixteen threads per query that share the splits (a load instruction of the workgroup reads
// sixteen runs of 128 contiguous bytes), then the sixteen partial results of a query through LDS.
__global__ __launch_bounds__(256) void k_match_merge_wide(const uint32_t* __restrict__ partial, int Q, int q_stride,
                                                          const int32_t* __restrict__ nq, int tsplit,
                                                          int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                          int32_t* __restrict__ second_dist)
{
    __shared__ uint2 s_k[16][17];
    const int b = blockIdx.y;
    const int ql = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int qi = blockIdx.x * 16 + ql;
    const int nQ = nq ? min(nq[b], q_stride) : Q;
    uint32_t best = kKeyNone, second = kKeyNone;
    if (qi < nQ) {
        const uint2* p = reinterpret_cast<const uint2*>(partial) + (size_t)b * tsplit * q_stride + qi;
#pragma unroll 4
        for (int s = g; s < tsplit; s += 16) {
            const uint2 k = p[(size_t)s * q_stride];
            push_key(best, second, k.x);
            push_key(best, second, k.y);
        }
    }
    s_k[g][ql] = make_uint2(best, second);
    __syncthreads();
    if (g != 0 || qi >= nQ) return;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const uint2 k = s_k[i][ql];
        push_key(best, second, k.x);
        push_key(best, second, k.y);
    }
    const size_t o = (size_t)b * q_stride + qi;
    best_idx[o] = (best >> 22) >= 256u ? -1 : (int32_t)(best & 0x3FFFFFu);
    best_dist[o] = (int32_t)(best >> 22);
    second_dist[o] = (int32_t)(second >> 22);
}

// ---------------------------------------------------------------------------
// Brute force on the matrix cores.  Hamming(a, b) = |a| + |b| - 2 <a, b> for bit vectors, and the 256-bit inner
// products of 32 train x 32 query descriptors are one chain of four v_mfma_scale_f32_32x32x64_f8f6f4 with both
// operands e2m1 (FP4): 64 bits of the descriptors per instruction, a bit expanded to a nibble through a 256-entry
// table in LDS.  With the trains as the A operand and the queries as B, lane l of the 32x32 result holds query l & 31
// against 16 trains (rows (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)), so the (best, second) update is lane-local: two
// vector instructions per pair instead of the ~19 of the xor/popcount form.  A wavefront keeps the expanded nibbles of
// kMfmaQT * 32 queries in registers and streams the train tiles past them; key order, and therefore the tie rule
// "smallest distance, then smallest index", is that of k_match_bruteforce.
//
// A set bit becomes the nibble 0b0010 = 1.0 in the trains and 0b1010 = -1.0 in the queries, and the queries' block
// scale is 2^13, so a common bit contributes -2 * 4096.  Which of a lane's 32 nibbles stands for which bit does not
// matter as long as trains and queries expand alike: the instruction pairs equal positions.
//
// The accumulator IS the key.  Row r of a tile starts (C operand, the same in every column) at
//     1.5 * 2^23 + (|b_r| + 256) * 4096 + position of r in the current run of 4096 train descriptors,
// all sums stay integers below 2^24 and so exact in f32, and the bit pattern of the result is
//     0x4B400000 + (|b_r| + 256 - 2 <a, b_r>) * 4096 + position:
// ordered like (distance, index) for one query (|a| is the same for every candidate of a query and is added at the
// end), so the best two are kept by comparing accumulator registers as they come, without a shift or an add per pair.
// After every 4096 train descriptors the two survivors are rewritten as keys on the index in the whole train set.
// (The first matrix form, eight v_mfma_i32_32x32x32_i8 per tile on bits expanded to bytes with a shift-add per pair
// for the key, took 1.5 times as long on long sweeps and 1.2-1.4 times on short ones.)
// ---------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));
constexpr int kMfmaQT = 2;              // query tiles (of 32) per wavefront
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
constexpr int kFp4Waves = 4;                    // one k-step (64 bits) of the train tile per wavefront
constexpr int kFp4RunBits = 12;                 // positions per run: 10 bits of distance field above them fill the mantissa below bit 22
constexpr uint32_t kFp4Base = 0x4B400000u;      // 1.5 * 2^23 as a bit pattern

__device__ __forceinline__ v4i_t expand32_fp4(const uint32_t* __restrict__ lut, uint32_t bits)
{
    v4i_t r;
    r[0] = (int)lut[bits & 0xffu];
    r[1] = (int)lut[(bits >> 8) & 0xffu];
    r[2] = (int)lut[(bits >> 16) & 0xffu];
    r[3] = (int)lut[bits >> 24];
    return r;
}

// c_rows: the start values of the tile's 32 rows (one set of reads serves every accumulator: the first matrix
// instruction of the second accumulator takes the first one's start registers as its C operand)
template <int QT>
__device__ __forceinline__ void fp4_sweep(const v4i_t* __restrict__ a_frag, const float* __restrict__ c_rows, int lane, int half,
                                          const v4i_t (&bq)[QT][4], uint32_t (&best)[QT], uint32_t (&second)[QT])
{
    constexpr int kScaleA = 0x7f7f7f7f;                         // E8M0 2^0
    constexpr int kScaleB = (int)0x8c8c8c8cu;                   // E8M0 2^13: twice the unit of the distance field
    v16f_t acc[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int row0 = 4 * half;
#pragma unroll
        for (int g = 0; g < 4; g++) {                           // register 4 g + i holds row 8 g + 4 half + i
            const float4 c = *reinterpret_cast<const float4*>(c_rows + row0 + 8 * g);
            acc[qt][4 * g] = c.x; acc[qt][4 * g + 1] = c.y; acc[qt][4 * g + 2] = c.z; acc[qt][4 * g + 3] = c.w;
        }
    }
    // the fragment of k-step s + 1 is on its way from LDS while the matrix instructions of step s run
    v4i_t a_cur = a_frag[lane], a_nxt = a_cur;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s < 3) a_nxt = a_frag[(s + 1) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        const v4i_t a = a_cur;
        a_cur = a_nxt;
        const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            const v8i_t b8 = {bq[qt][s][0], bq[qt][s][1], bq[qt][s][2], bq[qt][s][3], 0, 0, 0, 0};
            acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[qt], 4, 4, 0, kScaleA, 0, kScaleB);
        }
    }
    // The best two of the tile's 16 keys and the two kept so far, by triples: the minimum and the median of three keys
    // are all that can matter of them (2 instructions for 3 keys), three such sorted pairs merge into one in 4
    // (minimum and median of the minima, minimum of the medians), two in 3: 23 instructions instead of the 32 of
    // sixteen single updates.  Equal keys (only the "no candidate" pattern repeats) pass through unchanged.
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        uint32_t k[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            // (through a scalar copy: __builtin_bit_cast applied to the element expression itself reads element 0 of
            // the vector whatever the index)
            const float f = acc[qt][e];
            k[e] = __float_as_uint(f);
        }
        uint32_t m[6], d[6];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            m[i] = umin3(k[3 * i], k[3 * i + 1], k[3 * i + 2]);
            d[i] = umed3(k[3 * i], k[3 * i + 1], k[3 * i + 2]);
        }
        m[5] = best[qt];
        d[5] = second[qt];
        push_key(m[5], d[5], k[15]);
        const uint32_t ma = umin3(m[0], m[1], m[2]), da = min(umed3(m[0], m[1], m[2]), umin3(d[0], d[1], d[2]));
        const uint32_t mb = umin3(m[3], m[4], m[5]), db = min(umed3(m[3], m[4], m[5]), umin3(d[3], d[4], d[5]));
        best[qt] = min(ma, mb);
        second[qt] = umin3(max(ma, mb), da, db);
    }
}

// 16 bytes per lane from memory straight into LDS: lane l's bytes land at lds_base + 16 * l.  Written as an assembly
// statement on purpose: for the builtin of this instruction the compiler waits for every such load in flight before
// each later LDS read (it cannot tell the buffer being filled from the one being read), which would put the memory
// latency in front of every tile.  The caller waits for the loads itself (fp4_wait_loads) before the barrier that
// publishes the buffer.  Loads complete in order, so the compiler's own counts for its own loads only get stricter.
__device__ __forceinline__ void fp4_load_to_lds(const v4i_t* __restrict__ src_lane, uint32_t lds_base)
{
    // (m0 is not the compiler's to allocate: it loads it itself right before each instruction of its own that reads
    // it; one wait state between a scalar write of m0 and an LDS load that reads it)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src_lane), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ __forceinline__ void fp4_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Diagnostic builds (make EXTRA=-DORBX_MATCH_PROFILE): shader-clock cycles of one wavefront per part of a group of
// tiles (expansion of the next fragments, wait at the barrier, sweeps), summed over all wavefronts.
#ifdef ORBX_MATCH_PROFILE
__device__ unsigned long long g_match_prof[8];
#define MP_DECL unsigned long long mp_t = __builtin_readcyclecounter(), mp_acc[4] = {0, 0, 0, 0}
#define MP_MARK(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); mp_acc[k] += n_ - mp_t; mp_t = n_; } while (0)
#define MP_FLUSH() do { if (lane == 0) { for (int k_ = 0; k_ < 4; k_++) atomicAdd(&g_match_prof[k_], mp_acc[k_]); atomicAdd(&g_match_prof[4], 1ull); } } while (0)
#else
#define MP_DECL
#define MP_MARK(k)
#define MP_FLUSH()
#endif

// (four wavefronts per SIMD: at the 132 registers the compiler would take, three are resident and the sweep of eight
// frames against one map is 3 % slower)
// kPre: the train fragments and bit counts come expanded from memory (k_fp4_expand_trains wrote them once for all the
// workgroups that sweep the same train set) instead of being expanded here by every workgroup: `t` is then the block
// of fragments, [set][tile][k-step][lane] x 16 bytes, followed by the bit counts [set][tile][row], `tiles` tiles a set.
template <int QT, bool kPre>
__global__ __launch_bounds__(64 * kFp4Waves) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_match_bruteforce_fp4(
    const uint32_t* __restrict__ q, int Q, int q_stride, const int32_t* __restrict__ nq, const uint32_t* __restrict__ t, int T,
    int t_stride, const int32_t* __restrict__ nt, int tsplit, uint32_t* __restrict__ partial, int32_t* __restrict__ best_idx,
    int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist, int tiles, int nsets)
{
    constexpr int NT = 4;                                       // tiles per barrier (1 / 2 / 4: 0.855 / 0.813 / 0.792 ms for 8 x 8000 x 100000)
    __shared__ v4i_t s_a[2][NT][4 * 64];                        // expanded train fragments: [buffer][tile of the group][k-step][lane]
    __shared__ __align__(16) uint32_t s_c[3][32 * NT];          // start value (bit pattern) of every train row: [group mod 3][row]
    __shared__ uint32_t s_lut[256];                             // byte -> eight e2m1 values (bit i -> nibble i = 1.0)
    const int b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
    {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= ((threadIdx.x >> i) & 1u) << (4 * i + 1);
        s_lut[threadIdx.x] = v;
    }
    constexpr uint32_t kRowBase = kFp4Base + (256u << kFp4RunBits);
    // Rows past the end of the sweep's train descriptors (the last tile of a sweep may be short; its fragments repeat
    // the last descriptor) start kFp4Past distance units higher: above every key of a real row and, once the queries'
    // own bit counts are restored, at a distance above 256, which is read as "none".
    constexpr uint32_t kFp4Past = 300u << kFp4RunBits;
    const int nQ = nq ? min(nq[b], q_stride) : Q;
    const int nT = nt ? min(nt[b], T) : T;     // T is the per-batch maximum
    const int chunk = (((nT + tsplit - 1) / tsplit) + 31) & ~31;         // whole tiles per split
    const int j0 = min((int)blockIdx.y * chunk, nT), j1 = min(j0 + chunk, nT);
    // the first group's rows: positions 0.. of the first run
    if (!kPre && threadIdx.x < 32 * NT) s_c[0][threadIdx.x] = kRowBase + threadIdx.x + (j0 + (int)threadIdx.x < j1 ? 0u : kFp4Past);
    __syncthreads();
    const int q0 = (blockIdx.x * kFp4Waves + wave) * (32 * QT);
    if (blockIdx.x * kFp4Waves * (32 * QT) >= nQ) return;          // whole workgroup past the queries
    const bool active = q0 < nQ;                                          // idle wavefronts still help expanding

    // the first tile's train dwords are on their way while the queries are expanded
    const uint32_t* tp = t + (size_t)b * t_stride * 8 + 2 * wave + half;
    uint32_t mine = 0;
    if (!kPre && j0 < j1) mine = tp[8 * (size_t)min(j0 + r, j1 - 1)];
    // expanded form: this lane's 16 bytes of tile i are frag[i * 256]; the row counts of the set follow all fragments
    const size_t set = t_stride ? (size_t)b : 0;
    const v4i_t* frag = reinterpret_cast<const v4i_t*>(t) + set * tiles * 256 + wave * 64 + lane;
    const uint32_t* rowpop = t + (size_t)nsets * tiles * 1024 + set * tiles * 32;
    constexpr int kRunMask = (1 << kFp4RunBits) - 1;
    auto start_value = [&](int jg, uint32_t pop) {            // of row jg + threadIdx.x, for threads below 32 * NT
        return kRowBase + (pop << kFp4RunBits) + (uint32_t)(((jg - j0) & kRunMask) + (int)threadIdx.x) +
               (jg + (int)threadIdx.x < j1 ? 0u : kFp4Past);
    };
    uint32_t pop_next = 0;
    if (kPre) {
        // the first group's fragments and start values; the second group's counts are on their way
#pragma unroll
        for (int k = 0; k < NT; k++)
            if (j0 + 32 * k < j1) s_a[0][k][wave * 64 + lane] = frag[(size_t)(j0 / 32 + k) * 256];
        if (threadIdx.x < 32 * NT) {
            const int row = j0 + (int)threadIdx.x;
            s_c[0][threadIdx.x] = start_value(j0, row < tiles * 32 ? rowpop[row] : 0u);
            const int row2 = row + 32 * NT;
            if (row2 < tiles * 32) pop_next = rowpop[row2];
        }
    }

    // queries: this lane's query of every tile, expanded once (B operand: the 32 bits 64 s + 32 half .. of k-step s)
    v4i_t bq[QT][4];
    int pa[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int qi = min(q0 + 32 * qt + r, nQ - 1);
        const uint4* qp = reinterpret_cast<const uint4*>(q + ((size_t)b * q_stride + qi) * 8);
        const uint4 x = qp[0], y = qp[1];
        pa[qt] = __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w);
        // (selects written out: an array indexed by `half` would live in scratch memory)
        const uint32_t mine4[4] = {half ? x.y : x.x, half ? x.w : x.z, half ? y.y : y.x, half ? y.w : y.z};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bq[qt][s] = expand32_fp4(s_lut, mine4[s]);
            bq[qt][s] = bq[qt][s] | (bq[qt][s] << 2);          // 0b0010 -> 0b1010: the B operand holds 0 / -1.0
        }
    }
    // best two of the current run (accumulator bit patterns) and of the runs before it (keys on the whole train set)
    uint32_t best[QT], second[QT], gbest[QT], gsecond[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) { best[qt] = second[qt] = gbest[qt] = gsecond[qt] = 0xffffffffu; }
    auto fold = [&](int run0) {
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            const uint32_t kk[2] = {best[qt], second[qt]};
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (kk[i] == 0xffffffffu) continue;
                const uint32_t v = kk[i] - kFp4Base;
                push_key(gbest[qt], gsecond[qt], ((v >> kFp4RunBits) << 22) | (uint32_t)(run0 + (int)(v & ((1u << kFp4RunBits) - 1))));
            }
            best[qt] = second[qt] = 0xffffffffu;
        }
    };

    // Train row r of a tile: wavefront w expands k-step w, i.e. dword 2 w + half
    // of the row, and adds that dword's share of |b_r| to the row's start value, which a wavefront set up one group of
    // tiles ahead (three buffers: the group being swept, the group being built, and the one after it being set up).  Runs
    // of 4096 positions start at j0.
    int buf = 0, cb = 0;
    uint32_t mn[NT];
    mn[0] = mine;
    if (!kPre) {
#pragma unroll
        for (int k = 1; k < NT; k++) mn[k] = j0 + 32 * k < j1 ? tp[8 * (size_t)min(j0 + 32 * k + r, j1 - 1)] : 0u;
    }
    MP_DECL;
    for (int j = j0; j < j1; j += 32 * NT, buf ^= 1) {
        MP_MARK(3);
        int cn;
        if (kPre) {
            // Expanded form: while this group is swept, this wavefront's share of the next group's fragments goes from
            // memory straight into the other buffer; the start values of the next group are written at the end (two
            // buffers: every wavefront has passed this group's barrier, so nobody still reads the last group but one).
            cn = cb ^ 1;
            MP_MARK(0);
            fp4_wait_loads();                                               // this group's fragments have landed
            __syncthreads();
            MP_MARK(1);
            const float* c_rows = reinterpret_cast<const float*>(s_c[cb]);
            const int jn = j + 32 * NT;
#pragma unroll
            for (int k = 0; k < NT; k++)
                if (jn + 32 * k < j1)                                       // wave-uniform
                    fp4_load_to_lds(frag + (size_t)(jn / 32 + k) * 256, (uint32_t)(uintptr_t)&s_a[buf ^ 1][k][wave * 64]);
#pragma unroll
            for (int k = 0; k < NT; k++)
                if (active && j + 32 * k < j1) fp4_sweep<QT>(s_a[buf][k], c_rows + 32 * k, lane, half, bq, best, second);
            if (threadIdx.x < 32 * NT) {
                s_c[cn][threadIdx.x] = start_value(jn, pop_next);
                const int row2 = jn + 32 * NT + (int)threadIdx.x;
                pop_next = row2 < tiles * 32 ? rowpop[row2] : 0u;
            }
            if (active && ((jn - j0) & kRunMask) == 0) fold(j0 + ((j - j0) & ~kRunMask));
            MP_MARK(2);
            cb = cn;
            continue;
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            atomicAdd(&s_c[cb][32 * k + r], (uint32_t)__popc(mn[k]) << kFp4RunBits);
            s_a[buf][k][wave * 64 + lane] = expand32_fp4(s_lut, mn[k]);
        }
        cn = cb == 2 ? 0 : cb + 1;
        if (threadIdx.x < 32 * NT) s_c[cn][threadIdx.x] = start_value(j + 32 * NT, 0u);
#pragma unroll
        for (int k = 0; k < NT; k++)
            if (j + 32 * (NT + k) < j1) mn[k] = tp[8 * (size_t)min(j + 32 * (NT + k) + r, j1 - 1)];
        MP_MARK(0);
        __syncthreads();
        MP_MARK(1);
        if (active) {
            const float* c_rows = reinterpret_cast<const float*>(s_c[cb]);
#pragma unroll
            for (int k = 0; k < NT; k++) {
                if (j + 32 * k < j1) fp4_sweep<QT>(s_a[buf][k], c_rows + 32 * k, lane, half, bq, best, second);
            }
            if (((j + 32 * NT - j0) & kRunMask) == 0) fold(j0 + ((j - j0) & ~kRunMask));
        }
        MP_MARK(2);
        cb = cn;
    }
    if (active) MP_FLUSH();
    if (!active) return;
    fold(j0 + ((j1 - 1 - j0) & ~kRunMask));   // (a whole number of runs: nothing left, and the fold of nothing is harmless)

    // the two halves of the wavefront saw different trains of the same query: merge, then restore |a|
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const uint32_t ob = __shfl_xor(gbest[qt], 32, 64), os = __shfl_xor(gsecond[qt], 32, 64);
        push_key(gbest[qt], gsecond[qt], ob);
        push_key(gbest[qt], gsecond[qt], os);
        const int qi = q0 + 32 * qt + r;
        if (half != 0 || qi >= nQ) continue;
        const uint32_t fix = (uint32_t)(pa[qt] - 256) << 22;
        uint32_t kb = gbest[qt] == 0xffffffffu ? kKeyNone : gbest[qt] + fix;
        uint32_t ks = gsecond[qt] == 0xffffffffu ? kKeyNone : gsecond[qt] + fix;
        if ((kb >> 22) >= 256u) kb = kKeyNone;                  // a row past the end, or all 256 bits different
        if ((ks >> 22) >= 256u) ks = kKeyNone;
        const size_t o = (size_t)b * q_stride + qi;
        if (tsplit == 1) {
            best_idx[o] = (kb >> 22) >= 256u ? -1 : (int32_t)(kb & 0x3FFFFFu);
            best_dist[o] = (int32_t)(kb >> 22);
            second_dist[o] = (int32_t)(ks >> 22);
        } else {
            uint32_t* p = partial + (((size_t)b * tsplit + blockIdx.y) * q_stride + qi) * 2;
            p[0] = kb;
            p[1] = ks;
        }
    }
}

#ifdef ORBX_MATCH_PROFILE
}  // namespace orbx
extern "C" int orbx_debug_match_profile(unsigned long long* out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(orbx::g_match_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(orbx::g_match_prof), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
namespace orbx {
#endif

// Train descriptors -> the matrix form of k_match_bruteforce_fp4<.., true>: one workgroup per tile of 32 descriptors,
// wavefront w expands k-step w (dword 2 w + half of row lane & 31) exactly as the sweep itself would, the rows past
// row T - 1 of the last tile repeat that row; the rows' bit counts go behind all fragments.
constexpr int kExpandTiles = 4;         // tiles per workgroup of k_fp4_expand_trains (one table set-up and one barrier for all)
__global__ __launch_bounds__(64 * kFp4Waves) void k_fp4_expand_trains(const uint32_t* __restrict__ t, int T, int t_stride,
                                                                      uint32_t* __restrict__ out, int tiles, int nsets)
{
    __shared__ uint32_t s_lut[256];
    __shared__ uint32_t s_pop[kExpandTiles][32];
    const int set = blockIdx.y, tile0 = blockIdx.x * kExpandTiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
    {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= ((threadIdx.x >> i) & 1u) << (4 * i + 1);
        s_lut[threadIdx.x] = v;
    }
    if (threadIdx.x < 32 * kExpandTiles) (&s_pop[0][0])[threadIdx.x] = 0u;
    // all T rows of the set, whatever the batches' own counts (a set shared by the whole batch has one count per batch;
    // the sweep ignores the rows past its batch's count)
    uint32_t dw[kExpandTiles];
#pragma unroll
    for (int k = 0; k < kExpandTiles; k++)
        dw[k] = tile0 + k < tiles ? t[((size_t)set * t_stride + (size_t)min(32 * (tile0 + k) + r, T - 1)) * 8 + 2 * wave + half] : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kExpandTiles; k++) {
        if (tile0 + k >= tiles) break;
        reinterpret_cast<v4i_t*>(out)[((size_t)set * tiles + tile0 + k) * 256 + wave * 64 + lane] = expand32_fp4(s_lut, dw[k]);
        atomicAdd(&s_pop[k][r], (uint32_t)__popc(dw[k]));
    }
    __syncthreads();
    const int k = threadIdx.x >> 5;
    if (k < kExpandTiles && tile0 + k < tiles)
        out[(size_t)nsets * tiles * 1024 + ((size_t)set * tiles + tile0 + k) * 32 + (threadIdx.x & 31)] = s_pop[k][threadIdx.x & 31];
}

// The environment switches of the brute-force search are read once per process; only a process started with
// ORBX_SWITCHES_PER_CALL set (the tests, which compare the forms) reads them again at every call.
// ORBX_MATCH_FORM=scalar: the xor / popcount sweep for every size.  ORBX_MATCH_EXPAND=0 / 1: never / whenever possible
// expand the train set once in memory instead of in every workgroup.
static bool match_want_scalar()
{
    auto read = [] { const char* e = getenv("ORBX_MATCH_FORM"); return e && e[0] == 's'; };
    static const bool per_call = getenv("ORBX_SWITCHES_PER_CALL") != nullptr;
    static const bool at_start = read();
    return per_call ? read() : at_start;
}
static int match_expand_mode()
{
    auto read = [] { const char* e = getenv("ORBX_MATCH_EXPAND"); return e ? atoi(e) : -1; };
    static const bool per_call = getenv("ORBX_SWITCHES_PER_CALL") != nullptr;
    static const int at_start = read();
    return per_call ? read() : at_start;
}

// Bytes of the expanded train sets (fragments + bit counts) when the search is to use them, 0 when every workgroup
// expands for itself: the expanded form pays when many workgroups sweep the same set (each tile of 256 queries is one).
size_t match_expand_bytes(int q_stride, int T, int t_stride, int nbatch)
{
    if (match_want_scalar() || T < 64 || q_stride < 32) return 0;
    const int mode = match_expand_mode();
    if (mode == 0) return 0;
    const long sweeps_per_set = (long)((q_stride + 255) / 256) * (t_stride ? 1 : nbatch);
    // (measured: 32 sweeps of one set of 100 000 lose 4 us to the extra launch, 256 sweeps gain 24 us)
    if (mode < 0 && (sweeps_per_set < 64 || T < 4096)) return 0;
    const size_t tiles = (size_t)(T + 31) / 32, nsets = t_stride ? (size_t)nbatch : 1;
    return nsets * tiles * (4096 + 128);
}

int match_bruteforce_tsplit(int Q, int T, int nbatch)
{
    // chunks of >= 256 train descriptors
    long qblocks = (long)((Q + 255) / 256) * nbatch;
    // Workgroups wanted: one round on the chip's 1024 workgroup slots for small problems (start and end of a workgroup
    // weigh as much as its sweep), two or four rounds for large ones (the rounds cover each other's starts and ends).
    // Measured, 1024 / 2048 / 4096: 8 x 8000 x 100000 0.817 / 0.782 / 0.775 ms, 8000 x 100000 0.110 / 0.110 / 0.114,
    // 128 x 1000 x 1000 0.0246 / 0.0264 / 0.0262, 1000 x 100000 0.0257 / 0.0290 / 0.0290.
    const long tiles = qblocks * ((T + 31) / 32);            // tiles of 256 queries x 32 train descriptors
    const long target = tiles >= 200 * 1024 ? 4096 : (tiles >= 60 * 1024 ? 2048 : 1024);
    long want = (target + qblocks - 1) / qblocks;
    long maxsplit = T / 256;
    if (maxsplit < 1) maxsplit = 1;
    long s = want < maxsplit ? want : maxsplit;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

void launch_match_bruteforce(const uint32_t* d_q, int Q, int q_stride, const int32_t* d_nq,
                             const uint32_t* d_t, int T, int t_stride, const int32_t* d_nt, int nbatch,
                             int tsplit, uint32_t* d_partial, uint32_t* d_expand, int32_t* d_best_idx, int32_t* d_best_dist,
                             int32_t* d_second_dist, hipStream_t s)
{
    // the matrix-core sweep needs whole tiles to pay: tiny problems keep the scalar form; d_expand: room for
    // match_expand_bytes() when that is not 0
    if (!match_want_scalar() && T >= 64 && q_stride >= 32) {
        const int per_wg = kFp4Waves * 32 * kMfmaQT;
        dim3 grid((q_stride + per_wg - 1) / per_wg, tsplit, nbatch);
        const int tiles = (T + 31) / 32, nsets = t_stride ? nbatch : 1;
        if (d_expand) {
            hipLaunchKernelGGL(k_fp4_expand_trains, dim3((tiles + kExpandTiles - 1) / kExpandTiles, nsets), dim3(64 * kFp4Waves), 0, s, d_t, T, t_stride,
                               d_expand, tiles, nsets);
            hipLaunchKernelGGL((k_match_bruteforce_fp4<kMfmaQT, true>), grid, dim3(64 * kFp4Waves), 0, s, d_q, Q, q_stride, d_nq,
                               d_expand, T, t_stride, d_nt, tsplit, d_partial, d_best_idx, d_best_dist, d_second_dist, tiles,
                               nsets);
        } else {
            hipLaunchKernelGGL((k_match_bruteforce_fp4<kMfmaQT, false>), grid, dim3(64 * kFp4Waves), 0, s, d_q, Q, q_stride, d_nq,
                               d_t, T, t_stride, d_nt, tsplit, d_partial, d_best_idx, d_best_dist, d_second_dist, tiles, nsets);
        }
    } else {
        dim3 grid((q_stride + 255) / 256, tsplit, nbatch);
        hipLaunchKernelGGL(k_match_bruteforce, grid, dim3(256), 0, s, d_q, Q, q_stride, d_nq, d_t, T, t_stride, d_nt,
                           tsplit, d_partial, d_best_idx, d_best_dist, d_second_dist);
    }
    if (tsplit > 1) {
        if (tsplit >= 16) {
            dim3 g2((q_stride + 15) / 16, nbatch);
            hipLaunchKernelGGL(k_match_merge_wide, g2, dim3(256), 0, s, d_partial, Q, q_stride, d_nq, tsplit, d_best_idx,
                               d_best_dist, d_second_dist);
        } else {
            dim3 g2((q_stride + 255) / 256, nbatch);
            hipLaunchKernelGGL(k_match_merge, g2, dim3(256), 0, s, d_partial, Q, q_stride, d_nq, tsplit, d_best_idx,
                               d_best_dist, d_second_dist);
        }
    }
}

// ---------------------------------------------------------------------------
// windowed search over CSR candidate lists: thread = one query.  The key's
// position field is the position in the query's list, so that the first
// candidate in list order wins ties (src/ORBmatcher.cc:212).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_windowed(const uint32_t* __restrict__ q, int Q,
                                                        const uint32_t* __restrict__ t,
                                                        const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                        const int32_t* __restrict__ toct,
                                                        int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                        int32_t* __restrict__ second_dist,
                                                        int32_t* __restrict__ best_lvl, int32_t* __restrict__ second_lvl)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= Q) return;
    uint32_t qd[8];
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
    uint4 a = qp[0], c = qp[1];
    qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w;
    qd[4] = c.x; qd[5] = c.y; qd[6] = c.z; qd[7] = c.w;
    const int k0 = off[qi], k1 = off[qi + 1];
    uint32_t best = kKeyNone, second = kKeyNone;
    for (int k = k0; k < k1; k++) {
        const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)idx[k] * 8);
        const int d = hamming256(qd, tp[0], tp[1]);
        push_key(best, second, ((uint32_t)d << 22) | (uint32_t)(k - k0));
    }
    const int bi = (best >> 22) >= 256u ? -1 : idx[k0 + (int)(best & 0x3FFFFFu)];
    const int si = (second >> 22) >= 256u ? -1 : idx[k0 + (int)(second & 0x3FFFFFu)];
    best_idx[qi] = bi;
    best_dist[qi] = (int32_t)(best >> 22);
    second_dist[qi] = (int32_t)(second >> 22);
    if (best_lvl) best_lvl[qi] = bi < 0 ? -1 : (toct ? toct[bi] : 0);
    if (second_lvl) second_lvl[qi] = si < 0 ? -1 : (toct ? toct[si] : 0);
}

void launch_match_windowed(const uint32_t* d_q, int Q, const uint32_t* d_t, const int32_t* d_off,
                           const int32_t* d_idx, const int32_t* d_toct, int32_t* d_best_idx, int32_t* d_best_dist,
                           int32_t* d_second_dist, int32_t* d_best_lvl, int32_t* d_second_lvl, hipStream_t s)
{
    if (Q <= 0) return;
    hipLaunchKernelGGL(k_match_windowed, dim3((Q + 255) / 256), dim3(256), 0, s, d_q, Q, d_t, d_off, d_idx, d_toct,
                       d_best_idx, d_best_dist, d_second_dist, d_best_lvl, d_second_lvl);
}

__global__ __launch_bounds__(256) void k_hamming_csr(const uint32_t* __restrict__ q, int Q,
                                                     const uint32_t* __restrict__ t,
                                                     const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                     uint16_t* __restrict__ out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= Q) return;
    uint32_t qd[8];
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
    uint4 a = qp[0], c = qp[1];
    qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w;
    qd[4] = c.x; qd[5] = c.y; qd[6] = c.z; qd[7] = c.w;
    for (int k = off[qi]; k < off[qi + 1]; k++) {
        const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)idx[k] * 8);
        out[k] = (uint16_t)hamming256(qd, tp[0], tp[1]);
    }
}

void launch_hamming_csr(const uint32_t* d_q, int Q, const uint32_t* d_t, const int32_t* d_off,
                        const int32_t* d_idx, uint16_t* d_out, hipStream_t s)
{
    if (Q <= 0) return;
    hipLaunchKernelGGL(k_hamming_csr, dim3((Q + 255) / 256), dim3(256), 0, s, d_q, Q, d_t, d_off, d_idx, d_out);
}

// ---------------------------------------------------------------------------
// Scratch for the entry point that allocates nothing (orbx_match_bruteforce_device_static): the
// per-split (best, second) keys of k_match_bruteforce[_fp4] that k_match_merge combines.  Every GPU the
// library runs on holds its own copy of this array, at its own address, until the process ends.
// The name below is only a host-side handle: kernels get the address hipGetSymbolAddress reports
// for the current device.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[kMatchStaticScratchBytes];

namespace {
std::mutex g_scratch_mutex;
constexpr int kMaxScratchDevices = 64;
void* g_scratch_ptr[kMaxScratchDevices] = {};
}

// `device` must be the current device.  Returns the device address of this GPU's scratch array
// (looked up once per device), or nullptr with *err set.
uint32_t* match_static_scratch(int device, hipError_t* err)
{
    *err = hipSuccess;
    if (device < 0 || device >= kMaxScratchDevices) { *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { *err = e != hipSuccess ? e : hipErrorInvalidSymbol; return nullptr; }
        g_scratch_ptr[device] = p;
    }
    return static_cast<uint32_t*>(g_scratch_ptr[device]);
}

// ---------------------------------------------------------------------------
// Stereo: Frame::PrepareStereoCandidates (include/Frame.h:217-251) and the per-keypoint body of
// Frame::ComputeStereoMatches_Undistorted (src/Frame.cc:1302-1395), which is also the body of
// ComputeStereoMatch_OnePoint (:1189-1257) and the Hamming scan of ComputeStereoMatches (:1023-1068).
// A right keypoint is a candidate of left keypoint iL iff int(vL) lies in its row band
// [max(0, floor(y-r)), min(nRows-1, ceil(y+r))], r = 2*scaleFactor[octave]; the reference visits a
// row's candidates in ascending iR and keeps the first minimum below TH_HIGH, i.e. the minimum of
// (distance, iR), so the visiting order is free: one wavefront per left keypoint, its lanes
// stride over the right keypoints, test band / octave / disparity window, and the wave minimum of
// distance << 22 | iR is taken with DPP moves.  The lanes only visit the right keypoints of the rows around
// the left keypoint's (k_stereo_rows orders them by row first).  Work of a pair spreads over the whole
// device and there is no limit on the number of keypoints.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    // lanes without a source (and rows masked out) keep the identity
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast15 into rows 1, 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct StereoArgs {
    const orbx_keypoint *kl, *kr;
    const uint32_t *dl, *dr;
    const int32_t *nl, *nr;
    const uint8_t* active;          // per left keypoint, or NULL = all
    const float *min_d, *max_d;     // per left keypoint disparity window, or NULL = [0, mbf / mb)
    int32_t* listed;                // mvDistIdx as an array: accepted distance per left keypoint, -1 = not listed
    int32_t *best_idx, *best_dist;
    float *u_right, *depth;
    int32_t* nmatched;
    int stride, nrows, th_orb_dist;
    float mbf, mb, min_x;
    const int32_t* row_start;       // per pair nrows + 1 offsets into sorted: right keypoints ordered by clamp(floor(y))
    const int32_t* sorted;          // per pair `stride` right keypoint indices
    int rband;                      // a band never reaches further than this many rows from floor(y)
    int classic;                    // Frame::ComputeStereoMatches: row test of :1031-1033, then the SAD refinement
    int32_t* sad;                   // classic: accepted best SAD per left keypoint of a pair (`sad_stride` ints per pair), -1 = none
    int sad_stride;
    int init_outputs;               // u_right = depth = listed = -1 for every left keypoint first
    int median_cut;
};

// The reference's row table (mvRowIndices) lists, per image row, the right keypoints whose band covers it; here
// the right keypoints of a pair are only ordered by their own row, clamp(floor(y)) -- a counting sort in LDS --
// and a left keypoint then looks at the rows within `rband` of its own: every right keypoint whose band
// covers the left keypoint's row is among them (|row - y| <= r + 1 <= rband), and the exact band test
// picks those.  One workgroup per pair; dynamic LDS: 2 * (nrows + 1) ints.
__global__ __launch_bounds__(256) void k_stereo_rows(const orbx_keypoint* __restrict__ kr, const int32_t* __restrict__ nr_arr,
                                                     int stride, int nrows, int32_t* __restrict__ row_start,
                                                     int32_t* __restrict__ sorted)
{
    extern __shared__ int s_rows[];          // [0, nrows]: counts, then cursors; [nrows + 1, 2 nrows + 1]: offsets
    __shared__ int s_part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nr = min(nr_arr[b], stride);
    kr += (size_t)b * stride;
    row_start += (size_t)b * (nrows + 1);
    sorted += (size_t)b * stride;
    int* cnt = s_rows;
    int* offs = s_rows + nrows + 1;
    for (int i = tid; i <= nrows; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        const int r = min(max((int)floorf(kr[i].y), 0), nrows - 1);
        atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: a contiguous chunk per thread, then the chunk sums
    const int chunk = (nrows + 256) / 256;
    const int c0 = min(tid * chunk, nrows + 1), c1 = min(c0 + chunk, nrows + 1);
    int sum = 0;
    for (int i = c0; i < c1; i++) sum += cnt[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int t = s_part[i]; s_part[i] = run; run += t; }
    }
    __syncthreads();
    int run = s_part[tid];
    for (int i = c0; i < c1; i++) {
        const int t = cnt[i];
        offs[i] = run;
        row_start[i] = run;
        run += t;
    }
    __syncthreads();
    for (int i = tid; i <= nrows; i += 256) cnt[i] = offs[i];
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        const int r = min(max((int)floorf(kr[i].y), 0), nrows - 1);
        sorted[atomicAdd(&cnt[r], 1)] = i;
    }
}

// Classic stereo refinement of one left keypoint, Frame::ComputeStereoMatches src/Frame.cc:1070-1155, on one
// wavefront: an 11x11 SAD window (each patch minus its centre pixel) slides +-5 px over the right pyramid
// level kpL.octave; work item (row dy, shift incR) sums |a-b| over the 11 columns, 11 lanes then add the rows
// (exact integers); parabola sub-pixel fit and disparity gating.  Returns the best SAD of an accepted match
// (depth / u_right written by lane 0), -1 otherwise.  `part`: 128 ints of LDS owned by this wavefront.
struct SadPlanes {
    const uint8_t *pyrL, *pyrR;     // pyramids of the left and right extractor contexts (frame 0 of the launch)
    ScaleTab inv_sc;
};
__device__ __forceinline__ int sad_refine(const orbx_keypoint& kpL, float uR0, const uint8_t* __restrict__ pyrL,
                                          const uint8_t* __restrict__ pyrR, const FrameGeom& g, const ScaleTab& sc,
                                          const ScaleTab& inv_sc, float mbf, float mb, float* u_right, float* depth,
                                          int* part, int lane)
{
    const int lvl = min(max(kpL.octave, 0), g.nlevels - 1);
    const LevelGeom& L = g.lv[lvl];
    const float minD = 0.0f, maxD = __fdiv_rn(mbf, mb);
    const float sf = inv_sc.v[lvl];
    const float scaleduL = roundf(__fmul_rn(kpL.x, sf)), scaledvL = roundf(__fmul_rn(kpL.y, sf));
    const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
    const float endu = __fadd_rn(scaleduR0, 11.0f);
    bool ok = !(scaleduR0 < 0.0f || endu >= (float)L.w);
    const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
    // keep every access inside the stored plane even for keypoints the extractor would never produce
    ok = ok && cy >= 5 - kPadY && cy < L.h + kPadY - 5 && cxL >= 5 - kPadX && cxL < L.w + kPadX - 5 &&
         cxR0 >= 10 - kPadX && cxR0 < L.w + kPadX - 10;
    if (!ok) return -1;                                 // wave-uniform
    const uint8_t* imL = pyrL + L.plane_off + (size_t)(kPadY + cy) * L.pitch + kPadX + cxL;
    const uint8_t* imR = pyrR + L.plane_off + (size_t)(kPadY + cy) * L.pitch + kPadX + cxR0;
    const int cL = imL[0];
    for (int it = lane; it < 121; it += 64) {
        const int dy = it / 11 - 5, inc = it % 11 - 5;
        const int cR = imR[inc];
        const uint8_t* pl = imL + dy * L.pitch;
        const uint8_t* pr = imR + dy * L.pitch + inc;
        int acc = 0;
#pragma unroll
        for (int dx = -5; dx <= 5; dx++) acc += abs(((int)pl[dx] - cL) - ((int)pr[dx] - cR));
        part[it] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int dist = 0x7fffffff;
    if (lane < 11) {
        dist = 0;
        for (int r = 0; r < 11; r++) dist += part[r * 11 + lane];
    }
    // first minimum over the 11 shifts (strict '<' in the reference: lowest incR wins ties)
    const int key = lane < 11 ? ((dist << 4) | lane) : 0x7fffffff;
    int kmin = key;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) kmin = min(kmin, __shfl_xor(kmin, d, 16));
    kmin = __builtin_amdgcn_readfirstlane(kmin);
    const int bestSlot = kmin & 15, bestSad = kmin >> 4;
    const float d1 = (float)__shfl(dist, max(bestSlot - 1, 0), 64);
    const float d2 = (float)__shfl(dist, bestSlot, 64);
    const float d3 = (float)__shfl(dist, min(bestSlot + 1, 10), 64);
    int accepted = -1;
    if (bestSlot != 0 && bestSlot != 10) {
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (!(deltaR < -1.0f || deltaR > 1.0f)) {
            float bestuR = __fmul_rn(sc.v[lvl], __fadd_rn(__fadd_rn(scaleduR0, (float)(bestSlot - 5)), deltaR));
            float disparity = __fsub_rn(kpL.x, bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = (float)((double)kpL.x - 0.01);
                }
                if (lane == 0) {
                    *depth = __fdiv_rn(mbf, disparity);
                    *u_right = bestuR;
                }
                accepted = bestSad;
            }
        }
    }
    return accepted;
}

template <bool CLASSIC>
__global__ __launch_bounds__(256) void k_stereo_scan(StereoArgs a, ScaleTab sc, FrameGeom g, SadPlanes sp)
{
    __shared__ int s_part[CLASSIC ? 4 : 1][CLASSIC ? 128 : 1];
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int iL = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nl = min(a.nl[b], a.stride);
    if (iL >= nl) return;
    const size_t o = (size_t)b * a.stride + iL;
    if (a.init_outputs && lane == 0) {
        a.u_right[o] = -1.0f;
        a.depth[o] = -1.0f;
        if (a.listed) a.listed[o] = -1;
    }
    if (a.active && !a.active[o]) return;
    const orbx_keypoint k = a.kl[o];
    const float minD = a.min_d ? a.min_d[o] : 0.0f;
    const float maxD = a.max_d ? a.max_d[o] : __fdiv_rn(a.mbf, a.mb);
    const float vL = k.y, uL = k.x;
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    const int row = (int)vL;
    const bool row_ok = a.classic ? (row >= 0 && row < a.nrows) : !(vL < 0 || vL > (float)(a.nrows - 1));
    uint32_t key = 0xffffffffu;
    bool any = false;
    if (row_ok && !(maxU < a.min_x)) {
        uint32_t qd[8];
        const uint4* qp = reinterpret_cast<const uint4*>(a.dl + o * 8);
        const uint4 q0 = qp[0], q1 = qp[1];
        qd[0] = q0.x; qd[1] = q0.y; qd[2] = q0.z; qd[3] = q0.w;
        qd[4] = q1.x; qd[5] = q1.y; qd[6] = q1.z; qd[7] = q1.w;
        const orbx_keypoint* kr = a.kr + (size_t)b * a.stride;
        const uint32_t* dr = a.dr + (size_t)b * a.stride * 8;
        const float last_row = (float)(a.nrows - 1);
        const int32_t* rs = a.row_start + (size_t)b * (a.nrows + 1);
        const int32_t* sx = a.sorted + (size_t)b * a.stride;
        const int k0 = rs[max(row - a.rband, 0)], k1 = rs[min(row + a.rband + 1, a.nrows)];
        for (int kk = k0 + lane; kk < k1; kk += 64) {
            const int iR = sx[kk];
            const float yR = kr[iR].y, uR = kr[iR].x;
            const int oR = kr[iR].octave;
            const float r = __fmul_rn(2.0f, sc.v[min(max(oR, 0), kMaxLevels - 1)]);
            const int hi = (int)fminf(last_row, ceilf(__fadd_rn(yR, r)));
            const int lo = (int)fmaxf(0.0f, floorf(__fsub_rn(yR, r)));
            if (row < lo || row > hi) continue;
            any = true;
            if (oR < k.octave - 1 || oR > k.octave + 1) continue;
            if (uR >= minU && uR <= maxU) {
                const uint4* tp = reinterpret_cast<const uint4*>(dr + (size_t)iR * 8);
                const int d = hamming256(qd, tp[0], tp[1]);
                key = min(key, ((uint32_t)d << 22) | (uint32_t)iR);
            }
        }
    }
    any = __ballot(any) != 0ull;
    key = wave_min_u32(key);
    const bool found = (key >> 22) < 100u;          // bestDist starts at TH_HIGH, strict '<'
    const int bdist = found ? (int)(key >> 22) : 100;
    const int bidx = found ? (int)(key & 0x3FFFFFu) : 0;
    if (CLASSIC) {
        // the refinement of this keypoint follows on the same wavefront (src/Frame.cc:1070: bestDist < thOrbDist)
        int sad = -1;
        if (any && bdist < a.th_orb_dist)
            sad = sad_refine(k, a.kr[(size_t)b * a.stride + bidx].x, sp.pyrL + (size_t)b * g.pyr_frame_bytes,
                             sp.pyrR + (size_t)b * g.pyr_frame_bytes, g, sc, sp.inv_sc, a.mbf, a.mb, a.u_right + o, a.depth + o,
                             s_part[threadIdx.x >> 6], lane);
        if (lane == 0) a.sad[(size_t)b * a.sad_stride + iL] = sad;
    }
    if (lane != 0) return;
    if (!any) {                                     // left before the scan: 'continue' / 'return false'
        a.best_idx[o] = -1;
        a.best_dist[o] = 100;
        return;
    }
    a.best_idx[o] = bidx;
    a.best_dist[o] = bdist;
    if (!CLASSIC && bdist < a.th_orb_dist) {
        float bestuR = a.kr[(size_t)b * a.stride + bidx].x;
        float disparity = __fsub_rn(uL, bestuR);
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01f;
                bestuR = (float)((double)uL - 0.01);
            }
            a.depth[o] = __fdiv_rn(a.mbf, disparity);
            a.u_right[o] = bestuR;
            if (a.listed) a.listed[o] = bdist;
        }
    }
}

// nmatched of src/Frame.cc:1397 (keypoints of this call that reached the scan) and the cut at
// 2.1 x the median of all listed distances (:1400-1419).  One workgroup per pair.
__global__ __launch_bounds__(256) void k_stereo_finish(StereoArgs a)
{
    __shared__ int s_hist[257];
    __shared__ int s_scanned, s_nv, s_removed, s_median;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nl = min(a.nl[b], a.stride);
    const size_t base = (size_t)b * a.stride;
    for (int i = tid; i < 257; i += 256) s_hist[i] = 0;
    if (tid == 0) { s_scanned = 0; s_nv = 0; s_removed = 0; s_median = -1; }
    __syncthreads();
    int scanned = 0, nv = 0;
    for (int iL = tid; iL < nl; iL += 256) {
        const bool act = !a.active || a.active[base + iL];
        if (act && a.best_idx[base + iL] >= 0) scanned++;
        if (a.median_cut && a.listed) {
            const int d = a.listed[base + iL];
            if (d >= 0) { atomicAdd(&s_hist[min(d, 256)], 1); nv++; }
        }
    }
    if (scanned) atomicAdd(&s_scanned, scanned);
    if (nv) atomicAdd(&s_nv, nv);
    __syncthreads();
    if (tid == 0 && s_nv > 0) {          // element nv/2 of the sorted distances
        const int target = s_nv / 2;
        int cum = 0;
        for (int d = 0; d < 257; d++) {
            cum += s_hist[d];
            if (cum > target) { s_median = d; break; }
        }
    }
    __syncthreads();
    if (s_median >= 0) {
        const float thDist = __fmul_rn(1.5f * 1.4f, (float)s_median);
        int removed = 0;
        for (int iL = tid; iL < nl; iL += 256) {
            const int d = a.listed[base + iL];
            if (d >= 0 && !((float)d < thDist)) {
                a.u_right[base + iL] = -1.0f;
                a.depth[base + iL] = -1.0f;
                removed++;
            }
        }
        if (removed) atomicAdd(&s_removed, removed);
    }
    __syncthreads();
    if (tid == 0) a.nmatched[b] = s_scanned - s_removed;
}

// ---------------------------------------------------------------------------
// Tail of Frame::ComputeStereoMatches, src/Frame.cc:1157-1172: the cut at 2.1 x the median SAD of the accepted
// matches.  One workgroup per stereo pair over the per-keypoint SADs the scan launch left (-1 = no accepted
// match).  The median is element nacc / 2 of the SADs in ascending order (:1159-1160), found by two 256-bin
// histograms (an SAD is below 2^16); the number of keypoints is not limited.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_sad_finish(const int32_t* __restrict__ nl_arr, int stride,
                                                           const int32_t* __restrict__ sad_all, int sad_stride,
                                                           float* __restrict__ u_right, float* __restrict__ depth,
                                                           int32_t* __restrict__ nmatched)
{
    __shared__ int s_hist[256];
    __shared__ int s_nacc, s_median, s_bin, s_rank;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nl = min(nl_arr[b], stride);
    const int32_t* sad = sad_all + (size_t)b * sad_stride;
    u_right += (size_t)b * stride; depth += (size_t)b * stride;
    if (tid == 0) s_nacc = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 256) mine += sad[i] >= 0;
    if (mine) atomicAdd(&s_nacc, mine);
    __syncthreads();
    const int nacc = s_nacc;
    if (tid == 0) { s_median = -1; s_bin = 0; s_rank = nacc / 2; }
    for (int pass = 0; pass < 2 && nacc > 0; pass++) {
        s_hist[tid] = 0;
        __syncthreads();
        const int bin = s_bin;
        for (int i = tid; i < nl; i += 256) {
            const int v = sad[i];
            if (v < 0) continue;
            const int di = min(v, 0xffff);
            if (pass == 0) atomicAdd(&s_hist[di >> 8], 1);
            else if ((di >> 8) == bin) atomicAdd(&s_hist[di & 255], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int cum = 0, k = 0;
            for (; k < 255; k++) {
                if (cum + s_hist[k] > s_rank) break;
                cum += s_hist[k];
            }
            if (pass == 0) { s_bin = k; s_rank -= cum; }
            else s_median = (bin << 8) | k;
        }
        __syncthreads();
    }
    __syncthreads();
    int removed = 0;
    if (nacc > 0) {
        const float thDist = __fmul_rn(1.5f * 1.4f, (float)s_median);
        for (int i = tid; i < nl; i += 256) {
            const int v = sad[i];
            if (v >= 0 && !((float)v < thDist)) {
                u_right[i] = -1.0f;
                depth[i] = -1.0f;
                removed++;
            }
        }
    }
    if (removed) atomicSub(&s_nacc, removed);
    __syncthreads();
    if (tid == 0) nmatched[b] = s_nacc;
}

// rows a band can reach from floor(y): r = 2 * scaleFactor[octave] (include/Frame.h:240), floor / ceil add one
static int stereo_rband(const float* scale8)
{
    float m = 1.0f;
    for (int i = 0; i < kMaxLevels; i++) m = scale8[i] > m ? scale8[i] : m;
    return (int)ceilf(2.0f * m) + 1;
}

// ints of scratch the row ordering of one launch needs: per pair nrows + 1 offsets and `stride` indices
size_t stereo_rows_ints(int nbatch, int stride, int nrows) { return (size_t)nbatch * ((size_t)nrows + 1 + (size_t)stride); }
// ints of scratch of the classic flow: the row ordering, then the three lists of the SAD refinement per pair
size_t stereo_classic_ints(int nbatch, int stride, int nrows) { return stereo_rows_ints(nbatch, stride, nrows) + (size_t)nbatch * 3 * (size_t)stride; }

// k_stereo_rows keeps two ints per image row in dynamic LDS: beyond 64 KB in all (rows above about 8000) the kernel
// has to be told, once per process and device (the entry points accept up to kStereoMaxRows rows)
static void stereo_rows_allow_lds(int nrows)
{
    if ((size_t)(2 * (nrows + 1)) * 4 + 1024 <= 64 * 1024) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (2 * (kStereoMaxRows + 1)) * 4);
    done[dev] = true;
}

void launch_stereo_match(const StereoLaunch& L, const float* scale8, hipStream_t s)
{
    stereo_rows_allow_lds(L.nrows);
    ScaleTab sc;
    for (int i = 0; i < kMaxLevels; i++) sc.v[i] = scale8[i];
    StereoArgs a;
    a.kl = L.d_kl; a.kr = L.d_kr; a.dl = L.d_dl; a.dr = L.d_dr; a.nl = L.d_nl; a.nr = L.d_nr;
    a.active = L.d_active; a.min_d = L.d_min_d; a.max_d = L.d_max_d; a.listed = L.d_listed;
    a.best_idx = L.d_best_idx; a.best_dist = L.d_best_dist; a.u_right = L.d_u_right; a.depth = L.d_depth;
    a.nmatched = L.d_nmatched;
    a.stride = L.stride; a.nrows = L.nrows; a.th_orb_dist = L.th_orb_dist;
    a.mbf = L.mbf; a.mb = L.mb; a.min_x = L.min_x;
    a.classic = 0; a.init_outputs = L.init_outputs; a.median_cut = L.median_cut;
    a.sad = nullptr; a.sad_stride = 0;
    a.row_start = L.d_rows; a.sorted = L.d_rows + (size_t)L.nbatch * (L.nrows + 1); a.rband = stereo_rband(scale8);
    hipLaunchKernelGGL(k_stereo_rows, dim3(L.nbatch), dim3(256), (size_t)(2 * (L.nrows + 1)) * 4, s, L.d_kr, L.d_nr, L.stride,
                       L.nrows, L.d_rows, L.d_rows + (size_t)L.nbatch * (L.nrows + 1));
    hipLaunchKernelGGL(k_stereo_scan<false>, dim3((L.stride + 3) / 4, L.nbatch), dim3(256), 0, s, a, sc, FrameGeom{}, SadPlanes{});
    hipLaunchKernelGGL(k_stereo_finish, dim3(L.nbatch), dim3(256), 0, s, a);
}

void launch_stereo_classic(const orbx_keypoint* d_kl, const uint32_t* d_dl, const int32_t* d_nl,
                           const orbx_keypoint* d_kr, const uint32_t* d_dr, const int32_t* d_nr,
                           int nbatch, int stride, const uint8_t* pyrL, const uint8_t* pyrR, const FrameGeom& g,
                           const float* scale8, const float* inv_scale8, float mbf, float mb,
                           int32_t* d_best_idx, int32_t* d_best_dist, float* d_u_right, float* d_depth,
                           int32_t* d_nmatched, int32_t* d_rows, hipStream_t s)
{
    ScaleTab sc, isc;
    for (int i = 0; i < kMaxLevels; i++) { sc.v[i] = scale8[i]; isc.v[i] = inv_scale8[i]; }
    const int nrows = g.lv[0].h;
    stereo_rows_allow_lds(nrows);
    StereoArgs a;
    a.kl = d_kl; a.kr = d_kr; a.dl = d_dl; a.dr = d_dr; a.nl = d_nl; a.nr = d_nr;
    a.active = nullptr; a.min_d = nullptr; a.max_d = nullptr; a.listed = nullptr;
    a.best_idx = d_best_idx; a.best_dist = d_best_dist; a.u_right = d_u_right; a.depth = d_depth;
    a.nmatched = d_nmatched;
    a.stride = stride; a.nrows = nrows; a.th_orb_dist = 75;
    a.mbf = mbf; a.mb = mb; a.min_x = 0.0f;
    a.classic = 1; a.init_outputs = 1; a.median_cut = 0;
    // the per-keypoint SADs take the first `stride` ints of each pair's share of the refinement scratch
    a.sad = d_rows + stereo_rows_ints(nbatch, stride, nrows); a.sad_stride = 3 * stride;
    a.row_start = d_rows; a.sorted = d_rows + (size_t)nbatch * (nrows + 1); a.rband = stereo_rband(scale8);
    SadPlanes sp;
    sp.pyrL = pyrL; sp.pyrR = pyrR; sp.inv_sc = isc;
    hipLaunchKernelGGL(k_stereo_rows, dim3(nbatch), dim3(256), (size_t)(2 * (nrows + 1)) * 4, s, d_kr, d_nr, stride, nrows,
                       d_rows, d_rows + (size_t)nbatch * (nrows + 1));
    hipLaunchKernelGGL(k_stereo_scan<true>, dim3((stride + 3) / 4, nbatch), dim3(256), 0, s, a, sc, g, sp);
    hipLaunchKernelGGL(k_stereo_sad_finish, dim3(nbatch), dim3(256), 0, s, d_nl, stride, a.sad, a.sad_stride, d_u_right,
                       d_depth, d_nmatched);
}

// ---------------------------------------------------------------------------
// Frame grid: AssignFeaturesToGrid / PosInGrid (src/Frame.cc:536-551, :724-734)
// and GetFeaturesInArea (:669-722), 64 x 48 cells (include/Frame.h:92-93).
// k_grid_build: one workgroup per frame; counting sort of the keypoints by cell
// that keeps ascending keypoint index inside a cell (push_back order).
// k_grid_search: one thread per query walks its cell range in the reference's
// order (cell columns, then rows, then the cell's list) and either counts /
// lists the candidates or feeds them straight into the best / second-best keys.
// ---------------------------------------------------------------------------
constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;

__global__ __launch_bounds__(256) void k_grid_build(const orbx_keypoint* __restrict__ kps, const int32_t* __restrict__ n_arr,
                                                    int N, int stride, float minX, float minY, float invW, float invH,
                                                    int* __restrict__ cell_start, int* __restrict__ cell_list)
{
    __shared__ int s_cnt[kGridCells];
    __shared__ int s_wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_arr ? n_arr[b] : N, stride);
    kps += (size_t)b * stride;
    cell_start += (size_t)b * (kGridCells + 1);
    cell_list += (size_t)b * stride;
    auto cell_of = [&](int i) -> int {
        const int posX = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, minX), invW));
        const int posY = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, minY), invH));
        return (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) ? -1 : posX * kGridRows + posY;
    };
    for (int c = tid; c < kGridCells; c += 256) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = cell_of(i);
        if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of the 3072 counts: 12 per thread
    int local[12], sum = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) { local[k] = s_cnt[tid * 12 + k]; sum += local[k]; }
    int incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    __syncthreads();
    const int first = base;
#pragma unroll
    for (int k = 0; k < 12; k++) { s_cnt[tid * 12 + k] = base; cell_start[tid * 12 + k] = base; base += local[k]; }
    if (tid == 255) cell_start[kGridCells] = base;
    __syncthreads();
    // scatter in any order, then every thread puts the short lists of its 12 cells into ascending keypoint
    // order (the push_back order of the reference)
    for (int i = tid; i < n; i += 256) {
        const int c = cell_of(i);
        if (c >= 0) cell_list[atomicAdd(&s_cnt[c], 1)] = i;
    }
    __syncthreads();
    int lo = first;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int hi = lo + local[k];
        for (int a = lo + 1; a < hi; a++) {
            const int v = cell_list[a];
            int p = a - 1;
            while (p >= lo && cell_list[p] > v) { cell_list[p + 1] = cell_list[p]; p--; }
            cell_list[p + 1] = v;
        }
        lo = hi;
    }
}

struct GridQueryArgs {
    const float *qx, *qy, *qr;
    const int32_t *qmin, *qmax;
    const uint32_t* qdesc;          // NULL: count / list only
    const int32_t* nq;              // per batch, or NULL
    int Q, q_stride;
};

template <bool LIST>
__global__ __launch_bounds__(256) void k_grid_search(const orbx_keypoint* __restrict__ kps, const uint32_t* __restrict__ desc,
                                                     const int* __restrict__ cell_start, const int* __restrict__ cell_list,
                                                     int stride, GridQueryArgs a, float minX, float minY, float invW, float invH,
                                                     int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                     int32_t* __restrict__ second_dist, int32_t* __restrict__ best_lvl,
                                                     int32_t* __restrict__ second_lvl, int32_t* __restrict__ ncand,
                                                     const int32_t* __restrict__ cand_off, int32_t* __restrict__ cand_idx,
                                                     uint16_t* __restrict__ cand_dist)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int nQ = a.nq ? min(a.nq[b], a.q_stride) : a.Q;
    if (q >= nQ) return;
    kps += (size_t)b * stride;
    cell_start += (size_t)b * (kGridCells + 1);
    cell_list += (size_t)b * stride;
    const size_t qo = (size_t)b * a.q_stride + q;
    const float x = a.qx[qo], y = a.qy[qo], r = a.qr[qo];
    const int minLevel = a.qmin[qo], maxLevel = a.qmax[qo];
    const bool has_desc = a.qdesc != nullptr;
    uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (has_desc) {
        const uint4* qp = reinterpret_cast<const uint4*>(a.qdesc + qo * 8);
        uint4 u = qp[0], v = qp[1];
        qd[0] = u.x; qd[1] = u.y; qd[2] = u.z; qd[3] = u.w;
        qd[4] = v.x; qd[5] = v.y; qd[6] = v.z; qd[7] = v.w;
        desc += (size_t)b * stride * 8;
    }
    uint32_t best = kKeyNone, second = kKeyNone;
    int bi = -1, si = -1;       // keypoint indices behind the two keys
    int count = 0;
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
    const int nMaxCellX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
    const int nMaxCellY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
    const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    int wr = LIST ? cand_off[qo] : 0;
    // cells are numbered column by column (ix * 48 + iy), so the lists of the cells iy = nMinCellY .. nMaxCellY of one
    // column are one contiguous run of cell_list, in the reference's visiting order
    if (!empty)
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
            const int j1 = cell_start[ix * kGridRows + nMaxCellY + 1];
            for (int j = cell_start[ix * kGridRows + nMinCellY]; j < j1; j++) {
                const int idx = cell_list[j];
                const orbx_keypoint kp = kps[idx];
                if (checkLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r) {
                    if (LIST) cand_idx[wr] = idx;
                    if (has_desc) {
                        const uint4* tp = reinterpret_cast<const uint4*>(desc + (size_t)idx * 8);
                        const int d = hamming256(qd, tp[0], tp[1]);
                        if (LIST && cand_dist) cand_dist[wr] = (uint16_t)d;   // every candidate's distance, for searches that exclude
                        const uint32_t key = ((uint32_t)d << 22) | (uint32_t)min(count, 0x3FFFFF);
                        if (key < best) { second = best; si = bi; best = key; bi = idx; }
                        else if (key < second) { second = key; si = idx; }
                    }
                    if (LIST) wr++;
                    count++;
                }
            }
        }
    if (ncand) ncand[qo] = count;
    if (has_desc && best_idx) {
        const bool bok = (best >> 22) < 256u, sok = (second >> 22) < 256u;
        best_idx[qo] = bok ? bi : -1;
        best_dist[qo] = (int32_t)(best >> 22);
        second_dist[qo] = (int32_t)(second >> 22);
        if (best_lvl) best_lvl[qo] = bok ? kps[bi].octave : -1;
        if (second_lvl) second_lvl[qo] = sok ? kps[si].octave : -1;
    }
}

// The fused search (query descriptors given, no lists): 16 lanes per query, lane = cell column of the query's
// window.  A candidate's place in the reference's visiting order is its position in cell_list (columns ascending,
// then rows, then the cell's list), so key = distance << 22 | position orders candidates exactly like the scan of
// src/ORBmatcher.cc:186-225 (first candidate wins ties), in any evaluation order: every lane scans its columns'
// runs, then the 16 (best, second) pairs are merged.
__global__ __launch_bounds__(256) void k_grid_search_coop(const orbx_keypoint* __restrict__ kps, const uint32_t* __restrict__ desc,
                                                          const int* __restrict__ cell_start, const int* __restrict__ cell_list,
                                                          int stride, GridQueryArgs a, float minX, float minY, float invW, float invH,
                                                          int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                          int32_t* __restrict__ second_dist, int32_t* __restrict__ best_lvl,
                                                          int32_t* __restrict__ second_lvl, int32_t* __restrict__ ncand)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int nQ = a.nq ? min(a.nq[b], a.q_stride) : a.Q;
    if (q >= nQ) return;                      // whole 16-lane groups leave together
    kps += (size_t)b * stride;
    desc += (size_t)b * stride * 8;
    cell_start += (size_t)b * (kGridCells + 1);
    cell_list += (size_t)b * stride;
    const size_t qo = (size_t)b * a.q_stride + q;
    const float x = a.qx[qo], y = a.qy[qo], r = a.qr[qo];
    const int minLevel = a.qmin[qo], maxLevel = a.qmax[qo];
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
    const int nMaxCellX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
    const int nMaxCellY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
    const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    uint32_t best = kKeyNone, second = kKeyNone;
    int count = 0;
    if (!empty && nMinCellX + sub <= nMaxCellX) {
        uint32_t qd[8];
        const uint4* qp = reinterpret_cast<const uint4*>(a.qdesc + qo * 8);
        const uint4 u = qp[0], v = qp[1];
        qd[0] = u.x; qd[1] = u.y; qd[2] = u.z; qd[3] = u.w;
        qd[4] = v.x; qd[5] = v.y; qd[6] = v.z; qd[7] = v.w;
        for (int ix = nMinCellX + sub; ix <= nMaxCellX; ix += 16) {
            const int j1 = cell_start[ix * kGridRows + nMaxCellY + 1];
            for (int j = cell_start[ix * kGridRows + nMinCellY]; j < j1; j++) {
                const int idx = cell_list[j];
                const orbx_keypoint kp = kps[idx];
                if (checkLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r) {
                    const uint4* tp = reinterpret_cast<const uint4*>(desc + (size_t)idx * 8);
                    const int d = hamming256(qd, tp[0], tp[1]);
                    push_key(best, second, ((uint32_t)d << 22) | (uint32_t)j);
                    count++;
                }
            }
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)best, d, 16), os = (uint32_t)__shfl_xor((int)second, d, 16);
        count += __shfl_xor(count, d, 16);
        const uint32_t nb = min(best, ob);
        second = min(max(best, ob), min(second, os));
        best = nb;
    }
    if (sub != 0) return;
    if (ncand) ncand[qo] = count;
    const bool bok = (best >> 22) < 256u, sok = (second >> 22) < 256u;
    const int bi = bok ? cell_list[best & 0x3FFFFFu] : -1, si = sok ? cell_list[second & 0x3FFFFFu] : -1;
    best_idx[qo] = bi;
    best_dist[qo] = (int32_t)(best >> 22);
    second_dist[qo] = (int32_t)(second >> 22);
    if (best_lvl) best_lvl[qo] = bok ? kps[bi].octave : -1;
    if (second_lvl) second_lvl[qo] = sok ? kps[si].octave : -1;
}

// The candidate lists of the queries (Frame::GetFeaturesInArea, src/Frame.cc:669-722) with 16 lanes per query, lane =
// cell column of the query's window: the count pass (FILL = false) writes every query's list length, the fill pass,
// given the offsets, writes the lists in the reference's visiting order (columns ascending, each column's run of
// cell_list as it stands) with every candidate's distance.  A lane counts its column first, the 16 counts are scanned,
// then the lane writes its column at its place.  One thread per query (k_grid_search<true>) walks the same lists
// serially: a chain of dependent loads per candidate, 70 us for a frame's 1000 queries where this takes a fraction.
template <bool FILL>
__global__ __launch_bounds__(256) void k_grid_lists_coop(const orbx_keypoint* __restrict__ kps, const uint32_t* __restrict__ desc,
                                                         const int* __restrict__ cell_start, const int* __restrict__ cell_list,
                                                         int stride, GridQueryArgs a, float minX, float minY, float invW, float invH,
                                                         int32_t* __restrict__ ncand, const int32_t* __restrict__ cand_off,
                                                         int32_t* __restrict__ cand_idx, uint16_t* __restrict__ cand_dist)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int nQ = a.nq ? min(a.nq[b], a.q_stride) : a.Q;
    if (q >= nQ) return;                      // whole 16-lane groups leave together
    kps += (size_t)b * stride;
    cell_start += (size_t)b * (kGridCells + 1);
    cell_list += (size_t)b * stride;
    const size_t qo = (size_t)b * a.q_stride + q;
    const float x = a.qx[qo], y = a.qy[qo], r = a.qr[qo];
    const int minLevel = a.qmin[qo], maxLevel = a.qmax[qo];
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
    const int nMaxCellX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
    const int nMaxCellY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
    const bool empty = nMinCellX >= kGridCols || nMaxCellX < 0 || nMinCellY >= kGridRows || nMaxCellY < 0;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (FILL && a.qdesc) {
        const uint4* qp = reinterpret_cast<const uint4*>(a.qdesc + qo * 8);
        const uint4 u = qp[0], v = qp[1];
        qd[0] = u.x; qd[1] = u.y; qd[2] = u.z; qd[3] = u.w;
        qd[4] = v.x; qd[5] = v.y; qd[6] = v.z; qd[7] = v.w;
        desc += (size_t)b * stride * 8;
    }
    auto inside = [&](const orbx_keypoint& kp) -> bool {
        if (checkLevels) {
            if (kp.octave < minLevel) return false;
            if (maxLevel >= 0 && kp.octave > maxLevel) return false;
        }
        return fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r;
    };
    const int base = FILL ? cand_off[qo] : 0;
    int total = 0;
    if (!empty)
        for (int ix0 = nMinCellX; ix0 <= nMaxCellX; ix0 += 16) {      // the same trip count for the 16 lanes of a query
            const int ix = ix0 + sub;
            int j0 = 0, j1 = 0;
            if (ix <= nMaxCellX) {
                j0 = cell_start[ix * kGridRows + nMinCellY];
                j1 = cell_start[ix * kGridRows + nMaxCellY + 1];
            }
            int cnt = 0;
            for (int j = j0; j < j1; j++) cnt += inside(kps[cell_list[j]]) ? 1 : 0;
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int t = __shfl_up(incl, d, 16);
                if (sub >= d) incl += t;
            }
            if (FILL) {
                int wr = base + total + incl - cnt;
                for (int j = j0; j < j1; j++) {
                    const int idx = cell_list[j];
                    if (!inside(kps[idx])) continue;
                    cand_idx[wr] = idx;
                    if (cand_dist) {
                        const uint4* tp = reinterpret_cast<const uint4*>(desc + (size_t)idx * 8);
                        cand_dist[wr] = (uint16_t)hamming256(qd, tp[0], tp[1]);
                    }
                    wr++;
                }
            }
            total += __shfl(incl, 15, 16);
        }
    if (!FILL && sub == 0) ncand[qo] = total;
}

// count pass (cand_idx == nullptr: list lengths into ncand) or fill pass (lists at cand_off, distances if qdesc and cand_dist)
void launch_grid_lists(const orbx_keypoint* d_kps, const uint32_t* d_desc, const int* d_cell_start, const int* d_cell_list,
                       int stride, int nbatch, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                       const int32_t* qmax, const uint32_t* qdesc, const int32_t* d_nq, int Q, int q_stride, float minX, float minY,
                       float invW, float invH, int32_t* ncand, const int32_t* cand_off, int32_t* cand_idx, uint16_t* cand_dist,
                       hipStream_t s)
{
    if (q_stride <= 0) return;
    GridQueryArgs a{qx, qy, qr, qmin, qmax, qdesc, d_nq, Q, q_stride};
    const dim3 grid((q_stride + 15) / 16, nbatch);
    if (cand_idx)
        hipLaunchKernelGGL(k_grid_lists_coop<true>, grid, dim3(256), 0, s, d_kps, d_desc, d_cell_start, d_cell_list, stride, a, minX,
                           minY, invW, invH, ncand, cand_off, cand_idx, qdesc ? cand_dist : nullptr);
    else
        hipLaunchKernelGGL(k_grid_lists_coop<false>, grid, dim3(256), 0, s, d_kps, d_desc, d_cell_start, d_cell_list, stride, a, minX,
                           minY, invW, invH, ncand, cand_off, cand_idx, cand_dist);
}

void launch_grid_build(const orbx_keypoint* d_kps, const int32_t* d_n, int N, int stride, int nbatch, float minX,
                       float minY, float invW, float invH, int* d_cell_start, int* d_cell_list, hipStream_t s)
{
    hipLaunchKernelGGL(k_grid_build, dim3(nbatch), dim3(256), 0, s, d_kps, d_n, N, stride, minX, minY, invW, invH,
                       d_cell_start, d_cell_list);
}

void launch_grid_search(const orbx_keypoint* d_kps, const uint32_t* d_desc, const int* d_cell_start,
                        const int* d_cell_list, int stride, int nbatch, const float* qx, const float* qy, const float* qr,
                        const int32_t* qmin, const int32_t* qmax, const uint32_t* qdesc, const int32_t* d_nq, int Q,
                        int q_stride, float minX, float minY, float invW, float invH, int32_t* best_idx,
                        int32_t* best_dist, int32_t* second_dist, int32_t* best_lvl, int32_t* second_lvl, int32_t* ncand,
                        const int32_t* cand_off, int32_t* cand_idx, hipStream_t s, uint16_t* cand_dist)
{
    if (q_stride <= 0) return;
    GridQueryArgs a{qx, qy, qr, qmin, qmax, qdesc, d_nq, Q, q_stride};
    dim3 grid((q_stride + 255) / 256, nbatch);
    if (!cand_idx && qdesc && stride < (1 << 22)) {        // the key's position field holds 22 bits
        hipLaunchKernelGGL(k_grid_search_coop, dim3((q_stride + 15) / 16, nbatch), dim3(256), 0, s, d_kps, d_desc, d_cell_start,
                           d_cell_list, stride, a, minX, minY, invW, invH, best_idx, best_dist, second_dist, best_lvl,
                           second_lvl, ncand);
        return;
    }
    if (cand_idx)
        hipLaunchKernelGGL(k_grid_search<true>, grid, dim3(256), 0, s, d_kps, d_desc, d_cell_start, d_cell_list, stride, a,
                           minX, minY, invW, invH, best_idx, best_dist, second_dist, best_lvl, second_lvl, ncand, cand_off,
                           cand_idx, cand_dist);
    else
        hipLaunchKernelGGL(k_grid_search<false>, grid, dim3(256), 0, s, d_kps, d_desc, d_cell_start, d_cell_list, stride, a,
                           minX, minY, invW, invH, best_idx, best_dist, second_dist, best_lvl, second_lvl, ncand, cand_off,
                           cand_idx, cand_dist);
}

} // namespace orbx
