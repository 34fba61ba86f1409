// The tile loop of the brute-force matcher built up piece by piece, to see which piece keeps the matrix pipe and the
// vector ALU from running at the same time.  Four-wavefront workgroups, four of them per compute unit (LDS sized for
// that), every wavefront: groups of four tiles, per tile 8 FP4 matrix instructions (two accumulators) and the 46 min3 /
// med3 instructions of the best-two update.  Flags: 1 the A fragments come from LDS (one 16-byte read per k-step),
// 2 the accumulators' start values come from LDS (four 16-byte reads per tile), 4 a workgroup barrier per group,
// 8 each wavefront writes four 16-byte fragments per group to the other LDS buffer (as the expansion does).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o sweep_bench sweep_bench.hip && ./sweep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

template <int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sweep(uint32_t* out, unsigned long long* cycles, int groups)
{
    __shared__ v4i_t s_a[2][4][256];
    __shared__ __align__(16) float s_c[2][128];
    __shared__ uint32_t s_pad[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    // (operand values matter: with random bits in every nibble, as real descriptors have, the chip draws more power
    // and clocks lower than on the near-zero patterns of flags < 16)
    auto rnd = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return (int)(x & 0x22222222u); };
    for (int i = threadIdx.x; i < 2 * 4 * 256; i += 256)
        (&s_a[0][0][0])[i] = (F & 16) ? v4i_t{rnd(4 * i + blockIdx.x), rnd(4 * i + 1), rnd(4 * i + 2), rnd(4 * i + 3)} : v4i_t{i, i * 3, i * 5, i * 7};
    s_c[0][threadIdx.x & 127] = 12582912.0f + threadIdx.x;
    s_c[1][threadIdx.x & 127] = 12582912.0f + threadIdx.x;
    s_pad[threadIdx.x] = threadIdx.x;
    v4i_t bq[2][4];
    for (int qt = 0; qt < 2; qt++)
        for (int s = 0; s < 4; s++)
            bq[qt][s] = (F & 16) ? v4i_t{rnd(threadIdx.x * 64 + qt * 16 + s * 4) * 5, rnd(threadIdx.x * 64 + qt * 16 + s * 4 + 1) * 5,
                                         rnd(threadIdx.x * 64 + qt * 16 + s * 4 + 2) * 5, rnd(threadIdx.x * 64 + qt * 16 + s * 4 + 3) * 5}
                                 : v4i_t{lane ^ (5 + s), lane ^ (9 + qt), lane + 1, lane + 2};
    uint32_t best[2] = {0xffffffffu, 0xffffffffu}, second[2] = {0xffffffffu, 0xffffffffu};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int buf = 0;
    for (int g = 0; g < groups; g++, buf ^= 1) {
        if (F & 8) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_a[buf ^ 1][k][wave * 64 + lane] = (F & 16) ? s_a[buf][k][wave * 64 + ((lane + 1) & 63)] : v4i_t{g, lane, k, wave};
        }
        if (F & 4) __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v16f_t acc[2];
#pragma unroll
            for (int e = 0; e < 16; e++) acc[0][e] = acc[1][e] = (F & 2) ? s_c[buf][32 * k + 4 * half + 8 * (e >> 2) + (e & 3)] : 12582912.0f + e;
            v4i_t a_cur = (F & 1) ? s_a[buf][k][lane] : bq[0][0], a_nxt = a_cur;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if ((F & 1) && s < 3) a_nxt = s_a[buf][k][(s + 1) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const v4i_t a = a_cur;
                a_cur = a_nxt;
                const v8i_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
                for (int qt = 0; qt < 2; qt++) {
                    const v8i_t b8 = {bq[qt][s][0], bq[qt][s][1], bq[qt][s][2], bq[qt][s][3], 0, 0, 0, 0};
                    acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[qt], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                uint32_t kk[16];
#pragma unroll
                for (int e = 0; e < 16; e++) { const float f = acc[qt][e]; kk[e] = __float_as_uint(f); }
                uint32_t m[6], d[6];
#pragma unroll
                for (int i = 0; i < 5; i++) { m[i] = umin3(kk[3 * i], kk[3 * i + 1], kk[3 * i + 2]); d[i] = umed3(kk[3 * i], kk[3 * i + 1], kk[3 * i + 2]); }
                m[5] = min(best[qt], kk[15]);
                d[5] = umed3(best[qt], second[qt], kk[15]);
                const uint32_t ma = umin3(m[0], m[1], m[2]), da = min(umed3(m[0], m[1], m[2]), umin3(d[0], d[1], d[2]));
                const uint32_t mb = umin3(m[3], m[4], m[5]), db = min(umed3(m[3], m[4], m[5]), umin3(d[3], d[4], d[5]));
                best[qt] = min(ma, mb);
                second[qt] = umin3(max(ma, mb), da, db);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = best[0] ^ second[0] ^ best[1] ^ second[1] ^ s_pad[lane];
    if (lane == 0) atomicMax(&cycles[blockIdx.x], t1 - t0);
}

template <int F>
static void run(const char* name)
{
    const int blocks = 1024, groups = 400;
    uint32_t* d_out;
    unsigned long long* d_c;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMalloc(&d_c, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d_c, 0, blocks * 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_sweep<F>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, groups);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : c) sum += (double)v;
    // 4 wavefronts per SIMD, each `groups` x 4 tiles
    std::printf("flags %2d  %-58s %7.1f cycles per SIMD and tile, %.3f ms = %.1f ns per SIMD and tile\n", F, name,
                sum / blocks / (groups * 4) / 4, ms, ms * 1e6 / (groups * 4) / 4);
    hipFree(d_out);
    hipFree(d_c);
}

int main()
{
    run<0>("registers only");
    run<1>("A fragments from LDS");
    run<2>("start values from LDS");
    run<3>("A fragments and start values from LDS");
    run<4>("registers only, barrier per group");
    run<7>("LDS reads and barrier");
    run<11>("LDS reads and fragment writes");
    run<15>("LDS reads, fragment writes and barrier");
    run<31>("the same on random bits (nibbles 0 / 1.0 against 0 / -1.0)");
    run<15>("LDS reads, fragment writes and barrier, again");
    return 0;
}
