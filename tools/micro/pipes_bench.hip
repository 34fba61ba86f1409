// Do the matrix pipe and the vector ALU of a SIMD run at the same time?  Every wavefront repeats "8 FP4 matrix
// instructions (two accumulators, four k-steps), then NV vector min3 / med3 instructions", as the brute-force matcher does
// per tile, with 1, 2 or 4 wavefronts per SIMD.  Modes: 0 both, the vector instructions read the accumulators; 1 both,
// the vector instructions work on registers of their own; 2 matrix instructions only; 3 vector instructions only.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o pipes_bench pipes_bench.hip && ./pipes_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

template <int MODE>
__global__ __launch_bounds__(1024) void k_pipes(uint32_t* out, unsigned long long* cycles, int iters)
{
    const int lane = threadIdx.x & 63;
    v8i_t a = {lane, lane * 3, lane * 5, lane * 7, 0, 0, 0, 0}, b = {lane ^ 5, lane ^ 9, lane + 1, lane + 2, 0, 0, 0, 0};
    v16f_t acc0, acc1;
    for (int e = 0; e < 16; e++) { acc0[e] = (float)(lane + e); acc1[e] = (float)(lane - e); }
    uint32_t best0 = 0xffffffffu, second0 = 0xffffffffu, best1 = 0xffffffffu, second1 = 0xffffffffu;
    uint32_t w[16];
    for (int e = 0; e < 16; e++) w[e] = lane * 2654435761u + e * 40503u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE != 3) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        if (MODE != 2) {
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                uint32_t k[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    if (MODE == 0) { const float f = qt ? acc1[e] : acc0[e]; k[e] = __float_as_uint(f); }
                    else { w[e] += (uint32_t)it; k[e] = w[e]; }     // (one more instruction per key than mode 0)
                }
                uint32_t m[6], d[6];
#pragma unroll
                for (int i = 0; i < 5; i++) { m[i] = umin3(k[3 * i], k[3 * i + 1], k[3 * i + 2]); d[i] = umed3(k[3 * i], k[3 * i + 1], k[3 * i + 2]); }
                uint32_t& best = qt ? best1 : best0;
                uint32_t& second = qt ? second1 : second0;
                m[5] = min(best, k[15]);
                d[5] = umed3(best, second, k[15]);
                const uint32_t ma = umin3(m[0], m[1], m[2]), da = min(umed3(m[0], m[1], m[2]), umin3(d[0], d[1], d[2]));
                const uint32_t mb = umin3(m[3], m[4], m[5]), db = min(umed3(m[3], m[4], m[5]), umin3(d[3], d[4], d[5]));
                best = min(ma, mb);
                second = umin3(max(ma, mb), da, db);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = best0 ^ second0 ^ best1 ^ second1;
    for (int e = 0; e < 16; e++) r ^= __float_as_uint(acc0[e]) ^ __float_as_uint(acc1[e]) ^ w[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    // (the oldest wavefront of a SIMD wins every arbitration: the slowest one is what counts)
    if (lane == 0) atomicMax(&cycles[blockIdx.x], t1 - t0);
}

template <int MODE>
static void run(int waves_per_simd, const char* name)
{
    const int blocks = 256, threads = 64 * 4 * waves_per_simd, iters = 2000;
    uint32_t* d_out;
    unsigned long long* d_c;
    hipMalloc(&d_out, (size_t)blocks * threads * 4);
    hipMalloc(&d_c, blocks * 8);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d_c, 0, blocks * 8);
        hipLaunchKernelGGL(k_pipes<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, d_c, iters);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : c) sum += (double)v;
    std::printf("%-44s %d wavefronts per SIMD: %7.1f cycles per tile of one wavefront, %7.1f per SIMD and tile\n", name, waves_per_simd,
                sum / blocks / iters, sum / blocks / iters / waves_per_simd);
    hipFree(d_out);
    hipFree(d_c);
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>(1, "matrix + vector on the accumulators"); run<1>(1, "matrix + vector on other registers"); run<2>(1, "matrix only"); run<3>(1, "vector only"); }
        if (w == 2) { run<0>(2, "matrix + vector on the accumulators"); run<1>(2, "matrix + vector on other registers"); run<2>(2, "matrix only"); run<3>(2, "vector only"); }
        if (w == 4) { run<0>(4, "matrix + vector on the accumulators"); run<1>(4, "matrix + vector on other registers"); run<2>(4, "matrix only"); run<3>(4, "vector only"); }
    }
    return 0;
}
