// What a workgroup barrier stretch costs on gfx950 as a function of the workgroup size: ITER iterations of
// {one dependent LDS read-modify-write, __syncthreads()} in 64 workgroups (one per compute unit at most).
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_bench barrier_bench.hip ; run: ./barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT>
__global__ __launch_bounds__(NT) void k(int* out, int iters)
{
    __shared__ int lds[1024];
    const int tid = threadIdx.x;
    lds[tid] = tid;
    __syncthreads();
    int v = 0;
    for (int i = 0; i < iters; i++) {
        v += lds[(tid + i) & (NT - 1)];
        __syncthreads();
        lds[tid] = v;
        __syncthreads();
    }
    if (v == 0x7fffffff) out[blockIdx.x] = v;
}

template <int NT>
void run(int* d_out)
{
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<NT>, dim3(64), dim3(NT), 0, 0, d_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NT>, dim3(64), dim3(NT), 0, 0, d_out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::printf("%4d threads: %.3f us per barrier stretch (%d stretches)\n", NT, ms * 1e3 / (2.0 * iters), 2 * iters);
}

int main()
{
    int* d_out;
    hipMalloc(&d_out, 64 * sizeof(int));
    run<64>(d_out);
    run<256>(d_out);
    run<512>(d_out);
    run<1024>(d_out);
    hipFree(d_out);
    return 0;
}
