// Micro-benchmark: what straight-line code costs when its instructions are not in the instruction cache.
// A kernel runs N unrolled independent fma instructions once (s_memtime around them).  "cold": a different big kernel
// (the evictor: > 64 KB of code) ran in between, as in a training step made of several large kernels; "warm": the same
// kernel is launched twice in a row.  Build: hipcc -O3 --offload-arch=gfx950 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int N, int SALT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, float s) {
    float a0 = threadIdx.x * 1e-3f + SALT, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        a0 = fmaf(a0, s, 1.f + i); a1 = fmaf(a1, s, 2.f + i); a2 = fmaf(a2, s, 3.f + i); a3 = fmaf(a3, s, 4.f + i);
        a4 = fmaf(a4, s, 5.f + i); a5 = fmaf(a5, s, 6.f + i); a6 = fmaf(a6, s, 7.f + i); a7 = fmaf(a7, s, 8.f + i);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int N>
void run(int blocks) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&t, 4096 * 8);
    std::vector<unsigned long long> h(blocks);
    auto mean = [&]() { hipDeviceSynchronize(); hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost); double m = 0; for (auto v : h) m += v; return m / blocks; };
    // warm: same kernel twice
    hipLaunchKernelGGL((k<N, 0>), dim3(blocks), dim3(256), 0, 0, out, t, 0.999f);
    hipLaunchKernelGGL((k<N, 0>), dim3(blocks), dim3(256), 0, 0, out, t, 0.999f);
    const double warm = mean();
    // cold: evict with two other 16384-instruction kernels (2 x 128 KB of code), then run
    hipLaunchKernelGGL((k<16384, 1>), dim3(256), dim3(256), 0, 0, out, t, 0.999f);
    hipLaunchKernelGGL((k<16384, 2>), dim3(256), dim3(256), 0, 0, out, t, 0.999f);
    hipLaunchKernelGGL((k<N, 0>), dim3(blocks), dim3(256), 0, 0, out, t, 0.999f);
    const double cold = mean();
    printf("N=%5d fma (%3d KB of code), %4d workgroups of 4 waves: warm %8.0f cycles (%.1f per instr), after eviction %8.0f cycles (%.1f per instr, %.0f per 64-byte line)\n",
           N, N * 8 / 1024, blocks, warm, warm / N, cold, cold / N, cold / (N * 8 / 64.0));
    hipFree(out); hipFree(t);
}

int main() {
    for (int b : {256, 1024}) { run<512>(b); run<2048>(b); run<8192>(b); }
    return 0;
}
