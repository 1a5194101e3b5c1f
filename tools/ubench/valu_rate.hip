// Micro-benchmark: cycles per VALU instruction for a wave's stream at 1 and 2 waves per SIMD (s_memtime around an
// unrolled stream).  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, float s) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    __shared__ float lds[8192];
    lds[threadIdx.x] = a0;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        if (MODE == 0) {            // 8 independent chains of fma
            a0 = fmaf(a0, s, 1.f); a1 = fmaf(a1, s, 1.f); a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f);
            a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f);
        } else if (MODE == 1) {     // one dependent chain
            a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f);
            a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f); a0 = fmaf(a0, s, 1.f);
        } else if (MODE == 2) {     // epilogue-like: mul, max, cndmask, LDS store (2 chains)
            float m0 = a0 * s, m1 = a1 * s;
            m0 = fmaxf(a0, m0); m1 = fmaxf(a1, m1);
            a0 = (threadIdx.x & 1) ? m0 : 1.f; a1 = (threadIdx.x & 2) ? m1 : 0.f;
            *reinterpret_cast<float2*>(&lds[(threadIdx.x * 2 + 64 * (i & 7)) & 8190]) = make_float2(a0, a1);
            a2 += a0; a3 += a1; a4 = a2 * s; a5 = a3 * s;
        } else {                    // LDS round trip: write then dependent read
            lds[(threadIdx.x + 64 * (i & 15)) & 8191] = a0;
            a0 = lds[(threadIdx.x ^ 1) + 64 * (i & 15) & 8191] + 1.f;
            a1 = fmaf(a1, s, a0); a2 = fmaf(a2, s, a1); a3 = fmaf(a3, s, 1.f); a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, t, 0.999f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= 256;
    printf("%-28s %4d threads: %8.0f cycles for 256 iterations x 8 instr-ish -> %.2f cycles per (8-instr) iteration\n", name, threads, m, m / 256);
    hipFree(out); hipFree(t);
}

int main() {
    for (int th : {256, 512}) {
        run<0>("8 independent fma", th);
        run<1>("8 dependent fma", th);
        run<2>("epilogue-like (8 valu + st64)", th);
        run<3>("lds write->read + 6 fma", th);
    }
    return 0;
}
