// Micro-benchmark: issue rate of plain, packed and transcendental fp32 vector instructions at 1, 2 and 4 waves per SIMD
// (s_memtime around an unrolled stream of independent instructions; the figure printed is SIMD cycles per wave64
// instruction = wave time / instructions / waves per SIMD).  What the pairwise kernels are priced against.
// Build: hipcc -O3 --offload-arch=gfx950 valu_pk_rate.hip -o valu_pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* t, float s, int reps) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 s2 = {s, s * 0.999f}, one = {1.f, 0.5f};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r)
#pragma unroll
    for (int i = 0; i < 128; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(s), "v"(one.x));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(s2), "v"(one));
            else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j].x));
            else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(one));
            else if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(s2));
            else {  // the pairwise mix: 3 pk + 1 exp + 1 plain per 5
                if (j % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j].x));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(s2), "v"(one));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&t, 256 * 16 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, t, 0.999f, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), t, 256 * 16 * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double m = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m += h[b * 16 + w];
    m /= 256.0 * waves;
    const double wps = waves / 4.0;
    printf("%-22s %d waves/SIMD: %8.0f cycles per wave for 1024 instr -> %.2f cycles per instr per wave, %.2f SIMD cycles per instr\n",
           name, waves / 4, m, m / 1024, m / 1024 / wps);
    // the same stream 64 times over, whole chip, timed by events: cycles / wall time = the clock the chip holds under this load
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, t, 0.999f, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, t, 0.999f, 64);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), t, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double m2 = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m2 += h[b * 16 + w];
    m2 /= 256.0 * waves;
    printf("    x64, 256 workgroups: %.0f cycles in %.1f us -> %.2f GHz\n", m2, ms * 1e3, m2 / (ms * 1e3) / 1e3);
    hipFree(out); hipFree(t);
}

int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", th);
        run<1>("v_pk_fma_f32", th);
        run<2>("v_exp_f32", th);
        run<3>("v_pk_add_f32", th);
        run<4>("v_pk_mul_f32", th);
        run<5>("3 pk_fma + 1 exp", th);
    }
    return 0;
}
