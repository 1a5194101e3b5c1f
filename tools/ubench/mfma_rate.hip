// Micro-benchmark: cycles per v_mfma_f32_16x16x4_f32 on one SIMD for streams with NACC accumulators in round-robin
// order, at 1 and 2 waves per SIMD, bare and with the operand reads of a tile-pair product (3 x ds_read_b64 per 4 MFMAs,
// two k-groups ahead).  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_MFMA 256

template <int NACC, int LOADS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, float s) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * (i & 63);
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const float* p = lds + 2 * lane + 128 * (threadIdx.x >> 6);
    float a = s + lane, b = s - lane;
    f32x2 ring[3][3];
    if (LOADS) {
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 3; ++j) ring[g][j] = *reinterpret_cast<const f32x2*>(p + 1024 * ((3 * g + j) & 7));
    }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(b)::"memory");
#pragma unroll
    for (int g = 0; g < N_MFMA / 4; ++g) {
        if (LOADS) {
            for (int j = 0; j < 3; ++j) ring[(g + 2) % 3][j] = *reinterpret_cast<const f32x2*>(p + 1024 * ((3 * g + j + 6) & 7));
            __builtin_amdgcn_sched_barrier(0);
            const f32x2 x = ring[g % 3][0], y = ring[g % 3][1], z = ring[g % 3][2];
            acc[(4 * g + 0) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc[(4 * g + 0) % NACC], 0, 0, 0);
            acc[(4 * g + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.y, acc[(4 * g + 1) % NACC], 0, 0, 0);
            acc[(4 * g + 2) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, z.x, acc[(4 * g + 2) % NACC], 0, 0, 0);
            acc[(4 * g + 3) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, z.y, acc[(4 * g + 3) % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[(4 * g + i) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(4 * g + i) % NACC], 0, 0, 0);
        }
    }
    f32x4 r = acc[0];
    for (int i = 1; i < NACC; ++i) r += acc[i];
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(r)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int NACC, int LOADS>
void run(int threads) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NACC, LOADS>), dim3(256), dim3(threads), 0, 0, out, t, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= 256;
    const int wps = threads / 256;
    printf("%d accumulators, %s, %d wave(s) per SIMD: %7.0f cycles for %d MFMAs per wave -> %.1f cycles per MFMA on the SIMD's pipe\n",
           NACC, LOADS ? "3 ds_read_b64 per 4 MFMAs" : "no loads", wps, m, N_MFMA, m / (N_MFMA * wps));
    hipFree(out); hipFree(t);
}

int main() {
    for (int th : {256, 512}) {
        run<1, 0>(th); run<2, 0>(th); run<4, 0>(th);
        run<2, 1>(th); run<4, 1>(th);
    }
    return 0;
}
