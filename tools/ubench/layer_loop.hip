// Micro-benchmark of ONE hidden layer of the fused kernel (same device functions, included from the source) in a loop:
// what do the product, the epilogue and the barrier cost per iteration, at 1 and 2 waves per SIMD, in lockstep or not?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ivae_amd/csrc tools/ubench/layer_loop.hip -o build/layer_loop
#include "../../ivae_amd/csrc/fused_train.hip"
#include <stdio.h>
#include <vector>

using C = FCfg<10, 10, 40, 50>;

// MODE bits: 1 = epilogue (lrelu + stores), 2 = workgroup barrier per layer, 4 = waves 4-7 start half a layer late,
// 8 = pair-split mapping (wave = row tile x column half, 26 MFMAs) instead of wave-local (52 MFMAs)
template <int MODE>
__global__ __launch_bounds__(FT_THREADS) void k_layers(float* out, unsigned long long* t, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < C::LDS_FLOATS; i += FT_THREADS) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    float* T0 = lds;
    float* T1 = T0 + C::TILE;
    float* T2 = T1 + C::TILE;
    float* T4 = T2 + 2 * C::TILE;
    float* arena = T4 + C::TILE + 5 * C::SMALL;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= nwaves) return;
    const int rt = w & 3;
    f32x4 keep[2][2];
    if ((MODE & 4) && w >= 4) __builtin_amdgcn_s_sleep(26);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        float* in = (w < 4) ? ((it & 1) ? T2 : T0) : ((it & 1) ? T4 : T1);
        float* ou = (w < 4) ? ((it & 1) ? T0 : T2) : ((it & 1) ? T1 : T4);
        const int lane = opaque_lane();
        if (MODE & 8) {
            const float* const W[1] = {arena + C::off_of(L_G1)};
            f32x4 k1[1][2];
            fwd_product<C, 1, C::KS_H>(in, C::ST, W, w, lane, k1, NoFill());
            if (MODE & 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fwd_epilogue_row<C>(k1[0], r, ou, 0.1f, w, lane);
            } else { keep[0][0] += k1[0][0]; keep[0][1] += k1[0][1]; }
        } else {
            fwd_rows_product<C, C::KS_H>(in, C::ST, arena + C::off_of(L_G1), rt, lane, keep, NoFill());
            if (MODE & 1) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fwd_epilogue_row<C>(keep[p], r, ou, 0.1f, rt + 4 * p, lane);
            }
        }
        if (MODE & 2) lds_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * FT_THREADS + threadIdx.x] = keep[0][0][0] + keep[1][1][3] + keep[0][1][1] + keep[1][0][2];
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + w] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nwaves) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8 * 8);
    hipMemset(t, 0, 256 * 8 * 8);
    hipFuncSetAttribute((const void*)k_layers<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_layers<MODE>, dim3(256), dim3(FT_THREADS), C::LDS_BYTES, 0, out, t, iters, nwaves);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), t, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nwaves; ++w) { m += h[b * 8 + w]; ++n; }
    m /= n;
    const int mf = (MODE & 8) ? 26 : 52;
    printf("%-58s %d waves: %7.0f cycles/layer  (MFMA pipe time per SIMD %d)\n", name, nwaves, m / iters, mf * 32 * (nwaves > 4 ? 2 : 1));
    hipFree(out); hipFree(t);
}

int main() {
    run<0>("wave-local product only", 4);
    run<0>("wave-local product only", 8);
    run<1>("wave-local product + epilogue", 4);
    run<1>("wave-local product + epilogue", 8);
    run<5>("wave-local product + epilogue, waves 4-7 half a layer late", 8);
    run<3>("wave-local product + epilogue + barrier", 8);
    run<8>("pair-split product only", 8);
    run<9>("pair-split product + epilogue (no sync: timing only)", 8);
    run<11>("pair-split product + epilogue + barrier", 8);
    return 0;
}
