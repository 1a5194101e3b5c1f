// Same hidden layer as layer_loop.hip (64 rows x 64 columns, K = 52, lrelu epilogue, one barrier per layer), mapped on
// 16 waves (4 per SIMD, <= 128 registers): wave = (row tile, column quarter), ONE 16x16 tile per wave.
#include "../../ivae_amd/csrc/fused_train.hip"
#include <stdio.h>
#include <vector>
using C = FCfg<10, 10, 40, 50>;

template <int MODE>
__global__ __launch_bounds__(1024) void k_layers16(float* out, unsigned long long* t, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < C::LDS_FLOATS; i += 1024) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    float* T0 = lds;
    float* T2 = T0 + 2 * C::TILE;
    float* arena = T0 + 5 * C::TILE + 5 * C::SMALL;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt = w & 3, cq = w >> 2;
    float accsum = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        float* in = (it & 1) ? T2 : T0;
        float* ou = (it & 1) ? T0 : T2;
        const int lane = opaque_lane();
        const int c = lane & 15, q = lane >> 4;
        const int col = 16 * cq + c;
        const f32x4 acc = mm_fwd1<C::KS_H>(in + mul24(16 * rt + c, C::ST), arena + C::off_of(L_G1) + mul24(min(col, C::H - 1), C::KS_H), q);
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = col < C::H ? lrelu(acc[r], 0.1f) : (col == C::H ? 1.f : 0.f);
                if (col < C::ST) ou[mul24(16 * rt + 4 * q + r, C::ST) + col] = v;
            }
        } else accsum += acc[0] + acc[3];
        if (MODE & 2) lds_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 1024 + threadIdx.x] = accsum;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + w] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&t, 256 * 16 * 8);
    hipFuncSetAttribute((const void*)k_layers16<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_layers16<MODE>, dim3(256), dim3(1024), C::LDS_BYTES, 0, out, t, iters);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), t, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    printf("%-50s 16 waves: %7.0f cycles/layer (MFMA pipe time per SIMD 1664)  [%s]\n", name, m / iters, hipGetErrorString(e));
}

int main() {
    run<1>("one tile per wave, product + epilogue");
    run<3>("one tile per wave, product + epilogue + barrier");
    return 0;
}
