// Floor of a launch shaped like k_train_steps (256 workgroups x 512 threads, 160 KB of dynamic LDS each) that does nothing,
// and of the same launch moving k_train_steps's state (61 MB in, 61 MB out), by HIP events recorded around the launch.
// build: hipcc --offload-arch=gfx950 -O2 launch_floor.hip -o ../../var/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_empty(float* p) {
    extern __shared__ float lds[];
    if (p == (float*)1) lds[threadIdx.x] = 1.f;
}
__global__ __launch_bounds__(512) void k_state(const float* in, float* out, int per_block) {
    extern __shared__ float lds[];
    const float* src = in + (size_t)blockIdx.x * per_block;
    float* dst = out + (size_t)blockIdx.x * per_block;
    float r[120];
#pragma unroll
    for (int j = 0; j < 120; ++j) { const int i = threadIdx.x + j * 512; r[j] = i < per_block ? src[i] : 0.f; }
    lds[threadIdx.x] = r[0];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 120; ++j) { const int i = threadIdx.x + j * 512; if (i < per_block) dst[i] = r[j] + lds[(threadIdx.x + j) & 511]; }
}
int main() {
    const int per_block = 3 * 19940;           // parameters + two moments of one model
    float *a, *b;
    hipMalloc(&a, (size_t)256 * per_block * 4); hipMalloc(&b, (size_t)256 * per_block * 4);
    hipMemset(a, 0, (size_t)256 * per_block * 4);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipFuncSetAttribute((const void*)k_state, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f, sum = 0;
        for (int it = 0; it < 30; ++it) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 163840, 0, a);
            else hipLaunchKernelGGL(k_state, dim3(256), dim3(512), 163840, 0, a, b, per_block);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 5) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%s: mean %.1f us, best %.1f us\n", which ? "state in + out (61 MB + 61 MB)" : "empty kernel", sum / 25 * 1e3, best * 1e3);
    }
    return 0;
}
