// Fused Adam update over a flat parameter arena (one launch for all tensors / models).
//
// Arithmetic follows torch.optim.Adam's single-tensor CPU path (torch 2.10
// optim/adam.py:457-547), the optimiser the reference constructs at runners/ivae.py:38 and
// models/wrappers.py:44,127:   m.lerp_(g, 1-b1);  v.mul_(b2).addcmul_(g, g, 1-b2);
//   denom = sqrt(v)/sqrt(1-b2^t) + eps;  p.addcdiv_(m, denom, -lr/(1-b1^t)).
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                              const float* __restrict__ g, int64_t n, float w1, float b2, float w2,
                                              float neg_step, float bc2_sqrt, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update(pi, mi, vi, g[i], w1, b2, w2, neg_step, bc2_sqrt, eps);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

hipError_t launch_adam(float* p, float* m, float* v, const float* g, int64_t n, double lr, double b1, double b2,
                       double eps, int64_t step, hipStream_t st) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float neg_step = (float)(-lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, st, p, m, v, g, n, (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), neg_step, bc2_sqrt, (float)eps);
    return hipGetLastError();
}

// Packed models with their own learning rates (every run of a packed seed sweep has its own ReduceLROnPlateau,
// runners/ivae.py:39,113-114): blockIdx.y = model, its step size travels in the kernel arguments.
struct AdamModelSteps { float neg_step[ADAM_MODELS_PER_LAUNCH]; };

__global__ __launch_bounds__(256) void k_adam_models(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                     const float* __restrict__ g, int64_t n_per_model, float w1, float b2,
                                                     float w2, AdamModelSteps S, float bc2_sqrt, float eps) {
    const int64_t base = (int64_t)blockIdx.y * n_per_model;
    const float neg_step = S.neg_step[blockIdx.y];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_per_model; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + j;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update(pi, mi, vi, g[i], w1, b2, w2, neg_step, bc2_sqrt, eps);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

hipError_t launch_adam_models(float* p, float* m, float* v, const float* g, int64_t n_per_model, int n_models,
                              const double* model_lr, double b1, double b2, double eps, int64_t step, hipStream_t st) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const float bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)step));
    int64_t bx = (n_per_model + 255) / 256;
    if (bx > 64) bx = 64;
    for (int m0 = 0; m0 < n_models; m0 += ADAM_MODELS_PER_LAUNCH) {
        const int nm = n_models - m0 < ADAM_MODELS_PER_LAUNCH ? n_models - m0 : ADAM_MODELS_PER_LAUNCH;
        AdamModelSteps S;
        for (int i = 0; i < nm; ++i) S.neg_step[i] = (float)(-model_lr[m0 + i] / bc1);
        const int64_t off = (int64_t)m0 * n_per_model;
        hipLaunchKernelGGL(k_adam_models, dim3((unsigned)bx, (unsigned)nm), dim3(256), 0, st, p + off, m + off, v + off, g + off,
                           n_per_model, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), S, bc2_sqrt, (float)eps);
    }
    return hipGetLastError();
}
