// Internal launcher declarations (one per .hip translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include "layout.h"

hipError_t launch_forward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                          const float* eps, float* z_out, float* ws, float* grads_for_dec_bwd, hipStream_t st, bool full = false,
                          int* done = nullptr);
// Adam update fused into the slab reduction of launch_backward (ivae_elbo_adam_steps): the thread that forms a gradient
// element applies it, so a training step needs no separate optimiser launch.  Step sizes per model (at most
// ADAM_MODELS_PER_LAUNCH models; the caller falls back to the separate launch beyond that).
#define ADAM_MODELS_PER_LAUNCH 256
struct AdamFuse {
    float* p; float* m; float* v;
    float w1, b2, w2, bc2_sqrt, eps;
    float neg_step[ADAM_MODELS_PER_LAUNCH];
};
// The update of torch.optim.Adam's single-tensor path (optim/adam.py:457-547), shared by every kernel that applies it.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float w1, float b2, float w2,
                                            float neg_step, float bc2_sqrt, float eps) {
    // contractions written out: the same bits whichever kernel this is inlined into
    const float mi = __fmaf_rn(w1, __fsub_rn(g, m), m);
    const float vi = __fmaf_rn(__fmul_rn(w2, g), g, __fmul_rn(v, b2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), bc2_sqrt), eps);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(neg_step, mi), denom));
    m = mi;
    v = vi;
}
// fuse != null: the slab reduction also applies Adam; *fused tells the caller whether it did (a model whose gradient
// needs no slab reduction has no k_reduce launch to ride on: the caller then launches the optimiser itself).
// enc_bwd_done: launch_forward's row-chain kernel already ran the encoder / prior input-gradient chains.
hipError_t launch_backward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                           const float* eps, float* loss_out, float* grads, float* ws, bool dec_bwd_done, hipStream_t st,
                           const AdamFuse* fuse = nullptr, bool* fused = nullptr, bool enc_bwd_done = false);
hipError_t launch_pairwise(int n_rows, int n_cols, int d, const IvaeCoef& C, const float* z, const float* g,
                           const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz,
                           float* cg, float* cv, float* ws, int n_models, size_t mstride, hipStream_t st);
size_t pairwise_ws_floats(int n_rows, int n_cols, int d);
hipError_t launch_adam(float* p, float* m, float* v, const float* g, int64_t n, double lr, double b1, double b2,
                       double eps, int64_t step, hipStream_t st);
hipError_t launch_adam_models(float* p, float* m, float* v, const float* g, int64_t n_per_model, int n_models,
                              const double* model_lr, double b1, double b2, double eps, int64_t step, hipStream_t st);
hipError_t launch_corr(const float* x, const float* y, int64_t n, int d, double* cc, double* ws, hipStream_t st);
size_t corr_ws_bytes(int64_t n, int d);
hipError_t launch_corr_batched(const float* x, const float* y, int64_t n_blocks, int64_t n, int d, double* cc, hipStream_t st);
hipError_t launch_corr_static(const float* x, const float* y, int64_t n, int d, double* cc, hipStream_t st);
#define CORR_STATIC_MAX_D 40

// Dynamic LDS above 64 KB needs the function attribute.  hipFuncSetAttribute SETS the limit (it does not take a maximum),
// and host threads may launch the same kernel with different sizes at the same moment (sweep.py trains one pack per
// thread and stream): check, call and cache update therefore happen under one lock, so the attribute ON THE CURRENT
// DEVICE only ever grows and the cached value never exceeds it.  The lock-free early return is safe for the same
// reason: a cached value was written after the attribute had been raised to it.
#define IVAE_MAX_DEVICES 64
struct SmemCache { int v[IVAE_MAX_DEVICES]; };
inline std::mutex& smem_mutex() {
    static std::mutex m;
    return m;
}
static inline hipError_t set_smem(const void* fn, int bytes, SmemCache& cache) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < IVAE_MAX_DEVICES;
    if (cached && bytes <= __atomic_load_n(&cache.v[dev], __ATOMIC_ACQUIRE)) return hipSuccess;
    std::lock_guard<std::mutex> lock(smem_mutex());
    if (cached && bytes <= cache.v[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && cached) __atomic_store_n(&cache.v[dev], bytes, __ATOMIC_RELEASE);
    return e;      // (a device index past the cache: the attribute is set on every launch, still under the lock)
}
