// extern "C" surface of libivae_amd.so (declared in include/ivae_amd.h).
// Validates shapes on the host before any launch, never allocates device memory, never
// synchronises, never throws: every failure is an IVAE_E_* return code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "kernels.h"
#include "fused_train.h"

static thread_local char g_err[256] = "";

static int hip_fail(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return IVAE_E_HIP;
}

extern "C" {

const char* ivae_last_hip_error(void) { return g_err; }
#ifdef IVAE_DIAG
const char* ivae_version(void) { return "ivae_amd 0.3 (gfx950) DIAG build: timing switches and stamps, results may be wrong"; }
#else
const char* ivae_version(void) { return "ivae_amd 0.3 (gfx950)"; }
#endif

int ivae_param_count(const IvaeDims* dims, int64_t* out) {
    if (!out) return IVAE_E_NULLPTR;
    Layout L;
    int rc = ivae_make_layout(dims, false, &L);
    if (rc) return rc;
    *out = L.n_params;
    return IVAE_OK;
}

int ivae_workspace_bytes(const IvaeDims* dims, int32_t pairwise, size_t* out) {
    if (!out) return IVAE_E_NULLPTR;
    Layout L;
    int rc = ivae_make_layout(dims, pairwise != 0, &L);
    if (rc) return rc;
    *out = (size_t)L.ws.per_model * L.n_models * sizeof(float);
    return IVAE_OK;
}

int ivae_workspace_layout(const IvaeDims* dims, int32_t pairwise, IvaeWsLayout* out) {
    if (!out) return IVAE_E_NULLPTR;
    Layout L;
    int rc = ivae_make_layout(dims, pairwise != 0, &L);
    if (rc) return rc;
    *out = L.ws;
    return IVAE_OK;
}

static int check_coef(const IvaeDims* dims, const IvaeCoef* c) {
    if (!c) return IVAE_E_NULLPTR;
    if (c->global_batch < dims->batch || !(c->decoder_var > 0.f)) return IVAE_E_BADDIMS;
    if (c->pairwise && !(c->dataset_size > 0.f)) return IVAE_E_BADDIMS;
    return IVAE_OK;
}

int ivae_forward(const IvaeDims* dims, const IvaeCoef* coef, const float* params, const float* x, const float* u,
                 const float* eps, float* z_out, void* workspace, size_t workspace_bytes, void* stream) {
    Layout L;
    if (!coef) return IVAE_E_NULLPTR;
    int rc = ivae_make_layout(dims, coef->pairwise != 0, &L);
    if (rc) return rc;
    if ((rc = check_coef(dims, coef))) return rc;
    if (!params || !x || !eps || !z_out || !workspace || (L.M > 0 && !u)) return IVAE_E_NULLPTR;
    if (workspace_bytes < (size_t)L.ws.per_model * L.n_models * sizeof(float)) return IVAE_E_WORKSPACE;
    hipError_t e = launch_forward(L, *coef, params, x, u, eps, z_out, (float*)workspace, nullptr, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_forward");
}

// the pairwise kernels form element indices (row * latent_dim, slice geometry) in 32-bit integers
#define IVAE_PAIR_MAX_N (1 << 24)
int ivae_pairwise_workspace_bytes(int32_t n_rows, int32_t n_cols, int32_t latent_dim, size_t* out) {
    if (!out) return IVAE_E_NULLPTR;
    if (n_rows <= 0 || n_cols <= 0 || latent_dim <= 0) return IVAE_E_BADDIMS;
    if (latent_dim > 64 || n_rows > IVAE_PAIR_MAX_N || n_cols > IVAE_PAIR_MAX_N) return IVAE_E_TOOLARGE;
    *out = pairwise_ws_floats(n_rows, n_cols, latent_dim) * sizeof(float);
    return IVAE_OK;
}

int ivae_pairwise(int32_t n_rows, int32_t n_cols, int32_t latent_dim, const IvaeCoef* coef, const float* z,
                  const float* g_cols, const float* v_cols, float* lq, float* lqi, float* lse_s, float* lse_k,
                  float* rz, float* cg, float* cv, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_rows <= 0 || n_cols <= 0 || latent_dim <= 0) return IVAE_E_BADDIMS;
    if (latent_dim > 64 || n_rows > IVAE_PAIR_MAX_N || n_cols > IVAE_PAIR_MAX_N) return IVAE_E_TOOLARGE;
    if (!coef || !z || !g_cols || !v_cols || !lq || !lqi || !lse_s || !lse_k || !rz || !cg || !cv) return IVAE_E_NULLPTR;
    if (!(coef->dataset_size > 0.f) || coef->global_batch <= 0) return IVAE_E_BADDIMS;
    const size_t need = pairwise_ws_floats(n_rows, n_cols, latent_dim) * sizeof(float);
    if (need > 0 && !workspace) return IVAE_E_NULLPTR;
    if (workspace_bytes < need) return IVAE_E_WORKSPACE;
    hipError_t e = launch_pairwise(n_rows, n_cols, latent_dim, *coef, z, g_cols, v_cols, lq, lqi, lse_s, lse_k, rz, cg,
                                   cv, (float*)workspace, 1, 0, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_pairwise");
}

int ivae_backward(const IvaeDims* dims, const IvaeCoef* coef, const float* params, const float* x, const float* u,
                  const float* eps, float* loss_out, float* grads, void* workspace, size_t workspace_bytes,
                  void* stream) {
    Layout L;
    if (!coef) return IVAE_E_NULLPTR;
    int rc = ivae_make_layout(dims, coef->pairwise != 0, &L);
    if (rc) return rc;
    if ((rc = check_coef(dims, coef))) return rc;
    if (!params || !x || !eps || !grads || !workspace || (L.M > 0 && !u)) return IVAE_E_NULLPTR;
    if (workspace_bytes < (size_t)L.ws.per_model * L.n_models * sizeof(float)) return IVAE_E_WORKSPACE;
    hipError_t e = launch_backward(L, *coef, params, x, u, eps, loss_out, grads, (float*)workspace, false, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_backward");
}

// fuse != null: the slab reduction also applies the Adam update (*fused says whether it did)
static int fwd_bwd_impl(const IvaeDims* dims, const IvaeCoef* coef, const float* params, const float* x, const float* u,
                        const float* eps, float* z_out, float* loss_out, float* grads, void* workspace,
                        size_t workspace_bytes, void* stream, const AdamFuse* fuse, bool* fused) {
    Layout L;
    if (fused) *fused = false;
    if (!coef) return IVAE_E_NULLPTR;
    int rc = ivae_make_layout(dims, coef->pairwise != 0, &L);
    if (rc) return rc;
    if ((rc = check_coef(dims, coef))) return rc;
    if (!params || !x || !eps || !z_out || !grads || !workspace || (L.M > 0 && !u)) return IVAE_E_NULLPTR;
    if (workspace_bytes < (size_t)L.ws.per_model * L.n_models * sizeof(float)) return IVAE_E_WORKSPACE;
    if (coef->pairwise && coef->global_batch != dims->batch) return IVAE_E_BADDIMS;   // DP callers run the phases themselves
    // the decoder's backward rides in the decoder's forward launch (it needs nothing from other rows)
    int done = 0;               // plain ELBO: the row-chain kernel (where it serves the layout) also runs the encoder backward
    hipError_t e = launch_forward(L, *coef, params, x, u, eps, z_out, (float*)workspace, grads, (hipStream_t)stream, true, &done);
    if (e != hipSuccess) return hip_fail(e, "ivae_elbo_fwd_bwd");
    if (coef->pairwise) {       // all packed models in one launch sequence (their arrays repeat with the workspace stride)
        float* w = (float*)workspace;
        e = launch_pairwise(L.B, L.B, L.dl, *coef, w + L.ws.z, w + L.ws.g, w + L.ws.v, w + L.ws.lq, w + L.ws.lqi,
                            w + L.ws.lse_s, w + L.ws.lse_k, w + L.ws.rz, w + L.ws.cg, w + L.ws.cv, w + L.ws_pair,
                            L.n_models, (size_t)L.ws.per_model, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "ivae_elbo_fwd_bwd");
    }
    e = launch_backward(L, *coef, params, x, u, eps, loss_out, grads, (float*)workspace, true, (hipStream_t)stream, fuse, fused,
                        (done & 4) != 0);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_elbo_fwd_bwd");
}

int ivae_elbo_fwd_bwd(const IvaeDims* dims, const IvaeCoef* coef, const float* params, const float* x, const float* u,
                      const float* eps, float* z_out, float* loss_out, float* grads, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return fwd_bwd_impl(dims, coef, params, x, u, eps, z_out, loss_out, grads, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int ivae_adam_step(float* params, float* exp_avg, float* exp_avg_sq, const float* grads, int64_t n, double lr,
                   double beta1, double beta2, double eps, int64_t step, void* stream) {
    if (!params || !exp_avg || !exp_avg_sq || !grads) return IVAE_E_NULLPTR;
    if (n <= 0 || step < 1) return IVAE_E_BADDIMS;
    hipError_t e = launch_adam(params, exp_avg, exp_avg_sq, grads, n, lr, beta1, beta2, eps, step, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_adam_step");
}

int ivae_elbo_adam_steps(const IvaeDims* dims, const IvaeCoef* coef, const IvaeCoef* coef_steps, float* params,
                         float* exp_avg, float* exp_avg_sq,
                         const float* x_steps, const float* u_steps, const float* eps_steps, int32_t n_steps,
                         int64_t step0, double lr, const double* model_lr, double beta1, double beta2, double adam_eps, float* loss_out,
                         float* z_steps, float* grads, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = ivae_check_dims(dims);
    if (rc) return rc;
    if (n_steps <= 0 || step0 < 0) return IVAE_E_BADDIMS;
    if (!coef && !coef_steps) return IVAE_E_NULLPTR;
    if (!params || !exp_avg || !exp_avg_sq || !x_steps || !eps_steps || !loss_out || !z_steps || !grads ||
        (dims->aux_dim > 0 && !u_steps))
        return IVAE_E_NULLPTR;
    int64_t P = 0;
    if ((rc = ivae_param_count(dims, &P))) return rc;
    const size_t rows = (size_t)dims->n_models * dims->batch;
    AdamFuse F;
    F.p = params; F.m = exp_avg; F.v = exp_avg_sq;
    F.w1 = (float)(1.0 - beta1); F.b2 = (float)beta2; F.w2 = (float)(1.0 - beta2); F.eps = (float)adam_eps;
    const bool can_fuse = dims->n_models <= ADAM_MODELS_PER_LAUNCH;
    for (int s = 0; s < n_steps; ++s) {
        if (can_fuse) {          // same constants as launch_adam / launch_adam_models form for this step
            const double bc1 = 1.0 - pow(beta1, (double)(step0 + s + 1));
            F.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)(step0 + s + 1)));
            for (int i = 0; i < dims->n_models; ++i) F.neg_step[i] = (float)(-(model_lr ? model_lr[i] : lr) / bc1);
        }
        bool fused = false;
        rc = fwd_bwd_impl(dims, coef_steps ? coef_steps + s : coef, params, x_steps + s * rows * dims->data_dim,
                          u_steps ? u_steps + s * rows * dims->aux_dim : nullptr, eps_steps + s * rows * dims->latent_dim,
                          z_steps + s * rows * dims->latent_dim, loss_out + (size_t)s * dims->n_models, grads, workspace,
                          workspace_bytes, stream, can_fuse ? &F : nullptr, &fused);
        if (rc) return rc;
        if (fused) continue;     // the gradient reduction applied the update
        if (model_lr) {
            hipError_t e = launch_adam_models(params, exp_avg, exp_avg_sq, grads, P, dims->n_models, model_lr, beta1, beta2,
                                              adam_eps, step0 + s + 1, (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "ivae_elbo_adam_steps");
        } else {
            rc = ivae_adam_step(params, exp_avg, exp_avg_sq, grads, P * dims->n_models, lr, beta1, beta2, adam_eps, step0 + s + 1,
                                stream);
            if (rc) return rc;
        }
    }
    return IVAE_OK;
}

int ivae_train_steps_supported(const IvaeDims* d) {
    if (ivae_check_dims(d) != IVAE_OK) return 0;
    if (!(d->slope >= 0.f && d->slope <= 1.f)) return 0;      // the fused kernel computes lrelu as max(p, slope*p)
    return fused_supported(d->data_dim, d->latent_dim, d->aux_dim, d->hidden_dim, d->n_layers, d->activation, d->batch) ? 1 : 0;
}

static int g_fault = 0;       // ivae_debug_fault_inject

int ivae_debug_fault_inject(int32_t mask) {
    g_fault = mask;
    return IVAE_OK;
}

// ivae_debug_launch_events: events for the NEXT ivae_train_steps call of this thread (then forgotten)
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

int ivae_debug_launch_events(void* start_event, void* stop_event) {
    g_ev_start = (hipEvent_t)start_event;
    g_ev_stop = (hipEvent_t)stop_event;
    return IVAE_OK;
}

int ivae_train_steps(const IvaeDims* dims, const IvaeCoef* coef, float* params, float* exp_avg, float* exp_avg_sq,
                     const float* x_data, const float* u_data, int64_t n_data, const int32_t* batch_idx,
                     int64_t batch_idx_stride, const float* eps, uint64_t seed, int32_t n_steps, int64_t step0, double lr,
                     const double* model_lr, double beta1, double beta2, double adam_eps, float* loss_out, float* z_out,
                     float* z_steps, void* stream) {
    int rc = ivae_check_dims(dims);
    if (rc) return rc;
    if ((rc = check_coef(dims, coef))) return rc;
    if (!ivae_train_steps_supported(dims) || coef->pairwise || coef->global_batch != dims->batch) return IVAE_E_TOOLARGE;
    if (!params || !exp_avg || !exp_avg_sq || !x_data || !batch_idx || !loss_out || (dims->aux_dim > 0 && !u_data))
        return IVAE_E_NULLPTR;
    if (n_steps <= 0 || n_data <= 0 || step0 < 0) return IVAE_E_BADDIMS;
    if (n_data > 0x7fffffff) return IVAE_E_TOOLARGE;          // row indices are int32
    if (batch_idx_stride == 0) batch_idx_stride = (int64_t)n_steps * 64;
    if (batch_idx_stride < (int64_t)n_steps * 64) return IVAE_E_BADDIMS;
    if (model_lr) {           // the step table is formed for `lr`; a model's own rate enters as a factor on it
        if (!(lr != 0.0)) return IVAE_E_BADDIMS;
    }
    int64_t P = 0;
    if ((rc = ivae_param_count(dims, &P))) return rc;
    const size_t dd = dims->data_dim, dl = dims->latent_dim, da = dims->aux_dim;
    FusedArgs A;
    A.n_data = n_data; A.seed = seed;
    A.a = coef->a; A.b = coef->b; A.d = coef->d; A.decoder_var = coef->decoder_var; A.slope = dims->slope;
    A.w1 = (float)(1.0 - beta1); A.beta2 = (float)beta2; A.w2 = (float)(1.0 - beta2); A.adam_eps = (float)adam_eps;
    A.n_steps_total = n_steps;
    A.fault = g_fault;
    hipEvent_t ev_start = g_ev_start, ev_stop = g_ev_stop;       // the call's first launch starts the clock, its last one stops it
    g_ev_start = g_ev_stop = nullptr;
    for (int begin = 0; begin < n_steps; begin += FT_MAX_STEPS) {
        const int n = n_steps - begin < FT_MAX_STEPS ? n_steps - begin : FT_MAX_STEPS;
        A.n_steps = n; A.step_begin = begin; A.step0 = step0 + begin;
        for (int s = 0; s < n; ++s) {
            const double t = (double)(step0 + begin + s + 1);
            A.neg_step[s] = (float)(-lr / (1.0 - pow(beta1, t)));
            A.inv_bc2_sqrt[s] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
        }
        // a launch carries the per-model rate factors of at most FT_MAX_MODELS models: longer packs go in slices (every
        // per-model array advances by its own stride; the kernel numbers a slice's models from 0)
        for (int m0 = 0; m0 < dims->n_models; m0 += FT_MAX_MODELS) {
            const int nm = dims->n_models - m0 < FT_MAX_MODELS ? dims->n_models - m0 : FT_MAX_MODELS;
            const size_t m = (size_t)m0;
            A.params = params + m * P; A.exp_avg = exp_avg + m * P; A.exp_avg_sq = exp_avg_sq + m * P;
            A.x_data = x_data + m * n_data * dd;
            A.u_data = u_data ? u_data + m * n_data * da : nullptr;
            A.batch_idx = batch_idx + m * (size_t)batch_idx_stride;
            A.idx_stride = batch_idx_stride;
            A.eps = eps ? eps + m * n_steps * 64 * dl : nullptr;
            A.loss_out = loss_out + m * n_steps;
            A.z_steps = z_steps ? z_steps + m * n_steps * 64 * dl : nullptr;
            A.z_out = (begin + n == n_steps && z_out) ? z_out + m * 64 * dl : nullptr;
            A.model0 = m0;
            for (int i = 0; i < nm; ++i) A.lr_scale[i] = model_lr ? (float)(model_lr[m0 + i] / lr) : 1.f;
            for (int i = nm; i < FT_MAX_MODELS; ++i) A.lr_scale[i] = 1.f;
            const bool last = begin + n == n_steps && m0 + nm == dims->n_models;
            hipError_t e = launch_train_steps(dims->data_dim, dims->latent_dim, dims->aux_dim, dims->hidden_dim, A, nm,
                                              (hipStream_t)stream, ev_start, last ? ev_stop : nullptr);
            ev_start = nullptr;
            if (e != hipSuccess) return hip_fail(e, "ivae_train_steps");
        }
    }
    return IVAE_OK;
}

int ivae_corr_workspace_bytes(int64_t n, int32_t d, size_t* out) {
    if (!out) return IVAE_E_NULLPTR;
    if (n < 2 || d <= 0) return IVAE_E_BADDIMS;
    if (d > 64) return IVAE_E_TOOLARGE;
    *out = corr_ws_bytes(n, d);
    return IVAE_OK;
}

int ivae_corr(const float* x, const float* y, int64_t n, int32_t d, double* cc_out, void* workspace,
              size_t workspace_bytes, void* stream) {
    if (!x || !y || !cc_out || !workspace) return IVAE_E_NULLPTR;
    if (n < 2 || d <= 0) return IVAE_E_BADDIMS;
    if (d > 64) return IVAE_E_TOOLARGE;
    if (workspace_bytes < corr_ws_bytes(n, d)) return IVAE_E_WORKSPACE;
    hipError_t e = launch_corr(x, y, n, d, cc_out, (double*)workspace, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_corr");
}

int ivae_corr_static(const float* x, const float* y, int64_t n, int32_t d, double* cc_out, void* stream) {
    if (!x || !y || !cc_out) return IVAE_E_NULLPTR;
    if (n < 2 || d <= 0) return IVAE_E_BADDIMS;
    if (d > CORR_STATIC_MAX_D) {
        snprintf(g_err, sizeof(g_err), "ivae_corr_static: d = %d exceeds the built-in scratch (d <= %d); use ivae_corr", (int)d,
                 CORR_STATIC_MAX_D);
        return IVAE_E_TOOLARGE;
    }
    hipError_t e = launch_corr_static(x, y, n, d, cc_out, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_corr_static");
}

int ivae_corr_batched(const float* x, const float* y, int64_t n_blocks, int64_t n, int32_t d, double* cc_out,
                      void* stream) {
    if (!x || !y || !cc_out) return IVAE_E_NULLPTR;
    if (n_blocks <= 0 || n < 2 || d <= 0) return IVAE_E_BADDIMS;
    if (d > 64 || n_blocks > 0x7fffffff) return IVAE_E_TOOLARGE;
    hipError_t e = launch_corr_batched(x, y, n_blocks, n, d, cc_out, (hipStream_t)stream);
    return e == hipSuccess ? IVAE_OK : hip_fail(e, "ivae_corr_batched");
}

}  // extern "C"
