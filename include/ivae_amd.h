/*
 * ivae_amd.h -- C ABI of libivae_amd.so: the MI355X (gfx950) implementation of the
 * iVAE training hot path (ELBO forward/backward + Adam + MCC correlation).
 *
 * The reference (ilkhem/iVAE) is pure Python with no FFI; the interface it would bind
 * is the set of torch calls its nn.Module methods make.  Every entry point below names
 * the reference lines (relative to the reference root) it replaces.
 *
 * Conventions
 *   - all tensors are fp32, row-major (batch, feature), resident in device memory and
 *     owned by the caller; the library never allocates or frees device memory;
 *   - nn.Linear.weight is (out, in); a model's parameters travel as ONE flat fp32 arena
 *     in `parameters()` order: for each MLP in [logl (iVAE only), f, g, logv], for each
 *     layer: weight (out*in) then bias (out)           (models/nets.py:422-430, 99-106);
 *   - n_models > 1 = batched-model packing: every per-model array (params, grads, x, u,
 *     eps, z, loss, workspace) is the single-model array repeated with a fixed stride;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no internal sync;
 *   - return value 0 = ok, negative IVAE_E_* otherwise; never aborts, never prints.
 */
#ifndef IVAE_AMD_H
#define IVAE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { IVAE_ACT_NONE = 0, IVAE_ACT_LRELU = 1, IVAE_ACT_XTANH = 2, IVAE_ACT_SIGMOID = 3 };

enum {
    IVAE_OK = 0,
    IVAE_E_BADDIMS = -1,      /* non-positive / unsupported dimension */
    IVAE_E_BADACT = -2,       /* unknown activation id */
    IVAE_E_NULLPTR = -3,      /* a required pointer is NULL */
    IVAE_E_TOOLARGE = -4,     /* shape exceeds what the kernels' LDS tiling supports */
    IVAE_E_HIP = -5,          /* a HIP call failed; see ivae_last_hip_error() */
    IVAE_E_WORKSPACE = -6     /* workspace smaller than ivae_workspace_bytes() */
};

/* Shape of one model and of one call.  Mirrors the constructor arguments at
 * models/nets.py:125,190,392-393,550-551. */
typedef struct IvaeDims {
    int32_t batch;        /* rows in this call (the local shard under data parallelism) */
    int32_t data_dim;
    int32_t latent_dim;
    int32_t aux_dim;      /* 0 selects the VAE family: no prior MLP, prior variance 1 */
    int32_t hidden_dim;
    int32_t n_layers;     /* >= 1; n_layers-1 hidden layers of width hidden_dim */
    int32_t activation;   /* IVAE_ACT_* (models/nets.py:86-97) */
    float   slope;        /* lrelu negative slope / xtanh linear term */
    int32_t n_models;     /* >= 1 */
    /* Tuning of the shape-generic kernels; 0 = the library picks from the launch geometry.  They are part
     * of the shape because they decide the workspace layout: pass the SAME values to the size query and to the calls.
     * Every combination meets the same parity bar (tests/test_gpu_parity.py). */
    int32_t tile_rows;    /* 0 | 16 | 32 | 64: batch rows per workgroup */
    int32_t waves;        /* 0 | 4 | 8 | 16: waves per workgroup (a streamed operand chunk holds 16*waves columns) */
    int32_t wgrad_rows;   /* 0 | 16 | 32 | 64 | 128 | 256: batch rows per weight-gradient workgroup (sets the slab count) */
} IvaeDims;

/* Loss definition.  loss = -(1/Bg) * sum_i [ a*lpx - b*(lqc-lq) - c*(lq-lqi) - d*(lqi-lp) ]
 * (models/nets.py:184,242,471) when pairwise != 0, else -(1/Bg) * sum_i [a*lpx - b*lqc + d*lp]
 * (models/nets.py:475,599 with a=b=d=1).  Bg = global_batch. */
typedef struct IvaeCoef {
    float   a, b, c, d;
    float   dataset_size;   /* N in log(M*N), models/nets.py:181,468 */
    float   decoder_var;    /* .01 (nets.py:425,574) or .1 (nets.py:140,204) */
    int32_t pairwise;       /* 1: include the BxBxd total-correlation terms */
    int32_t global_batch;   /* batch of the whole step; == dims.batch unless data parallel */
} IvaeCoef;

/* Float offsets (per model) of the row tensors inside the workspace, for callers that
 * exchange them between ranks (data-parallel pairwise term) or read diagnostics. */
typedef struct IvaeWsLayout {
    int64_t per_model;      /* floats per model */
    int64_t g, v, l, z, f;  /* (batch, latent|data) forward tensors, nets.py:450-455 */
    int64_t lpx, lqc, lp, lq, lqi;   /* (batch,) per-row log-densities, nets.py:459-469 */
    int64_t rz, cg, cv;     /* (batch, latent) pairwise gradient terms */
    int64_t lse_s, lse_k;   /* (batch,), (batch, latent) log-sum-exp statistics */
} IvaeWsLayout;

/* ---- sizes -------------------------------------------------------------------------- */

/* Number of fp32 parameters of one model (sum over nn.Linear layers). */
int ivae_param_count(const IvaeDims* dims, int64_t* out);

/* Bytes of caller-provided scratch the calls below need for `dims` (all models) when they are given a coef with this
 * `pairwise` flag.  pairwise = 0: linear in the batch.  pairwise != 0: plus the pairwise kernels' scratch for a
 * (batch, batch, latent_dim) problem, which grows roughly as batch^2 * latent_dim / 341 -- so query with 0 for plain
 * calls (a full-dataset ivae_forward).  A workspace sized with pairwise != 0 also serves pairwise = 0 calls.
 * Layouts beyond 2^44 floats over all models return IVAE_E_TOOLARGE (the byte count is formed in size_t). */
int ivae_workspace_bytes(const IvaeDims* dims, int32_t pairwise, size_t* out);

/* Row-tensor offsets for the same (dims, pairwise); only per_model depends on `pairwise`. */
int ivae_workspace_layout(const IvaeDims* dims, int32_t pairwise, IvaeWsLayout* out);

/* ---- ELBO forward + backward ---------------------------------------------------------
 * Replaces  model.elbo(x,u[,N,a,b,c,d]) -> loss.backward()
 *   runners/ivae.py:76,82   models/wrappers.py:68-69,149-150
 *   models/nets.py:172-185 (cleanIVAE), 230-243 (cleanVAE), 457-475 (iVAE), 593-599 (VAE)
 * Inputs : params (n_models, P); x (n_models, batch, data_dim); u (n_models, batch, aux_dim)
 *          or NULL when aux_dim == 0; eps (n_models, batch, latent_dim) standard-normal draws
 *          (the torch.randn of nets.py:147,302).
 * Outputs: z_out (n_models, batch, latent_dim); loss_out (n_models,) = the quantity to
 *          minimise restricted to these rows (sums to the full loss over DP shards);
 *          grads (n_models, P) = d loss / d params, same layout as params.
 * Phases may also be called one by one (data-parallel pairwise term):
 *   ivae_forward -> [ivae_pairwise] -> ivae_backward. */
int ivae_elbo_fwd_bwd(const IvaeDims* dims, const IvaeCoef* coef, const float* params,
                      const float* x, const float* u, const float* eps,
                      float* z_out, float* loss_out, float* grads,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Forward only: models/nets.py:450-455, 587-591, 165-170, 224-228 plus the three per-row
 * log-densities (nets.py:459-461).  Fills the workspace row tensors and z_out. */
int ivae_forward(const IvaeDims* dims, const IvaeCoef* coef, const float* params,
                 const float* x, const float* u, const float* eps, float* z_out,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Pairwise log q(z_i | x_j) statistics and their gradient terms, never materialising the
 * (B,B,d) tensor of models/nets.py:180-182, 466-469.  Rows i: z (n_rows, d) of this shard.
 * Columns j: g_cols, v_cols (n_cols, d) of the GLOBAL batch.  Writes lq, lqi, rz (n_rows,*)
 * and cg, cv (n_cols, d) (column sums over this shard's rows).  Single model only.
 * workspace: at least ivae_pairwise_workspace_bytes(n_rows, n_cols, latent_dim) bytes of caller-owned
 * scratch (per-slice log-sum-exp statistics and gradient partial sums, merged in a fixed order).
 * Limits: latent_dim <= 64; n_rows, n_cols <= 2^24 (element indices are 32-bit): IVAE_E_TOOLARGE beyond. */
int ivae_pairwise_workspace_bytes(int32_t n_rows, int32_t n_cols, int32_t latent_dim, size_t* out);
int ivae_pairwise(int32_t n_rows, int32_t n_cols, int32_t latent_dim, const IvaeCoef* coef,
                  const float* z, const float* g_cols, const float* v_cols,
                  float* lq, float* lqi, float* lse_s, float* lse_k,
                  float* rz, float* cg, float* cv,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Backward of everything ivae_forward (+ ivae_pairwise) computed; autograd at
 * runners/ivae.py:82, models/wrappers.py:69. */
int ivae_backward(const IvaeDims* dims, const IvaeCoef* coef, const float* params,
                  const float* x, const float* u, const float* eps,
                  float* loss_out, float* grads,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- many steps on the shape-generic kernels, one call -----------------------------------
 * n_steps consecutive iterations of   zero_grad -> elbo -> backward -> Adam.step
 *   runners/ivae.py:75-91   models/wrappers.py:64-70
 * for ANY shape / activation / loss weights ivae_elbo_fwd_bwd takes, enqueued back to back by one call (the host
 * side of a training epoch is then one call, not one per step).  Step s reads batch s of
 *   x_steps (n_steps, n_models, batch, data_dim), u_steps (n_steps, n_models, batch, aux_dim) or NULL,
 *   eps_steps (n_steps, n_models, batch, latent_dim)
 * and writes loss_out (n_steps, n_models) and z_steps (n_steps, n_models, batch, latent_dim).
 * params / exp_avg / exp_avg_sq (n_models, P) in/out; grads (n_models, P): scratch for the flat gradient;
 * step0 = Adam steps already taken; lr is constant over the call.
 * coef_steps: NULL (the loss weights `coef` hold for every step), or n_steps loss definitions in HOST memory, read before
 * the call returns: step s runs with coef_steps[s] (`iVAE.anneal` changes a, b, c, d every iteration and drops the
 * pairwise terms after max_iter / 1.6 iterations: models/nets.py:477-486, called per step at models/wrappers.py:62-63).
 * When any of them has pairwise != 0 the workspace must be sized with pairwise = 1.
 * model_lr: NULL, or n_models learning rates in HOST memory that replace `lr` model by model (packed seeds of one
 * `main.py --n-sims K` run: each has its own ReduceLROnPlateau, runners/ivae.py:39,113-114); read before the call returns. */
int ivae_elbo_adam_steps(const IvaeDims* dims, const IvaeCoef* coef, const IvaeCoef* coef_steps,
                         float* params, float* exp_avg, float* exp_avg_sq,
                         const float* x_steps, const float* u_steps, const float* eps_steps,
                         int32_t n_steps, int64_t step0,
                         double lr, const double* model_lr, double beta1, double beta2, double adam_eps,
                         float* loss_out, float* z_steps, float* grads,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused persistent training ---------------------------------------------------------
 * n_steps consecutive iterations of   zero_grad -> elbo -> backward -> Adam.step
 *   runners/ivae.py:75-91   models/wrappers.py:64-70   (plain ELBO: models/nets.py:475,599)
 * for n_models independent models in ONE launch sequence: one workgroup per model keeps the
 * parameters in LDS and the Adam moments in registers across the steps; batches are gathered by
 * row index from datasets resident in device memory (DataLoader(shuffle=True) semantics are the
 * caller's choice of indices, runners/ivae.py:25).  Only shapes for which
 * ivae_train_steps_supported() returns 1 (3 layers, lrelu, batch 64, plain ELBO, LDS fit).
 *   params/exp_avg/exp_avg_sq (n_models, P) in/out;  x_data (n_models, n_data, data_dim);
 *   u_data (n_models, n_data, aux_dim) or NULL;  batch_idx (n_models, n_steps, 64) int32 rows -- or, with
 *   batch_idx_stride != 0, model m's table starts batch_idx_stride ints after model m-1's (the call may then point into
 *   an epoch's whole permutation table, (n_models, n_data), at the first step it is to run: no per-call slice);
 *   eps (n_models, n_steps, 64, latent_dim) or NULL -> in-kernel Philox4x32-10 + Box-Muller (seed);
 *   step0 = Adam steps already taken;  loss_out (n_models, n_steps);
 *   z_out (n_models, 64, latent_dim) of the LAST step, or NULL;
 *   z_steps (n_models, n_steps, 64, latent_dim) of EVERY step (the per-step z_est that
 *   models/wrappers.py:75 scores against the true sources), or NULL.
 * coef: a, b, d, decoder_var are used (pairwise must be 0, global_batch 64).
 * model_lr: NULL, or n_models learning rates in HOST memory, read before the call returns: model m trains with
 *   model_lr[m] instead of `lr` (every packed run has its own ReduceLROnPlateau, runners/ivae.py:39,113-114;
 *   models/wrappers.py:45,83); `lr` must then be non-zero (the step table is formed for it, a model's rate is a factor).
 * A hand-off between the waves of a model that times out inside the kernel (a protocol error, never seen) turns that
 * model's losses into NaN from that step on: check loss_out before trusting the parameters. */
int ivae_train_steps_supported(const IvaeDims* dims);
int ivae_train_steps(const IvaeDims* dims, const IvaeCoef* coef,
                     float* params, float* exp_avg, float* exp_avg_sq,
                     const float* x_data, const float* u_data, int64_t n_data,
                     const int32_t* batch_idx, int64_t batch_idx_stride, const float* eps, uint64_t seed,
                     int32_t n_steps, int64_t step0,
                     double lr, const double* model_lr, double beta1, double beta2, double adam_eps,
                     float* loss_out, float* z_out, float* z_steps, void* stream);

/* ---- Adam ---------------------------------------------------------------------------
 * Replaces optimizer.step() of torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8)
 *   runners/ivae.py:38,91   models/wrappers.py:44,70,127,151
 * over a flat arena of n floats.  `step` is the 1-based step count of THIS update.
 * lr is read per call (ReduceLROnPlateau mutates it, runners/ivae.py:39,113-114).  Hyper-parameters
 * are doubles, as torch holds them (1-beta2 must be formed in double to match). */
int ivae_adam_step(float* params, float* exp_avg, float* exp_avg_sq, const float* grads,
                   int64_t n, double lr, double beta1, double beta2, double eps, int64_t step,
                   void* stream);

/* ---- MCC correlation ------------------------------------------------------------------
 * Replaces np.corrcoef(x, y, rowvar=False)[:d, d:] of metrics/mcc.py:328 (float64
 * arithmetic on fp32 inputs).  x, y: (n, d) fp32; cc_out: (d, d) float64 on device.
 * workspace: at least ivae_corr_workspace_bytes(n, d) bytes.  The Hungarian assignment
 * (metrics/mcc.py:334) stays on the host.  A column of zero variance gives nan in its row / column of cc_out, as
 * np.corrcoef does (0 / 0; its clip to [-1, 1] keeps nan): the reference's runner scores such a batch 0
 * (runners/ivae.py:85-88). */
int ivae_corr_workspace_bytes(int64_t n, int32_t d, size_t* out);
int ivae_corr(const float* x, const float* y, int64_t n, int32_t d, double* cc_out,
              void* workspace, size_t workspace_bytes, void* stream);

/* ivae_corr without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  The scratch ivae_corr takes as `workspace` lives in a fixed array inside the library instead
 * (3,605,120 bytes per GPU, held until the process ends), which bounds the problem: d <= IVAE_CORR_STATIC_MAX_D (any n).
 * Larger d returns IVAE_E_TOOLARGE (ivae_last_hip_error() names the limit); use ivae_corr there.  Same kernels, same
 * arithmetic and summation order as ivae_corr: results are bit-identical.  The device that `stream` belongs to must be
 * current.  All calls on one GPU share that GPU's array: two calls that overlap on one GPU (different streams or host
 * threads without ordering between them) race; calls ordered on one stream are fine. */
#define IVAE_CORR_STATIC_MAX_D 40
int ivae_corr_static(const float* x, const float* y, int64_t n, int32_t d, double* cc_out, void* stream);

/* The same correlation block for n_blocks independent (n, d) problems stored back to back -- the per-step batch
 * scores of a training epoch (runners/ivae.py:86, models/wrappers.py:75 call mcc once per step on the batch):
 * x, y (n_blocks, n, d) fp32; cc_out (n_blocks, d, d) float64.  One launch, no workspace. */
int ivae_corr_batched(const float* x, const float* y, int64_t n_blocks, int64_t n, int32_t d,
                      double* cc_out, void* stream);

/* ---- diagnostics ---------------------------------------------------------------------- */
const char* ivae_last_hip_error(void);   /* text of the last failing HIP call (thread-local) */
const char* ivae_version(void);
/* Test hook, no counterpart in the reference: bit 0 makes ivae_train_steps's kernel skip one hand-off between the waves
 * of a model, so that the time-out report above can be exercised (tests/test_gpu_fused.py).  0 = normal operation. */
int ivae_debug_fault_inject(int32_t mask);
/* Measurement hook, no counterpart in the reference: the NEXT ivae_train_steps call of the calling thread records
 * `start_event` on its stream directly before its first kernel launch and `stop_event` directly after its last
 * (hipEvent_t handles created with timing; either may be NULL), so that the pair brackets the kernels and not the
 * host work of the call around them (argument checks, step tables: ~20 us that events recorded by the caller include;
 * bench.py's roofline figure).  The pair is used once. */
int ivae_debug_launch_events(void* start_event, void* stop_event);

#ifdef __cplusplus
}
#endif
#endif /* IVAE_AMD_H */
