// Arguments of the fused persistent train-step kernel (fused_train.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define FT_MAX_STEPS 128
#define FT_MAX_MODELS 256     // models per launch (a longer pack is launched in slices)

struct FusedArgs {
    float* params;            // (n_models, P) in/out
    float* exp_avg;           // (n_models, P) in/out
    float* exp_avg_sq;        // (n_models, P) in/out
    const float* x_data;      // (n_models, n_data, data_dim)
    const float* u_data;      // (n_models, n_data, aux_dim) or null
    const int* batch_idx;     // model m's rows of step s: batch_idx[m * idx_stride + s * 64 + 0..63]
    int64_t idx_stride;       // ints between the tables of consecutive models
    const float* eps;         // (n_models, n_steps, 64, latent) or null -> Philox(seed)
    float* loss_out;          // (n_models, n_steps)
    float* z_out;             // (n_models, 64, latent) of the last step, or null
    float* z_steps;           // (n_models, n_steps, 64, latent) of every step, or null
    int64_t n_data;
    unsigned long long seed;
    int64_t step0;            // Adam steps done before this launch
    int n_steps;              // steps in THIS launch (<= FT_MAX_STEPS)
    int n_steps_total;        // step extent of batch_idx / eps / loss_out
    int step_begin;           // first step of this launch inside that extent
    float a, b, d, decoder_var, slope;
    float w1, beta2, w2, adam_eps;      // 1-beta1, beta2, 1-beta2, eps
    float neg_step[FT_MAX_STEPS];       // -lr / (1 - beta1^t)
    float inv_bc2_sqrt[FT_MAX_STEPS];   // 1 / sqrt(1 - beta2^t)
    float lr_scale[FT_MAX_MODELS];      // model_lr[m] / lr (1 when the pack shares one learning rate)
    int model0;                         // number of this launch's first model inside the pack (Philox stream id)
    int fault;                          // fault injection for tests (ivae_debug_fault_inject); 0 in normal use
};

bool fused_supported(int dd, int dl, int m, int h, int n_layers, int act, int batch);
// ev_start / ev_stop (either may be null): events recorded on `st` directly before / after the launch
hipError_t launch_train_steps(int dd, int dl, int m, int h, const FusedArgs& A, int n_models, hipStream_t st,
                              hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
