// Fused persistent train-step kernel: one 256-thread workgroup (4 waves, one per SIMD) trains one model for n_steps
// consecutive SGD steps without leaving the CU.
//
//   per step:  gather batch rows by index -> encoder/prior/decoder forward (fp32 MFMA) ->
//              reparameterisation + Gaussian log-densities + loss -> analytic backward
//              (dgrad + wgrad on MFMA) -> Adam on every layer             (plain ELBO)
//
// Reference sequence replaced (per step): runners/ivae.py:75-91 / models/wrappers.py:64-70 with
// models/nets.py:450-475 (iVAE) or 587-599 (VAE) and torch.optim.Adam.
//
// On-chip residency:
//   * parameter arena: LDS, exactly the torch `parameters()` layout (weights (out,in) row-major,
//     then bias), so a weight row stride is its in-dim (50, 10: = 2 mod 4 -> conflict-free
//     operand reads);
//   * Adam exp_avg / exp_avg_sq: registers, flat per-layer ownership (thread t owns elements
//     t, t+256, ... of each layer), loaded once per launch, written back once;
//   * activations: five (64, SA) LDS tiles + four (64, SS) row tensors; hidden activations the
//     backward pass needs later are stashed in the producing wave's registers (MFMA C layout)
//     and written back to LDS when their MLP's backward starts: no recompute, no HBM round trip;
//   * weight gradients of one layer at a time are staged in an LDS tile and consumed by Adam
//     right away: no gradient arena exists.
// Every layer product (fwd, dgrad, wgrad) is v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) on
// 16x16 tiles with operands read from LDS; contraction dims are zero-padded to a multiple of 4,
// output tiles past the real dims are discarded in the epilogues.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"
#include "fused_train.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (make prof): s_memtime stamps at phase boundaries of model 0 / wave 0 go to a
// buffer of their own; no output depends on them and the shipped library contains none.
#ifdef FT_PROF
__device__ unsigned long long* g_ft_prof = nullptr;
#define FT_NSTAMP 64
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_ft_prof) {                                    \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_ft_prof[(size_t)s * FT_NSTAMP + (i)] = t_;                                           \
        }                                                                                          \
    } while (0)
#else
#define STAMP(i)
#endif

#define FT_THREADS 256
#define FT_B 64
#define LOG2PI_F 1.8378770664093453f

__host__ __device__ constexpr int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------- config
template <int DD_, int DL_, int M_, int H_>
struct FCfg {
    static constexpr int DD = DD_, DL = DL_, M = M_, H = H_, KIN = DD_ + M_;
    static constexpr bool HAS_U = M_ > 0;
    // arena (torch order): [logl] f g logv ; per MLP: W1 b1 W2 b2 W3 b3
    static constexpr int P_L0 = HAS_U ? H * M + H : 0, P_L1 = HAS_U ? H * H + H : 0, P_L2 = HAS_U ? DL * H + DL : 0;
    static constexpr int P_F0 = H * DL + H, P_F1 = H * H + H, P_F2 = DD * H + DD;
    static constexpr int P_E0 = H * KIN + H, P_E1 = H * H + H, P_E2 = DL * H + DL;
    static constexpr int OFF_L0 = 0, OFF_L1 = OFF_L0 + P_L0, OFF_L2 = OFF_L1 + P_L1;
    static constexpr int OFF_F0 = OFF_L2 + P_L2, OFF_F1 = OFF_F0 + P_F0, OFF_F2 = OFF_F1 + P_F1;
    static constexpr int OFF_G0 = OFF_F2 + P_F2, OFF_G1 = OFF_G0 + P_E0, OFF_G2 = OFF_G1 + P_E1;
    static constexpr int OFF_V0 = OFF_G2 + P_E2, OFF_V1 = OFF_V0 + P_E0, OFF_V2 = OFF_V1 + P_E1;
    static constexpr int NPARAM = OFF_V2 + P_E2;
    // Adam register slots per layer (flat ownership: thread t owns t, t+256, ...)
    static constexpr int C_L0 = cdiv(P_L0, FT_THREADS), C_L1 = cdiv(P_L1, FT_THREADS), C_L2 = cdiv(P_L2, FT_THREADS);
    static constexpr int C_F0 = cdiv(P_F0, FT_THREADS), C_F1 = cdiv(P_F1, FT_THREADS), C_F2 = cdiv(P_F2, FT_THREADS);
    static constexpr int C_E0 = cdiv(P_E0, FT_THREADS), C_E1 = cdiv(P_E1, FT_THREADS), C_E2 = cdiv(P_E2, FT_THREADS);
    static constexpr int R_L0 = 0, R_L1 = R_L0 + C_L0, R_L2 = R_L1 + C_L1;
    static constexpr int R_F0 = R_L2 + C_L2, R_F1 = R_F0 + C_F0, R_F2 = R_F1 + C_F1;
    static constexpr int R_G0 = R_F2 + C_F2, R_G1 = R_G0 + C_E0, R_G2 = R_G1 + C_E1;
    static constexpr int R_V0 = R_G2 + C_E2, R_V1 = R_V0 + C_E0, R_V2 = R_V1 + C_E1;
    static constexpr int NADAM = R_V2 + C_E2;
    // LDS activation tiles
    static constexpr int WMAX = cmax(cmax(H, KIN), cmax(DD, DL));
    static constexpr int SA = pad4(WMAX) + 2;                 // = 2 (mod 4): conflict-free row-strided reads
    static constexpr int SS = pad4(DL) + 2;
    static constexpr int ARENA_PAD = (NPARAM + 3) & ~3;
    static constexpr int TILE = FT_B * SA;
    static constexpr int SMALL = FT_B * SS;
    static constexpr int LDS_FLOATS = ARENA_PAD + 5 * TILE + 4 * SMALL + 96;   // +96: column sums, loss partials, slack for tile over-reads
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    static_assert(H <= 64 && KIN <= 64 && DD <= 16 && DL <= 16, "tile plan assumes <=64 wide hidden/input, <=16 wide heads");
    static_assert(LDS_BYTES <= 160 * 1024, "does not fit the 160 KB LDS");
};

// ------------------------------------------------------------------------------ primitives
// Lane id behind an optimisation barrier: keeps LICM from hoisting every phase's per-lane LDS
// addresses out of the step loop (dozens of phases x several VGPRs each -> spills).
__device__ __forceinline__ int opaque_lane() {
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
}

// Workgroup barrier that waits only for this wave's LDS traffic.  All cross-wave hand-offs inside the
// step loop go through LDS; __syncthreads() would also drain vmcnt and so expose the latency of the
// next batch's prefetch (global loads into registers that nothing reads until the next step).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[m][n] += A(rows i) x B(cols j) over ksteps*4; operands in LDS.
// A element (i,k) at A[i*a_si + k*a_sk], B element (k,j) at Bp[j*b_sj + k*b_sk]; pointers already at
// the block origin.  Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15].
template <int BM, int BN>
__device__ __forceinline__ void mma_block(const float* A, int a_si, int a_sk, const float* Bp, int b_sj, int b_sk,
                                          int ksteps, f32x4 (&acc)[BM][BN]) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const float* ap = A + c * a_si + q * a_sk;
    const float* bp = Bp + c * b_sj + q * b_sk;
#pragma unroll 2
    for (int s = 0; s < ksteps; ++s) {
        float a[BM], b[BN];
#pragma unroll
        for (int m = 0; m < BM; ++m) a[m] = ap[m * 16 * a_si];
#pragma unroll
        for (int n = 0; n < BN; ++n) b[n] = bp[n * 16 * b_sj];
#pragma unroll
        for (int m = 0; m < BM; ++m)
#pragma unroll
            for (int n = 0; n < BN; ++n) acc[m][n] = mfma4(a[m], b[n], acc[m][n]);
        ap += 4 * a_sk;
        bp += 4 * b_sk;
    }
}

__device__ __forceinline__ float lrelu(float p, float slope) { return p > 0.f ? p : p * slope; }

// Forward layer block: Y[b][n] = epi(bias[n] + sum_k X[b][k] W[n][k]); C rows = b, cols = n.
// EPI 0: lrelu -> out (stride so), columns >= N written as 0 (keeps contraction padding zero)
// EPI 1: identity -> out      EPI 2: exp -> out
template <int BM, int BN, int EPI>
__device__ __forceinline__ void fwd_block(const float* X, int sx, const float* W, int K, const float* bias, int N,
                                          float* out, int so, int row0, int col0, float slope, f32x4 (&acc)[BM][BN]) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int n = 0; n < BN; ++n) {
        const int col = col0 + n * 16 + c;
        const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < BM; ++m) acc[m][n] = (f32x4){bv, bv, bv, bv};
    }
    mma_block<BM, BN>(X + row0 * sx, sx, 1, W + col0 * K, K, 1, pad4(K) / 4, acc);
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) {
            const int col = col0 + n * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + m * 16 + 4 * q + r;
                float val = acc[m][n][r];
                if (EPI == 0) val = lrelu(val, slope);
                else if (EPI == 2) val = expf(val);
                if (col >= N) val = 0.f;
                acc[m][n][r] = val;
                if (col < so) out[row * so + col] = val;
            }
        }
}

// Write a stashed block (post-epilogue values kept in registers) back to an LDS tile.
template <int BM, int BN>
__device__ __forceinline__ void restore_block(const f32x4 (&st)[BM][BN], float* out, int so, int row0, int col0) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) {
            const int col = col0 + n * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col < so) out[(row0 + m * 16 + 4 * q + r) * so + col] = st[m][n][r];
        }
}

// dgrad block: dX[b][k] = mask(h[b][k]) * sum_n dY[b][n] W[n][k]; C rows = b, cols = k.
// MASK 0: none (input of the MLP), 1: lrelu mask from post-activation tile `hm` (stride sh).
template <int BM, int BN, int MASK>
__device__ __forceinline__ void dgrad_block(const float* dY, int sy, int N, const float* W, int K, const float* hm,
                                            int sh, float* out, int so, int row0, int col0, float slope) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    f32x4 acc[BM][BN];
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A(i=b, k=n) = dY[b][n]; B(k=n, j=k) = W[n][k]
    mma_block<BM, BN>(dY + row0 * sy, sy, 1, W + col0, 1, K, pad4(N) / 4, acc);
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) {
            const int col = col0 + n * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + m * 16 + 4 * q + r;
                float val = acc[m][n][r];
                if (col >= K) val = 0.f;
                else if (MASK == 1) val *= (hm[row * sh + col] > 0.f ? 1.f : slope);
                if (col < so) out[row * so + col] = val;
            }
        }
}

// wgrad block: dW[n][k] = sum_b dY[b][n] X[b][k]; C rows = n, cols = k; written to staging [n*K + k].
template <int BM, int BN>
__device__ __forceinline__ void wgrad_block(const float* dY, int sy, int N, const float* X, int sx, int K,
                                            float* stage, int row0, int col0) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    f32x4 acc[BM][BN];
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A(i=n, k=b) = dY[b][n]; B(k=b, j=k) = X[b][k]
    mma_block<BM, BN>(dY + row0, 1, sy, X + col0, 1, sx, FT_B / 4, acc);
#pragma unroll
    for (int m = 0; m < BM; ++m)
#pragma unroll
        for (int n = 0; n < BN; ++n) {
            const int col = col0 + n * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + m * 16 + 4 * q + r;
                if (row < N && col < K) stage[row * K + col] = acc[m][n][r];
            }
        }
}

// Column sums of a (64, >=64-wide or narrow) delta tile = the bias gradient of its layer.  Wave w owns
// columns 16w..16w+15; lane = (row group of 16 rows) x (column); two xor-shuffles finish the sum.
__device__ __forceinline__ void colsum64(const float* dY, int sy, int ncols, float* cs, int w) {
    const int lane = opaque_lane();
    const int c = lane & 15, rg = lane >> 4;
    if (16 * w < ncols) {
        const float* p = dY + (rg * 16) * sy + 16 * w + c;
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 16; ++b) sum += p[b * sy];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (rg == 0) cs[16 * w + c] = sum;
    }
}

// Adam for one layer: gradient from the staging tile (weights) / the column sums `cs` (bias).
// Same update as k_adam (adam.hip) / torch.optim.Adam; sqrt and the final divide use the hardware
// v_sqrt_f32 / v_rcp_f32 (1 ulp) -- a <=2e-7 relative change of a step of size ~lr.
template <int P, int NK, int CNT, int REG0, int NA>
__device__ __forceinline__ void adam_layer(float* arena_layer, const float* stage, const float* cs,
                                           float (&am)[NA], float (&av)[NA], float w1, float b2, float w2,
                                           float neg_step, float inv_bc2, float eps) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const int idx = threadIdx.x + j * FT_THREADS;
        if (idx < P) {
            const float g = idx < NK ? stage[idx] : cs[idx - NK];
            const float m0 = am[REG0 + j], v0 = av[REG0 + j];
            const float mi = m0 + w1 * (g - m0);
            const float vi = v0 * b2 + (w2 * g) * g;
            const float denom = __builtin_amdgcn_sqrtf(vi) * inv_bc2 + eps;
            arena_layer[idx] = arena_layer[idx] + (neg_step * mi) * __builtin_amdgcn_rcpf(denom);
            am[REG0 + j] = mi;
            av[REG0 + j] = vi;
        }
    }
}

template <int P, int CNT, int REG0, int NA>
__device__ __forceinline__ void adam_load(const float* gm, const float* gv, float (&am)[NA], float (&av)[NA]) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const int idx = threadIdx.x + j * FT_THREADS;
        am[REG0 + j] = idx < P ? gm[idx] : 0.f;
        av[REG0 + j] = idx < P ? gv[idx] : 0.f;
    }
}

template <int P, int CNT, int REG0, int NA>
__device__ __forceinline__ void adam_store(float* gm, float* gv, const float (&am)[NA], const float (&av)[NA]) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const int idx = threadIdx.x + j * FT_THREADS;
        if (idx < P) { gm[idx] = am[REG0 + j]; gv[idx] = av[REG0 + j]; }
    }
}

// Philox4x32-10 (Salmon et al. 2011) -> 4 standard normals via Box-Muller.
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2,
                                               unsigned c3, float (&out)[4]) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u3 = ((float)(c3 >> 8) + 0.5f) * (1.f / 16777216.f);
    // Box-Muller on the hardware transcendental units: v_log_f32 (log2), v_sin/v_cos (argument in turns)
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    const float s0 = __builtin_amdgcn_sinf(u1), c0f = __builtin_amdgcn_cosf(u1);
    const float s1 = __builtin_amdgcn_sinf(u3), c1f = __builtin_amdgcn_cosf(u3);
    out[0] = r0 * c0f; out[1] = r0 * s0; out[2] = r1 * c1f; out[3] = r1 * s1;
}

// Gather one batch [x | u] (zero-padded to SA columns) into registers, element e = tid + j*256, in two
// stages so that neither load latency is exposed: the row indices of the slots (issued one phase
// earlier), then the elements (branch-free: out-of-range slots read a dummy address and are zeroed).
template <class C>
__device__ __forceinline__ void load_rows(int (&rows)[cdiv(FT_B * C::SA, FT_THREADS)], const int* idx) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));      // keep the per-thread (row, column) decode out of the step loop's live set
#pragma unroll
    for (int j = 0; j < cdiv(FT_B * C::SA, FT_THREADS); ++j) {
        const int b = (t + j * FT_THREADS) / C::SA;
        rows[j] = idx[b < FT_B ? b : FT_B - 1];
    }
}

template <class C>
__device__ __forceinline__ void load_xu(float (&xu)[cdiv(FT_B * C::SA, FT_THREADS)],
                                        const int (&rows)[cdiv(FT_B * C::SA, FT_THREADS)], const float* xd,
                                        const float* ud) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int j = 0; j < cdiv(FT_B * C::SA, FT_THREADS); ++j) {
        const int e = t + j * FT_THREADS;
        const int b = e / C::SA, k = e - b * C::SA;
        const bool valid = b < FT_B && k < C::KIN;
        const size_t row = (size_t)rows[j];
        const float* p = k < C::DD ? xd + row * C::DD + k : ud + row * C::M + (k - C::DD);
        const float val = *(valid ? p : xd);
        xu[j] = valid ? val : 0.f;
    }
}

// ---------------------------------------------------------------------------------- kernel
// 4 waves, one per SIMD (512-register budget each: Adam moments + stashed activations stay in
// registers without spilling).  Every wave runs the same phase sequence on its own output blocks:
//   2x2-tile block (r22, c22) of a 64x64 output, or row/column tile `rt` of a 16-wide head.
// LDS tiles: A0..A3 activations / deltas, A4 = XU (forward) then gradient staging (backward).
template <class C>
__global__ __launch_bounds__(FT_THREADS, 1) void k_train_steps(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* arena = lds;
    float* A0 = arena + C::ARENA_PAD;
    float* A1 = A0 + C::TILE;
    float* A2 = A1 + C::TILE;
    float* A3 = A2 + C::TILE;
    float* A4 = A3 + C::TILE;
    float* sG = A4 + C::TILE;
    float* sV = sG + C::SMALL;
    float* sL = sV + C::SMALL;
    float* sZ = sL + C::SMALL;
    float* csum = sZ + C::SMALL;         // 64 column sums (bias gradients); the slack also absorbs tile over-reads
    float* red = csum + 64;              // 4 loss partials

    constexpr int SA = C::SA, SS = C::SS, H = C::H, DD = C::DD, DL = C::DL, M = C::M, KIN = C::KIN;
    const int tid = threadIdx.x;
    const int w_init = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int model = blockIdx.x;
    float* gP = A.params + (size_t)model * C::NPARAM;
    float* gM = A.exp_avg + (size_t)model * C::NPARAM;
    float* gV = A.exp_avg_sq + (size_t)model * C::NPARAM;
    const float* xd = A.x_data + (size_t)model * A.n_data * DD;
    const float* ud = C::HAS_U ? A.u_data + (size_t)model * A.n_data * M : A.x_data;
    const int* idxp = A.batch_idx + ((size_t)model * A.n_steps_total + A.step_begin) * FT_B;
    const float slope = A.slope;
    const float wB = 1.f / (float)FT_B;

    // ---- load state: zero LDS (finite padding), arena -> LDS, Adam moments -> registers
    for (int i = tid; i < C::LDS_FLOATS; i += FT_THREADS) lds[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < C::NPARAM; i += FT_THREADS) arena[i] = gP[i];
    float am[C::NADAM], av[C::NADAM];
    if (C::HAS_U) {
        adam_load<C::P_L0, C::C_L0, C::R_L0>(gM + C::OFF_L0, gV + C::OFF_L0, am, av);
        adam_load<C::P_L1, C::C_L1, C::R_L1>(gM + C::OFF_L1, gV + C::OFF_L1, am, av);
        adam_load<C::P_L2, C::C_L2, C::R_L2>(gM + C::OFF_L2, gV + C::OFF_L2, am, av);
    }
    adam_load<C::P_F0, C::C_F0, C::R_F0>(gM + C::OFF_F0, gV + C::OFF_F0, am, av);
    adam_load<C::P_F1, C::C_F1, C::R_F1>(gM + C::OFF_F1, gV + C::OFF_F1, am, av);
    adam_load<C::P_F2, C::C_F2, C::R_F2>(gM + C::OFF_F2, gV + C::OFF_F2, am, av);
    adam_load<C::P_E0, C::C_E0, C::R_G0>(gM + C::OFF_G0, gV + C::OFF_G0, am, av);
    adam_load<C::P_E1, C::C_E1, C::R_G1>(gM + C::OFF_G1, gV + C::OFF_G1, am, av);
    adam_load<C::P_E2, C::C_E2, C::R_G2>(gM + C::OFF_G2, gV + C::OFF_G2, am, av);
    adam_load<C::P_E0, C::C_E0, C::R_V0>(gM + C::OFF_V0, gV + C::OFF_V0, am, av);
    adam_load<C::P_E1, C::C_E1, C::R_V1>(gM + C::OFF_V1, gV + C::OFF_V1, am, av);
    adam_load<C::P_E2, C::C_E2, C::R_V2>(gM + C::OFF_V2, gV + C::OFF_V2, am, av);
    __syncthreads();

    const float w1 = A.w1, b2 = A.beta2, w2 = A.w2, aeps = A.adam_eps;
    constexpr int NXU = cdiv(FT_B * C::SA, FT_THREADS);
    float xu[NXU];
    int rows[NXU];
    load_rows<C>(rows, idxp);
    load_xu<C>(xu, rows, xd, ud);

    for (int s = 0; s < A.n_steps; ++s) {
        // block coordinates (wave-uniform), re-derived per step behind an optimisation barrier so the
        // scalar/vector address arithmetic of ~30 phases is not hoisted out of the step loop
        int w = w_init;
        asm volatile("" : "+s"(w));
        const int lane = opaque_lane();
        const int r22 = (w >> 1) * 32, c22 = (w & 1) * 32;
        const int rt = w * 16;
        const float neg_step = A.neg_step[s], inv_bc2 = A.inv_bc2_sqrt[s];
        const int* idx = idxp + (size_t)s * FT_B;

        STAMP(0);
        // ---- P0: the batch [x | u] was prefetched into registers during the previous step -> A4; eps -> sZ
#pragma unroll
        for (int j = 0; j < NXU; ++j) { const int e = tid + j * FT_THREADS; if (e < FT_B * SA) A4[e] = xu[j]; }
        if (A.eps) {
            for (int e = tid; e < FT_B * SS; e += FT_THREADS) {
                const int b = e / SS, k = e - b * SS;
                if (k < DL) sZ[e] = A.eps[(((size_t)model * A.n_steps_total + A.step_begin + s) * FT_B + b) * DL + k];
            }
        } else if (tid * 4 < FT_B * DL) {
            float nrm[4];
            philox_normal4(A.seed, (unsigned)tid, (unsigned)(A.step0 + s), (unsigned)model, 0x1234567u, nrm);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int el = tid * 4 + i; sZ[(el / DL) * SS + el % DL] = nrm[i]; }
        }
        lds_barrier(); STAMP(1);

        // ---- encoder g / logv layers 1-2; outputs stashed in registers for the backward pass
        f32x4 st1g[2][2], st1v[2][2], st2g[2][2], st2v[2][2], st3[2][2], st4[2][2];
        fwd_block<2, 2, 0>(A4, SA, arena + C::OFF_G0, KIN, arena + C::OFF_G0 + H * KIN, H, A0, SA, r22, c22, slope, st1g);
        fwd_block<2, 2, 0>(A4, SA, arena + C::OFF_V0, KIN, arena + C::OFF_V0 + H * KIN, H, A1, SA, r22, c22, slope, st1v);
        lds_barrier(); STAMP(2);
        fwd_block<2, 2, 0>(A0, SA, arena + C::OFF_G1, H, arena + C::OFF_G1 + H * H, H, A2, SA, r22, c22, slope, st2g);
        fwd_block<2, 2, 0>(A1, SA, arena + C::OFF_V1, H, arena + C::OFF_V1 + H * H, H, A3, SA, r22, c22, slope, st2v);
        lds_barrier(); STAMP(3);
        // ---- heads g -> sG, v = exp(logv) -> sV (row tile rt); prior layer 1 shares the phase (A0 is free)
        {
            f32x4 acc[1][1];
            fwd_block<1, 1, 1>(A2, SA, arena + C::OFF_G2, H, arena + C::OFF_G2 + DL * H, DL, sG, SS, rt, 0, slope, acc);
            fwd_block<1, 1, 2>(A3, SA, arena + C::OFF_V2, H, arena + C::OFF_V2 + DL * H, DL, sV, SS, rt, 0, slope, acc);
        }
        if (C::HAS_U) {
            fwd_block<2, 2, 0>(A4 + DD, SA, arena + C::OFF_L0, M, arena + C::OFF_L0 + H * M, H, A0, SA, r22, c22, slope, st3);
            lds_barrier(); STAMP(4);
            fwd_block<2, 2, 0>(A0, SA, arena + C::OFF_L1, H, arena + C::OFF_L1 + H * H, H, A1, SA, r22, c22, slope, st4);
            lds_barrier(); STAMP(5);
            f32x4 acc[1][1];
            fwd_block<1, 1, 2>(A1, SA, arena + C::OFF_L2, H, arena + C::OFF_L2 + DL * H, DL, sL, SS, rt, 0, slope, acc);
        }
        lds_barrier(); STAMP(6);
        // ---- Z: reparameterisation + latent log-densities (nets.py:301-314)
        float lsum = 0.f;
        for (int e = tid; e < FT_B * SS; e += FT_THREADS) {
            const int b = e / SS, k = e - b * SS;
            float zz = 0.f;
            if (k < DL) {
                const float ep = sZ[e];
                const float gg = sG[e], vv = sV[e];
                const float ll = C::HAS_U ? sL[e] : 1.f;
                zz = ep * sqrtf(vv) + gg;
                const float dq = zz - gg;
                const float lqc = -0.5f * (LOG2PI_F + logf(vv) + dq * dq / vv);
                const float lp = -0.5f * (LOG2PI_F + logf(ll) + zz * zz / ll);
                lsum += A.d * lp - A.b * lqc;
                if (!C::HAS_U) sL[e] = 1.f;
                if (A.z_out && s == A.n_steps - 1) A.z_out[((size_t)model * FT_B + b) * DL + k] = zz;
            }
            sZ[e] = zz;
        }
        lds_barrier(); STAMP(7);
        // ---- decoder f: z -> h1f (A0) -> h2f (A1) -> f; dOut = dLoss/df -> A2, log p(x|z)
        {
            f32x4 acc[2][2];
            fwd_block<2, 2, 0>(sZ, SS, arena + C::OFF_F0, DL, arena + C::OFF_F0 + H * DL, H, A0, SA, r22, c22, slope, acc);
            lds_barrier(); STAMP(8);
            fwd_block<2, 2, 0>(A0, SA, arena + C::OFF_F1, H, arena + C::OFF_F1 + H * H, H, A1, SA, r22, c22, slope, acc);
            lds_barrier(); STAMP(9);
        }
        {
            // A2 becomes the zero-padded dOut tile: columns >= 16 zeroed by everyone, the head tile written below
            for (int e = tid; e < FT_B * SA; e += FT_THREADS) { const int k = e % SA; if (k >= 16) A2[e] = 0.f; }
            f32x4 acc[1][1];
            const int c = lane & 15, q = lane >> 4;
            const float bv = c < DD ? arena[C::OFF_F2 + DD * H + c] : 0.f;
            acc[0][0] = (f32x4){bv, bv, bv, bv};
            mma_block<1, 1>(A1 + rt * SA, SA, 1, arena + C::OFF_F2, H, 1, pad4(H) / 4, acc);
            const float dv = A.decoder_var, logdv = logf(A.decoder_var);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt + 4 * q + r;
                float dout = 0.f;
                if (c < DD) {
                    const float df = A4[row * SA + c] - acc[0][0][r];        // x - f
                    lsum += A.a * (-0.5f * (LOG2PI_F + logdv + df * df / dv));
                    dout = -A.a * wB * df / dv;
                }
                A2[row * SA + c] = dout;
            }
        }
        // loss reduction (deterministic): wave shuffle -> 4 partials -> thread 0
        {
            float v = lsum;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) red[w] = v;
        }
        lds_barrier(); STAMP(10);
        if (tid == 0) {
            const float t = (red[0] + red[1]) + (red[2] + red[3]);
            A.loss_out[(size_t)model * A.n_steps_total + A.step_begin + s] = -t * wB;
        }

#define ADAM(P_, NK_, CNT_, REG_, OFF_, DY_, SY_) \
        adam_layer<P_, NK_, CNT_, REG_>(arena + OFF_, A4, csum, am, av, w1, b2, w2, neg_step, inv_bc2, aeps)

        // ---- decoder backward.  layer 3: d2f -> A3
        dgrad_block<2, 2, 1>(A2, SA, DD, arena + C::OFF_F2, H, A1, SA, A3, SA, r22, c22, slope);
        wgrad_block<1, 1>(A2, SA, DD, A1, SA, H, A4, 0, rt);
        colsum64(A2, SA, DD, csum, w);
        lds_barrier(); STAMP(11);
        ADAM(C::P_F2, DD * H, C::C_F2, C::R_F2, C::OFF_F2, A2, SA);
        lds_barrier(); STAMP(12);
        // layer 2: d1f -> A1
        dgrad_block<2, 2, 1>(A3, SA, H, arena + C::OFF_F1, H, A0, SA, A1, SA, r22, c22, slope);
        wgrad_block<2, 2>(A3, SA, H, A0, SA, H, A4, r22, c22);
        colsum64(A3, SA, H, csum, w);
        lds_barrier(); STAMP(13);
        ADAM(C::P_F1, H * H, C::C_F1, C::R_F1, C::OFF_F1, A3, SA);
        lds_barrier(); STAMP(14);
        // layer 1: dz -> A2 (16 columns)
        dgrad_block<1, 1, 0>(A1, SA, H, arena + C::OFF_F0, DL, nullptr, 0, A2, SA, rt, 0, slope);
        wgrad_block<1, 1>(A1, SA, H, sZ, SS, DL, A4, rt, 0);
        colsum64(A1, SA, H, csum, w);
        lds_barrier(); STAMP(15);
        ADAM(C::P_F0, H * DL, C::C_F0, C::R_F0, C::OFF_F0, A1, SA);
        // ---- E: head deltas in place (SURVEY §8 a-bwd): sG <- dg, sV <- dlogv, sL <- dlogl
        for (int e = tid; e < FT_B * SS; e += FT_THREADS) {
            const int b = e / SS, k = e - b * SS;
            float dg = 0.f, dv = 0.f, dl = 0.f;
            if (k < DL) {
                const float zz = sZ[e], gg = sG[e], vv = sV[e], ll = sL[e];
                const float dq = zz - gg;
                const float rr = dq / vv;
                const float Gz = A2[b * SA + k] - wB * (A.b * rr - A.d * zz / ll);
                dg = wB * A.b * rr + Gz;
                dv = -wB * A.b * 0.5f * (1.f - dq * rr) + Gz * 0.5f * dq;
                dl = wB * A.d * 0.5f * (1.f - zz * zz / ll);
            }
            sG[e] = dg; sV[e] = dv; sL[e] = dl;
        }
        // restore g's hidden activations: h2g -> A0 (h1f dead), h1g -> A3 (d2f dead).  A1 is still read by the
        // bias sums of the decoder's first layer above, A2 (dz) by the delta loop.
        restore_block<2, 2>(st2g, A0, SA, r22, c22);
        restore_block<2, 2>(st1g, A3, SA, r22, c22);
        lds_barrier(); STAMP(16);

        // ---- g backward: d2g -> A2, d1g -> A1
        if (s + 1 < A.n_steps) load_rows<C>(rows, idx + FT_B);     // next batch's row indices, used one phase later
        dgrad_block<2, 2, 1>(sG, SS, DL, arena + C::OFF_G2, H, A0, SA, A2, SA, r22, c22, slope);
        wgrad_block<1, 1>(sG, SS, DL, A0, SA, H, A4, 0, rt);
        colsum64(sG, SS, DL, csum, w);
        lds_barrier(); STAMP(17);
        ADAM(C::P_E2, DL * H, C::C_E2, C::R_G2, C::OFF_G2, sG, SS);
        lds_barrier(); STAMP(18);
        dgrad_block<2, 2, 1>(A2, SA, H, arena + C::OFF_G1, H, A3, SA, A1, SA, r22, c22, slope);
        wgrad_block<2, 2>(A2, SA, H, A3, SA, H, A4, r22, c22);
        // XU again (registers -> A0, h2g is dead) for the first-layer weight gradients of g / logv / logl,
        // then start fetching the next step's batch: its latency hides behind the rest of this step
#pragma unroll
        for (int j = 0; j < NXU; ++j) { const int e = tid + j * FT_THREADS; if (e < FT_B * SA) A0[e] = xu[j]; }
        if (s + 1 < A.n_steps) load_xu<C>(xu, rows, xd, ud);
        colsum64(A2, SA, H, csum, w);
        lds_barrier(); STAMP(19);
        ADAM(C::P_E1, H * H, C::C_E1, C::R_G1, C::OFF_G1, A2, SA);
        lds_barrier(); STAMP(20);
        wgrad_block<2, 2>(A1, SA, H, A0, SA, KIN, A4, r22, c22);
        colsum64(A1, SA, H, csum, w);
        lds_barrier(); STAMP(21);
        ADAM(C::P_E0, H * KIN, C::C_E0, C::R_G0, C::OFF_G0, A1, SA);
        // restore logv's hidden activations: h2v -> A2 (d2g dead), h1v -> A3 (h1g dead); A1 busy (bias sums)
        restore_block<2, 2>(st2v, A2, SA, r22, c22);
        restore_block<2, 2>(st1v, A3, SA, r22, c22);
        lds_barrier(); STAMP(22);

        // ---- logv backward: d2v -> A1, d1v -> A2
        dgrad_block<2, 2, 1>(sV, SS, DL, arena + C::OFF_V2, H, A2, SA, A1, SA, r22, c22, slope);
        wgrad_block<1, 1>(sV, SS, DL, A2, SA, H, A4, 0, rt);
        colsum64(sV, SS, DL, csum, w);
        lds_barrier(); STAMP(23);
        ADAM(C::P_E2, DL * H, C::C_E2, C::R_V2, C::OFF_V2, sV, SS);
        lds_barrier(); STAMP(24);
        dgrad_block<2, 2, 1>(A1, SA, H, arena + C::OFF_V1, H, A3, SA, A2, SA, r22, c22, slope);
        wgrad_block<2, 2>(A1, SA, H, A3, SA, H, A4, r22, c22);
        colsum64(A1, SA, H, csum, w);
        lds_barrier(); STAMP(25);
        ADAM(C::P_E1, H * H, C::C_E1, C::R_V1, C::OFF_V1, A1, SA);
        lds_barrier(); STAMP(26);
        wgrad_block<2, 2>(A2, SA, H, A0, SA, KIN, A4, r22, c22);
        colsum64(A2, SA, H, csum, w);
        lds_barrier(); STAMP(27);
        ADAM(C::P_E0, H * KIN, C::C_E0, C::R_V0, C::OFF_V0, A2, SA);

        // ---- logl backward: h2l -> A1 (d2v dead), h1l -> A3 (h1v dead); d2l -> A2 (after the sync), d1l -> A1
        if (C::HAS_U) {
            restore_block<2, 2>(st4, A1, SA, r22, c22);
            restore_block<2, 2>(st3, A3, SA, r22, c22);
            lds_barrier(); STAMP(28);
            dgrad_block<2, 2, 1>(sL, SS, DL, arena + C::OFF_L2, H, A1, SA, A2, SA, r22, c22, slope);
            wgrad_block<1, 1>(sL, SS, DL, A1, SA, H, A4, 0, rt);
            colsum64(sL, SS, DL, csum, w);
            lds_barrier(); STAMP(29);
            ADAM(C::P_L2, DL * H, C::C_L2, C::R_L2, C::OFF_L2, sL, SS);
            lds_barrier(); STAMP(30);
            dgrad_block<2, 2, 1>(A2, SA, H, arena + C::OFF_L1, H, A3, SA, A1, SA, r22, c22, slope);
            wgrad_block<2, 2>(A2, SA, H, A3, SA, H, A4, r22, c22);
            colsum64(A2, SA, H, csum, w);
            lds_barrier(); STAMP(31);
            ADAM(C::P_L1, H * H, C::C_L1, C::R_L1, C::OFF_L1, A2, SA);
            lds_barrier(); STAMP(32);
            wgrad_block<2, 2>(A1, SA, H, A0 + DD, SA, M, A4, r22, c22);
            colsum64(A1, SA, H, csum, w);
            lds_barrier(); STAMP(33);
            ADAM(C::P_L0, H * M, C::C_L0, C::R_L0, C::OFF_L0, A1, SA);
        }
#undef ADAM
        lds_barrier();
    }

    // ---- write state back
    for (int i = tid; i < C::NPARAM; i += FT_THREADS) gP[i] = arena[i];
    if (C::HAS_U) {
        adam_store<C::P_L0, C::C_L0, C::R_L0>(gM + C::OFF_L0, gV + C::OFF_L0, am, av);
        adam_store<C::P_L1, C::C_L1, C::R_L1>(gM + C::OFF_L1, gV + C::OFF_L1, am, av);
        adam_store<C::P_L2, C::C_L2, C::R_L2>(gM + C::OFF_L2, gV + C::OFF_L2, am, av);
    }
    adam_store<C::P_F0, C::C_F0, C::R_F0>(gM + C::OFF_F0, gV + C::OFF_F0, am, av);
    adam_store<C::P_F1, C::C_F1, C::R_F1>(gM + C::OFF_F1, gV + C::OFF_F1, am, av);
    adam_store<C::P_F2, C::C_F2, C::R_F2>(gM + C::OFF_F2, gV + C::OFF_F2, am, av);
    adam_store<C::P_E0, C::C_E0, C::R_G0>(gM + C::OFF_G0, gV + C::OFF_G0, am, av);
    adam_store<C::P_E1, C::C_E1, C::R_G1>(gM + C::OFF_G1, gV + C::OFF_G1, am, av);
    adam_store<C::P_E2, C::C_E2, C::R_G2>(gM + C::OFF_G2, gV + C::OFF_G2, am, av);
    adam_store<C::P_E0, C::C_E0, C::R_V0>(gM + C::OFF_V0, gV + C::OFF_V0, am, av);
    adam_store<C::P_E1, C::C_E1, C::R_V1>(gM + C::OFF_V1, gV + C::OFF_V1, am, av);
    adam_store<C::P_E2, C::C_E2, C::R_V2>(gM + C::OFF_V2, gV + C::OFF_V2, am, av);
}

// ------------------------------------------------------------------------------ host side
template <class C>
static hipError_t launch_cfg(const FusedArgs& A, int n_models, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_train_steps<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_train_steps<C>, dim3(n_models), dim3(FT_THREADS), C::LDS_BYTES, st, A);
    return hipGetLastError();
}

#define FT_SHAPES(X) X(10, 10, 40, 50) X(10, 10, 0, 50) X(5, 5, 40, 50) X(5, 5, 0, 50)

#ifdef FT_PROF
extern "C" int ivae_debug_set_prof(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ft_prof), &buf, sizeof(buf));
}
#endif

bool fused_supported(int dd, int dl, int m, int h, int n_layers, int act, int batch) {
    if (n_layers != 3 || act != IVAE_ACT_LRELU || batch != FT_B) return false;
#define X(a, b, c, d) if (dd == a && dl == b && m == c && h == d) return true;
    FT_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_train_steps(int dd, int dl, int m, int h, const FusedArgs& A, int n_models, hipStream_t st) {
#define X(a, b, c, d) if (dd == a && dl == b && m == c && h == d) return launch_cfg<FCfg<a, b, c, d>>(A, n_models, st);
    FT_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
