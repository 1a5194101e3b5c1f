// Pairwise log q(z_i | x_j) kernels for the total-correlation decomposition.
//
// Reference (models/nets.py:180-182, 238-240, 466-469) materialises T[i,j,k] = log N(z_ik; g_jk, v_jk) as a
// (B,B,d) tensor plus autograd copies.  Here T is never stored.  Outputs:
//   lse_s[i]   = LSE_j sum_k T        lse_k[i,k] = LSE_j T[.,.,k]
//   lq[i]      = lse_s - log(M*N)     lqi[i]     = sum_k (lse_k - log(M*N))
//   rz[i,k]    = sum_j Wt * dT/dz_ik           Wt = (b-c) softmax_j(sum_k T) + (c-d) softmax_j(T_k)
//   cg[j,k]    = sum_i Wt * dT/dg_jk           cv[j,k] = sum_i Wt * dT/dlogv_jk
//
// Layout of the work (d <= 32): a LANE owns a row i (64 rows per wave, 256 per workgroup) and walks over a slice of
// columns j whose per-column constants  g_jk, w_jk = 1/v_jk, a_jk = -(log 2pi + log v_jk)/2  were computed once per
// workgroup into LDS, so that  T = a - (z-g)^2 w / 2  costs three VALU operations with wave-uniform (broadcast)
// operands: no logarithm or division per (i,j,k), no cross-lane reduction at all.  The column range is split into
// slices over blockIdx.y to fill the chip; the per-slice log-sum-exp statistics (max, sum) and gradient partial sums
// go to a caller-provided workspace and are merged in a fixed order (deterministic results).  The column gradients
// cg, cv use the mirrored map (a lane owns a column, walks over a slice of rows).  Four launches:
//   k_pair_stats -> k_pair_combine -> k_pair_grads (both roles) -> k_pair_finish (+ lq, lqi)
// exp() inside the sums is v_exp_f32 (exp2(x log2 e)): arguments are <= 0 and the terms that matter have small
// magnitude, so the sums keep ~1e-6 relative accuracy; the final log / merge steps use the full-precision functions.
// 32 < d <= 64 takes the wave-per-row kernels at the end of this file.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

#define LOG2PI 1.8378770664093453f
#define NEG_BIG -3.0e38f
#define PB 256                     // rows (columns) per workgroup: one per lane

struct PairGeom {                  // host-computed slicing (pair_geometry)
    int n_rows, n_cols, d, dp;     // dp: d rounded up to a multiple of 4 (row stride of the staged constants)
    int CS, NS;                    // columns per slice, slices        (stats, row gradients)
    int RS, NR;                    // rows per slice, slices           (column gradients)
    size_t o_pm, o_ps;             // (NS, d+1, n_rows) per-slice max / sum (log2 units); plane d: the sum-over-k statistic
    size_t o_rz;                   // (NS, d, n_rows)
    size_t o_cg, o_cv;             // (NR, d, n_cols)
    size_t total;                  // floats
    int n_models;                  // packed models: blockIdx.x = model * (blocks per model) + block
    size_t mstride;                // floats between consecutive models in EVERY array (inputs, outputs, scratch)
};

static PairGeom pair_geometry(int n_rows, int n_cols, int d) {
    PairGeom G;
    G.n_rows = n_rows; G.n_cols = n_cols; G.d = d; G.dp = (d + 3) & ~3;
    const int rb = (n_rows + PB - 1) / PB, cb = (n_cols + PB - 1) / PB;
    // slices: enough workgroups for the 256 CUs (x2), bounded by 48 KB of LDS for the staged constants
    int want = (512 + rb - 1) / rb;
    G.CS = (n_cols + want - 1) / want;
    if (G.CS < 32) G.CS = 32;
    if (G.CS > 4096 / G.dp) G.CS = 4096 / G.dp;
    G.NS = (n_cols + G.CS - 1) / G.CS;
    want = (512 + cb - 1) / cb;
    G.RS = (n_rows + want - 1) / want;
    if (G.RS < 32) G.RS = 32;
    if (G.RS > 12288 / (2 * G.dp + 1)) G.RS = 12288 / (2 * G.dp + 1);
    G.NR = (n_rows + G.RS - 1) / G.RS;
    size_t o = 0;
    G.o_pm = o; o += (size_t)G.NS * (d + 1) * n_rows;
    G.o_ps = o; o += (size_t)G.NS * (d + 1) * n_rows;
    G.o_rz = o; o += (size_t)G.NS * d * n_rows;
    G.o_cg = o; o += (size_t)G.NR * d * n_cols;
    G.o_cv = o; o += (size_t)G.NR * d * n_cols;
    G.total = o;
    G.n_models = 1; G.mstride = 0;
    return G;
}

size_t pairwise_ws_floats(int n_rows, int n_cols, int d) {
    if (d > 32) return 0;
    return pair_geometry(n_rows, n_cols, d).total;
}

// The inner loops work in log2 units: T2 = T * log2(e) = A2 - dq^2 * W2 with A2 = a * log2(e), W2 = w * log2(e) / 2,
// so that exp(T - m) is one v_exp_f32 of a difference.  Padded k (d <= k < DP) carry A2 = W2 = 0, z = g = 0: T2 = 0,
// which adds nothing to the sum over k; their own statistics are never read.
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Round 3: the walks are bound by vector-instruction issue (tools/ubench/valu_pk_rate.hip: a packed fp32 instruction costs
// what a plain one costs and does two elements), so the arithmetic runs on PAIRS of adjacent latent dimensions k, k+1 --
// operands of a pair are adjacent registers / adjacent LDS words already -- as v_pk_add / v_pk_mul / v_pk_fma_f32.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fexp2(f2 x) { return f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }

// Column constants of slice [j0, j0+nj) -> LDS, three (nj, DP) arrays.
template <int DP>
__device__ __forceinline__ void stage_cols(const float* __restrict__ g, const float* __restrict__ v, int j0, int nj, int d,
                                           float* G, float* W2, float* A2) {
    for (int e = threadIdx.x; e < nj * DP; e += PB) {
        const int j = e / DP, k = e - j * DP;
        float gg = 0.f, w2 = 0.f, a2 = 0.f;
        if (k < d) {
            const float vv = v[(size_t)(j0 + j) * d + k];
            gg = g[(size_t)(j0 + j) * d + k];
            w2 = (0.5f * LOG2E) / vv;
            a2 = (-0.5f * LOG2E) * (LOG2PI + logf(vv));
        }
        G[e] = gg; W2[e] = w2; A2[e] = a2;
    }
}

// Statistics of one row over staged columns [j0, j1): the exact maxima first (a log-sum-exp reference that is merely an
// upper bound -- the largest A2 of the slice, which would save this pass -- underflows for a row that is far from every
// column of the slice relative to the columns' variances: every trained encoder; built and measured in round 3, then
// dropped), then the sums of exp2(T2 - max); packed over k pairs.  T2 = A2 - dq * (dq * W2): the form the gradient walks use.
template <int DP>
__device__ __forceinline__ void row_stats(const float* zi, const float* G, const float* W2, const float* A2, int j0, int j1,
                                          float* m, float* s, float& mS, float& sS) {
    static_assert(DP % 2 == 0, "pairs of latent dimensions");
    f2 z2[DP / 2], m2[DP / 2], s2[DP / 2];
#pragma unroll
    for (int h = 0; h < DP / 2; ++h) { z2[h] = f2{zi[2 * h], zi[2 * h + 1]}; m2[h] = splat(NEG_BIG); s2[h] = splat(0.f); }
    float mx = NEG_BIG;
    for (int j = j0; j < j1; ++j) {
        const f2* Gj = reinterpret_cast<const f2*>(G + j * DP);
        const f2* Wj = reinterpret_cast<const f2*>(W2 + j * DP);
        const f2* Aj = reinterpret_cast<const f2*>(A2 + j * DP);
        f2 S2 = splat(0.f);
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) {
            const f2 dq = z2[h] - Gj[h];
            const f2 t = Aj[h] - dq * (dq * Wj[h]);
            m2[h] = f2{fmaxf(m2[h].x, t.x), fmaxf(m2[h].y, t.y)};
            S2 += t;
        }
        mx = fmaxf(mx, S2.x + S2.y);
    }
    float acc = 0.f;
    for (int j = j0; j < j1; ++j) {
        const f2* Gj = reinterpret_cast<const f2*>(G + j * DP);
        const f2* Wj = reinterpret_cast<const f2*>(W2 + j * DP);
        const f2* Aj = reinterpret_cast<const f2*>(A2 + j * DP);
        f2 S2 = splat(0.f);
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) {
            const f2 dq = z2[h] - Gj[h];
            const f2 t = Aj[h] - dq * (dq * Wj[h]);
            s2[h] += fexp2(t - m2[h]);
            S2 += t;
        }
        acc += fexp2(S2.x + S2.y - mx);
    }
#pragma unroll
    for (int h = 0; h < DP / 2; ++h) { m[2 * h] = m2[h].x; m[2 * h + 1] = m2[h].y; s[2 * h] = s2[h].x; s[2 * h + 1] = s2[h].y; }
    mS = mx; sS = acc;
}

// Per-slice statistics: max and sum-of-exp2 over the slice's columns, for every k and for the sum over k.
template <int DP>
__global__ __launch_bounds__(PB) void k_pair_stats(PairGeom P, const float* __restrict__ z, const float* __restrict__ g,
                                                   const float* __restrict__ v, float* __restrict__ ws, int gx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int bx = blockIdx.x % gx;
    const size_t mo = (size_t)(blockIdx.x / gx) * P.mstride;
    z += mo; g += mo; v += mo; ws += mo;
    const int d = P.d, sl = blockIdx.y, j0 = sl * P.CS, nj = min(P.CS, P.n_cols - j0);
    float* G = sm; float* W2 = G + P.CS * DP; float* A2 = W2 + P.CS * DP;
    stage_cols<DP>(g, v, j0, nj, d, G, W2, A2);
    __syncthreads();
    const int i = bx * PB + threadIdx.x;
    const int ic = min(i, P.n_rows - 1);
    float zi[DP], m[DP], s[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) zi[k] = k < d ? z[(size_t)ic * d + k] : 0.f;
    float mS, sS;
    row_stats<DP>(zi, G, W2, A2, 0, nj, m, s, mS, sS);
    if (i < P.n_rows) {
        float* pm = ws + P.o_pm + (size_t)sl * (d + 1) * P.n_rows + i;
        float* ps = ws + P.o_ps + (size_t)sl * (d + 1) * P.n_rows + i;
#pragma unroll
        for (int k = 0; k < DP; ++k)
            if (k < d) { pm[(size_t)k * P.n_rows] = m[k]; ps[(size_t)k * P.n_rows] = s[k]; }
        pm[(size_t)d * P.n_rows] = mS;
        ps[(size_t)d * P.n_rows] = sS;
    }
}

// Merge the slices (log2 units in, natural-log LSE out): thread = (plane k <= d, row i).
__global__ __launch_bounds__(PB) void k_pair_combine(PairGeom P, const float* __restrict__ ws,
                                                     float* __restrict__ lse_s, float* __restrict__ lse_k, int gx) {
    const size_t mo = (size_t)(blockIdx.x / gx) * P.mstride;
    ws += mo; lse_s += mo; lse_k += mo;
    const int i = (blockIdx.x % gx) * PB + threadIdx.x, k = blockIdx.y;
    if (i >= P.n_rows) return;
    const size_t plane = (size_t)(P.d + 1) * P.n_rows, at = (size_t)k * P.n_rows + i;
    const float* pm = ws + P.o_pm + at;
    const float* ps = ws + P.o_ps + at;
    // 16 slices at a time, their loads in flight together (one memory round trip per group instead of one per unrolled
    // batch of 8); same arithmetic and order as a plain loop over the slices: global max first, then the sum
    float M = NEG_BIG;
    for (int s0 = 0; s0 < P.NS; s0 += 16) {
        float m16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m16[e] = pm[min(s0 + e, P.NS - 1) * plane];
#pragma unroll
        for (int e = 0; e < 16; ++e) M = fmaxf(M, m16[e]);          // (a clamped repeat of the last slice changes nothing)
    }
    float s = 0.f;
    for (int s0 = 0; s0 < P.NS; s0 += 16) {
        float m16[16], p16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int sl = min(s0 + e, P.NS - 1);
            m16[e] = pm[sl * plane];
            p16[e] = ps[sl * plane];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (s0 + e < P.NS) s += p16[e] * exp2f(m16[e] - M);
    }
    const float lse = M * LN2 + logf(s);
    if (k == P.d) lse_s[i] = lse;
    else lse_k[(size_t)i * P.d + k] = lse;
}

// Gradient partial sums.  blockIdx.z == 0: row role (lane = row i, slice of columns) -> rz partials;
// blockIdx.z == 1: column role (lane = column j, slice of rows) -> cg, cv partials.
template <int DP>
__global__ __launch_bounds__(PB) void k_pair_grads(PairGeom P, IvaeCoef C, const float* __restrict__ z,
                                                   const float* __restrict__ g, const float* __restrict__ v,
                                                   const float* __restrict__ lse_s, const float* __restrict__ lse_k,
                                                   float* __restrict__ ws, int gx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int bx = blockIdx.x % gx;
    const size_t mo = (size_t)(blockIdx.x / gx) * P.mstride;
    z += mo; g += mo; v += mo; lse_s += mo; lse_k += mo; ws += mo;
    const int d = P.d;
    const float cP = C.b - C.c, cQ = C.c - C.d;
    if (blockIdx.z == 0) {
        if (bx * PB >= P.n_rows || (int)blockIdx.y >= P.NS) return;
        const int sl = blockIdx.y, j0 = sl * P.CS, nj = min(P.CS, P.n_cols - j0);
        float* G = sm; float* W2 = G + P.CS * DP; float* A2 = W2 + P.CS * DP;
        stage_cols<DP>(g, v, j0, nj, d, G, W2, A2);
        __syncthreads();
        const int i = bx * PB + threadIdx.x;
        const int ic = min(i, P.n_rows - 1);
        f2 zi[DP / 2], lk[DP / 2], acc[DP / 2];
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) {
            const int k = 2 * h;
            zi[h] = f2{k < d ? z[(size_t)ic * d + k] : 0.f, k + 1 < d ? z[(size_t)ic * d + k + 1] : 0.f};
            lk[h] = f2{k < d ? lse_k[(size_t)ic * d + k] * LOG2E : 0.f, k + 1 < d ? lse_k[(size_t)ic * d + k + 1] * LOG2E : 0.f};
            acc[h] = splat(0.f);
        }
        const float LS = lse_s[ic] * LOG2E;
        const f2 cQ2 = splat(cQ);
        for (int j = 0; j < nj; ++j) {
            const f2* Gj = reinterpret_cast<const f2*>(G + j * DP);
            const f2* Wj = reinterpret_cast<const f2*>(W2 + j * DP);
            const f2* Aj = reinterpret_cast<const f2*>(A2 + j * DP);
            f2 tt[DP / 2], rr[DP / 2];
            f2 S2 = splat(0.f);
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 dq = zi[h] - Gj[h];
                rr[h] = dq * Wj[h];                                // (z - g) / v * log2(e) / 2
                tt[h] = Aj[h] - dq * rr[h];
                S2 += tt[h];
            }
            const f2 Pij = splat(cP * fexp2(S2.x + S2.y - LS));
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) acc[h] -= (Pij + cQ2 * fexp2(tt[h] - lk[h])) * rr[h];
        }
        if (i < P.n_rows) {
            float* pr = ws + P.o_rz + (size_t)sl * d * P.n_rows + i;
#pragma unroll
            for (int k = 0; k < DP; ++k)
                if (k < d) pr[(size_t)k * P.n_rows] = (k & 1 ? acc[k / 2].y : acc[k / 2].x) * (2.f * LN2);
        }
    } else {
        if (bx * PB >= P.n_cols || (int)blockIdx.y >= P.NR) return;
        const int rs = blockIdx.y, i0 = rs * P.RS, ni = min(P.RS, P.n_rows - i0);
        float* Z = sm; float* LK = Z + P.RS * DP; float* LSs = LK + P.RS * DP;
        for (int e = threadIdx.x; e < ni * DP; e += PB) {
            const int i = e / DP, k = e - i * DP;
            Z[e] = k < d ? z[(size_t)(i0 + i) * d + k] : 0.f;
            LK[e] = k < d ? lse_k[(size_t)(i0 + i) * d + k] * LOG2E : 0.f;
        }
        for (int e = threadIdx.x; e < ni; e += PB) LSs[e] = lse_s[i0 + e] * LOG2E;
        __syncthreads();
        const int j = bx * PB + threadIdx.x;
        const int jc = min(j, P.n_cols - 1);
        f2 gj[DP / 2], wj[DP / 2], aj[DP / 2], ag[DP / 2], av[DP / 2];
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const float vv = k < d ? v[(size_t)jc * d + k] : 1.f;
            const float g_ = k < d ? g[(size_t)jc * d + k] : 0.f;
            const float w_ = k < d ? (0.5f * LOG2E) / vv : 0.f;
            const float a_ = k < d ? (-0.5f * LOG2E) * (LOG2PI + logf(vv)) : 0.f;
            if (k & 1) { gj[k / 2].y = g_; wj[k / 2].y = w_; aj[k / 2].y = a_; }
            else { gj[k / 2].x = g_; wj[k / 2].x = w_; aj[k / 2].x = a_; }
        }
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) { ag[h] = splat(0.f); av[h] = splat(0.f); }
        const f2 cQ2 = splat(cQ), half2 = splat(0.5f), nln2 = splat(-LN2);
        for (int i = 0; i < ni; ++i) {
            const f2* Zi = reinterpret_cast<const f2*>(Z + i * DP);
            const f2* Li = reinterpret_cast<const f2*>(LK + i * DP);
            f2 tt[DP / 2], rr[DP / 2], uu[DP / 2];
            f2 S2 = splat(0.f);
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 dq = Zi[h] - gj[h];
                rr[h] = dq * wj[h];
                uu[h] = dq * rr[h];                                // dq^2 / v * log2(e) / 2
                tt[h] = aj[h] - uu[h];
                S2 += tt[h];
            }
            const f2 Pij = splat(cP * fexp2(S2.x + S2.y - LSs[i]));
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 wt = Pij + cQ2 * fexp2(tt[h] - Li[h]);
                ag[h] += wt * rr[h];
                av[h] -= wt * (half2 + nln2 * uu[h]);              // -1/2 (1 - dq^2 / v),  dq^2 / v = 2 ln2 * uu
            }
        }
        if (j < P.n_cols) {
            float* pg = ws + P.o_cg + (size_t)rs * d * P.n_cols + j;
            float* pv = ws + P.o_cv + (size_t)rs * d * P.n_cols + j;
#pragma unroll
            for (int k = 0; k < DP; ++k)
                if (k < d) {
                    pg[(size_t)k * P.n_cols] = (k & 1 ? ag[k / 2].y : ag[k / 2].x) * (2.f * LN2);
                    pv[(size_t)k * P.n_cols] = k & 1 ? av[k / 2].y : av[k / 2].x;
                }
        }
    }
}

// Fixed-order sums of the gradient partials: thread = (k, row or column); blockIdx.z: 0 rz (+ lq, lqi), 1 cg, 2 cv.
__global__ __launch_bounds__(PB) void k_pair_finish(PairGeom P, float logMN, const float* __restrict__ ws,
                                                    const float* __restrict__ lse_s, const float* __restrict__ lse_k,
                                                    float* __restrict__ lq, float* __restrict__ lqi, float* __restrict__ rz,
                                                    float* __restrict__ cg, float* __restrict__ cv, int gx) {
    const size_t mo = (size_t)(blockIdx.x / gx) * P.mstride;
    ws += mo; lse_s += mo; lse_k += mo; lq += mo; lqi += mo; rz += mo; cg += mo; cv += mo;
    const int which = blockIdx.z, k = blockIdx.y;
    const int n = which == 0 ? P.n_rows : P.n_cols, ns = which == 0 ? P.NS : P.NR;
    const int i = (blockIdx.x % gx) * PB + threadIdx.x;
    if (i >= n) return;
    if (which == 0 && k == 0) {          // lq, lqi of row i from the merged statistics
        float acc = 0.f;
        for (int kk = 0; kk < P.d; ++kk) acc += lse_k[(size_t)i * P.d + kk] - logMN;
        lq[i] = lse_s[i] - logMN;
        lqi[i] = acc;
    }
    const float* src = ws + (which == 0 ? P.o_rz : which == 1 ? P.o_cg : P.o_cv) + (size_t)k * n + i;
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < ns; ++s) acc += src[(size_t)s * P.d * n];
    float* dst = which == 0 ? rz : which == 1 ? cg : cv;
    dst[(size_t)i * P.d + k] = acc;
}

template <int DP>
static hipError_t launch_sliced(const PairGeom& P, const IvaeCoef& C, float logMN, const float* z, const float* g,
                                const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz, float* cg,
                                float* cv, float* ws, hipStream_t st) {
    const int d = P.d;
    const int rb = (P.n_rows + PB - 1) / PB, cb = (P.n_cols + PB - 1) / PB;
    const size_t sm_cols = (size_t)3 * P.CS * DP * sizeof(float);
    const size_t sm_rows = (size_t)(2 * P.RS * DP + P.RS) * sizeof(float);
    const int nm = P.n_models;
    hipLaunchKernelGGL(k_pair_stats<DP>, dim3(rb * nm, P.NS), dim3(PB), sm_cols, st, P, z, g, v, ws, rb);
    hipLaunchKernelGGL(k_pair_combine, dim3(rb * nm, d + 1), dim3(PB), 0, st, P, ws, lse_s, lse_k, rb);
    const int gx = rb > cb ? rb : cb, gy = P.NS > P.NR ? P.NS : P.NR;
    hipLaunchKernelGGL(k_pair_grads<DP>, dim3(gx * nm, gy, 2), dim3(PB), sm_cols > sm_rows ? sm_cols : sm_rows, st, P, C, z, g, v,
                       lse_s, lse_k, ws, gx);
    hipLaunchKernelGGL(k_pair_finish, dim3(gx * nm, d, 3), dim3(PB), 0, st, P, logMN, ws, lse_s, lse_k, lq, lqi, rz, cg, cv, gx);
    return hipGetLastError();
}

// ------------------------------------------------------------------ small problems: one workgroup, one launch
// n_rows, n_cols <= 64 and d <= 12 (every yaml config of the reference: batch 64, d = 5 or 10): the four stages above run
// inside ONE workgroup per model, so a training step spends one launch on the pairwise terms instead of four.
// Thread = (lane l, wave p): in the row role lane l owns row l and wave p walks PSQ = 64 / NWV columns; in the column
// role lane l owns column l and wave p walks PSQ rows.  The NWV waves' partial statistics / sums meet in LDS and are
// merged in a fixed order (deterministic), with the same arithmetic as the sliced kernels.  The walks are serial
// instruction streams: 16 waves (four per SIMD, 4 columns each) run them about twice as fast as 4 waves of 16 columns,
// one per SIMD, did (a lone wave issues an instruction every ~5 cycles); d > 8 takes 8 waves (LDS for the partials).
#define PSM 64                     // rows / columns of a small problem
template <int DP, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_pair_small(int n_rows, int n_cols, int d, IvaeCoef C, float logMN,
                                                   const float* __restrict__ z, const float* __restrict__ g,
                                                   const float* __restrict__ v, float* __restrict__ lq,
                                                   float* __restrict__ lqi, float* __restrict__ lse_s,
                                                   float* __restrict__ lse_k, float* __restrict__ rz,
                                                   float* __restrict__ cg, float* __restrict__ cv, size_t mstride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const size_t mo = (size_t)blockIdx.x * mstride;
    z += mo; g += mo; v += mo; lq += mo; lqi += mo; lse_s += mo; lse_k += mo; rz += mo; cg += mo; cv += mo;
    float* G = sm;                          // (PSM, DP) column constants
    float* W2 = G + PSM * DP;
    float* A2 = W2 + PSM * DP;
    float* Z = A2 + PSM * DP;               // (PSM, DP) rows
    float* LK = Z + PSM * DP;               // (PSM, DP) lse_k (natural log)
    float* LS = LK + PSM * DP;              // (PSM) lse_s
    float* part = LS + PSM;                 // stage 1-2: pm, ps (4, DP + 1, PSM) each; stage 3-4: rz, cg, cv partials (4, DP, PSM) each
    constexpr int PSQ = PSM / NWV, NTH = 64 * NWV;
    float* pm = part;
    float* ps = pm + NWV * (DP + 1) * PSM;
    float* pr = part;
    float* pg = pr + NWV * DP * PSM;
    float* pv = pg + NWV * DP * PSM;
    const int tid = threadIdx.x, l = tid & 63, p = tid >> 6;
    const float cP = C.b - C.c, cQ = C.c - C.d;

    for (int e = tid; e < PSM * DP; e += NTH) {            // column constants (stage_cols' arithmetic) and rows -> LDS
        const int j = e / DP, k = e - j * DP;
        float gg = 0.f, w2 = 0.f, a2 = 0.f, zz = 0.f;
        if (k < d && j < n_cols) {
            const float vv = v[(size_t)j * d + k];
            gg = g[(size_t)j * d + k];
            w2 = (0.5f * LOG2E) / vv;
            a2 = (-0.5f * LOG2E) * (LOG2PI + logf(vv));
        }
        if (k < d && j < n_rows) zz = z[(size_t)j * d + k];
        G[e] = gg; W2[e] = w2; A2[e] = a2; Z[e] = zz; LK[e] = 0.f;       // (LK: the padded k stay 0)
    }
    __syncthreads();

    // ---- stage 1: per-wave statistics of row l over the wave's columns
    const int ic = min(l, n_rows - 1);
    const int j0 = PSQ * p, j1 = min(j0 + PSQ, n_cols);
    float zi[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) zi[k] = Z[ic * DP + k];
    {
        float m[DP], s[DP];
        float mS, sS;
        row_stats<DP>(zi, G, W2, A2, j0, j1, m, s, mS, sS);
#pragma unroll
        for (int k = 0; k < DP; ++k) { pm[(p * (DP + 1) + k) * PSM + l] = m[k]; ps[(p * (DP + 1) + k) * PSM + l] = s[k]; }
        pm[(p * (DP + 1) + DP) * PSM + l] = mS;
        ps[(p * (DP + 1) + DP) * PSM + l] = sS;
    }
    __syncthreads();

    // ---- stage 2: merge the four waves (log2 units in, natural-log LSE out); plane DP is the sum-over-k statistic
    for (int e = tid; e < (d + 1) * PSM; e += NTH) {
        const int kk = e / PSM, i = e - kk * PSM;
        const int k = kk == d ? DP : kk;
        float M = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NWV; ++w) M = fmaxf(M, pm[(w * (DP + 1) + k) * PSM + i]);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) sum += ps[(w * (DP + 1) + k) * PSM + i] * exp2f(pm[(w * (DP + 1) + k) * PSM + i] - M);
        const float lse = M * LN2 + logf(sum);
        if (kk == d) { LS[i] = lse; if (i < n_rows) lse_s[i] = lse; }
        else { LK[i * DP + kk] = lse; if (i < n_rows) lse_k[(size_t)i * d + kk] = lse; }
    }
    __syncthreads();
    if (tid < n_rows) {
        float acc = 0.f;
        for (int kk = 0; kk < d; ++kk) acc += LK[tid * DP + kk] - logMN;
        lq[tid] = LS[tid] - logMN;
        lqi[tid] = acc;
    }

    // ---- stage 3a: row role -> rz partials of the wave's columns
    {
        f2 z2[DP / 2], lk[DP / 2], acc[DP / 2];
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) {
            z2[h] = f2{zi[2 * h], zi[2 * h + 1]};
            lk[h] = f2{2 * h < d ? LK[ic * DP + 2 * h] * LOG2E : 0.f, 2 * h + 1 < d ? LK[ic * DP + 2 * h + 1] * LOG2E : 0.f};
            acc[h] = splat(0.f);
        }
        const float LSi = LS[ic] * LOG2E;
        const f2 cQ2 = splat(cQ);
        for (int j = j0; j < j1; ++j) {
            const f2* Gj = reinterpret_cast<const f2*>(G + j * DP);
            const f2* Wj = reinterpret_cast<const f2*>(W2 + j * DP);
            const f2* Aj = reinterpret_cast<const f2*>(A2 + j * DP);
            f2 tt[DP / 2], rr[DP / 2];
            f2 S2 = splat(0.f);
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 dq = z2[h] - Gj[h];
                rr[h] = dq * Wj[h];
                tt[h] = Aj[h] - dq * rr[h];
                S2 += tt[h];
            }
            const f2 Pij = splat(cP * fexp2(S2.x + S2.y - LSi));
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) acc[h] -= (Pij + cQ2 * fexp2(tt[h] - lk[h])) * rr[h];
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) pr[(p * DP + k) * PSM + l] = (k & 1 ? acc[k / 2].y : acc[k / 2].x) * (2.f * LN2);
    }
    // ---- stage 3b: column role (lane = column l) -> cg, cv partials of the wave's rows
    {
        const int jc = min(l, n_cols - 1);
        const int i0 = PSQ * p, i1 = min(i0 + PSQ, n_rows);
        f2 gj[DP / 2], wj[DP / 2], aj[DP / 2], ag[DP / 2], av[DP / 2];
#pragma unroll
        for (int h = 0; h < DP / 2; ++h) {
            gj[h] = *reinterpret_cast<const f2*>(G + jc * DP + 2 * h);
            wj[h] = *reinterpret_cast<const f2*>(W2 + jc * DP + 2 * h);
            aj[h] = *reinterpret_cast<const f2*>(A2 + jc * DP + 2 * h);
            ag[h] = splat(0.f); av[h] = splat(0.f);
        }
        const f2 cQ2 = splat(cQ), half2 = splat(0.5f), nln2 = splat(-LN2), l2e = splat(LOG2E);
        for (int i = i0; i < i1; ++i) {
            const f2* Zi = reinterpret_cast<const f2*>(Z + i * DP);
            const f2* Li = reinterpret_cast<const f2*>(LK + i * DP);
            f2 tt[DP / 2], rr[DP / 2], uu[DP / 2];
            f2 S2 = splat(0.f);
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 dq = Zi[h] - gj[h];
                rr[h] = dq * wj[h];
                uu[h] = dq * rr[h];
                tt[h] = aj[h] - uu[h];
                S2 += tt[h];
            }
            const f2 Pij = splat(cP * fexp2(S2.x + S2.y - LS[i] * LOG2E));
#pragma unroll
            for (int h = 0; h < DP / 2; ++h) {
                const f2 wt = Pij + cQ2 * fexp2(tt[h] - Li[h] * l2e);
                ag[h] += wt * rr[h];
                av[h] -= wt * (half2 + nln2 * uu[h]);
            }
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            pg[(p * DP + k) * PSM + l] = (k & 1 ? ag[k / 2].y : ag[k / 2].x) * (2.f * LN2);
            pv[(p * DP + k) * PSM + l] = k & 1 ? av[k / 2].y : av[k / 2].x;
        }
    }
    __syncthreads();

    // ---- stage 4: fixed-order sums of the four waves' partials
    for (int e = tid; e < d * PSM; e += NTH) {
        const int k = e / PSM, i = e - k * PSM;
        float ar = 0.f, ag = 0.f, av = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            ar += pr[(w * DP + k) * PSM + i];
            ag += pg[(w * DP + k) * PSM + i];
            av += pv[(w * DP + k) * PSM + i];
        }
        if (i < n_rows) rz[(size_t)i * d + k] = ar;
        if (i < n_cols) { cg[(size_t)i * d + k] = ag; cv[(size_t)i * d + k] = av; }
    }
}

template <int DP, int NWV>
static hipError_t launch_small_w(int n_rows, int n_cols, int d, const IvaeCoef& C, float logMN, const float* z, const float* g,
                                 const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz, float* cg,
                                 float* cv, int n_models, size_t mstride, hipStream_t st) {
    const int np = 2 * NWV * (DP + 1) > 3 * NWV * DP ? 2 * NWV * (DP + 1) : 3 * NWV * DP;
    const size_t bytes = (size_t)(5 * PSM * DP + PSM + np * PSM) * sizeof(float);      // 109 KB at DP = 8 with 16 waves, 90 KB at DP = 12 with 8
    static SmemCache cur = {};
    hipError_t e = set_smem((const void*)k_pair_small<DP, NWV>, (int)bytes, cur);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pair_small<DP, NWV>), dim3(n_models), dim3(64 * NWV), bytes, st, n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi,
                       lse_s, lse_k, rz, cg, cv, mstride);
    return hipGetLastError();
}
// a few models: as many waves as the LDS allows (the walks are latency-bound); a chip-filling pack: four waves per model,
// several models per CU
template <int DP>
static hipError_t launch_small(int n_rows, int n_cols, int d, const IvaeCoef& C, float logMN, const float* z, const float* g,
                               const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz, float* cg,
                               float* cv, int n_models, size_t mstride, hipStream_t st) {
    if (n_models >= 128)
        return launch_small_w<DP, 4>(n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, n_models, mstride, st);
    return launch_small_w<DP, (DP <= 8 ? 16 : 8)>(n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, n_models,
                                                  mstride, st);
}

// ------------------------------------------------------------------ 32 < d <= 64: one wavefront per row / column
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

__device__ __forceinline__ void lse_push(float& m, float& s, float t) {
    if (t > m) { s = s * expf(m - t) + 1.f; m = t; }
    else s += expf(t - m);
}

__global__ __launch_bounds__(256) void k_pair_rows_wide(int n_rows, int n_cols, int d, IvaeCoef C, float logMN,
                                                        const float* __restrict__ z, const float* __restrict__ g,
                                                        const float* __restrict__ v, float* __restrict__ lq,
                                                        float* __restrict__ lqi, float* __restrict__ lse_s,
                                                        float* __restrict__ lse_k, float* __restrict__ rz) {
    constexpr int DMAX = 64;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;            // wave-uniform exit
    float zi[DMAX], mk[DMAX], sk[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        zi[k] = k < d ? z[(size_t)i * d + k] : 0.f;
        mk[k] = NEG_BIG; sk[k] = 0.f;
    }
    float mS = NEG_BIG, sS = 0.f;
    for (int j = lane; j < n_cols; j += 64) {
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                const float gg = g[(size_t)j * d + k], vv = v[(size_t)j * d + k];
                const float dq = zi[k] - gg;
                // (the form the gradient walks below and k_pair_cols_wide use: a weight exp(T - lse) is then formed from the
                // very T that went into the lse -- with dq * dq / vv here a single far column weighed 1.017 instead of 1)
                const float t = -0.5f * (LOG2PI + logf(vv) + dq * (dq / vv));
                S += t;
                lse_push(mk[k], sk[k], t);
            }
        }
        lse_push(mS, sS, S);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lse_merge(mS, sS, __shfl_xor(mS, off), __shfl_xor(sS, off));
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if (k < d) lse_merge(mk[k], sk[k], __shfl_xor(mk[k], off), __shfl_xor(sk[k], off));
    }
    const float LS = mS + logf(sS);
    float lqi_acc = 0.f;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        if (k < d) { mk[k] = mk[k] + logf(sk[k]); lqi_acc += mk[k] - logMN; }   // mk now holds lse_k
    }
    if (lane == 0) { lse_s[i] = LS; lq[i] = LS - logMN; lqi[i] = lqi_acc; }
    const float cP = C.b - C.c, cQ = C.c - C.d;
    float acc[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) acc[k] = 0.f;
    for (int j = lane; j < n_cols; j += 64) {
        float S = 0.f;
        float t[DMAX], rr[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                const float gg = g[(size_t)j * d + k], vv = v[(size_t)j * d + k];
                const float dq = zi[k] - gg;
                rr[k] = dq / vv;
                t[k] = -0.5f * (LOG2PI + logf(vv) + dq * rr[k]);
                S += t[k];
            }
        }
        const float Pij = cP * expf(S - LS);
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if (k < d) acc[k] -= (Pij + cQ * expf(t[k] - mk[k])) * rr[k];
    }
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        if (k < d) {
            float a = acc[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0) { rz[(size_t)i * d + k] = a; lse_k[(size_t)i * d + k] = mk[k]; }
        }
    }
}

__global__ __launch_bounds__(256) void k_pair_cols_wide(int n_rows, int n_cols, int d, IvaeCoef C,
                                                        const float* __restrict__ z, const float* __restrict__ g,
                                                        const float* __restrict__ v, const float* __restrict__ lse_s,
                                                        const float* __restrict__ lse_k, float* __restrict__ cg,
                                                        float* __restrict__ cv) {
    constexpr int DMAX = 64;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_cols) return;
    float gj[DMAX], ivj[DMAX], lvj[DMAX], ag[DMAX], av[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        const float vv = k < d ? v[(size_t)j * d + k] : 1.f;
        gj[k] = k < d ? g[(size_t)j * d + k] : 0.f;
        ivj[k] = vv; lvj[k] = logf(vv);
        ag[k] = 0.f; av[k] = 0.f;
    }
    const float cP = C.b - C.c, cQ = C.c - C.d;
    for (int i = lane; i < n_rows; i += 64) {
        float S = 0.f;
        float t[DMAX], rr[DMAX], dq[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                dq[k] = z[(size_t)i * d + k] - gj[k];
                rr[k] = dq[k] / ivj[k];
                t[k] = -0.5f * (LOG2PI + lvj[k] + dq[k] * rr[k]);
                S += t[k];
            }
        }
        const float Pij = cP * expf(S - lse_s[i]);
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                const float wt = Pij + cQ * expf(t[k] - lse_k[(size_t)i * d + k]);
                ag[k] += wt * rr[k];
                av[k] += wt * -0.5f * (1.f - dq[k] * rr[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        if (k < d) {
            float a = ag[k], b = av[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
            if (lane == 0) { cg[(size_t)j * d + k] = a; cv[(size_t)j * d + k] = b; }
        }
    }
}

// n_models packed problems of one shape: every pointer is the first model's, model m lives `mstride` floats further.
hipError_t launch_pairwise(int n_rows, int n_cols, int d, const IvaeCoef& C, const float* z, const float* g,
                           const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz,
                           float* cg, float* cv, float* ws, int n_models, size_t mstride, hipStream_t st) {
    const float logMN = (float)log((double)C.global_batch * (double)C.dataset_size);   // np.log(M * N), nets.py:181
    if (d > 32) {
        for (int m = 0; m < n_models; ++m) {
            const size_t o = (size_t)m * mstride;
            hipLaunchKernelGGL(k_pair_rows_wide, dim3((n_rows + 3) / 4), dim3(256), 0, st, n_rows, n_cols, d, C, logMN, z + o,
                               g + o, v + o, lq + o, lqi + o, lse_s + o, lse_k + o, rz + o);
            hipLaunchKernelGGL(k_pair_cols_wide, dim3((n_cols + 3) / 4), dim3(256), 0, st, n_rows, n_cols, d, C, z + o, g + o,
                               v + o, lse_s + o, lse_k + o, cg + o, cv + o);
        }
        return hipGetLastError();
    }
    if (n_rows <= PSM && n_cols <= PSM && d <= 12) {
        if (d <= 4) return launch_small<4>(n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, n_models, mstride, st);
        if (d <= 8) return launch_small<8>(n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, n_models, mstride, st);
        return launch_small<12>(n_rows, n_cols, d, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, n_models, mstride, st);
    }
    PairGeom P = pair_geometry(n_rows, n_cols, d);
    P.n_models = n_models; P.mstride = mstride;
#define PAIR_CASE(DP_) case DP_: return launch_sliced<DP_>(P, C, logMN, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, ws, st);
    switch (P.dp) {
        PAIR_CASE(4) PAIR_CASE(8) PAIR_CASE(12) PAIR_CASE(16) PAIR_CASE(20) PAIR_CASE(24) PAIR_CASE(28) PAIR_CASE(32)
    }
#undef PAIR_CASE
    return hipErrorInvalidValue;
}
