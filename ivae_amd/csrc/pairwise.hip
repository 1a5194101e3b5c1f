// Pairwise log q(z_i | x_j) kernels for the total-correlation decomposition.
//
// Reference (models/nets.py:180-182, 238-240, 466-469) materialises T[i,j,k] =
// log N(z_ik; g_jk, v_jk) as a (B,B,d) tensor plus autograd copies.  Here T is never stored:
// one 64-lane wavefront owns a row i (k_pair_rows) or a column j (k_pair_cols), its lanes
// stride over the other index, and the log-sum-exp statistics are combined with wavefront
// shuffles (online max/sum, flash-attention style).  Outputs:
//   lse_s[i]   = LSE_j sum_k T        lse_k[i,k] = LSE_j T[.,.,k]
//   lq[i]      = lse_s - log(M*N)     lqi[i]     = sum_k (lse_k - log(M*N))
//   rz[i,k]    = sum_j Wt * dT/dz_ik           Wt = (b-c) softmax_j(sum_k T) + (c-d) softmax_j(T_k)
//   cg[j,k]    = sum_i Wt * dT/dg_jk           cv[j,k] = sum_i Wt * dT/dlogv_jk
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

#define LOG2PI 1.8378770664093453f
#define NEG_BIG -3.0e38f

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

__device__ __forceinline__ void lse_push(float& m, float& s, float t) {
    if (t > m) { s = s * expf(m - t) + 1.f; m = t; }
    else s += expf(t - m);
}

template <int DMAX>
__global__ __launch_bounds__(256) void k_pair_rows(int n_rows, int n_cols, int d, IvaeCoef C, float logMN,
                                                   const float* __restrict__ z, const float* __restrict__ g,
                                                   const float* __restrict__ v, float* __restrict__ lq,
                                                   float* __restrict__ lqi, float* __restrict__ lse_s,
                                                   float* __restrict__ lse_k, float* __restrict__ rz) {
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
                const float t = -0.5f * (LOG2PI + logf(vv) + dq * dq / vv);
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
    // second sweep: row gradient term
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

template <int DMAX>
__global__ __launch_bounds__(256) void k_pair_cols(int n_rows, int n_cols, int d, IvaeCoef C,
                                                   const float* __restrict__ z, const float* __restrict__ g,
                                                   const float* __restrict__ v, const float* __restrict__ lse_s,
                                                   const float* __restrict__ lse_k, float* __restrict__ cg,
                                                   float* __restrict__ cv) {
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

template <int DMAX>
static hipError_t launch_t(int n_rows, int n_cols, int d, const IvaeCoef& C, const float* z, const float* g,
                           const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz,
                           float* cg, float* cv, hipStream_t st) {
    const float logMN = (float)log((double)C.global_batch * (double)C.dataset_size);   // np.log(M * N), nets.py:181
    hipLaunchKernelGGL(k_pair_rows<DMAX>, dim3((n_rows + 3) / 4), dim3(256), 0, st, n_rows, n_cols, d, C, logMN, z, g, v,
                       lq, lqi, lse_s, lse_k, rz);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pair_cols<DMAX>, dim3((n_cols + 3) / 4), dim3(256), 0, st, n_rows, n_cols, d, C, z, g, v,
                       lse_s, lse_k, cg, cv);
    return hipGetLastError();
}

hipError_t launch_pairwise(int n_rows, int n_cols, int d, const IvaeCoef& C, const float* z, const float* g,
                           const float* v, float* lq, float* lqi, float* lse_s, float* lse_k, float* rz,
                           float* cg, float* cv, hipStream_t st) {
    if (d <= 8) return launch_t<8>(n_rows, n_cols, d, C, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, st);
    if (d <= 16) return launch_t<16>(n_rows, n_cols, d, C, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, st);
    if (d <= 32) return launch_t<32>(n_rows, n_cols, d, C, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, st);
    return launch_t<64>(n_rows, n_cols, d, C, z, g, v, lq, lqi, lse_s, lse_k, rz, cg, cv, st);
}
