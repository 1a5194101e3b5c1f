// Shape-generic ELBO forward / backward kernels (any dims, any activation, any depth).
//
// One 256-thread workgroup owns a tile of TR batch rows of one model.  Activations live in
// LDS tiles with an odd leading dimension (bank-conflict-free row access), weights are read
// through L1/L2 (they are shared by every tile of the model), hidden pre-activations are
// kept in the caller's workspace for the backward pass.  Weight gradients are written as
// per-tile slabs and summed in a fixed order by k_reduce, so results are bitwise
// reproducible run to run (no float atomics).
//
// Reference arithmetic restated here: models/nets.py:114-121 (MLP.forward), 301-314
// (Normal.sample / log_pdf), 45-52 (log_normal), 450-475, 587-599, 165-185, 224-243 (forward /
// elbo of the four model classes); backward = what autograd derives from those
// (SURVEY.md §8 a-bwd).
#include <hip/hip_runtime.h>
#include "layout.h"
#include "kernels.h"

#define NT 256
#define LOG2PI 1.8378770664093453f

__device__ __forceinline__ float act_fwd(float p, int act, float slope) {
    switch (act) {
        case IVAE_ACT_LRELU: return p > 0.f ? p : p * slope;
        case IVAE_ACT_XTANH: return tanhf(p) + slope * p;
        case IVAE_ACT_SIGMOID: return 1.f / (1.f + expf(-p));
        default: return p;
    }
}

__device__ __forceinline__ float act_bwd(float p, int act, float slope) {
    switch (act) {
        case IVAE_ACT_LRELU: return p > 0.f ? 1.f : slope;
        case IVAE_ACT_XTANH: { float t = tanhf(p); return 1.f - t * t + slope; }
        case IVAE_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-p)); return s * (1.f - s); }
        default: return 1.f;
    }
}

// out[r][n] = epi(bias[n] + sum_k in[r][k] * W[n][k]); thread = (row r, 4 output columns).
// epi: 0 none, 1 activation, 2 exp.
__device__ void dense_fwd(const float* in, int ld_in, int K, const float* __restrict__ W,
                          const float* __restrict__ bias, int N, float* out, int ld_out,
                          float* __restrict__ pre_g, int nrows, int TR, int epi, int act, float slope) {
    const int tid = threadIdx.x;
    const int r = tid & (TR - 1);
    const int grp = tid / TR;
    const int NG = NT / TR;
    const float* xr = in + r * ld_in;
    for (int n0 = grp * 4; n0 < N; n0 += NG * 4) {
        const int n1 = min(n0 + 1, N - 1), n2 = min(n0 + 2, N - 1), n3 = min(n0 + 3, N - 1);
        const float* w0 = W + (size_t)n0 * K;
        const float* w1 = W + (size_t)n1 * K;
        const float* w2 = W + (size_t)n2 * K;
        const float* w3 = W + (size_t)n3 * K;
        float a0 = bias[n0], a1 = bias[n1], a2 = bias[n2], a3 = bias[n3];
        for (int k = 0; k < K; ++k) {
            const float xv = xr[k];
            a0 = fmaf(xv, w0[k], a0);
            a1 = fmaf(xv, w1[k], a1);
            a2 = fmaf(xv, w2[k], a2);
            a3 = fmaf(xv, w3[k], a3);
        }
        const float acc[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j;
            if (n < N) {
                const float p = acc[j];
                if (pre_g && r < nrows) pre_g[(size_t)r * N + n] = p;
                float o = p;
                if (epi == 1) o = act_fwd(p, act, slope);
                else if (epi == 2) o = expf(p);
                out[r * ld_out + n] = o;
            }
        }
    }
}

// Full MLP forward from an LDS input tile; hidden tiles ping-pong between h0/h1.
__device__ void mlp_fwd(const Layout& L, const MlpDesc& D, const float* __restrict__ P, float* ws_model,
                        const float* in, int ld_in, float* h0, float* h1, int ldh,
                        float* out, int ld_out, int final_epi, int row0, int nrows, int TR) {
    const float* cur = in;
    int ldc = ld_in;
    for (int li = 0; li < L.L; ++li) {
        const bool last = li == L.L - 1;
        float* dst = last ? out : ((li & 1) ? h1 : h0);
        const int ldd = last ? ld_out : ldh;
        float* pre = last ? nullptr : ws_model + D.pre_off[li] + (size_t)row0 * D.dout[li];
        dense_fwd(cur, ldc, D.din[li], P + D.w_off[li], P + D.b_off[li], D.dout[li], dst, ldd, pre,
                  nrows, TR, last ? final_epi : 1, L.act, L.slope);
        __syncthreads();
        cur = dst;
        ldc = ldd;
    }
}

__global__ __launch_bounds__(NT) void k_forward(Layout L, IvaeCoef C, const float* __restrict__ params,
                                                const float* __restrict__ x, const float* __restrict__ u,
                                                const float* __restrict__ eps, float* __restrict__ z_out,
                                                float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TR = L.tr_fwd;
    const int ldf = L.maxw + 1, lds = L.dl + 1;
    float* xin = smem;
    float* h0 = xin + TR * ldf;
    float* h1 = h0 + TR * ldf;
    float* sg = h1 + TR * ldf;
    float* sv = sg + TR * lds;
    float* sl = sv + TR * lds;
    float* sz = sl + TR * lds;
    const int tid = threadIdx.x;
    const int model = blockIdx.y;
    const int row0 = blockIdx.x * TR;
    const int nrows = min(TR, L.B - row0);
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : nullptr;
    const float* em = eps + ((size_t)model * L.B + row0) * L.dl;
    const int Kin = L.dd + L.M;

    for (int idx = tid; idx < TR * Kin; idx += NT) {
        const int r = idx / Kin, k = idx - r * Kin;
        float val = 0.f;
        if (r < nrows) val = k < L.dd ? xm[(size_t)r * L.dd + k] : um[(size_t)r * L.M + (k - L.dd)];
        xin[r * ldf + k] = val;
    }
    __syncthreads();

    mlp_fwd(L, L.mlp[IVAE_MLP_G], P, wsm, xin, ldf, h0, h1, ldf, sg, lds, 0, row0, nrows, TR);
    mlp_fwd(L, L.mlp[IVAE_MLP_LOGV], P, wsm, xin, ldf, h0, h1, ldf, sv, lds, 2, row0, nrows, TR);
    if (L.M > 0) {
        mlp_fwd(L, L.mlp[IVAE_MLP_LOGL], P, wsm, xin + L.dd, ldf, h0, h1, ldf, sl, lds, 2, row0, nrows, TR);
    } else {
        for (int idx = tid; idx < TR * L.dl; idx += NT) sl[(idx / L.dl) * lds + idx % L.dl] = 1.f;
    }
    // reparameterisation z = eps*sqrt(v) + g      (models/nets.py:301-304, 145-149)
    for (int idx = tid; idx < TR * L.dl; idx += NT) {
        const int r = idx / L.dl, k = idx - r * L.dl;
        float zz = 0.f;
        if (r < nrows) {
            const float e = em[(size_t)r * L.dl + k];
            const float vv = sv[r * lds + k], gg = sg[r * lds + k];
            zz = e * sqrtf(vv) + gg;
            const size_t o = (size_t)(row0 + r) * L.dl + k;
            z_out[(size_t)model * L.B * L.dl + o] = zz;
            wsm[L.ws.z + o] = zz;
            wsm[L.ws.g + o] = gg;
            wsm[L.ws.v + o] = vv;
            wsm[L.ws.l + o] = sl[r * lds + k];
        }
        sz[r * lds + k] = zz;
    }
    __syncthreads();
    // decoder; its output tile lands in xin (free by now)
    mlp_fwd(L, L.mlp[IVAE_MLP_F], P, wsm, sz, lds, h0, h1, ldf, xin, ldf, 0, row0, nrows, TR);
    for (int idx = tid; idx < nrows * L.dd; idx += NT) {
        const int r = idx / L.dd, k = idx - r * L.dd;
        wsm[L.ws.f + (size_t)(row0 + r) * L.dd + k] = xin[r * ldf + k];
    }
    // per-row Gaussian log-densities   (models/nets.py:306-314, 45-52)
    if (tid < nrows) {
        const int r = tid;
        const float dv = C.decoder_var;
        const float logdv = logf(dv);
        float lpx = 0.f, lqc = 0.f, lp = 0.f;
        for (int k = 0; k < L.dd; ++k) {
            const float df = xm[(size_t)r * L.dd + k] - xin[r * ldf + k];
            lpx += -0.5f * (LOG2PI + logdv + df * df / dv);
        }
        for (int k = 0; k < L.dl; ++k) {
            const float zz = sz[r * lds + k], gg = sg[r * lds + k], vv = sv[r * lds + k], ll = sl[r * lds + k];
            const float dq = zz - gg;
            lqc += -0.5f * (LOG2PI + logf(vv) + dq * dq / vv);
            lp += -0.5f * (LOG2PI + logf(ll) + zz * zz / ll);
        }
        wsm[L.ws.lpx + row0 + r] = lpx;
        wsm[L.ws.lqc + row0 + r] = lqc;
        wsm[L.ws.lp + row0 + r] = lp;
    }
}

// ------------------------------------------------------------------------------ backward
struct In0 {          // layer-0 input of an MLP, read straight from global memory
    const float* a; int na;   // first na columns from a (row stride na)
    const float* b; int nb;   // next nb columns from b (row stride nb)
    __device__ __forceinline__ float operator()(int r, int k) const {
        return k < na ? a[(size_t)r * na + k] : b[(size_t)r * nb + (k - na)];
    }
};

// Backward of one MLP for a row tile.  delta: LDS (TR, ld) holding dLoss/d(output), rows >=
// nrows zero.  Writes the tile's weight/bias gradient slab; if dx_out != null leaves
// dLoss/d(input) there (LDS, ld_dx).
// noinline: inlining the four call sites crashes hipcc 7.2 (InstCombine segfault) and buys nothing here.
__device__ __noinline__ void mlp_bwd(const Layout& L, const MlpDesc& D, const float* __restrict__ P, const float* wsm,
                        float* slab, float* delta, float* dnext, float* hin, int ld, In0 in0,
                        float* dx_out, int ld_dx, int row0, int nrows, int TR) {
    const int tid = threadIdx.x;
    for (int li = L.L - 1; li >= 0; --li) {
        const int N = D.dout[li], K = D.din[li];
        const float* W = P + D.w_off[li];
        if (li > 0) {   // layer input = act(pre-activation of the layer below)
            const float* pre = wsm + D.pre_off[li - 1] + (size_t)row0 * K;
            for (int idx = tid; idx < TR * K; idx += NT) {
                const int r = idx / K, k = idx - r * K;
                hin[r * ld + k] = r < nrows ? act_fwd(pre[(size_t)r * K + k], L.act, L.slope) : 0.f;
            }
            __syncthreads();
        }
        // weight gradient dW[n][k] = sum_r delta[r][n] * in[r][k]
        float* gW = slab + D.w_off[li];
        for (int idx = tid; idx < N * K; idx += NT) {
            const int n = idx / K, k = idx - n * K;
            float acc = 0.f;
            if (li > 0) {
                for (int r = 0; r < nrows; ++r) acc = fmaf(delta[r * ld + n], hin[r * ld + k], acc);
            } else {
                for (int r = 0; r < nrows; ++r) acc = fmaf(delta[r * ld + n], in0(r, k), acc);
            }
            gW[idx] = acc;
        }
        float* gB = slab + D.b_off[li];
        for (int n = tid; n < N; n += NT) {
            float acc = 0.f;
            for (int r = 0; r < nrows; ++r) acc += delta[r * ld + n];
            gB[n] = acc;
        }
        // input gradient
        const bool need = li > 0 || dx_out != nullptr;
        if (need) {
            float* dst = li > 0 ? dnext : dx_out;
            const int ldd = li > 0 ? ld : ld_dx;
            const float* pre = li > 0 ? wsm + D.pre_off[li - 1] + (size_t)row0 * K : nullptr;
            for (int idx = tid; idx < TR * K; idx += NT) {
                const int r = idx / K, k = idx - r * K;
                float acc = 0.f;
                if (r < nrows) {
                    for (int n = 0; n < N; ++n) acc = fmaf(delta[r * ld + n], W[(size_t)n * K + k], acc);
                    if (li > 0) acc *= act_bwd(pre[(size_t)r * K + k], L.act, L.slope);
                }
                dst[r * ldd + k] = acc;
            }
        }
        __syncthreads();
        if (li > 0) { float* t = delta; delta = dnext; dnext = t; }
    }
}

__device__ void copy_delta(const float* src, int lds, float* dst, int ld, int TR, int dl) {
    for (int idx = threadIdx.x; idx < TR * dl; idx += NT) {
        const int r = idx / dl, k = idx - r * dl;
        dst[r * ld + k] = src[r * lds + k];
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void k_backward(Layout L, IvaeCoef C, const float* __restrict__ params,
                                                 const float* __restrict__ x, const float* __restrict__ u,
                                                 const float* __restrict__ eps, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TR = L.tr_bwd;
    const int ld = L.maxh + 1, lds = L.dl + 1;
    float* hin = smem;
    float* dA = hin + TR * ld;
    float* dB = dA + TR * ld;
    float* sdz = dB + TR * ld;       // dLoss/dz through the decoder, then Gz
    float* sdg = sdz + TR * lds;
    float* sdv = sdg + TR * lds;
    float* sdl = sdv + TR * lds;
    const int tid = threadIdx.x;
    const int model = blockIdx.y;
    const int row0 = blockIdx.x * TR;
    const int nrows = min(TR, L.B - row0);
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    float* slab = wsm + L.ws_partial + (size_t)blockIdx.x * L.n_params;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : nullptr;
    const float* em = eps + ((size_t)model * L.B + row0) * L.dl;
    const float w = 1.f / (float)C.global_batch;

    // dLoss/df = -a*w*(x-f)/var
    const float* fm = wsm + L.ws.f + (size_t)row0 * L.dd;
    for (int idx = tid; idx < TR * L.dd; idx += NT) {
        const int r = idx / L.dd, k = idx - r * L.dd;
        dA[r * ld + k] = r < nrows ? -C.a * w * (xm[(size_t)r * L.dd + k] - fm[(size_t)r * L.dd + k]) / C.decoder_var : 0.f;
    }
    __syncthreads();
    In0 zin{wsm + L.ws.z + (size_t)row0 * L.dl, L.dl, nullptr, 0};
    mlp_bwd(L, L.mlp[IVAE_MLP_F], P, wsm, slab, dA, dB, hin, ld, zin, sdz, lds, row0, nrows, TR);

    // output-layer deltas of the encoder / prior heads (SURVEY.md §8 a-bwd)
    for (int idx = tid; idx < TR * L.dl; idx += NT) {
        const int r = idx / L.dl, k = idx - r * L.dl;
        float dg = 0.f, dv = 0.f, dlg = 0.f;
        if (r < nrows) {
            const size_t o = (size_t)(row0 + r) * L.dl + k;
            const float zz = wsm[L.ws.z + o], gg = wsm[L.ws.g + o], vv = wsm[L.ws.v + o], ll = wsm[L.ws.l + o];
            const float e = em[(size_t)r * L.dl + k];
            const float dq = zz - gg;
            const float rr = dq / vv;
            float Gz = sdz[r * lds + k] - w * (C.b * rr - C.d * zz / ll);
            dg = w * C.b * rr;
            dv = -w * C.b * 0.5f * (1.f - dq * dq / vv);
            if (C.pairwise) {
                Gz -= w * wsm[L.ws.rz + o];
                dg -= w * wsm[L.ws.cg + o];
                dv -= w * wsm[L.ws.cv + o];
            }
            dg += Gz;
            dv += Gz * 0.5f * e * sqrtf(vv);
            dlg = w * C.d * 0.5f * (1.f - zz * zz / ll);
        }
        sdg[r * lds + k] = dg;
        sdv[r * lds + k] = dv;
        sdl[r * lds + k] = dlg;
    }
    __syncthreads();
    // each head's delta goes into a wide tile (mlp_bwd ping-pongs delta/dnext with stride ld)
    const In0 xuin{xm, L.dd, um, L.M};
    const In0 uin{um, L.M, nullptr, 0};
    copy_delta(sdg, lds, dA, ld, TR, L.dl);
    mlp_bwd(L, L.mlp[IVAE_MLP_G], P, wsm, slab, dA, dB, hin, ld, xuin, nullptr, 0, row0, nrows, TR);
    copy_delta(sdv, lds, dA, ld, TR, L.dl);
    mlp_bwd(L, L.mlp[IVAE_MLP_LOGV], P, wsm, slab, dA, dB, hin, ld, xuin, nullptr, 0, row0, nrows, TR);
    if (L.M > 0) {
        copy_delta(sdl, lds, dA, ld, TR, L.dl);
        mlp_bwd(L, L.mlp[IVAE_MLP_LOGL], P, wsm, slab, dA, dB, hin, ld, uin, nullptr, 0, row0, nrows, TR);
    }
}

// Sum the per-tile slabs (fixed order) into grads and reduce the loss.
__global__ __launch_bounds__(NT) void k_reduce(Layout L, IvaeCoef C, const float* __restrict__ ws,
                                               float* __restrict__ grads, float* __restrict__ loss_out) {
    const int model = blockIdx.y;
    const float* wsm = ws + (size_t)model * L.ws.per_model;
    const int nblk_p = (L.n_params + NT - 1) / NT;
    if ((int)blockIdx.x < nblk_p) {
        const int p = blockIdx.x * NT + threadIdx.x;
        if (p < L.n_params) {
            const float* s = wsm + L.ws_partial + p;
            float acc = 0.f;
            for (int t = 0; t < L.n_tiles_bwd; ++t) acc += s[(size_t)t * L.n_params];
            grads[(size_t)model * L.n_params + p] = acc;
        }
        return;
    }
    if (!loss_out) return;
    __shared__ float red[NT];
    float acc = 0.f;
    for (int i = threadIdx.x; i < L.B; i += NT) {
        const float lpx = wsm[L.ws.lpx + i], lqc = wsm[L.ws.lqc + i], lp = wsm[L.ws.lp + i];
        float row;
        if (C.pairwise) {
            const float lq = wsm[L.ws.lq + i], lqi = wsm[L.ws.lqi + i];
            row = C.a * lpx - C.b * (lqc - lq) - C.c * (lq - lqi) - C.d * (lqi - lp);
        } else {
            row = C.a * lpx - C.b * lqc + C.d * lp;
        }
        acc += row;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[model] = -red[0] / (float)C.global_batch;
}

// ------------------------------------------------------------------------------ launchers
static size_t smem_fwd(const Layout& L) { return (size_t)L.tr_fwd * (3 * (L.maxw + 1) + 4 * (L.dl + 1)) * sizeof(float); }
static size_t smem_bwd(const Layout& L) { return (size_t)L.tr_bwd * (3 * (L.maxh + 1) + 4 * (L.dl + 1)) * sizeof(float); }

hipError_t launch_forward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                          const float* eps, float* z_out, float* ws, hipStream_t st) {
    dim3 grid(L.n_tiles_fwd, L.n_models);
    hipLaunchKernelGGL(k_forward, grid, dim3(NT), smem_fwd(L), st, L, C, params, x, u, eps, z_out, ws);
    return hipGetLastError();
}

hipError_t launch_backward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                           const float* eps, float* loss_out, float* grads, float* ws, hipStream_t st) {
    dim3 grid(L.n_tiles_bwd, L.n_models);
    hipLaunchKernelGGL(k_backward, grid, dim3(NT), smem_bwd(L), st, L, C, params, x, u, eps, ws);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 g2((L.n_params + NT - 1) / NT + 1, L.n_models);
    hipLaunchKernelGGL(k_reduce, g2, dim3(NT), 0, st, L, C, ws, grads, loss_out);
    return hipGetLastError();
}
