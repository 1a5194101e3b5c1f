// Shape-generic ELBO forward / backward kernels on the fp32 matrix cores (any dims, activation, depth).
//
// One workgroup of NW = 4 / 8 / 16 waves owns a tile of TR = 16/32/64 batch rows of one model (k_wgrad: 16 .. 256 rows).
// Every layer product (forward, input gradient, weight gradient) is the same "row-row" contraction
//
//        out[i][j] = sum_k  A[i][k] * Bm[j][k]          (v_mfma_f32_16x16x4_f32, exact fp32 FMA chain)
//
// with A resident in LDS (activation / delta tiles, row stride 4*odd floats so that 128-bit operand reads spread
// over the banks) and Bm streamed through LDS in chunks of 16 NW output columns x 64 contraction steps:
//   forward      A = layer input  [row][k]      Bm = W[n][k]  (+ bias as column K: inputs carry a constant-1 column)
//   dgrad        A = delta        [row][n]      Bm = W^T[k][n]
//   wgrad        A = delta^T      [n][row]      Bm = input^T[k][row]  (+ a row of ones: bias gradient)
// A lane loads 4 consecutive k of its row as one ds_read_b128 and feeds them to 4 MFMAs; both operands use the same
// k permutation, so sums are unchanged up to fp32 order.  The chunks are double-buffered in LDS and their global loads
// are issued one chunk ahead into registers (also across layer boundaries), so the L2 latency of the weights hides
// behind the previous chunk's MFMAs.  Hidden activations and their derivatives go to the caller's workspace feature-major
// ((H, Bp), Bp = batch rounded up to 64) so that the backward pass reads them as transposed operands directly.
// The step is split by data dependency: k_fwd_enc (three MLP chains as concurrent workgroups), k_dec (decoder forward,
// row log-densities, decoder input-gradient chain), k_bwd_enc (head deltas + input-gradient chains), then k_wgrad --
// nothing on the dependent chain of layers waits for a weight gradient, so every layer's is its own workgroup, fed by
// the layer deltas the chains left in the workspace.  Weight gradients are written as per-tile slabs and summed in a
// fixed order by k_reduce: results are bitwise reproducible run to run (no float atomics).
//
// Reference arithmetic restated here: models/nets.py:114-121 (MLP.forward), 301-314 (Normal.sample / log_pdf),
// 45-52 (log_normal), 450-475, 587-599, 165-185, 224-243 (forward / elbo of the four model classes); backward =
// what autograd derives from those (SURVEY.md §8 a-bwd).
#include <hip/hip_runtime.h>
#include "layout.h"
#include "kernels.h"

#define NR 256            // threads of a loss reduction
#define KCH 64            // contraction extent of a staged chunk
#define LDC 68            // chunk row stride (4 * odd)
// Geometry of a workgroup of NW waves (4, 8 or 16): NT threads, chunks of NCH = 16 NW rows (output columns) x KCH
#define GEOM(NW) constexpr int NT = 64 * (NW), NCH = 16 * (NW), CHUNK = NCH * LDC; (void)NT; (void)NCH; (void)CHUNK
#define LOG2PI 1.8378770664093453f

// Diagnostic build only (-DMF_PROF): s_memtime stamps of workgroup 0 / thread 0 go to a buffer of their own; no
// output depends on them and the shipped library contains none.
#ifdef MF_PROF
__device__ unsigned long long* g_mf_prof = nullptr;
#define MSTAMP(id)                                                                                  \
    do {                                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && g_mf_prof && St.n < 2000) {    \
            unsigned long long t_;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_mf_prof[2 * St.n] = (id);                                                             \
            g_mf_prof[2 * St.n + 1] = t_;                                                           \
            ++St.n;                                                                                 \
        }                                                                                           \
    } while (0)
extern "C" int ivae_debug_set_prof_mf(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mf_prof), &buf, sizeof(buf));
}
#else
#define MSTAMP(id)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ZERO4 ((f32x4){0.f, 0.f, 0.f, 0.f})

__device__ __forceinline__ int up16(int x) { return (x + 15) & ~15; }

// activation h = act(p) and its derivative d = act'(p) from ONE evaluation of the transcendental: the forward epilogue
// stores both (feature-major, workspace), so the backward pass evaluates no tanh / exp at all
__device__ __forceinline__ void act_both(float p, int act, float slope, float& h, float& d) {
    switch (act) {
        case IVAE_ACT_LRELU: h = p > 0.f ? p : p * slope; d = p > 0.f ? 1.f : slope; break;
        case IVAE_ACT_XTANH: { const float t = tanhf(p); h = t + slope * p; d = 1.f - t * t + slope; break; }
        case IVAE_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-p)); h = s; d = s * (1.f - s); break; }
        default: h = p; d = 1.f;
    }
}

// ------------------------------------------------------------------------------ streamed operand
// Source of the streamed operand Bm[j][k] (j: output column, k: contraction index) of one product.
//   SRC_W    forward:  Bm[j][k] = W[j][k] (W is (J, K) row-major at p), Bm[j][K] = bias[j] (at p + d2): the A tiles
//            carry a constant-1 column at k = K.  k in [hole0, hole0 + hole) is padding (zero) and later k shift
//            down by `hole` ([x | pad | u | 1] input tiles).
//   SRC_WT   dgrad:    Bm[j][k] = W[k][j]   (W is (K, J) row-major at p, row stride ld)
//   SRC_PRE  wgrad of a hidden layer:  Bm[j][k] = actT[j][k] (hidden activations feature-major at p, row stride ld,
//            k = batch row < K), Bm[J][k] = 1 (its result row is the bias gradient)
//   SRC_ROWS wgrad of a first layer:   Bm[j][k] = in[k][j], in = [a | b] row-major blocks (a: J1 columns at p, row
//            stride ld; b: J - J1 columns at p + d2, row stride ld2), Bm[J][k] = 1
// Everything outside the stated extents is zero.
enum { SRC_NONE = 0, SRC_W, SRC_WT, SRC_PRE, SRC_ROWS };
struct BSrc {
    const float* p;
    long d2;
    int kind;
    int J1, J, K;
    int ld, ld2;
    int hole0, hole;
    unsigned lim;                   // SRC_W / SRC_PRE: floats readable from p (range of the buffer descriptor)
};

// One 64 x 64 chunk (j0.., k0..) moves through 16 (+4) registers per thread.  LOAD: issue the global loads (unconditional:
// nothing here waits for memory).  !LOAD: registers -> LDS chunk buffer wb, applying the zero / one fill and the
// activation.  Thread (jq = tid >> 4, kq = tid & 15):
//   direct kinds (W, PRE: global memory contiguous along k): rows j = jq + 16e, e < 4, columns k = 4kq..4kq+3 as ONE
//     buffer_load_dwordx4 (dword-aligned; the descriptor's range check returns 0 past the arena, so no clamping) and ONE
//     ds_write_b128; per-row offsets are scalar (soffset), so a chunk costs one address computation per thread;
//   transposing kinds (WT, ROWS: contiguous along j), thread (w = tid >> 6, l = tid & 63), 16 scalar elements:
//     j = (l & 7) + 8w + 32(e & 1), k = (l >> 3) + 8(e >> 1)  (8 consecutive j x 8 k per wave keep both the 32-B global
//     segments and the LDS banks of the transposed store distinct); out-of-range elements read a clamped address.
template <bool LOAD, int NW>
__device__ __forceinline__ void chunk_io(float (&r)[16], float (&rb)[4], const BSrc& S, int j0, int k0, float* wb) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // opaque: keeps the per-element addresses of every call site from being hoisted
                                        // out of the layer loops (hundreds of live registers otherwise)
    const int w = tid >> 6, l = tid & 63;
    const int jq = tid >> 4, kq = tid & 15;
    const float* __restrict__ p = S.p;
    if (S.kind == SRC_W || S.kind == SRC_PRE) {
        const bool isw = S.kind == SRC_W;
        const int kf = k0 + 4 * kq;                                  // contraction index of my first column
        const int sh = (isw && kf >= S.hole0 + S.hole) ? S.hole : 0;   // (hole0 + hole is a multiple of 4: uniform per group)
        const int c0 = kf - sh;                                      // its column in memory
        if (LOAD) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(S.lim * 4u), 0x00020000);
            const int vo = ((j0 + jq) * S.ld + c0) * 4, vb = ((int)S.d2 + j0 + jq) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, e * 16 * NW * S.ld, 0));
                r[4 * e] = t[0]; r[4 * e + 1] = t[1]; r[4 * e + 2] = t[2]; r[4 * e + 3] = t[3];
                if (isw) rb[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vb, e * 16 * NW, 0));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + jq + 4 * NW * e;
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = kf + i, cc = c0 + i;
                    float v;
                    if (isw) {
                        const bool ok = j < S.J && !(k >= S.hole0 && k < S.hole0 + S.hole);
                        v = (ok && cc < S.K) ? r[4 * e + i] : (ok && cc == S.K) ? rb[e] : 0.f;
                    } else {
                        v = (k < S.K && j < S.J) ? r[4 * e + i] : (k < S.K && j == S.J) ? 1.f : 0.f;
                    }
                    o[i] = v;
                }
                *reinterpret_cast<f32x4*>(wb + (jq + 4 * NW * e) * LDC + 4 * kq) = o;
            }
        }
    } else if (S.kind == SRC_WT) {
        const int jl = (l & 7) + 8 * w, kl = l >> 3;
        const int ja = j0 + jl, jb = ja + 8 * NW;
        const int oj[2] = {min(ja, S.J - 1), min(jb, S.J - 1)};
        const bool vj[2] = {ja < S.J, jb < S.J};
        float* dst = wb + jl * LDC + kl;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = k0 + kl + 8 * (e >> 1);
            if (LOAD) r[e] = p[min(k, S.K - 1) * S.ld + oj[e & 1]];
            else dst[8 * NW * (e & 1) * LDC + 8 * (e >> 1)] = (vj[e & 1] && k < S.K) ? r[e] : 0.f;
        }
    } else if (S.kind == SRC_ROWS) {
        const int jl = (l & 7) + 8 * w, kl = l >> 3;
        const bool two = S.J > S.J1;
        const int ja = j0 + jl, jb = ja + 8 * NW;
        // column offset inside a row of block a (stride ld) or block b (stride ld2, at p + d2)
        const bool inb[2] = {two && ja >= S.J1, two && jb >= S.J1};
        const long cj[2] = {inb[0] ? S.d2 + (min(ja, S.J - 1) - S.J1) : (long)min(ja, S.J1 - 1),
                            inb[1] ? S.d2 + (min(jb, S.J - 1) - S.J1) : (long)min(jb, S.J1 - 1)};
        const int rs[2] = {inb[0] ? S.ld2 : S.ld, inb[1] ? S.ld2 : S.ld};
        const int vj[2] = {ja < S.J ? 1 : ja == S.J ? 2 : 0, jb < S.J ? 1 : jb == S.J ? 2 : 0};
        float* dst = wb + jl * LDC + kl;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = k0 + kl + 8 * (e >> 1);
            if (LOAD) r[e] = p[cj[e & 1] + (long)(min(k, S.K - 1) * rs[e & 1])];
            else dst[8 * NW * (e & 1) * LDC + 8 * (e >> 1)] = k < S.K ? (vj[e & 1] == 1 ? r[e] : vj[e & 1] == 2 ? 1.f : 0.f) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------ the contraction
struct Stream {           // chunk pipeline state carried from one product to the next
    float r[16];          // prefetched first chunk of the next product
    float rb[4];          //   (+ its bias entries)
    float* wbuf;          // two chunk buffers
    int par;              // which one the next chunk goes to
    int n;                // diagnostic build: stamps written so far
};
template <int NW>
__device__ __forceinline__ void chunk_load(Stream& St, const BSrc& S, int j0, int k0) {
    if (S.kind != SRC_NONE) chunk_io<true, NW>(St.r, St.rb, S, j0, k0, nullptr);
}
template <int NW>
__device__ __forceinline__ void chunk_store(Stream& St, const BSrc& S, int j0, int k0, float* wb) {
    chunk_io<false, NW>(St.r, St.rb, S, j0, k0, wb);
}

// out[i][j] = sum_k A[i][k] * Bm[j][k] for i < 16*RT (RT <= RTE row tiles), j < Jn, k < Kc.
// RTE (1, 2 or 4) fixes the wave -> tile map at compile time: wave w owns row tile w % RTE and the NTW = RTE column
// tiles (w / RTE) + (4 / RTE) * t of every 64-column block, so the MFMA loop has no data-dependent control flow; tiles
// and k-steps beyond the real extents multiply zero-filled chunk rows / columns (the epilogue drops them).
// On entry St.r holds chunk (0, 0) of S; on exit it holds chunk (0, 0) of `next` (if any).
// epi(rt, col0, acc, aux): lane (c, q) holds out[16*rt + 4q + r][col0 + c], r = 0..3; aux = pre(rt, col0), fetched
// BEFORE the contraction so that global operands of the epilogue (activation derivatives of the dgrad mask) arrive under it.
struct NoPre { __device__ __forceinline__ f32x4 operator()(int, int) const { return ZERO4; } };
template <int RTE, int NW, class EPI, class PRE = NoPre>
__device__ __forceinline__ void tile_gemm(const float* A, int lda, int RT, const BSrc& S, int Jn, int Kc, const BSrc& next,
                                          Stream& St, EPI&& epi, PRE&& pre = PRE()) {
    GEOM(NW);
    constexpr int WN = NW / RTE, NTW = RTE;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int rt = wave % RTE, cg = wave / RTE;
    const bool active = rt < RT;
    const int K16 = up16(Kc);
    for (int j0 = 0; j0 < Jn; j0 += NCH) {
        f32x4 acc[NTW], aux[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) { acc[t] = ZERO4; aux[t] = active ? pre(rt, j0 + 16 * (cg + WN * t)) : ZERO4; }
        for (int k0 = 0; k0 < K16; k0 += KCH) {
            const int nk = min(KCH, K16 - k0) >> 4;
            float* wb = St.wbuf + St.par * CHUNK;
            St.par ^= 1;
            MSTAMP(1);
            chunk_store<NW>(St, S, j0, k0, wb);
            MSTAMP(2);
            __syncthreads();
            MSTAMP(3);
            // prefetch: next k chunk, else next column block, else the following product's first chunk
            if (k0 + KCH < K16) chunk_load<NW>(St, S, j0, k0 + KCH);
            else if (j0 + NCH < Jn) chunk_load<NW>(St, S, j0 + NCH, 0);
            else chunk_load<NW>(St, next, 0, 0);
            MSTAMP(4);
            if (active) {
                const float* ap = A + (16 * rt + c) * lda + k0 + 4 * q;
                const float* bp = wb + (16 * cg + c) * LDC + 4 * q;
                // operands of k-step i+1 are read before the MFMAs of step i
                f32x4 av = *reinterpret_cast<const f32x4*>(ap), bv[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp + t * WN * 16 * LDC);
                for (int i = 0; i < nk; ++i) {
                    const f32x4 a0 = av;
                    f32x4 b0[NTW];
#pragma unroll
                    for (int t = 0; t < NTW; ++t) b0[t] = bv[t];
                    const int in = min(i + 1, nk - 1) * 16;       // (the last step re-reads itself: branch-free)
                    av = *reinterpret_cast<const f32x4*>(ap + in);
#pragma unroll
                    for (int t = 0; t < NTW; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp + t * WN * 16 * LDC + in);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int t = 0; t < NTW; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[x], b0[t][x], acc[t], 0, 0, 0);
                }
            }
        }
        MSTAMP(5);
        if (active) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) epi(rt, j0 + 16 * (cg + WN * t), acc[t], aux[t]);
        }
        MSTAMP(6);
    }
}

// ------------------------------------------------------------------------------ forward
struct FwdCtx {
    const float* P;        // this model's parameters
    float* wsm;            // this model's workspace
    int row0, nrows;
    float* h0;             // hidden tiles (TR, ldh), column H = 1
    float* h1;
};

__device__ __forceinline__ BSrc src_none() {
    BSrc S = {};
    S.hole0 = 0x3fffffff;
    return S;
}
// forward operand of layer li
__device__ __forceinline__ BSrc src_w(const Layout& L, const MlpDesc& D, const float* P, int li, bool holey) {
    BSrc S = src_none();
    S.kind = SRC_W; S.p = P + D.w_off[li]; S.d2 = D.b_off[li] - D.w_off[li];
    S.J1 = S.J = D.dout[li]; S.K = D.din[li]; S.ld = D.din[li]; S.lim = (unsigned)(L.n_params - D.w_off[li]);
    if (holey) { S.hole0 = L.dd; S.hole = L.uoff - L.dd; }
    return S;
}

// One MLP forward.  `in` (LDS, stride ld_in) holds the layer-0 input with a constant-1 column after its Kin0
// columns; `holey`: the input tile is [x | pad | u | 1] and W's columns skip the pad.  head(rt, col0, acc) receives
// the output layer's tiles.  `next`: first streamed operand of whatever product follows this MLP.
template <int RTE, int NW, class HEAD>
__device__ __forceinline__ void mlp_fwd(const Layout& L, const FwdCtx& X, const MlpDesc& D, const float* in, int ld_in, int Kin0, bool holey,
                                        const BSrc& next, Stream& St, HEAD&& head) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const float* cur = in;
    int ldc = ld_in;
    for (int li = 0; li < L.L; ++li) {
        const bool last = li == L.L - 1;
        const BSrc S = src_w(L, D, X.P, li, holey && li == 0);
        BSrc nx = next;
        if (!last) nx = src_w(L, D, X.P, li + 1, false);
        const int Kc = (li == 0 ? Kin0 : L.H) + 1;
        MSTAMP(7);
        const int N = D.dout[li];
        if (last) {
            tile_gemm<RTE, NW>(cur, ldc, RTE, S, N, Kc, nx, St, head);
        } else {
            float* dst = (li & 1) ? X.h1 : X.h0;
            float* actT = X.wsm + D.pre_off[li] + X.row0;
            float* derT = X.wsm + D.der_off[li] + X.row0;
            tile_gemm<RTE, NW>(cur, ldc, RTE, S, N, Kc, nx, St, [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                const int n = col0 + c;
                if (n < N) {
                    const int r0 = 16 * rt + 4 * q;
                    f32x4 hv, dv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float h, d;
                        act_both(acc[r], L.act, L.slope, h, d);
                        hv[r] = h; dv[r] = d;
                        dst[(r0 + r) * L.ldh + n] = h;
                    }
                    *reinterpret_cast<f32x4*>(actT + (size_t)n * L.Bp + r0) = hv;
                    *reinterpret_cast<f32x4*>(derT + (size_t)n * L.Bp + r0) = dv;
                }
            });
            cur = dst;
            ldc = L.ldh;
        }
    }
}

// Encoder / prior forward.  blockIdx.z picks the MLP: 0 = g, 1 = logv, 2 = logl; the three chains of a row tile
// run as three concurrent workgroups (a third of the dependent-layer latency, three times the workgroups).
// Writes g, v = exp(logv), l = exp(logl) rows to the workspace.
template <int RTE, int NW>
__global__ __launch_bounds__(64 * NW) void k_fwd_enc(Layout L, const float* __restrict__ params, const float* __restrict__ x,
                                                const float* __restrict__ u, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GEOM(NW);
    const int TR = L.tr;
    float* wbuf = smem;
    float* xin = wbuf + 2 * CHUNK;          // [x | pad | u | 1 | 0..]
    float* h0 = xin + TR * L.ldx;
    float* h1 = h0 + TR * L.ldh;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int model = blockIdx.y, row0 = blockIdx.x * TR, nrows = min(TR, L.B - row0);
    const int m = blockIdx.z == 0 ? IVAE_MLP_G : blockIdx.z == 1 ? IVAE_MLP_LOGV : IVAE_MLP_LOGL;
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : nullptr;
    const int Kx = L.uoff + L.M;            // contraction extent of the encoder's first layer (bias column index)
    const bool enc = m != IVAE_MLP_LOGL;

    Stream St;
    St.wbuf = wbuf; St.par = 0; St.n = 0;
    MSTAMP(100);
    chunk_load<NW>(St, src_w(L, L.mlp[m], P, 0, enc && L.M > 0), 0, 0);

    for (int i = 4 * tid; i < L.smem_enc / 4 - 2 * CHUNK; i += 4 * NT) *reinterpret_cast<f32x4*>(xin + i) = ZERO4;
    __syncthreads();
    {   // input rows -> LDS: a wave per row, lanes over the columns, four loads in flight (no division, no load
        // waited for one by one)
        const int wv = tid >> 6, W = L.dd + L.M;
        for (int r = wv; r < nrows; r += NW) {
            const float* xr = xm + (size_t)r * L.dd;
            const float* ur = um + (size_t)r * L.M - L.dd;          // indexed by k >= dd only (um is null when M == 0)
            float* dst = xin + r * L.ldx;
            for (int k0 = lane; k0 < W; k0 += 256) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + 64 * e;
                    v[e] = k < L.dd ? xr[k] : k < W ? ur[k] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + 64 * e;
                    if (k < W) dst[k < L.dd ? k : L.uoff + (k - L.dd)] = v[e];
                }
            }
        }
    }
    for (int r = tid; r < TR; r += NT) {
        xin[r * L.ldx + Kx] = 1.f;
        if (L.L > 1) { h0[r * L.ldh + L.H] = 1.f; h1[r * L.ldh + L.H] = 1.f; }
    }
    // (the first tile_gemm's barrier orders these writes before any operand read)
    FwdCtx X;
    X.P = P; X.wsm = wsm; X.row0 = row0; X.nrows = nrows; X.h0 = h0; X.h1 = h1;
    float* out = wsm + (m == IVAE_MLP_G ? L.ws.g : m == IVAE_MLP_LOGV ? L.ws.v : L.ws.l) + (size_t)row0 * L.dl;
    mlp_fwd<RTE, NW>(L, X, L.mlp[m], enc ? xin : xin + L.uoff, L.ldx, enc ? Kx : L.M, enc && L.M > 0, src_none(), St,
                 [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                     const int n = col0 + c;
                     if (n < L.dl) {
#pragma unroll
                         for (int r = 0; r < 4; ++r) {
                             const int row = 16 * rt + 4 * q + r;
                             if (row < nrows) out[(size_t)row * L.dl + n] = m == IVAE_MLP_G ? acc[r] : expf(acc[r]);
                         }
                     }
                 });
    MSTAMP(102);
}

// ------------------------------------------------------------------------------ backward
struct BwdCtx {
    const float* P;
    float* wsm;
    float* slab;
    int row0, nrows, TR;
    float* d0;             // delta tiles, row-major (TR, ldd)
    float* d1;
};

// Streamed operand of the input-gradient product of layer li: W^T.
__device__ __forceinline__ BSrc dgrad_src(const BwdCtx& X, const MlpDesc& D, int li) {
    BSrc S = src_none();
    S.kind = SRC_WT; S.p = X.P + D.w_off[li]; S.J1 = S.J = D.din[li]; S.K = D.dout[li]; S.ld = D.din[li];
    return S;
}
// Streamed operand of the weight-gradient product of layer li: the layer's input, transposed (hidden activations from
// the workspace, or the layer-0 input rows in_a (width na) followed by in_b (width nb) from global memory).
__device__ __forceinline__ BSrc wgrad_src(const Layout& L, const BwdCtx& X, const MlpDesc& D, int li, const float* in_a, int na,
                                          const float* in_b, int nb) {
    BSrc S = src_none();
    if (li > 0) {
        S.kind = SRC_PRE; S.p = X.wsm + D.pre_off[li - 1] + X.row0; S.J1 = S.J = D.din[li]; S.K = X.nrows; S.ld = L.Bp;
        { const long lim = L.ws.per_model - (D.pre_off[li - 1] + X.row0); S.lim = (unsigned)(lim < 0x3fffffffL ? lim : 0x3fffffffL); }
    } else {
        S.kind = SRC_ROWS; S.p = in_a; S.d2 = in_b ? in_b - in_a : 0L; S.J1 = na; S.J = na + nb; S.K = X.nrows;
        S.ld = na; S.ld2 = nb;
    }
    return S;
}

// Input-gradient chain of one MLP for a row tile.  On entry X.d0 holds dLoss/d(output) (rows >= nrows zero) and the
// caller has stored it feature-major at D.dlt_off[L-1].  Every hidden layer's delta goes to the workspace feature-major
// (D.dlt_off[li], (dout, Bp)): the weight gradients -- which nothing in this chain waits for -- are formed from those by
// k_wgrad, one workgroup per (row tile, layer), after the chains.  When dx_out != null the gradient with respect to the
// MLP's input is left there (LDS, stride ld_dx).
template <int RTE, int NW>
__device__ __forceinline__ void mlp_bwd(const Layout& L, const BwdCtx& X, const MlpDesc& D, float* dx_out, int ld_dx, const BSrc& next,
                                        Stream& St) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int n_prod = dx_out ? L.L : L.L - 1;
    int cur = 0;
    for (int t = 0; t < n_prod; ++t) {
        const int li = L.L - 1 - t;
        const int N = D.dout[li], K = D.din[li];
        const BSrc S = dgrad_src(X, D, li);
        BSrc nx = next;
        if (t + 1 < n_prod) nx = dgrad_src(X, D, li - 1);
        MSTAMP(7);
        // dX[r][k] = sum_n delta[r][n] * W[n][k]  (times act'(pre) for a hidden input)
        const float* dcur = cur ? X.d1 : X.d0;
        if (li > 0) {
            float* dnx = cur ? X.d0 : X.d1;
            const float* derT = X.wsm + D.der_off[li - 1] + X.row0;
            float* dltT = X.wsm + D.dlt_off[li - 1] + X.row0;
            tile_gemm<RTE, NW>(dcur, L.ldd, RTE, S, K, N, nx, St, [&](int rt, int col0, const f32x4& acc, const f32x4& pre) {
                const int k = col0 + c;
                if (k < K) {
                    const int r0 = 16 * rt + 4 * q;
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = r0 + r < X.nrows ? acc[r] * pre[r] : 0.f;
                        dnx[(r0 + r) * L.ldd + k] = o[r];
                    }
                    *reinterpret_cast<f32x4*>(dltT + (size_t)k * L.Bp + r0) = o;
                }
            }, [&](int rt, int col0) {
                const int k = min(col0 + c, K - 1);
                return *reinterpret_cast<const f32x4*>(derT + (size_t)k * L.Bp + 16 * rt + 4 * q);
            });
            cur ^= 1;
        } else {
            tile_gemm<RTE, NW>(dcur, L.ldd, RTE, S, K, N, nx, St, [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                const int k = col0 + c;
                if (k < K) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dx_out[(16 * rt + 4 * q + r) * ld_dx + k] = acc[r];
                }
            });
        }
    }
}

// Weight / bias gradient of ONE layer of one MLP for a row tile: dW[n][k] = sum_r delta[r][n] * in[r][k], column K of
// the result is the bias gradient.  blockIdx.z = mlp * L + layer (mlp order f, g, logv, logl).  delta^T comes from the
// workspace (written by k_dec / k_bwd_enc), the layer input streams through LDS as in every other product; the result
// goes to the tile's gradient slab.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_wgrad(Layout L, const float* __restrict__ params, const float* __restrict__ x,
                                                   const float* __restrict__ u, float* __restrict__ ws,
                                                   float* __restrict__ slabs, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GEOM(NW);
    const int TR = L.trw;                   // rows of this kernel's tiles (its own tiling: the contraction length)
    float* wbuf = smem;
    float* dT = wbuf + 2 * CHUNK;           // delta transposed (n, ldr)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int model = blockIdx.y, row0 = blockIdx.x * TR, nrows = min(TR, L.B - row0);
    const int mi = blockIdx.z / L.L, li = blockIdx.z - mi * L.L;
    const int m = mi == 0 ? IVAE_MLP_F : mi == 1 ? IVAE_MLP_G : mi == 2 ? IVAE_MLP_LOGV : IVAE_MLP_LOGL;
    const MlpDesc& D = L.mlp[m];
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : nullptr;
    const float* zrows = wsm + L.ws.z + (size_t)row0 * L.dl;

    BwdCtx X;
    X.P = params + (size_t)model * L.n_params; X.wsm = wsm;
    X.slab = slabs + (size_t)model * slab_stride + (size_t)blockIdx.x * L.slab_ld;
    X.row0 = row0; X.nrows = nrows; X.TR = TR; X.d0 = X.d1 = nullptr;
    const float* in_a = m == IVAE_MLP_F ? zrows : m == IVAE_MLP_LOGL ? um : xm;
    const int na = m == IVAE_MLP_F ? L.dl : m == IVAE_MLP_LOGL ? L.M : L.dd;
    const float* in_b = (m == IVAE_MLP_G || m == IVAE_MLP_LOGV) ? um : nullptr;
    const int nb = in_b ? L.M : 0;
    const int N = D.dout[li], K = D.din[li];
    const BSrc S = wgrad_src(L, X, D, li, in_a, na, in_b, nb);

    Stream St;
    St.wbuf = wbuf; St.par = 0; St.n = 1500;      // (diagnostic stamps of this kernel go to slots 1500..)
    chunk_load<NW>(St, S, 0, 0);
    for (int i = 4 * tid; i < L.smem_wg / 4 - 2 * CHUNK; i += 4 * NT) *reinterpret_cast<f32x4*>(dT + i) = ZERO4;
    __syncthreads();
    const float* dlt = wsm + D.dlt_off[li] + row0;
    {   // (TR is a power of two; four loads in flight)
        const int sh = __ffs(TR) - 1, tot = N * TR;
        for (int i0 = tid; i0 < tot; i0 += 4 * NT) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = i0 + e * NT, n = idx >> sh, r = idx & (TR - 1);
                v[e] = (idx < tot && r < nrows) ? dlt[(size_t)n * L.Bp + r] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = i0 + e * NT, n = idx >> sh, r = idx & (TR - 1);
                if (idx < tot) dT[n * L.ldr + r] = v[e];
            }
        }
    }
    // (the first tile_gemm's barrier orders these writes before any operand read)
    float* gW = X.slab + D.w_off[li];
    float* gB = X.slab + D.b_off[li];
    const int NTl = (N + 15) >> 4;
    for (int rc = 0; rc < NTl; rc += 4) {
        BSrc nx = src_none();
        if (rc + 4 < NTl) nx = S;
        tile_gemm<4, NW>(dT + rc * 16 * L.ldr, L.ldr, min(4, NTl - rc), S, K + 1, TR, nx, St,
                         [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                             const int k = col0 + c;
#pragma unroll
                             for (int r = 0; r < 4; ++r) {
                                 const int n = 16 * (rc + rt) + 4 * q + r;
                                 if (n < N) {
                                     if (k < K) gW[(size_t)n * K + k] = acc[r];
                                     else if (k == K) gB[n] = acc[r];
                                 }
                             }
                         });
    }
}

// Decoder kernel.  mode bit 0: reparameterisation, decoder forward, the three per-row log-densities; bit 1: decoder
// input-gradient chain (layer deltas of f and dLoss/dz through the decoder -> workspace).  The decoder's backward needs
// nothing from other rows, so ivae_elbo_fwd_bwd runs both bits in one launch; the encoder / prior head deltas,
// which take the pairwise terms, are formed later by k_bwd_enc.
template <int RTE, int NW>
__global__ __launch_bounds__(64 * NW) void k_dec(Layout L, IvaeCoef C, int mode, const float* __restrict__ params,
                                            const float* __restrict__ x, const float* __restrict__ eps,
                                            float* __restrict__ z_out, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GEOM(NW);
    const int TR = L.tr, lds = L.lds;
    const int ldt = max(L.ldh, L.ldd);
    float* wbuf = smem;
    float* t0 = wbuf + 2 * CHUNK;           // hidden tiles (forward) / delta tiles (backward)
    float* t1 = t0 + TR * ldt;
    float* sz = t1 + TR * ldt;              // z with its constant-1 column
    float* sdz = sz + TR * lds;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int model = blockIdx.y, row0 = blockIdx.x * TR, nrows = min(TR, L.B - row0);
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* em = eps + ((size_t)model * L.B + row0) * L.dl;
    float* fm = wsm + L.ws.f + (size_t)row0 * L.dd;
    const MlpDesc& DF = L.mlp[IVAE_MLP_F];

    BwdCtx B;
    B.P = P; B.wsm = wsm; B.slab = nullptr;
    B.row0 = row0; B.nrows = nrows; B.TR = TR; B.d0 = t0; B.d1 = t1;
    const BSrc bwd_first = dgrad_src(B, DF, L.L - 1);

    Stream St;
    St.wbuf = wbuf; St.par = 0; St.n = 0;
    MSTAMP(199);
    chunk_load<NW>(St, (mode & 1) ? src_w(L, DF, P, 0, false) : bwd_first, 0, 0);
    for (int i = 4 * tid; i < L.smem_dec / 4 - 2 * CHUNK; i += 4 * NT) *reinterpret_cast<f32x4*>(t0 + i) = ZERO4;
    __syncthreads();

    if (mode & 1) {
        for (int r = tid; r < TR; r += NT) {
            if (L.L > 1) { t0[r * L.ldh + L.H] = 1.f; t1[r * L.ldh + L.H] = 1.f; }
            sz[r * lds + L.dl] = 1.f;
        }
        // reparameterisation z = eps*sqrt(v) + g      (models/nets.py:301-304, 145-149)
        for (int idx = tid; idx < TR * L.dl; idx += NT) {
            const int r = idx / L.dl, k = idx - r * L.dl;
            float zz = 0.f;
            if (r < nrows) {
                const size_t o = (size_t)(row0 + r) * L.dl + k;
                zz = em[(size_t)r * L.dl + k] * sqrtf(wsm[L.ws.v + o]) + wsm[L.ws.g + o];
                z_out[(size_t)model * L.B * L.dl + o] = zz;
                wsm[L.ws.z + o] = zz;
                if (L.M == 0) wsm[L.ws.l + o] = 1.f;
            }
            sz[r * lds + k] = zz;
        }
        FwdCtx X;
        X.P = P; X.wsm = wsm; X.row0 = row0; X.nrows = nrows; X.h0 = t0; X.h1 = t1;
        mlp_fwd<RTE, NW>(L, X, DF, sz, lds, L.dl, false, (mode & 2) ? bwd_first : src_none(), St,
                     [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                         const int n = col0 + c;
                         if (n < L.dd) {
#pragma unroll
                             for (int r = 0; r < 4; ++r) {
                                 const int row = 16 * rt + 4 * q + r;
                                 if (row < nrows) fm[(size_t)row * L.dd + n] = acc[r];
                             }
                         }
                     });
        __syncthreads();
        // per-row Gaussian log-densities   (models/nets.py:306-314, 45-52): four lanes per row take every fourth
        // feature, then two xor-shuffles add the partial sums (fixed order)
        {
            const int r = tid >> 2, part = tid & 3;
            const bool live = r < nrows;                // (tid < 4 TR <= NT: whole waves take the same branch below)
            const int rr = live ? r : 0;
            const float dv = C.decoder_var;
            const float logdv = logf(dv);
            float lpx = 0.f, lqc = 0.f, lp = 0.f;
            if (tid < 4 * TR) {
                for (int k = part; k < L.dd; k += 4) {
                    const float df = xm[(size_t)rr * L.dd + k] - fm[(size_t)rr * L.dd + k];
                    lpx += -0.5f * (LOG2PI + logdv + df * df / dv);
                }
                for (int k = part; k < L.dl; k += 4) {
                    const size_t o = (size_t)(row0 + rr) * L.dl + k;
                    const float zz = sz[rr * lds + k], gg = wsm[L.ws.g + o], vv = wsm[L.ws.v + o];
                    const float ll = L.M > 0 ? wsm[L.ws.l + o] : 1.f;
                    const float dq = zz - gg;
                    lqc += -0.5f * (LOG2PI + logf(vv) + dq * dq / vv);
                    lp += -0.5f * (LOG2PI + logf(ll) + zz * zz / ll);
                }
                lpx += __shfl_xor(lpx, 1); lqc += __shfl_xor(lqc, 1); lp += __shfl_xor(lp, 1);
                lpx += __shfl_xor(lpx, 2); lqc += __shfl_xor(lqc, 2); lp += __shfl_xor(lp, 2);
                if (live && part == 0) {
                    wsm[L.ws.lpx + row0 + r] = lpx;
                    wsm[L.ws.lqc + row0 + r] = lqc;
                    wsm[L.ws.lp + row0 + r] = lp;
                }
            }
        }
    }
    MSTAMP(200);
    if (mode & 2) {
        // dLoss/df = -a*w*(x-f)/var, all TR rows (zero beyond the batch): the tiles may hold forward activations
        const float w = 1.f / (float)C.global_batch;
        float* dltF = wsm + DF.dlt_off[L.L - 1] + row0;
        __syncthreads();
        for (int idx = tid; idx < TR * L.dd; idx += NT) {
            const int r = idx / L.dd, k = idx - r * L.dd;
            const float v = r < nrows ? -C.a * w * (xm[(size_t)r * L.dd + k] - fm[(size_t)r * L.dd + k]) / C.decoder_var : 0.f;
            t0[r * L.ldd + k] = v;
            dltF[(size_t)k * L.Bp + r] = v;
        }
        mlp_bwd<RTE, NW>(L, B, DF, sdz, lds, src_none(), St);
        MSTAMP(201);
        __syncthreads();
        for (int idx = tid; idx < nrows * L.dl; idx += NT) {
            const int r = idx / L.dl, k = idx - r * L.dl;
            wsm[L.ws_dz + (size_t)(row0 + r) * L.dl + k] = sdz[r * lds + k];
        }
    }
}

// Per-row ELBO term of row i.
__device__ __forceinline__ float loss_row(const Layout& L, const IvaeCoef& C, const float* wsm, int i) {
    const float lpx = wsm[L.ws.lpx + i], lqc = wsm[L.ws.lqc + i], lp = wsm[L.ws.lp + i];
    if (C.pairwise) {
        const float lq = wsm[L.ws.lq + i], lqi = wsm[L.ws.lqi + i];
        return C.a * lpx - C.b * (lqc - lq) - C.c * (lq - lqi) - C.d * (lqi - lp);
    }
    return C.a * lpx - C.b * lqc + C.d * lp;
}
// Thread t's share (rows t, t + stride, ...) of the sum over rows of the per-row ELBO terms; four rows' loads in flight.
__device__ __forceinline__ float loss_partial(const Layout& L, const IvaeCoef& C, const float* wsm, int t, int stride) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = t;
    for (; i + 3 * stride < L.B; i += 4 * stride) {
        a0 += loss_row(L, C, wsm, i);
        a1 += loss_row(L, C, wsm, i + stride);
        a2 += loss_row(L, C, wsm, i + 2 * stride);
        a3 += loss_row(L, C, wsm, i + 3 * stride);
    }
    for (; i < L.B; i += stride) a0 += loss_row(L, C, wsm, i);
    return (a0 + a1) + (a2 + a3);
}

// Encoder / prior backward.  blockIdx.z picks the MLP: 0 = g, 1 = logv, 2 = logl (three concurrent workgroups per row
// tile).  Forms the output-layer delta of its head (SURVEY.md §8 a-bwd; pairwise terms included) and runs the
// input-gradient chain; the deltas go to the workspace for k_wgrad.
template <int RTE, int NW>
__global__ __launch_bounds__(64 * NW) void k_bwd_enc(Layout L, IvaeCoef C, const float* __restrict__ params,
                                                const float* __restrict__ x, const float* __restrict__ u,
                                                const float* __restrict__ eps, float* __restrict__ ws,
                                                float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GEOM(NW);
    const int TR = L.tr;
    float* wbuf = smem;
    float* dA = wbuf + 2 * CHUNK;
    float* dB = dA + TR * L.ldd;
    const int tid = threadIdx.x;
    const int model = blockIdx.y, row0 = blockIdx.x * TR, nrows = min(TR, L.B - row0);
    const int mi = blockIdx.z;
    const int m = mi == 0 ? IVAE_MLP_G : mi == 1 ? IVAE_MLP_LOGV : IVAE_MLP_LOGL;
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* em = eps + ((size_t)model * L.B + row0) * L.dl;
    const float w = 1.f / (float)C.global_batch;

    BwdCtx X;
    X.P = P; X.wsm = wsm; X.slab = nullptr;
    X.row0 = row0; X.nrows = nrows; X.TR = TR; X.d0 = dA; X.d1 = dB;
    float* dltT = wsm + L.mlp[m].dlt_off[L.L - 1] + row0;

    Stream St;
    St.wbuf = wbuf; St.par = 0; St.n = 1000;      // (diagnostic stamps of this kernel go to slots 1000..)
    chunk_load<NW>(St, L.L > 1 ? dgrad_src(X, L.mlp[m], L.L - 1) : src_none(), 0, 0);
    for (int i = 4 * tid; i < L.smem_benc / 4 - 2 * CHUNK; i += 4 * NT) *reinterpret_cast<f32x4*>(dA + i) = ZERO4;
    __syncthreads();
    {   // head deltas: element idx = r * dl + k of the tile sits at (row0 * dl + idx) in every (B, dl) workspace tensor, so
        // the loads need no division; two elements per pass keep both elements' loads in flight
        const int tot = nrows * L.dl;
        const size_t ob = (size_t)row0 * L.dl;
        auto head_delta = [&](int idx) -> float {
            const size_t o = ob + idx;
            const float zz = wsm[L.ws.z + o], gg = wsm[L.ws.g + o], vv = wsm[L.ws.v + o], ll = wsm[L.ws.l + o];
            const float dq = zz - gg;
            const float rr = dq / vv;
            float Gz = wsm[L.ws_dz + o] - w * (C.b * rr - C.d * zz / ll);
            if (C.pairwise) Gz -= w * wsm[L.ws.rz + o];
            float dlt;
            if (mi == 0) {
                dlt = w * C.b * rr + Gz;
                if (C.pairwise) dlt -= w * wsm[L.ws.cg + o];
            } else if (mi == 1) {
                dlt = -w * C.b * 0.5f * (1.f - dq * dq / vv) + Gz * 0.5f * em[idx] * sqrtf(vv);
                if (C.pairwise) dlt -= w * wsm[L.ws.cv + o];
            } else {
                dlt = w * C.d * 0.5f * (1.f - zz * zz / ll);
            }
            return dlt;
        };
        auto put = [&](int idx, float dlt) {
            const int r = idx / L.dl, k = idx - r * L.dl;
            dA[r * L.ldd + k] = dlt;
            dltT[(size_t)k * L.Bp + r] = dlt;
        };
        for (int i0 = tid; i0 < tot; i0 += 2 * NT) {
            const int i1 = i0 + NT;
            const float d0 = head_delta(i0);
            const float d1 = head_delta(min(i1, tot - 1));
            put(i0, d0);
            if (i1 < tot) put(i1, d1);
        }
    }
    mlp_bwd<RTE, NW>(L, X, L.mlp[m], nullptr, 0, src_none(), St);
    MSTAMP(202);
    // loss of the model (loss_out != null: no slab reduction follows, so k_reduce is not launched): one workgroup sums
    // the per-row terms with k_reduce's thread map and order (same bits)
    if (loss_out && blockIdx.x == 0 && blockIdx.z == 0) {
        __syncthreads();
        if (tid < NR) wbuf[tid] = loss_partial(L, C, wsm, tid, NR);
        __syncthreads();
        for (int s = NR / 2; s > 0; s >>= 1) {
            if (tid < s) wbuf[tid] += wbuf[tid + s];
            __syncthreads();
        }
        if (tid == 0) loss_out[model] = -wbuf[0] / (float)C.global_batch;
    }
}

// Sum the per-tile slabs into grads and reduce the loss.  A block sums 64 parameters: wave g adds the tiles
// t = g, g+4, ... with four independent accumulators (loads in flight), then the partials are combined in a fixed
// order, so the result does not depend on scheduling.
#define RED_P 256
#define RED_W 16          // waves of a k_reduce block: wave g sums the slabs g, g + 16, ...
template <bool ADAM>
__global__ __launch_bounds__(64 * RED_W) void k_reduce(Layout L, IvaeCoef C, const float* __restrict__ ws,
                                                       float* __restrict__ grads, float* __restrict__ loss_out, AdamFuse F) {
    const int model = blockIdx.y;
    const float* wsm = ws + (size_t)model * L.ws.per_model;
    const int nblk_p = L.n_wtiles > 1 ? (L.n_params + RED_P - 1) / RED_P : 0;   // one tile: the kernels wrote grads directly
    __shared__ f32x4 red4[64 * RED_W];
    __shared__ float red[64 * RED_W];
    if ((int)blockIdx.x < nblk_p) {
        // a lane owns 4 consecutive parameters (slab rows are slab_ld = 4k floats apart and 16-byte aligned: one
        // 128-bit load per slab, 1 KB contiguous per wave); 16 waves x 4 accumulators keep 64 slabs in flight per block
        const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int p4 = min((int)blockIdx.x * RED_P + 4 * l, L.slab_ld - 4);
        const float* s = wsm + L.ws_partial + p4;
        f32x4 a0 = ZERO4, a1 = ZERO4, a2 = ZERO4, a3 = ZERO4;
        int t = g;
        for (; t + 3 * RED_W < L.n_wtiles; t += 4 * RED_W) {
            a0 += *reinterpret_cast<const f32x4*>(s + (size_t)t * L.slab_ld);
            a1 += *reinterpret_cast<const f32x4*>(s + (size_t)(t + RED_W) * L.slab_ld);
            a2 += *reinterpret_cast<const f32x4*>(s + (size_t)(t + 2 * RED_W) * L.slab_ld);
            a3 += *reinterpret_cast<const f32x4*>(s + (size_t)(t + 3 * RED_W) * L.slab_ld);
        }
        for (; t < L.n_wtiles; t += RED_W) a0 += *reinterpret_cast<const f32x4*>(s + (size_t)t * L.slab_ld);
        red4[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (g == 0) {                            // fixed-order tree over the 16 waves
            f32x4 q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                q[i] = (red4[l + 64 * (4 * i)] + red4[l + 64 * (4 * i + 1)]) + (red4[l + 64 * (4 * i + 2)] + red4[l + 64 * (4 * i + 3)]);
            const f32x4 r = (q[0] + q[1]) + (q[2] + q[3]);
            const int p0 = (int)blockIdx.x * RED_P + 4 * l;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (p0 + e < L.n_params) {
                    const size_t i = (size_t)model * L.n_params + p0 + e;
                    grads[i] = r[e];
                    if (ADAM) {          // the optimiser step of this element, by the thread that just formed its gradient
                        float pi = F.p[i], mi = F.m[i], vi = F.v[i];
                        adam_update(pi, mi, vi, r[e], F.w1, F.b2, F.w2, F.neg_step[model], F.bc2_sqrt, F.eps);
                        F.p[i] = pi;
                        F.m[i] = mi;
                        F.v[i] = vi;
                    }
                }
        }
        return;
    }
    if (!loss_out) return;
    red[threadIdx.x] = loss_partial(L, C, wsm, threadIdx.x, 64 * RED_W);
    __syncthreads();
    for (int s = 32 * RED_W; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[model] = -red[0] / (float)C.global_batch;
}

// ------------------------------------------------------------------------------ launchers
template <int RTE, int NW>
static hipError_t launch_t(const Layout& L, const IvaeCoef& C, int what, const float* params, const float* x, const float* u,
                           const float* eps, float* z_out, float* ws, float* grads, hipStream_t st) {
    static SmemCache cur_enc = {}, cur_dec = {}, cur_benc = {}, cur_wg = {};
    constexpr int NT = 64 * NW;
    const int nz = L.M > 0 ? 3 : 2;
    // gradient slabs: per-tile partials in the workspace, or the caller's gradient buffer itself when a model is one tile
    float* slabs = L.n_wtiles > 1 ? ws + L.ws_partial : grads;
    const long slab_stride = L.n_wtiles > 1 ? (long)L.ws.per_model : (long)L.n_params;
    hipError_t e;
    if (what == 0) {            // encoder / prior forward
        if ((e = set_smem((const void*)k_fwd_enc<RTE, NW>, L.smem_enc, cur_enc)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_fwd_enc<RTE, NW>), dim3(L.n_tiles, L.n_models, nz), dim3(NT), L.smem_enc, st, L, params, x, u, ws);
    } else if (what <= 3) {     // decoder: mode = what
        if ((e = set_smem((const void*)k_dec<RTE, NW>, L.smem_dec, cur_dec)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_dec<RTE, NW>), dim3(L.n_tiles, L.n_models), dim3(NT), L.smem_dec, st, L, C, what, params, x, eps, z_out, ws);
    } else if (what == 5) {     // weight gradients of every layer of every MLP
        if ((e = set_smem((const void*)k_wgrad<NW>, L.smem_wg, cur_wg)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_wgrad<NW>), dim3(L.n_wtiles, L.n_models, (nz + 1) * L.L), dim3(NT), L.smem_wg, st, L, params, x, u, ws,
                           slabs, slab_stride);
    } else {                    // encoder / prior input-gradient chains
        if ((e = set_smem((const void*)k_bwd_enc<RTE, NW>, L.smem_benc, cur_benc)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_bwd_enc<RTE, NW>), dim3(L.n_tiles, L.n_models, nz), dim3(NT), L.smem_benc, st, L, C, params, x, u, eps, ws,
                           z_out /* phase 4: the loss output, or null */);
    }
    return hipGetLastError();
}

static hipError_t launch_phase(const Layout& L, const IvaeCoef& C, int what, const float* params, const float* x, const float* u,
                               const float* eps, float* z_out, float* ws, float* grads, hipStream_t st) {
#define GO(RTE_, NW_) return launch_t<RTE_, NW_>(L, C, what, params, x, u, eps, z_out, ws, grads, st)
    if (L.nw == 8) {
        if (L.tr == 16) GO(1, 8);
        if (L.tr == 32) GO(2, 8);
        GO(4, 8);
    }
    if (L.nw == 16) {
        if (L.tr == 16) GO(1, 16);
        if (L.tr == 32) GO(2, 16);
        GO(4, 16);
    }
    if (L.tr == 16) GO(1, 4);
    if (L.tr == 32) GO(2, 4);
    GO(4, 4);
#undef GO
}

// grads_for_dec_bwd != null: also run the decoder's backward inside the decoder launch (the caller then passes
// dec_bwd_done = true to launch_backward).
hipError_t launch_forward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                          const float* eps, float* z_out, float* ws, float* grads_for_dec_bwd, hipStream_t st) {
    hipError_t e = launch_phase(L, C, 0, params, x, u, eps, z_out, ws, nullptr, st);
    if (e != hipSuccess) return e;
    return launch_phase(L, C, grads_for_dec_bwd ? 3 : 1, params, x, u, eps, z_out, ws, grads_for_dec_bwd, st);
}

hipError_t launch_backward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                           const float* eps, float* loss_out, float* grads, float* ws, bool dec_bwd_done, hipStream_t st,
                           const AdamFuse* fuse, bool* fused) {
    if (fused) *fused = false;
    hipError_t e;
    if (!dec_bwd_done && (e = launch_phase(L, C, 2, params, x, u, eps, nullptr, ws, grads, st)) != hipSuccess) return e;
    const int nblk_p = L.n_wtiles > 1 ? (L.n_params + RED_P - 1) / RED_P : 0;
    // no slabs to sum: k_bwd_enc also reduces the loss and k_reduce is not launched
    if ((e = launch_phase(L, C, 4, params, x, u, eps, nblk_p == 0 ? loss_out : nullptr, ws, grads, st)) != hipSuccess) return e;
    if ((e = launch_phase(L, C, 5, params, x, u, eps, nullptr, ws, grads, st)) != hipSuccess) return e;
    if (nblk_p == 0) return hipSuccess;
    if (fuse && L.n_models <= ADAM_MODELS_PER_LAUNCH) {
        hipLaunchKernelGGL(k_reduce<true>, dim3(nblk_p + 1, L.n_models), dim3(64 * RED_W), 0, st, L, C, ws, grads, loss_out, *fuse);
        if (fused) *fused = true;
    } else {
        hipLaunchKernelGGL(k_reduce<false>, dim3(nblk_p + 1, L.n_models), dim3(64 * RED_W), 0, st, L, C, ws, grads, loss_out, AdamFuse{});
    }
    return hipGetLastError();
}
