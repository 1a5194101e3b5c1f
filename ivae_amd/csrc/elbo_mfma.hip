// Shape-generic ELBO forward / backward kernels on the fp32 matrix cores (any dims, activation, depth).
//
// One workgroup of NW = 4 / 8 / 16 waves owns a tile of TR = 16/32/64 batch rows of one model.  Every layer product
// (forward, input gradient, weight gradient) is the same "row-row" contraction
//
//        out[i][j] = sum_k  A[i][k] * Bm[j][k]          (v_mfma_f32_16x16x4_f32, exact fp32 FMA chain)
//
//   forward      A = layer input  [row][k]      Bm = W[n][k]  (+ bias as column K: inputs carry a constant-1 column)
//   dgrad        A = delta        [row][n]      Bm = W^T[k][n]
//   wgrad        A = delta^T      [n][row]      Bm = input^T[k][row]  (+ a row of ones: bias gradient)
// Forward / dgrad: A is resident in LDS (activation / delta tiles, row stride 4*odd floats so that 128-bit operand reads
// spread over the banks); a wave owns a 16-column output tile for all row tiles of the workgroup and loads its Bm
// fragments -- the weights: a few hundred KB, L2-resident -- STRAIGHT from global memory in MFMA operand layout, a whole
// tile ahead of their use (tile_gemm).  Nothing of Bm goes through LDS and a product costs one workgroup barrier.  A lane
// takes 4 consecutive k of its row per 128-bit load and feeds them to 4 MFMAs; both operands use the same k permutation,
// so sums are unchanged up to fp32 order.  Hidden activations and their derivatives go to the caller's workspace
// feature-major ((H, Bp), Bp = batch rounded up to an odd multiple of 64) so that the backward pass reads them as
// transposed operands directly.
// The step is split by data dependency.  Launches that fill the chip run everything that depends only on a tile's own
// rows in ONE kernel (k_chain: encoder / prior chains as three concurrent wave groups -> z -> decoder -> log-densities ->
// decoder backward -> [plain ELBO] encoder / prior backward); small launches and the data-parallel phases use the same
// stages as three kernels (k_fwd_enc, k_dec, k_bwd_enc: one chain per workgroup, three times the CUs).  Nothing on the
// dependent chain of layers waits for a weight gradient: k_wgrad forms them afterwards from the layer deltas the chains
// left in the workspace -- no LDS, no barrier, one wave per 4 x 4 block of output tiles, both operands streamed from
// global memory in fragment layout -- as per-tile slabs that k_reduce sums in a fixed order: results are bitwise
// reproducible run to run (no float atomics).
//
// Reference arithmetic restated here: models/nets.py:114-121 (MLP.forward), 301-314 (Normal.sample / log_pdf),
// 45-52 (log_normal), 450-475, 587-599, 165-185, 224-243 (forward / elbo of the four model classes); backward =
// what autograd derives from those (SURVEY.md §8 a-bwd).
#if !defined(IVAE_DIAG) && (defined(MF_PROF) || defined(MF_PROF_FINE) || defined(MF_WARM_ARGS) || defined(MF_TRANSPARENT) || \
                            defined(MF_FULL_BARRIER) || defined(WG_PD) || defined(WG_OCC) || defined(CH_NW) || defined(CH_OCC))
#error "MF_* / WG_* / CH_* are switches of the diagnostic build (make diag, -DIVAE_DIAG): never into libivae_amd.so"
#endif
#include <hip/hip_runtime.h>
#include "layout.h"
#include "kernels.h"

#define NR 256            // threads of a loss reduction
// Geometry of a workgroup of NW waves (4, 8 or 16): NT threads, NCH = 16 NW output columns per pass (one tile per wave)
#define GEOM(NW) constexpr int NT = 64 * (NW), NCH = 16 * (NW); (void)NT; (void)NCH
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
// k_wgrad: stamps of the job blockIdx = (0, 0, MF_PROF_Z), slots 1500 + i
#ifndef MF_PROF_Z
#define MF_PROF_Z 3
#endif
#define WSTAMP(i)                                                                                   \
    do {                                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 64 * MF_PROF_Z && g_mf_prof) {          \
            unsigned long long t_;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_mf_prof[2 * (1500 + (i))] = (i);                                                      \
            g_mf_prof[2 * (1500 + (i)) + 1] = t_;                                                   \
        }                                                                                           \
    } while (0)
#ifdef MF_PROF_FINE
#define MSTAMPF(id) MSTAMP(id)
#else
#define MSTAMPF(id)
#endif
#else
#define MSTAMP(id)
#define MSTAMPF(id)
#define WSTAMP(i)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ZERO4 ((f32x4){0.f, 0.f, 0.f, 0.f})

#ifdef MF_WARM_ARGS
__device__ __forceinline__ void warm_kernel_args() {
    typedef const int __attribute__((address_space(4))) cint4;
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    int any = 0;
#pragma unroll
    for (int i = 0; i < (int)((sizeof(Layout) + 63) / 64); ++i) any |= *(cint4*)(p + 64 * i);
    asm volatile("" ::"s"(any));
}
#else
__device__ __forceinline__ void warm_kernel_args() {}
#endif

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

// Leaky ReLU with a slope >= 0: the derivative is a function of the activation's sign (h > 0 <=> p > 0), so the forward
// pass does not store it and the backward pass reads the activation plane instead -- a quarter of the row chains' global
// stores (a pack of models is bound by that traffic, not by arithmetic).
__device__ __forceinline__ bool der_from_act(const Layout& L) { return L.act == IVAE_ACT_LRELU && L.slope >= 0.f; }

// ------------------------------------------------------------------------------ streamed operand
// Source of the register-streamed operand Bm[j][k] (j: output column, k: contraction index) of one product.
//   SRC_W    forward:  Bm[j][k] = W[j][k] (W is (J, K) row-major at p), Bm[j][K] = bias[j] (at p + d2): the A tiles
//            carry a constant-1 column at k = K.  k in [hole0, hole0 + hole) is padding (the A tile is zero there) and
//            later k shift down by `hole` ([x | pad | u | 1] input tiles).
//   SRC_WT   dgrad:    Bm[j][k] = W[k][j]   (W is (K, J) row-major at p, row stride ld)
// Contraction indices k >= K multiply exact zeros (the last 16-wide k block is masked in registers); output columns
// j past the real extent are computed from whatever the range-checked loads return and dropped by the epilogues.
enum { SRC_NONE = 0, SRC_W, SRC_WT };
struct BSrc {                       // (kept small: it lives in scalar registers, two or three at a time)
    const float* p;
    unsigned lim;                   // floats readable from p (range of the buffer descriptor)
    int kind;
    int boff;                       // SRC_W: float offset of the bias from p
    int K;                          // real contraction extent
    int Jt, Kc;                     // extents of the product: output columns, contraction steps (bias column and hole included)
    int ld;
    int hole0, hole;
};

// ------------------------------------------------------------------------------ the contraction
// A wave owns the 16-column tile col0 = 16 wave (+ 16 NW per pass) of a product for all RTE row tiles of the workgroup
// and fetches its Bm fragments STRAIGHT from global memory / L2 into MFMA operand layout -- lane (c, q) holds
// Bm[col0 + c][16 blk + 4q .. + 3], one 128-bit load per 16 contraction steps for the k-contiguous kind, four 32-bit
// loads for the transposing one -- one block ahead of its use inside a tile.  Nothing of the streamed operand goes through
// LDS and no barrier separates the k blocks: a product costs ONE workgroup barrier (its A tile, written by the previous
// epilogue, must be complete).
// What was measured on MI355X and is NOT here any more: fragments prefetched across tile and product boundaries, in groups
// of 2 / 4 / 8 blocks.  Every deeper variant was slower (configs[3]: 59 us with this loop, 61 / 61 / 66 / 101 us with
// groups of 1 / 2 / 4 / 8 blocks carried from product to product; batch 32768: 242 against 252 ... 523 us): the weights
// come from L2 quickly, and the registers and loads of blocks that do not exist cost more than the latency they hid.
// Likewise ALL of a tile's fragments requested at once for products of <= 4 blocks (exact counts, one round trip per
// product): 256 packed models 134 us against 113 us per step -- the registers cost k_chain<2> its second workgroup per CU,
// and occupancy is what hides the latency of a pack's weights, which come from HBM rather than L2.
#define ONES4 ((f32x4){1.f, 1.f, 1.f, 1.f})
struct Stream {           // (diagnostic build: stamps written so far)
    int n;
};

// Issue the loads of block blk of tile col0 of S (+ the tile's bias entry, SRC_W).  KIND is a compile-time copy of S.kind
// and every call issues the same number of loads whatever the extents are (a block past them reads range-checked
// addresses and is never used): the compiler can then count outstanding loads exactly, and a wait for the block that is
// needed now does not also wait for the one issued after it.
template <int KIND, bool BIAS = true>
__device__ __forceinline__ void frag_load(f32x4& dst, float& bias, const BSrc& S, int col0, int blk) {
    int lane = threadIdx.x & 63;
#ifndef MF_TRANSPARENT
    asm volatile("" : "+v"(lane));      // opaque: keeps the per-lane addresses of every call site from being hoisted
#endif
    const int c = lane & 15, q = lane >> 4;
    const int j = col0 + c;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)S.p, 0, (int)(S.lim * 4u), 0x00020000);
    if (KIND == SRC_W) {
        const int kf = 16 * blk + 4 * q;
        const int sh = kf >= S.hole0 + S.hole ? S.hole : 0;
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (j * S.ld + kf - sh) * 4, 0, 0));
        if (BIAS) bias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (S.boff + j) * 4, 0, 0));
    } else {
        const int vo = ((16 * blk + 4 * q) * S.ld + j) * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x)
            dst[x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, x * S.ld * 4, 0));
    }
}
// a k block that reaches past the real contraction extent: those indices become exact zeros, the bias enters
template <int KIND>
__device__ __forceinline__ f32x4 frag_last(const BSrc& S, f32x4 b, int kf, float bias) {
    if (KIND == SRC_W) {
        const int sh = kf >= S.hole0 + S.hole ? S.hole : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = kf + i - sh;
            b[i] = cc < S.K ? b[i] : cc == S.K ? bias : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = kf + i < S.K ? b[i] : 0.f;
    }
    return b;
}

// out[i][j] = sum_k A[i][k] * Bm[j][k] for i < 16*RTE, j < S.Jt, k < S.Kc; A in LDS (row stride lda).  KIND = S.kind
// (compile-time copy).
// epi(rt, col0, acc, aux): lane (c, q) holds out[16*rt + 4q + r][col0 + c], r = 0..3; aux = pre(rt, col0): global operands
// of the epilogue (activation derivatives of the dgrad mask), fetched BEFORE the contraction so that they arrive under it.
// Workgroup barrier between the products of a row chain.  The hand-off is through LDS, so only this wave's LDS traffic
// has to be complete; global stores of the epilogues are read back (if at all) behind one of the kernels' full
// __syncthreads() at a stage boundary.
__device__ __forceinline__ void tile_barrier() {
#ifdef MF_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
struct NoPre { __device__ __forceinline__ f32x4 operator()(int, int) const { return ZERO4; } };
template <int RTE, int NW, int KIND, class EPI, class PRE = NoPre>
__device__ __forceinline__ void tile_gemm(const float* A, int lda, const BSrc& S, Stream& St, EPI&& epi, PRE&& pre = PRE()) {
    GEOM(NW);
    constexpr int NA = RTE == 1 ? 2 : 1;          // one row tile: two accumulators break the dependent MFMA chain
    int tid = threadIdx.x;
#ifndef MF_TRANSPARENT
    asm volatile("" : "+v"(tid));
#endif
    // (k_chain runs several chains as groups of NW waves in one workgroup: the tile map is relative to the group)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) % NW, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int nblk = up16(S.Kc) >> 4;
    const int kmask = KIND == SRC_W ? S.Kc - 1 : S.K;     // k blocks reaching past it are masked (SRC_W: only the last one)
    // The first fragment (weights: they do not depend on the barrier) and the epilogue operands are requested BEFORE the
    // barrier, which waits for this wave's LDS traffic only.
    int col0 = 16 * wave;
    f32x4 bq, aux[RTE];
    float bias = 0.f, nob;
    auto fetch = [&](int col) __attribute__((always_inline)) {
        frag_load<KIND>(bq, bias, S, col, 0);
#pragma unroll
        for (int rt = 0; rt < RTE; ++rt) aux[rt] = pre(rt, col);
    };
    fetch(col0);
    MSTAMPF(2);
    tile_barrier();                                // the A tile is complete
    MSTAMPF(3);
    while (col0 < S.Jt) {
        f32x4 acc[RTE][NA];
#pragma unroll
        for (int rt = 0; rt < RTE; ++rt)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[rt][a] = ZERO4;
        const float* ap = A + c * lda + 4 * q;
        for (int blk = 0; blk < nblk; ++blk) {
            f32x4 b = bq;
            frag_load<KIND, false>(bq, nob, S, col0, blk + 1);      // (the block past the last one: range-checked, unused)
            f32x4 a[RTE];
#pragma unroll
            for (int rt = 0; rt < RTE; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap + 16 * rt * lda + 16 * blk);
            if (16 * blk + 16 > kmask) b = frag_last<KIND>(S, b, 16 * blk + 4 * q, bias);
#pragma unroll
            for (int rt = 0; rt < RTE; ++rt)
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    acc[rt][x % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][x], b[x], acc[rt][x % NA], 0, 0, 0);
        }
        MSTAMPF(5);
#pragma unroll
        for (int rt = 0; rt < RTE; ++rt) epi(rt, col0, NA == 2 ? acc[rt][0] + acc[rt][NA - 1] : acc[rt][0], aux[rt]);
        MSTAMPF(6);
        col0 += NCH;
        if (col0 < S.Jt) fetch(col0);              // (wave-uniform)
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
    S.kind = SRC_W; S.p = P + D.lay[li].w_off; S.boff = D.lay[li].b_off - D.lay[li].w_off;
    S.Jt = D.lay[li].dout; S.K = D.lay[li].din; S.ld = D.lay[li].din; S.lim = (unsigned)(L.n_params - D.lay[li].w_off);
    if (holey) { S.hole0 = L.dd; S.hole = L.uoff - L.dd; }
    S.Kc = S.K + S.hole + 1;
    return S;
}

// One MLP forward.  `in` (LDS, stride ld_in) holds the layer-0 input with a constant-1 column after its Kin0
// columns; `holey`: the input tile is [x | pad | u | 1] and W's columns skip the pad.  head(rt, col0, acc) receives
// the output layer's tiles.
template <int RTE, int NW, class HEAD>
__device__ __forceinline__ void mlp_fwd(const Layout& L, const FwdCtx& X, const MlpDesc& D, const float* in, int ld_in, bool holey,
                                        Stream& St, HEAD&& head) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const float* cur = in;
    int ldc = ld_in;
    for (int li = 0; li < L.L; ++li) {
        const bool last = li == L.L - 1;
        const BSrc S = src_w(L, D, X.P, li, holey && li == 0);
        MSTAMP(7);
        const int N = D.lay[li].dout;
        if (last) {
            tile_gemm<RTE, NW, SRC_W>(cur, ldc, S, St, head);
        } else {
            float* dst = (li & 1) ? X.h1 : X.h0;
            float* actT = X.wsm + D.lay[li].pre_off + X.row0;
            float* derT = X.wsm + D.lay[li].der_off + X.row0;
            tile_gemm<RTE, NW, SRC_W>(cur, ldc, S, St, [&](int rt, int col0, const f32x4& acc, const f32x4&) {
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
                    if (!der_from_act(L)) *reinterpret_cast<f32x4*>(derT + (size_t)n * L.Bp + r0) = dv;
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
    float* xin = smem;                      // [x | pad | u | 1 | 0..]
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
    St.n = 0;
    MSTAMP(100);

    for (int i = 4 * tid; i < L.smem_enc / 4; i += 4 * NT) *reinterpret_cast<f32x4*>(xin + i) = ZERO4;
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
    MSTAMP(101);
    for (int r = tid; r < TR; r += NT) {
        xin[r * L.ldx + Kx] = 1.f;
        if (L.L > 1) { h0[r * L.ldh + L.H] = 1.f; h1[r * L.ldh + L.H] = 1.f; }
    }
    // (the first tile_gemm's barrier orders these writes before any operand read)
    FwdCtx X;
    X.P = P; X.wsm = wsm; X.row0 = row0; X.nrows = nrows; X.h0 = h0; X.h1 = h1;
    float* out = wsm + (m == IVAE_MLP_G ? L.ws.g : m == IVAE_MLP_LOGV ? L.ws.v : L.ws.l) + (size_t)row0 * L.dl;
    mlp_fwd<RTE, NW>(L, X, L.mlp[m], enc ? xin : xin + L.uoff, L.ldx, enc && L.M > 0, St,
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
    int np;                // parameters of a model (range of the weight loads)
    float* d0;             // delta tiles, row-major (TR, ldd)
    float* d1;
};

// Streamed operand of the input-gradient product of layer li: W^T.
__device__ __forceinline__ BSrc dgrad_src(const BwdCtx& X, const MlpDesc& D, int li) {
    BSrc S = src_none();
    S.kind = SRC_WT; S.p = X.P + D.lay[li].w_off; S.Jt = D.lay[li].din; S.K = D.lay[li].dout; S.ld = D.lay[li].din;
    S.lim = (unsigned)(X.np - D.lay[li].w_off);
    S.Kc = S.K;
    return S;
}
// Input-gradient chain of one MLP for a row tile.  On entry X.d0 holds dLoss/d(output) (rows >= nrows zero) and the
// caller has stored it feature-major at D.lay[L-1].dlt_off.  Every hidden layer's delta goes to the workspace feature-major
// (D.lay[li].dlt_off, (dout, Bp)): the weight gradients -- which nothing in this chain waits for -- are formed from those by
// k_wgrad, one workgroup per (row tile, layer), after the chains.  When dx_out != null the gradient with respect to the
// MLP's input is left there (LDS, stride ld_dx).
template <int RTE, int NW>
__device__ __forceinline__ void mlp_bwd(const Layout& L, const BwdCtx& X, const MlpDesc& D, float* dx_out, int ld_dx, Stream& St) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int n_prod = dx_out ? L.L : L.L - 1;
    int cur = 0;
    // activation derivatives of a product's output rows (the dgrad mask), feature-major in the workspace
    struct DerLoad {
        const float* p; int K, Bp, c, q;
        bool from_act; float slope;      // p holds the activations themselves: act' = 1 where h > 0, else the slope
        __device__ __forceinline__ f32x4 operator()(int rt, int col0) const {
            f32x4 d = *reinterpret_cast<const f32x4*>(p + (size_t)min(col0 + c, K - 1) * Bp + 16 * rt + 4 * q);
            if (from_act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = d[r] > 0.f ? 1.f : slope;
            }
            return d;
        }
    };
    auto der_of = [&](int li) {          // of the hidden layer that feeds layer li (li >= 1); li < 1: any valid source, unused
        const int l = max(li, 1);
        const bool fa = der_from_act(L);
        return DerLoad{X.wsm + (fa ? D.lay[l - 1].pre_off : D.lay[l - 1].der_off) + X.row0, D.lay[l].din, L.Bp, c, q, fa, L.slope};
    };
    for (int t = 0; t < n_prod; ++t) {
        const int li = L.L - 1 - t;
        const int K = D.lay[li].din;
        const BSrc S = dgrad_src(X, D, li);
        MSTAMP(7);
        // dX[r][k] = sum_n delta[r][n] * W[n][k]  (times act'(pre) for a hidden input)
        const float* dcur = cur ? X.d1 : X.d0;
        if (li > 0) {
            float* dnx = cur ? X.d0 : X.d1;
            float* dltT = X.wsm + D.lay[li - 1].dlt_off + X.row0;
            auto epi = [&](int rt, int col0, const f32x4& acc, const f32x4& pre) {
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
            };
            tile_gemm<RTE, NW, SRC_WT>(dcur, L.ldd, S, St, epi, der_of(li));
            cur ^= 1;
        } else {
            tile_gemm<RTE, NW, SRC_WT>(dcur, L.ldd, S, St, [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                const int k = col0 + c;
                if (k < K) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dx_out[(16 * rt + 4 * q + r) * ld_dx + k] = acc[r];
                }
            });
        }
    }
}

// Weight / bias gradients of every layer: dW[n][k] = sum_r delta[r][n] * in[r][k] over a tile of L.trw batch rows; column
// K of the result is the bias gradient (the input's constant-1 column).  No LDS, no barrier: a one-wave workgroup owns a
// block of WG_NT x WG_KT 16 x 16 tiles of one layer's gradient (16 accumulators) and fetches both operands straight from
// global memory in MFMA layout -- delta^T from the feature-major deltas the chains left in the workspace, the layer input
// from the feature-major activations (hidden layers: one 128-bit load per tile and 16 rows) or from the row-major
// x / u / z (first layers: four 32-bit loads) -- one 16-row block (64 MFMAs) ahead of its use, a fixed number of loads
// per block so that the waits are exact.  blockIdx.z enumerates (MLP, layer, k-tile group, n-tile group) in the MLP order
// f, g, logv, logl; the result goes to the row tile's gradient slab.
#ifndef WG_PD
#define WG_PD 1
#endif
#define WG_NT 4
#define WG_KT 4
struct WgradJob {
    const float* dlt;       // delta^T (N, Bp) at this tile's first row
    unsigned dlt_lim;
    const float* actT;      // hidden input, feature-major (K, Bp), at this tile's first row   (layer > 0)
    unsigned act_lim;
    const float* in_a;      // first-layer input rows: block a (na columns), block b (nb columns)
    const float* in_b;
    int na, nb;
    int N, K, Bp, nrows;
    float* slab;            // this row tile's gradient slab (parameter layout)
    int slab_bytes, w_off, b_off;
};
template <bool ROWS>
__device__ __forceinline__ void wgrad_block(const WgradJob& J, int kt0, int nt0) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int c = lane & 15, q = lane >> 4;
    const int nb = (J.nrows + 15) >> 4;        // 16-row blocks that hold rows of the batch
    const int ktv = min(WG_KT, (J.K + 16) / 16 - kt0);         // k tiles of this job that exist
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)J.dlt, 0, (int)(J.dlt_lim * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)J.actT, 0, (int)(J.act_lim * 4u), 0x00020000);
    const int last_row = J.nrows - 1;
    int jj[WG_KT], rstride[WG_KT];             // input feature of my result column in k tile kt (== K: the constant-1 column)
    const float* base[WG_KT];
#pragma unroll
    for (int kt = 0; kt < WG_KT; ++kt) {
        const int j = 16 * (kt0 + kt) + c;
        jj[kt] = j;
        const bool inb = j >= J.na;
        base[kt] = inb ? J.in_b + min(j - J.na, max(J.nb - 1, 0)) : J.in_a + j;       // (j > K: clamped, dropped)
        rstride[kt] = inb ? J.nb : J.na;
    }
    f32x4 acc[WG_NT][WG_KT], fa[WG_PD][WG_NT], fb[WG_PD][WG_KT];
#pragma unroll
    for (int t = 0; t < WG_NT; ++t)
#pragma unroll
        for (int kt = 0; kt < WG_KT; ++kt) acc[t][kt] = ZERO4;
    auto load_blk = [&](int u, int blk) {
        blk = min(blk, nb - 1);
        const int ro = 16 * blk + 4 * q;
#pragma unroll
        for (int t = 0; t < WG_NT; ++t)
            fa[u][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ((16 * (nt0 + t) + c) * J.Bp + ro) * 4, 0, 0));
#pragma unroll
        for (int kt = 0; kt < WG_KT; ++kt) {
            if (!ROWS) {
                fb[u][kt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (jj[kt] * J.Bp + ro) * 4, 0, 0));
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) fb[u][kt][x] = base[kt][__mul24(min(ro + x, last_row), rstride[kt])];   // (< 2^31 bytes: ivae_check_dims)
            }
        }
    };
    WSTAMP(1);
#pragma unroll
    for (int u = 0; u < WG_PD; ++u) load_blk(u, u);
    WSTAMP(2);
#ifdef MF_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(3);
#endif
    for (int b0 = 0; b0 < nb; b0 += WG_PD) {
#pragma unroll
        for (int u = 0; u < WG_PD; ++u) {
            const int blk = b0 + u;
            f32x4 a[WG_NT], b[WG_KT];
#pragma unroll
            for (int t = 0; t < WG_NT; ++t) a[t] = fa[u][t];
#pragma unroll
            for (int kt = 0; kt < WG_KT; ++kt) b[kt] = jj[kt] == J.K ? ONES4 : fb[u][kt];
            load_blk(u, blk + WG_PD);
            if (blk < nb) {
                if (16 * blk + 16 > J.nrows) {       // rows past the batch: both operands become exact zeros
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const bool ok = 16 * blk + 4 * q + x < J.nrows;
#pragma unroll
                        for (int kt = 0; kt < WG_KT; ++kt) b[kt][x] = ok ? b[kt][x] : 0.f;
#pragma unroll
                        for (int t = 0; t < WG_NT; ++t) a[t][x] = ok ? a[t][x] : 0.f;
                    }
                }
#pragma unroll
                for (int kt = 0; kt < WG_KT; ++kt)
                    if (kt < ktv) {
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int t = 0; t < WG_NT; ++t)
                                acc[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][x], b[kt][x], acc[t][kt], 0, 0, 0);
                    }
            }
        }
    }
    WSTAMP(4);
    // results -> the slab: branch-free, an element outside the layer's (N, K + 1) gets an offset past the slab's range
    // and the buffer store drops it.  Byte offsets advance by a per-lane stride (a weight column: K floats per output row;
    // the bias: one), so an element costs one 24-bit multiply-add, a compare against a scalar and a select.
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)J.slab, 0, J.slab_bytes, 0x00020000);
    const int n0 = 16 * nt0 + 4 * q;
#pragma unroll
    for (int kt = 0; kt < WG_KT; ++kt) {
        const int j = jj[kt];
        const int stride4 = j < J.K ? 4 * J.K : 4;
        const int el0 = (j < J.K ? 4 * (J.w_off + j) : 4 * J.b_off) + __mul24(n0, stride4);
        const int nlim = j <= J.K ? J.N - n0 : 0;             // rows n0 + e with e < nlim exist
#pragma unroll
        for (int t = 0; t < WG_NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 16 * t + r;
                const int off = e < nlim ? el0 + __mul24(e, stride4) : 0x7ffffff0;
                const float val = acc[t][kt][r];       // (bit_cast of a vector element reads element 0: go through a scalar)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(val), rg, off, 0, 0);
            }
    }
    WSTAMP(5);
}

// jobs of one layer: groups of WG_KT k tiles (din + 1 columns) x groups of WG_NT n tiles
__host__ __device__ __forceinline__ int wgrad_kgroups(int din) { return ((din + 16) / 16 + WG_KT - 1) / WG_KT; }
__host__ __device__ __forceinline__ int wgrad_ngroups(int dout) { return ((dout + 15) / 16 + WG_NT - 1) / WG_NT; }

#define WG_WAVES 4         // jobs (waves) per workgroup: the four waves of a workgroup sit on the four SIMDs of its CU
#define RED_W 16           // waves of a k_reduce block (its loss block's thread map is reproduced below)
__device__ __forceinline__ float loss_partial(const Layout& L, const IvaeCoef& C, const float* wsm, int t, int stride);
#ifdef WG_OCC
__attribute__((amdgpu_waves_per_eu(WG_OCC, WG_OCC)))
#endif
__global__ __launch_bounds__(64 * WG_WAVES) void k_wgrad(Layout L, const float* __restrict__ x, const float* __restrict__ u,
                                              float* __restrict__ ws, float* __restrict__ slabs, long slab_stride,
                                              IvaeCoef C, float* __restrict__ loss_out) {
    WSTAMP(0);
    warm_kernel_args();
    WSTAMP(8);
    const int model = blockIdx.y, row0 = blockIdx.x * L.trw, nrows = min(L.trw, L.B - row0);
    // blockIdx.z -> (MLP, layer, k group, n group)
    int m = -1, li = 0, job = 0;
    {   // (static loop bounds over one 64-byte table: two wide scalar loads, not a dependent load per layer)
        int z = blockIdx.z * WG_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int mm = mi == 0 ? IVAE_MLP_F : mi == 1 ? IVAE_MLP_G : mi == 2 ? IVAE_MLP_LOGV : IVAE_MLP_LOGL;
#pragma unroll
            for (int l = 0; l < IVAE_MAX_LAYERS; ++l) {
                const int nj = L.wg_jobs[mi][l];
                const bool hit = z >= 0 && z < nj;          // (selects, not branches: z only decreases, one interval holds it)
                m = hit ? mm : m; li = hit ? l : li; job = hit ? z : job;
                z -= nj;
            }
        }
    }
    WSTAMP(6);
    if (m >= 0) {
        const MlpDesc& D = L.mlp[m];
        const int kg = wgrad_kgroups(D.lay[li].din);
        const int kt0 = (job % kg) * WG_KT, nt0 = (job / kg) * WG_NT;
        float* wsm = ws + (size_t)model * L.ws.per_model;
        WgradJob J;
        J.N = D.lay[li].dout; J.K = D.lay[li].din; J.Bp = L.Bp; J.nrows = nrows;
        J.dlt = wsm + D.lay[li].dlt_off + row0;
        { const long lim = L.ws.per_model - (D.lay[li].dlt_off + row0); J.dlt_lim = (unsigned)(lim < 0x3fffffffL ? lim : 0x3fffffffL); }
        J.slab = slabs + (size_t)model * slab_stride + (size_t)blockIdx.x * L.slab_ld;
        J.slab_bytes = L.n_params * 4; J.w_off = D.lay[li].w_off; J.b_off = D.lay[li].b_off;
        J.actT = wsm; J.act_lim = 0; J.in_a = J.in_b = wsm; J.na = J.K; J.nb = 0;
#ifdef MF_PROF
        asm volatile("" ::"s"(J.N), "s"(J.K), "s"(J.dlt), "s"(J.w_off), "s"(J.b_off), "s"(J.dlt_lim));
        WSTAMP(7);
#endif
        if (li > 0) {
            J.actT = wsm + D.lay[li - 1].pre_off + row0;
            const long lim = L.ws.per_model - (D.lay[li - 1].pre_off + row0);
            J.act_lim = (unsigned)(lim < 0x3fffffffL ? lim : 0x3fffffffL);
            wgrad_block<false>(J, kt0, nt0);
        } else {
            const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
            const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : xm;
            if (m == IVAE_MLP_F) { J.in_a = wsm + L.ws.z + (size_t)row0 * L.dl; J.na = L.dl; J.in_b = J.in_a; J.nb = 0; }
            else if (m == IVAE_MLP_LOGL) { J.in_a = um; J.na = L.M; J.in_b = um; J.nb = 0; }
            else { J.in_a = xm; J.na = L.dd; J.in_b = um; J.nb = L.M; }
            wgrad_block<true>(J, kt0, nt0);
        }
    }
    // loss of the model (loss_out != null: the gradient needs no slab reduction and the row chains ran in k_chain, which
    // sees one tile's rows only -- no k_reduce launch follows): the model's first workgroup sums the per-row terms with
    // k_reduce's thread map and order (same bits), 64 * RED_W virtual threads on its 256 real ones
    if (loss_out && blockIdx.x == 0 && blockIdx.z == 0) {
        __shared__ float red[64 * RED_W];
        const float* wsl = ws + (size_t)model * L.ws.per_model;
        const int tid = threadIdx.x;
        for (int vt = tid; vt < 64 * RED_W; vt += 64 * WG_WAVES) red[vt] = loss_partial(L, C, wsl, vt, 64 * RED_W);
        __syncthreads();
        for (int sft = 32 * RED_W; sft > 0; sft >>= 1) {
            for (int vt = tid; vt < sft; vt += 64 * WG_WAVES) red[vt] += red[vt + sft];
            __syncthreads();
        }
        if (tid == 0) loss_out[model] = -red[0] / (float)C.global_batch;
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
    float* t0 = smem;                       // hidden tiles (forward) / delta tiles (backward)
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
    B.row0 = row0; B.nrows = nrows; B.TR = TR; B.np = L.n_params; B.d0 = t0; B.d1 = t1;

    Stream St;
    St.n = 0;
    MSTAMP(199);
    for (int i = 4 * tid; i < L.smem_dec / 4; i += 4 * NT) *reinterpret_cast<f32x4*>(t0 + i) = ZERO4;
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
        auto head = [&](int rt, int col0, const f32x4& acc, const f32x4&) {
            const int n = col0 + c;
            if (n < L.dd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + 4 * q + r;
                    if (row < nrows) fm[(size_t)row * L.dd + n] = acc[r];
                }
            }
        };
        mlp_fwd<RTE, NW>(L, X, DF, sz, lds, false, St, head);
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
        float* dltF = wsm + DF.lay[L.L - 1].dlt_off + row0;
        __syncthreads();
        for (int idx = tid; idx < TR * L.dd; idx += NT) {
            const int r = idx / L.dd, k = idx - r * L.dd;
            const float v = r < nrows ? -C.a * w * (xm[(size_t)r * L.dd + k] - fm[(size_t)r * L.dd + k]) / C.decoder_var : 0.f;
            t0[r * L.ldd + k] = v;
            dltF[(size_t)k * L.Bp + r] = v;
        }
        mlp_bwd<RTE, NW>(L, B, DF, sdz, lds, St);
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
    float* dA = smem;
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
    X.row0 = row0; X.nrows = nrows; X.TR = TR; X.np = L.n_params; X.d0 = dA; X.d1 = dB;
    float* dltT = wsm + L.mlp[m].lay[L.L - 1].dlt_off + row0;

    Stream St;
    St.n = 0;
    St.n = 1000;                                  // (diagnostic stamps of this kernel go to slots 1000..)
    for (int i = 4 * tid; i < L.smem_benc / 4; i += 4 * NT) *reinterpret_cast<f32x4*>(dA + i) = ZERO4;
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
    mlp_bwd<RTE, NW>(L, X, L.mlp[m], nullptr, 0, St);
    MSTAMP(202);
    // loss of the model (loss_out != null: no slab reduction follows, so k_reduce is not launched): one workgroup sums
    // the per-row terms with k_reduce's thread map and order (same bits)
    if (loss_out && blockIdx.x == 0 && blockIdx.z == 0) {
        __syncthreads();
        float* red = dA;                    // (the delta tiles are dead; 2 TR ldd + 64 >= NR floats)
        if (tid < NR) red[tid] = loss_partial(L, C, wsm, tid, NR);
        __syncthreads();
        for (int s = NR / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) loss_out[model] = -red[0] / (float)C.global_batch;
    }
}

// Fused row chain: everything of a step that depends only on a tile's own rows, in ONE launch -- encoder / prior forward
// (the g, logv, logl chains run concurrently as three groups of 4 waves, in lockstep through the same barriers),
// reparameterisation, decoder forward, per-row log-densities, [mode bit 1] the decoder's input-gradient chain and
// [mode bit 2, plain ELBO only: the pairwise terms couple the rows] the head deltas and the encoder / prior
// input-gradient chains.  g, v, l, z, f and dLoss/dz stay in LDS between the stages; what other kernels read (k_wgrad:
// activations and deltas; pairwise kernels and the loss: row tensors and per-row terms) goes to the workspace as before.
// Replaces k_fwd_enc + k_dec (+ k_bwd_enc) where its tiles fit the LDS (Layout.smem_chain != 0): two launch boundaries
// and every global round trip between the stages disappear from the dependent chain.
#ifndef CH_NW
#define CH_NW 4
#endif
#ifndef CH_OCC
#define CH_OCC 6          // waves per SIMD the register allocation leaves room for: two 12-wave workgroups per CU
#endif
template <int RTE>
__global__ __launch_bounds__(64 * CH_NW * 3) __attribute__((amdgpu_waves_per_eu(CH_OCC, CH_OCC))) void k_chain(Layout L, IvaeCoef C, int mode, const float* __restrict__ params,
                                                          const float* __restrict__ x, const float* __restrict__ u,
                                                          const float* __restrict__ eps, float* __restrict__ z_out,
                                                          float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = CH_NW;
    const int TR = L.tr, lds = L.lds, ldf = L.dd | 1;
    const int ldt = max(L.ldh, L.ldd);
    const int NT = blockDim.x;              // 64 * 4 * (number of encoder / prior chains)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6) / NW;          // 0: g (+ decoder), 1: logv, 2: logl
    const int ngrp = L.M > 0 ? 3 : 2;
    float* xin = smem;                      // [x | pad | u | 1 | 0..]
    float* tiles = xin + TR * L.ldx;        // per group two (TR, ldt) tiles: hidden activations, then deltas
    float* t0 = tiles + grp * 2 * TR * ldt;
    float* t1 = t0 + TR * ldt;
    float* sg = tiles + ngrp * 2 * TR * ldt;        // row tensors (TR, lds): g, v, l, z (+ constant-1 column), dLoss/dz, eps
    float* sv = sg + TR * lds;
    float* sl = sv + TR * lds;
    float* sz = sl + TR * lds;
    float* sdz = sz + TR * lds;
    float* se = sdz + TR * lds;
    float* sf = se + TR * lds;              // decoder output (TR, ldf)
    const int model = blockIdx.y, row0 = blockIdx.x * TR, nrows = min(TR, L.B - row0);
    const int m = grp == 0 ? IVAE_MLP_G : grp == 1 ? IVAE_MLP_LOGV : IVAE_MLP_LOGL;
    const float* P = params + (size_t)model * L.n_params;
    float* wsm = ws + (size_t)model * L.ws.per_model;
    const float* xm = x + ((size_t)model * L.B + row0) * L.dd;
    const float* um = L.M > 0 ? u + ((size_t)model * L.B + row0) * L.M : nullptr;
    const float* em = eps + ((size_t)model * L.B + row0) * L.dl;
    const int Kx = L.uoff + L.M;
    const bool enc = m != IVAE_MLP_LOGL;
    const MlpDesc& DF = L.mlp[IVAE_MLP_F];

    Stream St;
    St.n = 0;
    MSTAMP(299);
    for (int i = 4 * tid; i < L.smem_chain / 4; i += 4 * NT) *reinterpret_cast<f32x4*>(smem + i) = ZERO4;
    __syncthreads();
    {   // input rows and noise -> LDS: a wave takes two rows per pass, lanes over the columns: eight loads (+ two of the
        // noise) in flight, one memory round trip for a 16- or 24-row tile
        const int wv = tid >> 6, nwv = NT >> 6, W = L.dd + L.M;
        for (int r = wv; r < nrows; r += 2 * nwv) {
            const int r2 = r + nwv;
            const bool two = r2 < nrows;
            const int rb = two ? r2 : r;
            const float* xr[2] = {xm + (size_t)r * L.dd, xm + (size_t)rb * L.dd};
            const float* ur[2] = {um + (size_t)r * L.M - L.dd, um + (size_t)rb * L.M - L.dd};   // indexed by k >= dd only (um is null when M == 0)
            float* dst[2] = {xin + r * L.ldx, xin + rb * L.ldx};
            float e0 = 0.f, e1 = 0.f;
            if (lane < L.dl) { e0 = em[(size_t)r * L.dl + lane]; e1 = em[(size_t)rb * L.dl + lane]; }
            for (int k0 = lane; k0 < W; k0 += 256) {
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + 64 * e;
                        v[h][e] = k < L.dd ? xr[h][k] : k < W ? ur[h][k] : 0.f;
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + 64 * e;
                        if (k < W && (h == 0 || two)) dst[h][k < L.dd ? k : L.uoff + (k - L.dd)] = v[h][e];
                    }
            }
            if (lane < L.dl) { se[r * lds + lane] = e0; if (two) se[r2 * lds + lane] = e1; }
        }
    }
    for (int r = tid; r < TR; r += NT) {
        xin[r * L.ldx + Kx] = 1.f;
        sz[r * lds + L.dl] = 1.f;
        if (L.M == 0) for (int k = 0; k < L.dl; ++k) sl[r * lds + k] = 1.f;
    }
    if (L.L > 1)
        for (int r = tid & (64 * NW - 1); r < TR; r += 64 * NW) { t0[r * L.ldh + L.H] = 1.f; t1[r * L.ldh + L.H] = 1.f; }
    // (the first product's barrier orders these writes before any operand read)
    MSTAMP(300);

    // ---- encoder / prior forward: three chains in lockstep -> g, v, l rows (LDS + workspace)
    {
        FwdCtx X;
        X.P = P; X.wsm = wsm; X.row0 = row0; X.nrows = nrows; X.h0 = t0; X.h1 = t1;
        float* srow = grp == 0 ? sg : grp == 1 ? sv : sl;
        float* out = wsm + (grp == 0 ? L.ws.g : grp == 1 ? L.ws.v : L.ws.l) + (size_t)row0 * L.dl;
        mlp_fwd<RTE, NW>(L, X, L.mlp[m], enc ? xin : xin + L.uoff, L.ldx, enc && L.M > 0, St,
                     [&](int rt, int col0, const f32x4& acc, const f32x4&) {
                         const int n = col0 + c;
                         if (n < L.dl) {
#pragma unroll
                             for (int r = 0; r < 4; ++r) {
                                 const int row = 16 * rt + 4 * q + r;
                                 const float o = grp == 0 ? acc[r] : expf(acc[r]);
                                 srow[row * lds + n] = o;
                                 if (row < nrows) out[(size_t)row * L.dl + n] = o;
                             }
                         }
                     });
    }
    __syncthreads();
    MSTAMP(301);
    // ---- reparameterisation z = eps*sqrt(v) + g      (models/nets.py:301-304, 145-149)
    for (int idx = tid; idx < TR * L.dl; idx += NT) {
        const int r = idx / L.dl, k = idx - r * L.dl;
        float zz = 0.f;
        if (r < nrows) {
            const size_t o = (size_t)(row0 + r) * L.dl + k;
            zz = se[r * lds + k] * sqrtf(sv[r * lds + k]) + sg[r * lds + k];
            z_out[(size_t)model * L.B * L.dl + o] = zz;
            wsm[L.ws.z + o] = zz;
            if (L.M == 0) wsm[L.ws.l + o] = 1.f;
        }
        sz[r * lds + k] = zz;
    }
    // ---- decoder forward (group 0; the other groups keep step through the barriers) -> f (LDS + workspace)
    BwdCtx B;
    B.P = P; B.wsm = wsm; B.slab = nullptr;
    B.row0 = row0; B.nrows = nrows; B.TR = TR; B.np = L.n_params; B.d0 = t0; B.d1 = t1;
    float* fm = wsm + L.ws.f + (size_t)row0 * L.dd;
    MSTAMP(302);
    // The decoder of a two-row-tile workgroup runs as TWO groups, one row tile each (the g and the logv group, on their own
    // tiles): the groups that only kept step through the decoder's barriers were a third of this kernel's wave cycles.
    constexpr int DRT = RTE == 2 ? 1 : RTE;            // row tiles a decoder group works on
    const int ndg = RTE == 2 ? 2 : 1;                  // decoder groups
    const int ro = RTE == 2 ? 16 * grp : 0;            // first tile row of my decoder group
    const int nrd = RTE == 2 ? max(0, min(16, nrows - ro)) : nrows;
    if (grp < ndg) {
        // (the hidden tiles still hold the encoder chain's activations: every column a product reads is rewritten by the
        // previous one, the constant-1 column stays)
        FwdCtx X;
        X.P = P; X.wsm = wsm; X.row0 = row0 + ro; X.nrows = nrd; X.h0 = t0; X.h1 = t1;
        auto head = [&](int rt, int col0, const f32x4& acc, const f32x4&) {
            const int n = col0 + c;
            if (n < L.dd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ro + 16 * rt + 4 * q + r;
                    sf[row * ldf + n] = acc[r];
                    if (row < nrows) fm[(size_t)row * L.dd + n] = acc[r];
                }
            }
        };
        mlp_fwd<DRT, NW>(L, X, DF, sz + ro * lds, lds, false, St, head);
    } else {
        for (int li = 0; li < L.L; ++li) __syncthreads();
    }
    __syncthreads();
    MSTAMP(303);
    // ---- per-row Gaussian log-densities   (models/nets.py:306-314, 45-52): sixteen lanes per row take every sixteenth
    // feature, then four xor-shuffles add the partial sums (fixed order)
    {
        const int r = tid >> 4, part = tid & 15;
        const bool live = r < nrows;                // (tid < 16 TR <= NT: whole waves take the same branch below)
        const int rr = live ? r : 0;
        const float dv = C.decoder_var;
        const float logdv = logf(dv);
        float lpx = 0.f, lqc = 0.f, lp = 0.f;
        if (tid < 16 * TR) {
            for (int k = part; k < L.dd; k += 16) {
                const float df = xin[rr * L.ldx + k] - sf[rr * ldf + k];
                lpx += -0.5f * (LOG2PI + logdv + df * df / dv);
            }
            for (int k = part; k < L.dl; k += 16) {
                const float zz = sz[rr * lds + k], gg = sg[rr * lds + k], vv = sv[rr * lds + k], ll = sl[rr * lds + k];
                const float dq = zz - gg;
                lqc += -0.5f * (LOG2PI + logf(vv) + dq * dq / vv);
                lp += -0.5f * (LOG2PI + logf(ll) + zz * zz / ll);
            }
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                lpx += __shfl_xor(lpx, sft); lqc += __shfl_xor(lqc, sft); lp += __shfl_xor(lp, sft);
            }
            if (live && part == 0) {
                wsm[L.ws.lpx + row0 + r] = lpx;
                wsm[L.ws.lqc + row0 + r] = lqc;
                wsm[L.ws.lp + row0 + r] = lp;
            }
        }
    }
    MSTAMP(304);
    if (!(mode & 2)) return;
    // ---- decoder input-gradient chain (group 0): layer deltas of f -> workspace, dLoss/dz -> sdz
    const float w = 1.f / (float)C.global_batch;
    {
        // dLoss/df = -a*w*(x-f)/var, all TR rows (zero beyond the batch), over the first tile of the decoder group that
        // owns the row (dead activations)
        float* dltF = wsm + DF.lay[L.L - 1].dlt_off + row0;
        for (int idx = tid; idx < TR * L.dd; idx += NT) {
            const int r = idx / L.dd, k = idx - r * L.dd;
            const float v = r < nrows ? -C.a * w * (xin[r * L.ldx + k] - sf[r * ldf + k]) / C.decoder_var : 0.f;
            const int dg = RTE == 2 ? r >> 4 : 0;
            tiles[dg * 2 * TR * ldt + (r - (RTE == 2 ? 16 * dg : 0)) * L.ldd + k] = v;
            dltF[(size_t)k * L.Bp + r] = v;
        }
    }
    MSTAMP(305);
    if (grp < ndg) {
        BwdCtx Bd = B;
        Bd.row0 = row0 + ro; Bd.nrows = nrd;
        mlp_bwd<DRT, NW>(L, Bd, DF, sdz + ro * lds, lds, St);
    } else {
        for (int li = 0; li < L.L; ++li) __syncthreads();
    }
    __syncthreads();
    if (!(mode & 4)) {
        for (int idx = tid; idx < nrows * L.dl; idx += NT) {
            const int r = idx / L.dl, k = idx - r * L.dl;
            wsm[L.ws_dz + (size_t)(row0 + r) * L.dl + k] = sdz[r * lds + k];
        }
        return;
    }
    // ---- head deltas (plain ELBO: SURVEY.md §8 a-bwd) and the three encoder / prior input-gradient chains
    MSTAMP(306);
    {
        float* dltT = wsm + L.mlp[m].lay[L.L - 1].dlt_off + row0;
        const int gt = tid & (64 * NW - 1);          // thread index inside my group
        for (int idx = gt; idx < TR * L.dl; idx += 64 * NW) {
            const int r = idx / L.dl, k = idx - r * L.dl;
            float dlt = 0.f;
            if (r < nrows) {
                const float zz = sz[r * lds + k], gg = sg[r * lds + k], vv = sv[r * lds + k], ll = sl[r * lds + k];
                const float dq = zz - gg;
                const float rr = dq / vv;
                const float Gz = sdz[r * lds + k] - w * (C.b * rr - C.d * zz / ll);
                if (grp == 0) dlt = w * C.b * rr + Gz;
                else if (grp == 1) dlt = -w * C.b * 0.5f * (1.f - dq * dq / vv) + Gz * 0.5f * se[r * lds + k] * sqrtf(vv);
                else dlt = w * C.d * 0.5f * (1.f - zz * zz / ll);
            }
            t0[r * L.ldd + k] = dlt;
            if (r < nrows) dltT[(size_t)k * L.Bp + r] = dlt;
        }
    }
    MSTAMP(307);
    mlp_bwd<RTE, NW>(L, B, L.mlp[m], nullptr, 0, St);
    MSTAMP(308);
}

// Sum the per-tile slabs into grads and reduce the loss.  A block sums 64 parameters: wave g adds the tiles
// t = g, g+4, ... with four independent accumulators (loads in flight), then the partials are combined in a fixed
// order, so the result does not depend on scheduling.
#define RED_P 256          // (RED_W = 16 waves per block, defined above: wave g sums the slabs g, g + 16, ...)
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
    static SmemCache cur_enc = {}, cur_dec = {}, cur_benc = {};
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
        hipLaunchKernelGGL(k_wgrad, dim3(L.n_wtiles, L.n_models, (L.wg_total + WG_WAVES - 1) / WG_WAVES), dim3(64 * WG_WAVES), 0, st, L, x, u, ws, slabs, slab_stride,
                           C, z_out /* phase 5: the loss output, or null */);
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

template <int RTE>
static hipError_t launch_chain_t(const Layout& L, const IvaeCoef& C, int mode, const float* params, const float* x, const float* u,
                                 const float* eps, float* z_out, float* ws, hipStream_t st) {
    static SmemCache cur = {};
    hipError_t e;
    if ((e = set_smem((const void*)k_chain<RTE>, L.smem_chain, cur)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_chain<RTE>), dim3(L.n_tiles, L.n_models), dim3(64 * CH_NW * (L.M > 0 ? 3 : 2)), L.smem_chain, st, L, C, mode,
                       params, x, u, eps, z_out, ws);
    return hipGetLastError();
}

// grads_for_dec_bwd != null: also run the decoder's backward inside the forward launch (the caller then passes
// dec_bwd_done = true to launch_backward); `full` (plain ELBO only): the encoder / prior input-gradient chains too, when
// the fused row-chain kernel serves this layout.  *done (bits: 1 forward, 2 decoder backward, 4 encoder backward) says
// what ran.
hipError_t launch_forward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                          const float* eps, float* z_out, float* ws, float* grads_for_dec_bwd, hipStream_t st, bool full,
                          int* done) {
    if (L.smem_chain) {
        const int mode = 1 | (grads_for_dec_bwd ? 2 : 0) | (grads_for_dec_bwd && full && !C.pairwise ? 4 : 0);
        if (done) *done = mode;
        if (L.tr == 16) return launch_chain_t<1>(L, C, mode, params, x, u, eps, z_out, ws, st);
        return launch_chain_t<2>(L, C, mode, params, x, u, eps, z_out, ws, st);       // (64-row tiles: smem_chain == 0)
    }
    if (done) *done = grads_for_dec_bwd ? 3 : 1;
    hipError_t e = launch_phase(L, C, 0, params, x, u, eps, z_out, ws, nullptr, st);
    if (e != hipSuccess) return e;
    return launch_phase(L, C, grads_for_dec_bwd ? 3 : 1, params, x, u, eps, z_out, ws, grads_for_dec_bwd, st);
}

hipError_t launch_backward(const Layout& L, const IvaeCoef& C, const float* params, const float* x, const float* u,
                           const float* eps, float* loss_out, float* grads, float* ws, bool dec_bwd_done, hipStream_t st,
                           const AdamFuse* fuse, bool* fused, bool enc_bwd_done) {
    if (fused) *fused = false;
    hipError_t e;
    if (!dec_bwd_done && (e = launch_phase(L, C, 2, params, x, u, eps, nullptr, ws, grads, st)) != hipSuccess) return e;
    const int nblk_p = L.n_wtiles > 1 ? (L.n_params + RED_P - 1) / RED_P : 0;
    // no slabs to sum: k_bwd_enc also reduces the loss and k_reduce is not launched
    if (!enc_bwd_done &&
        (e = launch_phase(L, C, 4, params, x, u, eps, nblk_p == 0 ? loss_out : nullptr, ws, grads, st)) != hipSuccess)
        return e;
    // no slabs and the row chains ran in k_chain (which sees one tile's rows only): k_wgrad's first workgroup of a model
    // also reduces its loss
    if ((e = launch_phase(L, C, 5, params, x, u, eps, nblk_p == 0 && enc_bwd_done ? loss_out : nullptr, ws, grads, st)) != hipSuccess)
        return e;
    if (nblk_p == 0) return hipSuccess;
    if (fuse && L.n_models <= ADAM_MODELS_PER_LAUNCH) {
        hipLaunchKernelGGL(k_reduce<true>, dim3(nblk_p + 1, L.n_models), dim3(64 * RED_W), 0, st, L, C, ws, grads, loss_out, *fuse);
        if (fused) *fused = true;
    } else {
        hipLaunchKernelGGL(k_reduce<false>, dim3(nblk_p + 1, L.n_models), dim3(64 * RED_W), 0, st, L, C, ws, grads, loss_out, AdamFuse{});
    }
    return hipGetLastError();
}
