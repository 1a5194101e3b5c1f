// Fused persistent train-step kernel: one 512-thread workgroup (8 waves, two per SIMD) trains one
// model for n_steps consecutive SGD steps without leaving the CU.
//
//   per step:  gather batch rows by index -> encoder/prior/decoder forward (fp32 MFMA) ->
//              reparameterisation + Gaussian log-densities + loss -> analytic backward
//              (dgrad + wgrad on MFMA) -> Adam on every layer             (plain ELBO)
//
// Reference sequence replaced (per step): runners/ivae.py:75-91 / models/wrappers.py:64-70 with
// models/nets.py:450-475 (iVAE) or 587-599 (VAE) and torch.optim.Adam.
//
// On-chip residency:
//   * parameters: LDS, one (N, KS) matrix per nn.Linear with the bias folded in as column K (the
//     activation tiles carry a constant-1 column there), rows padded to KS = 4*odd floats so that
//     64-bit operand reads are bank-conflict free;
//   * Adam exp_avg / exp_avg_sq: registers, flat per-layer ownership (thread t owns elements
//     t, t+512, ... of each layer), loaded once per launch, written back once;
//   * activations: five (64, ST) LDS tiles + four (64, SS) row tensors; hidden activations the
//     backward pass needs later are stashed in the producing wave's registers (MFMA C layout) and
//     written back to LDS one phase before their MLP's backward needs them;
//   * the weight gradient of a layer is staged in a dead tile and consumed by Adam one phase later
//     (while the next layer's products run on the matrix pipe): no gradient arena exists.
// Every layer product (fwd, dgrad, wgrad) is v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  A wave
// owns a PAIR of 16x16 output tiles whose columns interleave (tile j holds columns col0+2c+j), so
// operands and results move as 64-bit LDS accesses; the contraction index is permuted the same
// way on both operands (lane q takes k = 8t+2q, 8t+2q+1), which leaves every sum unchanged up to
// fp32 summation order.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"
#include "fused_train.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (make diag: -DIVAE_DIAG -DFT_PROF, a library of another name): s_memtime stamps at phase
// boundaries of model 0 / waves 0 and 4 go to a buffer of their own; no output depends on them and the shipped
// library contains none.
#if defined(FT_PROF) && !defined(IVAE_DIAG)
#error "FT_PROF belongs to the diagnostic build (make diag, -DIVAE_DIAG): never into libivae_amd.so"
#endif
#ifdef FT_PROF
__device__ unsigned long long* g_ft_prof = nullptr;
#define FT_NSTAMP 128
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_ft_prof) {                                    \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_ft_prof[(size_t)s * FT_NSTAMP + (i)] = t_;                                           \
        }                                                                                          \
    } while (0)
// sub-phase stamps of wave 0 (slots 32+i) and wave 4 (slots 48+i), i < 16
#define STAMPW(i)                                                                                  \
    do {                                                                                           \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && g_ft_prof) {            \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_ft_prof[(size_t)s * FT_NSTAMP + 32 + (threadIdx.x >> 4) + (i)] = t_;                 \
        }                                                                                          \
    } while (0)
// backward-phase stamps of wave 0 (slots 64+i) and wave 4 (slots 96+i), i < 32
#define STAMPB(i)                                                                                  \
    do {                                                                                           \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && g_ft_prof) {            \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_ft_prof[(size_t)s * FT_NSTAMP + 64 + (threadIdx.x >> 3) + (i)] = t_;                 \
        }                                                                                          \
    } while (0)
// in-stream stamps (no wait at the stamp: the counter is read into scalar registers and stored by TSTORE at the end)
#define TDECL unsigned long long tq_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TMARK(i) asm volatile("s_memtime %0" : "=s"(tq_[i])::"memory")
#define TSTORE(step, base)                                                                         \
    do {                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && g_ft_prof)              \
            for (int i_ = 0; i_ < 8; ++i_) g_ft_prof[(size_t)(step) * FT_NSTAMP + 64 + (threadIdx.x >> 3) + (base) + i_] = tq_[i_]; \
    } while (0)
#else
#define STAMP(i)
#define STAMPW(i)
#define STAMPB(i)
#define TDECL
#define TMARK(i)
#define TSTORE(step, base)
#endif

#define FT_THREADS 512
#define FT_B 64
#define LOG2PI_F 1.8378770664093453f

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }
// smallest 4*odd >= x: a row stride s with s/2 = 2*odd puts the 32 lanes of one ds_read_b64 group
// (16 rows x 2 column pairs) on 32 distinct bank pairs
__host__ __device__ constexpr int odd4(int x) { return (((x + 3) >> 2) | 1) << 2; }

// ---------------------------------------------------------------------------------- config
// Layer ids: 0..2 logl (absent when M == 0), 3..5 f, 6..8 g, 9..11 logv  (torch parameters() order).
enum { L_L0 = 0, L_L1, L_L2, L_F0, L_F1, L_F2, L_G0, L_G1, L_G2, L_V0, L_V1, L_V2, N_LAYERS };

template <int DD_, int DL_, int M_, int H_>
struct FCfg {
    static constexpr int DD = DD_, DL = DL_, M = M_, H = H_;
    static constexpr bool HAS_U = M_ > 0;
    static constexpr int UOFF = HAS_U ? ((DD_ + 1) & ~1) : 0;       // first column of u in the [x | u] tile (even)
    static constexpr int KIN = HAS_U ? UOFF + M_ : DD_;             // its constant-1 (bias) column
    static constexpr int KS_E0 = odd4(KIN + 1), KS_H = odd4(H_ + 1), KS_L0 = odd4(M_ + 1), KS_F0 = odd4(DL_ + 1);
    static constexpr int ST = cmax(KS_E0, KS_H);                    // activation tile row stride
    static constexpr int SS = odd4(cmax(DL_, DD_) + 1);             // row-tensor stride
    // per layer: output rows N, LDS row stride KS, bias column KB, torch in-features KR
    static constexpr int n_of(int l) { return (l < 3 && !HAS_U) ? 0 : (l == L_L2 || l == L_G2 || l == L_V2) ? DL : (l == L_F2 ? DD : H); }
    static constexpr int ks_of(int l) { return l == L_L0 ? KS_L0 : l == L_F0 ? KS_F0 : (l == L_G0 || l == L_V0) ? KS_E0 : KS_H; }
    static constexpr int kb_of(int l) { return l == L_L0 ? M : l == L_F0 ? DL : (l == L_G0 || l == L_V0) ? KIN : H; }
    static constexpr int kr_of(int l) { return l == L_L0 ? M : l == L_F0 ? DL : (l == L_G0 || l == L_V0) ? DD + M : H; }
    static constexpr int off_of(int l) { int o = 0; for (int i = 0; i < l; ++i) o += n_of(i) * ks_of(i); return o; }      // LDS
    static constexpr int goff_of(int l) { int o = 0; for (int i = 0; i < l; ++i) o += n_of(i) * (kr_of(i) + 1); return o; } // torch arena
    static constexpr int NPARAM = goff_of(N_LAYERS);
    static constexpr int ARENA = (off_of(N_LAYERS) + 3) & ~3;
    // Adam register slots: thread t owns elements t, t+512, ... of the (N, KB+1) index space of a layer
    static constexpr int p_of(int l) { return n_of(l) * (kb_of(l) + 1); }
    static constexpr int cnt_of(int l) { return cdiv(p_of(l), FT_THREADS); }
    static constexpr int reg_of(int l) { int o = 0; for (int i = 0; i < l; ++i) o += cnt_of(i); return o; }
    static constexpr int NADAM = reg_of(N_LAYERS);
    static constexpr int TILE = FT_B * ST, SMALL = FT_B * SS;
    static constexpr int STG_F2 = H * KS_H;                         // offset of the decoder-head gradient stage inside its tile
    static constexpr int LDS_FLOATS = ARENA + 5 * TILE + 5 * SMALL;         // d = 10: exactly the 160 KB
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    static constexpr int NXD = cdiv(DD_, 8), NXM = cdiv(M_, 8);     // prefetch registers: thread = (row, every 8th column)
    static constexpr int NXU = NXD + NXM;
    static_assert(ST <= 64 && H_ <= 62, "tile-pair plan covers at most 64 columns");
    static_assert(DD_ <= 15 && DL_ <= 15, "heads are one 16-wide tile");
    static_assert(STG_F2 + DD_ * KS_H <= TILE && SMALL + H_ * KS_F0 + 16 <= TILE, "stage aliasing plan");
    static_assert(DL_ * KS_H <= SMALL, "head-gradient stage aliases the z row tensor");
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
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ void st2(float* p, float a, float b) { *reinterpret_cast<f32x2*>(p) = (f32x2){a, b}; }
// leaky ReLU for 0 <= slope <= 1 (checked on the host before launch): max(p, slope * p)
__device__ __forceinline__ float lrelu(float p, float slope) { return fmaxf(p, p * slope); }

// natural log on the hardware log2 unit (v_log_f32, 1 ulp in log2)
__device__ __forceinline__ float flog(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }

// e^x on the hardware exp2 unit (v_exp_f32, 1 ulp in exp2; the argument's rounding adds |x| * 6e-8 relative)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }

// Workgroup barrier number n of the step (the diagnostic build stamps its release).
#define BAR(n) do { lds_barrier(); STAMP(n); } while (0)

#define ZERO4 ((f32x4){0.f, 0.f, 0.f, 0.f})
// LDS offsets are far below 2^23: full-rate 24-bit multiply instead of the quarter-rate v_mul_lo_u32
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// Phase body in two halves executed in opposite order by the two waves of a SIMD (waves w and w+4 share one):
// while waves 0-3 run the matrix half, waves 4-7 run the vector half (Adam, restores), then they swap, so the
// matrix pipe and the VALU of every SIMD are busy together.  One copy of the code: a 2-trip loop whose branch
// is wave-uniform.  The halves must not depend on each other (they run concurrently on other waves anyway).
template <class FM, class FV>
__device__ __forceinline__ void stagger(int w, FM&& matrix_half, FV&& vector_half) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if ((half == 0) == (w < 4)) matrix_half(); else vector_half();
    }
}
#define LAMBDA [&]() __attribute__((always_inline))


// ---- products.  c = lane & 15, q = lane >> 4.  MFMA operand slots: A[i = c][k = q], B[k = q][j = c];
// result: C[i = 4q + r][j = c], r = 0..3.

// The compiler's scheduler sinks LDS reads down to their first use, which exposes the full LDS latency in front
// of every group of MFMAs.  Each product below therefore issues the operand reads of group t+1 BEFORE the MFMAs
// of group t and pins that order with scheduling barriers; the wait-count pass then emits counted lgkmcnt waits.
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Interleave request for the scheduler inside one fenced region: `n` times {1 MFMA, then up to `valu` VALU and `ds`
// LDS instructions}.  Independent vector work placed in the same region as a run of MFMAs is thereby issued in the
// shadow of the matrix instructions (an MFMA holds the SIMD's issue port for 8 of its 32 cycles).
template <int N, int VALU, int DS>
__device__ __forceinline__ void interleave_mfma() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (VALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
        if (DS > 0) __builtin_amdgcn_sched_group_barrier(0x080, DS, 0);
    }
}

struct NoFill { __device__ __forceinline__ void operator()(int) const {} };

// acc[w][j] += X[row][k] * W_w[n_j][k], k in [0, KS), KS = 8G + 4.  xrow = X + row*sx, wrow[w][j] = W_w + n_j*KS.
// Operand reads run PF k-groups ahead of the MFMAs (a ring of PF+1 register sets): one group of 4 MFMAs lasts
// 128 cycles when the wave has the matrix pipe to itself, less than an LDS round trip under load.
#ifndef FT_PF
#define FT_PF 2
#endif
template <int NW, int KS, class F>
__device__ __forceinline__ void mm_fwd(const float* xrow, const float* const (&wrow)[NW][2], int q, f32x4 (&acc)[NW][2], F&& fill) {
    constexpr int G = (KS - 4) / 8, PF = FT_PF, R = PF + 1;
    static_assert(KS == 8 * G + 4 && G >= 1, "row strides are 4*odd, >= 12");
    f32x2 a[R], b[R][NW][2];
    float at = 0.f, bt[NW][2];
    auto fetch = [&](int t) {
        if (t < G) {
            a[t % R] = ld2(xrow + 8 * t + 2 * q);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
#if defined(IVAE_DIAG) && defined(FT_EXP_FWD_NOB)    // timing experiment (wrong results): no weight operand reads
                asm volatile("" : "=v"(b[t % R][w][0]), "=v"(b[t % R][w][1]));
#elif defined(IVAE_DIAG) && defined(FT_EXP_FWD_LINB) // timing experiment (wrong results): conflict-free weight reads
                b[t % R][w][0] = ld2(wrow[0][0] - mul24(min(2 * (int)(threadIdx.x & 15), 49), KS) + 2 * (threadIdx.x & 63) + 128 * (2 * w) + 8 * t);
                b[t % R][w][1] = ld2(wrow[0][0] - mul24(min(2 * (int)(threadIdx.x & 15), 49), KS) + 2 * (threadIdx.x & 63) + 128 * (2 * w + 1) + 8 * t);
#else
                b[t % R][w][0] = ld2(wrow[w][0] + 8 * t + 2 * q);
                b[t % R][w][1] = ld2(wrow[w][1] + 8 * t + 2 * q);
#endif
            }
        } else if (t == G) {
            at = xrow[8 * G + q];
#pragma unroll
            for (int w = 0; w < NW; ++w) { bt[w][0] = wrow[w][0][8 * G + q]; bt[w][1] = wrow[w][1][8 * G + q]; }
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch(t);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int cur = t % R;
        fetch(t + PF);
        SCHED_FENCE();
#pragma unroll
        for (int w = 0; w < NW; ++w) { acc[w][0] = mfma4(a[cur].x, b[cur][w][0].x, acc[w][0]); acc[w][1] = mfma4(a[cur].x, b[cur][w][1].x, acc[w][1]); }
#pragma unroll
        for (int w = 0; w < NW; ++w) { acc[w][0] = mfma4(a[cur].y, b[cur][w][0].y, acc[w][0]); acc[w][1] = mfma4(a[cur].y, b[cur][w][1].y, acc[w][1]); }
        fill(t);                                 // independent vector work, issued between this group's MFMAs
        interleave_mfma<4 * NW, 4, 1>();
        SCHED_FENCE();
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) { acc[w][0] = mfma4(at, bt[w][0], acc[w][0]); acc[w][1] = mfma4(at, bt[w][1], acc[w][1]); }
}

// single 16-wide tile: acc += X[row][k] * W[n][k]; all operands are read up front (2G+2 registers), two
// accumulators break the dependent chain
template <int KS>
__device__ __forceinline__ f32x4 mm_fwd1(const float* xrow, const float* wrow, int q) {
    constexpr int G = (KS - 4) / 8;
    f32x2 a[G], b[G];
#pragma unroll
    for (int t = 0; t < G; ++t) { a[t] = ld2(xrow + 8 * t + 2 * q); b[t] = ld2(wrow + 8 * t + 2 * q); }
    const float at = xrow[8 * G + q], bt = wrow[8 * G + q];
    SCHED_FENCE();
    f32x4 e = ZERO4, o = ZERO4;
#pragma unroll
    for (int t = 0; t < G; ++t) { e = mfma4(a[t].x, b[t].x, e); o = mfma4(a[t].y, b[t].y, o); }
    e = mfma4(at, bt, e);
    return e + o;
}

// acc[j] += dY[row][n] * W[n][col0 + j], n in [0, NS), NS = 8G + 4; rows n >= N (zero columns of dY) are
// clamped to a real row.  yrow = dY + row*sy, wcol = W + col0.
template <int NS, int N, int KS>
__device__ __forceinline__ void mm_dgrad(const float* yrow, const float* wcol, int q, f32x4 (&acc)[2]) {
    constexpr int G = (NS - 4) / 8, PF = FT_PF, R = PF + 1;
    static_assert(NS == 8 * G + 4 && G >= 1, "contraction lengths are 4*odd, >= 12");
    f32x2 a[R], b0[R], b1[R], bt = {0.f, 0.f};
    float at = 0.f;
    auto fetch = [&](int t) {
        if (t < G) {
            int n0 = 8 * t + 2 * q, n1 = n0 + 1;
            if (8 * t + 7 >= N) { n0 = min(n0, N - 1); n1 = min(n1, N - 1); }
            a[t % R] = ld2(yrow + 8 * t + 2 * q);
            b0[t % R] = ld2(wcol + mul24(n0, KS));
            b1[t % R] = ld2(wcol + mul24(n1, KS));
        } else if (t == G) {
            int n = 8 * G + q;
            if (8 * G + 3 >= N) n = min(n, N - 1);
            at = yrow[8 * G + q];
            bt = ld2(wcol + mul24(n, KS));
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch(t);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int cur = t % R;
        fetch(t + PF);
        SCHED_FENCE();
        acc[0] = mfma4(a[cur].x, b0[cur].x, acc[0]);
        acc[1] = mfma4(a[cur].x, b0[cur].y, acc[1]);
        acc[0] = mfma4(a[cur].y, b1[cur].x, acc[0]);
        acc[1] = mfma4(a[cur].y, b1[cur].y, acc[1]);
        SCHED_FENCE();
    }
    acc[0] = mfma4(at, bt.x, acc[0]);
    acc[1] = mfma4(at, bt.y, acc[1]);
}

// single tile, scalar column: acc += dY[row][n] * W[n][col]; operands read up front
template <int NS, int N, int KS>
__device__ __forceinline__ f32x4 mm_dgrad1(const float* yrow, const float* wcol, int q) {
    constexpr int G = (NS - 4) / 8;
    f32x2 a[G];
    float b0[G], b1[G];
#pragma unroll
    for (int t = 0; t < G; ++t) {
        int n0 = 8 * t + 2 * q, n1 = n0 + 1;
        if (8 * t + 7 >= N) { n0 = min(n0, N - 1); n1 = min(n1, N - 1); }
        a[t] = ld2(yrow + 8 * t + 2 * q); b0[t] = wcol[n0 * KS]; b1[t] = wcol[n1 * KS];
    }
    int n = 8 * G + q;
    if (8 * G + 3 >= N) n = min(n, N - 1);
    const float at = yrow[8 * G + q], bt = wcol[n * KS];
    SCHED_FENCE();
    f32x4 e = ZERO4, o = ZERO4;
#pragma unroll
    for (int t = 0; t < G; ++t) { e = mfma4(a[t].x, b0[t], e); o = mfma4(a[t].y, b1[t], o); }
    e = mfma4(at, bt, e);
    return e + o;
}

// acc[j] += dY[b][n] * X[b][col0 + j], b in [0, 64).  ycol = dY + n, xcol = X + col0.  WG_G groups of 2 batch
// k-steps (4 MFMAs each); fill(t) is independent vector work issued between the MFMAs of group t.
#define WG_G (FT_B / 8)
// Which batch rows a weight-gradient MFMA contracts: k-slot q of the k-step (t, i) takes row 16*(t/2) + 2*(t%2) + i +
// {0, 8, 4, 12}[q] (any assignment is a valid contraction as long as both operands use it).  With row strides of 52
// floats (= 52 mod 64 banks) the four k-slots then sit 0, 32, 16, 48 banks apart: the 32-bit reads of the delta operand
// (16 consecutive floats per slot) and the 64-bit reads of the activation operand (32 consecutive floats per slot, two
// slots per LDS cycle) are conflict-free, which consecutive rows q = 0..3 (52 banks apart) are not.
__device__ __forceinline__ constexpr int wg_row(int t, int i) { return 16 * (t / 2) + 2 * (t % 2) + i; }
__device__ __forceinline__ int wg_qoff(int q) { return ((q & 1) << 3) | ((q >> 1) << 2); }
template <class F>
__device__ __forceinline__ void mm_wgrad(const float* ycol, int sy, const float* xcol, int sx, int q, f32x4 (&acc)[2], F&& fill) {
    constexpr int PF = 2, R = PF + 1;
    float a[R][2];
    f32x2 b[R][2];
    auto fetch = [&](int t) {
        if (t < WG_G) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[t % R][i] = ycol[(wg_row(t, i) + wg_qoff(q)) * sy];
                b[t % R][i] = ld2(xcol + (wg_row(t, i) + wg_qoff(q)) * sx);
            }
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch(t);
#pragma unroll
    for (int t = 0; t < WG_G; ++t) {
        const int cur = t % R;
        fetch(t + PF);
        SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[0] = mfma4(a[cur][i], b[cur][i].x, acc[0]); acc[1] = mfma4(a[cur][i], b[cur][i].y, acc[1]); }
        fill(t);
        interleave_mfma<4, 5, 1>();
        SCHED_FENCE();
    }
}

// mm_dgrad followed by mm_wgrad as ONE instruction stream: the weight-gradient product's first operands are requested
// before the input-gradient product's last MFMAs, so the matrix pipe sees no gap between the two, and fill(t) -- the
// input-gradient epilogue, which needs accd complete -- is issued between the weight-gradient MFMAs.
template <int NS, int N, int KS, class F>
__device__ __forceinline__ void mm_dgrad_wgrad(const float* yrow, const float* wcol, const float* ycol, int sy, const float* xcol,
                                               int sx, int q, f32x4 (&accd)[2], f32x4 (&accw)[2], F&& fill, int prof_step = -1) {
    TDECL;
    TMARK(0);
    constexpr int G = (NS - 4) / 8, PF = FT_PF, R = PF + 1;
    static_assert(NS == 8 * G + 4 && G >= 2, "contraction lengths are 4*odd, >= 20");
    f32x2 a[R], b0[R], b1[R], bt = {0.f, 0.f};
    float at = 0.f;
#if defined(IVAE_DIAG) && defined(FT_EXP_LINEAR)     // timing experiment (wrong results): conflict-free operand addresses
    const float* lin = (const float*)((size_t)yrow & ~(size_t)0xfff) + 2 * (threadIdx.x & 63);
#endif
    auto fetch = [&](int t) {
        if (t < G) {
            int n0 = 8 * t + 2 * q, n1 = n0 + 1;
            if (8 * t + 7 >= N) { n0 = min(n0, N - 1); n1 = min(n1, N - 1); }
#if defined(IVAE_DIAG) && defined(FT_EXP_LINEAR)
            a[t % R] = ld2(lin + 384 * t); b0[t % R] = ld2(lin + 384 * t + 128); b1[t % R] = ld2(lin + 384 * t + 256);
#elif defined(IVAE_DIAG) && defined(FT_EXP_NOLOAD)
            asm volatile("" : "=v"(a[t % R]), "=v"(b0[t % R]), "=v"(b1[t % R]));
#else
            a[t % R] = ld2(yrow + 8 * t + 2 * q);
            b0[t % R] = ld2(wcol + mul24(n0, KS));
            b1[t % R] = ld2(wcol + mul24(n1, KS));
#endif
        } else if (t == G) {
            int n = 8 * G + q;
            if (8 * G + 3 >= N) n = min(n, N - 1);
            at = yrow[8 * G + q];
            bt = ld2(wcol + mul24(n, KS));
        }
    };
    constexpr int WR = 3;
    float wa[WR][2];
    f32x2 wb[WR][2];
    auto wfetch = [&](int t) {
        if (t < WG_G) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#if defined(IVAE_DIAG) && defined(FT_EXP_LINEAR)
                wa[t % WR][i] = lin[384 * t + 192 * i - (threadIdx.x & 63)]; wb[t % WR][i] = ld2(lin + 384 * t + 192 * i + 64);
#elif defined(IVAE_DIAG) && defined(FT_EXP_NOLOAD)
                asm volatile("" : "=v"(wa[t % WR][i]), "=v"(wb[t % WR][i]));
#else
                wa[t % WR][i] = ycol[(wg_row(t, i) + wg_qoff(q)) * sy];
                wb[t % WR][i] = ld2(xcol + (wg_row(t, i) + wg_qoff(q)) * sx);
#endif
            }
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch(t);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int cur = t % R;
        fetch(t + PF);
        if (t == G - 2) wfetch(0);
        if (t == G - 1) wfetch(1);
        SCHED_FENCE();
        if (t == 0) TMARK(1);
        if (t == 3) TMARK(2);
        accd[0] = mfma4(a[cur].x, b0[cur].x, accd[0]);
        accd[1] = mfma4(a[cur].x, b0[cur].y, accd[1]);
        accd[0] = mfma4(a[cur].y, b1[cur].x, accd[0]);
        accd[1] = mfma4(a[cur].y, b1[cur].y, accd[1]);
        SCHED_FENCE();
    }
    accd[0] = mfma4(at, bt.x, accd[0]);
    accd[1] = mfma4(at, bt.y, accd[1]);
    TMARK(3);
#pragma unroll
    for (int t = 0; t < WG_G; ++t) {
        const int cur = t % WR;
        wfetch(t + 2);
        SCHED_FENCE();
        if (t == 4) TMARK(4);
#pragma unroll
        for (int i = 0; i < 2; ++i) { accw[0] = mfma4(wa[cur][i], wb[cur][i].x, accw[0]); accw[1] = mfma4(wa[cur][i], wb[cur][i].y, accw[1]); }
        fill(t);
        interleave_mfma<4, 5, 1>();
        SCHED_FENCE();
    }
    TMARK(5);
#ifdef FT_PROF
    if (prof_step >= 0) TSTORE(prof_step, 0);
#endif
}

// single tile: operands of all 16 k-steps read up front (32 registers)
__device__ __forceinline__ f32x4 mm_wgrad1(const float* ycol, int sy, const float* xcol, int sx, int q) {
    float a[FT_B / 4], b[FT_B / 4];
#pragma unroll
    for (int t = 0; t < FT_B / 4; ++t) { a[t] = ycol[(16 * (t / 4) + t % 4 + wg_qoff(q)) * sy]; b[t] = xcol[(16 * (t / 4) + t % 4 + wg_qoff(q)) * sx]; }
    SCHED_FENCE();
    f32x4 e = ZERO4, o = ZERO4;
#pragma unroll
    for (int t = 0; t < FT_B / 4; t += 2) { e = mfma4(a[t], b[t], e); o = mfma4(a[t + 1], b[t + 1], o); }
    return e + o;
}

// ---- layer blocks.  Wave w owns rows 16*(w&3).. and the column half 32*(w>>2) + 2c + j of 64-wide outputs.

// ---- wave-local forward chains.  In the forward stage a wave owns a whole 16-row tile of ONE chain (all 64 output
// columns = four 16-wide tiles), so layer l+1 of the chain reads only rows the same wave wrote: no workgroup barrier
// between the layers of a chain.  keep[p][j] is the tile of columns 32p + 16j + c: lane c reads weight row 32p + 16j + c,
// so the 16 lanes of an operand read walk CONSECUTIVE weight rows (stride KS = 4*odd floats: conflict-free 64-bit reads;
// interleaved column pairs 2c, 2c + 1 put lanes c and c + 8 on the same banks -- measured: 0.8 us of the 34 us step).
// The price is 32-bit result stores (columns c and c + 16 are not adjacent): 16 per layer instead of 8.
template <class C, int KS, class F>
__device__ __forceinline__ void fwd_rows_product(const float* X, int sx, const float* W, int rt, int lane,
                                                 f32x4 (&keep)[2][2], F&& fill) {
    const int c = lane & 15, q = lane >> 4;
    const float* wrow[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * p + 16 * j + c;
            wrow[p][j] = W + mul24(32 * p + 16 * j + 15 < C::H ? col : min(col, C::H - 1), KS);
            keep[p][j] = ZERO4;
        }
    mm_fwd<2, KS>(X + mul24(16 * rt + c, sx), wrow, q, keep, fill);
}

// Epilogue of a wave-local hidden layer: activation, the constant-1 column H (bias column of the next layer) and zeros
// beyond it, stores, and the values kept for the stash.  Lanes whose column lies past the tile's last one (ST - 1) write
// that column's value (0) there, as its owner does: no exec-masked store, no branch.
template <class C>
__device__ __forceinline__ void fwd_rows_epilogue(f32x4 (&keep)[2][2], float* out, float slope, int rt, int lane) {
    const int c = lane & 15, q = lane >> 4;
    static_assert(C::ST - 1 > C::H, "the clamped column holds no product column and not the constant 1");
    float* base = out + mul24(16 * rt + 4 * q, C::ST);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * p + 16 * j + c;
            const bool whole = 32 * p + 16 * j + 15 < C::H;            // compile time: every lane's column is a product column
            float* dst = base + (whole ? col : min(col, C::ST - 1));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = lrelu(keep[p][j][r], slope);
                if (!whole) v = col < C::H ? v : (col == C::H ? 1.f : 0.f);
                keep[p][j][r] = v;
                dst[r * C::ST] = v;
            }
        }
}

// out[b][n] = lrelu(X[b][:] . W[n][:]) for the 16 rows of tile rt, every column; values kept for the stash.
template <class C, int KS>
__device__ __forceinline__ void fwd_rows(const float* X, int sx, const float* W, float* out, float slope, int rt,
                                         f32x4 (&keep)[2][2]) {
    const int lane = opaque_lane();
    fwd_rows_product<C, KS>(X, sx, W, rt, lane, keep, NoFill());
    fwd_rows_epilogue<C>(keep, out, slope, rt, lane);
}

// ---- the same chains as continuous MFMA streams (round 3).  A layer's product runs as two passes, tile pair p = 0
// (columns 0..31) then p = 1 (columns 32..63), and no epilogue ever runs with the matrix pipe idle: the epilogue of pair 0 is
// issued between the MFMAs of pair 1, and the epilogue of pair 1 is left PENDING -- the next layer of the chain performs it
// between the MFMAs of its own first pass, whose first four k-groups (k < 32) only read columns the pair-0 epilogue wrote.
// The operand ring of a pass is requested inside the pass before it, so a pass starts on operands that are already there.
template <int KS>
struct PairOps {
    static constexpr int G = (KS - 4) / 8, R = 3;
    static_assert(KS == 8 * G + 4 && G >= 1, "row strides are 4*odd, >= 12");
    const float* xrow;
    const float* w0;
    const float* w1;
    f32x2 a[R], b0[R], b1[R];
    float at, bt0, bt1;
    __device__ __forceinline__ void fetch(int t, int q) {
        if (t < G) {
            a[t % R] = ld2(xrow + 8 * t + 2 * q);
            b0[t % R] = ld2(w0 + 8 * t + 2 * q);
            b1[t % R] = ld2(w1 + 8 * t + 2 * q);
        } else if (t == G) {
            at = xrow[8 * G + q];
            bt0 = w0[8 * G + q];
            bt1 = w1[8 * G + q];
        }
    }
};

// operands of pair p of the layer X -> W for row tile rt; the first two k-groups are requested right away
template <class C, int KS>
__device__ __forceinline__ void pair_ops_start(PairOps<KS>& o, const float* X, int sx, const float* W, int rt, int lane, int p) {
    const int c = lane & 15, q = lane >> 4;
    o.xrow = X + mul24(16 * rt + c, sx);
    o.w0 = W + mul24(min(32 * p + c, C::H - 1), KS);
    o.w1 = W + mul24(min(32 * p + 16 + c, C::H - 1), KS);
    o.fetch(0, q);
    o.fetch(1, q);
}

// one pass: acc[j] += X W_j^T over the whole contraction; fill(t) is issued between the MFMAs of k-group t
template <int KS, class F>
__device__ __forceinline__ void pair_pass(PairOps<KS>& o, int q, f32x4 (&acc)[2], F&& fill) {
    constexpr int G = PairOps<KS>::G, R = PairOps<KS>::R;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int cur = t % R;
        o.fetch(t + 2, q);
        SCHED_FENCE();
        acc[0] = mfma4(o.a[cur].x, o.b0[cur].x, acc[0]);
        acc[1] = mfma4(o.a[cur].x, o.b1[cur].x, acc[1]);
        acc[0] = mfma4(o.a[cur].y, o.b0[cur].y, acc[0]);
        acc[1] = mfma4(o.a[cur].y, o.b1[cur].y, acc[1]);
        fill(t);
        interleave_mfma<4, 7, 2>();
        SCHED_FENCE();
    }
    acc[0] = mfma4(o.at, o.bt0, acc[0]);
    acc[1] = mfma4(o.at, o.bt1, acc[1]);
}

// epilogue of tile (p, j) = columns 32p + 16j + c of a hidden layer, rows 16rt + 4q + r: activation, constant-1 column H,
// zeros beyond, 32-bit stores; lanes past the tile's last column write that column's value (0) there, as its owner does
template <class C>
__device__ __forceinline__ void tile_epilogue(f32x4& v, int p, int j, float* out, float slope, int rt, int lane) {
    const int c = lane & 15, q = lane >> 4;
    const int col = 32 * p + 16 * j + c;
    const bool whole = 32 * p + 16 * j + 15 < C::H;                // every lane's column is a product column (p, j: constants)
    float* dst = out + mul24(16 * rt + 4 * q, C::ST) + (whole ? col : min(col, C::ST - 1));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = lrelu(v[r], slope);
        if (!whole) x = col < C::H ? x : (col == C::H ? 1.f : 0.f);
        v[r] = x;
        dst[r * C::ST] = x;
    }
}

// One hidden layer of a chain.  o0: operands of pair 0, already started.  keep: this layer's four tiles -- on return
// keep[0][*] hold the epilogued pair 0 (stored to `out`), keep[1][*] the RAW accumulators of pair 1 (pending).  PEND: the
// previous layer's pending pair (`pend`, epilogued here into `pend_out` = this layer's input tile).  next(): called late in
// the second pass, for the caller to start the operands of whatever follows.
template <class C, int KS, bool PEND, class NEXT>
__device__ __forceinline__ void chain_layer(PairOps<KS>& o0, const float* X, int sx, const float* W, float* out, float slope,
                                            int rt, int lane, f32x4 (&keep)[2][2], f32x4 (&pend)[2], float* pend_out,
                                            NEXT&& next) {
    constexpr int G = PairOps<KS>::G;
    const int q = lane >> 4;
    PairOps<KS> o1;
#pragma unroll
    for (int p = 0; p < 2; ++p) { keep[p][0] = ZERO4; keep[p][1] = ZERO4; }
    pair_pass<KS>(o0, q, keep[0], [&](int t) __attribute__((always_inline)) {
        if (PEND && t < 2) tile_epilogue<C>(pend[t], 1, t, pend_out, slope, rt, lane);       // before k-group 4 is requested
        if (t == (G >= 2 ? G - 2 : 0)) pair_ops_start<C, KS>(o1, X, sx, W, rt, lane, 1);
    });
    pair_pass<KS>(o1, q, keep[1], [&](int t) __attribute__((always_inline)) {
        if (t < 2) tile_epilogue<C>(keep[0][t], 0, t, out, slope, rt, lane);
        if (G == 1 && t == 0) tile_epilogue<C>(keep[0][1], 0, 1, out, slope, rt, lane);
        if (t == (G >= 2 ? G - 2 : 0)) next();
    });
}

// Head of a chain (one 16-wide tile, contraction over the H + 1 columns of the last hidden tile X) with that layer's
// pending pair: the k < 32 part first, the pending epilogue between its MFMAs, then the rest.
template <class C>
__device__ __forceinline__ f32x4 chain_head(const float* xrow, const float* wrow, int q, f32x4 (&pend)[2], float* pend_out,
                                            float slope, int rt, int lane) {
    constexpr int KS = C::KS_H, G = (KS - 4) / 8;
    static_assert(G >= 5, "the head's contraction covers k < 32 in its first four k-groups");
    f32x2 a[G], b[G];
#pragma unroll
    for (int t = 0; t < 4; ++t) { a[t] = ld2(xrow + 8 * t + 2 * q); b[t] = ld2(wrow + 8 * t + 2 * q); }
    SCHED_FENCE();
    f32x4 e = ZERO4, o = ZERO4;
#pragma unroll
    for (int t = 0; t < 4; ++t) { e = mfma4(a[t].x, b[t].x, e); o = mfma4(a[t].y, b[t].y, o); }
    tile_epilogue<C>(pend[0], 1, 0, pend_out, slope, rt, lane);
    tile_epilogue<C>(pend[1], 1, 1, pend_out, slope, rt, lane);
    interleave_mfma<8, 6, 2>();
    SCHED_FENCE();
#pragma unroll
    for (int t = 4; t < G; ++t) { a[t] = ld2(xrow + 8 * t + 2 * q); b[t] = ld2(wrow + 8 * t + 2 * q); }
    const float at = xrow[8 * G + q], bt = wrow[8 * G + q];
    SCHED_FENCE();
#pragma unroll
    for (int t = 4; t < G; ++t) { e = mfma4(a[t].x, b[t].x, e); o = mfma4(a[t].y, b[t].y, o); }
    e = mfma4(at, bt, e);
    return e + o;
}

// Hand-offs between the two waves that own one row tile (waves rt and rt + 4: the two waves of one SIMD) inside the
// forward stage.  A flag is one LDS word holding the number of the step it was last raised in; it lives in the unused
// pad column of the resident input tile (column KIN + 1 of T3, which multiplies zero weight columns only).  The
// producer drains its LDS traffic before raising it; LDS operations of a CU complete in issue order, so a consumer that
// sees the flag sees the data.  The wait is bounded: a protocol error ends in wrong numbers, never in a hung GPU.
// (Explicit LDS address space: through a generic pointer these would be FLAT accesses, whose wait also drains the
// global loads of the next batch's prefetch.)
typedef __attribute__((address_space(3))) volatile int lds_flag_t;
#ifndef FT_POLL_SLEEP
#define FT_POLL_SLEEP 1
#endif
__device__ __forceinline__ void flag_raise(float* flag, int step_no) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(lds_flag_t*)flag = step_no;
}
// A wait that runs out of polls (2^16 of them: tens of milliseconds, a thousand times the longest legitimate wait) raises
// the model's error word; from that step on every loss the model reports is NaN (k_train_steps, loss out), so a missed
// hand-off can never pass as a trained model.
__device__ __forceinline__ void flag_wait(const float* flag, int step_no, float* err) {
#if defined(IVAE_DIAG) && defined(FT_EXP_NOFLAGWAIT)     // timing experiment (wrong results)
    return;
#endif
    bool seen = false;
#pragma unroll 1
    for (int spin = 0; spin < (1 << 16); ++spin) {
        if (*(lds_flag_t*)flag >= step_no) { seen = true; break; }
        __builtin_amdgcn_s_sleep(FT_POLL_SLEEP);
    }
    if (!seen) *(lds_flag_t*)err = 1;
    asm volatile("" ::: "memory");
}

// Stash of a wave-local layer (post-epilogue values kept in registers, fwd_rows_epilogue's layout) -> its rows of an LDS tile.
template <class C>
__device__ __forceinline__ void restore_rows(const f32x4 (&keep)[2][2], float* out, int rt) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    float* base = out + mul24(16 * rt + 4 * q, C::ST);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* dst = base + min(32 * p + 16 * j + c, C::ST - 1);       // clamped like the epilogue that filled `keep`
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * C::ST] = keep[p][j][r];
        }
}

// Head forward (one 16-wide tile per 16 rows; 4 waves): out[b][n] = epi(sum_k X[b][k] W[n][k]), n < N.  EXP: exp().
template <class C, int N, bool EXP>
__device__ __forceinline__ void fwd_head(const float* X, const float* W, float* out, int rt) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const f32x4 acc = mm_fwd1<C::KS_H>(X + mul24(16 * rt + c, C::ST), W + mul24(min(c, N - 1), C::KS_H), q);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c < N) out[mul24(16 * rt + 4 * q, C::SS) + r * C::SS + c] = EXP ? fexp(acc[r]) : acc[r];
}

// Hidden-layer dgrad: out[b][k] = mask(h[b][k]) * sum_n dY[b][n] W[n][k] for k < KB, 0 beyond.
template <class C, int NS, int N, int KS, int KB>
__device__ __forceinline__ void dgrad_hidden(const float* dY, int sy, const float* W, const float* hm, float* out,
                                             float slope, int w) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const int rt = w & 3, col0 = 32 * (w >> 2) + 2 * c;
    f32x4 acc[2] = {ZERO4, ZERO4};
    mm_dgrad<NS, N, KS>(dY + mul24(16 * rt + c, sy), W + min(col0, KS - 2), q, acc);
    const bool in0 = col0 < KB, in1 = col0 + 1 < KB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = mul24(16 * rt + 4 * q, C::ST) + r * C::ST;      // row offset in floats
        if (col0 < C::ST) {
            const f32x2 h = ld2(hm + row + col0);
            const float v0 = in0 ? acc[0][r] * (h.x > 0.f ? 1.f : slope) : 0.f;
            const float v1 = in1 ? acc[1][r] * (h.y > 0.f ? 1.f : slope) : 0.f;
            st2(out + row + col0, v0, v1);
        }
    }
}

// Weight gradient of a layer with > 16 input columns: stage[n][k] = sum_b dY[b][n] X[b][k] for k <= KB (k == KB
// is the bias gradient: X carries a constant-1 column there), 0 beyond.  Wave w: rows 16*(w&3).., column half w>>2.
template <class C, int N, int KS, int KB, bool ROWGUARD = false, class F = NoFill>
__device__ __forceinline__ void wgrad_pair(const float* dY, int sy, const float* X, int sx, float* stage, int w, F&& fill = F()) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const int nt = w & 3, col0 = 32 * (w >> 2) + 2 * c;
    f32x4 acc[2] = {ZERO4, ZERO4};
    mm_wgrad(dY + min(16 * nt + c, sy - 1), sy, X + min(col0, KS - 2), sx, q, acc, fill);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 16 * nt + 4 * q + r;       // rows >= N land in the tail of the (64-row) stage tile unless guarded
        if (col0 < KS && (!ROWGUARD || n < N)) st2(stage + n * KS + col0, col0 <= KB ? acc[0][r] : 0.f, col0 + 1 <= KB ? acc[1][r] : 0.f);
    }
}

// The middle phase of an MLP's backward pass as one stream per wave (wave w: rows 16*(w&3).., column half w>>2 of both
// products): d1 = mask(h1) * (d2 W1) -> out, then the weight gradient of the same layer dW1 = d2^T [h1 | 1] -> stage.
// The input-gradient epilogue (mask from h1, read before the first product) is issued between the weight-gradient MFMAs.
template <class C, bool ROWGUARD>
__device__ __forceinline__ void dgrad_wgrad_hidden(const float* dY, const float* W, const float* h, float* out, float* stage,
                                                   float slope, int w, int prof_step = -1) {
    constexpr int ST = C::ST, KS = C::KS_H, H = C::H;
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const int rt = w & 3, col0 = 32 * (w >> 2) + 2 * c;
    // lanes whose column pair lies past the tile write zeros to its last pair, as that pair's owner does (columns H, H + 1)
    const int colw = min(col0, ST - 2);
    static_assert(ST - 2 >= H, "the clamped column pair holds no product column");
    const float* hrow = h + mul24(16 * rt + 4 * q, ST) + colw;
    f32x2 hm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hm[r] = ld2(hrow + r * ST);
    f32x4 accd[2] = {ZERO4, ZERO4}, accw[2] = {ZERO4, ZERO4};
    const bool in0 = col0 < H, in1 = col0 + 1 < H;
    float* orow = out + mul24(16 * rt + 4 * q, ST) + colw;
    mm_dgrad_wgrad<KS, H, KS>(dY + mul24(16 * rt + c, ST), W + min(col0, KS - 2), dY + min(16 * rt + c, ST - 1), ST,
                              h + min(col0, KS - 2), ST, q, accd, accw, [&](int t) __attribute__((always_inline)) {
        if (t < 4) {
            const float v0 = in0 ? accd[0][t] * (hm[t].x > 0.f ? 1.f : slope) : 0.f;
            const float v1 = in1 ? accd[1][t] * (hm[t].y > 0.f ? 1.f : slope) : 0.f;
            st2(orow + t * ST, v0, v1);
        }
    }, prof_step);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 16 * rt + 4 * q + r;       // rows >= H land in the tail of the (64-row) stage tile unless guarded
        if (col0 < KS && (!ROWGUARD || n < H)) st2(stage + n * KS + col0, col0 <= H ? accw[0][r] : 0.f, col0 + 1 <= H ? accw[1][r] : 0.f);
    }
}

// Single-tile weight gradient: rows 16*nt.., columns 16*kt + c.
template <class C, int N, int KS, int KB>
__device__ __forceinline__ void wgrad_tile(const float* dY, int sy, const float* X, int sx, float* stage, int nt, int kt) {
    const int lane = opaque_lane();
    const int c = lane & 15, q = lane >> 4;
    const int col = 16 * kt + c;
    const f32x4 acc = mm_wgrad1(dY + min(16 * nt + c, sy - 1), sy, X + min(col, KS - 1), sx, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 16 * nt + 4 * q + r;
        if (n < N && col < KS) stage[n * KS + col] = col <= KB ? acc[r] : 0.f;
    }
}

// Adam for layer L: gradient from `stage` (same (N, KS) layout as the layer in the arena).
// Same update as k_adam (adam.hip) / torch.optim.Adam; sqrt and the final divide use the hardware
// v_sqrt_f32 / v_rcp_f32 (1 ulp) -- a <=2e-7 relative change of a step of size ~lr.
template <class C, int L>
struct AdamOp {
    static constexpr int P = C::p_of(L), KC = C::kb_of(L) + 1, KS = C::ks_of(L), R0 = C::reg_of(L), CNT = C::cnt_of(L);
    static constexpr int NS = CNT > 0 ? CNT : 1;
    unsigned t;
    float g[NS], p[NS];
    float* base;

    __device__ __forceinline__ unsigned offset(int j) const {
        const unsigned idx = min(t + j * FT_THREADS, (unsigned)(P - 1));
        const unsigned n = idx / (unsigned)KC;
        return n * KS + (idx - n * KC);
    }

    // Branch-free: out-of-range slots work on a clamped (valid) element and only skip the store; their moment
    // registers hold garbage that is never written back.  All LDS reads of the layer are issued here, before any math.
    __device__ __forceinline__ void fetch(float* arena, const float* stage) {
        base = arena + C::off_of(L);
        t = threadIdx.x;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
            const unsigned o = offset(j);
            g[j] = stage[o];
            p[j] = base[o];
        }
    }

    // Same update as k_adam (adam.hip) / torch.optim.Adam; sqrt and the final divide use the hardware
    // v_sqrt_f32 / v_rcp_f32 (1 ulp) -- a <=2e-7 relative change of a step of size ~lr.
    template <int NA>
    __device__ __forceinline__ void slot(int j, float (&am)[NA], float (&av)[NA], float w1, float b2, float w2,
                                         float neg_step, float inv_bc2, float eps) {
        const float m0 = am[R0 + j], v0 = av[R0 + j];
        const float mi = m0 + w1 * (g[j] - m0);
        const float vi = v0 * b2 + (w2 * g[j]) * g[j];
        const float denom = __builtin_amdgcn_sqrtf(vi) * inv_bc2 + eps;
        const float pn = p[j] + (neg_step * mi) * __builtin_amdgcn_rcpf(denom);
        am[R0 + j] = mi;
        av[R0 + j] = vi;
        if ((j + 1) * FT_THREADS <= P || t + j * FT_THREADS < (unsigned)P) base[offset(j)] = pn;
    }
};

// Adam for layer L: gradient from `stage` (same (N, KS) layout as the layer in the arena).
template <class C, int L, int NA>
__device__ __forceinline__ void adam_layer(float* arena, const float* stage, float (&am)[NA], float (&av)[NA], float w1,
                                           float b2, float w2, float neg_step, float inv_bc2, float eps) {
    if constexpr (C::cnt_of(L) > 0) {
        AdamOp<C, L> op;
        op.fetch(arena, stage);
        SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < C::cnt_of(L); ++j) op.slot(j, am, av, w1, b2, w2, neg_step, inv_bc2, eps);
    }
}

// torch arena index of element (n, k) of layer L's (N, KB+1) index space, or -1 for the hole column that
// keeps u on an even column of the encoder input when data_dim is odd.
template <class C, int L>
__device__ __forceinline__ int torch_index(int n, int k) {
    constexpr int KB = C::kb_of(L), KR = C::kr_of(L), G0 = C::goff_of(L), N = C::n_of(L);
    if (k == KB) return G0 + N * KR + n;                      // bias
    if ((L == L_G0 || L == L_V0) && C::HAS_U && C::UOFF != C::DD) {
        if (k >= C::DD && k < C::UOFF) return -1;
        if (k >= C::UOFF) k -= C::UOFF - C::DD;
    }
    return G0 + n * KR + k;
}

template <class C, int L, int NA>
__device__ __forceinline__ void state_load(const float* gP, const float* gM, const float* gV, float* arena,
                                           float (&am)[NA], float (&av)[NA]) {
    constexpr int P = C::p_of(L), KC = C::kb_of(L) + 1, KS = C::ks_of(L), R0 = C::reg_of(L);
#pragma unroll
    for (int j = 0; j < C::cnt_of(L); ++j) {
        const int idx = threadIdx.x + j * FT_THREADS;
        float m = 0.f, v = 0.f;
        if (idx < P) {
            const int n = idx / KC, k = idx - n * KC;
            const int gi = torch_index<C, L>(n, k);
            if (gi >= 0) { arena[C::off_of(L) + n * KS + k] = gP[gi]; m = gM[gi]; v = gV[gi]; }
        }
        am[R0 + j] = m;
        av[R0 + j] = v;
    }
}

template <class C, int L, int NA>
__device__ __forceinline__ void state_store(float* gP, float* gM, float* gV, const float* arena, const float (&am)[NA],
                                            const float (&av)[NA]) {
    constexpr int P = C::p_of(L), KC = C::kb_of(L) + 1, KS = C::ks_of(L), R0 = C::reg_of(L);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));        // re-derive the indices here: values shared with state_load would live (spilled) across the step loop
#pragma unroll
    for (int j = 0; j < C::cnt_of(L); ++j) {
        const int idx = tid + j * FT_THREADS;
        if (idx < P) {
            const int n = idx / KC, k = idx - n * KC;
            const int gi = torch_index<C, L>(n, k);
            if (gi >= 0) { gP[gi] = arena[C::off_of(L) + n * KS + k]; gM[gi] = am[R0 + j]; gV[gi] = av[R0 + j]; }
        }
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

// Next batch [x | u] into registers: thread = (row b = tid >> 3, column phase p = tid & 7); it fetches x[row][p + 8j]
// and u[row][p + 8j].  Every load instruction reads one array and is unconditional (out-of-range columns are
// clamped to a valid element and simply not stored), so nothing here branches and no load waits on another.
template <class C>
__device__ __forceinline__ void load_xu(float (&xu)[C::NXU], int row, const float* xd, const float* ud) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int ph = t & 7;
    const float* xr = xd + (size_t)row * C::DD;
#pragma unroll
    for (int j = 0; j < C::NXD; ++j) xu[j] = xr[min(ph + 8 * j, C::DD - 1)];
    if constexpr (C::HAS_U) {
        const float* ur = ud + (size_t)row * C::M;
#pragma unroll
        for (int j = 0; j < C::NXM; ++j) xu[C::NXD + j] = ur[min(ph + 8 * j, C::M - 1)];
    }
}

// ... and from the registers into the resident [x | u | 1] tile (the constant-1 and zero columns are written once
// at kernel start and never touched again).
template <class C>
__device__ __forceinline__ void store_xu(const float (&xu)[C::NXU], float* tile) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int ph = t & 7;
    float* dst = tile + (t >> 3) * C::ST + ph;
#pragma unroll
    for (int j = 0; j < C::NXD; ++j) if (8 * j + 7 < C::DD || ph + 8 * j < C::DD) dst[8 * j] = xu[j];
    if constexpr (C::HAS_U) {
#pragma unroll
        for (int j = 0; j < C::NXM; ++j) if (8 * j + 7 < C::M || ph + 8 * j < C::M) dst[C::UOFF + 8 * j] = xu[C::NXD + j];
    }
}

// ---------------------------------------------------------------------------------- kernel
template <class C>
__global__ __launch_bounds__(FT_THREADS) void k_train_steps(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T0 = lds;
    float* T1 = T0 + C::TILE;
    float* T2 = T1 + C::TILE;
    float* T3 = T2 + C::TILE;          // [x | u | 1] of the current batch, resident for the whole step
    float* T4 = T3 + C::TILE;
    float* sG = T4 + C::TILE;          // g, then dLoss/dg
    float* sV = sG + C::SMALL;         // v = exp(logv), then dLoss/dlogv
    float* sL = sV + C::SMALL;         // l = exp(logl), then dLoss/dlogl
    float* sZ = sL + C::SMALL;         // eps, then z (with its constant-1 column), then the head-gradient stage
    float* sO = sZ + C::SMALL;         // dLoss/df (decoder head delta), forward stage..BF3
    float* arena = sO + C::SMALL;      // parameters, last: anything read past the end of a tile is still a finite LDS word
    float* stF0 = T1 + C::SMALL;       // gradient stage of the decoder's first layer, BF1..G3
    float* red = stF0 + C::H * C::KS_F0;   // 8 loss partials, BF1..G3
    // hand-off flags of the forward stage: pad column of T3, rows 0..11
    float* flagV = T3 + C::KIN + 1;                  // [rt]: v rows of tile rt are in sV, and T4's rows rt are free
    float* flagG = flagV + 4 * C::ST;                // [rt]: the g chain has read its last word of T0's rows rt
    float* flagZ = flagV + 8 * C::ST;                // [rt]: z rows of tile rt are in sZ
    float* flagL = flagV + 12 * C::ST;               // [rt]: h1l rows of tile rt are in T1
    float* flagE = flagV + 16 * C::ST;               // error word: a hand-off wait timed out (sticky for the launch)
    static_assert(C::KIN + 1 < C::ST, "the input tile has a pad column for the flags");

    constexpr int ST = C::ST, SS = C::SS, H = C::H, DD = C::DD, DL = C::DL, M = C::M;
    constexpr int KS_H = C::KS_H, KS_E0 = C::KS_E0, KS_L0 = C::KS_L0, KS_F0 = C::KS_F0;
    const int tid = threadIdx.x;
    const int w_init = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int model = blockIdx.x;
    float* gP = A.params + (size_t)model * C::NPARAM;
    float* gM = A.exp_avg + (size_t)model * C::NPARAM;
    float* gV = A.exp_avg_sq + (size_t)model * C::NPARAM;
    const float* xd = A.x_data + (size_t)model * A.n_data * DD;
    const float* ud = C::HAS_U ? A.u_data + (size_t)model * A.n_data * M : A.x_data;
    const int* idxp = A.batch_idx + (size_t)model * A.idx_stride + (size_t)A.step_begin * FT_B;
    const float slope = A.slope;
    const float wB = 1.f / (float)FT_B;

    // ---- load state: zero LDS (padding must be finite and the bias/zero columns exact), parameters -> LDS,
    // Adam moments -> registers
    for (int i = tid; i < C::LDS_FLOATS; i += FT_THREADS) lds[i] = 0.f;
    __syncthreads();
    float am[C::NADAM], av[C::NADAM];
    if constexpr (C::HAS_U) {
        state_load<C, L_L0>(gP, gM, gV, arena, am, av);
        state_load<C, L_L1>(gP, gM, gV, arena, am, av);
        state_load<C, L_L2>(gP, gM, gV, arena, am, av);
    }
    state_load<C, L_F0>(gP, gM, gV, arena, am, av);
    state_load<C, L_F1>(gP, gM, gV, arena, am, av);
    state_load<C, L_F2>(gP, gM, gV, arena, am, av);
    state_load<C, L_G0>(gP, gM, gV, arena, am, av);
    state_load<C, L_G1>(gP, gM, gV, arena, am, av);
    state_load<C, L_G2>(gP, gM, gV, arena, am, av);
    state_load<C, L_V0>(gP, gM, gV, arena, am, av);
    state_load<C, L_V1>(gP, gM, gV, arena, am, av);
    state_load<C, L_V2>(gP, gM, gV, arena, am, av);
    if (tid < FT_B) T3[tid * ST + C::KIN] = 1.f;      // constant-1 (bias) column of the resident input tile
    __syncthreads();

    const float w1 = A.w1, b2 = A.beta2, w2 = A.w2, aeps = A.adam_eps;
    float xu[C::NXU];
    // row indices are clamped to the dataset: a bad index reads a wrong (valid) row instead of faulting the GPU
    const int last_row = (int)(A.n_data - 1);
    load_xu<C>(xu, min(max(idxp[tid >> 3], 0), last_row), xd, ud);
    float neg_step = 0.f, inv_bc2 = 1.f;       // of the previous step (its last layer's Adam runs in P0)
    const float lr_scale = A.lr_scale[model];  // this model's learning rate relative to the table's (1: exact)

// Weight-gradient product of one layer with Adam of another layer (LA, gradient in stage_a) issued between its MFMAs.
#define WGRAD_ADAM(N_, KS_, KB_, DY_, SY_, X_, SX_, STAGE_, LA_, STAGE_A_)                                              \
    do {                                                                                                               \
        AdamOp<C, LA_> op_;                                                                                            \
        op_.fetch(arena, STAGE_A_);                                                                                    \
        wgrad_pair<C, N_, KS_, KB_>(DY_, SY_, X_, SX_, STAGE_, w, [&](int t_) __attribute__((always_inline)) {           \
            constexpr int CNT_ = AdamOp<C, LA_>::CNT;                                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < CNT_; ++j_)                                                        \
                if (j_ * WG_G / (CNT_ > 0 ? CNT_ : 1) == t_) op_.slot(j_, am, av, w1, b2, w2, neg_step, inv_bc2, aeps);   \
        });                                                                                                            \
    } while (0)

#define ADAM(L_, STAGE_) adam_layer<C, L_>(arena, STAGE_, am, av, w1, b2, w2, neg_step, inv_bc2, aeps)

    for (int s = 0; s < A.n_steps; ++s) {
        // wave id / lane re-derived per step behind an optimisation barrier so the address arithmetic of ~20
        // phases is not hoisted out of the step loop
        int w = w_init;
        asm volatile("" : "+s"(w));
        const int lane = opaque_lane();
        const int c = lane & 15, q = lane >> 4;
        const int rt = w & 3;

        STAMP(0);
        // ---- P0: Adam of the previous step's last layer; [x | u | 1] (prefetched) -> T3; eps -> sZ
        if (s > 0) {
            if constexpr (C::HAS_U) ADAM(L_L0, T0); else ADAM(L_V0, T1);
        }
        neg_step = A.neg_step[s] * lr_scale;
        inv_bc2 = A.inv_bc2_sqrt[s];
        store_xu<C>(xu, T3);
        if (A.eps) {
            for (int e = tid; e < FT_B * DL; e += FT_THREADS) {
                const int b = e / DL, k = e - b * DL;
                sZ[b * SS + k] = A.eps[(((size_t)model * A.n_steps_total + A.step_begin + s) * FT_B + b) * DL + k];
            }
        } else if (tid * 4 < FT_B * DL) {
            float nrm[4];
            philox_normal4(A.seed, (unsigned)tid, (unsigned)(A.step0 + s), (unsigned)(A.model0 + model), 0x1234567u, nrm);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int el = tid * 4 + i; sZ[(el / DL) * SS + el % DL] = nrm[i]; }
        }
        // row index of the next step's batch (the last step re-reads its own: unconditional, so that the wait-count
        // pass never sees a load that is pending on one path only and answers with a vmcnt(0) mid-phase)
        const int next_row = min(max(idxp[(size_t)min(s + 1, A.n_steps - 1) * FT_B + (tid >> 3)], 0), last_row);
        BAR(1);

        // ---- forward stage, wave-local chains (no workgroup barrier inside).  Row tile rt = w & 3 is owned by the
        // two waves of one SIMD: wave rt runs the g chain and then the decoder, wave rt + 4 the logv chain and then the
        // prior.  The chains of a pair are independent instruction streams, so one wave's epilogues and LDS round
        // trips run under the other's MFMAs.  Tiles (rows rt only): g: T3 -> T0 -> T2 -> sG; v: T3 -> T1 -> T4 -> sV;
        // prior: u(T3) -> T1 -> T0 -> sL; decoder: sZ -> T2 -> T4 -> sO.  Hand-offs between the pair: flags.
        // stash (48 registers a lane): waves 0-3 hold h1g, h2g and -- read back from the partner's tile at the end of the
        // stage, so that both waves of a pair carry the same load -- h1l; waves 4-7 hold h1v, h2v, h2l
        f32x4 s1[2][2], s2[2][2], s3[2][2];
        float lsum = 0.f;
        const int step_no = s + 1;
        load_xu<C>(xu, next_row, xd, ud);                  // next batch's rows start to arrive in registers
#if defined(IVAE_DIAG) && defined(FT_EXP_FWD_ONLY_G)      // timing experiment (wrong results): waves 4-7 idle in the forward stage
        if (w >= 4) {} else
#elif defined(IVAE_DIAG) && defined(FT_EXP_FWD_ONLY_V)    // ... waves 0-3 idle
        if (w < 4) {} else
#endif
        if (w < 4) {
            STAMPW(0);
            f32x4 none[2];                   // (no pending pair in front of a chain's first layer)
            {
                PairOps<KS_E0> oG0;
                PairOps<KS_H> oG1;
                pair_ops_start<C, KS_E0>(oG0, T3, ST, arena + C::off_of(L_G0), rt, lane, 0);
                chain_layer<C, KS_E0, false>(oG0, T3, ST, arena + C::off_of(L_G0), T0, slope, rt, lane, s1, none, nullptr,
                                             LAMBDA { pair_ops_start<C, KS_H>(oG1, T0, ST, arena + C::off_of(L_G1), rt, lane, 0); });
                STAMPW(1);
                chain_layer<C, KS_H, true>(oG1, T0, ST, arena + C::off_of(L_G1), T2, slope, rt, lane, s2, s1[1], T0, LAMBDA {});
            }
            STAMPW(2);
            if constexpr (C::HAS_U) flag_raise(flagG + rt * ST, step_no);          // T0's rows rt may be overwritten (h2l)
            {
                const f32x4 acc = chain_head<C>(T2 + mul24(16 * rt + c, ST), arena + C::off_of(L_G2) + mul24(min(c, DL - 1), KS_H), q,
                                                s2[1], T2, slope, rt, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c < DL) sG[mul24(16 * rt + 4 * q, SS) + r * SS + c] = acc[r];
            }
            STAMPW(3);
            flag_wait(flagV + rt * ST, step_no, flagE);
            STAMPW(4);

            // reparameterisation z = g + eps * sqrt(v) (nets.py:301-304), rows of this tile: lane = (row, a quarter of the
            // SS columns); all loads first, v_sqrt_f32 (1 ulp)
            {
                constexpr int QC = SS / 4;
                static_assert(SS % 4 == 0, "row-tensor stride is a multiple of 4");
                const int ln = opaque_lane();
                const int b = 16 * rt + (ln >> 2), k0 = (ln & 3) * QC, i0 = b * SS + k0;
                float ev[QC], vv[QC], gv[QC];
#pragma unroll
                for (int j = 0; j < QC; ++j) { ev[j] = sZ[i0 + j]; vv[j] = sV[i0 + j]; gv[j] = sG[i0 + j]; }
#pragma unroll
                for (int j = 0; j < QC; ++j) {
                    const int k = k0 + j;
                    float zz = k == DL ? 1.f : 0.f;
                    if (k < DL) zz = ev[j] * __builtin_amdgcn_sqrtf(vv[j]) + gv[j];
                    ev[j] = zz;
                    sZ[i0 + j] = zz;
                }
                if (A.z_out && s == A.n_steps - 1) {
#pragma unroll
                    for (int j = 0; j < QC; ++j) if (k0 + j < DL) A.z_out[((size_t)model * FT_B + b) * DL + k0 + j] = ev[j];
                }
                if (A.z_steps) {
#pragma unroll
                    for (int j = 0; j < QC; ++j)
                        if (k0 + j < DL) A.z_steps[(((size_t)model * A.n_steps_total + A.step_begin + s) * FT_B + b) * DL + k0 + j] = ev[j];
                }
            }
            flag_raise(flagZ + rt * ST, step_no);
            STAMPW(5);
            f32x4 kF1[2][2];
            {
                f32x4 kF0[2][2];
                PairOps<KS_F0> oF0;
                PairOps<KS_H> oF1;
                pair_ops_start<C, KS_F0>(oF0, sZ, SS, arena + C::off_of(L_F0), rt, lane, 0);
                chain_layer<C, KS_F0, false>(oF0, sZ, SS, arena + C::off_of(L_F0), T2, slope, rt, lane, kF0, none, nullptr,       // h1f
                                             LAMBDA { pair_ops_start<C, KS_H>(oF1, T2, ST, arena + C::off_of(L_F1), rt, lane, 0); });
                STAMPW(6);
                chain_layer<C, KS_H, true>(oF1, T2, ST, arena + C::off_of(L_F1), T4, slope, rt, lane, kF1, kF0[1], T2, LAMBDA {});  // h2f
                STAMPW(7);
            }
            // decoder head f (with h2f's pending columns), log p(x|z) and dLoss/df -> sO
            const f32x4 acc = chain_head<C>(T4 + mul24(16 * rt + c, ST), arena + C::off_of(L_F2) + mul24(min(c, DD - 1), KS_H), q,
                                            kF1[1], T4, slope, rt, lane);
            const float inv_dv = 1.f / A.decoder_var, logdv = logf(A.decoder_var);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * q + r;
                float dout = 0.f;
                if (c < DD) {
                    const float df = T3[row * ST + c] - acc[r];             // x - f
                    lsum += A.a * (-0.5f * (LOG2PI_F + logdv + df * df * inv_dv));
                    dout = -A.a * wB * df * inv_dv;
                }
                if (c < SS) sO[row * SS + c] = dout;
            }
            STAMPW(8);
            if constexpr (C::HAS_U) {          // take over the partner's h1l (same register layout as a stash)
                flag_wait(flagL + rt * ST, step_no, flagE);
                const int ln = opaque_lane();
                const float* base = T1 + mul24(16 * rt + 4 * (ln >> 4), ST);
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s3[p][j][r] = base[r * ST + min(32 * p + 16 * j + (ln & 15), ST - 1)];
            }
        } else {
            STAMPW(0);
            // The variance chain is what the partner's reparameterisation waits for, and with two 2-accumulator MFMA streams
            // on one SIMD the older wave (the g chain) would take the pipe until its stream ends: priority for this chain.
            __builtin_amdgcn_s_setprio(1);
            f32x4 none[2];
            {
                PairOps<KS_E0> oV0;
                PairOps<KS_H> oV1;
                pair_ops_start<C, KS_E0>(oV0, T3, ST, arena + C::off_of(L_V0), rt, lane, 0);
                chain_layer<C, KS_E0, false>(oV0, T3, ST, arena + C::off_of(L_V0), T1, slope, rt, lane, s1, none, nullptr,
                                             LAMBDA { pair_ops_start<C, KS_H>(oV1, T1, ST, arena + C::off_of(L_V1), rt, lane, 0); });
                STAMPW(1);
                chain_layer<C, KS_H, true>(oV1, T1, ST, arena + C::off_of(L_V1), T4, slope, rt, lane, s2, s1[1], T1, LAMBDA {});
            }
            STAMPW(2);
            {
                const f32x4 acc = chain_head<C>(T4 + mul24(16 * rt + c, ST), arena + C::off_of(L_V2) + mul24(min(c, DL - 1), KS_H), q,
                                                s2[1], T4, slope, rt, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c < DL) sV[mul24(16 * rt + 4 * q, SS) + r * SS + c] = fexp(acc[r]);
            }
            if (!(A.fault & 1)) flag_raise(flagV + rt * ST, step_no);      // (fault injection: tests/test_gpu_fused.py)
            __builtin_amdgcn_s_setprio(0);
            STAMPW(3);
            if constexpr (C::HAS_U) {
                f32x4 kL0[2][2];
                PairOps<KS_L0> oL0;
                PairOps<KS_H> oL1;
                pair_ops_start<C, KS_L0>(oL0, T3 + C::UOFF, ST, arena + C::off_of(L_L0), rt, lane, 0);
                chain_layer<C, KS_L0, false>(oL0, T3 + C::UOFF, ST, arena + C::off_of(L_L0), T1, slope, rt, lane, kL0, none, nullptr,
                                             LAMBDA { pair_ops_start<C, KS_H>(oL1, T1, ST, arena + C::off_of(L_L1), rt, lane, 0); });
                STAMPW(4);
                flag_wait(flagG + rt * ST, step_no, flagE);                  // T0's rows rt are free (the g chain has read h1g)
                chain_layer<C, KS_H, true>(oL1, T1, ST, arena + C::off_of(L_L1), T0, slope, rt, lane, s3, kL0[1], T1, LAMBDA {});
                flag_raise(flagL + rt * ST, step_no);                        // h1l (with its pending columns) is in T1
                STAMPW(5);
                const f32x4 acc = chain_head<C>(T0 + mul24(16 * rt + c, ST), arena + C::off_of(L_L2) + mul24(min(c, DL - 1), KS_H), q,
                                                s3[1], T0, slope, rt, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c < DL) sL[mul24(16 * rt + 4 * q, SS) + r * SS + c] = fexp(acc[r]);
                STAMPW(6);
            }
            flag_wait(flagZ + rt * ST, step_no, flagE);
            STAMPW(7);
            // latent log-densities log q(z|x,u), log p(z|u) (nets.py:306-314), rows of this tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // own sV / sL rows
            for (int e = lane; e < 16 * DL; e += 64) {
                const int b = 16 * rt + e / DL, k = e % DL, i = b * SS + k;
                const float zz = sZ[i], vv = sV[i], dq = zz - sG[i];
                const float lqc = -0.5f * (LOG2PI_F + flog(vv) + dq * dq * __builtin_amdgcn_rcpf(vv));
                float lp = -0.5f * (LOG2PI_F + zz * zz);
                if constexpr (C::HAS_U) { const float ll = sL[i]; lp = -0.5f * (LOG2PI_F + flog(ll) + zz * zz * __builtin_amdgcn_rcpf(ll)); }
                lsum += A.d * lp - A.b * lqc;
            }
        }
        BAR(8);

        // ---- BF3: d2f -> T1; decoder-head gradient -> T0 + STG_F2 (waves 0-3)
        dgrad_hidden<C, SS, DD, KS_H, H>(sO, SS, arena + C::off_of(L_F2), T4, T1, slope, w);
        if (w < 4) wgrad_tile<C, DD, KS_H, H>(sO, SS, T4, ST, T0 + C::STG_F2, 0, w);
        BAR(9);
        // ---- BF2: d1f -> T4; layer-2 gradient -> T0; Adam(F2)
        dgrad_wgrad_hidden<C, true>(T1, arena + C::off_of(L_F1), T2, T4, T0, slope, w);  // guarded: the tile's tail holds the F2 stage
        ADAM(L_F2, T0 + C::STG_F2);
        BAR(10);
        // ---- BF1: dLoss/dz (waves 0-3) and, by the lanes that hold it, the head deltas in place (SURVEY §8 a-bwd): sG <- dg,
        //           sV <- dlogv, sL <- dlogl (rows of the wave's own tile: nothing else reads them in this phase); h2g -> T2;
        //           layer-1 gradient -> stF0 (waves 4-7); Adam(F1)
        stagger(w, LAMBDA {
            if (w < 4) {
                // operands of the element-wise part first: they do not depend on the product
                float zz[4], vv[4], gg[4], ll[4];
                const int e0 = mul24(16 * rt + 4 * q, SS) + min(c, SS - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    zz[r] = sZ[e0 + r * SS]; vv[r] = sV[e0 + r * SS]; gg[r] = sG[e0 + r * SS];
                    ll[r] = C::HAS_U ? sL[e0 + r * SS] : 1.f;
                }
                const f32x4 acc = mm_dgrad1<KS_H, H, KS_F0>(T4 + (16 * rt + c) * ST, arena + C::off_of(L_F0) + min(c, KS_F0 - 1), q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dq = zz[r] - gg[r];
                    const float rr = dq * __builtin_amdgcn_rcpf(vv[r]);
                    const float zl = C::HAS_U ? zz[r] * __builtin_amdgcn_rcpf(ll[r]) : zz[r];
                    const float Gz = acc[r] - wB * (A.b * rr - A.d * zl);
                    if (c < DL) {
                        sG[e0 + r * SS] = wB * A.b * rr + Gz;
                        sV[e0 + r * SS] = -wB * A.b * 0.5f * (1.f - dq * rr) + Gz * 0.5f * dq;
                        if constexpr (C::HAS_U) sL[e0 + r * SS] = wB * A.d * 0.5f * (1.f - zz[r] * zl);
                    }
                }
                restore_rows<C>(s2, T2, rt);
            } else {
                wgrad_tile<C, H, KS_F0, DL>(T4, ST, sZ, SS, stF0, w - 4, 0);
            }
        }, LAMBDA { ADAM(L_F1, T0); });
        {   // loss reduction (deterministic): wave shuffle -> 8 partials -> thread 0 (next phase)
            float v = lsum;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) red[w] = v;
        }
        BAR(11);

        // ---- G3: d2g -> T0; head gradient -> sZ (z is dead); loss out; Adam(F0); h1g -> T4
        if (tid == 0) {
            const float t = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            A.loss_out[(size_t)model * A.n_steps_total + A.step_begin + s] = *(lds_flag_t*)flagE ? __builtin_nanf("") : -t * wB;
        }
        stagger(w, LAMBDA {
                    dgrad_hidden<C, SS, DL, KS_H, H>(sG, SS, arena + C::off_of(L_G2), T2, T0, slope, w);
                    if (w < 4) wgrad_tile<C, DL, KS_H, H>(sG, SS, T2, ST, sZ, 0, w);
                }, LAMBDA { ADAM(L_F0, stF0); if (w < 4) restore_rows<C>(s1, T4, rt); });
        BAR(13);
        // ---- G2: d1g -> T2; layer-2 gradient -> T1; Adam(G2)
#ifdef FT_PROF
        dgrad_wgrad_hidden<C, false>(T0, arena + C::off_of(L_G1), T4, T2, T1, slope, w, s);
        STAMPB(8);
        ADAM(L_G2, sZ);
        STAMPB(9);
#else
        dgrad_wgrad_hidden<C, false>(T0, arena + C::off_of(L_G1), T4, T2, T1, slope, w);
        ADAM(L_G2, sZ);
#endif
        BAR(14);
        // ---- G1: layer-1 gradient -> T0; Adam(G1); h2v -> T4
        WGRAD_ADAM(H, KS_E0, C::KIN, T2, ST, T3, ST, T0, L_G1, T1);
        if (w >= 4) restore_rows<C>(s2, T4, rt);
        BAR(15);

        // ---- V3: d2v -> T1; head gradient -> sZ; Adam(G0); h1v -> T2
        stagger(w, LAMBDA {
                    dgrad_hidden<C, SS, DL, KS_H, H>(sV, SS, arena + C::off_of(L_V2), T4, T1, slope, w);
                    if (w < 4) wgrad_tile<C, DL, KS_H, H>(sV, SS, T4, ST, sZ, 0, w);
                }, LAMBDA { ADAM(L_G0, T0); if (w >= 4) restore_rows<C>(s1, T2, rt); });
        BAR(16);
        // ---- V2: d1v -> T4; layer-2 gradient -> T0; Adam(V2)
        dgrad_wgrad_hidden<C, false>(T1, arena + C::off_of(L_V1), T2, T4, T0, slope, w);
        ADAM(L_V2, sZ);
        BAR(17);
        // ---- V1: layer-1 gradient -> T1; Adam(V1); h2l -> T2
        WGRAD_ADAM(H, KS_E0, C::KIN, T4, ST, T3, ST, T1, L_V1, T0);
        if constexpr (C::HAS_U) { if (w >= 4) restore_rows<C>(s3, T2, rt); }
        BAR(18);

        if constexpr (C::HAS_U) {
            // ---- L3: d2l -> T0; head gradient -> sZ; Adam(V0); h1l -> T4
            stagger(w, LAMBDA {
                        dgrad_hidden<C, SS, DL, KS_H, H>(sL, SS, arena + C::off_of(L_L2), T2, T0, slope, w);
                        if (w < 4) wgrad_tile<C, DL, KS_H, H>(sL, SS, T2, ST, sZ, 0, w);
                    }, LAMBDA { ADAM(L_V0, T1); if (w < 4) restore_rows<C>(s3, T4, rt); });
            BAR(19);
            // ---- L2: d1l -> T2; layer-2 gradient -> T1; Adam(L2)
            dgrad_wgrad_hidden<C, false>(T0, arena + C::off_of(L_L1), T4, T2, T1, slope, w);
            ADAM(L_L2, sZ);
            BAR(20);
            // ---- L1: layer-1 gradient -> T0; Adam(L1)
            WGRAD_ADAM(H, KS_L0, M, T2, ST, T3 + C::UOFF, ST, T0, L_L1, T1);
            BAR(21);
        }
    }
    // Adam of the last step's last layer
    if (A.n_steps > 0) {
        if constexpr (C::HAS_U) ADAM(L_L0, T0); else ADAM(L_V0, T1);
    }
#undef ADAM
    __syncthreads();

    // ---- write state back
    if constexpr (C::HAS_U) {
        state_store<C, L_L0>(gP, gM, gV, arena, am, av);
        state_store<C, L_L1>(gP, gM, gV, arena, am, av);
        state_store<C, L_L2>(gP, gM, gV, arena, am, av);
    }
    state_store<C, L_F0>(gP, gM, gV, arena, am, av);
    state_store<C, L_F1>(gP, gM, gV, arena, am, av);
    state_store<C, L_F2>(gP, gM, gV, arena, am, av);
    state_store<C, L_G0>(gP, gM, gV, arena, am, av);
    state_store<C, L_G1>(gP, gM, gV, arena, am, av);
    state_store<C, L_G2>(gP, gM, gV, arena, am, av);
    state_store<C, L_V0>(gP, gM, gV, arena, am, av);
    state_store<C, L_V1>(gP, gM, gV, arena, am, av);
    state_store<C, L_V2>(gP, gM, gV, arena, am, av);
}

// ------------------------------------------------------------------------------ host side
template <class C>
static hipError_t launch_cfg(const FusedArgs& A, int n_models, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    static SmemCache cur = {};      // per device: a process may drive this entry point on several GPUs
    hipError_t e = set_smem((const void*)k_train_steps<C>, (int)C::LDS_BYTES, cur);
    if (e != hipSuccess) return e;
    // (hipExtLaunchKernelGGL would hand the events the dispatch's own time stamps, but it costs the launch ~30 us of host
    //  time -- measured: 721 against 689 us of wall per 20-step launch; the records sit directly around the launch instead)
    if (ev_start && (e = hipEventRecord(ev_start, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_train_steps<C>, dim3(n_models), dim3(FT_THREADS), C::LDS_BYTES, st, A);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return ev_stop ? hipEventRecord(ev_stop, st) : hipSuccess;
}

#ifndef FT_SHAPES
// d = 4..10 with 40 segments (the LDS holds no more: d = 11 overflows) and their VAE counterparts
#define FT_SHAPES(X)                                                                                                  \
    X(10, 10, 40, 50) X(10, 10, 0, 50) X(5, 5, 40, 50) X(5, 5, 0, 50) X(4, 4, 40, 50) X(4, 4, 0, 50) X(6, 6, 40, 50)        \
    X(6, 6, 0, 50) X(7, 7, 40, 50) X(7, 7, 0, 50) X(8, 8, 40, 50) X(8, 8, 0, 50) X(9, 9, 40, 50) X(9, 9, 0, 50)
#endif

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

hipError_t launch_train_steps(int dd, int dl, int m, int h, const FusedArgs& A, int n_models, hipStream_t st, hipEvent_t ev_start,
                              hipEvent_t ev_stop) {
#define X(a, b, c, d) if (dd == a && dl == b && m == c && h == d) return launch_cfg<FCfg<a, b, c, d>>(A, n_models, st, ev_start, ev_stop);
    FT_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
