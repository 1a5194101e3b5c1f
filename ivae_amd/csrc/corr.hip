// Pearson cross-correlation block for the MCC metric, float64 arithmetic on fp32 inputs.
//
// Replaces np.corrcoef(x, y, rowvar=False)[:d, d:] (metrics/mcc.py:328; torch spec at
// mcc.py:174-253).  ONE pass over the rows: moments of the SHIFTED data a = x - kx, b = y - ky
// (sums, squares, cross products, all float64), where the shift k is the mean of a fixed sample of
// up to 256 rows spread over the array -- every workgroup forms the same k from the same rows in the
// same order -- and the centred moments follow as S_ab - S_a S_b / n.  The shift only has to be
// NEAR the mean: with k the mean of a sample S of the data, n (k - mean)^2 <= (n / |S|) * M2, so
// the correction term never exceeds n / |S| times the quantity it corrects and float64 keeps
// >= 12 digits at n = 10^6 whatever the data (offsets, trends, outliers; measured: tools/fuzz_corr.py);
// a constant column is shifted to exact zeros (the sample sum of equal fp32 values is exact), so its
// variance is exactly 0 and the correlation nan as in numpy.  The cross moments are the contraction
// A^T B over the rows: v_mfma_f64_16x16x4_f64, operands loaded from HBM straight into fragment
// layout (lane (c, q): row r0 + q, column 16 t + c), converted and shifted in registers; sums and
// squares ride along on the vector unit.  Waves walk 4-row groups grid-strided, per-wave partials
// meet in LDS and per-block partials in the workspace, both summed in a fixed order (bitwise
// reproducible for a given n, d).  The Hungarian step (mcc.py:334) stays on the host.
// (Until round 3 this was two passes -- column means, then centred moments: 130 us at n = 10^6,
// d = 20, each pass at 2.4 TB/s; the data are now read once.)
#include <hip/hip_runtime.h>
#include "kernels.h"

#define CORR_BLOCKS 256
#define CORR_ROWS 64

// The size the two-pass version needed (means, per-block sums, per-block centred moments).  The one-pass kernels use its
// first CORR_BLOCKS * (d*d + 4d) doubles (2d less than this); the figure is kept: callers have sized buffers with it and the
// built-in scratch of ivae_corr_static is declared from it.
size_t corr_ws_bytes(int64_t n, int d) {
    (void)n;
    return sizeof(double) * ((size_t)2 * d + (size_t)CORR_BLOCKS * 2 * d + (size_t)CORR_BLOCKS * ((size_t)d * d + 2 * d));
}

// np.corrcoef clips to [-1, 1] with np.clip, which PROPAGATES nan: a column of zero variance gives 0 / 0 there and the
// block is not finite -- the reference's Hungarian step then raises and the runner scores the batch 0
// (runners/ivae.py:85-88).  fmin / fmax would turn that nan into -1, i.e. into a perfect |correlation|.
__device__ __forceinline__ double clip_unit(double c) { return c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c); }

// Sums of the nblk <= CORR_BLOCKS per-block partials of up to 64 consecutive quantities [q0, q0 + cnt) by a 1024-thread
// workgroup: wave w takes blocks 16 w .. 16 w + 15 (16 loads in flight, 512 contiguous bytes each, summed in block order),
// lane l quantity q0 + l; the sixteen wave sums meet in LDS and are added in wave order.  Fixed order: deterministic.
// dst[l] = sum * scale for l < cnt.  (One thread per quantity with eight loads in flight, in ONE workgroup for all
// quantities, took 57 us at d = 20 -- as long as a streaming pass; the final kernels are therefore grids over the
// quantities.)  red: (16, 64) doubles of LDS.  Ends with a barrier.
__device__ __forceinline__ void reduce_partials64(const double* __restrict__ p, int nout, int nblk, int q0, int cnt, double scale,
                                                  double* red, double* dst) {
    static_assert(CORR_BLOCKS == 256, "16 waves x 16 blocks");
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = q0 + min(l, cnt - 1);
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int blk = min(w * 16 + e, nblk - 1);
        v[e] = p[(size_t)blk * nout + q];
    }
    double t = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) t += w * 16 + e < nblk ? v[e] : 0.0;
    red[w * 64 + l] = t;
    __syncthreads();
    if (w == 0 && l < cnt) {
        double tot = 0.0;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) tot += red[ww * 64 + l];
        dst[l] = tot * scale;
    }
    __syncthreads();
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// NT = ceil(d / 16) column tiles of x and of y; WAVES per workgroup (NT <= 2: 16, four per SIMD; else 8: the NT*NT
// accumulator tiles take 8 registers each).  MFMA slots: A[i = c][k = q] = a[row q][column 16 ta + c],
// B[k = q][j = c] = b[row q][column 16 tb + c]; D[i = 4 r + q][j = c] in register pair r (probed on gfx950: the fp64
// shape interleaves rows where the fp32 16x16x4 shape blocks them).
// Per-block output (nout = d*d + 4d doubles): S_ab (d*d) | S_aa (d) | S_bb (d) | S_a (d) | S_b (d).
#define CORR_SAMPLE 256
template <int NT, int WAVES, int U>
__global__ __launch_bounds__(64 * WAVES) void k_corr_moments(const float* __restrict__ x, const float* __restrict__ y,
                                                            int64_t n, int d, double* __restrict__ pmom) {
    extern __shared__ __attribute__((aligned(16))) double part[];    // (nout) the block's partial moments
    __shared__ double red[64 * WAVES];
    __shared__ double shift[128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int nout = d * d + 4 * d;
    {   // the shift: mean of S sample rows r_s = s * (n / S), S a power of two (dividing by it is exact) <= n, <= 256, and
        // at most 16 rows per thread (all loads in flight at once).  Thread (column c2, sub) sums rows sub, sub + nsub, ...
        const int c2 = threadIdx.x % (2 * d), sub = threadIdx.x / (2 * d), nsub = 64 * WAVES / (2 * d);
        int S = 1;
        while (2 * S <= CORR_SAMPLE && 2 * S <= n && 2 * S <= 16 * nsub) S *= 2;
        const int64_t rstride = n / S;
        const float* src = c2 < d ? x + c2 : y + (c2 - d);
        float vals[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int sidx = sub + e * nsub;
            vals[e] = sub < nsub && sidx < S ? src[(int64_t)sidx * rstride * d] : 0.f;
        }
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) t += (double)vals[e];
        red[threadIdx.x] = sub < nsub ? t : 0.0;
        __syncthreads();
        if ((int)threadIdx.x < 2 * d) {
            double tot = 0.0;
            for (int k = 0; k < nsub; ++k) tot += red[k * 2 * d + threadIdx.x];
            shift[threadIdx.x] = tot / (double)S;
        }
        __syncthreads();
    }
    f64x4 acc[NT][NT];
    double kx[NT], ky[NT], sxx[NT], syy[NT], sx[NT], sy[NT];
    int col[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        col[t] = min(16 * t + c, d - 1);                     // (a clamped column is loaded and zeroed below)
        kx[t] = shift[col[t]]; ky[t] = shift[d + col[t]];
        sxx[t] = 0.0; syy[t] = 0.0; sx[t] = 0.0; sy[t] = 0.0;
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    // U 4-row groups per batch of loads, two batches in flight per wave (the next batch is requested before the current one
    // is consumed: a wave that waits for its loads, then computes, then asks again leaves the memory idle while it computes
    // and the matrix pipe idle while it waits -- 4 waves per SIMD do not cover that)
    // (U = 2 where 4 would spill: the NT * NT accumulator tiles take 8 registers each)
    const int64_t ngroups = (n + 3) / 4;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    auto fetch = [&](int64_t g0, float (&xv)[U][NT], float (&yv)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = min((g0 + u * stride) * 4 + q, n - 1);     // (clamped: zeroed in consume)
#pragma unroll
            for (int t = 0; t < NT; ++t) { xv[u][t] = x[row * d + col[t]]; yv[u][t] = y[row * d + col[t]]; }
        }
    };
    auto consume = [&](int64_t g0, const float (&xv)[U][NT], const float (&yv)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (g0 + u * stride) * 4 + q;
            const bool rok = g0 + u * stride < ngroups && row < n;
            double a[NT], b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool ok = rok && 16 * t + c < d;
                a[t] = ok ? (double)xv[u][t] - kx[t] : 0.0;
                b[t] = ok ? (double)yv[u][t] - ky[t] : 0.0;
                sx[t] += a[t];
                sy[t] += b[t];
                sxx[t] += a[t] * a[t];
                syy[t] += b[t] * b[t];
            }
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    };
    // The walk is bound by vector-instruction issue, not by memory (timing-only ablations: every vector instruction taken
    // out of the loop shortens it by its issue time, fp64 ones by 8 cycles each), so the steady state spends none on
    // addresses or masks: batches whose rows all exist are loaded through a buffer descriptor -- per-lane byte offsets
    // fixed for the whole walk, the batch position in the scalar offset -- and consumed unmasked (a lane whose column is
    // clamped computes on a duplicate column; nothing it produces is written).  Two batches per trip, the next one
    // requested before the current one is consumed.  Arrays of 4 GB and more, and the last (partial) batches, take the
    // masked 64-bit path below.
    int64_t g0 = (int64_t)blockIdx.x * WAVES + w;
    const int64_t step = U * stride;
    if ((uint64_t)n * (uint64_t)d * 4u < (1ull << 32)) {
        const int64_t nfull = n / 4;                         // groups whose four rows all exist
        // batches of this wave that are complete: the last group of batch b is g0 + (b U + U - 1) stride
        const int64_t last0 = g0 + (int64_t)(U - 1) * stride;
        const int full = last0 < nfull ? (int)((nfull - 1 - last0) / step) + 1 : 0;
        const int trips = __builtin_amdgcn_readfirstlane(full / 2);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((uint64_t)n * d * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)((uint64_t)n * d * 4u), 0x00020000);
        int vo[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) vo[t] = (int)(((g0 * 4 + q) * d + col[t]) * 4);
        const int ustep = __builtin_amdgcn_readfirstlane((int)(stride * 16 * d));      // bytes from a group to the wave's next
        auto bfetch = [&](int so, float (&xv)[U][NT], float (&yv)[U][NT]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    xv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[t], so + u * ustep, 0));
                    yv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, vo[t], so + u * ustep, 0));
                }
        };
        auto bconsume = [&](const float (&xv)[U][NT], const float (&yv)[U][NT]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double a[NT], b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a[t] = (double)xv[u][t] - kx[t];
                    b[t] = (double)yv[u][t] - ky[t];
                    sx[t] += a[t];
                    sy[t] += b[t];
                    sxx[t] += a[t] * a[t];
                    syy[t] += b[t] * b[t];
                }
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        };
        if (trips > 0) {
            float xa[U][NT], ya[U][NT], xb[U][NT], yb[U][NT];
            const int bstep = U * ustep;
            const int so_last = (2 * trips - 1) * bstep;      // the last complete batch of this wave
            int so = 0;
            bfetch(so, xa, ya);
            for (int it = 0; it < trips; ++it) {
                bfetch(so + bstep, xb, yb);
                __builtin_amdgcn_sched_barrier(0);
                bconsume(xa, ya);
                __builtin_amdgcn_sched_barrier(0);
                // (the request of the last trip has no batch to fetch: it reads the last one again, unused; no branch here,
                //  so that the wait counts stay counted.  A request past the array would be harmless too -- on gfx950 the
                //  scalar offset is part of a raw buffer load's range check, tools/ubench/soffset_range.hip -- but is not made)
                bfetch(min(so + 2 * bstep, so_last), xa, ya);
                __builtin_amdgcn_sched_barrier(0);
                bconsume(xb, yb);
                __builtin_amdgcn_sched_barrier(0);
                so += 2 * bstep;
            }
            g0 += (int64_t)trips * 2 * step;
        }
    }
    {
        float xa[U][NT], ya[U][NT];
        for (; g0 < ngroups; g0 += step) {
            fetch(g0, xa, ya);
            consume(g0, xa, ya);
        }
    }
    // wave partials -> one LDS block, wave after wave (a fixed order: deterministic).  Cross moments from the accumulators;
    // sums and squares: the four q lanes of a column summed in q order first
    auto over_q = [&](double v) {
        const double v1 = __shfl(v, c + 16), v2 = __shfl(v, c + 32), v3 = __shfl(v, c + 48);
        return ((v + v1) + v2) + v3;
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) { sxx[t] = over_q(sxx[t]); syy[t] = over_q(syy[t]); sx[t] = over_q(sx[t]); sy[t] = over_q(sy[t]); }
    for (int ww = 0; ww < WAVES; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ia = 16 * ta + 4 * r + q, jb = 16 * tb + c;
                        if (ia < d && jb < d) part[ia * d + jb] = (ww ? part[ia * d + jb] : 0.0) + acc[ta][tb][r];
                    }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (q == 0 && 16 * t + c < d) {
                    const int o = d * d + 16 * t + c;
                    part[o] = (ww ? part[o] : 0.0) + sxx[t];
                    part[o + d] = (ww ? part[o + d] : 0.0) + syy[t];
                    part[o + 2 * d] = (ww ? part[o + 2 * d] : 0.0) + sx[t];
                    part[o + 3 * d] = (ww ? part[o + 3 * d] : 0.0) + sy[t];
                }
        }
        __syncthreads();
    }
    for (int o = threadIdx.x; o < nout; o += 64 * WAVES) pmom[(size_t)blockIdx.x * nout + o] = part[o];
}

// Workgroup b: correlations of the cross moments o in [64 b, 64 b + 64); it reduces those and (again in every workgroup:
// 4d more quantities) the squares and sums that centre and normalise them.
__global__ __launch_bounds__(1024) void k_corr_final(const double* __restrict__ pmom, int64_t n, int d, int nblk,
                                                     double* __restrict__ cc) {
    __shared__ double red[16 * 64];
    __shared__ double cross[64];
    __shared__ double lin[256];               // S_aa (d) | S_bb (d) | S_a (d) | S_b (d)
    const int nout = d * d + 4 * d, o0 = blockIdx.x * 64, cnt = min(64, d * d - o0);
    reduce_partials64(pmom, nout, nblk, o0, cnt, 1.0, red, cross);
    for (int a0 = 0; a0 < 4 * d; a0 += 64) reduce_partials64(pmom, nout, nblk, d * d + a0, min(64, 4 * d - a0), 1.0, red, lin + a0);
    if ((int)threadIdx.x < cnt) {
        const int o = o0 + threadIdx.x, a = o / d, b = o - a * d;
        const double inv_n = 1.0 / (double)n, scale = 1.0 / (double)(n - 1);     // np.cov normalisation (cancels in the ratio)
        const double sa = lin[2 * d + a], sb = lin[3 * d + b];
        const double cov = (cross[threadIdx.x] - sa * sb * inv_n) * scale;
        const double va = (lin[a] - sa * sa * inv_n) * scale, vb = (lin[d + b] - sb * sb * inv_n) * scale;
        double c = cov / sqrt(va);
        c = c / sqrt(vb);
        cc[o] = clip_unit(c);
    }
}

// Many small problems in ONE launch: problem b owns rows [b*n, (b+1)*n) of x and y (the per-step batch scores of a
// training epoch, runners/ivae.py:86, wrappers.py:75: n = batch size).  ONE WAVE per problem, four problems per workgroup,
// no LDS and no barrier: the same one-pass shifted moments as above with the shift taken from the problem's first (up to)
// four rows -- exact zeros for a constant column, and n / 4 * 1e-16 bounds the relative error of the correction -- the
// cross moments on v_mfma_f64_16x16x4_f64 straight from fragment-layout loads, sums and squares per lane and across the
// four row lanes at the end, the correlation block written from the accumulators.  (Until round 3 a 256-thread workgroup
// per problem staged float64 tiles in LDS in two passes: 290 us for 40 000 problems of 64 x 5 -- 0.35 TB/s of input.)
template <int NT>
__global__ __launch_bounds__(256) void k_corr_batched(const float* __restrict__ x, const float* __restrict__ y, int64_t n_blocks,
                                                      int64_t n, int d, double* __restrict__ cc) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, q = lane >> 4;
    const int64_t prob = (int64_t)blockIdx.x * 4 + w;
    if (prob >= n_blocks) return;                            // (whole waves leave; nothing below synchronises)
    const float* xb = x + (size_t)prob * n * d;
    const float* yb = y + (size_t)prob * n * d;
    auto all_q = [&](double v) {                             // the four row lanes of a column, in q order, in every lane
        const double v0 = __shfl(v, c), v1 = __shfl(v, c + 16), v2 = __shfl(v, c + 32), v3 = __shfl(v, c + 48);
        return ((v0 + v1) + v2) + v3;
    };
    const int S = n >= 4 ? 4 : (n >= 2 ? 2 : 1);             // a power of two: the division is exact
    f64x4 acc[NT][NT];
    double kx[NT], ky[NT], sxx[NT], syy[NT], sx[NT], sy[NT];
    int col[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        col[t] = min(16 * t + c, d - 1);                     // (a clamped column computes on a duplicate; never written)
        const double px = q < S ? (double)xb[(int64_t)q * d + col[t]] : 0.0;
        const double py = q < S ? (double)yb[(int64_t)q * d + col[t]] : 0.0;
        kx[t] = all_q(px) / (double)S;
        ky[t] = all_q(py) / (double)S;
        sxx[t] = 0.0; syy[t] = 0.0; sx[t] = 0.0; sy[t] = 0.0;
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    constexpr int U = NT == 1 ? 8 : (NT == 2 ? 4 : 2);       // 4-row groups per batch of loads: 2 NT U loads in flight
    const int64_t ngroups = (n + 3) / 4;
    int64_t g0 = 0;
    if ((uint64_t)n * (uint64_t)d * 4u < (1ull << 32)) {
        // batches of complete groups: no address arithmetic (per-lane byte offsets fixed, the batch in the scalar offset of a
        // buffer load over this problem's rows) and no masks -- the walk is bound by vector-instruction issue (see above)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((uint64_t)n * d * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (int)((uint64_t)n * d * 4u), 0x00020000);
        int vo[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) vo[t] = (q * d + col[t]) * 4;
        const int gbytes = 16 * d;                           // a 4-row group
        const int nbatch = (int)((n / 4) / U);
        for (int bi = 0; bi < nbatch; ++bi) {
            const int so = bi * U * gbytes;
            float xv[U][NT], yv[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    xv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[t], so + u * gbytes, 0));
                    yv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, vo[t], so + u * gbytes, 0));
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double a[NT], b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a[t] = (double)xv[u][t] - kx[t];
                    b[t] = (double)yv[u][t] - ky[t];
                    sx[t] += a[t];
                    sy[t] += b[t];
                    sxx[t] += a[t] * a[t];
                    syy[t] += b[t] * b[t];
                }
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        g0 = (int64_t)nbatch * U;
    }
    for (; g0 < ngroups; g0 += U) {                          // what is left: partial batches, masked
        float xv[U][NT], yv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = min((g0 + u) * 4 + q, n - 1);                  // (clamped: zeroed below)
#pragma unroll
            for (int t = 0; t < NT; ++t) { xv[u][t] = xb[row * d + col[t]]; yv[u][t] = yb[row * d + col[t]]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rok = (g0 + u) * 4 + q < n;
            double a[NT], b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a[t] = rok ? (double)xv[u][t] - kx[t] : 0.0;
                b[t] = rok ? (double)yv[u][t] - ky[t] : 0.0;
                sx[t] += a[t];
                sy[t] += b[t];
                sxx[t] += a[t] * a[t];
                syy[t] += b[t] * b[t];
            }
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    }
    const double inv_n = 1.0 / (double)n, scale = 1.0 / (double)(n - 1);       // np.cov normalisation (cancels in the ratio)
    double vx[NT], vy[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sx[t] = all_q(sx[t]); sy[t] = all_q(sy[t]);
        vx[t] = (all_q(sxx[t]) - sx[t] * sx[t] * inv_n) * scale;
        vy[t] = (all_q(syy[t]) - sy[t] * sy[t] * inv_n) * scale;
    }
    double* out = cc + (size_t)prob * d * d;
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ia = 16 * ta + 4 * r + q;                                // this lane's output row; its x column lives in lane 4 r + q
            const double sa = __shfl(sx[ta], 4 * r + q), va = __shfl(vx[ta], 4 * r + q);
#pragma unroll
            for (int tb = 0; tb < NT; ++tb) {
                const int jb = 16 * tb + c;
                double v = (acc[ta][tb][r] - sa * sy[tb] * inv_n) * scale / sqrt(va);
                v = v / sqrt(vy[tb]);
                if (ia < d && jb < d) out[ia * d + jb] = clip_unit(v);
            }
        }
}

hipError_t launch_corr_batched(const float* x, const float* y, int64_t n_blocks, int64_t n, int d, double* cc, hipStream_t st) {
    const dim3 grid((unsigned)((n_blocks + 3) / 4)), block(256);
    switch ((d + 15) / 16) {
        case 1: hipLaunchKernelGGL(k_corr_batched<1>, grid, block, 0, st, x, y, n_blocks, n, d, cc); break;
        case 2: hipLaunchKernelGGL(k_corr_batched<2>, grid, block, 0, st, x, y, n_blocks, n, d, cc); break;
        case 3: hipLaunchKernelGGL(k_corr_batched<3>, grid, block, 0, st, x, y, n_blocks, n, d, cc); break;
        case 4: hipLaunchKernelGGL(k_corr_batched<4>, grid, block, 0, st, x, y, n_blocks, n, d, cc); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_corr(const float* x, const float* y, int64_t n, int d, double* cc, double* ws, hipStream_t st) {
    double* pmom = ws;                      // (nblk, d*d + 4d): inside what corr_ws_bytes has always asked for
    int nblk = (int)((n + CORR_ROWS - 1) / CORR_ROWS);
    if (nblk > CORR_BLOCKS) nblk = CORR_BLOCKS;
    const size_t smem = (size_t)(d * d + 4 * d) * sizeof(double);          // d = 64: 34 816 B
    switch ((d + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_corr_moments<1, 16, 4>), dim3(nblk), dim3(1024), smem, st, x, y, n, d, pmom); break;
        case 2: hipLaunchKernelGGL((k_corr_moments<2, 16, 2>), dim3(nblk), dim3(1024), smem, st, x, y, n, d, pmom); break;
        case 3: hipLaunchKernelGGL((k_corr_moments<3, 8, 4>), dim3(nblk), dim3(512), smem, st, x, y, n, d, pmom); break;
        case 4: hipLaunchKernelGGL((k_corr_moments<4, 8, 2>), dim3(nblk), dim3(512), smem, st, x, y, n, d, pmom); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_corr_final, dim3((d * d + 63) / 64), dim3(1024), 0, st, pmom, n, d, nblk, cc);
    return hipGetLastError();
}

// ---- the same computation with the scratch inside the library: ivae_corr_static (callers that cannot allocate) ----
// `scratch` holds what ivae_corr's workspace argument holds (the per-block moments; sized as corr_ws_bytes asks)
// for d <= CORR_STATIC_MAX_D.  In host code the array's NAME is only a stand-in: the address the kernels get is
// the one hipGetSymbolAddress returns for the CURRENT device (every GPU holds its own copy), looked up once per device
// and kept in scratch_ptr[].  Nothing here must start at zero: every word a kernel reads was written by the kernel
// before it in the same call (blocks 0..nblk-1 only), so a call does not depend on what the last one left.
#define CORR_STATIC_DOUBLES (2 * CORR_STATIC_MAX_D + CORR_BLOCKS * 2 * CORR_STATIC_MAX_D + \
                             CORR_BLOCKS * (CORR_STATIC_MAX_D * CORR_STATIC_MAX_D + 2 * CORR_STATIC_MAX_D))
static __device__ __align__(256) double scratch[CORR_STATIC_DOUBLES];      // 3,605,120 B per GPU

static double* scratch_ptr[IVAE_MAX_DEVICES];       // per device; written once (racing host threads write the same value)

hipError_t launch_corr_static(const float* x, const float* y, int64_t n, int d, double* cc, hipStream_t st) {
    if (d > CORR_STATIC_MAX_D || corr_ws_bytes(n, d) > sizeof(double) * (size_t)CORR_STATIC_DOUBLES) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= IVAE_MAX_DEVICES) return hipErrorInvalidDevice;
    double* ptr = __atomic_load_n(&scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!ptr) {
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        __atomic_store_n(&scratch_ptr[dev], ptr, __ATOMIC_RELEASE);
    }
    return launch_corr(x, y, n, d, cc, ptr, st);
}
