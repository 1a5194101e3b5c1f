// Pearson cross-correlation block for the MCC metric, float64 arithmetic on fp32 inputs.
//
// Replaces np.corrcoef(x, y, rowvar=False)[:d, d:] (metrics/mcc.py:328; torch spec at
// mcc.py:174-253): two-pass centred moments.  Pass 1: column sums -> means.  Pass 2: the centred
// cross moments are the contraction Xc^T Yc over the rows: v_mfma_f64_16x16x4_f64, operands
// loaded from HBM straight into fragment layout (lane (c, q): row r0 + q, column 16 t + c),
// converted and centred in registers; the auto moments ride along on the vector unit.  Waves
// walk 4-row groups grid-strided, per-wave partials meet in LDS and per-block partials in the
// workspace, both summed in a fixed order (bitwise reproducible for a given n, d).  The
// Hungarian step (mcc.py:334) stays on the host.
#include <hip/hip_runtime.h>
#include "kernels.h"

#define CORR_BLOCKS 256
#define CORR_ROWS 64

size_t corr_ws_bytes(int64_t n, int d) {
    (void)n;
    return sizeof(double) * ((size_t)2 * d + (size_t)CORR_BLOCKS * 2 * d + (size_t)CORR_BLOCKS * ((size_t)d * d + 2 * d));
}

// np.corrcoef clips to [-1, 1] with np.clip, which PROPAGATES nan: a column of zero variance gives 0 / 0 there and the
// block is not finite -- the reference's Hungarian step then raises and the runner scores the batch 0
// (runners/ivae.py:85-88).  fmin / fmax would turn that nan into -1, i.e. into a perfect |correlation|.
__device__ __forceinline__ double clip_unit(double c) { return c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c); }

#define CORR_SUM_THREADS 1024
__global__ __launch_bounds__(CORR_SUM_THREADS) void k_corr_sums(const float* __restrict__ x, const float* __restrict__ y,
                                                                int64_t n, int d, double* __restrict__ psum) {
    // thread c < 2d owns a column; rows strided over blocks.  (d is small: <= 64.)  16 waves per CU: the walk is bound by
    // the bytes in flight
    __shared__ double red[CORR_SUM_THREADS];
    const int c = threadIdx.x % (2 * d);
    const int sub = threadIdx.x / (2 * d);
    const int nsub = CORR_SUM_THREADS / (2 * d);
    double acc = 0.0;
    if (sub < nsub) {
        const float* src = c < d ? x + c : y + (c - d);
        // eight rows' loads in flight per thread (a dependent one-load-per-iteration walk is bound by memory latency:
        // 0.97 ms at n = 10^6, d = 20); the sum keeps its row order
        const int64_t stride = (int64_t)gridDim.x * nsub;
        for (int64_t r = (int64_t)blockIdx.x * nsub + sub; r < n; r += 8 * stride) {
            float vals[8];
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) {
                const int64_t rr = r + u_ * stride;
                vals[u_] = rr < n ? src[rr * d] : 0.f;
            }
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) acc += (double)vals[u_];
        }
    }
    red[threadIdx.x] = sub < nsub ? acc : 0.0;
    __syncthreads();
    if (threadIdx.x < 2 * d) {
        double t = 0.0;
        for (int s = 0; s < nsub; ++s) t += red[s * 2 * d + threadIdx.x];
        psum[(size_t)blockIdx.x * 2 * d + threadIdx.x] = t;
    }
}

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

__global__ __launch_bounds__(1024) void k_corr_mean(const double* __restrict__ psum, int64_t n, int d, int nblk, double* __restrict__ mean) {
    __shared__ double red[16 * 64];
    const int q0 = blockIdx.x * 64;
    reduce_partials64(psum, 2 * d, nblk, q0, min(64, 2 * d - q0), 1.0 / (double)n, red, mean + q0);
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// NT = ceil(d / 16) column tiles of x and of y; WAVES per workgroup (NT <= 2: 16, four per SIMD; else 8: the NT*NT
// accumulator tiles take 8 registers each).  MFMA slots: A[i = c][k = q] = xc[row q][column 16 ta + c],
// B[k = q][j = c] = yc[row q][column 16 tb + c]; D[i = 4 r + q][j = c] in register pair r (probed on gfx950: the fp64
// shape interleaves rows where the fp32 16x16x4 shape blocks them).
template <int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_corr_moments(const float* __restrict__ x, const float* __restrict__ y,
                                                            int64_t n, int d, const double* __restrict__ mean,
                                                            double* __restrict__ pmom) {
    extern __shared__ __attribute__((aligned(16))) double part[];    // (nout) the block's partial moments
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int nout = d * d + 2 * d;
    f64x4 acc[NT][NT];
    double mx[NT], my[NT], sxx[NT], syy[NT];
    int col[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        col[t] = min(16 * t + c, d - 1);                     // (a clamped column is loaded and zeroed below)
        mx[t] = mean[col[t]]; my[t] = mean[d + col[t]];
        sxx[t] = 0.0; syy[t] = 0.0;
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    constexpr int U = 4;                                      // 4-row groups in flight per wave: 2 NT U loads
    const int64_t ngroups = (n + 3) / 4;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t g0 = (int64_t)blockIdx.x * WAVES + w; g0 < ngroups; g0 += U * stride) {
        float xv[U][NT], yv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = min((g0 + u * stride) * 4 + q, n - 1);     // (clamped: zeroed below)
#pragma unroll
            for (int t = 0; t < NT; ++t) { xv[u][t] = x[row * d + col[t]]; yv[u][t] = y[row * d + col[t]]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (g0 + u * stride) * 4 + q;
            const bool rok = g0 + u * stride < ngroups && row < n;
            double a[NT], b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool ok = rok && 16 * t + c < d;
                a[t] = ok ? (double)xv[u][t] - mx[t] : 0.0;
                b[t] = ok ? (double)yv[u][t] - my[t] : 0.0;
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
    // wave partials -> one LDS block, wave after wave (a fixed order: deterministic).  Cross moments from the accumulators;
    // auto moments: the four q lanes of a column summed in q order first
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double vx = sxx[t], vy = syy[t];
        const double x1 = __shfl(vx, c + 16), x2 = __shfl(vx, c + 32), x3 = __shfl(vx, c + 48);
        const double y1 = __shfl(vy, c + 16), y2 = __shfl(vy, c + 32), y3 = __shfl(vy, c + 48);
        sxx[t] = ((vx + x1) + x2) + x3;
        syy[t] = ((vy + y1) + y2) + y3;
    }
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
                    const int ox = d * d + 16 * t + c, oy = ox + d;
                    part[ox] = (ww ? part[ox] : 0.0) + sxx[t];
                    part[oy] = (ww ? part[oy] : 0.0) + syy[t];
                }
        }
        __syncthreads();
    }
    for (int o = threadIdx.x; o < nout; o += 64 * WAVES) pmom[(size_t)blockIdx.x * nout + o] = part[o];
}

// Workgroup b: correlations of the cross moments o in [64 b, 64 b + 64); it reduces those and (again in every workgroup:
// 2d more quantities) the auto moments it divides by.
__global__ __launch_bounds__(1024) void k_corr_final(const double* __restrict__ pmom, int64_t n, int d, int nblk,
                                                     double* __restrict__ cc) {
    __shared__ double red[16 * 64];
    __shared__ double cross[64];
    __shared__ double autos[128];
    const int nout = d * d + 2 * d, o0 = blockIdx.x * 64, cnt = min(64, d * d - o0);
    const double scale = 1.0 / (double)(n - 1);            // np.cov normalisation (cancels in the ratio)
    reduce_partials64(pmom, nout, nblk, o0, cnt, scale, red, cross);
    for (int a0 = 0; a0 < 2 * d; a0 += 64) reduce_partials64(pmom, nout, nblk, d * d + a0, min(64, 2 * d - a0), scale, red, autos + a0);
    if ((int)threadIdx.x < cnt) {
        const int o = o0 + threadIdx.x, a = o / d, b = o - a * d;
        double c = cross[threadIdx.x] / sqrt(autos[a]);
        c = c / sqrt(autos[d + b]);
        cc[o] = clip_unit(c);
    }
}

// Many small problems in ONE launch: block b owns rows [b*n, (b+1)*n) of x and y (the per-step batch scores of a
// training epoch, runners/ivae.py:86, wrappers.py:75: n = batch size).  Same two-pass centred moments in float64,
// everything inside the workgroup: column means -> centred 64-row chunks in LDS -> moments -> correlation block.
__global__ __launch_bounds__(256) void k_corr_batched(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                      int d, double* __restrict__ cc) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ld = 2 * d + 1, nout = d * d + 2 * d;
    double* tile = sm;                          // CORR_ROWS x (2d+1)
    double* mean = tile + CORR_ROWS * ld;       // 2d
    double* mom = mean + 2 * d;                 // d*d + 2d
    double* red = mom + nout;                   // 256
    const float* xb = x + (size_t)blockIdx.x * n * d;
    const float* yb = y + (size_t)blockIdx.x * n * d;
    {   // pass 1: column means
        const int c = threadIdx.x % (2 * d), sub = threadIdx.x / (2 * d), nsub = 256 / (2 * d);
        double acc = 0.0;
        if (sub < nsub) {
            const float* src = c < d ? xb + c : yb + (c - d);
            for (int64_t r = sub; r < n; r += nsub) acc += (double)src[r * d];
        }
        red[threadIdx.x] = sub < nsub ? acc : 0.0;
        __syncthreads();
        if (threadIdx.x < 2 * d) {
            double t = 0.0;
            for (int s = 0; s < nsub; ++s) t += red[s * 2 * d + threadIdx.x];
            mean[threadIdx.x] = t / (double)n;
        }
        __syncthreads();
    }
    double acc[17];
#pragma unroll
    for (int q = 0; q < 17; ++q) acc[q] = 0.0;
    for (int64_t r0 = 0; r0 < n; r0 += CORR_ROWS) {
        const int nr = (int)((n - r0) < CORR_ROWS ? (n - r0) : CORR_ROWS);
        for (int idx = threadIdx.x; idx < CORR_ROWS * 2 * d; idx += 256) {
            const int r = idx / (2 * d), c = idx - r * 2 * d;
            double val = 0.0;
            if (r < nr) val = (c < d ? (double)xb[(r0 + r) * d + c] : (double)yb[(r0 + r) * d + (c - d)]) - mean[c];
            tile[r * ld + c] = val;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 17; ++q) {
            const int o = threadIdx.x + q * 256;
            if (o < nout) {
                int ca, cb;
                if (o < d * d) { ca = o / d; cb = d + (o - ca * d); }
                else { ca = cb = o - d * d; }
                double s = 0.0;
                for (int r = 0; r < nr; ++r) s += tile[r * ld + ca] * tile[r * ld + cb];
                acc[q] += s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 17; ++q) {
        const int o = threadIdx.x + q * 256;
        if (o < nout) mom[o] = acc[q] / (double)(n - 1);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d * d; o += 256) {
        const int a = o / d, b = o - a * d;
        double c = mom[o] / sqrt(mom[d * d + a]);
        c = c / sqrt(mom[d * d + d + b]);
        cc[(size_t)blockIdx.x * d * d + o] = clip_unit(c);
    }
}

hipError_t launch_corr_batched(const float* x, const float* y, int64_t n_blocks, int64_t n, int d, double* cc, hipStream_t st) {
    static SmemCache cur = {};
    const size_t smem = sizeof(double) * ((size_t)CORR_ROWS * (2 * d + 1) + 2 * d + (size_t)d * d + 2 * d + 256);
    hipError_t e = set_smem((const void*)k_corr_batched, (int)smem, cur);      // d > ~40: above the default dynamic-LDS limit
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_corr_batched, dim3((unsigned)n_blocks), dim3(256), smem, st, x, y, n, d, cc);
    return hipGetLastError();
}

hipError_t launch_corr(const float* x, const float* y, int64_t n, int d, double* cc, double* ws, hipStream_t st) {
    double* mean = ws;
    double* psum = mean + 2 * d;
    double* pmom = psum + (size_t)CORR_BLOCKS * 2 * d;
    int nblk = (int)((n + CORR_ROWS - 1) / CORR_ROWS);
    if (nblk > CORR_BLOCKS) nblk = CORR_BLOCKS;
    const size_t smem = (size_t)(d * d + 2 * d) * sizeof(double);          // d = 64: 33 792 B
    hipLaunchKernelGGL(k_corr_sums, dim3(nblk), dim3(CORR_SUM_THREADS), 0, st, x, y, n, d, psum);
    hipLaunchKernelGGL(k_corr_mean, dim3((2 * d + 63) / 64), dim3(1024), 0, st, psum, n, d, nblk, mean);
    switch ((d + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_corr_moments<1, 16>), dim3(nblk), dim3(1024), smem, st, x, y, n, d, mean, pmom); break;
        case 2: hipLaunchKernelGGL((k_corr_moments<2, 16>), dim3(nblk), dim3(1024), smem, st, x, y, n, d, mean, pmom); break;
        case 3: hipLaunchKernelGGL((k_corr_moments<3, 8>), dim3(nblk), dim3(512), smem, st, x, y, n, d, mean, pmom); break;
        case 4: hipLaunchKernelGGL((k_corr_moments<4, 8>), dim3(nblk), dim3(512), smem, st, x, y, n, d, mean, pmom); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_corr_final, dim3((d * d + 63) / 64), dim3(1024), 0, st, pmom, n, d, nblk, cc);
    return hipGetLastError();
}

// ---- the same computation with the scratch inside the library: ivae_corr_static (callers that cannot allocate) ----
// `scratch` holds what ivae_corr's workspace argument holds (column means, per-block column sums, per-block centred
// moments) for d <= CORR_STATIC_MAX_D.  In host code the array's NAME is only a stand-in: the address the kernels get is
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
