// Pearson cross-correlation block for the MCC metric, float64 arithmetic on fp32 inputs.
//
// Replaces np.corrcoef(x, y, rowvar=False)[:d, d:] (metrics/mcc.py:328; torch spec at
// mcc.py:174-253): two-pass centred moments.  Pass 1: column sums -> means.  Pass 2: centred
// cross/auto moments, 64-row chunks staged in LDS as doubles, per-block partials reduced in a
// fixed order (bitwise reproducible).  The Hungarian step (mcc.py:334) stays on the host.
#include <hip/hip_runtime.h>
#include "kernels.h"

#define CORR_BLOCKS 256
#define CORR_ROWS 64

size_t corr_ws_bytes(int64_t n, int d) {
    (void)n;
    return sizeof(double) * ((size_t)2 * d + (size_t)CORR_BLOCKS * 2 * d + (size_t)CORR_BLOCKS * ((size_t)d * d + 2 * d));
}

__global__ __launch_bounds__(256) void k_corr_sums(const float* __restrict__ x, const float* __restrict__ y,
                                                   int64_t n, int d, double* __restrict__ psum) {
    // thread c < 2d owns a column; rows strided over blocks.  (d is small: <= 64.)
    __shared__ double red[256];
    const int c = threadIdx.x % (2 * d);
    const int sub = threadIdx.x / (2 * d);
    const int nsub = 256 / (2 * d);
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

__global__ void k_corr_mean(const double* __restrict__ psum, int64_t n, int d, int nblk, double* __restrict__ mean) {
    const int c = threadIdx.x;
    if (c < 2 * d) {
        double t = 0.0;
        for (int b0 = 0; b0 < nblk; b0 += 8) {          // eight partials in flight, summed in block order
            double p8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p8[e] = b0 + e < nblk ? psum[(size_t)(b0 + e) * 2 * d + c] : 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (b0 + e < nblk) t += p8[e];
        }
        mean[c] = t / (double)n;
    }
}

__global__ __launch_bounds__(256) void k_corr_moments(const float* __restrict__ x, const float* __restrict__ y,
                                                      int64_t n, int d, const double* __restrict__ mean,
                                                      double* __restrict__ pmom) {
    extern __shared__ __attribute__((aligned(16))) double tile[];   // CORR_ROWS x (2d+1)
    const int ld = 2 * d + 1;
    const int nout = d * d + 2 * d;
    // every thread keeps up to ceil(nout/256) accumulators; nout <= 64*64+128 -> <= 17
    double acc[17];
#pragma unroll
    for (int q = 0; q < 17; ++q) acc[q] = 0.0;
    // the next chunk's elements are requested (into registers) before the current chunk's products run, so the memory
    // round trip of a chunk hides under the arithmetic of the one before; CORR_ROWS * 2d / 256 <= 32 elements per thread
    float nxt[32];
    auto fetch = [&](int64_t r0) {
        const int nr = (int)((n - r0) < CORR_ROWS ? (n - r0) : CORR_ROWS);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int idx = threadIdx.x + e * 256;
            const int r = idx / (2 * d), c = idx - r * 2 * d;
            float val = 0.f;
            if (idx < CORR_ROWS * 2 * d && r < nr) val = c < d ? x[(r0 + r) * d + c] : y[(r0 + r) * d + (c - d)];
            nxt[e] = val;
        }
    };
    const int64_t step = (int64_t)gridDim.x * CORR_ROWS;
    if ((int64_t)blockIdx.x * CORR_ROWS < n) fetch((int64_t)blockIdx.x * CORR_ROWS);
    for (int64_t r0 = (int64_t)blockIdx.x * CORR_ROWS; r0 < n; r0 += step) {
        const int nr = (int)((n - r0) < CORR_ROWS ? (n - r0) : CORR_ROWS);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int idx = threadIdx.x + e * 256;
            if (idx < CORR_ROWS * 2 * d) {
                const int r = idx / (2 * d), c = idx - r * 2 * d;
                tile[r * ld + c] = r < nr ? (double)nxt[e] - mean[c] : 0.0;
            }
        }
        __syncthreads();
        if (r0 + step < n) fetch(r0 + step);
#pragma unroll
        for (int q = 0; q < 17; ++q) {
            const int o = threadIdx.x + q * 256;
            if (o < nout) {
                int ca, cb;
                if (o < d * d) { ca = o / d; cb = d + (o - ca * d); }
                else { ca = cb = o - d * d; }
                double s = 0.0;
                // rows past the chunk's end are zeros in the tile: a fixed trip count lets the loads of several rows be in flight
#pragma unroll 8
                for (int r = 0; r < CORR_ROWS; ++r) s += tile[r * ld + ca] * tile[r * ld + cb];
                acc[q] += s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 17; ++q) {
        const int o = threadIdx.x + q * 256;
        if (o < nout) pmom[(size_t)blockIdx.x * nout + o] = acc[q];
    }
}

__global__ __launch_bounds__(256) void k_corr_final(const double* __restrict__ pmom, int64_t n, int d, int nblk,
                                                    double* __restrict__ cc) {
    extern __shared__ __attribute__((aligned(16))) double mom[];   // d*d + 2d
    const int nout = d * d + 2 * d;
    for (int o = threadIdx.x; o < nout; o += 256) {
        double t = 0.0;
        for (int b0 = 0; b0 < nblk; b0 += 8) {          // eight partials in flight, summed in block order
            double p8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p8[e] = b0 + e < nblk ? pmom[(size_t)(b0 + e) * nout + o] : 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (b0 + e < nblk) t += p8[e];
        }
        mom[o] = t / (double)(n - 1);        // np.cov normalisation (cancels, kept for rounding parity)
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d * d; o += 256) {
        const int a = o / d, b = o - a * d;
        double c = mom[o] / sqrt(mom[d * d + a]);
        c = c / sqrt(mom[d * d + d + b]);
        cc[o] = fmin(1.0, fmax(-1.0, c));     // np.corrcoef clips to [-1, 1]
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
        cc[(size_t)blockIdx.x * d * d + o] = fmin(1.0, fmax(-1.0, c));
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
    static SmemCache cur = {};
    double* mean = ws;
    double* psum = mean + 2 * d;
    double* pmom = psum + (size_t)CORR_BLOCKS * 2 * d;
    int nblk = (int)((n + CORR_ROWS - 1) / CORR_ROWS);
    if (nblk > CORR_BLOCKS) nblk = CORR_BLOCKS;
    const size_t smem = (size_t)CORR_ROWS * (2 * d + 1) * sizeof(double);          // d = 64: 66048 B, above the default limit
    hipError_t e = set_smem((const void*)k_corr_moments, (int)smem, cur);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_corr_sums, dim3(nblk), dim3(256), 0, st, x, y, n, d, psum);
    hipLaunchKernelGGL(k_corr_mean, dim3(1), dim3(128), 0, st, psum, n, d, nblk, mean);
    hipLaunchKernelGGL(k_corr_moments, dim3(nblk), dim3(256), smem, st, x, y, n, d, mean, pmom);
    hipLaunchKernelGGL(k_corr_final, dim3(1), dim3(256), (size_t)(d * d + 2 * d) * sizeof(double), st, pmom, n, d, nblk, cc);
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
