// Micro-benchmark: LDS bank conflicts of the fused kernel's access patterns, one kernel instance per pattern so that
// rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS attributes the counters.  c = lane & 15, q = lane >> 4.
// Build: hipcc -O3 --offload-arch=gfx950 lds_conflicts.hip -o lds_conflicts
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ST 52
#define KS 52
enum { FWD_A, FWD_B_OLD, FWD_B_NEW, DGRAD_B, WGRAD_A_OLD, WGRAD_A_NEW, WGRAD_B_OLD, WGRAD_B_NEW, EPI_ST64, EPI_ST32, ADAM_RD, XU_ST, LIN32, STRIDE2_32, LIN64, N_PAT };
static const char* NAMES[N_PAT] = {"fwd/dgrad A: ld64 X[(16rt+c)*ST + 8t+2q]", "fwd B old: ld64 W[(2c+j)*KS + 8t+2q]", "fwd B new: ld64 W[(16j+c)*KS + 8t+2q]",
    "dgrad B: ld64 W[(8t+2q)*KS + 32h+2c]", "wgrad A old: ld32 dY[(8t+q)*ST + 16nt+c]", "wgrad A new: ld32 dY[(b0+{0,8,4,12}[q])*ST + 16nt+c]",
    "wgrad B old: ld64 X[(8t+q)*ST + 32h+2c]", "wgrad B new: ld64 X[(b0+{0,8,4,12}[q])*ST + 32h+2c]", "epilogue st64 out[(4q+r)*ST + 32h+2c]",
    "epilogue st32 out[(4q+r)*ST + 16j+c]", "adam ld32 stage[idx + idx/51], idx = tid + 512j", "store_xu st32 T[(tid>>3)*ST + (tid&7) + 8j]",
    "reference: ld32 lds[lane + 64t] (conflict-free)", "reference: ld32 lds[2*lane + 128t] (2-way)", "reference: ld64 lds[2*lane + 128t] (conflict-free)"};

template <int PAT>
__global__ __launch_bounds__(512) void k(float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 1e-3f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4, rt = w & 3, h = w >> 2;
    const int qoff = ((q & 1) << 3) | ((q >> 1) << 2);
    float acc = 0.f;
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            int a = 0;
            bool wide = true, store = false;
            if (PAT == FWD_A) a = (16 * rt + c) * ST + 8 * t + 2 * q;
            if (PAT == FWD_B_OLD) a = (32 * h + 2 * c + (t & 1)) * KS + 8 * t + 2 * q;
            if (PAT == FWD_B_NEW) a = (32 * h + 16 * (t & 1) + c) * KS + 8 * t + 2 * q;
            if (PAT == DGRAD_B) a = (8 * t + 2 * q) * KS + 32 * h + 2 * c;
            if (PAT == WGRAD_A_OLD) { a = (8 * t + q) * ST + 16 * rt + c; wide = false; }
            if (PAT == WGRAD_A_NEW) { a = (16 * (t / 2) + 2 * (t % 2) + qoff) * ST + 16 * rt + c; wide = false; }
            if (PAT == WGRAD_B_OLD) a = (8 * t + q) * ST + 32 * h + 2 * c;
            if (PAT == WGRAD_B_NEW) a = (16 * (t / 2) + 2 * (t % 2) + qoff) * ST + 32 * h + 2 * c;
            if (PAT == EPI_ST64) { a = (16 * rt + 4 * q + (t & 3)) * ST + min(32 * h + 2 * c, ST - 2); store = true; }
            if (PAT == EPI_ST32) { a = (16 * rt + 4 * q + (t & 3)) * ST + min(16 * (t & 1) + 32 * h + c, ST - 1); wide = false; store = true; }
            if (PAT == ADAM_RD) { const int idx = threadIdx.x + 512 * (t % 5); a = idx + idx / 51; wide = false; }
            if (PAT == XU_ST) { a = (threadIdx.x >> 3) * ST + (threadIdx.x & 7) + 8 * t; wide = false; store = true; }
            if (PAT == LIN32) { a = lane + 64 * t + 512 * w; wide = false; }
            if (PAT == STRIDE2_32) { a = 2 * lane + 128 * t + 1024 * w; wide = false; }
            if (PAT == LIN64) a = 2 * lane + 128 * t + 1024 * w;
            a &= wide ? 16382 : 16383;
            if (store) {
                if (wide) *reinterpret_cast<f32x2*>(lds + a) = (f32x2){acc, acc}; else lds[a] = acc;
            } else if (wide) {
                const f32x2 v = *reinterpret_cast<volatile f32x2*>(lds + a);
                acc += v.x + v.y;
            } else {
                acc += *reinterpret_cast<volatile float*>(lds + a);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int PAT>
void run(float* out) {
    hipLaunchKernelGGL(k<PAT>, dim3(64), dim3(512), 0, 0, out);
    hipDeviceSynchronize();
    printf("k<%d> = %s\n", PAT, NAMES[PAT]);
}

int main() {
    float* out;
    hipMalloc(&out, 64 * 512 * 4);
    run<FWD_A>(out); run<FWD_B_OLD>(out); run<FWD_B_NEW>(out); run<DGRAD_B>(out); run<WGRAD_A_OLD>(out); run<WGRAD_A_NEW>(out);
    run<WGRAD_B_OLD>(out); run<WGRAD_B_NEW>(out); run<EPI_ST64>(out); run<EPI_ST32>(out); run<ADAM_RD>(out); run<XU_ST>(out); run<LIN32>(out); run<STRIDE2_32>(out); run<LIN64>(out);
    hipFree(out);
    return 0;
}
