// Does a raw buffer load's range check include the scalar offset on gfx950?  A 4 KB allocation filled with 1.0f, a
// descriptor of 64 bytes over its start; loads at (voffset, soffset) pairs inside and outside the 64 bytes.  Out-of-range
// loads return 0.  build: hipcc --offload-arch=gfx950 -O2 soffset_range.hip -o ../../var/soffset_range
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 64, 0x00020000);
    const int l = threadIdx.x;
    // lane 0: v=0,s=0 (in) | 1: v=60,s=0 (in) | 2: v=64,s=0 (out by v) | 3: v=0,s=64 (out by s) | 4: v=32,s=32 (sum out) | 5: v=32,s=28 (sum in)
    const int vo[6] = {0, 60, 64, 0, 32, 32};
    float r = -1.f;
    if (l < 6) {
        const int v = vo[l];
        int s = l == 3 ? 64 : l == 4 ? 32 : l == 5 ? 28 : 0;
        s = __builtin_amdgcn_readfirstlane(s);      // (wave-uniform per call below)
        r = 0.f;
    }
    // one call per scalar offset (soffset must be uniform)
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l < 6 ? vo[l] : 0, 0, 0));
    float r64 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l < 6 ? vo[l] : 0, 64, 0));
    float r32 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l < 6 ? vo[l] : 0, 32, 0));
    float r28 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l < 6 ? vo[l] : 0, 28, 0));
    if (l < 6) { out[4 * l] = r0; out[4 * l + 1] = r64; out[4 * l + 2] = r32; out[4 * l + 3] = r28; }
    (void)r;
}
int main() {
    float *p, *o, h[1024], ho[24];
    for (int i = 0; i < 1024; ++i) h[i] = 1.f;
    hipMalloc(&p, 4096); hipMalloc(&o, sizeof(ho));
    hipMemcpy(p, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const int vo[6] = {0, 60, 64, 0, 32, 32};
    const int so[4] = {0, 64, 32, 28};
    for (int l = 0; l < 6; ++l)
        for (int j = 0; j < 4; ++j)
            printf("voffset %2d soffset %2d (sum %3d, limit 64): %s\n", vo[l], so[j], vo[l] + so[j], ho[4 * l + j] == 1.f ? "data" : "zero");
    return 0;
}
