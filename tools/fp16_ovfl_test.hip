// MODE.FP16_OVFL probe for gfx950: does an f32 -> f16 conversion of an out-of-range value saturate to +-65504 when the bit
// is set?  hipcc --offload-arch=gfx950 -O3 tools/fp16_ovfl_test.hip -o tools/fp16_ovfl_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, int n, int mode) {
    if (mode) __builtin_amdgcn_s_setreg(1473, 1);     // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
    const int i = threadIdx.x;
    if (i < n) {
        f32x2 v = {in[i], -in[i]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2 b = __builtin_convertvector(h, f32x2);
        const f32x2 lo = v - b;
        const f16x2 l = __builtin_convertvector(lo, f16x2);
        out[4 * i] = b.x; out[4 * i + 1] = b.y; out[4 * i + 2] = (float)l.x; out[4 * i + 3] = (float)(_Float16)in[i];
    }
}
int main() {
    const float h[8] = {1.0f, 65504.f, 65520.f, 70000.f, 1e6f, INFINITY, NAN, 3e38f};
    float *d, *o, r[32];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, 8, mode);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int i = 0; i < 8; ++i) printf("  in %g -> pk hi %g / %g  lo %g  scalar cvt %g\n", h[i], r[4*i], r[4*i+1], r[4*i+2], r[4*i+3]);
    }
    return 0;
}
