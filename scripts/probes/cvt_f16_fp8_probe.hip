// Probe (round 4): v_cvt_scalef32_pk_fp8_f16 — scale direction, saturation, byte order — for deriving the hi8 operand of the MX cross-term
// MFMA from the f16 fragments in registers (gemm256x.hip, 96-byte LDS rows).   hipcc --offload-arch=gfx950 cvt_f16_fp8_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef short v2i16 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale) {
    const int i = threadIdx.x;
    f16x2 v = {(f16_t)in[2 * i], (f16_t)in[2 * i + 1]};
    v2i16 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, v, scale, false);      // low half
    v2i16 r2 = {(short)0x1234, (short)0x5678};
    r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r2, v, scale, true);     // high half, low preserved?
    out[2 * i] = __builtin_bit_cast(unsigned, r);
    out[2 * i + 1] = __builtin_bit_cast(unsigned, r2);
}
static float e4m3(unsigned b) {
    b &= 255; int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return s ? -v : v;
}
int main() {
    const float h[16] = {1.0f, -2.5f, 0.0156f, 300.0f, 448.0f, 449.0f, 500.0f, 1000.0f, 60000.0f, -700.0f, 0.001f, 0.3f, 17.0f, 18.0f, 19.0f, 20.0f};
    float* d; unsigned* o;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 16 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 0.25f, 4.0f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, scale);
        unsigned r[16];
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int i = 0; i < 8; ++i)
            printf("  in (%g, %g) -> lo-half word %08x = (%g, %g) | hi-half word %08x = hi (%g, %g) low kept %04x\n", h[2 * i], h[2 * i + 1], r[2 * i], e4m3(r[2 * i]), e4m3(r[2 * i] >> 8),
                   r[2 * i + 1], e4m3(r[2 * i + 1] >> 16), e4m3(r[2 * i + 1] >> 24), r[2 * i + 1] & 0xffff);
    }
    return 0;
}
