// Developer probe (GPU, round 4): issue cost of the VALU instructions the attention softmax and the operand conversions are made of — one
// wave per SIMD, 8 independent chains, 64 instructions per loop iteration, cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float x[8];
    int w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = 1.0f + 1e-3f * (lane + i); w[i] = 0x3c003c00 + lane + i; p[i] = (f32x2){x[i], x[i] + 1.f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], 0.999f, 1e-4f);
                else if (OP == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
                else if (OP == 2) w[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[(i + 1) & 7], w[i], false);
                else if (OP == 3) { v2i16 t = __builtin_bit_cast(v2i16, w[i]); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, x[i], x[(i + 1) & 7], 0.5f, false); w[i] = __builtin_bit_cast(int, t); }
                else if (OP == 4) { v2i16 t = __builtin_bit_cast(v2i16, w[i]); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(t, __builtin_bit_cast(f16x2, w[(i + 1) & 7]), 0.5f, false); w[i] = __builtin_bit_cast(int, t); }
                else if (OP == 5) { f16x2 t = {(_Float16)x[i], (_Float16)x[(i + 1) & 7]}; w[i] ^= __builtin_bit_cast(int, t); }      // v_cvt_pk_f16_f32 (+ xor)
                else if (OP == 6) p[i] = p[i] * (f32x2){0.999f, 0.998f} + (f32x2){1e-4f, 2e-4f};                                         // v_pk_fma_f32
                else if (OP == 7) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 7]), x[(i + 2) & 7]) * 0.999f;               // v_max3 (+ mul)
                else if (OP == 8) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(x[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]), "v"(w[i]));
                else if (OP == 9) w[i] = (int)__builtin_amdgcn_perm((unsigned)w[i], (unsigned)w[(i + 1) & 7], 0x07050301u);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i] + (float)w[i] + p[i][0] + p[i][1];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int OP> static void run(const char* what, float* out, unsigned long long* cyc, double per_iter) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    unsigned long long h[1024];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    printf("%-44s %6.2f cycles per instruction (one wave per SIMD, independent chains)\n", what, s / 1024 / iters / per_iter);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 1024 * 8);
    run<0>("v_fma_f32", out, cyc, 64);
    run<1>("v_exp_f32", out, cyc, 64);
    run<2>("v_cvt_pk_fp8_f32", out, cyc, 64);
    run<3>("v_cvt_scalef32_pk_fp8_f32", out, cyc, 64);
    run<4>("v_cvt_scalef32_pk_fp8_f16", out, cyc, 64);
    run<5>("v_cvt_pk_f16_f32 + v_xor", out, cyc, 128);
    run<6>("v_pk_fma_f32", out, cyc, 64);
    run<7>("v_max3_f32 + v_mul_f32", out, cyc, 128);
    run<8>("v_fma_mix_f32", out, cyc, 64);
    run<9>("v_perm_b32", out, cyc, 64);
    return 0;
}
