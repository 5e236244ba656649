// Developer probe (GPU): operand layout, scale semantics and issue rate of the gfx950 block-scaled MFMA
//   v_mfma_scale_f32_16x16x128_f8f6f4  (fp8 e4m3 operands, e8m0 block scales)
// against the f16 16x16x32 MFMA — the building block of the "low-precision correction terms" idea (docs/LOG_r01-r05.md §9).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// one wave: a[lane][32 bytes], b[lane][32 bytes] -> c[lane][4]
__global__ void k_once(const uint8_t* A, const uint8_t* B, float* C, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + lane * 32);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + lane * 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A fp8*/, 0 /*B fp8*/, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) C[lane * 4 + r] = c[r];
}

template <int MODE>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + lane + i; b[i] = 0x38383838 - lane - i; }
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static uint8_t fp8_of(int v) { static const uint8_t t[5] = {0x00, 0x38, 0x40, 0x44, 0x48}; return t[v]; }   // 0,1,2,3,4 in e4m3

int main() {
    const int L = 64;
    uint8_t hA[64 * 32], hB[64 * 32];
    int Am[16][128], Bm[128][16];
    srand(7);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) Am[i][k] = rand() % 4;
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) Bm[k][j] = rand() % 4;
    float ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 128; ++k) s += Am[i][k] * Bm[k][j]; ref[i][j] = s; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 64 * 4 * sizeof(float));
    // hypotheses for (lane, byte) -> k:  H0: k = 32*(lane>>4) + byte ;  H1: k = 16*(lane>>4) + (byte & 15) + 64*(byte >> 4)
    for (int hyp = 0; hyp < 2; ++hyp) {
        for (int l = 0; l < L; ++l) for (int by = 0; by < 32; ++by) {
            const int g = l >> 4, r = l & 15;
            const int k = hyp == 0 ? 32 * g + by : 16 * g + (by & 15) + 64 * (by >> 4);
            hA[l * 32 + by] = fp8_of(Am[r][k]);
            hB[l * 32 + by] = fp8_of(Bm[k][r]);
        }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0x7f7f7f7f, 0x7f7f7f7f);
        float hC[256]; hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        int bad = 0; double maxd = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {      // C/D map of the 16x16 family: col = lane & 15, row = 4 * (lane >> 4) + reg
            const float want = ref[4 * (l >> 4) + r][l & 15];
            const double d = fabs(hC[l * 4 + r] - want); if (d > 1e-3) ++bad; if (d > maxd) maxd = d;
        }
        printf("layout hypothesis %d (k = %s): %d / 256 wrong, max diff %.3f  (sample C[0][0] gpu %.1f ref %.1f)\n", hyp,
               hyp == 0 ? "32*(lane>>4) + byte" : "16*(lane>>4) + (byte&15) + 64*(byte>>4)", bad, maxd, hC[0], ref[0][0]);
        if (bad == 0) {
            // scales: e8m0 byte 127 = 1.0; 128 = 2.0 on A, 126 = 0.5 on B; which byte does opsel 0 take, and is the scale per lane?
            hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0x7f7f7f80, 0x7f7f7f7f);
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            printf("  scale_a low byte 0x80 (x2): C[0][0] = %.1f (x%.2f)\n", hC[0], hC[0] / ref[0][0]);
            hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0x807f7f7f, 0x7f7f7f7f);
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            printf("  scale_a high byte 0x80:     C[0][0] = %.1f (x%.2f)\n", hC[0], hC[0] / ref[0][0]);
            hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0x7f7f7f7f, 0x7f7f7f7e);
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            printf("  scale_b low byte 0x7e (x0.5): C[0][0] = %.1f (x%.2f)\n", hC[0], hC[0] / ref[0][0]);
        }
    }
    // issue rate: 4 independent accumulators, one wave per SIMD (256 threads), cycles per MFMA
    float* dOut; unsigned long long* dCyc; hipMalloc(&dOut, 256 * 256 * sizeof(float)); hipMalloc(&dCyc, 8);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, dOut, dCyc, iters);
            else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, dOut, dCyc, iters);
            hipDeviceSynchronize();
        }
        unsigned long long cyc; hipMemcpy(&cyc, dCyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)cyc / (4.0 * iters);
        const double flop = mode == 0 ? 2.0 * 16 * 16 * 128 : 2.0 * 16 * 16 * 32;
        printf("%s: %.1f s_memtime ticks per MFMA (one wave per SIMD) -> %.0f FLOP per tick per SIMD\n",
               mode == 0 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8)" : "v_mfma_f32_16x16x32_f16", per, flop / per);
    }
    return 0;
}
