// Probe (developer): GY rows (glc_common.h) — fp32 -> GY (A order) -> fp32 round trip, and the scale bytes / packed parts of one block.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../../gliclass/c_amd/csrc/glc_common.h"
__global__ void enc(const float* src, unsigned char* dst, int rows, int K) {
    const int bi = blockIdx.x * 256 + threadIdx.x;
    if (bi >= rows * (K >> 4)) return;
    const int row = bi / (K >> 4), e0 = (bi % (K >> 4)) * 16;
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = src[row * K + e0 + e];
    gy_store16<false>(dst + row * gy_row_bytes(K), K, e0, v);
}
__global__ void dec(const unsigned char* src, float* dst, int rows, int K) {
    const int bi = blockIdx.x * 256 + threadIdx.x;
    if (bi >= rows * (K >> 4)) return;
    const int row = bi / (K >> 4), e0 = (bi % (K >> 4)) * 16;
    float v[16];
    gy_load16(src + row * gy_row_bytes(K), K, e0, v);
    for (int e = 0; e < 16; ++e) dst[row * K + e0 + e] = v[e];
}
int main() {
    const int rows = 8, K = 128;
    std::vector<float> h(rows * K), o(rows * K);
    unsigned s = 99u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f - 1.f) * 2.0f; }
    float *d, *d2; unsigned char* g;
    hipMalloc(&d, h.size() * 4); hipMalloc(&d2, h.size() * 4); hipMalloc(&g, rows * gy_row_bytes(K));
    hipMemset(g, 0xEE, rows * gy_row_bytes(K));
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(enc, dim3(1), dim3(256), 0, 0, d, g, rows, K);
    hipLaunchKernelGGL(dec, dim3(1), dim3(256), 0, 0, g, d2, rows, K);
    hipMemcpy(o.data(), d2, o.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned char> raw(rows * gy_row_bytes(K));
    hipMemcpy(raw.data(), g, raw.size(), hipMemcpyDeviceToHost);
    double md = 0, mr = 0;
    for (size_t i = 0; i < h.size(); ++i) { md = fmax(md, fabs((double)o[i] - h[i]) / fmax(1e-3, fabs(h[i]))); mr = fmax(mr, fabs(h[i])); }
    printf("row bytes %zu; round trip max relative error %.3e (f16 alone would be 4.9e-4; hi + e2m3 lo: ~3e-5)\n", gy_row_bytes(K), md);
    printf("row 0 scale bytes:"); for (int i = 0; i < K / 16; ++i) printf(" %d", raw[(K / 32) * 112 + i]); printf("\n");
    printf("row 0 group 0 fp6 area:"); for (int i = 64; i < 112; ++i) printf(" %02x", raw[i]); printf("\n");
    printf("row 0 first values: %.5f %.5f %.5f -> %.5f %.5f %.5f\n", h[0], h[1], h[2], o[0], o[1], o[2]);
    return 0;
}
