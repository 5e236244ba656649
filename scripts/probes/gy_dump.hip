#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../../gliclass/c_amd/csrc/glc_common.h"
__global__ void enc(unsigned char* dstA, unsigned char* dstW, int K) {
    const int e0 = threadIdx.x * 16;
    if (e0 >= K) return;
    float a[16], w[16];
    for (int e = 0; e < 16; ++e) { a[e] = 1.0f + ldexpf(1.f, -12); w[e] = 1.0f; }
    gy_store16<false>(dstA, K, e0, a);
    gy_store16<true>(dstW, K, e0, w);
}
static float dec6(int v) { const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7; const float x = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1); return s ? -x : x; }
int main() {
    const int K = 64;
    unsigned char *dA, *dW; hipMalloc(&dA, gy_row_bytes(K)); hipMalloc(&dW, gy_row_bytes(K));
    hipLaunchKernelGGL(enc, dim3(1), dim3(64), 0, 0, dA, dW, K);
    std::vector<unsigned char> A(gy_row_bytes(K)), W(gy_row_bytes(K));
    hipMemcpy(A.data(), dA, A.size(), hipMemcpyDeviceToHost); hipMemcpy(W.data(), dW, W.size(), hipMemcpyDeviceToHost);
    for (int which = 0; which < 2; ++which) {
        const std::vector<unsigned char>& R = which ? W : A;
        printf("%s row: scale bytes %d %d %d %d; group 0 block 0 values:", which ? "W" : "A", R[2 * 112], R[2 * 112 + 1], R[2 * 112 + 2], R[2 * 112 + 3]);
        unsigned char blk[24];
        for (int i = 0; i < 16; ++i) blk[i] = R[64 + i];
        for (int i = 0; i < 8; ++i) blk[16 + i] = R[96 + i];
        for (int k = 0; k < 32; ++k) {
            const int bit = 6 * k; int v = 0;
            for (int b = 0; b < 6; ++b) v |= ((blk[(bit + b) >> 3] >> ((bit + b) & 7)) & 1) << b;
            printf(" %.3g", dec6(v));
        }
        printf("\n");
    }
    return 0;
}
