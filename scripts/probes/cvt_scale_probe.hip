#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v2i16 __attribute__((ext_vector_type(2)));
__global__ void k(float* in, float scale, unsigned* out) {
    v2i16 old = {0, 0};
    v2i16 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[0], in[1], scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, in[2], in[3], scale, true);
    out[threadIdx.x] = ((unsigned)(unsigned short)r[0]) | ((unsigned)(unsigned short)r[1] << 16);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(in[0], in[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(in[2], in[3], w, true);
    out[64 + threadIdx.x] = (unsigned)w;
}
int main() {
    float h[4] = {1.0f, 0.75f, 300.0f, -2.5f}, *d; unsigned *o, ho[128];
    hipMalloc((void**)&d, 16); hipMalloc((void**)&o, 512); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    float scales[] = {1.0f, 2.0f, 0.5f, 1.0f / 2048.0f};
    for (float sc : scales) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, sc, o); hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
        printf("scale %g: scalef32 pk -> %08x   plain pk -> %08x\n", sc, ho[0], ho[64]);
    }
    return 0;
}
