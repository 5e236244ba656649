// Developer probe (GPU, round 3): how fast can a workgroup stream L2-resident operand rows into LDS?  The GEMM's loop is bound by its
// LDS-DMA stream (docs/LOG_r01-r05.md §3e); this compares, on the GEMM's access shape (8 waves, 1 KiB per wave-instruction = 8 rows x 128 B,
// a 64 KiB stage = 256 rows x 128 B x two operands, every CU re-reading buffers that fit the L2), per CU and second:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), one 64 KiB stage in flight, vmcnt(0) + barrier per stage  (what gemm256x does)
//   mode 1  the same with two stages in flight (ring of 2 x 64 KiB)
//   mode 2  global_load_dwordx4 into registers, ds_write_b128 a stage later (register staging, 8 x 16 B per thread in flight)
//   mode 3  register staging with 16 x 16 B per thread in flight
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/stage_probe.hip -o /tmp/stage_probe && /tmp/stage_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ void glds16(const unsigned char* gsrc_lane, void* lds_wave_base) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(la), "v"(gsrc_lane) : "memory");
}

// rows: A = tile_m * 256 rows, W = tile_n * 256 rows, row pitch = K * 4 bytes (GX rows), stage g = bytes [128 g, 128 g + 128) of each row
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_stream(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W, int K, int ntm, int ntn,
                                                   int tiles_per_wg, unsigned* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t pitch = (size_t)K * 4;
    const int ngroups = K / 32;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = (blockIdx.x * tiles_per_wg + t);
        const int tm = (tile / ntn) % ntm, tn = tile % ntn;
        // this thread's 16-byte pieces of a stage: operand o (0 A, 1 W), piece p = wave * 4 + i (32 pieces of 8 rows per operand), row = 8 p + lane / 8, chunk = lane % 8
        const unsigned char* srcA = A + ((size_t)tm * 256 + (lane >> 3)) * pitch + (lane & 7) * 16;
        const unsigned char* srcW = W + ((size_t)tn * 256 + (lane >> 3)) * pitch + (lane & 7) * 16;
        if (MODE <= 1) {
            constexpr int NS = MODE == 0 ? 1 : 2;
            auto issue = [&](int g) {
                unsigned char* st = smem + (size_t)(g % (NS + 1)) * 65536 * 0 + (size_t)(g & 1) * 65536;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = wave * 4 + i;
                    glds16(srcA + (size_t)(8 * p) * pitch + (size_t)g * 128, st + p * 1024);
                    glds16(srcW + (size_t)(8 * p) * pitch + (size_t)g * 128, st + 32768 + p * 1024);
                }
            };
            issue(0);
            if (NS == 2 && ngroups > 1) issue(1);
            for (int g = 0; g < ngroups; ++g) {
                if (NS == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                else { if (g + 1 < ngroups) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
                acc += *reinterpret_cast<const unsigned*>(smem + (size_t)(g & 1) * 65536 + threadIdx.x * 4);      // touch the stage
                __builtin_amdgcn_s_barrier();
                if (g + NS < ngroups) issue(g + NS);
            }
        } else {
            constexpr int NR = MODE == 2 ? 8 : 16;           // 16-byte registers in flight per thread (8 = one stage)
            u32x4 r[NR];
            auto load = [&](int g, int base) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = wave * 4 + i;
                    r[base + 2 * i] = *reinterpret_cast<const u32x4*>(srcA + (size_t)(8 * p) * pitch + (size_t)g * 128);
                    r[base + 2 * i + 1] = *reinterpret_cast<const u32x4*>(srcW + (size_t)(8 * p) * pitch + (size_t)g * 128);
                }
            };
            auto store = [&](int g, int base) {
                unsigned char* st = smem + (size_t)(g & 1) * 65536;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = wave * 4 + i;
                    *reinterpret_cast<u32x4*>(st + p * 1024 + lane * 16) = r[base + 2 * i];
                    *reinterpret_cast<u32x4*>(st + 32768 + p * 1024 + lane * 16) = r[base + 2 * i + 1];
                }
            };
            load(0, 0);
            if (NR == 16 && ngroups > 1) load(1, 8);
            for (int g = 0; g < ngroups; ++g) {
                const int base = NR == 16 ? 8 * (g & 1) : 0;
                store(g, base);                               // (the compiler waits for exactly these registers)
                if (g + NR / 8 < ngroups) load(g + NR / 8, base);
                __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                acc += *reinterpret_cast<const unsigned*>(smem + (size_t)(g & 1) * 65536 + threadIdx.x * 4);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE> static void run(const unsigned char* A, const unsigned char* W, int K, int ntm, int ntn, unsigned* sink, unsigned long long* cyc, const char* what) {
    const int nwg = 256, tiles = 12;
    hipFuncSetAttribute((const void*)k_stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream<MODE>, dim3(nwg), dim3(512), 131072, 0, A, W, K, ntm, ntn, tiles, sink, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nwg * tiles * (K / 32) * 65536.0;
        if (rep == 2) printf("%-58s %8.3f ms  %7.2f TB/s aggregate  %6.1f GB/s per CU  (%s)\n", what, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
    }
}

int main() {
    const int K = 768, ntm = 256, ntn = 12;                 // FFN1's shape: A 65536 x 768 GX rows (201 MB), W 3072 x 768 (9.4 MB)
    unsigned char *A, *W; unsigned* sink; unsigned long long* cyc;
    hipMalloc((void**)&A, (size_t)ntm * 256 * K * 4); hipMalloc((void**)&W, (size_t)ntn * 256 * K * 4);
    hipMalloc((void**)&sink, 64); hipMalloc((void**)&cyc, 256 * 8);
    hipMemset(A, 1, (size_t)ntm * 256 * K * 4); hipMemset(W, 2, (size_t)ntn * 256 * K * 4);
    printf("FFN1-shaped streams (tile order n fastest: 12 workgroups share an A tile at a time), 256 workgroups x 12 tiles x 24 stages of 64 KiB\n");
    run<0>(A, W, K, ntm, ntn, sink, cyc, "LDS-DMA, one stage in flight");
    run<1>(A, W, K, ntm, ntn, sink, cyc, "LDS-DMA, two stages in flight");
    run<2>(A, W, K, ntm, ntn, sink, cyc, "register staging, 8 x 16 B per thread in flight");
    run<3>(A, W, K, ntm, ntn, sink, cyc, "register staging, 16 x 16 B per thread in flight");
    printf("small working set (A = 8 tiles, all L2 / MALL resident)\n");
    run<0>(A, W, K, 8, ntn, sink, cyc, "LDS-DMA, one stage in flight");
    run<1>(A, W, K, 8, ntn, sink, cyc, "LDS-DMA, two stages in flight");
    run<2>(A, W, K, 8, ntn, sink, cyc, "register staging, 8 x 16 B per thread in flight");
    run<3>(A, W, K, 8, ntn, sink, cyc, "register staging, 16 x 16 B per thread in flight");
    return 0;
}
