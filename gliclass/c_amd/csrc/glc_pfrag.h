// Position blocks that stay in registers over several key tiles (attention_mx.hip; round 5).
#pragma once
#include "glc_common.h"

typedef __attribute__((ext_vector_type(8))) int glc_i32x8;
typedef __attribute__((ext_vector_type(4))) int glc_i32x4;
// the same tile as eight 16-byte pieces: a position block that stays in registers over several steps and is replaced IN PLACE by a
// conditional request (as one C++ value per block the replacement is a 32-register phi: hipcc answers with whole-fragment copies and spills)
struct PFrag { f16x8 f[4]; glc_i32x4 xa[2], xb[2]; };
// rows of a position table into `f` if `cond` (wave-uniform) — one asm block, tied operands, a scalar branch inside.  vf / vx: load_rows.
// The loads are NOT tracked by the compiler's s_waitcnt insertion: pfrag_wait before the first use.
__device__ __forceinline__ void pfrag_load_if(int cond, const unsigned char* base, unsigned vf, unsigned vx, PFrag& f) {
    asm volatile("s_cmp_eq_u32 %[c], 0\n\ts_cbranch_scc1 .Lpfskip%=\n\t"
                 "global_load_dwordx4 %[f0], %[vf], %[b]\n\tglobal_load_dwordx4 %[f1], %[vf], %[b] offset:1024\n\t"
                 "global_load_dwordx4 %[f2], %[vf], %[b] offset:2048\n\tglobal_load_dwordx4 %[f3], %[vf], %[b] offset:3072\n\t"
                 "global_load_dwordx4 %[a0], %[vx], %[b]\n\tglobal_load_dwordx4 %[b0], %[vx], %[b] offset:16\n\t"
                 "global_load_dwordx4 %[a1], %[vx], %[b] offset:2048\n\tglobal_load_dwordx4 %[b1], %[vx], %[b] offset:2064\n"
                 ".Lpfskip%=:"
                 : [f0] "+v"(f.f[0]), [f1] "+v"(f.f[1]), [f2] "+v"(f.f[2]), [f3] "+v"(f.f[3]), [a0] "+v"(f.xa[0]), [b0] "+v"(f.xb[0]), [a1] "+v"(f.xa[1]), [b1] "+v"(f.xb[1])
                 : [c] "s"(__builtin_amdgcn_readfirstlane(cond)), [vf] "v"(vf), [vx] "v"(vx), [b] "s"(base)
                 : "scc", "memory");
}
// the same from the PLANAR copy of a table (round 6): all eight pieces on ONE offset register, f16 units from `b`, the MX planes from `bx` = b + 4096
__device__ __forceinline__ void pfrag_load_planar_if(int cond, const unsigned char* b, const unsigned char* bx, unsigned vo, PFrag& f) {
    asm volatile("s_cmp_eq_u32 %[c], 0\n\ts_cbranch_scc1 .Lpfskip%=\n\t"
                 "global_load_dwordx4 %[f0], %[vo], %[b]\n\tglobal_load_dwordx4 %[f1], %[vo], %[b] offset:1024\n\t"
                 "global_load_dwordx4 %[f2], %[vo], %[b] offset:2048\n\tglobal_load_dwordx4 %[f3], %[vo], %[b] offset:3072\n\t"
                 "global_load_dwordx4 %[a0], %[vo], %[bx]\n\tglobal_load_dwordx4 %[b0], %[vo], %[bx] offset:1024\n\t"
                 "global_load_dwordx4 %[a1], %[vo], %[bx] offset:2048\n\tglobal_load_dwordx4 %[b1], %[vo], %[bx] offset:3072\n"
                 ".Lpfskip%=:"
                 : [f0] "+v"(f.f[0]), [f1] "+v"(f.f[1]), [f2] "+v"(f.f[2]), [f3] "+v"(f.f[3]), [a0] "+v"(f.xa[0]), [b0] "+v"(f.xb[0]), [a1] "+v"(f.xa[1]), [b1] "+v"(f.xb[1])
                 : [c] "s"(__builtin_amdgcn_readfirstlane(cond)), [vo] "v"(vo), [b] "s"(b), [bx] "s"(bx)
                 : "scc", "memory");
}
// every vector-memory request of this wave has landed; the uses of both position blocks stay behind the wait.  (NOT a counted wait that
// leaves younger LDS-DMA pieces in flight: on gfx950 `global_load_lds` pieces and register loads do not retire in issue order relative to
// each other — `s_waitcnt vmcnt(2)` behind [8 register loads, 2 DMA pieces] let the MFMAs read rows that had not arrived; measured, round 5.)
__device__ __forceinline__ void pfrag_wait_all(PFrag& f, PFrag& g) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(f.f[0]), "+v"(f.f[1]), "+v"(f.f[2]), "+v"(f.f[3]), "+v"(f.xa[0]), "+v"(f.xb[0]), "+v"(f.xa[1]), "+v"(f.xb[1]),
                                        "+v"(g.f[0]), "+v"(g.f[1]), "+v"(g.f[2]), "+v"(g.f[3]), "+v"(g.xa[0]), "+v"(g.xb[0]), "+v"(g.xa[1]), "+v"(g.xb[1]) :: "memory");
}
