// Developer (EXPERIMENTS.md E48): is there a wait state missing between a VALU instruction that writes a
// register and an MFMA that reads it as A (B is symmetric) on gfx950?  hipcc (ROCm 7.2) inserts none: the hardware is
// supposed to interlock.  One wave per trial: the operand register holds an OLD value; a VALU instruction writes the NEW
// value; DIST wait states later (s_nop, inline asm: the hazard recogniser does not look inside) the MFMA under test reads
// it; a reference MFMA with the same registers runs long after.  Any lane whose two results differ counts as a mismatch.
// All values are chosen so that the products are exact in fp32 (no rounding: the result does not depend on summation order).
//   writers: v_perm_b32, v_sub_f32, v_and_b32, v_xor_b32, v_mov_b32     MFMAs: v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x2_f32
//   modes  : 0 one wave per SIMD (idle matrix pipe)   1 two trial waves per SIMD
//            2..5 every other wave is a partner: VALU chain / fp32 MFMA chain / bf16 MFMA chain / LDS traffic
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_valu_mfma_hazard.hip -o /tmp/bench_vmh && /tmp/bench_vmh [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N15 "s_nop 15\n s_nop 15\n"
// compare v[72:87] with v[88:103] into v104
#define CMP                                                                                                           \
    "v_xor_b32 v104, v72, v88\n v_xor_b32 v105, v73, v89\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v74, v90\n"   \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v75, v91\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v76, v92\n"  \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v77, v93\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v78, v94\n"  \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v79, v95\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v80, v96\n"  \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v81, v97\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v82, v98\n"  \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v83, v99\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v84, v100\n" \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v85, v101\n v_or_b32 v104, v104, v105\n v_xor_b32 v105, v86, v102\n" \
    "v_or_b32 v104, v104, v105\n v_xor_b32 v105, v87, v103\n v_or_b32 v104, v104, v105\n"
#define CLOB                                                                                                              \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", \
        "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
        "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105"

// bf16 MFMA: A = v[64:67] (the register under test: v67), B = v[68:71]
#define TRIAL_BF(WR, NOPS)                                                                                            \
    asm volatile("v_mov_b32 v64, %[a0]\n v_mov_b32 v65, %[a1]\n v_mov_b32 v66, %[a2]\n v_mov_b32 v67, %[old]\n"       \
                 "v_mov_b32 v68, %[b0]\n v_mov_b32 v69, %[b1]\n v_mov_b32 v70, %[b2]\n v_mov_b32 v71, %[b3]\n" N15 N15 \
                 WR NOPS "v_mfma_f32_32x32x16_bf16 v[72:87], v[64:67], v[68:71], 0\n" N15 N15                        \
                 "v_mfma_f32_32x32x16_bf16 v[88:103], v[64:67], v[68:71], 0\n" N15 N15 CMP "v_mov_b32 %[flag], v104\n" \
                 : [flag] "=v"(flag)                                                                                  \
                 : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [old] "v"(old), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), \
                   [b3] "v"(b3), [x0] "v"(x0), [x1] "v"(x1), [sel] "s"(0x07060302u)                                   \
                 : CLOB)
// fp32 MFMA: A = v67, B = v68
#define TRIAL_F32(WR, NOPS)                                                                                           \
    asm volatile("v_mov_b32 v67, %[old]\n v_mov_b32 v68, %[b0]\n" N15 N15 WR NOPS                                     \
                 "v_mfma_f32_32x32x2_f32 v[72:87], v67, v68, 0\n" N15 N15 N15                                        \
                 "v_mfma_f32_32x32x2_f32 v[88:103], v67, v68, 0\n" N15 N15 N15 CMP "v_mov_b32 %[flag], v104\n"        \
                 : [flag] "=v"(flag)                                                                                  \
                 : [old] "v"(old), [b0] "v"(b0), [x0] "v"(x0), [x1] "v"(x1), [sel] "s"(0x07060302u)                   \
                 : CLOB)

#define W_PERM "v_perm_b32 v67, %[x1], %[x0], %[sel]\n"
#define W_SUB "v_sub_f32 v67, %[x0], %[x1]\n"
#define W_AND "v_and_b32 v67, 0xffff0000, %[x0]\n"
#define W_XOR "v_xor_b32 v67, %[x0], %[x1]\n"
#define W_MOV "v_mov_b32 v67, %[x0]\n"
#define D0 ""
#define D1 "s_nop 0\n"
#define D2 "s_nop 1\n"
#define D3 "s_nop 2\n"
#define D4 "s_nop 3\n"
#define D6 "s_nop 5\n"
#define D8 "s_nop 7\n"

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
// a pair of bf16 in [1, 2) with 3 random mantissa bits each (products of two: 8 bits; sums of 16: exact in fp32)
__device__ __forceinline__ unsigned bfpair(unsigned r) { return 0x3f803f80u | (r & 0x00700070u); }
// an fp32 in [1, 2) with 8 random mantissa bits (products 18 bits, sums of 2 + accumulate: exact)
__device__ __forceinline__ unsigned f32v(unsigned r) { return 0x3f800000u | (r & 0x007f8000u); }

__device__ int g_done;

template <int WR, int MF, int DIST>
__device__ __forceinline__ unsigned trial(unsigned& s) {
    unsigned flag = 0;
    if (MF == 0) {
        const unsigned a0 = bfpair(lcg(s)), a1 = bfpair(lcg(s)), a2 = bfpair(lcg(s)), old = bfpair(lcg(s));
        const unsigned b0 = bfpair(lcg(s)), b1 = bfpair(lcg(s)), b2 = bfpair(lcg(s)), b3 = bfpair(lcg(s));
        unsigned x0 = bfpair(lcg(s)), x1 = bfpair(lcg(s));
        if (WR == 1) { x0 = 0x40000000u | (x0 & 0x007f0070u); x1 = 0x3f800000u; }     // [2, 4) - 1: no NaN in either half
        if (WR == 3) x1 &= 0x00700070u;
#define GO(WS, DS) TRIAL_BF(WS, DS)
#define PICKD(WS)                                                                                  \
        if (DIST == 0) GO(WS, D0); else if (DIST == 1) GO(WS, D1); else if (DIST == 2) GO(WS, D2); \
        else if (DIST == 3) GO(WS, D3); else if (DIST == 4) GO(WS, D4); else if (DIST == 6) GO(WS, D6); else GO(WS, D8);
        if (WR == 0) { PICKD(W_PERM) } else if (WR == 1) { PICKD(W_SUB) } else if (WR == 2) { PICKD(W_AND) }
        else if (WR == 3) { PICKD(W_XOR) } else { PICKD(W_MOV) }
#undef GO
    } else {
        const unsigned old = f32v(lcg(s)), b0 = f32v(lcg(s));
        unsigned x0 = f32v(lcg(s)), x1 = f32v(lcg(s));
        if (WR == 1) x0 = (x0 & 0x007f8000u) | 0x40000000u;           // [2, 4) - [1, 2): exact, in (0, 3)
        if (WR == 3) x1 &= 0x007f8000u;
        if (WR == 0) { x0 = 0x80000000u; }                            // perm: upper half of x1 | upper half of x0
#define GO(WS, DS) TRIAL_F32(WS, DS)
        if (WR == 0) { PICKD(W_PERM) } else if (WR == 1) { PICKD(W_SUB) } else if (WR == 2) { PICKD(W_AND) }
        else if (WR == 3) { PICKD(W_XOR) } else { PICKD(W_MOV) }
#undef GO
#undef PICKD
    }
    return flag;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// partner workloads (mode 2..5), run until every trial wave is done
__device__ void partner(int mode, int ntrial, float* sink) {
    __shared__ float lds[1024];
    const int lane = threadIdx.x;
    float v = 1.0f + lane * 1e-3f, w = 0.5f;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)1.0f; pb[i] = (__bf16)0.5f; }
    for (int round = 0; round < (1 << 20); ++round) {
        if (__hip_atomic_load(&g_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ntrial) break;
        for (int k = 0; k < 64; ++k) {
            if (mode == 2) { v = fmaf(v, 0.999f, w); w = fmaf(w, 1.001f, -v * 1e-4f); v = fmaf(v, 0.5f, w); w = fmaf(w, 0.25f, v); }
            else if (mode == 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, w, acc, 0, 0, 0);
            else if (mode == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
            else { lds[(lane * 17 + k) & 1023] = v; v += lds[(lane * 5 + k * 3) & 1023]; }
        }
    }
    sink[blockIdx.x * 64 + lane] = v + w + acc[0] + acc[7];
}

template <int WR, int MF, int DIST>
__global__ void __launch_bounds__(64) hz(unsigned long long* out, int iters, int mode, int ntrial, float* sink) {
    if (mode >= 2 && (blockIdx.x & 1)) { partner(mode, ntrial, sink); return; }
    unsigned s = 0x9e3779b9u * (blockIdx.x * 64 + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) bad += trial<WR, MF, DIST>(s) != 0;
    for (int off = 32; off >= 1; off >>= 1) bad += __shfl_xor(bad, off);
    if (threadIdx.x == 0) {
        atomicAdd(out, bad);
        __hip_atomic_fetch_add(&g_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int WR, int MF, int DIST>
static void run(const char* wname, int iters, unsigned long long* d_out, float* sink) {
    static const char* mnames[6] = {"1 wave/SIMD", "2 trial waves/SIMD", "partner VALU", "partner f32 MFMA", "partner bf16 MFMA", "partner LDS"};
    printf("%-10s -> %-28s dist %d:", wname, MF == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", DIST);
    for (int mode = 0; mode < 6; ++mode) {
        const int grid = mode == 0 ? 1024 : 2048;
        const int ntrial = mode >= 2 ? grid / 2 : grid;
        int zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_done), &zero, sizeof(int));
        (void)hipMemset(d_out, 0, 8);
        hipLaunchKernelGGL((hz<WR, MF, DIST>), dim3(grid), dim3(64), 0, 0, d_out, iters, mode, ntrial, sink);
        hipError_t e = hipDeviceSynchronize();
        unsigned long long bad = 0;
        (void)hipMemcpy(&bad, d_out, 8, hipMemcpyDeviceToHost);
        printf("  [%s] %llu/%.1e%s", mnames[mode], bad, (double)ntrial * 64 * iters, e == hipSuccess ? "" : " HIP ERROR");
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* d_out;
    float* sink;
    (void)hipMalloc(&d_out, 8);
    (void)hipMalloc(&sink, 2048 * 64 * 4);
#define ALLD(WR, MF, NAME) run<WR, MF, 0>(NAME, iters, d_out, sink); run<WR, MF, 1>(NAME, iters, d_out, sink); run<WR, MF, 2>(NAME, iters, d_out, sink); \
    run<WR, MF, 4>(NAME, iters, d_out, sink); run<WR, MF, 8>(NAME, iters, d_out, sink);
    ALLD(0, 0, "v_perm_b32") ALLD(1, 0, "v_sub_f32") ALLD(2, 0, "v_and_b32") ALLD(3, 0, "v_xor_b32") ALLD(4, 0, "v_mov_b32")
    ALLD(0, 1, "v_perm_b32") ALLD(1, 1, "v_sub_f32") ALLD(3, 1, "v_xor_b32") ALLD(4, 1, "v_mov_b32")
    return 0;
}
