// Developer (EXPERIMENTS.md E48): reproducer attempt for the one instruction site that loses a term in the 4-wave build
// beside bf16 MFMAs.  Offline analysis of 20 events (scripts/e48_event_analysis.py) shows: the forward-solve sum tp1 of the
// diagonal chain misses exactly ONE term, always register 7 of the operand block, in lanes 48-63, in an early k-step.  In the
// kernel's assembly that term is the LAST of three back-to-back in-place packed FMAs, directly in front of an fp32 MFMA:
//     v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]
//     s_waitcnt lgkmcnt(0)
//     v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]
//     s_nop 0
//     v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]      <- its low half (v76) is lost in lanes 48-63
//     v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]
// Here: "victim" waves run exactly this sequence (inline asm, same registers) inside a loop with outstanding buffer loads
// and LDS reads, and check v76 / v77 against the same sums made by scalar FMAs in a quiet section; "partner" waves on the
// same SIMDs run v_mfma_f32_32x32x16_bf16 streams with VALU work between them.
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_pkfma_waw.hip -o /tmp/bench_waw && /tmp/bench_waw [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ float lds_f[];

// VAR 0: the kernel's sequence; 1: s_nop 3 instead of s_nop 0; 2: s_nop 3 between the last packed FMA and the MFMA;
// 3: the three FMAs not in place (ping-pong destinations); 4: scalar v_fma_f32 pairs instead of packed
#define SEQ_HEAD                                                                                              \
    "v_mov_b32 v76, %[t1]\n v_mov_b32 v77, %[t0]\n"                                                           \
    "v_mov_b32 v134, %[a5]\n v_mov_b32 v135, %[b5]\n v_mov_b32 v100, %[a6]\n v_mov_b32 v101, %[b6]\n"         \
    "v_mov_b32 v80, %[a7]\n v_mov_b32 v81, %[b7]\n v_mov_b32 v131, %[m]\n"                                    \
    "ds_read2_b32 v[140:141], %[za] offset0:8 offset1:9\n ds_read2_b32 v[142:143], %[za] offset0:10 offset1:11\n" \
    "s_waitcnt lgkmcnt(1)\n"
#define SEQ_TAIL                                                                                              \
    "v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n" \
    "v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n" \
    "s_nop 15\n s_nop 15\n v_mov_b32 %[o1], v76\n v_mov_b32 %[o0], v77\n v_mov_b32 %[z5], v141\n v_mov_b32 %[z6], v142\n v_mov_b32 %[z7], v143\n"
#define CLOBS "v76", "v77", "v80", "v81", "v100", "v101", "v131", "v134", "v135", "v140", "v141", "v142", "v143", "v82", "v83", "v84", \
              "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v78", "v79", "memory"
#define OUTS [o1] "=v"(o1), [o0] "=v"(o0), [z5] "=v"(z5), [z6] "=v"(z6), [z7] "=v"(z7)
#define INS [t1] "v"(t1), [t0] "v"(t0), [a5] "v"(a5), [b5] "v"(b5), [a6] "v"(a6), [b6] "v"(b6), [a7] "v"(a7), [b7] "v"(b7), [m] "v"(mm), [za] "v"(zaddr)

// VAR 5: the whole head of the kernel's loop iteration: three fp32 MFMAs IN FLIGHT while the eight FMA pairs issue
#define FULL_CLOBS "v76", "v77", "v80", "v81", "v100", "v101", "v131", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
    "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97",                                   \
    "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161",                   \
    "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "memory"
template <int NOP>
__device__ __forceinline__ void seq_full(float t1, float t0, float a5, float b5, float a6, float b6, float a7, float b7, float mm, int zaddr,
                                         float& o1, float& o0, float (&z)[8]) {
#define FULL_BODY(NOPSTR)                                                                                                   \
    asm volatile("v_mov_b32 v76, %[t1]\n v_mov_b32 v77, %[t0]\n v_mov_b32 v134, %[a5]\n v_mov_b32 v135, %[b5]\n v_mov_b32 v100, %[a6]\n"     \
                 "v_mov_b32 v101, %[b6]\n v_mov_b32 v80, %[a7]\n v_mov_b32 v81, %[b7]\n v_mov_b32 v131, %[m]\n s_nop 7\n"                     \
                 "v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n"                                                    \
                 "ds_read2_b32 v[136:137], %[za] offset1:1\n ds_read2_b32 v[138:139], %[za] offset0:2 offset1:3\n"             \
                 "ds_read2_b32 v[140:141], %[za] offset0:8 offset1:9\n ds_read2_b32 v[142:143], %[za] offset0:10 offset1:11\n" \
                 "v_mfma_f32_32x32x2_f32 v[146:161], v131, v131, v[146:161]\n v_mfma_f32_32x32x2_f32 v[162:177], v131, v131, v[162:177]\n" \
                 "s_waitcnt lgkmcnt(3)\n"                                                                                      \
                 "v_fma_f32 v76, v134, v136, v76\n v_fma_f32 v77, v135, v136, v77\n v_fma_f32 v76, v100, v137, v76\n v_fma_f32 v77, v101, v137, v77\n" \
                 "s_waitcnt lgkmcnt(2)\n"                                                                                      \
                 "v_fma_f32 v76, v80, v138, v76\n v_fma_f32 v77, v81, v138, v77\n v_fma_f32 v76, v134, v139, v76\n v_fma_f32 v77, v135, v139, v77\n"   \
                 "s_waitcnt lgkmcnt(1)\n"                                                                                      \
                 "v_fma_f32 v76, v100, v140, v76\n v_fma_f32 v77, v101, v140, v77\n"                                           \
                 "v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"              \
                 "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]\n s_nop 0\n"                        \
                 "v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n" NOPSTR                                 \
                 "v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n v_mfma_f32_32x32x2_f32 v[146:161], v131, v131, v[146:161]\n" \
                 "v_mfma_f32_32x32x2_f32 v[162:177], v131, v131, v[162:177]\n v_mfma_f32_32x32x2_f32 v[82:97], v131, v131, v[82:97]\n" \
                 "v_mfma_f32_32x32x2_f32 v[146:161], v131, v131, v[146:161]\n v_mfma_f32_32x32x2_f32 v[162:177], v131, v131, v[162:177]\n" \
                 "s_nop 15\n s_nop 15\n v_mov_b32 %[o1], v76\n v_mov_b32 %[o0], v77\n v_mov_b32 %[z0], v136\n v_mov_b32 %[z1], v137\n"           \
                 "v_mov_b32 %[z2], v138\n v_mov_b32 %[z3], v139\n v_mov_b32 %[z4], v140\n v_mov_b32 %[z5], v141\n v_mov_b32 %[z6], v142\n v_mov_b32 %[z7], v143\n" \
                 : [o1] "=v"(o1), [o0] "=v"(o0), [z0] "=v"(z[0]), [z1] "=v"(z[1]), [z2] "=v"(z[2]), [z3] "=v"(z[3]), [z4] "=v"(z[4]), [z5] "=v"(z[5]), \
                   [z6] "=v"(z[6]), [z7] "=v"(z[7])                                                                           \
                 : INS : FULL_CLOBS)
    if (NOP) { FULL_BODY("s_nop 3\n"); } else { FULL_BODY(""); }
#undef FULL_BODY
}

// VAR 7: as VAR 5, with the kernel's register roles: v76 / v77 hold operand register 0 of the two blocks, the three MFMAs in
// flight read them as A / B, the first FMA pair overwrites them in place (running sums come in through v204 / v205)
__device__ __forceinline__ void seq_roles(float t1, float t0, float a5, float b5, float a6, float b6, float a7, float b7, float mm, int zaddr,
                                          float& o1, float& o0, float (&z)[8]) {
    asm volatile("v_mov_b32 v204, %[t1]\n v_mov_b32 v205, %[t0]\n v_mov_b32 v76, %[a5]\n v_mov_b32 v77, %[b5]\n v_mov_b32 v134, %[a5]\n v_mov_b32 v135, %[b5]\n"
                 "v_mov_b32 v100, %[a6]\n v_mov_b32 v101, %[b6]\n v_mov_b32 v80, %[a7]\n v_mov_b32 v81, %[b7]\n s_nop 7\n"
                 "v_mfma_f32_32x32x2_f32 v[82:97], v77, v77, v[82:97]\n"
                 "ds_read2_b32 v[136:137], %[za] offset1:1\n ds_read2_b32 v[138:139], %[za] offset0:2 offset1:3\n"
                 "ds_read2_b32 v[140:141], %[za] offset0:8 offset1:9\n ds_read2_b32 v[142:143], %[za] offset0:10 offset1:11\n"
                 "v_mfma_f32_32x32x2_f32 v[146:161], v77, v76, v[146:161]\n v_mfma_f32_32x32x2_f32 v[162:177], v76, v76, v[162:177]\n"
                 "s_waitcnt lgkmcnt(3)\n"
                 "v_fma_f32 v76, v76, v136, v204\n v_fma_f32 v77, v77, v136, v205\n v_fma_f32 v76, v100, v137, v76\n v_fma_f32 v77, v101, v137, v77\n"
                 "s_waitcnt lgkmcnt(2)\n"
                 "v_fma_f32 v76, v80, v138, v76\n v_fma_f32 v77, v81, v138, v77\n v_fma_f32 v76, v134, v139, v76\n v_fma_f32 v77, v135, v139, v77\n"
                 "s_waitcnt lgkmcnt(1)\n"
                 "v_fma_f32 v76, v100, v140, v76\n v_fma_f32 v77, v101, v140, v77\n"
                 "v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"
                 "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]\n s_nop 0\n"
                 "v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n"
                 "v_mfma_f32_32x32x2_f32 v[82:97], v135, v135, v[82:97]\n v_mfma_f32_32x32x2_f32 v[146:161], v135, v134, v[146:161]\n"
                 "v_mfma_f32_32x32x2_f32 v[162:177], v134, v134, v[162:177]\n v_mfma_f32_32x32x2_f32 v[82:97], v101, v101, v[82:97]\n"
                 "v_mfma_f32_32x32x2_f32 v[146:161], v101, v100, v[146:161]\n v_mfma_f32_32x32x2_f32 v[162:177], v100, v100, v[162:177]\n"
                 "s_nop 15\n s_nop 15\n v_mov_b32 %[o1], v76\n v_mov_b32 %[o0], v77\n v_mov_b32 %[z0], v136\n v_mov_b32 %[z1], v137\n"
                 "v_mov_b32 %[z2], v138\n v_mov_b32 %[z3], v139\n v_mov_b32 %[z4], v140\n v_mov_b32 %[z5], v141\n v_mov_b32 %[z6], v142\n v_mov_b32 %[z7], v143\n"
                 : [o1] "=v"(o1), [o0] "=v"(o0), [z0] "=v"(z[0]), [z1] "=v"(z[1]), [z2] "=v"(z[2]), [z3] "=v"(z[3]), [z4] "=v"(z[4]), [z5] "=v"(z[5]),
                   [z6] "=v"(z[6]), [z7] "=v"(z[7])
                 : INS : FULL_CLOBS, "v204", "v205");
}

template <int VAR>
__device__ __forceinline__ void seq(float t1, float t0, float a5, float b5, float a6, float b6, float a7, float b7, float mm, int zaddr,
                                    float& o1, float& o0, float& z5, float& z6, float& z7) {
    if (VAR == 0)
        asm volatile(SEQ_HEAD "v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"
                     "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]\n s_nop 0\n"
                     "v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n" SEQ_TAIL : OUTS : INS : CLOBS);
    else if (VAR == 1)
        asm volatile(SEQ_HEAD "v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"
                     "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]\n s_nop 3\n"
                     "v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n" SEQ_TAIL : OUTS : INS : CLOBS);
    else if (VAR == 2)
        asm volatile(SEQ_HEAD "v_pk_fma_f32 v[76:77], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"
                     "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[76:77] op_sel_hi:[1,0,1]\n s_nop 0\n"
                     "v_pk_fma_f32 v[76:77], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n s_nop 3\n" SEQ_TAIL : OUTS : INS : CLOBS);
    else if (VAR == 3)
        asm volatile(SEQ_HEAD "v_pk_fma_f32 v[78:79], v[134:135], v[140:141], v[76:77] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n"
                     "v_pk_fma_f32 v[76:77], v[100:101], v[142:143], v[78:79] op_sel_hi:[1,0,1]\n s_nop 0\n"
                     "v_pk_fma_f32 v[78:79], v[80:81], v[142:143], v[76:77] op_sel:[0,1,0]\n v_mov_b32 v76, v78\n v_mov_b32 v77, v79\n" SEQ_TAIL : OUTS : INS : CLOBS);
    else
        asm volatile(SEQ_HEAD "v_fma_f32 v76, v134, v141, v76\n v_fma_f32 v77, v135, v141, v77\n s_waitcnt lgkmcnt(0)\n"
                     "v_fma_f32 v76, v100, v142, v76\n v_fma_f32 v77, v101, v142, v77\n"
                     "v_fma_f32 v76, v80, v143, v76\n v_fma_f32 v77, v81, v143, v77\n" SEQ_TAIL : OUTS : INS : CLOBS);
}

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float rnd(unsigned& s) { return (float)((int)(lcg(s) >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }

// out[0..3] lanes with a wrong v76 (low half) by lane quarter, [4..7] wrong v77, [8] sequences checked (lanes), [9] of those: the
// wrong value equals the sum WITHOUT the last term
template <int VAR>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ big, unsigned long long* out, int* census, int iters, int force_kind,
                                            int prio, float* sink, int noise) {
    __shared__ int kind_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w, h = lane >> 5;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
        const int cu = (int)((xcc << 8) | ((hw >> 8) & 0xff));
        kind_s = force_kind >= 0 ? force_kind : (atomicAdd(&census[cu], 1) & 1);
    }
    for (int i = threadIdx.x; i < 4096; i += 256) lds_f[i] = 0.25f + 0.001f * (float)(i % 97);
    if (threadIdx.x < 8) lds_f[17000 + threadIdx.x] = 0.f;
    if (threadIdx.x == 0) lds_f[18000] = 0.f;
    __syncthreads();
    if (kind_s == 0 && noise && w > 0) {
        // sibling waves of the chain wave in the real kernel: LDS block writes / reads (parked k-loops, factor copies), flag
        // polling with s_sleep, fp32 MFMAs, workspace loads and stores -- until wave 0 of the workgroup is done
        typedef float f4 __attribute__((ext_vector_type(4)));
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0, 0x7fffffff, 0x00020000);
        f4* area = reinterpret_cast<f4*>(lds_f + 8192 + 2048 * (w - 1));
        volatile float* stop = lds_f + 18000;
        float v = 1.0f + lane;
        for (int round = 0; round < (1 << 22); ++round) {
            if (*stop != 0.f) break;
            const f4 x = {v, v + 1.f, v + 2.f, v + 3.f};
            area[lane] = x; area[64 + lane] = x; area[128 + lane] = x; area[192 + lane] = x;
            const f4 a = area[(lane + 7) & 63], b = area[64 + ((lane + 13) & 63)];
            v = a[0] + b[1];
            for (int spin = 0; spin < 3; ++spin) { if (lds_f[17000 + w] > 1e30f) break; __builtin_amdgcn_s_sleep(2); }
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 0.5f, acc, 0, 0, 0);
            const int so = (int)(((unsigned)(round * 7919 + gw * 104729)) % 60000u) * 4096;
            u32x4 l0 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so, 16), l1 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + 1024, so, 16);
            v += __uint_as_float(l0[0]) * 1e-30f + __uint_as_float(l1[1]) * 1e-30f;
        }
        sink[(size_t)gw * 64 + lane] = v + acc[3];
        return;
    }
    if (kind_s == 0) {
        unsigned s = 0x9e3779b9u * (gw * 64 + lane + 1), su = 0x7f4a7c15u * (gw + 1);      // su: wave-uniform
        unsigned long long bad1 = 0, bad0 = 0, lost = 0;
        if (prio) __builtin_amdgcn_s_setprio(3);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0, 0x7fffffff, 0x00020000);
        float keep = 0.f;
        for (int it = 0; it < iters; ++it) {
            const float t1 = rnd(s), t0 = rnd(s), a5 = rnd(s), b5 = rnd(s), a6 = rnd(s), b6 = rnd(s), a7 = rnd(s), b7 = rnd(s), mm = rnd(s);
            // outstanding loads that return during the sequence (far apart: they miss the caches)
            const int so = (int)((lcg(su) >> 8) % 60000u) * 4096;
            u32x4 ld[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ld[q] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + 1024 * (q & 3), so + 4096 * (q >> 2), 16);
            const int zaddr = (int)((32 * (it & 63) + 4 * h) * 4);
            float o1, o0, z5, z6, z7, e1, e0, l1;
            if (VAR >= 5) {
                float z[8];
                if (VAR == 7) seq_roles(t1, t0, 1e-3f * a5, 1e-3f * b5, 1e-3f * a6, 1e-3f * b6, 1e-3f * a7, 1e-3f * b7, mm, zaddr, o1, o0, z);
                else seq_full<(VAR - 5) & 1>(t1, t0, a5, b5, a6, b6, a7, b7, 1e-3f * mm, zaddr, o1, o0, z);
                const float sc = VAR == 7 ? 1e-3f : 1.f;
                const float A[8] = {sc * a5, sc * a6, sc * a7, sc * a5, sc * a6, sc * a5, sc * a6, sc * a7},
                            B[8] = {sc * b5, sc * b6, sc * b7, sc * b5, sc * b6, sc * b5, sc * b6, sc * b7};
                e1 = t1; e0 = t0; l1 = t1;
#pragma unroll
                for (int q = 0; q < 8; ++q) { e1 = fmaf(A[q], z[q], e1); e0 = fmaf(B[q], z[q], e0); if (q < 7) l1 = fmaf(A[q], z[q], l1); }
            } else {
                seq<VAR>(t1, t0, a5, b5, a6, b6, a7, b7, mm, zaddr, o1, o0, z5, z6, z7);
                // reference in a quiet section: scalar FMAs in the same order
                e1 = fmaf(a7, z7, fmaf(a6, z6, fmaf(a5, z5, t1))); e0 = fmaf(b7, z7, fmaf(b6, z6, fmaf(b5, z5, t0)));
                l1 = fmaf(a6, z6, fmaf(a5, z5, t1));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) keep += __uint_as_float(ld[q][q & 3]) * 1e-30f;
            if (__float_as_uint(o1) != __float_as_uint(e1)) { ++bad1; if (__float_as_uint(o1) == __float_as_uint(l1)) ++lost; }
            if (__float_as_uint(o0) != __float_as_uint(e0)) ++bad0;
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        atomicAdd(&out[lane >> 4], bad1);
        atomicAdd(&out[4 + (lane >> 4)], bad0);
        atomicAdd(&out[8], (unsigned long long)iters);
        atomicAdd(&out[9], lost);
        sink[(size_t)gw * 64 + lane] = keep;
        if (noise && lane == 0) lds_f[18000] = 1.f;
    } else {
        // partner: bf16 MFMA stream with VALU work (perm / and / sub: the plane split) and loads in between
        f32x16 acc[2];
        for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        unsigned su = 0x85ebca6bu * (gw + 1);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0, 0x7fffffff, 0x00020000);
        for (int it = 0; it < iters * 3; ++it) {
            const int so = (int)((lcg(su) >> 8) % 60000u) * 4096;
            u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so, 16), x1 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + 1024, so, 16);
            u32x4 p0, p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p0[j] = __builtin_amdgcn_perm(x1[j], x0[j], 0x07060302u);
                const unsigned y0 = __float_as_uint(__uint_as_float(x0[j]) - __uint_as_float(x0[j] & 0xffff0000u));
                const unsigned y1 = __float_as_uint(__uint_as_float(x1[j]) - __uint_as_float(x1[j] & 0xffff0000u));
                p1[j] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                p2[j] = (y0 & 0xffff0000u) | (y1 >> 16);
                p3[j] = p0[j] ^ p1[j];
            }
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p0), __builtin_bit_cast(bf16x8, p1), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p2), __builtin_bit_cast(bf16x8, p3), acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p1), __builtin_bit_cast(bf16x8, p2), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p3), __builtin_bit_cast(bf16x8, p0), acc[1], 0, 0, 0);
            }
        }
        sink[(size_t)gw * 64 + lane] = acc[0][0] + acc[1][3];
    }
}

template <int VAR>
static void run(const char* name, const float* big, unsigned long long* out, int* census, int iters, float* sink, int noise = 0) {
    const char* modes[4] = {"beside bf16 partner, prio 3", "beside bf16 partner, prio 0", "victims only, prio 3", "victims only, prio 0"};
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(census, 0, 4096 * 4);
        (void)hipMemset(out, 0, 128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), 72 * 1024, 0, big, out, census, iters, mode >= 2 ? 0 : -1, (mode & 1) ? 0 : 1, sink, noise);
        hipError_t e = hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
        printf("%-44s %-30s: sequences (lanes) %llu; low half wrong by lane quarter [%llu %llu %llu %llu] (= sum without the last term: %llu); high half wrong [%llu %llu %llu %llu]%s\n",
               name, modes[mode], h[8], h[0], h[1], h[2], h[3], h[9], h[4], h[5], h[6], h[7], e == hipSuccess ? "" : " HIP ERROR");
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    float *big, *sink;
    unsigned long long* out;
    int* census;
    (void)hipMalloc(&big, (size_t)60010 * 4096 + 8192);
    (void)hipMemset(big, 0, (size_t)60010 * 4096 + 8192);
    (void)hipMalloc(&sink, (size_t)512 * 4 * 64 * 4);
    (void)hipMalloc(&out, 128);
    (void)hipMalloc(&census, 4096 * 4);
    run<7>("register roles, sibling waves make LDS/MFMA/VMEM noise", big, out, census, iters * 4, sink, 1);
    run<7>("loop head with the kernel's register roles", big, out, census, iters, sink);
    run<5>("whole loop head: 3 MFMAs in flight + 8 FMA pairs", big, out, census, iters, sink);
    run<6>("the same, s_nop 3 behind the last packed FMA", big, out, census, iters, sink);
    run<0>("the kernel's sequence", big, out, census, iters, sink);
    run<1>("s_nop 3 between the 2nd and 3rd packed FMA", big, out, census, iters, sink);
    run<2>("s_nop 3 between the 3rd packed FMA and the MFMA", big, out, census, iters, sink);
    run<3>("destinations not in place", big, out, census, iters, sink);
    run<4>("scalar v_fma_f32 pairs", big, out, census, iters, sink);
    return 0;
}
