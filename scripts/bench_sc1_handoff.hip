// Developer (EXPERIMENTS.md E48): minimal reproducer of the hand-off the dataflow sweep of gpsat_kernels.hip is built on.
// In every workgroup (two per CU, like the 4-wave build) wave 0 stores a 4-KiB block into the workgroup's own slab
// (4 x buffer_store_dwordx4, cache policy ST), drains (s_waitcnt vmcnt(0)), raises a flag in LDS; wave 1 has been spinning on
// that flag and loads the block at once (4 x buffer_load_dwordx4, cache policy LD) and checks every word against the stamp of
// this iteration.  Waves 2 and 3 make background traffic (stream the slab) or run bf16 MFMAs.  A stale word = the block's
// previous contents (an older stamp).  Reported per lane quarter (16 lanes x 16 B = 256 B of each 1-KiB row).
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_sc1_handoff.hip -o /tmp/bench_ho && /tmp/bench_ho [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ int lds_i[];

template <int AUX>
__device__ __forceinline__ void ld_blk(const unsigned* ws, int blk, int lane, u32x4 (&v)[4]) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(ws), 0, 0x7fffffff, 0x00020000);
    const int so = blk * 4096, vo = lane * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 1024 * q, so, AUX);
}
template <int AUX>
__device__ __forceinline__ void st_blk(unsigned* ws, int blk, int lane, const u32x4 (&v)[4]) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
    const int so = blk * 4096, vo = lane * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(v[q], r, vo + 1024 * q, so, AUX);
}

__device__ __forceinline__ unsigned stamp(int it, int q, int e, int lane) { return ((unsigned)it << 12) | (q << 10) | (e << 8) | lane; }

// out[0..3]: stale words per lane quarter; out[4]: hand-offs with any stale word; out[5]: stale words whose stamp is not the
// block's previous one (garbage, never expected); out[6]: hand-offs checked
template <int ST, int LD, int BG, int DELAY>
__global__ void __launch_bounds__(256, 2) ho(unsigned* slabs, unsigned long long* out, int iters, int nblk, float* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned* ws = slabs + (size_t)blockIdx.x * nblk * 1024;
    volatile int* flag = lds_i;          // [0] produced, [1] consumed, [2] stop
    if (threadIdx.x == 0) { lds_i[0] = 0; lds_i[1] = 0; lds_i[2] = 0; }
    __syncthreads();
    if (w == 0) {
        for (int it = 1; it <= iters; ++it) {
            const int b = (it * 37) % nblk;
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] = stamp(it, q, e, lane);
            st_blk<ST>(ws, b, lane, v);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (DELAY) __builtin_amdgcn_s_sleep(DELAY);
            if (lane == 0) __hip_atomic_store(&lds_i[0], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // the next store may go on at once: it is another block; a block is rewritten only nblk / gcd iterations later,
            // after the consumer has long checked it (the consumer acknowledges: at most 8 hand-offs ahead)
            while (__hip_atomic_load(&lds_i[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it - 8) __builtin_amdgcn_s_sleep(1);
        }
    } else if (w == 1) {
        unsigned long long stale[4] = {0, 0, 0, 0}, bad_ho = 0, garbage = 0;
        for (int it = 1; it <= iters; ++it) {
            const int b = (it * 37) % nblk;
            while (__hip_atomic_load(&lds_i[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            u32x4 v[4];
            ld_blk<LD>(ws, b, lane, v);
            int nbad = 0, ngar = 0;
            const int prev = it - nblk;              // 37 and nblk are coprime: the block's previous stamp
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned got = v[q][e];
                    if (got != stamp(it, q, e, lane)) { ++nbad; if (prev < 1 || got != stamp(prev, q, e, lane)) ++ngar; }
                }
            stale[lane >> 4] += nbad;
            garbage += ngar;
            const unsigned long long any = __ballot(nbad != 0);
            if (lane == 0 && any) ++bad_ho;
            if (lane == 0) __hip_atomic_store(&lds_i[1], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        atomicAdd(&out[lane >> 4], stale[lane >> 4]);
        atomicAdd(&out[5], garbage);
        if (lane == 0) { atomicAdd(&out[4], bad_ho); atomicAdd(&out[6], (unsigned long long)iters); lds_i[2] = 1; }
    } else {
        // background: until the consumer is done
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        bf16x8 pa, pb;
        for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)1.0f; pb[i] = (__bf16)0.5f; }
        unsigned sum = 0;
        // background traffic reads ANOTHER region (the second half of the allocation): it never touches hand-off blocks
        const unsigned* bgws = slabs + (size_t)(gridDim.x + blockIdx.x) * nblk * 1024;
        for (int round = 0; round < (1 << 24); ++round) {
            if (flag[2]) break;
            if (BG == 1) {
                u32x4 v[4];
                ld_blk<16>(bgws, (round * 2 + (w & 1)) % nblk, lane, v);
                sum += v[0][0] + v[1][1] + v[2][2] + v[3][3];
            } else if (BG == 2) {
                for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
            } else if (BG == 3) {
                for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 0.5f, acc, 0, 0, 0);
            } else {
                __builtin_amdgcn_s_sleep(32);
            }
        }
        sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[5] + (float)sum;
    }
}

template <int ST, int LD, int BG, int DELAY>
static void run(const char* name, unsigned* slabs, unsigned long long* d_out, float* sink, int iters, int nblk) {
    (void)hipMemset(d_out, 0, 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ho<ST, LD, BG, DELAY>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL((ho<ST, LD, BG, DELAY>), dim3(512), dim3(256), 72 * 1024, 0, slabs, d_out, iters, nblk, sink);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
    printf("%-58s hand-offs %llu  with stale words %llu  stale words by lane quarter [%llu %llu %llu %llu]  not-the-previous-stamp %llu%s\n", name, h[6],
           h[4], h[0], h[1], h[2], h[3], h[5], e == hipSuccess ? "" : "  HIP ERROR");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int nblk = 251;                 // blocks per slab (prime: every block is revisited every 251 iterations), ~1 MiB
    unsigned* slabs;
    unsigned long long* d_out;
    float* sink;
    (void)hipMalloc(&slabs, (size_t)2 * 512 * nblk * 4096);
    (void)hipMemset(slabs, 0, (size_t)2 * 512 * nblk * 4096);
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&sink, 512 * 256 * 4);
#define BGS(ST, LD, DELAY, NAME)                                                               \
    run<ST, LD, 0, DELAY>(NAME ", other waves idle", slabs, d_out, sink, iters, nblk);        \
    run<ST, LD, 1, DELAY>(NAME ", other waves stream sc1 loads", slabs, d_out, sink, iters, nblk); \
    run<ST, LD, 2, DELAY>(NAME ", other waves bf16 MFMA", slabs, d_out, sink, iters, nblk);   \
    run<ST, LD, 3, DELAY>(NAME ", other waves fp32 MFMA", slabs, d_out, sink, iters, nblk);
    BGS(16, 16, 0, "store sc1, load sc1")
    BGS(0, 0, 0, "store plain, load plain")
    BGS(16, 0, 0, "store sc1, load plain")
    BGS(0, 16, 0, "store plain, load sc1")
    BGS(17, 17, 0, "store sc0 sc1, load sc0 sc1")
    BGS(16, 16, 8, "store sc1, load sc1, s_sleep 8 before the flag")
    BGS(16, 16, 64, "store sc1, load sc1, s_sleep 64 before the flag")
    return 0;
}
