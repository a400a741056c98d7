// gpsat_ring.h -- the time-sliced tile queue of the persistent tile kernels (device side), shared by the fp32 and fp64 kernels.
#ifndef GPSAT_RING_H
#define GPSAT_RING_H
#include <hip/hip_runtime.h>
#include "gpsat_kernels.h"

namespace gpsat {

// ---------------------------------------------------------------------------------------------
// time slicing (KernelArgs::seg_cost > 0): the tiles circulate through a ring in device memory.  An entry is
// (sequence + 1) << 32 | resumed << 31 | tile; slot s lives at s & ring_mask.  The ring is larger than the number of tiles
// plus the number of workgroups, so a slot is never rewritten before the workgroup that claimed it has read it.
// One thread per workgroup calls these.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ int ring_pop(const KernelArgs& A) {
    const unsigned s = (unsigned)__hip_atomic_fetch_add(&A.ring_ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long* e = A.ring + (s & (unsigned)A.ring_mask);
    for (int spins = 0;; ++spins) {
        // relaxed sc1 polling, no acquire fence anywhere: a resumed tile's state is read with sc1 loads only (they bypass
        // this CU's L1), after the workgroup barrier that follows this poll (gp_tile_kernel)
        const unsigned long long v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == s + 1u) return (int)(unsigned)v;
        // nothing there yet: either a tile will be pushed back, or every tile is finished
        if (__hip_atomic_load(&A.ring_ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) return -1;
        // never reached by design (minutes of polling): a lost entry must not hang the GPU; the host sees unfinished tiles
        if (spins > (1 << 25)) return -1;
        __builtin_amdgcn_s_sleep(64);
    }
}

// The two halves of ring_pop for a workgroup that does something else while its slot is empty (cooperative tiles: it
// helps a running tile): claim a slot once, then look at it.  ring_look: the entry, -1 when every tile is finished,
// -2 when the slot is still empty.
static __device__ __forceinline__ unsigned ring_claim(const KernelArgs& A) {
    return (unsigned)__hip_atomic_fetch_add(&A.ring_ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static __device__ __forceinline__ int ring_look(const KernelArgs& A, unsigned s) {
    const unsigned long long v = __hip_atomic_load(A.ring + (s & (unsigned)A.ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == s + 1u) return (int)(unsigned)v;
    if (__hip_atomic_load(&A.ring_ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) return -1;
    return -2;
}

// entries pushed and not yet claimed (<= 0: whoever pops next waits): a tile that would be suspended now would only move to
// a workgroup that is waiting for it -- it may as well keep running where it is
static __device__ __forceinline__ int ring_waiting_tiles(const KernelArgs& A) {
    return __hip_atomic_load(&A.ring_ctl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
           __hip_atomic_load(&A.ring_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static __device__ __forceinline__ void ring_push(const KernelArgs& A, int tile) {
    const unsigned s = (unsigned)__hip_atomic_fetch_add(&A.ring_ctl[16], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long v = ((unsigned long long)(s + 1u) << 32) | 0x80000000ull | (unsigned)tile;
    __hip_atomic_store(A.ring + (s & (unsigned)A.ring_mask), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace gpsat
#endif
