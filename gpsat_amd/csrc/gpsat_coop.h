// gpsat_coop.h -- cooperative tiles: the control block through which the workgroup that owns a tile (the OWNER: it runs the
// optimiser, the diagonal chain of the sweep and the prediction) and the workgroups that help it (HELPERS: they pull groups
// of the owner's sweep and gradient queues) meet in device memory.  Included inside namespace gpsat.
//
// Visibility rules (MI355X_MICROARCH.md, "inter-workgroup visibility"; cdna_hip_programming.md Guideline 16): per-XCD L2s
// are not coherent with each other and a CU's L1 is never refreshed by another CU's stores, so
//   * every word of a CoopCtl is accessed by agent-scope atomics only (sc1 loads / stores, atomic adds), never plainly;
//   * every workspace byte another workgroup may read is stored sc1 (write-through) and loaded sc1 (buffer_load ... sc1
//     to registers: the kernels' ldg / stg always are);
//   * a flag or counter that announces data is written only after EVERY wave that stored that data has executed
//     s_waitcnt vmcnt(0); a wave that signals for itself does so right behind its own wait.
// Helpers are opportunistic: nothing waits FOR a helper.  Work is handed out by atomic queue heads, so an owner whose
// helpers never show up finishes its queues with its own waves; the one thing an owner waits for is that helpers which
// checked into a phase have checked out of it (`active`), and that wait -- like every spin here -- is bounded and ends in a
// failed evaluation instead of a hung GPU.
#ifndef GPSAT_COOP_H
#define GPSAT_COOP_H

typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) unsigned guint;
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) double gdouble;

enum { COOP_CLOSED = 0, COOP_SWEEP = 1, COOP_GRAD = 2, COOP_RELEASED = 3 };

struct CoopCtl {                  // 1 KiB per workgroup, zeroed by the host before every launch
    // Phase word and check-in count in ONE 64-bit word, changed by single atomics only:
    //   bits 63..32  (seq << 2) | kind (COOP_*); seq counts the phases this workgroup has opened as an owner
    //   bits 31..0   helper workgroups checked into the open phase
    // A helper checks in by compare-and-swap (count + 1 only while the phase half is what it read), the owner closes a phase by
    // an atomic AND on the kind bits and then waits for the count half of the SAME word: no pair of words whose accesses could
    // pass each other (round 3 had `phase` and `active` apart, relaxed: a store-buffering pattern).
    unsigned long long pa;
    int tile;                     // tile of the running optimisation
    int want_m;                   // the open sweep also builds M = L^-1 (gradient wanted)
    int score;                    // > 0: the running tile takes helpers (NB of the tile); 0: not now
    int hcap;                     // helpers wanted at most
    int pad0[2];
    double theta[8];              // parameters of the open evaluation
    // counters (agent-scope atomic adds / exchanges only)
    int helpers;                  // attached helper workgroups
    int unused0;
    int qhead;                    // group queue head of the open phase
    int done;                     // queue groups finished
    int fail;                     // the open evaluation has failed (not positive definite, or a spin gave up)
    // sweep flags (phase_pt): panels whose diagonal chain is complete / rows of panel s-1 of the columns of panel s in memory
    int ready, g0done;
    int pad1[1];
    int colrow[GPSAT_PT_MAXNB];   // per block column: panels whose rows are in memory
    // statistics of this workgroup (atomic adds; read by the host when GPSAT_DEBUG_COOP_STATS is set):
    // 0 cooperative evaluations as owner, 1 phases taken part in as helper, 2 groups run as helper, 3 flag waits that gave
    // up, 4 owner waits that gave up, 5 evaluations failed by a pivot, 6 helper loops unwound
    int stat[8];
    int pad2[256 - 32 - GPSAT_PT_MAXNB - 8];
};
static_assert(sizeof(CoopCtl) == 1024, "CoopCtl is one KiB");

typedef __attribute__((address_space(1))) CoopCtl gCoopCtl;

// generic -> global address space (the pointers come from kernel arguments: device memory)
__device__ __forceinline__ gfloat* as_gfloat(float* p) { return (gfloat*)p; }
__device__ __forceinline__ gCoopCtl* as_gctl(void* p) { return (gCoopCtl*)p; }

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_AGENT_CAS __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define PA_PHASE(v) ((unsigned)((v) >> 32))
#define PA_ACTIVE(v) ((unsigned)(v))
#define PA_MAKE(seq, kind) ((unsigned long long)(((unsigned)(seq) << 2) | (unsigned)(kind)) << 32)
#define COOP_STAT(ctl, i) __hip_atomic_fetch_add(&(ctl)->stat[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

// every storing wave, before the flag / counter that announces its stores
__device__ __forceinline__ void coop_drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float gld_f(const gfloat* p) {
    return __uint_as_float(__hip_atomic_load((const guint*)p, RLX_AGENT));
}
__device__ __forceinline__ void gst_f(gfloat* p, float v) {
    __hip_atomic_store((guint*)p, __float_as_uint(v), RLX_AGENT);
}
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ double gld_d(const gdouble* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const gu64*)p, RLX_AGENT));
}
__device__ __forceinline__ void gst_d(gdouble* p, double v) {
    __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), RLX_AGENT);
}

#endif
