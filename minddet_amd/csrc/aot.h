// aot.h -- host-side helpers shared by every entry point of libminddet_hip.so.
// Argument checking for the MindSpore AOT-operator ABI (see include/minddet_hip.h) and
// stream-ordered scratch.  Mutable globals: the LDS-attribute cache and the per-stream scratch pool below, nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/minddet_hip.h"

#define MD_HIP_TRY(expr)                       \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) return MD_ERR_HIP; \
    } while (0)

// Device side: ordering point between LDS writes by SOME lanes of a wave and reads of the same bytes by OTHER lanes of that wave (the
// wave-private transpose slabs of conv_pingpong_kernel<PERS> and bottleneck64_kernel).  To the language these are different threads with no
// synchronisation between them, so hipcc may move -- or, around a divergent `if`, duplicate into the non-writing lanes and run first -- the
// reads across the writes (seen r03: stale rows in exactly those lanes).  A wavefront-scope release / acquire pair orders the memory
// operations, and wave_barrier is a convergent operation: it can be neither duplicated into nor moved across a divergent region.  All
// three lower to no instruction (one wave's LDS operations execute in issue order).  r03 used an empty asm with a memory clobber here,
// which holds only as long as hipcc treats that asm as convergent (r03 ADVICE).
#define MD_WAVE_LDS_ORDER()                                         \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

// Device side: every 128-bit buffer store goes through this macro: the store, then `s_nop 1` in an asm statement that takes the stored value
// as an in / out operand.
// Why (r04, the root cause of r03's "result that was wrong only sometimes"): gfx950 needs two wait states between a VMEM store of more than
// 64 bits and a VALU write of its data registers (the store reads them over several cycles).  hipcc inserts them (`s_nop 1`) -- EXCEPT when
// the store's soffset operand is an SGPR: LLVM's GCNHazardRecognizer::createsVALUHazard exempts that form ("this hazard only exists if the
// instruction is not using a register in the soffset field"), and on this chip the exemption is wrong.  `buffer_store_dwordx4 v[14:17], v84,
// s[8:11], s26 offen` directly followed by `v_lshlrev_b32 v14, 16, v86` stored the NEW v14 in the last four lanes of every 16 -- in some
// tiles of some runs, depending on how long the store waited to issue (profiles/r04_store_data_hazard.txt: reproduced, bisected to one
// not-even-executed branch that changed the schedule, confirmed by two independent fixes).  Any soffset that is not an inline constant is
// exposed: a wave-dependent value, but also a literal above 64, which LLVM materialises in an SGPR.  Whether a build is hit depends only on
// what the scheduler happens to put behind the store -- r03 "fixed" it by rewriting an unrelated expression.
// The guard: the value is live INTO the asm, so its registers cannot be redefined before it, and the asm itself is the two wait states;
// anything the scheduler puts between the store and the asm writes other registers.  No register, 8 bytes of code per store, and it does
// not depend on hipcc's hazard table.  tools/isa_audit.py additionally scans every wide store of every kernel for a VALU write of its
// data registers within two wait states.  V must be a non-const lvalue.
#define MD_BUFFER_STORE_B128(V, RSRC, VOFF, SOFF, AUX)                                   \
    do {                                                                                 \
        __builtin_amdgcn_raw_buffer_store_b128((V), (RSRC), (int)(VOFF), (SOFF), (AUX)); \
        asm volatile("s_nop 1" : "+v"(V));                                               \
    } while (0)

namespace md {

static inline bool dtype_is(const char **dtypes, int i, const char *want) {
    // dtypes may be NULL when the caller (e.g. a plain C test) does not describe tensors.
    if (!dtypes || !dtypes[i]) return true;
    return strcmp(dtypes[i], want) == 0;
}

static inline int64_t dim(int *ndims, int64_t **shapes, int i, int d) {
    if (!ndims || !shapes || !shapes[i]) return -1;
    if (d < 0) d += ndims[i];
    if (d < 0 || d >= ndims[i]) return -1;
    return shapes[i][d];
}

static inline int64_t numel(int *ndims, int64_t **shapes, int i) {
    if (!ndims || !shapes || !shapes[i]) return -1;
    int64_t n = 1;
    for (int d = 0; d < ndims[i]; ++d) n *= shapes[i][d];
    return n;
}

// pool buffers grow in 1 MiB steps, doubling: few regrowths even when the first calls are small
static inline size_t align_up_pow2_chunk(size_t b) {
    size_t w = (size_t)1 << 20;
    while (w < b) w <<= 1;
    return w;
}

// Scratch: a caller-provided workspace (params[ws_index], size from shapes) or, when the caller passes none, a buffer of the library's
// per-(device, stream) pool.  r03: the fallback used to be hipMallocAsync + hipFreeAsync per call, which blocks the host behind queued
// work (measured ~7 ms per call) -- a MindSpore-side caller binding the ops as INTEGRATION 1 shows would have paid it on
// every NMS / top-k call.  The pool keeps ONE buffer per stream, grown on demand (hipMallocAsync / hipFreeAsync on that stream, so a
// queued kernel never loses its memory) and reused by every later call on the stream: steady state makes no driver call at all.  Reuse
// is safe because an op's kernels and the next op's kernels on one stream execute in order.  md_scratch_release() frees the pool.
// Not for use under stream capture: a pool buffer that GROWS inside a captured region would be allocated by the graph -- a captured
// caller passes the workspace param (as the Python host always does).
struct ScratchPoolEntry { int dev; hipStream_t stream; void *ptr; size_t bytes; };
struct ScratchPool {
    ScratchPoolEntry e[64];
    int n = 0;
    int lock = 0;
};
inline ScratchPool g_scratch_pool;
static inline void *pool_acquire(size_t bytes, hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    ScratchPool &P = g_scratch_pool;
    while (__atomic_exchange_n(&P.lock, 1, __ATOMIC_ACQUIRE)) { }
    void *out = nullptr;
    int slot = -1;
    for (int i = 0; i < P.n; ++i)
        if (P.e[i].dev == dev && P.e[i].stream == s) { slot = i; break; }
    if (slot < 0 && P.n < 64) {
        slot = P.n++;
        P.e[slot] = {dev, s, nullptr, 0};
    }
    if (slot >= 0) {
        ScratchPoolEntry &en = P.e[slot];
        if (en.bytes < bytes) {
            const size_t want = align_up_pow2_chunk(bytes);
            void *np = nullptr;
            if (hipMallocAsync(&np, want, s) == hipSuccess) {
                if (en.ptr) (void)hipFreeAsync(en.ptr, s);   // behind every kernel already queued on s that uses it
                en.ptr = np;
                en.bytes = want;
            }
        }
        if (en.bytes >= bytes) out = en.ptr;
    }
    __atomic_store_n(&P.lock, 0, __ATOMIC_RELEASE);
    return out;
}
static inline int pool_release() {
    ScratchPool &P = g_scratch_pool;
    while (__atomic_exchange_n(&P.lock, 1, __ATOMIC_ACQUIRE)) { }
    int rc = MD_OK;
    // hipFree, not hipFreeAsync: a stream the pool remembers may have been destroyed by its owner since (its handle would be invalid);
    // hipFree waits for the device, which is what a caller releasing memory before unloading the library wants anyway
    for (int i = 0; i < P.n; ++i)
        if (P.e[i].ptr && hipFree(P.e[i].ptr) != hipSuccess) rc = MD_ERR_HIP;
    P.n = 0;
    __atomic_store_n(&P.lock, 0, __ATOMIC_RELEASE);
    return rc;
}
struct Scratch {
    void *ptr = nullptr;
    bool owned = false;     // true: more than 64 (device, stream) pairs are live -- a one-call stream-ordered allocation
    hipStream_t stream = nullptr;
    int acquire(size_t bytes, int nparam, void **params, int *ndims, int64_t **shapes, int ws_index,
                hipStream_t s) {
        stream = s;
        if (bytes == 0) bytes = 16;
        if (ws_index < nparam && params[ws_index]) {
            int64_t have = numel(ndims, shapes, ws_index);
            if (have >= 0 && (size_t)have < bytes) return MD_ERR_SIZE;
            ptr = params[ws_index];
            owned = false;
            return MD_OK;
        }
        ptr = pool_acquire(bytes, s);
        if (ptr) return MD_OK;
        if (hipMallocAsync(&ptr, bytes, s) != hipSuccess) return MD_ERR_HIP;
        owned = true;
        return MD_OK;
    }
    ~Scratch() {
        if (owned && ptr) (void)hipFreeAsync(ptr, stream);
    }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per launch: it is a driver call on the
// launch path of every layer otherwise (r02 ADVICE; the YOLOv5s step is launch-bound) and would also run inside graph capture.
// Cache keyed by the kernel's host address (lock-free insert; a full table or a device id >= 8 just sets the attribute again).  The
// set-and-store of one slot is serialised by the slot's lock, so the cached value is always the value the driver holds (r03 ADVICE: two
// host threads raising one kernel's size concurrently could otherwise leave the cache above the driver's value).  This cache is the
// library's only mutable global besides the per-stream scratch pool below; `inline` gives the whole library ONE instance.
struct LdsAttrSlot { const void *k; int set[8]; int lock; };
inline LdsAttrSlot g_lds_attr[128];
static inline int ensure_dyn_lds(const void *k, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MD_ERR_HIP;
    const unsigned h0 = (unsigned)(((uintptr_t)k >> 3) * 2654435761u >> 16);
    for (unsigned p = 0; p < 128 && dev >= 0 && dev < 8; ++p) {
        LdsAttrSlot &sl = g_lds_attr[(h0 + p) & 127];
        const void *cur = __atomic_load_n(&sl.k, __ATOMIC_ACQUIRE);
        if (cur == nullptr) {
            const void *expected = nullptr;
            cur = __atomic_compare_exchange_n(&sl.k, &expected, k, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) ? k : expected;
        }
        if (cur != k) continue;
        if (__atomic_load_n(&sl.set[dev], __ATOMIC_ACQUIRE) >= lds) return MD_OK;
        while (__atomic_exchange_n(&sl.lock, 1, __ATOMIC_ACQUIRE)) { }
        int rc = MD_OK;
        if (sl.set[dev] < lds) {
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) rc = MD_ERR_HIP;
            else __atomic_store_n(&sl.set[dev], lds, __ATOMIC_RELEASE);
        }
        __atomic_store_n(&sl.lock, 0, __ATOMIC_RELEASE);
        return rc;
    }
    return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? MD_OK : MD_ERR_HIP;
}

// records which conv-family kernel the calling host thread launched last (md_conv2d_last_kernel); defined in conv.hip
void md_note_conv_kernel(int id);
// activation bytes above which a conv-family op slices the batch: md_conv_tune.chunk_limit of the call (tests lower it), default 2 GiB - 64 KiB
static inline long long md_chunk_limit(const md_conv_tune *t) {
    return t && t->chunk_limit > 0 && t->chunk_limit < 0x7fff0000 ? (long long)t->chunk_limit : 0x7fff0000LL;
}

}  // namespace md
