// aot.h -- host-side helpers shared by every entry point of libminddet_hip.so.
// Argument checking for the MindSpore AOT-operator ABI (see include/minddet_hip.h) and
// stream-ordered scratch.  No global mutable state except the write-once LDS-attribute cache below.
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

// Scratch: a caller-provided workspace (params[ws_index], size from shapes) or a
// stream-ordered allocation released by the destructor (hipFreeAsync on the same stream).
struct Scratch {
    void *ptr = nullptr;
    bool owned = false;
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
// Write-once cache keyed by the kernel's host address (lock-free insert; a full table or a device id >= 8 just sets the attribute again).
struct LdsAttrSlot { const void *k; int set[8]; };
static LdsAttrSlot g_lds_attr[128];
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
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MD_ERR_HIP;
        __atomic_store_n(&sl.set[dev], lds, __ATOMIC_RELEASE);
        return MD_OK;
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
