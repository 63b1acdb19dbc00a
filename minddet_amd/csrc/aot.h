// aot.h -- host-side helpers shared by every entry point of libminddet_hip.so.
// Argument checking for the MindSpore AOT-operator ABI (see include/minddet_hip.h) and
// stream-ordered scratch.  No global mutable state.
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

// records which conv-family kernel the calling host thread launched last (md_conv2d_last_kernel); defined in conv.hip
void md_note_conv_kernel(int id);
// activation bytes above which a conv-family op slices the batch (md_conv2d_set_chunk_limit; 2 GiB - 64 KiB unless a test lowered it)
long long md_chunk_limit();

}  // namespace md
