// Micro-benchmark: LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU vs waves per CU and DMAs in flight
// per wave.  Each wave repeatedly issues `depth` 1-KiB DMAs from an L2-resident buffer, waits vmcnt(0), repeats.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip ; run: ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(1024) void dma_kernel(const char *src, unsigned bytes, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    unsigned off = (blockIdx.x * nw + wave) * 65536u + lane * 16u;
    char *dst = smem + wave * DEPTH * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + d * 1024), 16, (int)((off + d * 1024u) % bytes), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off += 8192u * 7u;
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(float *)smem;
}

template <int DEPTH>
float run(const char *src, unsigned bytes, int waves, int iters, float *sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute((const void *)dma_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, bytes, 10, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(256), dim3(waves * 64), lds, 0, src, bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const unsigned bytes = 16u << 20;  // 16 MiB: L2/IC resident
    char *src; float *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4096);
    const int iters = 2000;
    printf("waves/CU depth  GB/s/CU  B/clk@2.4  chip TB/s\n");
    for (int waves : {1, 2, 4, 8, 16}) {
        for (int depth : {1, 2, 4, 8, 16}) {
            if ((size_t)waves * depth * 1024 > 150 * 1024) continue;
            float ms = 0;
            switch (depth) {
                case 1: ms = run<1>(src, bytes, waves, iters, sink); break;
                case 2: ms = run<2>(src, bytes, waves, iters, sink); break;
                case 4: ms = run<4>(src, bytes, waves, iters, sink); break;
                case 8: ms = run<8>(src, bytes, waves, iters, sink); break;
                case 16: ms = run<16>(src, bytes, waves, iters, sink); break;
            }
            double per_cu = (double)waves * depth * 1024 * iters / (ms * 1e-3) / 1e9;
            printf("%5d %6d %9.1f %9.1f %9.2f\n", waves, depth, per_cu, per_cu / 2.4, per_cu * 256 / 1e3);
        }
    }
    return 0;
}
