// How fast can a CU fill LDS from L2-resident data?  The 256 x 256 x 64 implicit-GEMM k-loops (csrc/conv_bf16.hip tile 32, csrc/conv_pw_lp.hip
// phases 1 / 5) move 43-64 KiB per k-tile into LDS by LDS-DMA and run at ~55-62 % of the matrix pipe: is the fill the bound, and does the
// register path (global_load -> VGPR -> ds_write) have a different ceiling?
//   mode 0: LDS-DMA only (global_load_lds_dwordx4, 1 KiB per wave instruction)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: half the pieces by DMA, half through registers
//   mode 3: global_load_dwordx4 -> VGPR only (no LDS write)
// 16 wavefronts per workgroup, one workgroup per CU, PIECES KiB per wave and iteration, all of an iteration's loads in flight before the wait.
// build: hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip ; run: ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PIECES>
__global__ __launch_bounds__(1024) void fill(const u32x4 *__restrict__ src, long src_chunks, int iters, unsigned long long *cycles, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    auto dma = [&](const u32x4 *p, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(lds) : "memory", "m0");
    };
    // a workgroup walks its own 4 MiB window of the (L2 / MALL resident) source, PIECES KiB per wave per iteration
    // footprint = src_chunks x 16 bytes, shared by every workgroup (2 MiB: each XCD's L2 holds it; 64 MiB: the Infinity Cache does)
    long base = ((long)blockIdx.x * 4099 * 64) % src_chunks;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const u32x4 *p = src + (base + (long)it * 16 * PIECES * 64) % src_chunks + (wave * PIECES) * 64 + lane;
        u32x4 r[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const bool by_dma = MODE == 0 || (MODE == 2 && (i & 1) == 0);
            if (by_dma) dma(p + i * 64, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(((wave * PIECES + i) * 64) * 16)));
            else        r[i] = __builtin_nontemporal_load(p + i * 64);
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const bool by_dma = MODE == 0 || (MODE == 2 && (i & 1) == 0);
            if (!by_dma) {
                if (MODE == 3) acc += r[i][0];
                else smem[(wave * PIECES + i) * 64 + lane] = r[i];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    acc += smem[tid][0];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int PIECES>
void run(const u32x4 *src, long chunks, unsigned long long *cyc, unsigned *sink, const char *name)
{
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void *)fill<MODE, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * PIECES * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill<MODE, PIECES>), dim3(blocks), dim3(1024), 16 * PIECES * 1024, 0, src, chunks, iters, cyc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double bytes = (double)iters * 16 * PIECES * 1024;
        if (rep) printf("%-44s %2d KiB/iter: %6.1f B/clk/CU (median WG), %6.1f cycles per KiB, chip %.2f TB/s, %.0f us\n", name, 16 * PIECES,
                        bytes / (double)h[blocks / 2], (double)h[blocks / 2] / (iters * 16.0 * PIECES), bytes * blocks / (ms * 1e-3) / 1e12, ms * 1e3);
    }
}

int main()
{
    const long bytes = 64L << 20;                      // 64 MiB source: Infinity-Cache resident, each XCD's L2 sees its workgroups' windows
    u32x4 *src; unsigned long long *cyc; unsigned *sink;
    hipMalloc(&src, bytes); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes);
    for (long foot : {2L << 20, 16L << 20, 60L << 20}) {
        const long chunks = foot / 16;                    // (the buffer has 4 MiB of slack behind the footprint)
        printf("-- source footprint %ld MiB\n", foot >> 20);
        run<0, 4>(src, chunks, cyc, sink, "LDS-DMA");
        run<1, 4>(src, chunks, cyc, sink, "global_load -> VGPR -> ds_write_b128");
        run<2, 4>(src, chunks, cyc, sink, "half DMA, half through registers");
        run<3, 4>(src, chunks, cyc, sink, "global_load -> VGPR only");
        run<0, 2>(src, chunks, cyc, sink, "LDS-DMA");
        run<1, 2>(src, chunks, cyc, sink, "global_load -> VGPR -> ds_write_b128");
    }
    return 0;
}
