// How fast can one CU pull L2-resident bytes into LDS, by method and by number of loader waves?
//   dma   : global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPRs)
//   reg   : global_load_dwordx4 -> ds_write_b128 (register staging)
//   ldonly: global_load_dwordx4 only (L2 -> L1 -> VGPR rate, the ceiling of both)
// One workgroup per CU (256 of them), W waves each; the waves of a workgroup walk a 64 KiB window of their own (16 MB over
// the chip: L2 resident after the first pass) DEPTH instructions deep.  Prints bytes / clk / CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/feed_probe.hip -o scripts/probes/feed_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void feed(const float *src, float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = src + (long)blockIdx.x * 16384 + lane * 4;                     // 64 KiB window per workgroup
    float *my = lds + wave * DEPTH * 256;                                             // DEPTH KiB of LDS per wave
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float *p = base + (((it * 16 + wave) * DEPTH * 256) & 16383);            // this wave's DEPTH KiB of the pass
        if (MODE == 3 || MODE == 4) {
            // one instruction = four 256-byte rows of a [rows][2304 float] matrix (row stride 9216 B), as the conv's
            // filter tile; MODE 4: the 16-byte chunks of a row permuted by XOR with (row & 15), as an LDS swizzle needs
            const int row4 = lane >> 4, pc = lane & 15;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int row = (wave * DEPTH + d) * 4 + row4;                       // < 16 * 8 * 4 = 512 rows
                const int ch = MODE == 4 ? (pc ^ (row & 15)) : pc;
                const float *q = src + (long)blockIdx.x * 16384 * 0 + (long)row * 2304 + ((it * 64) % 2304) + ch * 4;
                const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(my + d * 256));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(q), "s"(l) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(my + d * 256));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p + d * 256), "s"(l) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 r[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = *(const f32x4 *)(p + d * 256);
            if (MODE == 1) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) *(volatile f32x4 *)(my + d * 256 + lane * 4) = r[d];
            } else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc += r[d];
            }
        }
    }
    long t1 = __builtin_readcyclecounter();
    if (MODE == 2) out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    else out[blockIdx.x * 1024 + threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 1024] = (float)(t1 - t0);
}

template <int MODE, int DEPTH> void run(const float *src, float *out, int waves, const char *name)
{
    const int iters = 2000;
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute((const void *)feed<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(256), dim3(waves * 64), lds, 0, src, out, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(256), dim3(waves * 64), lds, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms, cyc; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&cyc, out + 256 * 1024, 4, hipMemcpyDeviceToHost);
    const double bytes = (double)waves * DEPTH * 1024 * iters;
    printf("%-7s depth %d  %2d waves/CU: %6.1f B/clk/CU  (%6.1f GB/s per CU, %5.2f TB/s chip)\n", name, DEPTH, waves, bytes / cyc,
           bytes / (ms * 1e6), bytes * 256 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *src, *out;
    hipMalloc(&src, 256L * 65536 + 65536); hipMemset(src, 0, 256L * 65536 + 65536);
    hipMalloc(&out, (256 * 1024 + 16) * 4);
    for (int w : {1, 2, 4, 8, 16}) run<0, 4>(src, out, w, "dma");
    for (int w : {4, 8, 16}) run<0, 8>(src, out, w, "dma");
    for (int w : {4, 8}) run<3, 2>(src, out, w, "dma-rows");
    for (int w : {8, 16}) run<3, 4>(src, out, w, "dma-rows");
    for (int w : {8, 16}) run<3, 8>(src, out, w, "dma-rows");
    for (int w : {8, 16}) run<4, 4>(src, out, w, "dma-rows-xor");
    for (int w : {4, 8}) run<4, 2>(src, out, w, "dma-rows-xor");
    for (int w : {4, 8}) run<0, 2>(src, out, w, "dma");
    for (int w : {1, 2, 4, 8, 16}) run<1, 4>(src, out, w, "reg");
    for (int w : {4, 8, 16}) run<1, 8>(src, out, w, "reg");
    for (int w : {1, 2, 4, 8, 16}) run<2, 4>(src, out, w, "ldonly");
    for (int w : {4, 8, 16}) run<2, 8>(src, out, w, "ldonly");
    return 0;
}
