// HBM ceiling probe for this box: read-only, copy and an 80/20 read/write mix (the GroupDW byte mix),
// with plain 16-byte loads and with LDS-DMA.  hipcc --offload-arch=gfx950 -O3 bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f4 *x, float *sink, long n4)
{
    f4 a = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) a += x[i];
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[0] = a[0];
}
__global__ __launch_bounds__(256) void k_read_u4(const f4 *x, float *sink, long n4)
{
    f4 a = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
        a += v0 + v1 + v2 + v3;
    }
    for (; i < n4; i += stride) a += x[i];
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[0] = a[0];
}
__global__ __launch_bounds__(256) void k_copy(const f4 *x, f4 *y, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = x[i];
}
// 4 reads : 1 write
__global__ __launch_bounds__(256) void k_mix(const f4 *x, f4 *y, long n4out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4out; i += (long)gridDim.x * 256)
        y[i] = x[i] + x[i + n4out] + x[i + 2 * n4out] + x[i + 3 * n4out];
}
// LDS-DMA streaming read: each wave pulls 1 KiB pieces into a private LDS ring, DEPTH in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void k_dma(const f4 *x, float *sink, long n4)
{
    __shared__ f4 ring[4][DEPTH][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + wave;
    const long pieces = n4 / 64;
    int slot = 0;
    float acc = 0.f;
    for (long p = w; p < pieces; p += nwaves) {
        const f4 *src = x + p * 64 + lane;
        const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&ring[wave][slot][0]);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = ring[wave][0][lane][0];
    if (acc == 123.456f) sink[0] = acc;
}

// 4 reads : 1 write of ONE interleaved stream: a block owns a contiguous span; element i of the
// output reads input elements 4i..4i+3 (adjacent 64 B) -> one read stream + one write stream
template <bool NT>
__global__ __launch_bounds__(256) void k_mix_span(const f4 *x, f4 *y, long n4out, long span)
{
    for (long base = (long)blockIdx.x * span; base < n4out; base += (long)gridDim.x * span) {
        const long end = base + span < n4out ? base + span : n4out;
        for (long i = base + threadIdx.x; i < end; i += 256) {
            const f4 *q = x + ((i & ~63L) * 4 + (i & 63));
            f4 v = q[0] + q[64] + q[128] + q[192];
            if (NT) __builtin_nontemporal_store(v, y + i); else y[i] = v;
        }
    }
}

// the GroupDW loader's read pattern: a block (one loader wave) streams 1 KiB DMA pieces; QUARTER:
// a piece = 4 pixels x 256 B at a 1 KB pixel stride (one 64-channel group of 256-channel pixels), the
// block's group = blockIdx & 3; else a piece = 1 KiB contiguous.  `depth` pieces in flight.
template <int DEPTH, bool QUARTER>
__global__ __launch_bounds__(64) void k_dma_pix(const f4 *x, float *sink, long npix, long pix_per_block)
{
    __shared__ f4 ring[DEPTH][64];
    const int lane = threadIdx.x;
    const int q = QUARTER ? (blockIdx.x & 3) : 0;
    const long blk = QUARTER ? (blockIdx.x >> 2) : blockIdx.x;
    const long nblk = QUARTER ? (gridDim.x >> 2) : gridDim.x;
    int slot = 0;
    for (long p0 = blk * pix_per_block; p0 < npix; p0 += nblk * pix_per_block) {
        const long steps = QUARTER ? pix_per_block / 4 : pix_per_block;     // pieces in this span
        for (long k = 0; k < steps; ++k) {
            if ((QUARTER ? p0 + k * 4 + 3 : p0 + k) >= npix) break;
            const f4 *src = QUARTER ? x + (p0 + k * 4 + (lane >> 4)) * 64 + q * 16 + (lane & 15)
                                    : x + (p0 + k) * 64 + lane;
            const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&ring[slot][0]);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
            slot = slot + 1 == DEPTH ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DEPTH - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float acc = ring[0][lane][0];
    if (acc == 123.456f) sink[0] = acc;
}

// Epilogue access patterns of the 1x1 expansion convs: a 64-row x 2 KB tile per 512-thread block, read R + write Y.
// FRAG: the MFMA accumulator layout (a wave owns a 256-byte column band; one instruction = 16 rows x four 16-byte
// pieces at a 32-byte stride); else one instruction = 1 KiB contiguous (half a row).
template <bool FRAG>
__global__ __launch_bounds__(512) void k_epi(const f4 *r, f4 *y, long ntiles)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long base = t * 64 * 128;            // f4 units: 64 rows x 128 f4 (2 KB)
        if (FRAG) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const long o = base + (long)(j * 16 + l15) * 128 + wave * 16 + ps * 8 + q * 2 + k;
                        y[o] = r[o] + f4{1.f, 1.f, 1.f, 1.f};
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long o = base + i * 512 + tid;
                y[o] = r[o] + f4{1.f, 1.f, 1.f, 1.f};
            }
        }
    }
}

template <typename F> float timeit(F f, int it = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

int main()
{
    const long bytes = 1L << 30;                 // 1 GiB source (beyond the 256 MiB Infinity Cache)
    const long n4 = bytes / 16;
    f4 *x, *y; float *sink;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&sink, 64);
    hipMemset(x, 0x3c, bytes); hipMemset(y, 0, bytes);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
        printf("read  f4        blocks %5d: %7.1f us  %7.1f GB/s\n", blocks, ms * 1e3, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_read_u4, dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
        printf("read  f4 x4     blocks %5d: %7.1f us  %7.1f GB/s\n", blocks, ms * 1e3, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, x, y, n4 / 2); });
        printf("copy  f4        blocks %5d: %7.1f us  %7.1f GB/s (read+write)\n", blocks, ms * 1e3, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, x, y, n4 / 4); });
        printf("mix 4r:1w f4    blocks %5d: %7.1f us  %7.1f GB/s (read+write)\n", blocks, ms * 1e3, (bytes + bytes / 4) / ms / 1e6);
    }
    for (int blocks : {256, 512, 1024, 2048})
        for (long span : {256L, 1024L, 4096L, 16384L}) {
            float ms = timeit([&] { hipLaunchKernelGGL(k_mix_span<false>, dim3(blocks), dim3(256), 0, 0, x, y, n4 / 4, span); });
            float ms2 = timeit([&] { hipLaunchKernelGGL(k_mix_span<true>, dim3(blocks), dim3(256), 0, 0, x, y, n4 / 4, span); });
            printf("mixspan blocks %5d span %6ld f4: %7.1f GB/s   nt: %7.1f GB/s\n", blocks, span, (bytes + bytes / 4) / ms / 1e6, (bytes + bytes / 4) / ms2 / 1e6);
        }
    {
        const long ntiles = (bytes / 2) / (64 * 2048);      // R and Y of 512 MiB each
        for (int blocks : {256, 512, 1024}) {
            float a = timeit([&] { hipLaunchKernelGGL((k_epi<true>), dim3(blocks), dim3(512), 0, 0, x, y, ntiles); });
            float b = timeit([&] { hipLaunchKernelGGL((k_epi<false>), dim3(blocks), dim3(512), 0, 0, x, y, ntiles); });
            printf("epilogue pattern R+Y, %4d blocks: accumulator-layout pieces %7.1f GB/s   contiguous KiB %7.1f GB/s (read+write)\n",
                   blocks, bytes / a / 1e6, bytes / b / 1e6);
        }
    }
    {
        const long npix = bytes / 1024;           // 1 KB pixels
        for (int blocks : {256, 512, 1024}) {
            float a = timeit([&] { hipLaunchKernelGGL((k_dma_pix<46, false>), dim3(blocks), dim3(64), 0, 0, x, sink, npix, 116L); });
            float b = timeit([&] { hipLaunchKernelGGL((k_dma_pix<46, true>), dim3(blocks), dim3(64), 0, 0, x, sink, npix, 116L); });
            float c = timeit([&] { hipLaunchKernelGGL((k_dma_pix<46, true>), dim3(blocks), dim3(64), 0, 0, x, sink, npix, 29L * 29 * 4 / 4 * 4); });
            printf("loader-pattern read, %4d single-wave blocks, 46 in flight: contiguous %7.1f  quarter(116 px spans) %7.1f  quarter(3364 px spans) %7.1f GB/s\n",
                   blocks, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6);
        }
    }
    for (int blocks : {256, 512, 1024, 2048}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_dma<8>, dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
        printf("read  lds-dma d8  blocks %5d: %7.1f us  %7.1f GB/s\n", blocks, ms * 1e3, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_dma<16>, dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
        printf("read  lds-dma d16 blocks %5d: %7.1f us  %7.1f GB/s\n", blocks, ms * 1e3, bytes / ms / 1e6);
    }
    float ms = timeit([&] { hipMemcpyAsync(y, x, bytes / 2, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D: %7.1f us  %7.1f GB/s (read+write)\n", ms * 1e3, bytes / ms / 1e6);
    ms = timeit([&] { hipMemsetAsync(y, 0, bytes, 0); });
    printf("hipMemset: %7.1f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
    return 0;
}
