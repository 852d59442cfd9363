// HBM ceiling probes of the box at hand, callable from bench.py: what a kernel with GroupDW's byte mix (4 bytes read per
// byte written; csrc/xcorr.hip) can reach at all on this GPU, so that `xcorr_hbm.achieved` is quoted against a measured
// ceiling and not only against the 8 TB/s specification.  Three access patterns, 16 bytes per lane:
//   mode 0  read-only                  sum of src
//   mode 1  copy                       dst[i] = src[i]
//   mode 2  4 reads : 1 write          ONE interleaved read stream (output element i reads the four adjacent 1 KiB rows
//                                      4i .. 4i+3 of its 64-element group) + non-temporal stores: the best of the variants
//                                      measured in scripts/probes/bw_probe.hip (separate read streams: 4.8-5.2 TB/s; this: 5.5-6.1)
//   mode 3  GroupDW's TRAFFIC, no compute  an address-level emulation of groupdw_dma_kernel's launch (csrc/xcorr.hip): a block =
//                                      (sample, 64-channel group) reads row after row of three 29 x 29 x 256-channel fp32 maps - its
//                                      256-byte quarter of every 1 KiB pixel row - and writes its quarter of a 25 x 25 x 256 map, 4 : 1,
//                                      non-temporal; what 8 192 independent streams of 256-byte granules reach on this box
//                                      (scripts/probes/granule_probe.hip: the product layout; blocked / interleaved layouts + 1...6 %)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// Round 4 (scripts/probes/copy_probe.hip, 2 GiB, random and constant fills alike): a read needs EIGHT non-temporal 16-byte
// loads in flight per lane to reach this box's ceiling (6.25-6.37 TB/s; four plain loads: 5.5-5.8), and a copy block-contiguous
// 32 KiB spans with non-temporal loads AND stores (5.40-5.49 TB/s from 512 blocks up; the plain grid-stride copy this probe
// used before reads 4.5-4.8 at 2048-8192 blocks and 5.45 only at exactly four resident blocks per CU).
__global__ __launch_bounds__(256) void bw_read_kernel(const f4 *__restrict__ x, float *sink, long n4)
{
    f4 a = {0.f, 0.f, 0.f, 0.f};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; i < n4; i += stride) a += x[i];
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[0] = a[0];        // keeps the loads live, never true in practice
}

__global__ __launch_bounds__(256) void bw_copy_kernel(const f4 *__restrict__ x, f4 *__restrict__ y, long n4)
{
    constexpr int U = 8;
    const long span = 256L * U;                                        // 32 KiB per block and trip
    long base = (long)blockIdx.x * span;
    for (; base + span <= n4; base += (long)gridDim.x * span) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(x + base + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], y + base + u * 256 + threadIdx.x);
    }
    for (long i = base + threadIdx.x; i < n4 && i < base + span; i += 256) y[i] = x[i];
}

__global__ __launch_bounds__(256) void bw_mix_kernel(const f4 *__restrict__ x, f4 *__restrict__ y, long n4out, long span)
{
    for (long base = (long)blockIdx.x * span; base < n4out; base += (long)gridDim.x * span) {
        const long end = base + span < n4out ? base + span : n4out;
        for (long i = base + threadIdx.x; i < end; i += 256) {
            const f4 *q = x + ((i & ~63L) * 4 + (i & 63));
            const f4 v = q[0] + q[64] + q[128] + q[192];
            __builtin_nontemporal_store(v, y + i);
        }
    }
}

// mode 3: x = three maps back to back [3][S][841][256] floats, y = [S][625][256] floats
__global__ __launch_bounds__(256) void bw_gdw_pattern_kernel(const f4 *__restrict__ x, f4 *__restrict__ y, int samples)
{
    const int s = blockIdx.x >> 2, g = blockIdx.x & 3;
    const int tid = threadIdx.x;
    constexpr int NREAD = 29 * 87 * 16, NWRITE = 625 * 16;             // 16-byte lane accesses of a block
    const long map = (long)samples * 841 * 64;                        // f4 elements per map
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto raddr = [&](int idx) -> const f4 * {
        const int gran = idx >> 4, l = idx & 15;
        const int r = gran / 87, rem = gran - r * 87, j = rem / 29, px = rem - j * 29;
        return x + j * map + ((long)s * 841 + r * 29 + px) * 64 + g * 16 + l;
    };
    int wi = tid;
    for (int base = 0; base < NREAD; base += 8 * 256) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            v[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (idx < NREAD) v[u] = __builtin_nontemporal_load(raddr(idx));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
#pragma unroll
        for (int u = 0; u < 2; ++u, wi += 256)                          // 2 stores per 8 loads: 4 : 1
            if (wi < NWRITE) __builtin_nontemporal_store(acc, y + ((long)s * 625 + (wi >> 4)) * 64 + g * 16 + (wi & 15));
    }
    for (; wi < NWRITE; wi += 256) __builtin_nontemporal_store(acc, y + ((long)s * 625 + (wi >> 4)) * 64 + g * 16 + (wi & 15));
}

}  // namespace

/* One launch of a bandwidth probe.  mode 0: reads `bytes` from src (dst = a 4-byte sink); mode 1: copies `bytes`;
 * mode 2: reads `bytes` from src and writes bytes / 4 to dst; mode 3: GroupDW's traffic pattern for S = bytes / (3 * 841 * 1024)
 * samples (reads S * 3 * 841 KiB from src, writes S * 625 KiB to dst).  bytes % 4096 == 0 (mode 3: % 1024), 16-byte aligned pointers. */
extern "C" int usot_bw_probe(void *stream, const void *src, void *dst, int64_t bytes, int mode)
{
    if (!src || !dst || bytes <= 0 || (bytes % (mode == 3 ? 1024 : 4096)) || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16) || mode < 0 || mode > 3) return USOT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long n4 = bytes / 16;
    if (mode == 3) {
        const int64_t S = bytes / (3 * 841 * 1024);
        if (S < 1 || S > (1 << 20)) return USOT_EINVAL;
        hipLaunchKernelGGL(bw_gdw_pattern_kernel, dim3((unsigned)(S * 4)), dim3(256), 0, s, (const f4 *)src, (f4 *)dst, (int)S);
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    if (mode == 0) hipLaunchKernelGGL(bw_read_kernel, dim3(8192), dim3(256), 0, s, (const f4 *)src, (float *)dst, n4);
    else if (mode == 1) hipLaunchKernelGGL(bw_copy_kernel, dim3(2048), dim3(256), 0, s, (const f4 *)src, (f4 *)dst, n4);
    else hipLaunchKernelGGL(bw_mix_kernel, dim3(256), dim3(256), 0, s, (const f4 *)src, (f4 *)dst, n4 / 4, (long)4096);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
