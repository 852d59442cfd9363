// HBM ceiling probes of the box at hand, callable from bench.py: what a kernel with GroupDW's byte mix (4 bytes read per
// byte written; csrc/xcorr.hip) can reach at all on this GPU, so that `xcorr_hbm.achieved` is quoted against a measured
// ceiling and not only against the 8 TB/s specification.  Three access patterns, 16 bytes per lane:
//   mode 0  read-only                  sum of src
//   mode 1  copy                       dst[i] = src[i]
//   mode 2  4 reads : 1 write          ONE interleaved read stream (output element i reads the four adjacent 1 KiB rows
//                                      4i .. 4i+3 of its 64-element group) + non-temporal stores: the best of the variants
//                                      measured in scripts/probes/bw_probe.hip (separate read streams: 4.8-5.2 TB/s; this: 5.5-6.1)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void bw_read_kernel(const f4 *__restrict__ x, float *sink, long n4)
{
    f4 a = {0.f, 0.f, 0.f, 0.f};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
        a += v0 + v1 + v2 + v3;
    }
    for (; i < n4; i += stride) a += x[i];
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[0] = a[0];        // keeps the loads live, never true in practice
}

__global__ __launch_bounds__(256) void bw_copy_kernel(const f4 *__restrict__ x, f4 *__restrict__ y, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = x[i];
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

}  // namespace

/* One launch of a bandwidth probe.  mode 0: reads `bytes` from src (dst = a 4-byte sink); mode 1: copies `bytes`;
 * mode 2: reads `bytes` from src and writes bytes / 4 to dst.  bytes % 4096 == 0, 16-byte aligned pointers. */
extern "C" int usot_bw_probe(void *stream, const void *src, void *dst, int64_t bytes, int mode)
{
    if (!src || !dst || bytes <= 0 || (bytes % 4096) || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16) || mode < 0 || mode > 2) return USOT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long n4 = bytes / 16;
    if (mode == 0) hipLaunchKernelGGL(bw_read_kernel, dim3(8192), dim3(256), 0, s, (const f4 *)src, (float *)dst, n4);
    else if (mode == 1) hipLaunchKernelGGL(bw_copy_kernel, dim3(4096), dim3(256), 0, s, (const f4 *)src, (f4 *)dst, n4);
    else hipLaunchKernelGGL(bw_mix_kernel, dim3(256), dim3(256), 0, s, (const f4 *)src, (f4 *)dst, n4 / 4, (long)4096);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
