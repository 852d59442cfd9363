// Copy-ceiling probe (round 4): why csrc/bw_probe.hip's copy mode reads 4.5 TB/s where MI355X_MICROARCH.md quotes 6.29 TB/s
// for a float4 copy.  Sweeps grid size, loads in flight per lane (unroll), non-temporal loads / stores, buffer size and the
// data fill (constant vs random: DVFS).  hipcc --offload-arch=gfx950 -O3 copy_probe.hip -o copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ x, f4 *__restrict__ y, long n4)
{
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) y[i] = x[i];
}

// block-contiguous spans: a block copies `span` consecutive f4 (span % 256 == 0) per trip, U pieces of 4 KiB in flight
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_span(const f4 *__restrict__ x, f4 *__restrict__ y, long n4)
{
    const long span = 256L * U;
    for (long base = (long)blockIdx.x * span; base + span <= n4; base += (long)gridDim.x * span) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(x + base + u * 256 + threadIdx.x) : x[base + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], y + base + u * 256 + threadIdx.x); else y[base + u * 256 + threadIdx.x] = v[u];
        }
    }
}

template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ x, float *sink, long n4)
{
    f4 a = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u];
    }
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[0] = a[0];
}

template <bool NTS>
__global__ __launch_bounds__(256) void k_write(f4 *__restrict__ y, long n4)
{
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if (NTS) __builtin_nontemporal_store(v, y + i); else y[i] = v;
    }
}

__global__ void k_fill(uint32_t *p, long n, uint32_t seed)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0x007fffffu) | 0x3f000000u | (h & 0x80000000u);      // random floats in +-[0.5, 1)
    }
}

template <typename F> float timeit(F f, int it = 10)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / it;
}

int main(int argc, char **argv)
{
    const long gib = argc > 1 ? atol(argv[1]) : 2;
    const long bytes = gib << 30;
    const long n4 = bytes / 16;
    f4 *x, *y; float *sink;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    for (int fill = 0; fill < 2; ++fill) {
        if (fill == 0) { hipMemset(x, 0x3c, bytes); hipMemset(y, 0, bytes); }
        else { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)x, bytes / 4, 17u); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)y, bytes / 4, 99u); }
        hipDeviceSynchronize();
        printf("== %ld GiB buffers, fill %s\n", gib, fill ? "random" : "constant 0x3c");
        for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
#define ROW(U, NTL, NTS) { float ms = timeit([&] { hipLaunchKernelGGL((k_copy<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, n4); }); \
            printf("copy  grid-stride U=%d ntl=%d nts=%d blocks %6d: %8.1f us %7.1f GB/s (r+w)\n", U, NTL, NTS, blocks, ms * 1e3, 2.0 * bytes / ms / 1e6); }
            ROW(1, false, false) ROW(1, false, true) ROW(1, true, true)
            ROW(4, false, false) ROW(4, false, true) ROW(4, true, true)
            ROW(8, false, true)
#undef ROW
        }
        for (int blocks : {256, 512, 1024, 2048, 4096}) {
#define ROW(U, NTL, NTS) { float ms = timeit([&] { hipLaunchKernelGGL((k_copy_span<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, n4); }); \
            printf("copy  block-span  U=%d ntl=%d nts=%d blocks %6d: %8.1f us %7.1f GB/s (r+w)\n", U, NTL, NTS, blocks, ms * 1e3, 2.0 * bytes / ms / 1e6); }
            ROW(4, false, false) ROW(4, false, true) ROW(8, false, true) ROW(8, true, true) ROW(16, false, true)
#undef ROW
        }
        for (int blocks : {2048, 8192, 32768}) {
            float ms = timeit([&] { hipLaunchKernelGGL((k_read<4, false>), dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
            float m2 = timeit([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(blocks), dim3(256), 0, 0, x, sink, n4); });
            float m3 = timeit([&] { hipLaunchKernelGGL((k_write<false>), dim3(blocks), dim3(256), 0, 0, y, n4); });
            float m4 = timeit([&] { hipLaunchKernelGGL((k_write<true>), dim3(blocks), dim3(256), 0, 0, y, n4); });
            printf("read U=4 %7.1f  read U=8 nt %7.1f  write %7.1f  write nt %7.1f GB/s   blocks %6d\n",
                   bytes / ms / 1e6, bytes / m2 / 1e6, bytes / m3 / 1e6, bytes / m4 / 1e6, blocks);
        }
        float ms = timeit([&] { hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0); });
        printf("hipMemcpy D2D: %8.1f us %7.1f GB/s (r+w)\n", ms * 1e3, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
