// Does the PIECE SIZE of a strided stream set the HBM rate?  (round 4)  The pixel-stationary 1x1 kernels of the batch-64 backbone
// (csrc/pw_panel.hip: conv3 + residual, 283 MB per launch) read the residual and write Y in 128-byte pieces per pixel and
// channel group, 2 KB apart (NHWC, 1024 channels), and move their bytes at 4.3 TB/s where a block-contiguous copy on the same
// box reaches 5.5.  This probe copies a [pixels][ROW bytes] array the way such a kernel walks it: a workgroup owns a panel of 256
// pixels and visits it ROW / PIECE times, each visit touching PIECE contiguous bytes of every pixel row (PIECE = ROW: one
// contiguous 512 KB block).   hipcc --offload-arch=gfx950 -O3 stride_probe.hip -o stride_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PIECE, bool NT, bool WR>       // PIECE bytes per pixel and visit; ROW = 2048
__global__ __launch_bounds__(512) void k_panel(const char *__restrict__ x, char *__restrict__ y, long npanels, float *sink)
{
    constexpr int ROW = 2048, PIX = 256;
    constexpr int LPP = PIECE / 16;                  // lanes per pixel
    constexpr int PPP = 512 / LPP;                   // pixels per pass
    f4 acc = {0, 0, 0, 0};
    for (long pn = blockIdx.x; pn < npanels; pn += gridDim.x) {
        const char *xb = x + pn * (long)(PIX * ROW);
        char *yb = y + pn * (long)(PIX * ROW);
        for (int g = 0; g < ROW / PIECE; ++g) {
            constexpr int NPS = PIX / PPP, B = NPS < 8 ? NPS : 8;       // at most 8 loads in flight per lane
            for (int p0 = 0; p0 < NPS; p0 += B) {
                f4 v[B];
#pragma unroll
                for (int ps = 0; ps < B; ++ps) {
                    const int pix = (p0 + ps) * PPP + threadIdx.x / LPP, sub = threadIdx.x % LPP;
                    const f4 *src = (const f4 *)(xb + (long)pix * ROW + g * PIECE + sub * 16);
                    v[ps] = NT ? __builtin_nontemporal_load(src) : *src;
                }
#pragma unroll
                for (int ps = 0; ps < B; ++ps) {
                    const int pix = (p0 + ps) * PPP + threadIdx.x / LPP, sub = threadIdx.x % LPP;
                    f4 *dst = (f4 *)(yb + (long)pix * ROW + g * PIECE + sub * 16);
                    if (WR) { if (NT) __builtin_nontemporal_store(v[ps], dst); else *dst = v[ps]; }
                    else acc += v[ps];
                }
            }
        }
    }
    if (!WR && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

__global__ void k_fill(uint32_t *p, long n, uint32_t seed)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0x007fffffu) | 0x3f000000u | (h & 0x80000000u);
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
    return ms / it;
}

int main()
{
    const long bytes = 1L << 30;                     // 2048 panels of 512 KB: far beyond the 256 MB Infinity Cache
    const long npanels = bytes / (256 * 2048);
    char *x, *y; float *sink;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)x, bytes / 4, 17u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)y, bytes / 4, 99u);
    hipDeviceSynchronize();
    for (int grid : {256, 512}) {
#define ROW(P, NT) { float ms = timeit([&] { hipLaunchKernelGGL((k_panel<P, NT, true>), dim3(grid), dim3(512), 0, 0, x, y, npanels, sink); }); \
        float mr = timeit([&] { hipLaunchKernelGGL((k_panel<P, NT, false>), dim3(grid), dim3(512), 0, 0, x, y, npanels, sink); }); \
        printf("piece %4d B nt=%d grid %3d: copy %7.1f GB/s (r+w)   read-only %7.1f GB/s\n", P, NT, grid, 2.0 * bytes / ms / 1e6, 1.0 * bytes / mr / 1e6); }
        ROW(64, false) ROW(128, false) ROW(128, true) ROW(256, false) ROW(256, true) ROW(512, false) ROW(1024, false) ROW(2048, false) ROW(2048, true)
#undef ROW
    }
    return 0;
}
