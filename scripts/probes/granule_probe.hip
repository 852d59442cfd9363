// Does GroupDW's HBM rate depend on its 256-byte granules?  Address-level emulation of groupdw_dma_kernel's traffic at
// 2048 samples: a block = (sample, 64-channel group) reads row r of three 29 x 29 x 256-channel fp32 maps (its quarter of
// every 1 KiB pixel row) and writes its quarter of a 25 x 25 x 256 output map, 4 : 1 —
//   layout 0  as the product: [sample][row][pixel][256 ch], a block touches 256-byte granules 1 KiB apart
//   layout 1  channel-group-blocked: [sample][group][row][pixel][64 ch], a block's bytes are contiguous
//   layout 2  as 0 with 128-channel groups (512-byte granules, 2 groups per sample)
//   layout 3  ONE stream per block: [sample][group][row][map][pixel][64 ch] - the three maps interleaved by row AND blocked by group
//   layout 4  the three maps interleaved by row only: [sample][row][map][pixel][256 ch] (256-byte granules, one region per sample)
// hipcc --offload-arch=gfx950 -O3 granule_probe.hip -o granule_probe && ./granule_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, bool NT>
__global__ __launch_bounds__(256) void k_gdw(const f4 *x0, const f4 *x1, const f4 *x2, f4 *y, int samples)
{
    constexpr int GPS = LAYOUT == 2 ? 2 : 4;                 // channel groups per sample
    constexpr int L = 64 / GPS;                              // 16-byte lanes per granule (16: 256 B, 32: 512 B)
    const int s = blockIdx.x / GPS, g = blockIdx.x % GPS;
    const f4 *xs[3] = {x0, x1, x2};
    const int tid = threadIdx.x;
    const int nread = 29 * 87 * L, nwrite = 625 * L;         // lane-loads / lane-stores of this block
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto raddr = [&](int idx) -> const f4 * {
        const int gran = idx / L, l = idx % L;
        const int r = gran / 87, rem = gran % 87, j = rem / 29, px = rem % 29;
        const long pos = (long)r * 29 + px;
        if (LAYOUT == 1) return xs[j] + (((long)s * 4 + g) * 841 + pos) * 16 + l;
        if (LAYOUT == 3) return xs[0] + (((long)s * 4 + g) * (3 * 841) + gran) * 16 + l;          // x0 holds all three maps
        if (LAYOUT == 4) return xs[0] + ((long)s * (3 * 841) + gran) * 64 + g * L + l;
        return xs[j] + ((long)s * 841 + pos) * 64 + g * L + l;
    };
    auto waddr = [&](int idx) -> f4 * {
        const int gran = idx / L, l = idx % L;
        if (LAYOUT == 1 || LAYOUT == 3) return y + (((long)s * 4 + g) * 625 + gran) * 16 + l;
        return y + ((long)s * 625 + gran) * 64 + g * L + l;
    };
    int wi = tid;
    for (int base = 0; base < nread; base += 8 * 256) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            v[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (idx < nread) v[u] = NT ? __builtin_nontemporal_load(raddr(idx)) : *raddr(idx);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
#pragma unroll
        for (int u = 0; u < 2; ++u, wi += 256)                // 2 stores per 8 loads
            if (wi < nwrite) {
                if (NT) __builtin_nontemporal_store(acc, waddr(wi)); else *waddr(wi) = acc;
            }
    }
    for (; wi < nwrite; wi += 256) *waddr(wi) = acc;
}

int main()
{
    const int S = 2048;
    const size_t xb = (size_t)S * 841 * 1024, yb = (size_t)S * 625 * 1024;
    f4 *x[3], *y;
    for (int j = 0; j < 3; ++j) { hipMalloc(&x[j], j ? xb : 3 * xb); hipMemset(x[j], 1, j ? xb : 3 * xb); }
    hipMalloc(&y, yb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 3.0 * xb + yb;
    auto run = [&](const char *name, auto kern, int blocks) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x[0], x[1], x[2], y, S);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
        }
    };
    run("layout 0: 256 B granules, 1 KiB apart", k_gdw<0, false>, S * 4);
    run("layout 0, non-temporal", k_gdw<0, true>, S * 4);
    run("layout 1: channel-group-blocked (contiguous)", k_gdw<1, false>, S * 4);
    run("layout 1, non-temporal", k_gdw<1, true>, S * 4);
    run("layout 2: 512 B granules (128-ch groups)", k_gdw<2, false>, S * 2);
    run("layout 2, non-temporal", k_gdw<2, true>, S * 2);
    run("layout 3: one contiguous stream per block", k_gdw<3, false>, S * 4);
    run("layout 3, non-temporal", k_gdw<3, true>, S * 4);
    run("layout 4: maps interleaved by row, 256 B granules", k_gdw<4, false>, S * 4);
    run("layout 4, non-temporal", k_gdw<4, true>, S * 4);
    return 0;
}
