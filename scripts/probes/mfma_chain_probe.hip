// How many independent accumulator chains does one wave per SIMD need to issue v_mfma_f32_16x16x4_f32 back to back?
// The batch-1 fp32 conv consumers (conv_igemm.hip v3/v4) run ONE wave per SIMD with 2 alternating accumulators
// and their MFMA phase measures ~52 cycles per MFMA instead of 32.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_chain_probe.hip -o scripts/probes/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int LDSREAD>
__global__ __launch_bounds__(256) void k(float *out, int iters, const float *in)
{
    __shared__ __attribute__((aligned(16))) float sm[64 * 68 * 2];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) sm[i] = in[i & 1023];
    __syncthreads();
    f32x4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = *(const f32x4 *)(sm + (lane & 15) * 68 + (lane >> 4) * 4);
    f32x4 b = *(const f32x4 *)(sm + 32 * 68 + (lane & 15) * 68 + (lane >> 4) * 4);
    long t0 = __builtin_readcyclecounter();
    const float *pa = sm + (lane & 15) * 68 + (lane >> 4) * 4, *pb = pa + 32 * 68;
    f32x4 a1 = a, b1 = b;
    for (int it = 0; it < iters; it += 2) {      // two fragment slots, no copies: the conv consumers' round structure
        if (LDSREAD) { a1 = *(const f32x4 *)(pa + ((it + 1) & 3) * 16); b1 = *(const f32x4 *)(pb + ((it + 1) & 3) * 16); }
        if (LDSREAD == 2) __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise sinks the reads below the MFMAs and waits at once
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], acc[c % NCH], 0, 0, 0);
        if (LDSREAD == 2) __builtin_amdgcn_sched_barrier(0);
        if (LDSREAD) { a = *(const f32x4 *)(pa + ((it + 2) & 3) * 16); b = *(const f32x4 *)(pb + ((it + 2) & 3) * 16); }
        if (LDSREAD == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b1[c], acc[c % NCH], 0, 0, 0);
    }
    long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
#pragma unroll
    for (int c = 1; c < NCH; ++c) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 256] = (float)(t1 - t0);
}

// The conv consumers' k-tile body: 4 rounds of (2 ds_read_b128 of a later round ; 4 MFMAs), straight-line, every
// read pinned in front of the MFMAs it overlaps with (sched_barrier), fragments DIST rounds ahead in 4 slots.
template <int DIST, int PIN>
__global__ __launch_bounds__(256) void ktile(float *out, int iters, const float *in)
{
    __shared__ __attribute__((aligned(16))) float sm[64 * 68 * 2];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) sm[i] = in[i & 1023];
    __syncthreads();
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float *pa = sm + (lane & 15) * 68 + (lane >> 4) * 4, *pb = pa + 32 * 68;
    f32x4 fa[4], fb[4];
#pragma unroll
    for (int r = 0; r < DIST; ++r) { fa[r] = *(const f32x4 *)(pa + r * 16); fb[r] = *(const f32x4 *)(pb + r * 16); }
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nr = (r + DIST) & 3;
            fa[nr] = *(const f32x4 *)(pa + nr * 16 + (it & 4)); fb[nr] = *(const f32x4 *)(pb + nr * 16 + (it & 4));
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][0], fb[r][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][1], fb[r][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][2], fb[r][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][3], fb[r][3], acc1, 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    long t1 = __builtin_readcyclecounter();
    f32x4 s = acc0 + acc1;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 256] = (float)(t1 - t0);
}

template <int DIST, int PIN> void run_ktile(float *out, const float *in, const char *name)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ktile<DIST, PIN>), dim3(256), dim3(256), 0, 0, out, 100, in);
    hipEventRecord(e0);
    hipLaunchKernelGGL((ktile<DIST, PIN>), dim3(256), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, out + 256 * 256, 4, hipMemcpyDeviceToHost);
    const double nm = 4.0 * iters;
    printf("%-44s %7.1f ns/MFMA  ticks/MFMA %.2f\n", name, ms * 1e6 / nm, cyc / nm);
}

// The same 32 x 32 x 64 k-tile on v_mfma_f32_32x32x2_f32 with the k range split over the four waves: each wave owns the
// whole 32 x 32 tile for 16 of the 64 k: 8 MFMAs (64 cycles each) and FOUR ds_read_b128 per k-tile instead of 16 + 8.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PIN>
__global__ __launch_bounds__(256) void ktile32(float *out, int iters, const float *in)
{
    __shared__ __attribute__((aligned(16))) float sm[64 * 68 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) sm[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float *pa = sm + (lane & 31) * 68 + wave * 16 + (lane >> 5) * 8, *pb = pa + 32 * 68;
    f32x4 fa[2][2], fb[2][2];
    fa[0][0] = *(const f32x4 *)pa; fa[0][1] = *(const f32x4 *)(pa + 4); fb[0][0] = *(const f32x4 *)pb; fb[0][1] = *(const f32x4 *)(pb + 4);
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                 // two k-tiles per trip: fragment buffers alternate without copies
            const int o = ((it + h + 1) & 1) * 64 * 68;
            fa[h ^ 1][0] = *(const f32x4 *)(pa + o); fa[h ^ 1][1] = *(const f32x4 *)(pa + o + 4);
            fb[h ^ 1][0] = *(const f32x4 *)(pb + o); fb[h ^ 1][1] = *(const f32x4 *)(pb + o + 4);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][j >> 2][j & 3], fb[h][j >> 2][j & 3], acc, 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 256] = (float)(t1 - t0);
}

template <int PIN> void run_ktile32(float *out, const float *in, const char *name)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ktile32<PIN>), dim3(256), dim3(256), 0, 0, out, 100, in);
    hipEventRecord(e0);
    hipLaunchKernelGGL((ktile32<PIN>), dim3(256), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, out + 256 * 256, 4, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f ns/k-tile  ticks/k-tile %.1f (16x16x4 floor 512)\n", name, ms * 1e6 / iters, cyc / iters);
}

template <int NCH, int LDSREAD> void run(float *out, const float *in, const char *name)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCH, LDSREAD>), dim3(256), dim3(256), 0, 0, out, 100, in);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, LDSREAD>), dim3(256), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, out + 256 * 256, 4, hipMemcpyDeviceToHost);
    const double nm = 4.0 * iters;
    printf("%-44s %7.1f ns/MFMA-equivalent  s_memtime ticks/MFMA %.2f  -> %.1f TFLOP/s chip\n", name, ms * 1e6 / nm, cyc / nm,
           256.0 * 4 * nm * 2048 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out, *in;
    hipMalloc(&out, (256 * 256 + 16) * 4); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    run<1, 0>(out, in, "1 chain");
    run<2, 0>(out, in, "2 chains");
    run<4, 0>(out, in, "4 chains");
    run<1, 1>(out, in, "1 chain  + 2 ds_read_b128 per 4 MFMA");
    run<2, 1>(out, in, "2 chains + 2 ds_read_b128 per 4 MFMA");
    run<4, 1>(out, in, "4 chains + 2 ds_read_b128 per 4 MFMA");
    run<1, 2>(out, in, "1 chain  + 2 ds_read_b128, reads pinned first");
    run<2, 2>(out, in, "2 chains + 2 ds_read_b128, reads pinned first");
    run_ktile<1, 0>(out, in, "k-tile body, 1 round ahead, hipcc order");
    run_ktile<1, 1>(out, in, "k-tile body, 1 round ahead, pinned");
    run_ktile<2, 0>(out, in, "k-tile body, 2 rounds ahead, hipcc order");
    run_ktile<2, 1>(out, in, "k-tile body, 2 rounds ahead, pinned");
    run_ktile32<0>(out, in, "32x32x2 k-split body, hipcc order");
    run_ktile32<1>(out, in, "32x32x2 k-split body, pinned");
    return 0;
}
