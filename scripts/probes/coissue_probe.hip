// Does a VALU / SALU / LDS / VMEM instruction of ANOTHER wave on the same SIMD delay a wave that issues fp32 MFMAs back to back?
// Block = 8 waves: waves 0-3 (one per SIMD) run a chain-free loop of v_mfma_f32_16x16x4_f32 (4 accumulators); waves 4-7 (their
// SIMD siblings) run, per iteration, NV instructions of one kind.  Prints the MFMA waves' cycles per MFMA for each (kind, NV).
//   hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip && ./coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV>
__global__ __launch_bounds__(512) void probe(float *out, long long *cyc, int iters, const float *src)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    if (wave < 4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const float x = 1.f + lane, y = 2.f;
        const long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
        out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        int v = lane, u = lane * 3;
        float f = lane;
        f32x4 g = {0, 0, 0, 0};
        long long w = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(u));
                if (KIND == 1) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(w) : "v"(u) : "vcc");
                if (KIND == 3) { f32x4 t; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((lane & 15) * 16)); g += t; }
                if (KIND == 4) { asm volatile("ds_write_b128 %0, %1" :: "v"(lane * 16), "v"(g)); }
                if (KIND == 5) { f32x4 t = *(const volatile f32x4 *)(src + lane * 4 + (q & 7) * 256); g += t; }
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(f) : "vcc");
            }
            if (NV == 0) asm volatile("s_nop 0");
        }
        out[threadIdx.x] = v + f + g[0] + g[1] + (float)w;
    }
}

// The MFMA wave's OWN fragment reads: per round NRD ds_read_b128 for the NEXT round (burst in front, or one behind each of the
// first NRD MFMAs), then 8 MFMAs on the previous round's fragments; siblings idle or issuing NW ds_write_b128 per round.
template <int NRD, int ILV, int NW, int NMW = 4, int NDMA = 0>
__global__ __launch_bounds__(512) void probe_self(float *out, long long *cyc, int iters, const float *src = nullptr)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 1.f;
    __syncthreads();
    if (wave < NMW) {
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        f32x4 fr[2][4];
        for (int q = 0; q < 4; ++q) fr[0][q] = fr[1][q] = f32x4{1, 2, 3, 4};
        const unsigned base = (unsigned)(((lane & 15) * 68 + (lane >> 4) * 4) * 4);      // LDS byte address (the array sits at 0)
        const long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int cur = r, nxt = r ^ 1;
                // fragments of the NEXT round: inline asm (ds_read_b128 proper, no compiler-placed wait), waited for at the round's end
                if (!ILV) {
#pragma unroll
                    for (int q = 0; q < NRD; ++q)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[nxt][q]) : "v"(base), "n"(q * 16 * 68 * 4 + r * 64));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[cur][0][c], fr[cur][(NRD > 1) ? 1 : 0][c], acc[c & 1], 0, 0, 0);
                    if (ILV && c < NRD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[nxt][c]) : "v"(base), "n"(c * 16 * 68 * 4 + r * 64));
                    acc[2 + (c & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[cur][(NRD > 2) ? 2 : 0][c], fr[cur][(NRD > 3) ? 3 : 0][c], acc[2 + (c & 1)], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
        out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        f32x4 g = {1, 2, 3, 4};
        const float *gp = src + (blockIdx.x * 4 + (wave - 4)) * 4096 + lane * 4;
        for (int i = 0; i < iters * 2; ++i) {
#pragma unroll
            for (int q = 0; q < NW; ++q) asm volatile("ds_write_b128 %0, %1" :: "v"(16384 + lane * 16 + q * 1024), "v"(g));
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {          // LDS-DMA pieces (1 KiB: lane x 16 bytes behind M0) of an L2-resident buffer
                unsigned keep;
                const unsigned dst = __builtin_amdgcn_readfirstlane(20480 + (wave - 4) * 2048 + q * 1024);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gp + ((i * NDMA + q) & 3) * 256), "s"(dst) : "memory");
            }
            if (NDMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (NW == 0 && NDMA == 0) asm volatile("s_nop 0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        out[threadIdx.x] = g[0];
    }
}

template <int NRD, int ILV, int NW, int NMW = 4, int NDMA = 0> void run_self(float *out, long long *cyc, const float *src = nullptr)
{
    const int iters = 2000;
    if (NMW == 8) printf("TWO MFMA waves per SIMD (cycles are per wave: 512 = the pipe shared evenly and full) -> ");
    if (NDMA) printf("siblings also issue %d LDS-DMA pieces per round each -> ", NDMA);
    hipLaunchKernelGGL((probe_self<NRD, ILV, NW, NMW, NDMA>), dim3(256), dim3(512), 0, 0, out, cyc, iters, src);
    hipDeviceSynchronize();
    static long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("own reads: %d ds_read_b128 per round (%s), siblings %d ds_write_b128 per round: %.0f cycles per round of 8 MFMAs (256 = MFMA-bound)\n",
           NRD, ILV ? "one behind each MFMA" : "burst in front", NW, s / 1024 / iters / 2);
}

template <int KIND, int NV> void run(const char *name, float *out, long long *cyc, const float *src)
{
    const int iters = 4000;
    hipLaunchKernelGGL((probe<KIND, NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, src);
    hipDeviceSynchronize();
    static long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-18s %2d sibling instr per 4 MFMAs: %.1f cycles per MFMA\n", name, NV, s / 1024 / iters / 4);
}

int main()
{
    float *out, *src; long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8192); hipMalloc(&src, 32 << 20); hipMemset(src, 0, 32 << 20);
    run<0, 0>("idle sibling", out, cyc, src);
    run<0, 4>("v_add_u32", out, cyc, src);  run<0, 16>("v_add_u32", out, cyc, src);  run<0, 32>("v_add_u32", out, cyc, src);
    run<6, 16>("v_cndmask", out, cyc, src);
    run<2, 8>("v_mad_u64_u32", out, cyc, src);
    run<1, 16>("s_add_u32", out, cyc, src);
    run<3, 4>("ds_read_b128+wait", out, cyc, src); run<3, 8>("ds_read_b128+wait", out, cyc, src);
    run<4, 4>("ds_write_b128", out, cyc, src); run<4, 8>("ds_write_b128", out, cyc, src);
    run<5, 4>("global_load_x4", out, cyc, src); run<5, 8>("global_load_x4", out, cyc, src);
    run_self<1, 0, 0>(out, cyc); run_self<2, 0, 0>(out, cyc); run_self<3, 0, 0>(out, cyc); run_self<4, 0, 0>(out, cyc);
    run_self<3, 1, 0>(out, cyc); run_self<4, 1, 0>(out, cyc);
    run_self<3, 0, 0, 8>(out, cyc); run_self<2, 0, 0, 8>(out, cyc); run_self<4, 0, 0, 8>(out, cyc); run_self<1, 0, 0, 8>(out, cyc);
    run_self<3, 0, 3>(out, cyc); run_self<3, 1, 3>(out, cyc); run_self<3, 0, 6>(out, cyc); run_self<1, 0, 6>(out, cyc);
    run_self<3, 0, 0, 4, 1>(out, cyc, src); run_self<3, 0, 1, 4, 1>(out, cyc, src); run_self<3, 0, 0, 4, 2>(out, cyc, src); run_self<3, 0, 2, 4, 0>(out, cyc, src);
    return 0;
}
