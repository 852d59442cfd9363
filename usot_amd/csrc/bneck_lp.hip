// Layer1's FIRST bottleneck of the batched low-precision backbone as ONE launch (round 4):
//     t1 = relu(bn1(conv1 1x1 64 -> 64 (x)))            modules.py:40-42
//     t2 = relu(bn2(conv2 3x3 64 -> 64 (t1)))           modules.py:44-46
//     y  = relu(bn3(conv3 1x1 64 -> 256 (t2)) + bn_d(downsample 1x1 64 -> 256 (x)))     modules.py:48-56, 108-113
//     t  = relu(bn1'(conv1' 1x1 256 -> 64 (y)))         the NEXT block's conv1 (modules.py:40-42), so that the rest of
//                                                       layer1 keeps its launches (halo 3x3, conv3 + next conv1 pairs)
// Before: four launches (shortcut conv, conv1, conv2, conv3 + next conv1) moving 32 + 130 | 32 + 32 | 32 + 32 | 32 + 130 +
// 130 + 32 MB at batch 64 — 39 + 22 + 28 + 77 us, all HBM time.  Here the 64-channel intermediates t1 and t2 never leave
// LDS, the shortcut conv is the second half of conv3's k axis ([w3 | wd] x [t2 ; x]: ONE accumulation, no rounded
// shortcut map), and the block's output tile feeds the next conv1 from LDS: 32 MB in (x 1.4 halo, mostly L2 hits), 130 +
// 32 MB out.
//
// A workgroup (8 waves, persistent, one per CU) owns an 8 x 16 tile of output pixels of one image:
//   X    the 10 x 18 x 64 halo tile of x -> LDS by LDS-DMA ([pixel][8 chunks of 16 B], chunk index XOR-swizzled by
//        (halo column >> 1) & 7 on the SOURCE side), double-buffered: the next tile's halo is issued at the top of this tile;
//   A    conv1 on all 180 halo pixels (x 1.4: recomputed halo), filter fragments from LDS; t1 -> LDS ([pixel][8 + 2 pad
//        chunks], the layout of csrc/conv3x3_halo.hip), ZERO at halo pixels outside the image (conv2 pads t1, not x);
//   B    conv2 as conv3x3_halo_kernel (B fragment of tap (kh, kw) = the t1 image at a shifted pixel offset, A fragments in
//        registers for the whole kernel), but wave (cb, rh) = 16 channels x 4 tile rows: 18 A fragments = 72 registers
//        (32 channels x 2 rows = 144 left the other phases no room: 340 bytes of spills); a B fragment then feeds ONE MFMA,
//        the LDS reads of the phase take as long as its MFMAs (2 304 cycles per tile each, side by side);
//   C    conv3 | downsample: wave w = output channels 32 w .. 32 w + 31 x all 128 pixels, k = 128: B fragments of k-steps
//        0, 1 from t2 (LDS), of k-steps 2, 3 from the CENTRE pixels of the x halo tile; epilogue from registers (filter rows
//        permuted so that a lane's 8 accumulators per pixel are 8 contiguous channels: 16-byte stores);
//   D    the next conv1: the rounded y tile goes to LDS in two halves of 64 pixels ([pixel][32 + 1 chunks]), wave (cb, ph) =
//        16 output channels x 2 pixel blocks, its 8 A fragments in registers for the whole kernel.
// t1, t2 and the y halves share ONE LDS region (barriers between the phases).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct BneckK {
    const uint16_t *x, *w1, *w2, *w3c, *wn;
    const float *b1, *b2, *b3c, *bn;
    uint16_t *y, *t;
    const uint16_t *zero;
    int N, H, W, tiles_x, tiles_y, ntiles;
};

__device__ __attribute__((aligned(16))) uint32_t bk_zero16[4] = {0u, 0u, 0u, 0u};

constexpr int TH = 8, TW = 16;              // output tile
constexpr int HR = TH + 2, HC = TW + 2;     // halo tile 10 x 18
constexpr int HPIX = HR * HC;               // 180
constexpr int TPS = 10;                     // chunks per pixel of t1 / t2 in LDS (8 data + 2 pad)
constexpr int XSLOT = HPIX * 8;             // 1440 chunks per x halo slot
constexpr int XPIECES = XSLOT / 60;         // 24 DMA instructions of 60 lanes
// LDS map, in 16-byte chunks
constexpr int L_W1F = 0;                    // [h 2][i 2][ks 2][64]
constexpr int L_W3F = L_W1F + 512;          // [w 8][i 2][ks 4][64]
constexpr int L_XR = L_W3F + 4096;          // [2][XSLOT]
constexpr int L_TR = L_XR + 2 * XSLOT;      // t1 (1800) | t2 (1280) | y half (64 pixels x 33 chunks = 2112)
constexpr int YPS = 33;                     // chunks per pixel of a y half: 32 data + 1 (odd pitch: bank slot = (pixel + chunk) mod 16)
constexpr int L_BIAS = L_TR + 64 * YPS;         // b1[64] b2[64] b3c[256] bn[64] floats = 112 chunks
constexpr int L_END = L_BIAS + 112;
constexpr int BK_LDS = L_END * 16;          // 154 368 B
constexpr int W2CH = 64 * 576 / 8;          // 4608 chunks: staged once over [L_XR, L_XR + 4608) (x slots + t region)
static_assert(W2CH <= 2 * XSLOT + 64 * YPS, "w2 staging fits over the x slots and the t region");
static_assert(BK_LDS <= 160 * 1024, "LDS");

template <bool F16> __device__ __forceinline__ f32x4 bk_mfma(u32x4 a, u32x4 b, f32x4 c)
{
#ifdef USOT_BKABL_NOMMA          // scripts/bneck_probe.py: timing builds with parts of the kernel removed
    c[0] += __builtin_bit_cast(float, a[0] ^ b[0]);
    return c;
#endif
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// x slot: physical chunk = logical chunk ^ swz(halo column).  By COLUMN (not by linear pixel) so that phase C's centre reads are
// one base address + a row immediate; 16 consecutive columns give 16 distinct (parity, swz) pairs = conflict-free ds_read_b128;
// phase A's linear runs of 16 pixels can pair column 16 / 17 with column 0 / 1 of the next row (2-way on a few reads)
__device__ __forceinline__ int bk_swz(int hx) { return (hx >> 1) & 7; }

template <bool F16> __device__ __forceinline__ float bk_unpack(uint32_t h)
{
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
    else               return __builtin_bit_cast(float, h << 16);
}

// 8 fp32 -> 8 storage-type values after bias + ReLU
template <bool F16> __device__ __forceinline__ u32x4 bk_pack8(f32x4 a, f32x4 b, f32x4 ba, f32x4 bb)
{
    a += ba; b += bb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = fmaxf(a[e], 0.f); b[e] = fmaxf(b[e], 0.f); }
    u32x4 o;
    o[0] = usot_pack2_lp<F16>(a[0], a[1]); o[1] = usot_pack2_lp<F16>(a[2], a[3]);
    o[2] = usot_pack2_lp<F16>(b[0], b[1]); o[3] = usot_pack2_lp<F16>(b[2], b[3]);
    return o;
}

template <bool F16>
__global__ __launch_bounds__(512) void bneck_first_kernel(const BneckK p)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int h = wave & 1, rq = wave >> 1;         // phases A, B: channel half; B: row pair; A: pixel quarter (= rq)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sm;

    auto dma16 = [&](const void *src, uint32_t lds_byte) {        // active lanes x 16 B -> LDS lds_byte + lane * 16
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
    };
    const int grid8 = gridDim.x >> 3;
    // tile of (iteration k, block b): the grid's eight XCD groups each walk a contiguous run of tiles, so that the tiles of one
    // image (whose halos overlap) are in flight on ONE XCD at the same time (block b runs on XCD b % 8: speed only)
    auto tile_of = [&](int k) { return (k * 8 + ((int)blockIdx.x & 7)) * grid8 + ((int)blockIdx.x >> 3); };
    auto tile_xy = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / (p.tiles_x * p.tiles_y);
        const int r2 = tile - n * p.tiles_x * p.tiles_y;
        ty = r2 / p.tiles_x;
        tx = r2 - ty * p.tiles_x;
    };
    // x halo tile of `tile` -> slot: linear position L = piece * 60 + lane (lanes 0..59), pixel P = L >> 3, physical chunk pc =
    // L & 7 holds logical chunk pc ^ swz(column).  The lane's halo coordinates and chunk are the same for every tile: packed
    // once (hy | hx << 8 | chunk byte offset << 16) — the per-tile address is then a dozen integer operations per piece, not
    // the division-laden sixty hipcc makes of the straight code (the kernel is instruction-issue-bound: ~1 500 instructions per
    // wave and tile on two waves per SIMD, scripts/bneck_probe.py: 6.5 us per tile with MFMAs, DMA and stores compiled out)
    uint32_t xpk[XPIECES / 8];
#pragma unroll
    for (int i = 0; i < XPIECES / 8; ++i) {
        const int L = (i * 8 + wave) * 60 + lane;
        const int P = L >> 3, pc = L & 7;
        const int hy = P / HC, hx = P - hy * HC;
        xpk[i] = (uint32_t)hy | ((uint32_t)hx << 8) | ((uint32_t)((pc ^ bk_swz(hx)) << 4) << 16);
    }
    auto issue_x = [&](int tile, int slot) {
        int n, ty, tx;
        tile_xy(tile, n, ty, tx);
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const char *xn = (const char *)(p.x + (long)n * p.H * p.W * 64);
#pragma unroll
        for (int i = 0; i < XPIECES / 8; ++i) {
            const int iy = y0 + (int)(xpk[i] & 0xffu), ix = x0 + (int)((xpk[i] >> 8) & 0xffu);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t off = (uint32_t)(iy * p.W + ix) * 128u + (xpk[i] >> 16);     // < 2^31: checked by the launcher
            const void *src = ok ? (const void *)(xn + off) : (const void *)p.zero;
            if (lane < 60) dma16(src, lds0 + (uint32_t)((L_XR + slot * XSLOT + (i * 8 + wave) * 60) * 16));
        }
    };

    int kit = 0;
    int tile = tile_of(0);
    if (tile >= p.ntiles) return;                    // (the launcher sizes the grid so that every block has a first tile)
    // ---- one-time staging: w2 fragments (through LDS, as conv3x3_halo_kernel), w1 / w3c fragments (to stay in LDS), biases
#pragma unroll
    for (int i = 0; i < W2CH / 512; ++i) {           // w2: fragment (cb, s) = chunk (cb*18 + s)*64 + ln  <-  w2[cb*16 + (ln&15)][s*32 + (ln>>4)*8]
        const int c = i * 512 + tid;
        const int ln = c & 63, fs = c >> 6, s = fs % 18, cbb = fs / 18;
        dma16(p.w2 + (long)(cbb * 16 + (ln & 15)) * 576 + s * 32 + (ln >> 4) * 8, lds0 + (uint32_t)((L_XR + i * 512 + wave * 64) * 16));
    }
    {   // w1: fragment (hh, i, ks) = chunk ((hh*2 + i)*2 + ks)*64 + ln  <-  w1[hh*32 + ((ln&15)>>2)*8 + i*4 + (ln&3)][ks*32 + (ln>>4)*8]
        const int c = tid;                           // 512 chunks
        const int ln = c & 63, f = c >> 6, ks = f & 1, i = (f >> 1) & 1, hh = f >> 2;
        const int ch = hh * 32 + ((ln & 15) >> 2) * 8 + i * 4 + (ln & 3);
        dma16(p.w1 + ch * 64 + ks * 32 + (ln >> 4) * 8, lds0 + (uint32_t)((L_W1F + wave * 64) * 16));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {                 // w3c: fragment (w, i, ks) = chunk ((w*2 + i)*4 + ks)*64 + ln
        const int c = it * 512 + tid;
        const int ln = c & 63, f = c >> 6, ks = f & 3, i = (f >> 2) & 1, w = f >> 3;
        const int ch = w * 32 + ((ln & 15) >> 2) * 8 + i * 4 + (ln & 3);
        dma16(p.w3c + ch * 128 + ks * 32 + (ln >> 4) * 8, lds0 + (uint32_t)((L_W3F + it * 512 + wave * 64) * 16));
    }
    {
        float *sb = (float *)(sm + L_BIAS);
        if (tid < 64) { sb[tid] = p.b1[tid]; sb[64 + tid] = p.b2[tid]; sb[384 + tid] = p.bn[tid]; }
        if (tid < 256) sb[128 + tid] = p.b3c[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int cb = wave & 3, rh = wave >> 2;        // phase B: channel block, row half; phase D: channel block, pixel-block pair
    u32x4 af2[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) af2[s] = sm[L_XR + (cb * 18 + s) * 64 + lane];
    u32x4 an[8];                                     // phase D: wn[cb*16 + l15][ks*32 + q*8 .. +7]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) an[ks] = *(const u32x4 *)(p.wn + (cb * 16 + l15) * 256 + ks * 32 + q * 8);
    __syncthreads();                                 // every wave holds its fragments: the staging space is free
    issue_x(tile, 0);
    const float *sb = (const float *)(sm + L_BIAS);
    int slot = 0;
    int stores_prev = 0;                             // first tile: wait for everything

    for (;; ++kit) {
        int n, ty, tx;
        tile_xy(tile, n, ty, tx);
        const int next = tile_of(kit + 1);
        // this tile's halo was issued one tile ago (or in the prologue): wait until only the previous tile's last stores are
        // outstanding — vmcnt retires in order, so the DMA pieces (older) have landed for this wave
        // (stores_prev = this wave's y and t stores of the previous tile, the only younger operations — the next conv1's fragments
        //  are loaded once, before the loop: wave-uniform counts of rows inside the image.  Waiting for the stores too — their
        //  acknowledgements, not their issue — would put a memory round trip into every tile)
        // The counted wait is correct only while this wave issues EXACTLY stores_prev vector-memory operations between the DMA and
        // here (one store per in-image row; a compiler-inserted extra op would be safe, a merged or sunk store a silent LDS race).
        // -DUSOT_BNECK_VMCNT0 builds wait for everything instead; tests/test_gpu_ops.py::test_bneck_counted_waits_equal_full_waits
        // compares the two builds bit for bit on ragged images.
#ifdef USOT_BNECK_VMCNT0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)stores_prev;
#else
        switch (stores_prev) {
#define BK_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        BK_W(12) BK_W(11) BK_W(10) BK_W(9) BK_W(8) BK_W(7) BK_W(6) BK_W(5) BK_W(4) BK_W(3) BK_W(2) BK_W(1)
#undef BK_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#endif
        __syncthreads();                             // #1: x(tile) visible; the other slot and the t region are free
#ifndef USOT_BKABL_NODMA
        if (next < p.ntiles) issue_x(next, slot ^ 1);
#endif
        const u32x4 *xs = sm + L_XR + slot * XSLOT;
        u32x4 *tr = sm + L_TR;

        // ---------------- A: conv1 on the halo pixels.  wave (h, pq = rq): channels h*32 .. +31 x pixel blocks pq*3 .. +2
        {
            f32x4 acc[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 a[2], b[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = sm[L_W1F + ((h * 2 + i) * 2 + ks) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int P = (rq * 3 + j) * 16 + l15;          // < 192: pixels 180.. read whatever follows the slot
                    b[j] = xs[P * 8 + ((ks * 4 + q) ^ bk_swz(P % HC))];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = bk_mfma<F16>(a[i], b[j], acc[i][j]);
            }
            const f32x4 ba = *(const f32x4 *)(sb + h * 32 + q * 8), bb = *(const f32x4 *)(sb + h * 32 + q * 8 + 4);
            const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int P = (rq * 3 + j) * 16 + l15;
                const int hy = P / HC, hx = P - hy * HC;
                const bool in = (unsigned)(y0 + hy) < (unsigned)p.H && (unsigned)(x0 + hx) < (unsigned)p.W;
                u32x4 o = bk_pack8<F16>(acc[0][j], acc[1][j], ba, bb);
                if (!in) o = u32x4{0u, 0u, 0u, 0u};
                if (P < HPIX) tr[P * TPS + h * 4 + q] = o;
            }
        }
        __syncthreads();                             // #2: t1 complete

        // ---------------- B: conv2 from the t1 halo image.  wave (cb, rh): channels cb*16 .. +15 x tile rows rh*4 .. rh*4 + 3
        f32x4 acc2[4];
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const u32x4 *hb = tr + ((rh * 4) * HC + l15) * TPS + q;
            u32x4 bf[2][4];
            auto read_b = [&](int s, u32x4 (&b)[4]) {
                const int t = s >> 1, ks = s & 1, kh = t / 3, kw = t - kh * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = hb[((j + kh) * HC + kw) * TPS + ks * 4];
            };
            read_b(0, bf[0]);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18) read_b(s + 1, bf[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[j] = bk_mfma<F16>(af2[s], bf[s & 1][j], acc2[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                             // #3: every wave has read t1; t2 takes its place
        {   // lane (l15, q): channels cb*16 + q*4 .. +3 of pixel (rh*4 + j, l15): 8 bytes of the pixel's 128
            const f32x4 bv = *(const f32x4 *)(sb + 64 + cb * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc2[j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]); o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                *(u32x2 *)((char *)(tr + ((rh * 4 + j) * 16 + l15) * TPS) + cb * 32 + q * 8) = o;
            }
        }
        __syncthreads();                             // #4: t2 complete

        // ---------------- C: conv3 | downsample.  wave w: channels w*32 .. +31 x the tile's 8 pixel blocks (= tile rows)
        u32x4 yo[8];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {             // four pixel blocks at a time (all eight: 64 accumulators + 32 fragment registers)
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc[i][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = sm[L_W3F + ((wave * 2 + i) * 4 + ks) * 64 + lane];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    if (ks < 2) {
                        b[pb] = tr[((hp * 4 + pb) * 16 + l15) * TPS + ks * 4 + q];
                    } else {
                        const int P = (hp * 4 + pb + 1) * HC + 1 + l15;      // centre pixel (row, l15) of the x halo tile
                        b[pb] = xs[P * 8 + (((ks - 2) * 4 + q) ^ bk_swz(1 + l15))];
                    }
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][pb] = bk_mfma<F16>(a[i], b[pb], acc[i][pb]);
            }
            const f32x4 ba = *(const f32x4 *)(sb + 128 + wave * 32 + q * 8), bb = *(const f32x4 *)(sb + 128 + wave * 32 + q * 8 + 4);
            const int ox = tx * TW + l15;
            char *yb = (char *)(p.y + (((long)n * p.H + ty * TH) * p.W + ox) * 256 + wave * 32 + q * 8);   // row ty*8 of the tile
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int oy = ty * TH + hp * 4 + pb;
                yo[hp * 4 + pb] = bk_pack8<F16>(acc[0][pb], acc[1][pb], ba, bb);
#ifdef USOT_BKABL_NOSTORE
                if (oy < p.H && ox < p.W && yo[hp * 4 + pb][0] == 0x12345678u)
#else
                if (oy < p.H && ox < p.W)
#endif
                    *(u32x4 *)(yb + (uint32_t)((hp * 4 + pb) * p.W) * 512u) = yo[hp * 4 + pb];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------- D: the next conv1 on the rounded y tile, 64 pixels at a time.  wave (cb, ph): 16 channels x 2 blocks
        const int ph = rh;
        // (the next conv1's 8 A fragments live in registers for the whole kernel, loaded before the tile loop: fetched per tile
        //  they sat behind the y stores in the in-order vmcnt queue, and waiting for them waited for the stores' acknowledgements)
        const f32x4 bnv = *(const f32x4 *)(sb + 384 + cb * 16 + q * 4);
#ifndef USOT_BKABL_NOD
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                         // #5 / #7: t2 (or the first y half) has been read by every wave
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) tr[(pb * 16 + l15) * YPS + wave * 4 + q] = yo[half * 4 + pb];
            __syncthreads();                         // #6 / #8
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                u32x4 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = tr[((ph * 2 + j) * 16 + l15) * YPS + ks * 4 + q];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = bk_mfma<F16>(an[ks], b[j], acc[j]);
            }
            const int ox = tx * TW + l15;
            char *tb = (char *)(p.t + (((long)n * p.H + ty * TH) * p.W + ox) * 64 + cb * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = ty * TH + half * 4 + ph * 2 + j;
                f32x4 v = acc[j] + bnv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]); o[1] = usot_pack2_lp<F16>(v[2], v[3]);
#ifdef USOT_BKABL_NOSTORE
                if (oy < p.H && ox < p.W && o[0] == 0x12345678u)
#else
                if (oy < p.H && ox < p.W)
#endif
                    *(u32x2 *)(tb + (uint32_t)((half * 4 + ph * 2 + j) * p.W) * 128u) = o;
            }
        }
#endif
        if (next >= p.ntiles) break;
        {   // stores this wave issued since the DMA: y rows ty*8 + 0..7 and t rows ty*8 + half*4 + ph*2 + j inside the image
            // (column 0 of a tile always is, so a row inside the image is a store instruction)
            const int r0 = p.H - (ty * TH + ph * 2);
            stores_prev = min(8, max(0, p.H - ty * TH)) + min(2, max(0, r0)) + min(2, max(0, r0 - 4));
        }
        tile = next;
        slot ^= 1;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The REST of a layer1 bottleneck whose conv1 the previous launch already made (bneck_first_kernel's phase D, or this
// kernel's): conv2 3x3 + BN + ReLU, conv3 1x1 + BN + identity residual + ReLU and the NEXT block's conv1 + BN + ReLU
// (modules.py:43-58 and :40-42 of the following block; CN = 64 inside layer1, 128 for layer2's first block) — phases B, C, D
// of the kernel above with the t1 halo tile coming from global memory (zero outside the image = conv2's padding) and the
// residual read from global memory into registers at the top of the tile.  Before: conv3x3_halo (32 + 32 MB) + the
// pixel-stationary pair (32 + 130 + 130 + 32 MB); here t2 never leaves LDS: 32 x 1.4 + 130 + 130 + 32 MB, one launch.
//   LDS: t1 halo [2 slots][180 pixels][8 + 2 pad chunks] | t2 [128][10] | y half [64][33] | w3 fragments [8][2][2][64] | biases
struct BneckTK {
    const uint16_t *t1, *res, *w2, *w3, *wn;
    const float *b2, *b3, *bn;
    uint16_t *y, *t;
    const uint16_t *zero;
    int N, H, W, tiles_x, tiles_y, ntiles;
};

constexpr int T1SLOT = HPIX * TPS;          // 1800 chunks
constexpr int T1PIECES = (T1SLOT + 63) / 64;        // 29 DMA instructions (the last one 8 lanes)
constexpr int M_T1 = 0;                     // [2][T1SLOT]
constexpr int M_T2 = M_T1 + 2 * T1SLOT;     // [128][TPS]
constexpr int M_YH = M_T2 + 128 * TPS;      // [64][YPS]
constexpr int M_W3F = M_YH + 64 * YPS;      // [w 8][i 2][ks 2][64]
constexpr int M_BIAS = M_W3F + 2048;        // b2[64] b3[256] bn[128] floats = 112 chunks
constexpr int M_END = M_BIAS + 112;
constexpr int BT_LDS = M_END * 16;          // 146 432 B
static_assert(W2CH <= M_W3F, "w2 staging fits in front of the w3 fragments");
static_assert(BT_LDS <= 160 * 1024, "LDS");

template <bool F16, int CN>
__global__ __launch_bounds__(512) void bneck_tail_kernel(const BneckTK p)
{
    static_assert(CN == 64 || CN == 128, "next conv1: 256 -> 64 | 128");
    constexpr int NCB = CN / 16;                    // channel blocks of the next conv1
    constexpr int WPC = 8 / NCB;                    // waves per channel block (2 | 1)
    constexpr int NPX = 4 / WPC;                    // pixel blocks per wave and y half (2 | 4)
    extern __shared__ __attribute__((aligned(16))) u32x4 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sm;
    auto dma16 = [&](const void *src, uint32_t lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
    };
    const int grid8 = gridDim.x >> 3;
    auto tile_of = [&](int k) { return (k * 8 + ((int)blockIdx.x & 7)) * grid8 + ((int)blockIdx.x >> 3); };
    auto tile_xy = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / (p.tiles_x * p.tiles_y);
        const int r2 = tile - n * p.tiles_x * p.tiles_y;
        ty = r2 / p.tiles_x;
        tx = r2 - ty * p.tiles_x;
    };
    // t1 halo tile -> slot: position L = piece * 64 + lane = pixel P = L / 10, chunk pc = L % 10 (8, 9: pad, not written)
    uint32_t tpk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (i * 8 + wave) * 64 + lane;
        const int P = L / TPS, pc = L - P * TPS;
        const int hy = P / HC, hx = P - hy * HC;
        tpk[i] = (L < T1SLOT && pc < 8) ? ((uint32_t)hy | ((uint32_t)hx << 8) | ((uint32_t)(pc << 4) << 16)) : 0xffffffffu;
    }
    auto issue_t1 = [&](int tile, int slot) {
        int n, ty, tx;
        tile_xy(tile, n, ty, tx);
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const char *tn = (const char *)(p.t1 + (long)n * p.H * p.W * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = y0 + (int)(tpk[i] & 0xffu), ix = x0 + (int)((tpk[i] >> 8) & 0xffu);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t off = (uint32_t)(iy * p.W + ix) * 128u + ((tpk[i] >> 16) & 0xffu);
            const void *src = ok ? (const void *)(tn + off) : (const void *)p.zero;
            if (tpk[i] != 0xffffffffu) dma16(src, lds0 + (uint32_t)((M_T1 + slot * T1SLOT + (i * 8 + wave) * 64) * 16));
        }
    };

    int kit = 0;
    int tile = tile_of(0);
    if (tile >= p.ntiles) return;
    // ---- one-time staging: w2 fragments through LDS (as above), w3 fragments (stay in LDS), biases
#pragma unroll
    for (int i = 0; i < W2CH / 512; ++i) {
        const int c = i * 512 + tid;
        const int ln = c & 63, fs = c >> 6, s = fs % 18, cbb = fs / 18;
        dma16(p.w2 + (long)(cbb * 16 + (ln & 15)) * 576 + s * 32 + (ln >> 4) * 8, lds0 + (uint32_t)((i * 512 + wave * 64) * 16));
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {                 // w3: fragment (w, i, ks) = chunk ((w*2 + i)*2 + ks)*64 + ln
        const int c = it * 512 + tid;
        const int ln = c & 63, f = c >> 6, ks = f & 1, i = (f >> 1) & 1, w = f >> 2;
        const int ch = w * 32 + ((ln & 15) >> 2) * 8 + i * 4 + (ln & 3);
        dma16(p.w3 + ch * 64 + ks * 32 + (ln >> 4) * 8, lds0 + (uint32_t)((M_W3F + it * 512 + wave * 64) * 16));
    }
    {
        float *sbw = (float *)(sm + M_BIAS);
        if (tid < 64) sbw[tid] = p.b2[tid];
        if (tid < 256) sbw[64 + tid] = p.b3[tid];
        if (tid < CN) sbw[320 + tid] = p.bn[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int cb = wave & 3, rh = wave >> 2;        // phase B: channel block, row half
    u32x4 af2[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) af2[s] = sm[(cb * 18 + s) * 64 + lane];
    const int dcb = wave / WPC, dph = wave % WPC;   // phase D: channel block, which NPX pixel blocks of a y half
    u32x4 an[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) an[ks] = *(const u32x4 *)(p.wn + (dcb * 16 + l15) * 256 + ks * 32 + q * 8);
    __syncthreads();
    issue_t1(tile, 0);
    const float *sb = (const float *)(sm + M_BIAS);
    int slot = 0;

    for (;; ++kit) {
        int n, ty, tx;
        tile_xy(tile, n, ty, tx);
        const int next = tile_of(kit + 1);
        // this tile's halo was issued one tile ago; the residual loads issued behind it were waited for in phase C, so — vmcnt
        // retires in order — the DMA has landed for this wave.  The first tile waits explicitly.
#ifdef USOT_BNECK_VMCNT0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (kit == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();                             // #1: t1(tile) visible; the other slot, t2 and the y half are free
        if (next < p.ntiles) issue_t1(next, slot ^ 1);
        // residual: lane (l15, q) of wave w = channels w*32 + q*8 .. +7 of pixel (row pb, column l15): one 16-byte load per row
        const int ox = tx * TW + l15;
        const long pix0 = ((long)n * p.H + ty * TH) * p.W + ox;
        const char *rb = (const char *)(p.res + pix0 * 256 + wave * 32 + q * 8);
        u32x4 rr[8];
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
            rr[pb] = u32x4{0u, 0u, 0u, 0u};
            if (ty * TH + pb < p.H && ox < p.W) rr[pb] = *(const u32x4 *)(rb + (uint32_t)(pb * p.W) * 512u);
        }
        const u32x4 *t1s = sm + M_T1 + slot * T1SLOT;
        u32x4 *t2 = sm + M_T2, *yh = sm + M_YH;

        // ---------------- B: conv2.  wave (cb, rh): channels cb*16 .. +15 x tile rows rh*4 .. rh*4 + 3
        {
            f32x4 acc2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const u32x4 *hb = t1s + ((rh * 4) * HC + l15) * TPS + q;
            u32x4 bf[2][4];
            auto read_b = [&](int s, u32x4 (&b)[4]) {
                const int t = s >> 1, ks = s & 1, kh = t / 3, kw = t - kh * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = hb[((j + kh) * HC + kw) * TPS + ks * 4];
            };
            read_b(0, bf[0]);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18) read_b(s + 1, bf[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[j] = bk_mfma<F16>(af2[s], bf[s & 1][j], acc2[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const f32x4 bv = *(const f32x4 *)(sb + cb * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc2[j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]); o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                *(u32x2 *)((char *)(t2 + ((rh * 4 + j) * 16 + l15) * TPS) + cb * 32 + q * 8) = o;
            }
        }
        __syncthreads();                             // #2: t2 complete

        // ---------------- C + D, 64 pixels (four tile rows) at a time
        const f32x4 ba = *(const f32x4 *)(sb + 64 + wave * 32 + q * 8), bb = *(const f32x4 *)(sb + 64 + wave * 32 + q * 8 + 4);
        const f32x4 bnv = *(const f32x4 *)(sb + 320 + dcb * 16 + q * 4);
        char *yb = (char *)(p.y + pix0 * 256 + wave * 32 + q * 8);
        char *tb = (char *)(p.t + pix0 * CN + dcb * 16 + q * 4);
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc[i][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = sm[M_W3F + ((wave * 2 + i) * 2 + ks) * 64 + lane];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) b[pb] = t2[((hp * 4 + pb) * 16 + l15) * TPS + ks * 4 + q];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][pb] = bk_mfma<F16>(a[i], b[pb], acc[i][pb]);
            }
            if (hp == 1) __syncthreads();            // #5: phase D of the first half has read the y half
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int row = hp * 4 + pb;
                const u32x4 r = rr[row];
                f32x4 v0 = acc[0][pb] + ba, v1 = acc[1][pb] + bb;
                v0[0] += bk_unpack<F16>(r[0] & 0xffffu); v0[1] += bk_unpack<F16>(r[0] >> 16);
                v0[2] += bk_unpack<F16>(r[1] & 0xffffu); v0[3] += bk_unpack<F16>(r[1] >> 16);
                v1[0] += bk_unpack<F16>(r[2] & 0xffffu); v1[1] += bk_unpack<F16>(r[2] >> 16);
                v1[2] += bk_unpack<F16>(r[3] & 0xffffu); v1[3] += bk_unpack<F16>(r[3] >> 16);
                const u32x4 o = bk_pack8<F16>(v0, v1, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f});
                if (ty * TH + row < p.H && ox < p.W) *(u32x4 *)(yb + (uint32_t)(row * p.W) * 512u) = o;
                yh[(pb * 16 + l15) * YPS + wave * 4 + q] = o;
            }
            __syncthreads();                         // #3 / #6: the y half is complete
            f32x4 accd[NPX];
#pragma unroll
            for (int j = 0; j < NPX; ++j) accd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                u32x4 b[NPX];
#pragma unroll
                for (int j = 0; j < NPX; ++j) b[j] = yh[((dph * NPX + j) * 16 + l15) * YPS + ks * 4 + q];
#pragma unroll
                for (int j = 0; j < NPX; ++j) accd[j] = bk_mfma<F16>(an[ks], b[j], accd[j]);
            }
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int row = hp * 4 + dph * NPX + j;
                f32x4 v = accd[j] + bnv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]); o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                if (ty * TH + row < p.H && ox < p.W) *(u32x2 *)(tb + (uint32_t)(row * p.W) * (uint32_t)(CN * 2)) = o;
            }
        }
        if (next >= p.ntiles) break;
        tile = next;
        slot ^= 1;
    }
}

}  // namespace

extern "C" int usot_bneck_first_supported(int Cin, int Cmid, int Cout, int Cnext)
{
    return Cin == 64 && Cmid == 64 && Cout == 256 && Cnext == 64;
}

/* Layer1's first bottleneck + the next block's conv1 in one launch (see the top of this file).  NHWC dense, storage type
 * dtype 0 = bf16 | 1 = fp16; w1 [64][64], w2 [64][576] (k = (kh*3 + kw)*64 + ci), w3c [256][128] = [conv3 | downsample] along
 * k, wn [64][256]; biases fp32 (b3c = conv3's + the downsample's); y [N][H][W][256], t [N][H][W][64]. */
extern "C" int usot_bneck_first_lp(void *stream, const usot_bneck_desc *d, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!d || !d->x || !d->w1 || !d->w2 || !d->w3c || !d->wn || !d->b1 || !d->b2 || !d->b3c || !d->bn || !d->y || !d->t) return USOT_EINVAL;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (((uintptr_t)d->x | (uintptr_t)d->w1 | (uintptr_t)d->w2 | (uintptr_t)d->w3c | (uintptr_t)d->wn | (uintptr_t)d->b1 |
         (uintptr_t)d->b2 | (uintptr_t)d->b3c | (uintptr_t)d->bn | (uintptr_t)d->y | (uintptr_t)d->t) & 15) return USOT_EINVAL;
    BneckK p;
    p.x = (const uint16_t *)d->x; p.w1 = (const uint16_t *)d->w1; p.w2 = (const uint16_t *)d->w2;
    p.w3c = (const uint16_t *)d->w3c; p.wn = (const uint16_t *)d->wn;
    p.b1 = d->b1; p.b2 = d->b2; p.b3c = d->b3c; p.bn = d->bn;
    p.y = (uint16_t *)d->y; p.t = (uint16_t *)d->t;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.tiles_x = usot_cdiv(d->W, TW); p.tiles_y = usot_cdiv(d->H, TH);
    const long nt = (long)p.tiles_x * p.tiles_y * d->N;
    if (nt > 0x7fffffffL || (long)d->H * d->W * 128 >= 0x7fffffffL) return USOT_EINVAL;      // 32-bit offsets inside an image
    p.ntiles = (int)nt;
    static const uint16_t *zero_page_d[USOT_MAX_DEV] = {};
    const uint16_t *&zero_page = zero_page_d[usot_dv];
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!zero_page) {
        void *zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(bk_zero16)) != hipSuccess || !zp) return USOT_ELAUNCH;
        zero_page = (const uint16_t *)zp;
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    p.zero = zero_page;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)bneck_first_kernel<true> : (const void *)bneck_first_kernel<false>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    // grid: a multiple of 8 (the XCD-grouped tile walk), at most one workgroup per CU, every block with a first tile:
    // tile_of(0) of block b = (b & 7) * (grid / 8) + (b >> 3) < grid <= ntiles
    int grid = cus - (cus & 7);
    if (grid > p.ntiles) grid = p.ntiles - (p.ntiles & 7);
    if (grid < 8) return USOT_EINVAL;               // fewer than eight tiles: not this kernel's regime
    if (dtype) hipLaunchKernelGGL(bneck_first_kernel<true>, dim3(grid), dim3(512), BK_LDS, (hipStream_t)stream, p);
    else       hipLaunchKernelGGL(bneck_first_kernel<false>, dim3(grid), dim3(512), BK_LDS, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_bneck_tail_supported(int Cmid, int Cout, int Cnext)
{
    return Cmid == 64 && Cout == 256 && (Cnext == 64 || Cnext == 128);
}

/* The rest of a layer1 bottleneck after its conv1 + the next block's conv1 in one launch (bneck_tail_kernel above).  NHWC dense,
 * storage type dtype 0 = bf16 | 1 = fp16: d->x = t1 [N][H][W][64] (this block's conv1 output, after ReLU), d->w1 = the residual
 * [N][H][W][256] (the block's input), d->w2 [64][576], d->w3c = w3 [256][64], d->wn [Cnext][256]; biases d->b2, d->b3c (= b3),
 * d->bn (d->b1 unused); y [N][H][W][256] = relu(conv3(relu(conv2 t1)) + residual), t [N][H][W][Cnext] = relu(conv1'(y)). */
extern "C" int usot_bneck_tail_lp(void *stream, const usot_bneck_desc *d, int Cnext, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!d || !d->x || !d->w1 || !d->w2 || !d->w3c || !d->wn || !d->b2 || !d->b3c || !d->bn || !d->y || !d->t) return USOT_EINVAL;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || (dtype != 0 && dtype != 1) || !usot_bneck_tail_supported(64, 256, Cnext)) return USOT_EINVAL;
    if (((uintptr_t)d->x | (uintptr_t)d->w1 | (uintptr_t)d->w2 | (uintptr_t)d->w3c | (uintptr_t)d->wn |
         (uintptr_t)d->b2 | (uintptr_t)d->b3c | (uintptr_t)d->bn | (uintptr_t)d->y | (uintptr_t)d->t) & 15) return USOT_EINVAL;
    BneckTK p;
    p.t1 = (const uint16_t *)d->x; p.res = (const uint16_t *)d->w1; p.w2 = (const uint16_t *)d->w2;
    p.w3 = (const uint16_t *)d->w3c; p.wn = (const uint16_t *)d->wn;
    p.b2 = d->b2; p.b3 = d->b3c; p.bn = d->bn;
    p.y = (uint16_t *)d->y; p.t = (uint16_t *)d->t;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.tiles_x = usot_cdiv(d->W, TW); p.tiles_y = usot_cdiv(d->H, TH);
    const long nt = (long)p.tiles_x * p.tiles_y * d->N;
    if (nt > 0x7fffffffL || (long)d->H * d->W * 512 >= 0x7fffffffL) return USOT_EINVAL;      // 32-bit offsets inside an image
    p.ntiles = (int)nt;
    static const uint16_t *zero_page_d[USOT_MAX_DEV] = {};
    const uint16_t *&zero_page = zero_page_d[usot_dv];
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!zero_page) {
        void *zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(bk_zero16)) != hipSuccess || !zp) return USOT_ELAUNCH;
        zero_page = (const uint16_t *)zp;
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    p.zero = zero_page;
    const int v = (dtype ? 2 : 0) + (Cnext == 128 ? 1 : 0);
    const void *fns[4] = {(const void *)bneck_tail_kernel<false, 64>, (const void *)bneck_tail_kernel<false, 128>,
                          (const void *)bneck_tail_kernel<true, 64>, (const void *)bneck_tail_kernel<true, 128>};
    static bool raised_d[USOT_MAX_DEV][4] = {};
    bool (&raised)[4] = raised_d[usot_dv];
    if (!raised[v]) {
        if (hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS) != hipSuccess) return USOT_ELAUNCH;
        raised[v] = true;
    }
    int grid = cus - (cus & 7);
    if (grid > p.ntiles) grid = p.ntiles - (p.ntiles & 7);
    if (grid < 8) return USOT_EINVAL;
    switch (v) {
    case 0: hipLaunchKernelGGL((bneck_tail_kernel<false, 64>), dim3(grid), dim3(512), BT_LDS, (hipStream_t)stream, p); break;
    case 1: hipLaunchKernelGGL((bneck_tail_kernel<false, 128>), dim3(grid), dim3(512), BT_LDS, (hipStream_t)stream, p); break;
    case 2: hipLaunchKernelGGL((bneck_tail_kernel<true, 64>), dim3(grid), dim3(512), BT_LDS, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL((bneck_tail_kernel<true, 128>), dim3(grid), dim3(512), BT_LDS, (hipStream_t)stream, p); break;
    }
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
