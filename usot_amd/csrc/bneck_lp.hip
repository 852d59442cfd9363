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
    // L & 7 holds logical chunk pc ^ swz(P)
    auto issue_x = [&](int tile, int slot) {
        int n, ty, tx;
        tile_xy(tile, n, ty, tx);
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
        for (int i = 0; i < XPIECES / 8; ++i) {
            const int piece = i * 8 + wave;
            const int L = piece * 60 + lane;
            const int P = L >> 3, pc = L & 7;
            const int hy = P / HC, hx = P - hy * HC;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint16_t *src = ok ? p.x + (((long)n * p.H + iy) * p.W + ix) * 64 + ((pc ^ bk_swz(hx)) << 3) : p.zero;
            if (lane < 60) dma16(src, lds0 + (uint32_t)((L_XR + slot * XSLOT + piece * 60) * 16));
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
        switch (stores_prev) {
#define BK_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        BK_W(12) BK_W(11) BK_W(10) BK_W(9) BK_W(8) BK_W(7) BK_W(6) BK_W(5) BK_W(4) BK_W(3) BK_W(2) BK_W(1)
#undef BK_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
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
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int oy = ty * TH + hp * 4 + pb;
                yo[hp * 4 + pb] = bk_pack8<F16>(acc[0][pb], acc[1][pb], ba, bb);
#ifdef USOT_BKABL_NOSTORE
                if (oy < p.H && ox < p.W && yo[hp * 4 + pb][0] == 0x12345678u)
#else
                if (oy < p.H && ox < p.W)
#endif
                    *(u32x4 *)(p.y + (((long)n * p.H + oy) * p.W + ox) * 256 + wave * 32 + q * 8) = yo[hp * 4 + pb];
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
                    *(u32x2 *)(p.t + (((long)n * p.H + oy) * p.W + ox) * 64 + cb * 16 + q * 4) = o;
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
    if (nt > 0x7fffffffL) return USOT_EINVAL;
    p.ntiles = (int)nt;
    static const uint16_t *zero_page = nullptr;
    static int cus = 0;
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
    static bool raised[2] = {false, false};
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
