// Shared helpers for the libusot_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define USOT_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH; \
    } while (0)

static inline int usot_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The launchers keep per-DEVICE facts in function-local statics: raised dynamic-LDS limits (hipFuncSetAttribute), the address
// of a __device__ zero page (hipGetSymbolAddress), the CU count.  All of these belong to ONE device; the execution model is
// Per-device launcher state (round 6).  Launchers cache what they learn about a device - the address of a zero page in device memory,
// whether a kernel's dynamic-LDS limit was raised, the CU count, resident-workgroup counts - in function-local statics.  Those are now
// arrays indexed by the CURRENT HIP device (usot_device_slot(): hipGetDevice, 0 .. USOT_MAX_DEV - 1), so one process may drive several
// GPUs (SURVEY 8e's "one process + per-device streams" form) as well as one; a device index beyond the table returns USOT_ESTATE.
// Initialisation races between host threads are benign (idempotent values).  usot_device_guard() stays in the ABI: USOT_OK when the
// current device has a slot.  Both are defined in head_ops.hip.
constexpr int USOT_MAX_DEV = 16;
extern "C" int usot_device_slot(void);
extern "C" int usot_device_guard(void);

// Running total of the finished k-blocks of an MFMA accumulator (4 floats per lane); see conv_igemm.hip: blocked_mma.
// USOT_TOT selects its arithmetic (a build-time experiment switch; scripts/tot_variants.py builds and times all of them).
// Measured on one MI355X box, frame graph replay / mean rms ratio of HIP-vs-float64 to reference-float32-vs-float64 over the
// 34 fixtures of tests/golden/f64_gate.py ('zero_dc', 'dc'):
//   1  plain float32 adds (default)     873.6 us   0.38 / 0.35    (K/64 sequential roundings of the growing total)
//   2  float64 total                    931.3 us   0.37 / 0.35    (v_cvt_f64_f32 + v_add_f64 per register and block)
//   4  float32 Kahan-compensated total  948.9 us   0.37 / 0.35
// i.e. once the products are summed in 64-wide blocks the rounding of the TOTAL is no longer what separates the frame from
// float64 (activation roundings between ~60 layers are), and the compensated forms buy nothing for 6-8 % of the frame.
#ifndef USOT_TOT
#define USOT_TOT 1
#endif
typedef float usot_f32x4 __attribute__((ext_vector_type(4)));
typedef double usot_f64x4 __attribute__((ext_vector_type(4)));
struct BlockTotal {
#if USOT_TOT == 2
    usot_f64x4 s;
    __device__ __forceinline__ void clear() { s = usot_f64x4{0., 0., 0., 0.}; }
    __device__ __forceinline__ void add(usot_f32x4 p) { s += __builtin_convertvector(p, usot_f64x4); }
    __device__ __forceinline__ usot_f32x4 get() const { return __builtin_convertvector(s, usot_f32x4); }
#elif USOT_TOT == 4
    usot_f32x4 s, c;                 // c = what the additions so far have lost, negated (Kahan)
    __device__ __forceinline__ void clear() { s = c = usot_f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void add(usot_f32x4 p)
    {
        const usot_f32x4 y = p - c;
        const usot_f32x4 t = s + y;
        c = (t - s) - y;             // no -ffast-math in this build: evaluated as written
        s = t;
    }
    __device__ __forceinline__ usot_f32x4 get() const { return s - c; }
#else
    usot_f32x4 s;
    __device__ __forceinline__ void clear() { s = usot_f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void add(usot_f32x4 p) { s += p; }
    __device__ __forceinline__ usot_f32x4 get() const { return s; }
#endif
};

// two floats -> one dword of the low-precision storage type (low half = a), round to nearest even: ONE v_cvt_pk_bf16_f32 on
// gfx950 (the shift / add / mask form of the same rounding is ~10 VALU instructions per pair; the epilogues of the HBM-bound
// 1x1 layers spend a third of their time there), v_cvt_f16_f32 x 2 + v_pack_b32_f16 for fp16
typedef float usot_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 usot_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 usot_f16x2 __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ uint32_t usot_pack2_lp(float a, float b)
{
    const usot_f32x2 v = {a, b};
    if constexpr (F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, usot_f16x2));
    else               return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, usot_bf16x2));
}

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
