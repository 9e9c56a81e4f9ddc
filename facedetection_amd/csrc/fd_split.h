// Two-way fp16 splitting of float32 values, the operand format of the split-precision
// convolution (fd_conv_split.hip).
//
// A float32 value v is carried as two fp16 numbers  hi = fp16(v),  lo = fp16(v - hi)
// (both round-to-nearest-even; v - hi is exact in float32).  hi + lo reproduces v to about 22
// significant bits, and a product a*b is formed on the f16 matrix cores as
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (each exact in the fp32 accumulator),
// dropping only a_lo*b_lo (<= 2^-22 |a*b|).  That is float32-grade arithmetic at three
// f16 MFMAs per multiply, 16/3 of the fp32-MFMA rate.
//
// Memory layout of a split tensor with C channels per pixel (C % 32 == 0): the same 4 bytes
// per element as float32, arranged per pixel and per group of 32 channels as
//     [ 32 x hi (64 B) | 32 x lo (64 B) ]
// so one 128-byte group is exactly one K-step row of the GEMM staging buffer.  The 4-channel
// staged image uses [ 4 x hi | 4 x lo ] (16 B) per pixel.
//
// Range: |v| is clamped to 65504 (fp16 max); values below 2^-14 lose relative (not absolute)
// precision as lo enters the subnormal range (spacing 2^-24).  Every split function counts the
// values it had to clamp (|v| > 65504, +-Inf) or could not represent (NaN) in a per-thread
// counter; a kernel adds its threads' counts to a device counter when it ends (split_report), so
// the host can tell that a result left the range in which this format is as good as float32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float F16_MAX = 65504.0f;

__device__ __forceinline__ void split_f32(float v, _Float16& hi, _Float16& lo, unsigned& bad) {
    bad += !(fabsf(v) <= F16_MAX);             // also true for NaN
    v = fminf(fmaxf(v, -F16_MAX), F16_MAX);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// relu(v) split: the ReLU and the upper clamp are one median-of-three
__device__ __forceinline__ void split_f32_relu(float v, _Float16& hi, _Float16& lo, unsigned& bad) {
    bad += !(v <= F16_MAX);                    // also true for NaN
    v = __builtin_amdgcn_fmed3f(v, 0.0f, F16_MAX);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// 8 consecutive channels -> 8 hi halves + 8 lo halves (16 bytes each)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo, unsigned& bad) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 h, l;
        split_f32(a[j], h, l, bad);
        hi[j] = h; lo[j] = l;
        split_f32(b[j], h, l, bad);
        hi[4 + j] = h; lo[4 + j] = l;
    }
}

// adds a thread's count of out-of-range values to the device counter (NULL: not monitored)
__device__ __forceinline__ void split_report(unsigned* counter, unsigned bad) {
    if (bad != 0u && counter) atomicAdd(counter, bad);
}

// byte offset of the hi half of channel c (c % 8 == 0 for 16-byte accesses) inside a pixel row of C channels
__device__ __forceinline__ size_t split_offset(int c) { return (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2; }

}  // namespace fd
