// 16-bit operand splits shared by the MFMA conv kernels (internal).
//   bf16x3: x = x1 + x2 + x3 exactly (24 bits), 6 bf16 MFMAs per product block;
//   f16x2 : x = x1 + 2^-11 x2' (22 bits, x2' = f16((x - x1) * 2^11) stays clear of the fp16 subnormals), 3 fp16 MFMAs into two
//           accumulators A += x1*w1, B += x1*w2' + x2'*w1, result A + 2^-11 B.  fp16 range only (|x| <= 65504).
#pragma once
#include "cd_common.h"

namespace cd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

__device__ __forceinline__ unsigned pack_h2(_Float16 lo, _Float16 hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}
// two-term split of 4 floats: t1 = f16(x), t2 = f16((x - t1) * 2^11)   (round to nearest even).
// Two-element vector form: gfx950's v_cvt_pk_f16_f32 converts a pair per instruction and the remainder is formed with packed
// fp32 arithmetic -- 6 VALU instructions per pair instead of 12 (the staging waves of the conv kernels are VALU-bound).
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(const f32x4 x, u32x2& t1, u32x2& t2) {
  const f32x2v a = {x[0], x[1]}, b = {x[2], x[3]};
  const f16x2v ha = __builtin_convertvector(a, f16x2v), hb = __builtin_convertvector(b, f16x2v);
  const f32x2v ra = (a - __builtin_convertvector(ha, f32x2v)) * 2048.f, rb = (b - __builtin_convertvector(hb, f32x2v)) * 2048.f;
  const f16x2v la = __builtin_convertvector(ra, f16x2v), lb = __builtin_convertvector(rb, f16x2v);
  t1 = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  t2 = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

}  // namespace cd
