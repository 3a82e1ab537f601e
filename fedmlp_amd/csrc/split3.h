// fp32 products on the bf16 matrix pipe (gfx950): the exact three-way split of an fp32 value into bf16 planes.
//
//   x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = x - h - m      (round to nearest even)
//
// Both subtractions are exact (x - h has at most 16 significant bits, x - h - m at most 8, so l IS a bf16), hence the nine
// products {h,m,l}(a) x {h,m,l}(b) sum to a*b exactly, and each of them is exact in fp32 (8 x 8 significant bits).
// v_mfma_f32_16x16x32_bf16 forms them and accumulates in fp32: 9 bf16 MFMAs (16 cycles each) replace the 8 fp32 MFMAs
// (32 cycles each) of the same 16x16x32 volume.  SP = 6 leaves out m*l, l*m and l*l: each is at most 2^-24 of |a*b|
// (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|; measured over 2.6e5 random pairs: 2^-24.3 for the three together), i.e. the K-term dot
// product's error bound of K unit roundoffs (2^-24 each, from the fp32 accumulation) grows by at most two: (K + 2) u for K = 64..4608.
// tests/test_split3_cpu.py restates the arithmetic in numpy.
// Inf turns into NaN (inf - inf); training tensors hold neither.
#pragma once
#include "common.h"

#if defined(__HIPCC__)
typedef unsigned int sp_u32x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));

// a - b as ONE v_sub_f32: beside MFMAs a packed v_pk_add_f32 (which -O3 forms from two adjacent subtractions) costs more
// issue time than the two plain instructions (MI355X_MICROARCH.md, per-instruction constants)
__device__ __forceinline__ float sp_sub(float a, float b)
{
#ifdef FM_SPLIT_PK
    return a - b;
#else
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
// planes of the pair (a, b): element a in the low half of each word, b in the high half
__device__ __forceinline__ void split3_pair(const sp_f32x2 e, unsigned& H, unsigned& M, unsigned& L)
{
    const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(e, sp_bf16x2));
    const sp_f32x2 r = {sp_sub(e.x, __builtin_bit_cast(float, hw << 16)), sp_sub(e.y, __builtin_bit_cast(float, hw & 0xffff0000u))};
    const unsigned mw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sp_bf16x2));
    const sp_f32x2 l = {sp_sub(r.x, __builtin_bit_cast(float, mw << 16)), sp_sub(r.y, __builtin_bit_cast(float, mw & 0xffff0000u))};
    H = hw;
    M = mw;
    L = __builtin_bit_cast(unsigned, __builtin_convertvector(l, sp_bf16x2));
}
// 8 values (two 16-B fragments) -> three 8 x bf16 MFMA operands
__device__ __forceinline__ void split3(const f32x4 x0, const f32x4 x1, sp_u32x4& H, sp_u32x4& M, sp_u32x4& L)
{
    unsigned h[4], m[4], l[4];
    split3_pair(sp_f32x2{x0[0], x0[1]}, h[0], m[0], l[0]);
    split3_pair(sp_f32x2{x0[2], x0[3]}, h[1], m[1], l[1]);
    split3_pair(sp_f32x2{x1[0], x1[1]}, h[2], m[2], l[2]);
    split3_pair(sp_f32x2{x1[2], x1[3]}, h[3], m[3], l[3]);
    H = sp_u32x4{h[0], h[1], h[2], h[3]};
    M = sp_u32x4{m[0], m[1], m[2], m[3]};
    L = sp_u32x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x4 mfma_bf16(sp_u32x4 a, sp_u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += a * b from the planes: SP = 9 every partial product, SP = 6 those that matter in fp32; small terms first
template <int SP>
__device__ __forceinline__ f32x4 mfma_split(const sp_u32x4 ah, const sp_u32x4 am, const sp_u32x4 al, const sp_u32x4 bh,
                                            const sp_u32x4 bm, const sp_u32x4 bl, f32x4 v)
{
    if constexpr (SP == 9) {
        v = mfma_bf16(al, bl, v);
        v = mfma_bf16(al, bm, v);
        v = mfma_bf16(am, bl, v);
    }
    v = mfma_bf16(al, bh, v);
    v = mfma_bf16(ah, bl, v);
    v = mfma_bf16(am, bm, v);
    v = mfma_bf16(am, bh, v);
    v = mfma_bf16(ah, bm, v);
    v = mfma_bf16(ah, bh, v);
    return v;
}
#endif
