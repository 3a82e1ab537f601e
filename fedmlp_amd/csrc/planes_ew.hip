// Producers of activation planes (ResNet-18 planes mode, pconv.hip / pwgrad.hip): the BatchNorm apply, max-pool and
// BatchNorm-backward apply passes of elementwise.hip, writing their result as block-major bf16 planes
// P[C/32][3][pixels][32] (x = h + m + l exactly, split3.h) for the conv GEMMs that consume it -- and as fp32 only where a
// caller still wants the tensor that way.  Same arithmetic, expression for expression, as the fp32 kernels they stand in for
// (the reference ops: BatchNorm2d / ReLU / MaxPool2d forward and backward inside net(images) and loss.backward(),
// utils/local_training.py:657, 674, 937-947, 965; torchvision resnet18, model/all_models.py:53-54).
//
// HBM-bound.  Thread mapping = the fp32 kernels': one thread = 4 consecutive channels of one pixel (one 16-B access per
// operand, 16+ consecutive lanes = one pixel's channels: whole 128-B lines per wave instruction).  A plane chunk is 8
// channels, 4g..4g+3 and 16+4g..16+4g+3 of a 32-channel block: lanes cq and cq + 4 of a pixel hold its two halves; the lower
// lane fetches the upper one's values by a lane shift, splits the 8 values and writes the three 16-B plane chunks (four such
// lanes = 64 contiguous bytes of each plane).  Tensors kept ONLY as planes are read back the same way: a thread re-forms its 4
// channels from 8 bytes of each plane (x = (h + m) + l, both additions exact), a ReLU mask needs the h plane's sign only.
#include "common.h"
#include "kernels.h"
#include "split3.h"

static inline int pe_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#if __HIP_DEVICE_COMPILE__
// byte offset of channels c .. c + 3 (c % 4 == 0) of pixel `pix` inside plane 0 of a planes tensor of P pixels
__device__ __forceinline__ size_t pe_off(int c, int64_t pix, int64_t P)
{
    return (((size_t)(c >> 5) * 3) * P + pix) * 64 + ((c & 15) >> 2) * 16 + ((c >> 4) & 1) * 8;
}
// v = this thread's channels c .. c + 3; every lane of the wave calls this (the exchange is a wave operation)
__device__ __forceinline__ void pe_store_planes(unsigned short* planes, int64_t P, int c, int64_t pix, f32x4 v, bool active)
{
    f32x4 up;          // the values of lane + 4: channels c + 16 .. c + 19 of the same pixel
#pragma unroll
    for (int k = 0; k < 4; ++k) up[k] = __shfl_down(v[k], 4);
    if (active && (c & 16) == 0) {
        sp_u32x4 H, M, L;
        split3(v, up, H, M, L);
        unsigned char* d = reinterpret_cast<unsigned char*>(planes) + (((size_t)(c >> 5) * 3) * P + pix) * 64 + ((c & 15) >> 2) * 16;
        *reinterpret_cast<sp_u32x4*>(d) = H;
        *reinterpret_cast<sp_u32x4*>(d + (size_t)P * 64) = M;
        *reinterpret_cast<sp_u32x4*>(d + (size_t)P * 128) = L;
    }
}
// channels c .. c + 3 back from the planes: x = (h + m) + l
__device__ __forceinline__ f32x4 pe_load_planes(const unsigned short* planes, int64_t P, int c, int64_t pix)
{
    const unsigned char* s = reinterpret_cast<const unsigned char*>(planes) + pe_off(c, pix, P);
    const uint2 H = *reinterpret_cast<const uint2*>(s);
    const uint2 M = *reinterpret_cast<const uint2*>(s + (size_t)P * 64);
    const uint2 L = *reinterpret_cast<const uint2*>(s + (size_t)P * 128);
    f32x4 v;
    v[0] = (__builtin_bit_cast(float, H.x << 16) + __builtin_bit_cast(float, M.x << 16)) + __builtin_bit_cast(float, L.x << 16);
    v[1] = (__builtin_bit_cast(float, H.x & 0xffff0000u) + __builtin_bit_cast(float, M.x & 0xffff0000u)) + __builtin_bit_cast(float, L.x & 0xffff0000u);
    v[2] = (__builtin_bit_cast(float, H.y << 16) + __builtin_bit_cast(float, M.y << 16)) + __builtin_bit_cast(float, L.y << 16);
    v[3] = (__builtin_bit_cast(float, H.y & 0xffff0000u) + __builtin_bit_cast(float, M.y & 0xffff0000u)) + __builtin_bit_cast(float, L.y & 0xffff0000u);
    return v;
}
// d masked by z > 0 where z is kept only as planes: the sign of its h plane (z > 0 <=> bf16(z) > 0 for every normal float)
__device__ __forceinline__ f32x4 pe_mask_h(f32x4 d, const unsigned short* zh, int64_t P, int c, int64_t pix)
{
    const uint2 hb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(zh) + pe_off(c, pix, P));
    d[0] = (short)(hb.x & 0xffffu) > 0 ? d[0] : 0.f;
    d[1] = (short)(hb.x >> 16) > 0 ? d[1] : 0.f;
    d[2] = (short)(hb.y & 0xffffu) > 0 ? d[2] : 0.f;
    d[3] = (short)(hb.y >> 16) > 0 ? d[3] : 0.f;
    return d;
}
#endif

// out = [relu]( y*scale+shift [+ res] [+ y2*scale2+shift2] ), per group (elementwise.hip bn_apply_kernel)
__global__ __launch_bounds__(256) void bn_apply_planes_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ res,
                                                              const float* __restrict__ y2, const float* __restrict__ scale2,
                                                              const float* __restrict__ shift2, float* __restrict__ out,
                                                              unsigned short* __restrict__ outp, int pix_per_group, int C, int relu,
                                                              const unsigned short* __restrict__ resp)
{
#if __HIP_DEVICE_COMPILE__
    const int grp = blockIdx.y;
    const int Q = C >> 2;
    const int64_t n4 = (int64_t)pix_per_group * Q;
    const int64_t P = (int64_t)gridDim.y * pix_per_group;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n4;
    const int64_t ii = active ? i : 0;
    const int c = (int)(ii % Q) * 4;
    const int64_t pix = (int64_t)grp * pix_per_group + ii / Q;
    const size_t o = (size_t)pix * C + c;
    f32x4 v = ld4(y + o) * ld4(scale + grp * C + c) + ld4(shift + grp * C + c);
    if (res) v += ld4(res + o);
    if (resp) v += pe_load_planes(resp, P, c, pix);
    if (y2) v += ld4(y2 + o) * ld4(scale2 + grp * C + c) + ld4(shift2 + grp * C + c);
    if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (out && active) st4(out + o, v);
    pe_store_planes(outp, P, c, pix, v, active);
#endif
}
void k_bn_apply_planes(const float* y, const float* scale, const float* shift, const float* res, const float* y2, const float* scale2,
                       const float* shift2, float* out, unsigned short* outp, int groups, int pix_per_group, int C, int relu,
                       hipStream_t s, const unsigned short* resp)
{
    const int64_t n4 = (int64_t)pix_per_group * (C / 4);
    hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(pe_cdiv(n4, 256), groups), dim3(256), 0, s, y, scale, shift, res, y2, scale2,
                       shift2, out, outp, pix_per_group, C, relu, resp);
}

// stem: pooled = maxpool3x3s2p1(relu(y*scale+shift)) (+ argmax code); scale == null -> plain max-pool (elementwise.hip stem_pool_kernel)
__global__ __launch_bounds__(256) void stem_pool_planes_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ pooled,
                                                               uint8_t* __restrict__ idx, unsigned short* __restrict__ pooledp,
                                                               int imgs_per_group, int H, int W, int C)
{
#if __HIP_DEVICE_COMPILE__
    const int grp = blockIdx.y;
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2;
    const int64_t ppg = (int64_t)imgs_per_group * Hp * Wp;          // pooled pixels per group
    const int64_t n = ppg * Q;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i0 < n;
    const int64_t i = active ? i0 : 0;
    const int cq = (int)(i % Q), c = cq * 4;
    const int64_t pixg = i / Q;
    int64_t t = pixg;
    const int ow = (int)(t % Wp); t /= Wp;
    const int oh = (int)(t % Hp);
    const int img = grp * imgs_per_group + (int)(t / Hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) { sc = ld4(scale + grp * C + c); sh = ld4(shift + grp * C + c); }
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int code[4] = {0, 0, 0, 0};
    // Round 6: the nine window loads are UNCONDITIONAL (clamped coordinates) and a position outside the map takes -inf afterwards
    // -- never chosen under the strict >, exactly like the skipped position it was: with `continue` in front of each load the
    // compiler waited for every load before it issued the next (nine dependent round trips per thread, tools/wait_chains.py)
    f32x4 win[9];
    bool inside[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        const bool rok = (unsigned)ih < (unsigned)H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            inside[kh * 3 + kw] = rok && (unsigned)iw < (unsigned)W;
            win[kh * 3 + kw] = ld4(y + ((size_t)(img * H + (rok ? ih : oh * 2)) * W + ((unsigned)iw < (unsigned)W ? iw : ow * 2)) * C + c);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        f32x4 v = win[t];
        if (scale) {
            v = v * sc + sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float u = inside[t] ? v[k] : -INFINITY;
            if (u > best[k]) { best[k] = u; code[k] = t; }
        }
    }
    const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + c;
    if (active) {
        if (pooled) st4(pooled + o, best);
        if (idx)
            *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((unsigned char)code[0], (unsigned char)code[1], (unsigned char)code[2],
                                                              (unsigned char)code[3]);
    }
    pe_store_planes(pooledp, (int64_t)gridDim.y * ppg, c, (int64_t)grp * ppg + pixg, best, active);
#endif
}
void k_stem_pool_planes(const float* y, const float* scale, const float* shift, float* pooled, uint8_t* idx, unsigned short* pooledp,
                        int groups, int imgs_per_group, int H, int W, int C, hipStream_t s)
{
    const int64_t n = (int64_t)imgs_per_group * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(stem_pool_planes_kernel, dim3(pe_cdiv(n, 256), groups), dim3(256), 0, s, y, scale, shift, pooled, idx, pooledp,
                       imgs_per_group, H, W, C);
}

// dy = ca*dyh + cb*y + cc, dyh = dz masked by z > 0 (z fp32, or the h plane zh of a planes-only z, or y*msc+msh > 0); optionally
// store dyh (elementwise.hip bn_bwd_apply_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                                  const float* __restrict__ y, const float* __restrict__ ca,
                                                                  const float* __restrict__ cb_, const float* __restrict__ cc, float* dy,
                                                                  unsigned short* __restrict__ dyp, float* dyh_out, int pix_per_group,
                                                                  int C, const float* __restrict__ msc, const float* __restrict__ msh,
                                                                  const unsigned short* __restrict__ zh)
{
#if __HIP_DEVICE_COMPILE__
    const int grp = blockIdx.y;
    const int Q = C >> 2;
    const int64_t n4 = (int64_t)pix_per_group * Q;
    const int64_t P = (int64_t)gridDim.y * pix_per_group;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n4;
    const int64_t ii = active ? i : 0;
    const int c = (int)(ii % Q) * 4;
    const int64_t pix = (int64_t)grp * pix_per_group + ii / Q;
    const size_t o = (size_t)pix * C + c;
    f32x4 d = ld4(dz + o);
    if (zh) d = pe_mask_h(d, zh, P, c, pix);
    if (z) {
        const f32x4 zz = ld4(z + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = zz[k] > 0.f ? d[k] : 0.f;
    }
    const f32x4 yy = ld4(y + o);
    if (msc) {
        const f32x4 sc = ld4(msc + grp * C + c), sh = ld4(msh + grp * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = __builtin_fmaf(yy[k], sc[k], sh[k]) > 0.f ? d[k] : 0.f;
    }
    const f32x4 r = ld4(ca + grp * C + c) * d + ld4(cb_ + grp * C + c) * yy + ld4(cc + grp * C + c);
    if (active) {
        if (dyh_out) st4(dyh_out + o, d);
        if (dy) st4(dy + o, r);
    }
    pe_store_planes(dyp, P, c, pix, r, active);
#endif
}
void k_bn_bwd_apply_planes(const float* dz, const float* z, const float* y, const float* ca, const float* cb, const float* cc, float* dy,
                           unsigned short* dyp, float* dyh_out, int groups, int pix_per_group, int C, hipStream_t s,
                           const float* mask_scale, const float* mask_shift, const unsigned short* zh)
{
    const int64_t n4 = (int64_t)pix_per_group * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(pe_cdiv(n4, 256), groups), dim3(256), 0, s, dz, z, y, ca, cb, cc, dy, dyp,
                       dyh_out, pix_per_group, C, mask_scale, mask_shift, zh);
}
