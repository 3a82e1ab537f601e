// Producers of activation planes (ResNet-18 planes mode, pconv.hip): the BatchNorm apply, max-pool and BatchNorm-backward apply
// passes of elementwise.hip, writing their result as block-major bf16 planes P[C/32][3][pixels][32] (x = h + m + l exactly,
// split3.h) for the conv GEMMs that consume it -- and as fp32 only where an elementwise consumer needs the tensor (residual
// adds, ReLU masks of a later backward, pooling).  Same arithmetic, expression for expression, as the fp32 kernels they stand
// in for (the reference ops: BatchNorm2d / ReLU / MaxPool2d forward and backward inside net(images) and loss.backward(),
// utils/local_training.py:657, 674, 937-947, 965; torchvision resnet18, model/all_models.py:53-54).
//
// HBM-bound.  One thread = one 8-channel chunk of one pixel: channels 4g..4g+3 and 16+4g..16+4g+3 of a 32-channel block (the chunk
// order of the planes), i.e. two 16-B reads 64 B apart per fp32 operand and three 16-B plane writes.  Threads are ordered
// (channel block, pixel, g): four lanes read one 128-B line, a wave writes 1 KB contiguous of each plane.
#include "common.h"
#include "kernels.h"
#include "split3.h"

static inline int pe_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#if __HIP_DEVICE_COMPILE__
__device__ __forceinline__ void pe_store_planes(unsigned short* planes, int64_t P, int cb, int64_t pix, int g, f32x4 v0, f32x4 v1)
{
    sp_u32x4 H, M, L;
    split3(v0, v1, H, M, L);
    unsigned char* d = reinterpret_cast<unsigned char*>(planes) + (((size_t)cb * 3) * P + pix) * 64 + g * 16;
    *reinterpret_cast<sp_u32x4*>(d) = H;
    *reinterpret_cast<sp_u32x4*>(d + (size_t)P * 64) = M;
    *reinterpret_cast<sp_u32x4*>(d + (size_t)P * 128) = L;
}
// the fp32 values back: x = (h + m) + l, both additions exact
__device__ __forceinline__ void pe_load_planes(const unsigned short* planes, int64_t P, int cb, int64_t pix, int g, f32x4& v0, f32x4& v1)
{
    const unsigned char* s = reinterpret_cast<const unsigned char*>(planes) + (((size_t)cb * 3) * P + pix) * 64 + g * 16;
    const sp_u32x4 H = *reinterpret_cast<const sp_u32x4*>(s);
    const sp_u32x4 M = *reinterpret_cast<const sp_u32x4*>(s + (size_t)P * 64);
    const sp_u32x4 L = *reinterpret_cast<const sp_u32x4*>(s + (size_t)P * 128);
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = (__builtin_bit_cast(float, H[i] << 16) + __builtin_bit_cast(float, M[i] << 16)) + __builtin_bit_cast(float, L[i] << 16);
        v[2 * i + 1] = (__builtin_bit_cast(float, H[i] & 0xffff0000u) + __builtin_bit_cast(float, M[i] & 0xffff0000u)) +
                       __builtin_bit_cast(float, L[i] & 0xffff0000u);
    }
    v0 = f32x4{v[0], v[1], v[2], v[3]};
    v1 = f32x4{v[4], v[5], v[6], v[7]};
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
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int cb = (int)(t / pix_per_group);
    if (cb >= (C >> 5)) return;
    const int64_t pix = (int64_t)grp * pix_per_group + (t - (int64_t)cb * pix_per_group);
    const int64_t P = (int64_t)gridDim.y * pix_per_group;
    const int c0 = cb * 32 + 4 * g;
    const size_t o = (size_t)pix * C + c0;
    f32x4 v[2], rp[2];
    if (resp) pe_load_planes(resp, P, cb, pix, g, rp[0], rp[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + 16 * h;
        v[h] = ld4(y + o + 16 * h) * ld4(scale + grp * C + c) + ld4(shift + grp * C + c);
        if (res) v[h] += ld4(res + o + 16 * h);
        if (resp) v[h] += rp[h];
        if (y2) v[h] += ld4(y2 + o + 16 * h) * ld4(scale2 + grp * C + c) + ld4(shift2 + grp * C + c);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[h][k] = fmaxf(v[h][k], 0.f);
        }
        if (out) st4(out + o + 16 * h, v[h]);
    }
    pe_store_planes(outp, P, cb, pix, g, v[0], v[1]);
#endif
}
void k_bn_apply_planes(const float* y, const float* scale, const float* shift, const float* res, const float* y2, const float* scale2,
                       const float* shift2, float* out, unsigned short* outp, int groups, int pix_per_group, int C, int relu,
                       hipStream_t s, const unsigned short* resp)
{
    const int64_t n = (int64_t)pix_per_group * (C / 8);
    hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(pe_cdiv(n, 256), groups), dim3(256), 0, s, y, scale, shift, res, y2, scale2,
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
    const int Hp = H / 2, Wp = W / 2;
    const int64_t ppg = (int64_t)imgs_per_group * Hp * Wp;          // pooled pixels per group
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(i & 3);
    const int64_t t = i >> 2;
    const int cb = (int)(t / ppg);
    if (cb >= (C >> 5)) return;
    int64_t r = t - (int64_t)cb * ppg;
    const int64_t pixg = r;
    const int ow = (int)(r % Wp); r /= Wp;
    const int oh = (int)(r % Hp);
    const int img = grp * imgs_per_group + (int)(r / Hp);
    const int c0 = cb * 32 + 4 * g;
    f32x4 best[2];
    int code[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + 16 * h;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) { sc = ld4(scale + grp * C + c); sh = ld4(shift + grp * C + c); }
        best[h] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int k = 0; k < 4; ++k) code[h][k] = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                f32x4 v = ld4(y + ((size_t)(img * H + ih) * W + iw) * C + c);
                if (scale) {
                    v = v * sc + sh;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[h][k]) { best[h][k] = v[k]; code[h][k] = kh * 3 + kw; }
            }
        }
        const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + c;
        if (pooled) st4(pooled + o, best[h]);
        if (idx)
            *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((unsigned char)code[h][0], (unsigned char)code[h][1],
                                                              (unsigned char)code[h][2], (unsigned char)code[h][3]);
    }
    pe_store_planes(pooledp, (int64_t)gridDim.y * ppg, cb, (int64_t)grp * ppg + pixg, g, best[0], best[1]);
#endif
}
void k_stem_pool_planes(const float* y, const float* scale, const float* shift, float* pooled, uint8_t* idx, unsigned short* pooledp,
                        int groups, int imgs_per_group, int H, int W, int C, hipStream_t s)
{
    const int64_t n = (int64_t)imgs_per_group * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(stem_pool_planes_kernel, dim3(pe_cdiv(n, 256), groups), dim3(256), 0, s, y, scale, shift, pooled, idx, pooledp,
                       imgs_per_group, H, W, C);
}

// dy = ca*dyh + cb*y + cc, dyh = dz masked by z > 0 (or by y*msc+msh > 0); optionally store dyh (elementwise.hip bn_bwd_apply_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                                  const float* __restrict__ y, const float* __restrict__ ca,
                                                                  const float* __restrict__ cb_, const float* __restrict__ cc, float* dy,
                                                                  unsigned short* __restrict__ dyp, float* dyh_out, int pix_per_group,
                                                                  int C, const float* __restrict__ msc, const float* __restrict__ msh,
                                                                  const unsigned short* __restrict__ zh)
{
#if __HIP_DEVICE_COMPILE__
    const int grp = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int cb = (int)(t / pix_per_group);
    if (cb >= (C >> 5)) return;
    const int64_t pix = (int64_t)grp * pix_per_group + (t - (int64_t)cb * pix_per_group);
    const int64_t P = (int64_t)gridDim.y * pix_per_group;
    const int c0 = cb * 32 + 4 * g;
    const size_t o = (size_t)pix * C + c0;
    f32x4 r[2];
    // planes mode: the ReLU mask of a tensor kept only as planes is the sign of its h plane (this thread's chunk: 8 bf16)
    sp_u32x4 hb = {0u, 0u, 0u, 0u};
    if (zh) hb = *reinterpret_cast<const sp_u32x4*>(reinterpret_cast<const unsigned char*>(zh) + (((size_t)cb * 3) * P + pix) * 64 + g * 16);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + 16 * h;
        f32x4 d = ld4(dz + o + 16 * h);
        if (zh) {
            d[0] = (short)(hb[2 * h] & 0xffffu) > 0 ? d[0] : 0.f;
            d[1] = (short)(hb[2 * h] >> 16) > 0 ? d[1] : 0.f;
            d[2] = (short)(hb[2 * h + 1] & 0xffffu) > 0 ? d[2] : 0.f;
            d[3] = (short)(hb[2 * h + 1] >> 16) > 0 ? d[3] : 0.f;
        }
        if (z) {
            const f32x4 zz = ld4(z + o + 16 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = zz[k] > 0.f ? d[k] : 0.f;
        }
        const f32x4 yy = ld4(y + o + 16 * h);
        if (msc) {
            const f32x4 sc = ld4(msc + grp * C + c), sh = ld4(msh + grp * C + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = __builtin_fmaf(yy[k], sc[k], sh[k]) > 0.f ? d[k] : 0.f;
        }
        r[h] = ld4(ca + grp * C + c) * d + ld4(cb_ + grp * C + c) * yy + ld4(cc + grp * C + c);
        if (dyh_out) st4(dyh_out + o + 16 * h, d);
        if (dy) st4(dy + o + 16 * h, r[h]);
    }
    pe_store_planes(dyp, P, cb, pix, g, r[0], r[1]);
#endif
}
void k_bn_bwd_apply_planes(const float* dz, const float* z, const float* y, const float* ca, const float* cb, const float* cc, float* dy,
                           unsigned short* dyp, float* dyh_out, int groups, int pix_per_group, int C, hipStream_t s,
                           const float* mask_scale, const float* mask_shift, const unsigned short* zh)
{
    const int64_t n = (int64_t)pix_per_group * (C / 8);
    hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(pe_cdiv(n, 256), groups), dim3(256), 0, s, dz, z, y, ca, cb, cc, dy, dyp,
                       dyh_out, pix_per_group, C, mask_scale, mask_shift, zh);
}
