// HBM-bound kernels of the FedMLP engine: layout changes, BatchNorm (train/eval,
// forward/backward), stem max-pool, Adam, split-K slab reduction.  gfx950.
//
// Reference ops replaced (all reached through net(images) / loss.backward() /
// optimizer.step() in utils/local_training.py:657-675, 937-966, 1178-1192):
// nn.BatchNorm2d (train: batch statistics + running-stat update, momentum 0.1,
// eps 1e-5; eval: folded affine), F.relu, residual add, nn.MaxPool2d(3,2,1),
// torch.optim.Adam (coupled L2 weight decay).  Roofline: HBM bandwidth; every
// kernel moves 16 B per lane, NHWC so that the channel axis is contiguous.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "split3.h"

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- layout ------
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix_total, int HW)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix_total) return;
    int64_t img = i / HW;
    int64_t r = i - img * HW;
    const float* s = x + img * 3 * HW + r;
    f32x4 v = {s[0], s[HW], s[2 * (int64_t)HW], 0.f};
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
}
// ---- stem im2col for the bf16 configuration --------------------------------------------------------------
// col[img][oh][ow][kh][kw_p = 4][4] (bf16, 16 B per (kh, kw) pair's 4 channel slots ... 32 B per kernel row) straight from the
// caller's NCHW fp32 batch: the K axis has exactly the engine's stem-weight layout [cout][kh][kw_p][4], so the stem
// becomes a K = k*16 pointwise convolution on the bf16 matrix pipe (forward of teacher and student and the weight
// gradient all read this ONE buffer; the fp32 implicit-GEMM stem cost 5 of the step's 68 ms).  Thread = (pixel, kh).
__global__ void stem_im2col_kernel(const float* __restrict__ x, bf16* __restrict__ col, int64_t n, int H, int W, int Ho, int Wo,
                                   int k, int stride, int pad_t, int pad_l)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int kh = (int)(i % k);
    int64_t p = i / k;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int64_t img = p / Ho;
    const int ih = oh * stride + kh - pad_t, iw0 = ow * stride - pad_l;
    const bool rv = (unsigned)ih < (unsigned)H;
    const float* xi = x + (img * 3) * (int64_t)H * W + (int64_t)(rv ? ih : 0) * W;
    float v[4][4];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
        const int iw = iw0 + kw;
        const bool ok = rv && kw < k && (unsigned)iw < (unsigned)W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = xi[(int64_t)c * H * W + (ok ? iw : 0)];
            v[kw][c] = ok ? t : 0.f;
        }
        v[kw][3] = 0.f;
    }
    bf16* o = col + i * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 a = {v[2 * h][0], v[2 * h][1], v[2 * h][2], v[2 * h][3]};
        const f32x4 b = {v[2 * h + 1][0], v[2 * h + 1][1], v[2 * h + 1][2], v[2 * h + 1][3]};
        const bf16x4 pa = __builtin_convertvector(a, bf16x4), pb = __builtin_convertvector(b, bf16x4);
        *reinterpret_cast<bf16x8*>(o + 8 * h) = __builtin_shufflevector(pa, pb, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
void k_stem_im2col(const float* x, void* col, int imgs, int H, int W, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                   hipStream_t s)
{
    const int64_t n = (int64_t)imgs * Ho * Wo * k;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, reinterpret_cast<bf16*>(col), n, H, W, Ho, Wo, k,
                       stride, pad_t, pad_l);
}

void k_nchw_to_nhwc4(const float* x, float* y, int imgs, int H, int W, hipStream_t s)
{
    int64_t n = (int64_t)imgs * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, n, H * W);
}

// engine row o starts at o*Ostride (Ostride >= H*Wpad*Ipad; the tail of a row, if any, is padding the caller keeps zero)
__global__ void oihw_to_ohwi_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int H,
                                    int W, int Wpad, int Ipad, int Ostride, int inverse)
{
    int64_t n = (int64_t)O * H * Wpad * Ipad;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int ip = i % Ipad;
    int64_t t = i / Ipad;
    int wp = t % Wpad; t /= Wpad;
    int h = t % H;
    int o = t / H;
    const bool real = wp < W && ip < I;
    const int64_t j = (((int64_t)o * I + ip) * H + h) * W + wp;     // OIHW index
    const int64_t d = (int64_t)o * Ostride + ((int64_t)h * Wpad + wp) * Ipad + ip;
    if (!inverse) dst[d] = real ? src[j] : 0.f;
    else if (real) dst[j] = src[d];
}
void k_oihw_to_ohwi(const float* src, float* dst, int O, int I, int H, int W, int Wpad, int Ipad, hipStream_t s, int Ostride)
{
    int64_t n = (int64_t)O * H * Wpad * Ipad;
    hipLaunchKernelGGL(oihw_to_ohwi_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, O, I, H, W, Wpad, Ipad,
                       Ostride ? Ostride : H * Wpad * Ipad, 0);
}
void k_ohwi_to_oihw(const float* src, float* dst, int O, int I, int H, int W, int Wpad, int Ipad, hipStream_t s, int Ostride)
{
    int64_t n = (int64_t)O * H * Wpad * Ipad;
    hipLaunchKernelGGL(oihw_to_ohwi_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, O, I, H, W, Wpad, Ipad,
                       Ostride ? Ostride : H * Wpad * Ipad, 1);
}

// input of the packed 7x7 stem: y [imgs][H + 2*fr_t...] -- a zero-framed NHWC3 image (frame written once at allocation,
// only the interior here).  src_nhwc3 = 0: x is the caller's NCHW fp32 batch; 1: x is [imgs][H][W][3] (test hook).
// yp (planes mode, stem_rows.hip): the same framed image as three bf16 planes x = h + m + l of FOUR channels per pixel (the fourth
// zero), [3][imgs][Hp][Wp][4]: 8 B per pixel and plane
__global__ void frame_nhwc3_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int H, int W, int Hp, int Wp,
                                   int top, int left, int src_nhwc3, unsigned short* __restrict__ yp, long long yp_plane_elems)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (img, h, w)
    if (i >= n) return;
    const int w = (int)(i % W);
    const int64_t t = i / W;
    const int h = (int)(t % H);
    const int64_t img = t / H;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        v[c] = src_nhwc3 ? x[i * 3 + c] : x[((img * 3 + c) * H + h) * (int64_t)W + w];
    float* o = y + ((img * Hp + h + top) * (int64_t)Wp + w + left) * 3;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
#if __HIP_DEVICE_COMPILE__
    if (yp) {
        sp_u32x4 Hh, Mm, Ll;
        split3(f32x4{v[0], v[1], v[2], 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, Hh, Mm, Ll);
        unsigned short* q = yp + ((img * Hp + h + top) * (int64_t)Wp + w + left) * 4;
        *reinterpret_cast<uint2*>(q) = make_uint2(Hh[0], Hh[1]);
        *reinterpret_cast<uint2*>(q + yp_plane_elems) = make_uint2(Mm[0], Mm[1]);
        *reinterpret_cast<uint2*>(q + 2 * yp_plane_elems) = make_uint2(Ll[0], Ll[1]);
    }
#endif
}
// packed stem: the 8th tap slot of every kernel row multiplies real pixels (the window over-reads one pixel) with a zero
// weight; its weight GRADIENT is not zero by itself and must not reach Adam
__global__ void stem3_mask_grad_kernel(float* __restrict__ g, int O)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (o, kh, 3 floats)
    if (i >= O * 7 * 3) return;
    const int o = i / 21, r = i - o * 21;
    g[o * 176 + (r / 3) * 24 + 21 + r % 3] = 0.f;
}
void k_stem3_mask_grad(float* g, int O, hipStream_t s)
{
    hipLaunchKernelGGL(stem3_mask_grad_kernel, dim3(cdiv(O * 21, 256)), dim3(256), 0, s, g, O);
}
void k_frame_nhwc3(const float* x, float* y, int imgs, int H, int W, int Hp, int Wp, int top, int left, int src_nhwc3,
                   hipStream_t s, unsigned short* yp, long long yp_plane_elems)
{
    const int64_t n = (int64_t)imgs * H * W;
    hipLaunchKernelGGL(frame_nhwc3_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, n, H, W, Hp, Wp, top, left, src_nhwc3, yp,
                       yp_plane_elems);
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int T, int Ci, TapList taps)
{
    // out[ci][j][co] = w[co][taps.t[j]][ci]; thread per output element, co fastest
    int64_t n = (int64_t)Ci * taps.n * Co;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int co = i % Co;
    int64_t t = i / Co;
    int j = t % taps.n;
    int ci = t / taps.n;
    out[i] = w[((int64_t)co * T + taps.t[j]) * Ci + ci];
}
void k_pack_dgrad(const float* w, float* out, int Co, int T, int Ci, TapList taps, hipStream_t s)
{
    int64_t n = (int64_t)Ci * taps.n * Co;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, w, out, Co, T, Ci, taps);
}

// every (conv, parity class) pack of the step in ONE launch; a block finds its job from the jobs' first-block table.
// out[ci][j][co] = w[co][taps[j]][ci] is a transpose per tap: a block moves one 32 (co) x 32 (ci) tile through LDS, so
// both the reads (along ci) and the writes (along co) are 128-B rows (the thread-per-output-element form gathered one
// float per cache line: 218 us per step for ResNet-18's 11 M weights).  Blocks per job: pack_job_blocks().
__global__ __launch_bounds__(256) void pack_dgrad_all_kernel(const float* __restrict__ state, const PackJob* __restrict__ jobs,
                                                             int njobs)
{
    __shared__ float tile[32][33];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
    const PackJob jb = jobs[j];
    const int tco = (jb.Co + 31) >> 5, tci = (jb.Ci + 31) >> 5;
    int b = (int)blockIdx.x - jb.blk0;
    const int tj = b / (tco * tci);
    b -= tj * (tco * tci);
    const int co0 = (b / tci) << 5, ci0 = (b % tci) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* w = state + jb.w_off;
    const int tap = jb.taps[tj];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + ty + 8 * r, ci = ci0 + tx;
        tile[ty + 8 * r][tx] = (co < jb.Co && ci < jb.Ci) ? w[((int64_t)co * jb.T + tap) * jb.Ci + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + 8 * r, co = co0 + tx;
        if (ci < jb.Ci && co < jb.Co) jb.out[((int64_t)ci * jb.ntaps + tj) * jb.Co + co] = tile[tx][ty + 8 * r];
    }
}
int pack_job_blocks(int Co, int Ci, int ntaps) { return ntaps * ((Co + 31) / 32) * ((Ci + 31) / 32); }
void k_pack_dgrad_all(const float* state, const PackJob* jobs, int njobs, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(pack_dgrad_all_kernel, dim3(nblocks), dim3(256), 0, s, state, jobs, njobs);
}

__global__ void scale_kernel(float* __restrict__ x, float w, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] *= w;
}
__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a, float b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * y[i] + b * x[i];
}
void k_axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(axpby_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, y, x, a, b, n);
}
void k_scale(float* x, float w, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, w, n);
}

// FedAvg of K client states that share this GPU (utils/FedAvg.py:7-14).  HBM-bound: (K + 1) x 4 B per element, 16 B per lane.
// The reference rounds every product and every sum separately (numpy / torch CPU fp32) and divides with an IEEE division:
// contraction into FMAs is switched off for these two kernels (hipcc's default is -ffp-contract=fast; HIP's __fmul_rn /
// __fadd_rn are header functions compiled under that default and are contracted all the same, so the operators are written
// out inside the pragma's scope); fp32 division is correctly rounded by default.
// `out` may be one of the inputs (every element is read before it is written): no __restrict__ on it.
__global__ void __launch_bounds__(256) fedavg_fold_kernel(FoldArgs a, int K, float tot, float* out, int64_t n4)
{
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(a.s[0])[i];
        float4 acc = {v.x * a.n[0], v.y * a.n[0], v.z * a.n[0], v.w * a.n[0]};
        for (int k = 1; k < K; ++k) {
            v = reinterpret_cast<const float4*>(a.s[k])[i];
            const float px = v.x * a.n[k], py = v.y * a.n[k], pz = v.z * a.n[k], pw = v.w * a.n[k];
            acc.x = acc.x + px; acc.y = acc.y + py; acc.z = acc.z + pz; acc.w = acc.w + pw;
        }
        acc.x = acc.x / tot; acc.y = acc.y / tot; acc.z = acc.z / tot; acc.w = acc.w / tot;
        reinterpret_cast<float4*>(out)[i] = acc;
    }
}
__global__ void fedavg_fold_tail_kernel(FoldArgs a, int K, float tot, float* out, int64_t i0, int64_t n)
{
#pragma clang fp contract(off)
    const int64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    float acc = a.s[0][i] * a.n[0];
    for (int k = 1; k < K; ++k) {
        const float p = a.s[k][i] * a.n[k];
        acc = acc + p;
    }
    out[i] = acc / tot;
}
void k_fedavg_fold(const FoldArgs& a, int K, float tot, float* out, int64_t n, hipStream_t s)
{
    // the 16-B path needs 16-B aligned pointers (fm_state_device()'s buffer and torch allocations are; a view at an odd element
    // offset is not): otherwise every element takes the scalar kernel
    bool al = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int k = 0; k < K; ++k) al = al && (reinterpret_cast<uintptr_t>(a.s[k]) & 15) == 0;
    if (!al) {
        for (int64_t i0 = 0; i0 < n; i0 += 1024)
            hipLaunchKernelGGL(fedavg_fold_tail_kernel, dim3(1), dim3(1024), 0, s, a, K, tot, out, i0, std::min<int64_t>(n, i0 + 1024));
        return;
    }
    const int64_t n4 = n / 4;
    if (n4) hipLaunchKernelGGL(fedavg_fold_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 256), 8192)), dim3(256), 0, s, a, K, tot, out, n4);
    if (n4 * 4 < n) hipLaunchKernelGGL(fedavg_fold_tail_kernel, dim3(1), dim3(64), 0, s, a, K, tot, out, n4 * 4, n);
}

// ------------------------------------------------------------ augmentation -----
// Replaces, for a uint8 cache of already-resized images kept in HBM, the per-sample CPU work of
// the reference's train transform (dataset/dataset.py:40-53): RandomAffine(10 deg, 2 %) with
// NEAREST sampling and fill 0, RandomHorizontalFlip, ToTensor, Normalize.  The random draws stay
// on the host; what crosses the ABI per sample is what Pillow's nearest-neighbour affine walks
// (libImaging/Geometry.c affine_fixed): the six coefficients in 16.16 fixed point
//   c2 = FIX(a2 + a0/2 + a1/2), c5 = FIX(a5 + a3/2 + a4/2), c0, c1, c3, c4 = FIX(a0, a1, a3, a4)
// and the source pixel of output (x, y) is ((c2 + c0 x + c1 y) >> 16, (c5 + c3 x + c4 y) >> 16):
// integer arithmetic, bit-exact with Pillow.  The flip is applied after the affine (output x reads
// the affine image at W-1-x).  One thread per output pixel, the three channel planes share the source
// coordinate.  HBM-bound: reads <= 3 B, writes 12 B per pixel.
__global__ void augment_kernel(const uint8_t* __restrict__ cache, const int* __restrict__ idx,
                               const int* __restrict__ params, float* __restrict__ out, int H, int W, float m0,
                               float m1, float m2, float s0, float s1, float s2)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const int* p = params + b * 8;
    const int xs = p[6] != 0 ? W - 1 - x : x;
    const int xin = (p[2] + p[0] * xs + p[1] * y) >> 16;        // arithmetic shift = floor, as in C on int
    const int yin = (p[5] + p[3] * xs + p[4] * y) >> 16;
    const bool ok = xin >= 0 && xin < W && yin >= 0 && yin < H;
    const uint8_t* src = cache + (size_t)idx[b] * 3 * H * W + (ok ? yin * W + xin : 0);
    const float v0 = ok ? (float)src[0] : 0.f, v1 = ok ? (float)src[(size_t)H * W] : 0.f,
                v2 = ok ? (float)src[(size_t)2 * H * W] : 0.f;
    float* o = out + (size_t)b * 3 * H * W + i;
    // ToTensor: float32 / 255 ; Normalize: (v - mean) / std -- IEEE division, no reciprocal, no contraction
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn(v0, 255.f), m0), s0);
    o[(size_t)H * W] = __fdiv_rn(__fsub_rn(__fdiv_rn(v1, 255.f), m1), s1);
    o[(size_t)2 * H * W] = __fdiv_rn(__fsub_rn(__fdiv_rn(v2, 255.f), m2), s2);
}
void k_augment(const uint8_t* cache, const int* idx, const int* params, float* out, int B, int H, int W,
               float m0, float m1, float m2, float s0, float s1, float s2, hipStream_t s)
{
    hipLaunchKernelGGL(augment_kernel, dim3(cdiv((int64_t)H * W, 256), B), dim3(256), 0, s, cache, idx, params, out,
                       H, W, m0, m1, m2, s0, s1, s2);
}

// ------------------------------------------------------------ BN forward -------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int groups, int tiles, int C, int count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ run_mean, float* __restrict__ run_var,
                                   float* __restrict__ mean, float* __restrict__ istd,
                                   float* __restrict__ scale, float* __restrict__ shift, float eps, float momentum,
                                   const int* __restrict__ skip)
{
    __shared__ double red[2][4][64];
    // (a kernel of this step reported a lost part, adam_kernel: the running statistics stay as they are too)
    const bool keep_running = skip && *skip;
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const bool chv = blockIdx.x * 64 + cl < C;             // C need not be a multiple of 64
    const int ch = chv ? blockIdx.x * 64 + cl : C - 1;
    for (int g = 0; g < groups; ++g) {
        double s1 = 0.0, s2 = 0.0;
        const float* st = stats + (size_t)g * tiles * 2 * C;
#pragma unroll 4
        for (int t = sl; t < tiles; t += 4) {                  // (eight loads in flight: the loop was one L2 round trip per tile)
            s1 += (double)st[(size_t)t * 2 * C + ch];
            s2 += (double)st[(size_t)t * 2 * C + C + ch];
        }
        red[0][sl][cl] = s1;
        red[1][sl][cl] = s2;
        __syncthreads();
        if (sl == 0 && chv) {
            s1 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            s2 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
            const double n = (double)count;
            const double mu = s1 / n;
            double var = s2 / n - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            const float ga = gamma[ch], be = beta[ch];
            mean[g * C + ch] = (float)mu;
            istd[g * C + ch] = is;
            const float sc = ga * is;
            scale[g * C + ch] = sc;
            shift[g * C + ch] = be - (float)mu * sc;
            const double unb = count > 1 ? var * n / (n - 1.0) : var;
            if (!keep_running) {
                run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * (float)mu;
                run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (float)unb;
            }
        }
        __syncthreads();
    }
}
// stage A of the statistics reduction: [groups][tiles][2C] -> [groups][32][2C], one block per
// (chunk of tiles, group); fixed summation order, so the result is run-to-run deterministic.
__global__ void bn_fold_tiles_kernel(const float* __restrict__ stats, float* __restrict__ out, int tiles, int C2)
{
    const int g = blockIdx.y, nch = gridDim.x;
    const int per = (tiles + nch - 1) / nch;
    const int t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    const float* st = stats + (size_t)g * tiles * C2;
    for (int c = threadIdx.x; c < C2; c += blockDim.x) {
        // (double sums, round 6: ~100 fp32 terms per chunk were the one fp32 accumulation in the statistics path)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int t = t0;
        for (; t + 3 < t1; t += 4) {
            a0 += (double)st[(size_t)t * C2 + c];
            a1 += (double)st[(size_t)(t + 1) * C2 + c];
            a2 += (double)st[(size_t)(t + 2) * C2 + c];
            a3 += (double)st[(size_t)(t + 3) * C2 + c];
        }
        for (; t < t1; ++t) a0 += (double)st[(size_t)t * C2 + c];
        out[((size_t)g * nch + blockIdx.x) * C2 + c] = (float)((a0 + a1) + (a2 + a3));
    }
}

void k_bn_finalize(const float* stats, int groups, int tiles, int C, int count, const float* gamma,
                   const float* beta, float* run_mean, float* run_var, float* mean, float* istd, float* scale,
                   float* shift, float eps, float momentum, hipStream_t s, const int* skip)
{
    const float* src = stats;
    if (tiles > 64) {
        // the folded partials live right behind the per-tile partials in the same workspace
        float* folded = const_cast<float*>(stats) + (size_t)groups * tiles * 2 * C;
        hipLaunchKernelGGL(bn_fold_tiles_kernel, dim3(32, groups), dim3(256), 0, s, stats, folded, tiles, 2 * C);
        src = folded;
        tiles = 32;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, s, src, groups, tiles, C, count, gamma,
                       beta, run_mean, run_var, mean, istd, scale, shift, eps, momentum, skip);
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float* __restrict__ scale, float* __restrict__ shift, int n, float eps)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sc = gamma[i] / sqrtf(rv[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - rm[i] * sc;
}
void k_bn_eval_affine(const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                      float* scale, float* shift, int n, float eps, hipStream_t s)
{
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, gamma, beta, run_mean, run_var,
                       scale, shift, n, eps);
}

__global__ void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ res,
                                const float* __restrict__ y2, const float* __restrict__ scale2,
                                const float* __restrict__ shift2, float* __restrict__ out, int pix_per_group, int C,
                                int relu)
{
    const int g = blockIdx.y;
    const int Q = C >> 2;
    const int64_t n4 = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y + base);
    const f32x4* r4 = res ? reinterpret_cast<const f32x4*>(res + base) : nullptr;
    const f32x4* z4 = y2 ? reinterpret_cast<const f32x4*>(y2 + base) : nullptr;
    f32x4* o4 = reinterpret_cast<f32x4*>(out + base);
    const f32x4* sc = reinterpret_cast<const f32x4*>(scale + g * C);
    const f32x4* sh = reinterpret_cast<const f32x4*>(shift + g * C);
    const f32x4* sc2 = y2 ? reinterpret_cast<const f32x4*>(scale2 + g * C) : nullptr;
    const f32x4* sh2 = y2 ? reinterpret_cast<const f32x4*>(shift2 + g * C) : nullptr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int cq = (int)(i % Q);
        f32x4 v = y4[i] * sc[cq] + sh[cq];
        if (r4) v += r4[i];
        if (z4) v += z4[i] * sc2[cq] + sh2[cq];
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        o4[i] = v;
    }
}
void k_bn_apply(const float* y, const float* scale, const float* shift, const float* res, const float* y2,
                const float* scale2, const float* shift2, float* out, int groups, int pix_per_group, int C,
                int relu, hipStream_t s)
{
    int64_t n4 = (int64_t)pix_per_group * (C / 4);
    dim3 grid(cdiv(n4, 256), groups);   // one 16-B element per thread: 6.3 TB/s vs 4.7 for a capped grid-stride loop (tools/ew_bw.hip)
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, s, y, scale, shift, res, y2, scale2, shift2, out,
                       pix_per_group, C, relu);
}

// test hook (fm_debug_stem_masks): the ReLU mask of the dense stem map, with the fused multiply-add stem_pool_kernel uses
__global__ void stem_relu_bits_kernel(const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                      uint8_t* __restrict__ bits, int64_t pix_per_group, int C)
{
    const int g = blockIdx.y, BQ = C >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pix_per_group * BQ) return;
    const int b = (int)(i % BQ);
    const int64_t pix = (int64_t)g * pix_per_group + i / BQ;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        m |= (__builtin_fmaf(y[pix * C + 8 * b + j], scale[g * C + 8 * b + j], shift[g * C + 8 * b + j]) > 0.f ? 1u : 0u) << j;
    bits[pix * BQ + b] = (uint8_t)m;
}
void k_stem_relu_bits(const float* y, const float* scale, const float* shift, uint8_t* bits, int groups, int64_t pix_per_group, int C,
                      hipStream_t s)
{
    hipLaunchKernelGGL(stem_relu_bits_kernel, dim3(cdiv(pix_per_group * (C / 8), 256), groups), dim3(256), 0, s, y, scale, shift, bits,
                       pix_per_group, C);
}

// ------------------------------------------------------------ stem pool --------
__global__ void stem_pool_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                 const float* __restrict__ shift, float* __restrict__ pooled,
                                 uint8_t* __restrict__ idx, int imgs_per_group, int H, int W, int C)
{
    const int g = blockIdx.y;
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2;     // 3x3 s2 p1 on even H,W -> H/2
    const int64_t n = (int64_t)imgs_per_group * Hp * Wp * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int ow = (int)(t % Wp); t /= Wp;
    const int oh = (int)(t % Hp);
    const int img = g * imgs_per_group + (int)(t / Hp);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) {
        sc = *reinterpret_cast<const f32x4*>(scale + g * C + cq * 4);
        sh = *reinterpret_cast<const f32x4*>(shift + g * C + cq * 4);
    }
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int code[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if ((unsigned)iw >= (unsigned)W) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(y + ((size_t)(img * H + ih) * W + iw) * C + cq * 4);
            if (scale) {
                v = v * sc + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] > best[k]) { best[k] = v[k]; code[k] = kh * 3 + kw; }
        }
    }
    const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + cq * 4;
    *reinterpret_cast<f32x4*>(pooled + o) = best;
    if (idx) {
        uchar4 c4 = make_uchar4((unsigned char)code[0], (unsigned char)code[1], (unsigned char)code[2],
                                (unsigned char)code[3]);
        *reinterpret_cast<uchar4*>(idx + o) = c4;
    }
}
void k_stem_pool(const float* y, const float* scale, const float* shift, float* pooled, uint8_t* idx, int groups,
                 int imgs_per_group, int H, int W, int C, hipStream_t s)
{
    int64_t n = (int64_t)imgs_per_group * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(stem_pool_kernel, dim3(cdiv(n, 256), groups), dim3(256), 0, s, y, scale, shift, pooled, idx,
                       imgs_per_group, H, W, C);
}

__global__ void stem_pool_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ pooled,
                                     const uint8_t* __restrict__ idx, float* __restrict__ dy, int imgs, int H, int W,
                                     int C)
{
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2;
    const int64_t n = (int64_t)imgs * H * W * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int img = (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1;      // equal when ih is even
    const int ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        if (oh >= Hp) continue;
        const int kh = ih - (2 * oh - 1);
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            if (ow >= Wp) continue;
            const int kw = iw - (2 * ow - 1);
            const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + cq * 4;
            const uchar4 c4 = *reinterpret_cast<const uchar4*>(idx + o);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pooled + o);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dp + o);
            const int code = kh * 3 + kw;
            if (c4.x == code && pv[0] > 0.f) acc[0] += g[0];
            if (c4.y == code && pv[1] > 0.f) acc[1] += g[1];
            if (c4.z == code && pv[2] > 0.f) acc[2] += g[2];
            if (c4.w == code && pv[3] > 0.f) acc[3] += g[3];
        }
    }
    *reinterpret_cast<f32x4*>(dy + i * 4) = acc;
}
void k_stem_pool_bwd(const float* dpooled, const float* pooled, const uint8_t* idx, float* dy, int imgs, int H,
                     int W, int C, hipStream_t s)
{
    int64_t n = (int64_t)imgs * H * W * (C / 4);
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, dpooled, pooled, idx, dy, imgs, H,
                       W, C);
}

// ------------------------------------------------------------ BN backward ------
int bn_bwd_blocks(int pix_per_group)
{
    // blocks per group of the channel reductions (<= 1024: ws_part is sized for that)
    static const int cap = std::min(1024, std::max(1, fm_tune("FM_BN_BLOCKS", 1024)));
    return max(1, min(cap, cdiv(pix_per_group, 64)));
}

// ---- stem: max-pool backward + BatchNorm backward without the dense intermediate ------------------------------------
// The unfused order writes the dense 112x112 gradient of relu(bn(y0)) (stem_pool_bwd), reduces it against y0 and re-reads
// both in the apply pass: 5.4 GB per bs-128 stage-1 step.  Here (a) the two BatchNorm-backward sums are taken over the
// POOLED positions -- every window's gradient goes to its argmax, whose y0 is gathered -- and (b) the apply pass forms the
// pooled-gradient sum of a dense position on the fly (the pool-backward loop) and applies ca*g + cb*y + cc in the same
// thread: 3.4 GB.  Sums: s1 = sum g, s2 = sum g * xhat(y0 at the argmax), g = dp where pooled > 0 (the ReLU mask).
__global__ void stem_pool_bn_reduce_kernel(const float* __restrict__ dp, const float* __restrict__ pooled,
                                           const uint8_t* __restrict__ idx, const float* __restrict__ y,
                                           const float* __restrict__ mean, const float* __restrict__ istd,
                                           float* __restrict__ part, int imgs_per_group, int H, int W, int C,
                                           const float* __restrict__ gamma, const float* __restrict__ beta)
{
    __shared__ f32x4 red[2][256];
    const int g = blockIdx.y, nblk = gridDim.x;
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2, P = 256 / Q;
    const int cq = threadIdx.x % Q, pl = threadIdx.x / Q;
    const int npool = imgs_per_group * Hp * Wp;
    const int TP = 8 * P;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + cq * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(istd + g * C + cq * 4);
    // round 6: the normalised value at the argmax comes from the POOLED value itself -- a pooled value > 0 is
    // fma(y0, gamma istd, beta - mean gamma istd) of its argmax, so xhat = (pooled - beta) / gamma -- instead of a 4-byte gather of
    // y0 out of the dense map per element (the gathers touched every line of the 112 x 112 map: the pass read 0.4 GB more than
    // its inputs).  A channel whose |gamma| is small keeps the gather (the division would magnify pooled's rounding).
    f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = ga;
    bool fast[4] = {false, false, false, false};
    if (gamma) {
        ga = *reinterpret_cast<const f32x4*>(gamma + cq * 4);
        be = *reinterpret_cast<const f32x4*>(beta + cq * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) fast[k] = fabsf(ga[k]) >= 0.0625f;
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = blockIdx.x * TP; t0 < npool; t0 += nblk * TP)
        for (int pp = t0 + pl; pp < min(npool, t0 + TP); pp += P) {
            const int ow = pp % Wp;
            const int t = pp / Wp;
            const int oh = t % Hp;
            const int img = g * imgs_per_group + t / Hp;
            const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + cq * 4;
            const f32x4 d = *reinterpret_cast<const f32x4*>(dp + o);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pooled + o);
            const uchar4 c4 = *reinterpret_cast<const uchar4*>(idx + o);
            const unsigned char code[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(pv[k] > 0.f)) continue;
                float xh;
                if (fast[k]) xh = (pv[k] - be[k]) / ga[k];
                else {
                    const int ih = 2 * oh - 1 + code[k] / 3, iw = 2 * ow - 1 + code[k] % 3;
                    xh = (y[((size_t)(img * H + ih) * W + iw) * C + cq * 4 + k] - mu[k]) * is[k];
                }
                s1[k] += d[k];
                s2[k] += d[k] * xh;
            }
        }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < P; ++k) {
            s1 += red[0][k * Q + cq];
            s2 += red[1][k * Q + cq];
        }
        float* o = part + ((size_t)(g * nblk + blockIdx.x) * 2) * C + cq * 4;
        *reinterpret_cast<f32x4*>(o) = s1;
        *reinterpret_cast<f32x4*>(o + C) = s2;
    }
}
// round 6, even H and W: one thread = the 2 x 2 dense positions (2 oh + a, 2 ow + b) of one pooled index and channel quad.  They
// are covered by the four windows (oh + i, ow + j) only -- window (i, j) reaches position (a, b) when i <= a and j <= b, with the
// window-local code (a - 2 i + 1) * 3 + (b - 2 j + 1) -- so a thread reads four windows' (argmax code, pooled value, gradient) for
// four outputs where the per-position kernel below read 2.25 windows per output, and does its index arithmetic once for four.
__global__ __launch_bounds__(256) void stem_pool_bn_apply2_kernel(const float* __restrict__ dp, const float* __restrict__ pooled,
                                                                  const uint8_t* __restrict__ idx, const float* __restrict__ y,
                                                                  const float* __restrict__ ca, const float* __restrict__ cb,
                                                                  const float* __restrict__ cc, float* __restrict__ dy,
                                                                  int imgs_per_group, int H, int W, int C)
{
    const int g = blockIdx.y;
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2;
    const int64_t n = (int64_t)imgs_per_group * Hp * Wp * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int ow = (int)(t % Wp); t /= Wp;
    const int oh = (int)(t % Hp);
    const int img = g * imgs_per_group + (int)(t / Hp);
    // gw[i][j][k] = the gradient window (i, j) sends, cw = the window-local code of its argmax (15: nothing to send)
    f32x4 gw[2][2];
    int cw[2][2][4];
#pragma unroll
    for (int wi = 0; wi < 2; ++wi)
#pragma unroll
        for (int wj = 0; wj < 2; ++wj) {
            const bool ok = oh + wi < Hp && ow + wj < Wp;
            const size_t o = ((size_t)(img * Hp + (ok ? oh + wi : oh)) * Wp + (ok ? ow + wj : ow)) * C + cq * 4;
            const uchar4 c4 = *reinterpret_cast<const uchar4*>(idx + o);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pooled + o);
            gw[wi][wj] = *reinterpret_cast<const f32x4*>(dp + o);
            const int code[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) cw[wi][wj][k] = (ok && pv[k] > 0.f) ? code[k] : 15;
        }
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ca + g * C + cq * 4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(cb + g * C + cq * 4);
    const f32x4 c4v = *reinterpret_cast<const f32x4*>(cc + g * C + cq * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // the same order of additions as the per-position kernel: windows by rising (oh, ow)
#pragma unroll
            for (int wi = 0; wi <= a; ++wi)
#pragma unroll
                for (int wj = 0; wj <= b; ++wj) {
                    const int code = (a - 2 * wi + 1) * 3 + (b - 2 * wj + 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cw[wi][wj][k] == code) acc[k] += gw[wi][wj][k];
                }
            const size_t od = ((size_t)(img * H + 2 * oh + a) * W + 2 * ow + b) * C + cq * 4;
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + od);
            *reinterpret_cast<f32x4*>(dy + od) = a4 * acc + b4 * yy + c4v;
        }
}
__global__ void stem_pool_bn_apply_kernel(const float* __restrict__ dp, const float* __restrict__ pooled,
                                          const uint8_t* __restrict__ idx, const float* __restrict__ y,
                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                          const float* __restrict__ cc, float* __restrict__ dy, int imgs_per_group, int H,
                                          int W, int C)
{
    const int g = blockIdx.y;
    const int Hp = H / 2, Wp = W / 2, Q = C >> 2;
    const int64_t n = (int64_t)imgs_per_group * H * W * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int img = g * imgs_per_group + (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1;      // equal when ih is even
    const int ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        if (oh >= Hp) continue;
        const int kh = ih - (2 * oh - 1);
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            if (ow >= Wp) continue;
            const int kw = iw - (2 * ow - 1);
            const size_t o = ((size_t)(img * Hp + oh) * Wp + ow) * C + cq * 4;
            const uchar4 c4 = *reinterpret_cast<const uchar4*>(idx + o);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pooled + o);
            const f32x4 gg = *reinterpret_cast<const f32x4*>(dp + o);
            const int code = kh * 3 + kw;
            if (c4.x == code && pv[0] > 0.f) acc[0] += gg[0];
            if (c4.y == code && pv[1] > 0.f) acc[1] += gg[1];
            if (c4.z == code && pv[2] > 0.f) acc[2] += gg[2];
            if (c4.w == code && pv[3] > 0.f) acc[3] += gg[3];
        }
    }
    const size_t od = ((size_t)(img * H + ih) * W + iw) * C + cq * 4;
    const f32x4 yy = *reinterpret_cast<const f32x4*>(y + od);
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ca + g * C + cq * 4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(cb + g * C + cq * 4);
    const f32x4 c4v = *reinterpret_cast<const f32x4*>(cc + g * C + cq * 4);
    *reinterpret_cast<f32x4*>(dy + od) = a4 * acc + b4 * yy + c4v;
}
int stem_pool_bn_blocks(int pooled_per_group) { return bn_bwd_blocks(pooled_per_group); }
void k_stem_pool_bn_reduce(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y, const float* mean,
                           const float* istd, float* part, int groups, int imgs_per_group, int H, int W, int C, hipStream_t s,
                           const float* gamma, const float* beta)
{
    dim3 grid(stem_pool_bn_blocks(imgs_per_group * (H / 2) * (W / 2)), groups);
    hipLaunchKernelGGL(stem_pool_bn_reduce_kernel, grid, dim3(256), 0, s, dpooled, pooled, idx, y, mean, istd, part,
                       imgs_per_group, H, W, C, gamma, beta);
}
void k_stem_pool_bn_apply(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y, const float* ca,
                          const float* cb, const float* cc, float* dy, int groups, int imgs_per_group, int H, int W, int C,
                          hipStream_t s)
{
    static const int blocks2 = fm_tune("FM_STEM_APPLY_2X2", 1);
    if (blocks2 && H % 2 == 0 && W % 2 == 0) {
        const int64_t n2 = (int64_t)imgs_per_group * (H / 2) * (W / 2) * (C / 4);
        hipLaunchKernelGGL(stem_pool_bn_apply2_kernel, dim3(cdiv(n2, 256), groups), dim3(256), 0, s, dpooled, pooled, idx, y, ca, cb,
                           cc, dy, imgs_per_group, H, W, C);
        return;
    }
    const int64_t n = (int64_t)imgs_per_group * H * W * (C / 4);
    hipLaunchKernelGGL(stem_pool_bn_apply_kernel, dim3(cdiv(n, 256), groups), dim3(256), 0, s, dpooled, pooled, idx, y, ca, cb,
                       cc, dy, imgs_per_group, H, W, C);
}


// ReLU mask of dz: from z (the stored post-activation, z > 0) or -- msc/msh given, z null -- recomputed from the conv
// output the kernel reads anyway: relu(bn(y)) > 0  <=>  y*scale + shift > 0 with the forward's own per-group scale /
// shift and the same fused multiply-add bn_apply_kernel uses (bit-identical mask, one tensor read less).
__device__ __forceinline__ f32x4 relu_mask_from_y(f32x4 d, f32x4 yy, f32x4 sc, f32x4 sh)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = __builtin_fmaf(yy[k], sc[k], sh[k]) > 0.f ? d[k] : 0.f;
    return d;
}
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                     const float* __restrict__ y, const float* __restrict__ mean,
                                     const float* __restrict__ istd, float* __restrict__ part, int pix_per_group,
                                     int C, const float* __restrict__ msc, const float* __restrict__ msh,
                                     const unsigned short* __restrict__ zh, long long zP)
{
    __shared__ f32x4 red[2][256];
    const int g = blockIdx.y, nblk = gridDim.x;
    const int Q = C >> 2, P = 256 / Q;
    const int cq = threadIdx.x % Q, pl = threadIdx.x / Q;
    // pixel tiles are dealt round-robin to the blocks (tile t -> block t % nblk), so the blocks running at any
    // moment read neighbouring memory: contiguous per-block chunks behave like a large-stride walk (4.7 vs 6.3 TB/s)
    const int TP = 8 * P;
    const size_t base = (size_t)g * pix_per_group * C;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + cq * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(istd + g * C + cq * 4);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (msc) {
        sc = *reinterpret_cast<const f32x4*>(msc + g * C + cq * 4);
        sh = *reinterpret_cast<const f32x4*>(msh + g * C + cq * 4);
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = blockIdx.x * TP; t0 < pix_per_group; t0 += nblk * TP)
#pragma unroll 4
    for (int p = t0 + pl; p < min(pix_per_group, t0 + TP); p += P) {
        const size_t o = base + (size_t)p * C + cq * 4;
        f32x4 d = *reinterpret_cast<const f32x4*>(dz + o);
        if (z) {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = zz[k] > 0.f ? d[k] : 0.f;
        }
        if (zh) {
            // planes mode: z lives only as block-major planes; its sign is the sign of the h plane (z > 0 <=> bf16(z) > 0 for every
            // normal float).  Channels 4 cq .. 4 cq + 3: chunk (c & 15) >> 2 of block c >> 5, second half of the chunk when c & 16
            const int c = cq * 4;
            const uint2 hb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(zh) +
                ((size_t)(c >> 5) * 3 * zP + (size_t)g * pix_per_group + p) * 64 + ((c & 15) >> 2) * 16 + ((c >> 4) & 1) * 8);
            d[0] = (short)(hb.x & 0xffffu) > 0 ? d[0] : 0.f;
            d[1] = (short)(hb.x >> 16) > 0 ? d[1] : 0.f;
            d[2] = (short)(hb.y & 0xffffu) > 0 ? d[2] : 0.f;
            d[3] = (short)(hb.y >> 16) > 0 ? d[3] : 0.f;
        }
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
        if (msc) d = relu_mask_from_y(d, yy, sc, sh);
        const f32x4 xh = (yy - mu) * is;
        s1 += d;
        s2 += d * xh;
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < P; ++k) {
            s1 += red[0][k * Q + cq];
            s2 += red[1][k * Q + cq];
        }
        float* o = part + ((size_t)(g * nblk + blockIdx.x) * 2) * C + cq * 4;
        *reinterpret_cast<f32x4*>(o) = s1;
        *reinterpret_cast<f32x4*>(o + C) = s2;
    }
}
void k_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean, const float* istd,
                     float* part, int groups, int pix_per_group, int C, hipStream_t s, const float* mask_scale,
                     const float* mask_shift, const unsigned short* zh)
{
    dim3 grid(bn_bwd_blocks(pix_per_group), groups);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, s, dz, z, y, mean, istd, part, pix_per_group, C,
                       mask_scale, mask_shift, zh, (long long)groups * pix_per_group);
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int groups, int nblk, int C, int count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ istd, float* __restrict__ ca,
                                       float* __restrict__ cb, float* __restrict__ cc, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta)
{
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const bool chv = blockIdx.x * 64 + cl < C;
    const int ch = chv ? blockIdx.x * 64 + cl : C - 1;
    double dg = 0.0, db = 0.0;
    for (int g = 0; g < groups; ++g) {
        double s1 = 0.0, s2 = 0.0;
        const float* pt = part + (size_t)g * nblk * 2 * C;
#pragma unroll 4
        for (int t = sl; t < nblk; t += 4) {
            s1 += (double)pt[(size_t)t * 2 * C + ch];
            s2 += (double)pt[(size_t)t * 2 * C + C + ch];
        }
        red[0][sl][cl] = s1;
        red[1][sl][cl] = s2;
        __syncthreads();
        if (sl == 0 && chv) {
            s1 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            s2 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
            const float is = istd[g * C + ch], mu = mean[g * C + ch];
            const float a = gamma[ch] * is;
            const float b = -a * is * (float)(s2 / (double)count);
            ca[g * C + ch] = a;
            cb[g * C + ch] = b;
            cc[g * C + ch] = -b * mu - a * (float)(s1 / (double)count);
            dg += s2;
            db += s1;
        }
        __syncthreads();
    }
    if (sl == 0 && chv) {
        dgamma[ch] = (float)dg;
        dbeta[ch] = (float)db;
    }
}
void k_bn_bwd_finalize(const float* part, int groups, int nblk, int C, int count, const float* gamma,
                       const float* mean, const float* istd, float* ca, float* cb, float* cc, float* dgamma,
                       float* dbeta, hipStream_t s)
{
    const float* src = part;
    if (nblk > 64) {      // fold the per-block partials with many blocks first (same kernel as the forward statistics)
        float* folded = const_cast<float*>(part) + (size_t)groups * nblk * 2 * C;
        hipLaunchKernelGGL(bn_fold_tiles_kernel, dim3(32, groups), dim3(256), 0, s, part, folded, nblk, 2 * C);
        src = folded;
        nblk = 32;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, s, src, groups, nblk, C, count, gamma,
                       mean, istd, ca, cb, cc, dgamma, dbeta);
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                    const float* __restrict__ y, const float* __restrict__ ca,
                                    const float* __restrict__ cb, const float* __restrict__ cc, float* dy,
                                    float* dyh_out, int pix_per_group, int C, const float* __restrict__ msc,
                                    const float* __restrict__ msh)
{
    const int g = blockIdx.y;
    const int Q = C >> 2;
    const int64_t n4 = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(ca + g * C);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(cb + g * C);
    const f32x4* c4 = reinterpret_cast<const f32x4*>(cc + g * C);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int cq = (int)(i % Q);
        const size_t o = base + (size_t)i * 4;
        f32x4 d = *reinterpret_cast<const f32x4*>(dz + o);
        if (z) {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = zz[k] > 0.f ? d[k] : 0.f;
        }
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
        if (msc)
            d = relu_mask_from_y(d, yy, *reinterpret_cast<const f32x4*>(msc + g * C + cq * 4),
                                 *reinterpret_cast<const f32x4*>(msh + g * C + cq * 4));
        const f32x4 r = a4[cq] * d + b4[cq] * yy + c4[cq];
        if (dyh_out) *reinterpret_cast<f32x4*>(dyh_out + o) = d;
        *reinterpret_cast<f32x4*>(dy + o) = r;
    }
}
void k_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* ca, const float* cb,
                    const float* cc, float* dy, float* dyh_out, int groups, int pix_per_group, int C,
                    hipStream_t s, const float* mask_scale, const float* mask_shift)
{
    int64_t n4 = (int64_t)pix_per_group * (C / 4);
    dim3 grid(cdiv(n4, 256), groups);   // one 16-B element per thread: 6.3 TB/s vs 4.7 for a capped grid-stride loop (tools/ew_bw.hip)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, dz, z, y, ca, cb, cc, dy, dyh_out,
                       pix_per_group, C, mask_scale, mask_shift);
}

// ------------------------------------------------------------ optimiser --------
// torch.optim.Adam single-tensor update order (exp_avg.lerp_, exp_avg_sq mul/addcmul,
// denom = sqrt(v)/sqrt(bc2) + eps, p -= lr/bc1 * m/denom), L2 decay folded into g.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, const int* __restrict__ skip)
{
    // a kernel of this step reported a lost part (pconv.hip's bounded stream-K wait): the gradients are poisoned, the weights
    // and moments stay as they are; the host learns of it through the word's mapped twin at its next call
    if (skip && *skip) return;
    const float step = lr / bc1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] + wd * pp;
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        mm = mm + (1.f - b1) * (gg - mm);
        vv = vv * b2 + (1.f - b2) * gg * gg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float den = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] = pp[k] - step * (mm[k] / den);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}
void k_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
            float wd, float bc1, float bc2_sqrt, hipStream_t s, const int* skip)
{
    const int64_t n4 = n / 4;     // engine pads the parameter arena to a multiple of 4
    hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, s, p, g, m, v, n4, lr, b1, b2,
                       eps, wd, bc1, bc2_sqrt, skip);
}

// out[i] = sum_s slab[s][i]; block = 64 float4 columns x 16 split lanes, lanes combined through
// LDS in a fixed order (deterministic), so a long split axis does not serialise on one thread.
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int64_t n4)
{
    __shared__ f32x4 red[16][64];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = sl;
        for (; s + 16 < splits; s += 32) {
            a += reinterpret_cast<const f32x4*>(slab)[(int64_t)s * n4 + i];
            b += reinterpret_cast<const f32x4*>(slab)[(int64_t)(s + 16) * n4 + i];
        }
        if (s < splits) a += reinterpret_cast<const f32x4*>(slab)[(int64_t)s * n4 + i];
    }
    red[sl][col] = a + b;
    __syncthreads();
    if (sl == 0 && i < n4) {
        f32x4 r = red[0][col];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += red[k][col];
        reinterpret_cast<f32x4*>(out)[i] = r;
    }
}
void k_reduce_slabs(const float* slab, float* out, int splits, int64_t n, hipStream_t s)
{
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n4, 64)), dim3(1024), 0, s, slab, out, splits, n4);
}
