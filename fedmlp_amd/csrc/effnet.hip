// HBM-bound kernels of the EfficientNet-B0 path (BASELINE configs 4-5), gfx950.
//
// Reference ops replaced: what efficientnet-pytorch 0.7.1's MBConvBlock does around its 1x1
// convolutions (the 1x1 convs themselves are GEMMs and run through igemm.hip / wgrad.hip):
// depthwise k3/k5 s1/s2 convolution with TF-"same" padding, BatchNorm(eps 1e-3, momentum 0.01)
// + Swish forward/backward, squeeze-and-excite (global pool -> 1x1 reduce + Swish -> 1x1 expand
// + sigmoid -> channel gate), drop-connect on the residual branch, dropout before `_fc`
// (model/efficientnet.py:28-33 builds the net; the trainer calls it at
// utils/local_training.py:657, 937-947, 983, 1030, 1178).  The model is HBM-bound on MI355X
// (10.4 FLOP/B in fp32): every kernel here moves 16 B per lane on NHWC tensors whose
// channel counts are padded to multiples of 16 (24->32, 40->48; padded channels are exactly 0).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// FAST = bf16 storage: v_exp_f32 / v_rcp_f32 (1 ulp-class fp32 results feeding 8-bit mantissas) instead of the
// IEEE expf / division sequences -- with 8 elements per 16 B the exact forms make these passes VALU-bound
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }
// FM_F32_FAST_SWISH (default on): the fp32 configuration takes the hardware exp / rcp in the streaming kernels too
// (both within ~2 ulp of the IEEE sequences: the step parity against the fp32 oracle is unchanged at its 2e-5 / 5e-4
// bounds, and the BN / squeeze-excite passes stop being issue-bound).  -DFM_F32_FAST_SWISH=0 restores expf and 1/x.
template <bool FAST> __device__ __forceinline__ float sigm_t(float v)
{
    if constexpr (FAST || FM_F32_FAST_SWISH) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    else return 1.f / (1.f + expf(-v));
}
template <bool FAST = false> __device__ __forceinline__ f32x4 act_fwd(f32x4 v, int act)
{
    if (act == 1) { for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f); }
    else if (act == 2) { for (int k = 0; k < 4; ++k) v[k] = v[k] * sigm_t<FAST>(v[k]); }
    return v;
}
// d act(v) / dv for act = swish
template <bool FAST = false> __device__ __forceinline__ float swish_grad(float v)
{
    const float s = sigm_t<FAST>(v);
    return s * (1.f + v * (1.f - s));
}

// ------------------------------------------------------------ BN(+act) apply ---
// Every thread moves ONE 16-B piece per tensor: 4 channels in fp32, 8 in bf16 (NV quads; the mixed fp32-y /
// bf16-activation form of the stem uses 4).  C % (4 NV) == 0 (channel counts are padded to 16).
template <typename TY, typename TA> struct NvOf { static constexpr int NV = (VecOf<TY>::NV == 2 && VecOf<TA>::NV == 2) ? 2 : 1; };

// out = act(y*scale+shift) * rowscale[img] + res      (per group scale/shift)
template <typename TY, typename TA>
__global__ void bnact_apply_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, const TA* __restrict__ res,
                                   const float* __restrict__ rowscale, TA* __restrict__ out, int pix_per_group,
                                   int HW, int C, int act)
{
    constexpr int NV = NvOf<TY, TA>::NV;
    const int g = blockIdx.y;
    const int Q = C / (4 * NV);
    const int64_t nq = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int cq = (int)(i % Q);
    const int64_t pix = i / Q;
    const size_t o = base + (size_t)i * (4 * NV);
    f32x4 v[NV], sc[NV], sh[NV];
    ldv<NV>(y + o, v);
    ldf<NV>(scale + g * C + cq * 4 * NV, sc);
    ldf<NV>(shift + g * C + cq * 4 * NV, sh);
#pragma unroll
    for (int h = 0; h < NV; ++h) v[h] = act_fwd<NV == 2>(v[h] * sc[h] + sh[h], act);
    if (rowscale) {
        const float rs = rowscale[(size_t)g * (pix_per_group / HW) + pix / HW];
#pragma unroll
        for (int h = 0; h < NV; ++h) v[h] = v[h] * rs;
    }
    if (res) {
        f32x4 r[NV];
        ldv<NV>(res + o, r);
#pragma unroll
        for (int h = 0; h < NV; ++h) v[h] += r[h];
    }
    stv<NV>(out + o, v);
}
// The same pass with EW_R pixels per thread.  The one-piece-per-thread form above issues 2 NV 16-B loads of per-channel
// parameters for every 16-B piece of data (4 : 1 in bf16), so the vector L1 moves several times the HBM bytes and the
// kernel streams at ~3.5 TB/s.  Here a block is P pixel rows x Q channel pieces (P*Q <= 256 threads, the rest idle): a
// thread keeps its channel piece, loads scale / shift once and walks EW_R rows P apart -- every load instruction is
// still one contiguous run of P*Q pieces, a block covers one contiguous chunk of EW_R*P pixels (DRAM locality), and
// the per-element arithmetic is untouched (bit-identical results).
constexpr int EW_R = 4;
// FM_EW_ROWS (bf16 storage, default 3) / FM_EW_ROWS_F32 (fp32 storage, default 1): bit 0 = apply pass, bit 1 =
// backward-apply pass.  Measured per step: bf16 58.5 -> 57.1 ms, all of it from the backward-apply (14 parameter loads
// per 2 data loads); fp32 apply 58.35 -> 58.04, fp32 backward-apply 58.35 -> 58.85 (slower: stays one piece per thread).
static inline bool ew_rows_on(int Q, int nv, int pass_bit)
{
    const char* v = getenv(nv == 2 ? "FM_EW_ROWS" : "FM_EW_ROWS_F32");      // read per call (tests compare both forms in one process)
    const int mode = v ? atoi(v) : (nv == 2 ? 3 : 1);
    return (mode & pass_bit) && Q >= 1 && Q <= 256;
}
template <typename TY, typename TA>
__global__ __launch_bounds__(256) void bnact_apply_rows_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const TA* __restrict__ res,
                                                               const float* __restrict__ rowscale, TA* __restrict__ out,
                                                               int pix_per_group, int HW, int C, int act)
{
    constexpr int NV = NvOf<TY, TA>::NV;
    const int g = blockIdx.y;
    const int Q = C / (4 * NV);
    const int T = blockDim.x;                          // (256 / Q) * Q threads: P pixel rows x Q channel pieces
    const int cq = threadIdx.x % Q;
    const int64_t nq = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    f32x4 sc[NV], sh[NV];
    ldf<NV>(scale + g * C + cq * 4 * NV, sc);
    ldf<NV>(shift + g * C + cq * 4 * NV, sh);
    f32x4 v[EW_R][NV];
    int64_t idx[EW_R];
#pragma unroll
    for (int k = 0; k < EW_R; ++k) {                   // all loads first (the tail re-reads the last valid piece)
        idx[k] = ((int64_t)blockIdx.x * EW_R + k) * T + threadIdx.x;
        ldv<NV>(y + base + (size_t)min(idx[k], nq - 1) * (4 * NV), v[k]);
    }
#pragma unroll
    for (int k = 0; k < EW_R; ++k) {
        if (idx[k] >= nq) break;
        const size_t o = base + (size_t)idx[k] * (4 * NV);
#pragma unroll
        for (int h = 0; h < NV; ++h) v[k][h] = act_fwd<NV == 2>(v[k][h] * sc[h] + sh[h], act);
        if (rowscale) {
            const float rs = rowscale[(size_t)g * (pix_per_group / HW) + (idx[k] / Q) / HW];
#pragma unroll
            for (int h = 0; h < NV; ++h) v[k][h] = v[k][h] * rs;
        }
        if (res) {
            f32x4 r[NV];
            ldv<NV>(res + o, r);
#pragma unroll
            for (int h = 0; h < NV; ++h) v[k][h] += r[h];
        }
        stv<NV>(out + o, v[k]);
    }
}
template <typename T> static inline const T* cp(const void* p) { return reinterpret_cast<const T*>(p); }
template <typename T> static inline T* mp(void* p) { return reinterpret_cast<T*>(p); }

// ty / ta: storage type of the raw conv output y and of the activations (res, out): DT_F32 or DT_BF16
// (the stem's y stays fp32 in the bf16 configuration: it comes from the fp32 implicit-GEMM kernel)
void k_bnact_apply(const void* y, int ty, const float* scale, const float* shift, const void* res, const float* rowscale,
                   void* out, int ta, int groups, int pix_per_group, int HW, int C, int act, hipStream_t s)
{
    const int nv = (ty == DT_BF16 && ta == DT_BF16) ? 2 : 1;
    const int Q = C / (4 * nv);
    if (ew_rows_on(Q, nv, 1)) {
        const int T = (256 / Q) * Q;
        const dim3 rgrid(cdiv((int64_t)pix_per_group * Q, (int64_t)EW_R * T), groups);
        if (ty == DT_F32 && ta == DT_F32)
            hipLaunchKernelGGL((bnact_apply_rows_kernel<float, float>), rgrid, dim3(T), 0, s, cp<float>(y), scale, shift,
                               cp<float>(res), rowscale, mp<float>(out), pix_per_group, HW, C, act);
        else if (ty == DT_F32)
            hipLaunchKernelGGL((bnact_apply_rows_kernel<float, bf16>), rgrid, dim3(T), 0, s, cp<float>(y), scale, shift,
                               cp<bf16>(res), rowscale, mp<bf16>(out), pix_per_group, HW, C, act);
        else
            hipLaunchKernelGGL((bnact_apply_rows_kernel<bf16, bf16>), rgrid, dim3(T), 0, s, cp<bf16>(y), scale, shift,
                               cp<bf16>(res), rowscale, mp<bf16>(out), pix_per_group, HW, C, act);
        return;
    }
    const dim3 grid(cdiv((int64_t)pix_per_group * (C / (4 * nv)), 256), groups);
    if (ty == DT_F32 && ta == DT_F32)
        hipLaunchKernelGGL((bnact_apply_kernel<float, float>), grid, dim3(256), 0, s, cp<float>(y), scale, shift, cp<float>(res),
                           rowscale, mp<float>(out), pix_per_group, HW, C, act);
    else if (ty == DT_F32)
        hipLaunchKernelGGL((bnact_apply_kernel<float, bf16>), grid, dim3(256), 0, s, cp<float>(y), scale, shift, cp<bf16>(res),
                           rowscale, mp<bf16>(out), pix_per_group, HW, C, act);
    else
        hipLaunchKernelGGL((bnact_apply_kernel<bf16, bf16>), grid, dim3(256), 0, s, cp<bf16>(y), scale, shift, cp<bf16>(res),
                           rowscale, mp<bf16>(out), pix_per_group, HW, C, act);
}

// ------------------------------------------------------------ channel reductions
// mode 0: (sum y, sum y^2)                        -> forward BN statistics
// mode 1: (sum dyh, sum dyh*xhat), dyh = dz * act'(v) * rowscale   -> BN backward sums
// part layout [groups][nblk][2][C] (what bn_finalize / bn_bwd_finalize consume)
template <typename TY, typename TA>
__global__ void chan_reduce_kernel(const TA* __restrict__ a, const TY* __restrict__ y,
                                   const float* __restrict__ mean, const float* __restrict__ istd,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ rowscale, float* __restrict__ part, int pix_per_group,
                                   int HW, int C, int mode, int act, const float* __restrict__ gate,
                                   const float* __restrict__ dsv)
{
    constexpr int NV = NvOf<TY, TA>::NV;
    __shared__ f32x4 red[2][NV][256];
    const int g = blockIdx.y, nblk = gridDim.x;
    const int Q = C / (4 * NV);                // 16-B pieces per pixel
    const int QT = Q < 256 ? Q : 256;          // pieces handled concurrently
    const int P = 256 / QT;                    // pixel lanes
    const int cq0 = threadIdx.x % QT, pl = threadIdx.x / QT;
    const bool active = pl < P;
    const int TP = 8 * P;                     // pixel tiles dealt round-robin to the blocks (DRAM locality)
    const size_t base = (size_t)g * pix_per_group * C;
    for (int cq = cq0; cq < Q; cq += QT) {
        const int c0 = cq * 4 * NV;
        f32x4 s1[NV], s2[NV];
#pragma unroll
        for (int h = 0; h < NV; ++h) { s1[h] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (active) {
            f32x4 mu[NV], is[NV], sc[NV], sh[NV];
#pragma unroll
            for (int h = 0; h < NV; ++h) { mu[h] = is[h] = sc[h] = sh[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (mode == 1) {
                ldf<NV>(mean + g * C + c0, mu);
                ldf<NV>(istd + g * C + c0, is);
                if (act == 2) {
                    ldf<NV>(scale + g * C + c0, sc);
                    ldf<NV>(shift + g * C + c0, sh);
                }
            }
            for (int t0 = blockIdx.x * TP; t0 < pix_per_group; t0 += nblk * TP)
#pragma unroll 4
            for (int p = t0 + pl; p < min(pix_per_group, t0 + TP); p += P) {
                const size_t o = base + (size_t)p * C + c0;
                f32x4 yy[NV];
                ldv<NV>(y + o, yy);
                if (mode == 0) {
#pragma unroll
                    for (int h = 0; h < NV; ++h) { s1[h] += yy[h]; s2[h] += yy[h] * yy[h]; }
                } else {
                    f32x4 d[NV];
                    ldv<NV>(a + o, d);
                    if (gate) {          // squeeze-excite backward folded in: d(a_d) = d(a_s)*gate + ds/HW
                        const size_t io = ((size_t)g * (pix_per_group / HW) + p / HW) * C + c0;
                        f32x4 gt[NV], dv[NV];
                        ldf<NV>(gate + io, gt);
                        ldf<NV>(dsv + io, dv);
#pragma unroll
                        for (int h = 0; h < NV; ++h) d[h] = d[h] * gt[h] + dv[h] * (1.f / (float)HW);
                    }
                    if (act == 2) {
#pragma unroll
                        for (int h = 0; h < NV; ++h) {
                            const f32x4 v = yy[h] * sc[h] + sh[h];
#pragma unroll
                            for (int k = 0; k < 4; ++k) d[h][k] *= swish_grad<NV == 2>(v[k]);
                        }
                    }
                    if (rowscale) {
                        const float rs = rowscale[(size_t)g * (pix_per_group / HW) + p / HW];
#pragma unroll
                        for (int h = 0; h < NV; ++h) d[h] = d[h] * rs;
                    }
#pragma unroll
                    for (int h = 0; h < NV; ++h) { s1[h] += d[h]; s2[h] += d[h] * ((yy[h] - mu[h]) * is[h]); }
                }
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int h = 0; h < NV; ++h) { red[0][h][pl * QT + cq0] = s1[h]; red[1][h][pl * QT + cq0] = s2[h]; }
        }
        __syncthreads();
        if (pl == 0) {
            float* o = part + ((size_t)(g * nblk + blockIdx.x) * 2) * C + c0;
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                for (int k = 1; k < P; ++k) { s1[h] += red[0][h][k * QT + cq0]; s2[h] += red[1][h][k * QT + cq0]; }
                st4(o + 4 * h, s1[h]);
                st4(o + C + 4 * h, s2[h]);
            }
        }
    }
}
void k_chan_reduce(const void* a, int ta, const void* y, int ty, const float* mean, const float* istd, const float* scale,
                   const float* shift, const float* rowscale, float* part, int groups, int pix_per_group, int HW,
                   int C, int mode, int act, const float* gate, const float* dsv, hipStream_t s)
{
    const dim3 grid(bn_bwd_blocks(pix_per_group), groups);
    if (ty == DT_F32 && ta == DT_F32)
        hipLaunchKernelGGL((chan_reduce_kernel<float, float>), grid, dim3(256), 0, s, cp<float>(a), cp<float>(y), mean, istd,
                           scale, shift, rowscale, part, pix_per_group, HW, C, mode, act, gate, dsv);
    else if (ty == DT_F32)
        hipLaunchKernelGGL((chan_reduce_kernel<float, bf16>), grid, dim3(256), 0, s, cp<bf16>(a), cp<float>(y), mean, istd,
                           scale, shift, rowscale, part, pix_per_group, HW, C, mode, act, gate, dsv);
    else
        hipLaunchKernelGGL((chan_reduce_kernel<bf16, bf16>), grid, dim3(256), 0, s, cp<bf16>(a), cp<bf16>(y), mean, istd,
                           scale, shift, rowscale, part, pix_per_group, HW, C, mode, act, gate, dsv);
}

// dy = ca*dyh + cb*y + cc with dyh = dz * act'(v) * rowscale
template <typename TY, typename TA>
__global__ void bnact_bwd_apply_kernel(const TA* __restrict__ dz, const TY* __restrict__ y,
                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                       const float* __restrict__ cc, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const float* __restrict__ rowscale,
                                       TY* __restrict__ dy, int pix_per_group, int HW, int C, int act,
                                       const float* __restrict__ gate, const float* __restrict__ dsv)
{
    constexpr int NV = NvOf<TY, TA>::NV;
    const int g = blockIdx.y;
    const int Q = C / (4 * NV);
    const int64_t nq = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int c0 = (int)(i % Q) * 4 * NV;
    const int64_t pix = i / Q;
    const size_t o = base + (size_t)i * (4 * NV);
    f32x4 d[NV], yy[NV];
    ldv<NV>(dz + o, d);
    ldv<NV>(y + o, yy);
    if (gate) {
        const size_t io = ((size_t)g * (pix_per_group / HW) + pix / HW) * C + c0;
        f32x4 gt[NV], dv[NV];
        ldf<NV>(gate + io, gt);
        ldf<NV>(dsv + io, dv);
#pragma unroll
        for (int h = 0; h < NV; ++h) d[h] = d[h] * gt[h] + dv[h] * (1.f / (float)HW);
    }
    if (act == 2) {
        f32x4 sc[NV], sh[NV];
        ldf<NV>(scale + g * C + c0, sc);
        ldf<NV>(shift + g * C + c0, sh);
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const f32x4 v = yy[h] * sc[h] + sh[h];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[h][k] *= swish_grad<NV == 2>(v[k]);
        }
    }
    if (rowscale) {
        const float rs = rowscale[(size_t)g * (pix_per_group / HW) + pix / HW];
#pragma unroll
        for (int h = 0; h < NV; ++h) d[h] = d[h] * rs;
    }
    f32x4 a_[NV], b_[NV], c_[NV];
    ldf<NV>(ca + g * C + c0, a_);
    ldf<NV>(cb + g * C + c0, b_);
    ldf<NV>(cc + g * C + c0, c_);
#pragma unroll
    for (int h = 0; h < NV; ++h) d[h] = a_[h] * d[h] + b_[h] * yy[h] + c_[h];
    stv<NV>(dy + o, d);
}
// EW_R pixels per thread (see bnact_apply_rows_kernel): ca / cb / cc / scale / shift -- up to 10 NV 16-B parameter loads
// against 2 data loads per piece in the form above -- are loaded once per thread; the per-image vectors (gate, d s,
// rowscale) stay per piece.  Same per-element arithmetic.
template <typename TY, typename TA>
__global__ __launch_bounds__(256) void bnact_bwd_apply_rows_kernel(const TA* __restrict__ dz, const TY* __restrict__ y,
                                                                   const float* __restrict__ ca, const float* __restrict__ cb,
                                                                   const float* __restrict__ cc, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ rowscale, TY* __restrict__ dy,
                                                                   int pix_per_group, int HW, int C, int act,
                                                                   const float* __restrict__ gate, const float* __restrict__ dsv)
{
    constexpr int NV = NvOf<TY, TA>::NV;
    const int g = blockIdx.y;
    const int Q = C / (4 * NV);
    const int T = blockDim.x;
    const int c0 = (int)(threadIdx.x % Q) * 4 * NV;
    const int64_t nq = (int64_t)pix_per_group * Q;
    const size_t base = (size_t)g * pix_per_group * C;
    f32x4 a_[NV], b_[NV], c_[NV], sc[NV], sh[NV];
    ldf<NV>(ca + g * C + c0, a_);
    ldf<NV>(cb + g * C + c0, b_);
    ldf<NV>(cc + g * C + c0, c_);
    if (act == 2) {
        ldf<NV>(scale + g * C + c0, sc);
        ldf<NV>(shift + g * C + c0, sh);
    }
    f32x4 d[EW_R][NV], yy[EW_R][NV];
    int64_t idx[EW_R];
#pragma unroll
    for (int k = 0; k < EW_R; ++k) {
        idx[k] = ((int64_t)blockIdx.x * EW_R + k) * T + threadIdx.x;
        const size_t o = base + (size_t)min(idx[k], nq - 1) * (4 * NV);
        ldv<NV>(dz + o, d[k]);
        ldv<NV>(y + o, yy[k]);
    }
#pragma unroll
    for (int k = 0; k < EW_R; ++k) {
        if (idx[k] >= nq) break;
        const int64_t pix = idx[k] / Q;
        if (gate) {
            const size_t io = ((size_t)g * (pix_per_group / HW) + pix / HW) * C + c0;
            f32x4 gt[NV], dv[NV];
            ldf<NV>(gate + io, gt);
            ldf<NV>(dsv + io, dv);
#pragma unroll
            for (int h = 0; h < NV; ++h) d[k][h] = d[k][h] * gt[h] + dv[h] * (1.f / (float)HW);
        }
        if (act == 2) {
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                const f32x4 v = yy[k][h] * sc[h] + sh[h];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[k][h][q] *= swish_grad<NV == 2>(v[q]);
            }
        }
        if (rowscale) {
            const float rs = rowscale[(size_t)g * (pix_per_group / HW) + pix / HW];
#pragma unroll
            for (int h = 0; h < NV; ++h) d[k][h] = d[k][h] * rs;
        }
#pragma unroll
        for (int h = 0; h < NV; ++h) d[k][h] = a_[h] * d[k][h] + b_[h] * yy[k][h] + c_[h];
        stv<NV>(dy + base + (size_t)idx[k] * (4 * NV), d[k]);
    }
}
// dz is stored as the activations are (ta), y and the result dy as the raw conv output is (ty)
void k_bnact_bwd_apply(const void* dz, int ta, const void* y, int ty, const float* ca, const float* cb, const float* cc,
                       const float* scale, const float* shift, const float* rowscale, void* dy, int groups,
                       int pix_per_group, int HW, int C, int act, const float* gate, const float* dsv, hipStream_t s)
{
    const int nv = (ty == DT_BF16 && ta == DT_BF16) ? 2 : 1;
    const int Q = C / (4 * nv);
    if (ew_rows_on(Q, nv, 2)) {
        const int T = (256 / Q) * Q;
        const dim3 rgrid(cdiv((int64_t)pix_per_group * Q, (int64_t)EW_R * T), groups);
        if (ty == DT_F32 && ta == DT_F32)
            hipLaunchKernelGGL((bnact_bwd_apply_rows_kernel<float, float>), rgrid, dim3(T), 0, s, cp<float>(dz), cp<float>(y), ca,
                               cb, cc, scale, shift, rowscale, mp<float>(dy), pix_per_group, HW, C, act, gate, dsv);
        else if (ty == DT_F32)
            hipLaunchKernelGGL((bnact_bwd_apply_rows_kernel<float, bf16>), rgrid, dim3(T), 0, s, cp<bf16>(dz), cp<float>(y), ca,
                               cb, cc, scale, shift, rowscale, mp<float>(dy), pix_per_group, HW, C, act, gate, dsv);
        else
            hipLaunchKernelGGL((bnact_bwd_apply_rows_kernel<bf16, bf16>), rgrid, dim3(T), 0, s, cp<bf16>(dz), cp<bf16>(y), ca, cb,
                               cc, scale, shift, rowscale, mp<bf16>(dy), pix_per_group, HW, C, act, gate, dsv);
        return;
    }
    const dim3 grid(cdiv((int64_t)pix_per_group * (C / (4 * nv)), 256), groups);
    if (ty == DT_F32 && ta == DT_F32)
        hipLaunchKernelGGL((bnact_bwd_apply_kernel<float, float>), grid, dim3(256), 0, s, cp<float>(dz), cp<float>(y), ca, cb, cc,
                           scale, shift, rowscale, mp<float>(dy), pix_per_group, HW, C, act, gate, dsv);
    else if (ty == DT_F32)
        hipLaunchKernelGGL((bnact_bwd_apply_kernel<float, bf16>), grid, dim3(256), 0, s, cp<bf16>(dz), cp<float>(y), ca, cb, cc,
                           scale, shift, rowscale, mp<float>(dy), pix_per_group, HW, C, act, gate, dsv);
    else
        hipLaunchKernelGGL((bnact_bwd_apply_kernel<bf16, bf16>), grid, dim3(256), 0, s, cp<bf16>(dz), cp<bf16>(y), ca, cb, cc,
                           scale, shift, rowscale, mp<bf16>(dy), pix_per_group, HW, C, act, gate, dsv);
}

// ------------------------------------------------------------ depthwise conv ---
// The shipped path for EfficientNet-B0's shapes (TF-"same" padding of even inputs) is the row-uniform family dw_rowu_*
// further down.  The generic per-pixel kernels remain as the fallback for other paddings / odd inputs and as the yardstick
// of tests/test_effnet_gpu.py (FM_DW_GENERIC=1); the register-blocked weight-gradient kernels dw_wgrad_blk* still run the
// fp32 3x3 layers (dw_wgrad_full).  The register-blocked forward / data-gradient kernels, the LDS-tiled kernel and the
// kernel-row weight gradient of rounds 1-2 lost to the row-uniform family on every layer and were removed in round 3.
#define LD4Z(row, idx, stride, ok) ((ok) ? ld4((row) + (size_t)(idx) * (stride)) : f32x4{0.f, 0.f, 0.f, 0.f})
#define ROW_FENCE()
#define DW_LB(n) __launch_bounds__(256)
#define ROW_SKIP(rv) if (!(rv)) continue
#define ROW_OK(rv) (rv)
// x [imgs][Hi][Wi][C], w [K*K][C], y [imgs][Ho][Wo][C]; pad_t/pad_l = TF-same top/left padding.
// Optional fused eval epilogue: y = act(y*scale+shift).
template <int K, typename T>
__global__ void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                              const float* __restrict__ scale, const float* __restrict__ shift, int imgs, int Hi,
                              int Wi, int Ho, int Wo, int C, int stride, int pad_t, int pad_l, int act)
{
    const int Q = C >> 2;
    const int64_t n = (int64_t)imgs * Ho * Wo * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int img = (int)(t / Ho);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int ih = oh * stride + kh - pad_t;
        if ((unsigned)ih >= (unsigned)Hi) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int iw = ow * stride + kw - pad_l;
            if ((unsigned)iw >= (unsigned)Wi) continue;
            acc += ld4(x + ((size_t)(img * Hi + ih) * Wi + iw) * C + cq * 4) *
                   ld4(w + (kh * K + kw) * C + cq * 4);
        }
    }
    if (scale) {
        acc = acc * ld4(scale + cq * 4) + ld4(shift + cq * 4);
        acc = act_fwd<VecOf<T>::NV == 2>(acc, act);
    }
    st4(y + i * 4, acc);
}
static inline bool dw_blk_ok(int K, int S, int Hi, int Wi, int pad_t, int pad_l)
{
    const int pt = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
    return (K == 3 || K == 5) && (S == 1 || S == 2) && pad_t == pt && pad_l == pt && !getenv("FM_DW_GENERIC");
}
// ---- row-uniform stride-1 depthwise convolution -----------------------------------------------------------------------
// The register-blocked kernels above spend ~5 VALU instructions per FMA on 64-bit addresses, bounds compares / selects
// and (bf16) one branch + s_waitcnt vmcnt(0) PER LOAD.  Here the geometry is wave-uniform: a wave owns one (image, pair
// of output rows) at a time and its lanes are (4-column block, channel quad) of that row pair, so
//  * an input row is ONE buffer descriptor (base = the row, num_records = its bytes; 0 records for rows above / below the
//    image): the hardware range check returns zeros for the left / right / top / bottom halo and drops the stores of
//    partial blocks -- no compare, no select, no branch;
//  * the per-lane byte offsets of the K+3 columns are computed once per kernel (32-bit), the row enters through SGPRs;
//  * the lane's K*K weights sit in LDS ([tap][lane], staged once per block; the 4 waves of a block share the lane ->
//    (column block, quad) map and take different row pairs), plus K rows of zeros that serve as "kernel row K";
//  * rows are double-buffered in packed registers (row r+1 is in flight while row r is computed) and the row loop is a
//    REAL loop, unrolled by two by hand so the two buffers and the two weight rows (the lower output row uses the
//    previous input row's weights) alternate without moves.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct RawOf;
template <> struct RawOf<float> { typedef u32x4 type; };
template <> struct RawOf<bf16> { typedef u32x2 type; };
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void raw_load(rsrc_t r, int voff, u32x4& v) { v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0); }
__device__ __forceinline__ void raw_load(rsrc_t r, int voff, u32x2& v) { v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0); }
__device__ __forceinline__ f32x4 raw_cvt(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ f32x4 raw_cvt(u32x2 v)
{
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
}
__device__ __forceinline__ void raw_store(rsrc_t r, int voff, f32x4 v, float*)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ void raw_store(rsrc_t r, int voff, f32x4 v, bf16*)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4)), r, voff, 0, 0);
}
// descriptor of image row `ih` of image `img` (wave-uniform arguments); rows outside the image get 0 records
template <typename T>
__device__ __forceinline__ rsrc_t row_rsrc(const T* x, int img, int ih, int H, int rowelems)
{
    const bool ok = (unsigned)ih < (unsigned)H;
    const T* base = x + ((size_t)img * H + (ok ? ih : 0)) * rowelems;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, ok ? rowelems * (int)sizeof(T) : 0, 0x00020000);
}
// ST: train-mode forward -- the per-channel sum and sum of squares of the STORED outputs (BN1's batch statistics) are
// taken here from the registers instead of by a separate pass over y: every block leaves one [2][64-lane] record
// (4 waves folded in a fixed order), dw_rowu_wgrad_reduce sums the records of a channel quad per statistics group.
template <typename T> __device__ __forceinline__ f32x4 stored_value(f32x4 v)
{
    if constexpr (sizeof(T) == 2) return __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
    else return v;
}
// (channel chunk, row group) of this block.  xr > 0: XCD-aware order -- the blocks of xr consecutive row groups (all their
// channel chunks) get ids congruent mod 8, i.e. ONE XCD: neighbouring row groups share their K-1 halo rows through that
// L2 instead of fetching them twice.  The grid is padded to whole batches of 8 groups; false = a padding block.
__device__ __forceinline__ bool rowu_block(int nchunk, int nrg, int xr, int& chunk, int& rg)
{
    const int id = blockIdx.x;
    if (xr <= 0) { chunk = id % nchunk; rg = id / nchunk; return rg < nrg; }
    const int G = nchunk * xr;
    const int x = id & 7, q = id >> 3;
    const int g = (q / G) * 8 + x, within = q - (q / G) * G;
    rg = g * xr + within / nchunk;
    chunk = within - (within / nchunk) * nchunk;
    return rg < nrg;
}
static inline int rowu_grid(int nchunk, int nrg, int xr)
{
    if (xr <= 0) return nchunk * nrg;
    const int ngroups = (nrg + xr - 1) / xr;
    return (ngroups + 7) / 8 * 8 * nchunk * xr;
}
static inline int rowu_xr() { static const int v = fm_tune("FM_DW_XR", 4); return v; }
// ST = 2 (data gradient of a block with an expand conv): the BN0-backward sums  S1 = sum dz*swish'(v),  S2 = sum dz*swish'(v)*xhat
// (dz = the gradient this kernel stores, v = y_e*scale+shift, xhat = (y_e-mean)*istd) are taken here as well: y_e is read
// at the output positions, the separate reduction pass over (dz, y_e) is gone.  bnq = {mean, istd, scale, shift} [groups][C].
struct BnQuad { const float *mean, *istd, *scale, *shift; };
template <int K, typename T, int ST = 0>
__global__ __launch_bounds__(256) void dw_rowu_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int nrp, int H, int W, int C, int act, int flip, int nchunk, int rpb,
                                                      f32x4* __restrict__ rec = nullptr, const T* __restrict__ ye = nullptr,
                                                      BnQuad bnq = BnQuad{nullptr, nullptr, nullptr, nullptr}, int rp_per_group = 1,
                                                      int xr = 0)
{
    constexpr int PT = (K - 1) / 2, NIN = K + 3, ES = (int)sizeof(T);
    constexpr bool FAST = VecOf<T>::NV == 2;
    typedef typename RawOf<T>::type raw_t;
    __shared__ f32x4 ws[(K * K + K) * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int chunk, rg;
    if (!rowu_block(nchunk, (nrp + rpb - 1) / rpb, xr, chunk, rg)) return;
    const int bidx = rg * nchunk + chunk;                      // record slot (independent of the block order)
    const int Q = C >> 2, WB = (W + 3) >> 2, HB = (H + 1) >> 1;
    const int id = chunk * 64 + lane;
    const bool lv = id < WB * Q;
    const int owb = lv ? id / Q : 0, cq = lv ? id - owb * Q : 0;
    for (int t = wave; t < K * K + K; t += 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < K * K) v = ld4(w + (flip ? K * K - 1 - t : t) * C + cq * 4);
        ws[t * 64 + lane] = v;
    }
    int voff[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) voff[j] = lv ? ((owb * 4 - PT + j) * C + cq * 4) * ES : 0x7f000000;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) { sc = ld4(scale + cq * 4); sh = ld4(shift + cq * 4); }
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    float cmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cmask[j] = (lv && owb * 4 + j < W) ? 1.f : 0.f;
    __syncthreads();
    const int rowelems = W * C;
    const int rp1 = min(nrp, (rg + 1) * rpb);
    const f32x4* wl = ws + lane;
    for (int rp = rg * rpb + wave; rp < rp1; rp += 4) {
        const int img = rp / HB, oh0 = (rp - img * HB) * 2;
        f32x4 acc[2][4], wa[K], wb[K];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kw = 0; kw < K; ++kw) wb[kw] = f32x4{0.f, 0.f, 0.f, 0.f};
        raw_t ra[NIN], rb[NIN];
        {
            const rsrc_t r = row_rsrc(x, img, oh0 - PT, H, rowelems);
#pragma unroll
            for (int j = 0; j < NIN; ++j) raw_load(r, voff[j], ra[j]);
        }
#pragma unroll 1
        for (int ir = 0; ir <= K; ir += 2) {
            {   // row ir+1 -> rb (always exists: K is odd)
                const rsrc_t r = row_rsrc(x, img, oh0 - PT + ir + 1, H, rowelems);
#pragma unroll
                for (int j = 0; j < NIN; ++j) raw_load(r, voff[j], rb[j]);
            }
            {   // compute row ir: taps ir (upper output row, wa) and ir-1 (lower output row, wb)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) wa[kw] = wl[(ir * K + kw) * 64];
                f32x4 xin[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) xin[j] = raw_cvt(ra[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        acc[0][j] += xin[j + kw] * wa[kw];
                        acc[1][j] += xin[j + kw] * wb[kw];
                    }
            }
            if (ir + 2 <= K) {
                const rsrc_t r = row_rsrc(x, img, oh0 - PT + ir + 2, H, rowelems);
#pragma unroll
                for (int j = 0; j < NIN; ++j) raw_load(r, voff[j], ra[j]);
            }
            {   // compute row ir+1: taps ir+1 (zeros when ir+1 == K) into wb, previous = wa
#pragma unroll
                for (int kw = 0; kw < K; ++kw) wb[kw] = wl[((ir + 1) * K + kw) * 64];
                f32x4 xin[NIN];
#pragma unroll
                for (int j = 0; j < NIN; ++j) xin[j] = raw_cvt(rb[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        acc[0][j] += xin[j + kw] * wb[kw];
                        acc[1][j] += xin[j + kw] * wa[kw];
                    }
            }
        }
        raw_t rye[ST == 2 ? 2 : 1][4];
        f32x4 bmu, bis, bsc, bsh;
        if constexpr (ST == 2) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const rsrc_t ry = row_rsrc(ye, img, oh0 + r, H, rowelems);
#pragma unroll
                for (int j = 0; j < 4; ++j) raw_load(ry, voff[j + PT], rye[r][j]);
            }
            const int g = rp / rp_per_group;
            bmu = ld4(bnq.mean + g * C + cq * 4); bis = ld4(bnq.istd + g * C + cq * 4);
            bsc = ld4(bnq.scale + g * C + cq * 4); bsh = ld4(bnq.shift + g * C + cq * 4);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const rsrc_t ro = row_rsrc(y, img, oh0 + r, H, rowelems);
            const float rm = oh0 + r < H ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[r][j];
                if (scale) v = act_fwd<FAST>(v * sc + sh, act);
                raw_store(ro, voff[j + PT], v, (T*)nullptr);
                if constexpr (ST == 1) {
                    const f32x4 q = stored_value<T>(v) * (rm * cmask[j]);
                    st1 += q;
                    st2 += q * q;
                }
                if constexpr (ST == 3) st1 += stored_value<T>(v) * (rm * cmask[j]);     // eval: squeeze-excite pooling
                if constexpr (ST == 2) {
                    const f32x4 yy = raw_cvt(rye[r][j]);
                    const f32x4 u = yy * bsc + bsh;
                    f32x4 d = stored_value<T>(v) * (rm * cmask[j]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] *= swish_grad<FAST>(u[k]);
                    st1 += d;
                    st2 += d * ((yy - bmu) * bis);
                }
            }
        }
    }
    if constexpr (ST != 0) {
        __shared__ f32x4 red[3][2][64];
        if (wave > 0) { red[wave - 1][0][lane] = st1; red[wave - 1][1][lane] = st2; }
        __syncthreads();
        if (wave == 0) {
            constexpr int NR = ST == 3 ? 1 : 2;       // the pooling record holds the sum only
            rec[((size_t)bidx * NR + 0) * 64 + lane] = ((st1 + red[0][0][lane]) + red[1][0][lane]) + red[2][0][lane];
            if constexpr (ST != 3)
                rec[((size_t)bidx * 2 + 1) * 64 + lane] = ((st2 + red[0][1][lane]) + red[1][1][lane]) + red[2][1][lane];
        }
    }
}
// stride-2 forward in the same row-uniform form: a wave step = one output row (img, oh), lanes = (4-column block of the
// output row, channel quad); K input rows of 6 + K columns each, double-buffered (K is odd: the last row is peeled).
template <int K, typename T, bool PF, int ST = 0>
__global__ __launch_bounds__(256) void dw_rowu_s2_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int nrows, int Hi, int Wi, int Ho, int Wo, int C, int act, int nchunk,
                                                         int rpb, f32x4* __restrict__ rec = nullptr)
{
    constexpr int PT = (K - 2) / 2, NIN = 6 + K, ES = (int)sizeof(T);
    constexpr bool FAST = VecOf<T>::NV == 2;
    typedef typename RawOf<T>::type raw_t;
    __shared__ f32x4 ws[K * K * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x % nchunk, rg = blockIdx.x / nchunk;
    const int Q = C >> 2, WB = (Wo + 3) >> 2;
    const int id = chunk * 64 + lane;
    const bool lv = id < WB * Q;
    const int owb = lv ? id / Q : 0, cq = lv ? id - owb * Q : 0;
    for (int t = wave; t < K * K; t += 4) ws[t * 64 + lane] = ld4(w + t * C + cq * 4);
    int voff[NIN], vout[4];
#pragma unroll
    for (int j = 0; j < NIN; ++j) voff[j] = lv ? ((owb * 8 - PT + j) * C + cq * 4) * ES : 0x7f000000;
#pragma unroll
    for (int j = 0; j < 4; ++j) vout[j] = lv ? ((owb * 4 + j) * C + cq * 4) * ES : 0x7f000000;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) { sc = ld4(scale + cq * 4); sh = ld4(shift + cq * 4); }
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    float cmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cmask[j] = (lv && owb * 4 + j < Wo) ? 1.f : 0.f;
    __syncthreads();
    const int in_row = Wi * C, out_row = Wo * C;
    const int r1 = min(nrows, (rg + 1) * rpb);
    const f32x4* wl = ws + lane;
    for (int row = rg * rpb + wave; row < r1; row += 4) {
        const int img = row / Ho, oh = row - img * Ho;
        const int ih0 = oh * 2 - PT;
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        raw_t ra[NIN], rb[PF ? NIN : 1];
        auto load = [&](raw_t (&dst)[NIN], int kh) {
            const rsrc_t r = row_rsrc(x, img, ih0 + kh, Hi, in_row);
#pragma unroll
            for (int j = 0; j < NIN; ++j) raw_load(r, voff[j], dst[j]);
        };
        auto comp = [&](const raw_t (&src)[NIN], int kh) {
            f32x4 wr[K], xin[NIN];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) wr[kw] = wl[(kh * K + kw) * 64];
#pragma unroll
            for (int j = 0; j < NIN; ++j) xin[j] = raw_cvt(src[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[j] += xin[2 * j + kw] * wr[kw];
        };
        if constexpr (PF) {
            load(ra, 0);
#pragma unroll 1
            for (int kh = 0; kh < K - 1; kh += 2) {
                load(rb, kh + 1);
                comp(ra, kh);
                load(ra, kh + 2);
                comp(rb, kh + 1);
            }
            comp(ra, K - 1);
        } else {
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                load(ra, kh);
                comp(ra, kh);
            }
        }
        const rsrc_t ro = row_rsrc(y, img, oh, Ho, out_row);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j];
            if (scale) v = act_fwd<FAST>(v * sc + sh, act);
            raw_store(ro, vout[j], v, (T*)nullptr);
            if constexpr (ST == 1) {
                const f32x4 q = stored_value<T>(v) * cmask[j];
                st1 += q;
                st2 += q * q;
            }
            if constexpr (ST == 3) st1 += stored_value<T>(v) * cmask[j];
        }
    }
    if constexpr (ST != 0) {
        __shared__ f32x4 red[3][2][64];
        if (wave > 0) { red[wave - 1][0][lane] = st1; red[wave - 1][1][lane] = st2; }
        __syncthreads();
        if (wave == 0) {
            constexpr int NR = ST == 3 ? 1 : 2;       // the pooling record holds the sum only
            rec[((size_t)blockIdx.x * NR + 0) * 64 + lane] = ((st1 + red[0][0][lane]) + red[1][0][lane]) + red[2][0][lane];
            if constexpr (ST != 3)
                rec[((size_t)blockIdx.x * 2 + 1) * 64 + lane] = ((st2 + red[0][1][lane]) + red[1][1][lane]) + red[2][1][lane];
        }
    }
}
// stride-2 data gradient, row-uniform: a wave step = one INPUT row (img, ih), lanes = (4-column block of that row, channel
// quad).  Only the kernel rows kh with (ih + PT - kh) even reach it -- a wave-uniform choice, (K+1)/2 candidate rows
// (a candidate beyond the kernel reads a zero tap row and a 0-record descriptor); which (column, kw) pairs are exact
// divisions is compile-time because the block starts at a multiple of 4.
template <int K, typename T, int ST = 0>
__global__ __launch_bounds__(256) void dw_rowu_dgrad_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                               T* __restrict__ dx, int nrows, int Hi, int Wi, int Ho, int Wo,
                                                               int C, int nchunk, int rpb, f32x4* __restrict__ rec = nullptr,
                                                               const T* __restrict__ ye = nullptr,
                                                               BnQuad bnq = BnQuad{nullptr, nullptr, nullptr, nullptr},
                                                               int rows_per_group = 1)
{
    constexpr int PT = (K - 2) / 2, ES = (int)sizeof(T);
    constexpr int OMIN = -((K - PT) / 2), OMAX = (3 + PT) / 2, NC = OMAX - OMIN + 1, NR = (K + 1) / 2;
    typedef typename RawOf<T>::type raw_t;
    __shared__ f32x4 ws[(K * K + K) * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x % nchunk, rg = blockIdx.x / nchunk;
    const int Q = C >> 2, WB = (Wi + 3) >> 2;
    const int id = chunk * 64 + lane;
    const bool lv = id < WB * Q;
    const int iwb = lv ? id / Q : 0, cq = lv ? id - iwb * Q : 0;
    for (int t = wave; t < K * K + K; t += 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < K * K) v = ld4(w + t * C + cq * 4);
        ws[t * 64 + lane] = v;
    }
    int voff[NC], vout[4];
#pragma unroll
    for (int c = 0; c < NC; ++c) voff[c] = lv ? ((iwb * 2 + OMIN + c) * C + cq * 4) * ES : 0x7f000000;
#pragma unroll
    for (int j = 0; j < 4; ++j) vout[j] = lv ? ((iwb * 4 + j) * C + cq * 4) * ES : 0x7f000000;
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    float cmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cmask[j] = (lv && iwb * 4 + j < Wi) ? 1.f : 0.f;
    __syncthreads();
    const int in_row = Wi * C, out_row = Wo * C;
    const int r1 = min(nrows, (rg + 1) * rpb);
    const f32x4* wl = ws + lane;
    for (int row = rg * rpb + wave; row < r1; row += 4) {
        const int img = row / Hi, ih = row - img * Hi;
        const int kh0 = (ih + PT) & 1;
        raw_t raw[NR][NC];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int kh = kh0 + 2 * r;
            const int a = ih + PT - kh;                                   // even; may be negative or beyond the last row
            const rsrc_t rs = row_rsrc(dy, img, kh < K ? a >> 1 : -1, Ho, out_row);
#pragma unroll
            for (int c = 0; c < NC; ++c) raw_load(rs, voff[c], raw[r][c]);
        }
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int kh = min(kh0 + 2 * r, K);                           // K = the zero tap row
            f32x4 wr[K], din[NC];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) wr[kw] = wl[(kh * K + kw) * 64];
#pragma unroll
            for (int c = 0; c < NC; ++c) din[c] = raw_cvt(raw[r][c]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const int d = j + PT - kw;
                    if (d % 2 != 0) continue;
                    acc[j] += din[d / 2 - OMIN] * wr[kw];
                }
        }
        const rsrc_t ro = row_rsrc(dx, img, ih, Hi, in_row);
#pragma unroll
        for (int j = 0; j < 4; ++j) raw_store(ro, vout[j], acc[j], (T*)nullptr);
        if constexpr (ST == 2) {
            raw_t rye[4];
            const rsrc_t ry = row_rsrc(ye, img, ih, Hi, in_row);
#pragma unroll
            for (int j = 0; j < 4; ++j) raw_load(ry, vout[j], rye[j]);
            const int g = row / rows_per_group;
            const f32x4 bmu = ld4(bnq.mean + g * C + cq * 4), bis = ld4(bnq.istd + g * C + cq * 4);
            const f32x4 bsc = ld4(bnq.scale + g * C + cq * 4), bsh = ld4(bnq.shift + g * C + cq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 yy = raw_cvt(rye[j]);
                const f32x4 u = yy * bsc + bsh;
                f32x4 d = stored_value<T>(acc[j]) * cmask[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] *= swish_grad<sizeof(T) == 2>(u[k]);
                st1 += d;
                st2 += d * ((yy - bmu) * bis);
            }
        }
    }
    if constexpr (ST != 0) {
        __shared__ f32x4 red[3][2][64];
        if (wave > 0) { red[wave - 1][0][lane] = st1; red[wave - 1][1][lane] = st2; }
        __syncthreads();
        if (wave == 0) {
            constexpr int NR = ST == 3 ? 1 : 2;       // the pooling record holds the sum only
            rec[((size_t)blockIdx.x * NR + 0) * 64 + lane] = ((st1 + red[0][0][lane]) + red[1][0][lane]) + red[2][0][lane];
            if constexpr (ST != 3)
                rec[((size_t)blockIdx.x * 2 + 1) * 64 + lane] = ((st2 + red[0][1][lane]) + red[1][1][lane]) + red[2][1][lane];
        }
    }
}
__global__ void dw_rowu_wgrad_reduce(const f32x4* __restrict__ part, float* __restrict__ out, int KK, int Q, int WB, int nchunk,
                                     int nrg, int nsplit);
// Train-mode statistics request of a depthwise forward: rec = record workspace, out = [groups][1][2][C] partials for
// k_bn_finalize (tiles = 1).  Honoured when the row steps of one statistics group fill whole blocks.
struct DwStats { float* rec; float* out; int groups; const void* ye; BnQuad bnq; float* pool; };   // ye: the BN0-backward sums;
// pool != null (eval forward): per-image channel sums of the activated output -> pool [imgs][C] (the squeeze-excite pooling)
static inline int dw_pool_rpb(int steps_per_image)
{
    for (int r = std::min(16, steps_per_image); r >= 1; --r)
        if (steps_per_image % r == 0) return r;
    return 1;
}
constexpr int DW_ST_SPLITS = 8;              // partials per group the statistics reducer leaves (k_bn_finalize tiles)
// row groups per statistics group: the largest divisor of the group's steps that keeps the launch near 3072 blocks
// (every block leaves a record: with the forward's 16-step blocks the 112x112 layer would leave 14 336 of them)
static inline int dw_stats_rowgroups(int steps_per_group, int nchunk, int groups)
{
    const int target = std::max(1, 3072 / (nchunk * groups));
    for (int d = std::min(target, steps_per_group); d >= 1; --d)
        if (steps_per_group % d == 0) return 4 * d >= target || d == steps_per_group ? d : 0;
    return 0;
}
template <int K, typename T>
static bool dw_rowu_launch(const T* x, const float* w, T* y, const float* scale, const float* shift, int imgs, int H, int W,
                           int C, int act, int flip, hipStream_t s, const DwStats* st = nullptr)
{
    const int WB = (W + 3) / 4, HB = (H + 1) / 2, Q = C / 4;
    const int nchunk = (WB * Q + 63) / 64, nrp = imgs * HB;
    static const int rpb_env = fm_tune("FM_DW_RPB", 16);
    int rpb = std::max(4, rpb_env);
    const int xr = rowu_xr();
    const BnQuad nobn{nullptr, nullptr, nullptr, nullptr};
    const T* noy = nullptr;
    if (st && st->pool) {          // eval: blocks stay inside one image, one record each, summed per image
        rpb = dw_pool_rpb(HB);
        const int gpi = HB / rpb;
        hipLaunchKernelGGL((dw_rowu_kernel<K, T, 3>), dim3(rowu_grid(nchunk, gpi * imgs, xr)), dim3(256), 0, s, x, w, y, scale, shift, nrp,
                           H, W, C, act, flip, nchunk, rpb, reinterpret_cast<f32x4*>(st->rec), noy, nobn, 1, xr);
        hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16, 1, imgs), dim3(256), 0, s, reinterpret_cast<const f32x4*>(st->rec),
                           st->pool, 1, Q, WB, nchunk, gpi, 1);
        return true;
    }
    if (st && nrp % st->groups == 0) {
        const int nrg_g = dw_stats_rowgroups(nrp / st->groups, nchunk, st->groups);
        if (nrg_g) {
            rpb = nrp / st->groups / nrg_g;
            const dim3 grid(rowu_grid(nchunk, nrg_g * st->groups, xr));
            if (st->ye)
                hipLaunchKernelGGL((dw_rowu_kernel<K, T, 2>), grid, dim3(256), 0, s, x, w, y, scale, shift, nrp, H, W, C, act, flip, nchunk,
                                   rpb, reinterpret_cast<f32x4*>(st->rec), reinterpret_cast<const T*>(st->ye), st->bnq,
                                   nrp / st->groups, xr);
            else
                hipLaunchKernelGGL((dw_rowu_kernel<K, T, 1>), grid, dim3(256), 0, s, x, w, y, scale, shift, nrp, H, W, C, act, flip, nchunk,
                                   rpb, reinterpret_cast<f32x4*>(st->rec), noy, nobn, 1, xr);
            hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16 * DW_ST_SPLITS, 2, st->groups), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(st->rec), st->out, 2, Q, WB, nchunk, nrg_g, DW_ST_SPLITS);
            return true;
        }
    }
    f32x4* norec = nullptr;
    hipLaunchKernelGGL((dw_rowu_kernel<K, T>), dim3(rowu_grid(nchunk, (nrp + rpb - 1) / rpb, xr)), dim3(256), 0, s, x, w, y, scale, shift,
                       nrp, H, W, C, act, flip, nchunk, rpb, norec, noy, nobn, 1, xr);
    return false;
}

template <int K, typename T>
static bool dw_rowu_s2_launch(const T* x, const float* w, T* y, const float* scale, const float* shift, int imgs, int Hi, int Wi,
                              int Ho, int Wo, int C, int act, hipStream_t s, const DwStats* st = nullptr)
{
    const int WB = (Wo + 3) / 4, Q = C / 4;
    const int nchunk = (WB * Q + 63) / 64, nrows = imgs * Ho;
    static const int rpb_env = fm_tune("FM_DW_RPB", 16);
    static const int pf_env = fm_tune("FM_DW_PF", -1);
    int rpb = std::max(4, rpb_env);
    const bool pf = pf_env >= 0 ? pf_env != 0 : true;      // fp32 5x5 stride 2: 0.44 -> 0.32 ms with the second row buffer
    if (st && st->pool) {
        rpb = dw_pool_rpb(Ho);
        const int gpi = Ho / rpb;
        hipLaunchKernelGGL((dw_rowu_s2_kernel<K, T, true, 3>), dim3(nchunk * gpi * imgs), dim3(256), 0, s, x, w, y, scale, shift, nrows, Hi,
                           Wi, Ho, Wo, C, act, nchunk, rpb, reinterpret_cast<f32x4*>(st->rec));
        hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16, 1, imgs), dim3(256), 0, s, reinterpret_cast<const f32x4*>(st->rec),
                           st->pool, 1, Q, WB, nchunk, gpi, 1);
        return true;
    }
    if (st && nrows % st->groups == 0) {
        const int nrg_g = dw_stats_rowgroups(nrows / st->groups, nchunk, st->groups);
        if (nrg_g) {
            rpb = nrows / st->groups / nrg_g;
            hipLaunchKernelGGL((dw_rowu_s2_kernel<K, T, true, 1>), dim3(nchunk * nrg_g * st->groups), dim3(256), 0, s, x, w, y, scale,
                               shift, nrows, Hi, Wi, Ho, Wo, C, act, nchunk, rpb, reinterpret_cast<f32x4*>(st->rec));
            hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16 * DW_ST_SPLITS, 2, st->groups), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(st->rec), st->out, 2, Q, WB, nchunk, nrg_g, DW_ST_SPLITS);
            return true;
        }
    }
    const dim3 grid(nchunk * ((nrows + rpb - 1) / rpb));
    if (pf) hipLaunchKernelGGL((dw_rowu_s2_kernel<K, T, true>), grid, dim3(256), 0, s, x, w, y, scale, shift, nrows, Hi, Wi, Ho, Wo, C, act, nchunk, rpb);
    else hipLaunchKernelGGL((dw_rowu_s2_kernel<K, T, false>), grid, dim3(256), 0, s, x, w, y, scale, shift, nrows, Hi, Wi, Ho, Wo, C, act, nchunk, rpb);
    return false;
}
template <int K, typename T>
static bool dw_rowu_dgrad_s2_launch(const T* dy, const float* w, T* dx, int imgs, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s,
                                    const DwStats* st = nullptr)
{
    const int WB = (Wi + 3) / 4, Q = C / 4;
    const int nchunk = (WB * Q + 63) / 64, nrows = imgs * Hi;
    static const int rpb_env = fm_tune("FM_DW_RPB", 16);
    int rpb = std::max(4, rpb_env);
    if (st && st->ye && nrows % st->groups == 0) {
        const int nrg_g = dw_stats_rowgroups(nrows / st->groups, nchunk, st->groups);
        if (nrg_g) {
            rpb = nrows / st->groups / nrg_g;
            hipLaunchKernelGGL((dw_rowu_dgrad_s2_kernel<K, T, 2>), dim3(nchunk * nrg_g * st->groups), dim3(256), 0, s, dy, w, dx, nrows, Hi,
                               Wi, Ho, Wo, C, nchunk, rpb, reinterpret_cast<f32x4*>(st->rec), reinterpret_cast<const T*>(st->ye), st->bnq,
                               nrows / st->groups);
            hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16 * DW_ST_SPLITS, 2, st->groups), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(st->rec), st->out, 2, Q, WB, nchunk, nrg_g, DW_ST_SPLITS);
            return true;
        }
    }
    const dim3 grid(nchunk * ((nrows + rpb - 1) / rpb));
    hipLaunchKernelGGL((dw_rowu_dgrad_s2_kernel<K, T>), grid, dim3(256), 0, s, dy, w, dx, nrows, Hi, Wi, Ho, Wo, C, nchunk, rpb);
    return false;
}

// returns true when the statistics request `st` was served (st->out then holds one partial per group)
template <typename T>
static bool dw_fwd_t(const T* x, const float* w, T* y, const float* scale, const float* shift, int imgs, int Hi, int Wi,
                     int Ho, int Wo, int C, int K, int stride, int pad_t, int pad_l, int act, hipStream_t s,
                     const float* psc = nullptr, const float* psh = nullptr, int ipg = 1, const DwStats* st = nullptr)
{
    if (!fm_tune("FM_DW_STATS", 1)) st = nullptr;
    const dim3 blk(256);
    if (!psc && stride == 1 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l)) {
        if (K == 3) return dw_rowu_launch<3, T>(x, w, y, scale, shift, imgs, Hi, Wi, C, act, 0, s, st);
        return dw_rowu_launch<5, T>(x, w, y, scale, shift, imgs, Hi, Wi, C, act, 0, s, st);
    }
    if (!psc && stride == 2 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l) && Wi == 2 * Wo && Hi == 2 * Ho) {
        if (K == 3) return dw_rowu_s2_launch<3, T>(x, w, y, scale, shift, imgs, Hi, Wi, Ho, Wo, C, act, s, st);
        return dw_rowu_s2_launch<5, T>(x, w, y, scale, shift, imgs, Hi, Wi, Ho, Wo, C, act, s, st);
    }
    const dim3 grid(cdiv((int64_t)imgs * Ho * Wo * (C / 4), 256));
    if (K == 3) hipLaunchKernelGGL((dw_fwd_kernel<3, T>), grid, blk, 0, s, x, w, y, scale, shift, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l, act);
    else hipLaunchKernelGGL((dw_fwd_kernel<5, T>), grid, blk, 0, s, x, w, y, scale, shift, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l, act);
    return false;
}
// dt: storage type of x and y (DT_F32 / DT_BF16); weights and the BN affine are fp32
// pool_out != null (eval mode, with stats_rec as record workspace): also leave the per-image channel SUMS of the stored
// (activated) output in pool_out [imgs][C] -- the squeeze-excite pooling without another pass over y.
// stats_rec != null (train mode): also leave the per-channel sum / sum of squares of y as ONE partial per group in
// stats_out [groups][dw_stats_tiles()][2][C]; returns false when the launch shape could not do it (the caller then reduces y itself)
bool k_dw_fwd(const void* x, const float* w, void* y, int dt, const float* scale, const float* shift, int imgs, int Hi,
              int Wi, int Ho, int Wo, int C, int K, int stride, int pad_t, int pad_l, int act, hipStream_t s, float* stats_rec,
              float* stats_out, int groups, float* pool_out)
{
    const DwStats st{stats_rec, stats_out, groups, nullptr, BnQuad{nullptr, nullptr, nullptr, nullptr}, pool_out};
    const DwStats* sp = stats_rec ? &st : nullptr;
    if (dt == DT_F32)
        return dw_fwd_t(cp<float>(x), w, mp<float>(y), scale, shift, imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, act, s, nullptr,
                        nullptr, 1, sp);
    return dw_fwd_t(cp<bf16>(x), w, mp<bf16>(y), scale, shift, imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, act, s, nullptr,
                    nullptr, 1, sp);
}

template <int K, typename T>
__global__ void dw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                int imgs, int Hi, int Wi, int Ho, int Wo, int C, int stride, int pad_t, int pad_l)
{
    const int Q = C >> 2;
    const int64_t n = (int64_t)imgs * Hi * Wi * Q;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cq = (int)(i % Q);
    int64_t t = i / Q;
    const int iw = (int)(t % Wi); t /= Wi;
    const int ih = (int)(t % Hi);
    const int img = (int)(t / Hi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int a = ih + pad_t - kh;
        if (a < 0 || a % stride) continue;
        const int oh = a / stride;
        if (oh >= Ho) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int b = iw + pad_l - kw;
            if (b < 0 || b % stride) continue;
            const int ow = b / stride;
            if (ow >= Wo) continue;
            acc += ld4(dy + ((size_t)(img * Ho + oh) * Wo + ow) * C + cq * 4) *
                   ld4(w + (kh * K + kw) * C + cq * 4);
        }
    }
    st4(dx + i * 4, acc);
}
template <typename T>
static bool dw_dgrad_t(const T* dy, const float* w, T* dx, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                       int stride, int pad_t, int pad_l, hipStream_t s, const DwStats* st = nullptr)
{
    if (!fm_tune("FM_DW_STATS", 1)) st = nullptr;
    const dim3 blk(256);
    if (stride == 1 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l)) {
        // stride 1: dx = dy (*) rot180(w), the forward kernel with the rotated kernel (Hi == Ho, Wi == Wo)
        const float* nul = nullptr;
        if (K == 3) return dw_rowu_launch<3, T>(dy, w, dx, nul, nul, imgs, Hi, Wi, C, 0, 1, s, st);
        return dw_rowu_launch<5, T>(dy, w, dx, nul, nul, imgs, Hi, Wi, C, 0, 1, s, st);
    }
    if (stride == 2 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l) && Wi == 2 * Wo && Hi == 2 * Ho) {
        if (K == 3) return dw_rowu_dgrad_s2_launch<3, T>(dy, w, dx, imgs, Hi, Wi, Ho, Wo, C, s, st);
        return dw_rowu_dgrad_s2_launch<5, T>(dy, w, dx, imgs, Hi, Wi, Ho, Wo, C, s, st);
    }
    const dim3 grid(cdiv((int64_t)imgs * Hi * Wi * (C / 4), 256));
    if (K == 3) hipLaunchKernelGGL((dw_dgrad_kernel<3, T>), grid, blk, 0, s, dy, w, dx, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l);
    else hipLaunchKernelGGL((dw_dgrad_kernel<5, T>), grid, blk, 0, s, dy, w, dx, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l);
    return false;
}
// ye != null: dx feeds act(bn0(y_e)) -- also leave the BN0-backward sums (chan_reduce mode 1, swish) as
// dw_stats_tiles() partials per group in stats_out; returns false when the launch shape could not do it
bool k_dw_dgrad(const void* dy, const float* w, void* dx, int dt, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                int stride, int pad_t, int pad_l, hipStream_t s, const void* ye, const float* mean, const float* istd,
                const float* scale, const float* shift, float* stats_rec, float* stats_out, int groups)
{
    const DwStats st{stats_rec, stats_out, groups, ye, BnQuad{mean, istd, scale, shift}, nullptr};
    const DwStats* sp = ye ? &st : nullptr;
    if (dt == DT_F32) return dw_dgrad_t(cp<float>(dy), w, mp<float>(dx), imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, s, sp);
    return dw_dgrad_t(cp<bf16>(dy), w, mp<bf16>(dx), imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, s, sp);
}

// dw[kh][kw][c] = sum over output pixels of dy * x(shifted).  Thread = (channel quad, pixel lane):
// QT quads x P lanes per block (QT*P <= 256 threads, see dw_map), each block owns a chunk of output
// pixels; K*K float4 accumulators per thread, folded over the pixel lanes through LDS; partial
// [nblk][K*K][C], summed later in a fixed order (reduce_slabs) -> run-to-run deterministic.
template <int K, typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       float* __restrict__ part, int imgs, int Hi, int Wi, int Ho,
                                                       int Wo, int C, int stride, int pad_t, int pad_l, int QT, int P)
{
    __shared__ f32x4 red[256];
    const int cq = blockIdx.y * QT + threadIdx.x % QT, pl = threadIdx.x / QT;
    const int npix = imgs * Ho * Wo, nblk = gridDim.x;
    const int chunk = (npix + nblk - 1) / nblk;
    const int pb = blockIdx.x * chunk, pe = min(npix, pb + chunk);
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = pb + pl; p < pe; p += P) {
        const int img = p / (Ho * Wo);
        const int rem = p - img * Ho * Wo;
        const int oh = rem / Wo, ow = rem - oh * Wo;
        const f32x4 d = ld4(dy + (size_t)p * C + cq * 4);
        const T* xi = x + (size_t)img * Hi * Wi * C + cq * 4;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = oh * stride + kh - pad_t;
            if ((unsigned)ih >= (unsigned)Hi) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int iw = ow * stride + kw - pad_l;
                if ((unsigned)iw >= (unsigned)Wi) continue;
                acc[kh * K + kw] += d * ld4(xi + (size_t)(ih * Wi + iw) * C);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        __syncthreads();
        red[threadIdx.x] = acc[t];
        __syncthreads();
        if (pl == 0) {
            f32x4 v = red[threadIdx.x];
            for (int k = 1; k < P; ++k) v += red[k * QT + threadIdx.x];
            st4(part + ((size_t)blockIdx.x * K * K + t) * C + cq * 4, v);
        }
    }
}
// wgrad, register-blocked: the pixel loop walks blocks of 4 consecutive output columns
template <int K, int S, typename T>
__global__ DW_LB(2) void dw_wgrad_blk_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           float* __restrict__ part, int imgs, int Hi, int Wi,
                                                           int Ho, int Wo, int C, int QT, int P)
{
    constexpr int PT = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
    constexpr int NIN = 3 * S + K;
    __shared__ f32x4 red[256];
    const int cq = blockIdx.y * QT + threadIdx.x % QT, pl = threadIdx.x / QT;
    const int WB = (Wo + 3) >> 2;
    const int npb = imgs * Ho * WB, nblk = gridDim.x;
    const int chunk = (npb + nblk - 1) / nblk;
    const int pb = blockIdx.x * chunk, pe = min(npb, pb + chunk);
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = pb + pl; p < pe; p += P) {
        const int img = p / (Ho * WB);
        const int rem = p - img * Ho * WB;
        const int oh = rem / WB, ow0 = (rem - oh * WB) * 4;
        const T* dr = dy + ((size_t)(img * Ho + oh) * Wo) * C + cq * 4;
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = ow0 + j < Wo;
            d[j] = LD4Z(dr, ow0 + j, C, ok);
        }
        const int iw0 = ow0 * S - PT;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = oh * S + kh - PT;
            const bool rv = (unsigned)ih < (unsigned)Hi;
            ROW_SKIP(rv);
            const T* xr = x + ((size_t)(img * Hi + (rv ? ih : 0)) * Wi) * C + cq * 4;
            f32x4 xin[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int iw = iw0 + j;
                const bool ok = rv && (unsigned)iw < (unsigned)Wi;
                xin[j] = LD4Z(xr, iw, C, ok);
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[kh * K + kw] += d[j] * xin[j * S + kw];
            ROW_FENCE();
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        __syncthreads();
        red[threadIdx.x] = acc[t];
        __syncthreads();
        if (pl == 0) {
            f32x4 v = red[threadIdx.x];
            for (int k = 1; k < P; ++k) v += red[k * QT + threadIdx.x];
            st4(part + ((size_t)blockIdx.x * K * K + t) * C + cq * 4, v);
        }
    }
}
// stride-1 wgrad, pixel blocks of 2 output rows x 4 columns: K+1 input rows serve both rows of dy
template <int K, typename T>
__global__ DW_LB(2) void dw_wgrad_blk2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            float* __restrict__ part, int imgs, int Hi, int Wi,
                                                            int Ho, int Wo, int C, int QT, int P)
{
    constexpr int PT = (K - 1) / 2;
    constexpr int NIN = 3 + K;
    __shared__ f32x4 red[256];
    const int cq = blockIdx.y * QT + threadIdx.x % QT, pl = threadIdx.x / QT;
    const int WB = (Wo + 3) >> 2, HB = (Ho + 1) >> 1;
    const int npb = imgs * HB * WB, nblk = gridDim.x;
    const int chunk = (npb + nblk - 1) / nblk;
    const int pb = blockIdx.x * chunk, pe = min(npb, pb + chunk);
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = pb + pl; p < pe; p += P) {
        const int img = p / (HB * WB);
        const int rem = p - img * HB * WB;
        const int oh0 = (rem / WB) * 2, ow0 = (rem % WB) * 4;
        f32x4 d[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const T* dr = dy + ((size_t)(img * Ho + min(oh0 + r, Ho - 1)) * Wo) * C + cq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = oh0 + r < Ho && ow0 + j < Wo;
                d[r][j] = LD4Z(dr, ow0 + j, C, ok);
            }
        }
        const int iw0 = ow0 - PT;
#pragma unroll
        for (int ir = 0; ir <= K; ++ir) {
            const int ih = oh0 + ir - PT;
            const bool rv = (unsigned)ih < (unsigned)Hi;
            ROW_SKIP(rv);
            const T* xr = x + ((size_t)(img * Hi + (rv ? ih : 0)) * Wi) * C + cq * 4;
            f32x4 xin[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int iw = iw0 + j;
                const bool ok = rv && (unsigned)iw < (unsigned)Wi;
                xin[j] = LD4Z(xr, iw, C, ok);
            }
            if (ir < K) {                     // kernel row ir against the upper dy row
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ir * K + kw] += d[0][j] * xin[j + kw];
            }
            if (ir >= 1) {                    // kernel row ir-1 against the lower dy row
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[(ir - 1) * K + kw] += d[1][j] * xin[j + kw];
            }
            ROW_FENCE();
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        __syncthreads();
        red[threadIdx.x] = acc[t];
        __syncthreads();
        if (pl == 0) {
            f32x4 v = red[threadIdx.x];
            for (int k = 1; k < P; ++k) v += red[k * QT + threadIdx.x];
            st4(part + ((size_t)blockIdx.x * K * K + t) * C + cq * 4, v);
        }
    }
}
// ---- row-uniform weight gradient ---------------------------------------------------------------------------------------
// Same wave geometry as dw_rowu_kernel: a wave step = one (image, RB rows of dy) (RB = 2 at stride 1, 1 at stride 2),
// lanes = (4-column block of dy, channel quad); every lane keeps ALL K*K tap accumulators of its quad and walks the
// (RB-1)*S + K input rows of the step (row r+1 in flight while row r is multiplied; rows fenced so the scheduler cannot
// hoist the whole window).  A block's 4 waves take different steps of the same lanes; they are folded through LDS in a
// fixed order and written as ONE [K*K][64] record per block; dw_rowu_wgrad_reduce sums the records of a quad (all row
// groups x all column blocks) in a fixed order -> run-to-run deterministic, no atomics.
template <int K, int S, typename T>
__global__ __launch_bounds__(256) void dw_rowu_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            f32x4* __restrict__ part, int nsteps, int Hi, int Wi, int Ho, int Wo,
                                                            int C, int nchunk, int spb)
{
    constexpr int PT = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
    constexpr int RB = S == 1 ? 2 : 1;
    constexpr int NIN = 3 * S + K, NROW = (RB - 1) * S + K, ES = (int)sizeof(T);
    typedef typename RawOf<T>::type raw_t;
    __shared__ f32x4 red[3][K][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x % nchunk, rg = blockIdx.x / nchunk;
    const int Q = C >> 2, WB = (Wo + 3) >> 2, HB = (Ho + RB - 1) / RB;
    const int id = chunk * 64 + lane;
    const bool lv = id < WB * Q;
    const int owb = lv ? id / Q : 0, cq = lv ? id - owb * Q : 0;
    int voff[NIN], vd[4];
#pragma unroll
    for (int j = 0; j < NIN; ++j) voff[j] = lv ? ((owb * 4 * S - PT + j) * C + cq * 4) * ES : 0x7f000000;
#pragma unroll
    for (int j = 0; j < 4; ++j) vd[j] = lv ? ((owb * 4 + j) * C + cq * 4) * ES : 0x7f000000;
    const int in_row = Wi * C, out_row = Wo * C;
    const int s1 = min(nsteps, (rg + 1) * spb);
    f32x4 acc[K][K];
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) acc[kh][kw] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int st = rg * spb + wave; st < s1; st += 4) {
        const int img = st / HB, oh0 = (st - img * HB) * RB;
        const int ih0 = oh0 * S - PT;
        raw_t rd[RB][4], ra[NIN], rb[NIN];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const rsrc_t rs = row_rsrc(dy, img, oh0 + r, Ho, out_row);
#pragma unroll
            for (int j = 0; j < 4; ++j) raw_load(rs, vd[j], rd[r][j]);
        }
        {
            const rsrc_t rs = row_rsrc(x, img, ih0, Hi, in_row);
#pragma unroll
            for (int j = 0; j < NIN; ++j) raw_load(rs, voff[j], ra[j]);
        }
        f32x4 d[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[r][j] = raw_cvt(rd[r][j]);
#pragma unroll
        for (int ir = 0; ir < NROW; ++ir) {
            raw_t (&cur)[NIN] = (ir & 1) ? rb : ra;
            raw_t (&nxt)[NIN] = (ir & 1) ? ra : rb;
            if (ir + 1 < NROW) {
                const rsrc_t rs = row_rsrc(x, img, ih0 + ir + 1, Hi, in_row);
#pragma unroll
                for (int j = 0; j < NIN; ++j) raw_load(rs, voff[j], nxt[j]);
            }
            f32x4 xin[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) xin[j] = raw_cvt(cur[j]);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int kh = ir - r * S;                    // compile-time after unrolling
                if (kh < 0 || kh >= K) continue;
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[kh][kw] += d[r][j] * xin[j * S + kw];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // fold the 4 waves (fixed order) one kernel row at a time and store the block's record
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) red[wave - 1][kw][lane] = acc[kh][kw];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const f32x4 v = ((acc[kh][kw] + red[0][kw][lane]) + red[1][kw][lane]) + red[2][kw][lane];
                part[((size_t)blockIdx.x * K * K + kh * K + kw) * 64 + lane] = v;
            }
        }
    }
}
// out[t][cq] = sum over row groups and column blocks of the block records above.  Block = 16 quads x 16 split lanes.
__global__ __launch_bounds__(256) void dw_rowu_wgrad_reduce(const f32x4* __restrict__ part, float* __restrict__ out, int KK, int Q,
                                                            int WB, int nchunk, int nrg, int nsplit)
{
    // blockIdx.x = (16-quad block, split of the row groups), .y = record slot t, .z = statistics group;
    // out [z][split][KK][Q*4] (the weight gradient uses one group, one split)
    __shared__ f32x4 red[16][16];
    const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int qb = blockIdx.x / nsplit, sp = blockIdx.x - qb * nsplit;
    const int t = blockIdx.y, cq = qb * 16 + ql;
    part += (size_t)blockIdx.z * nrg * nchunk * KK * 64;
    out += ((size_t)blockIdx.z * nsplit + sp) * KK * Q * 4;
    const int rg0 = (int)((int64_t)nrg * sp / nsplit), rg1 = (int)((int64_t)nrg * (sp + 1) / nsplit);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (cq < Q) {
        // four records in flight per thread (four running sums, folded in a fixed order): one at a time this was a chain of
        // dependent L2 round trips -- 20 us per launch, 63 launches per step
        const int n = (rg1 - rg0) * WB;
        f32x4 b[4] = {a, a, a, a};
        auto rec = [&](int i) {
            const int rg = rg0 + i / WB, m = i % WB;
            const int id = cq + m * Q;
            return part[((size_t)(rg * nchunk + (id >> 6)) * KK + t) * 64 + (id & 63)];
        };
        int i = sl;
        for (; i + 48 < n; i += 64) {
            const f32x4 v0 = rec(i), v1 = rec(i + 16), v2 = rec(i + 32), v3 = rec(i + 48);
            b[0] += v0; b[1] += v1; b[2] += v2; b[3] += v3;
        }
        for (int u = 0; i < n; i += 16, ++u) b[u] += rec(i);
        a = (b[0] + b[1]) + (b[2] + b[3]);
    }
    red[sl][ql] = a;
    __syncthreads();
    if (sl == 0 && cq < Q) {
        f32x4 r = red[0][ql];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += red[k][ql];
        st4(out + ((size_t)t * Q + cq) * 4, r);
    }
}
template <int K, int S, typename T>
static void dw_rowu_wgrad_launch(const T* dy, const T* x, float* part, float* out, int imgs, int Hi, int Wi, int Ho, int Wo, int C,
                                 hipStream_t s)
{
    constexpr int RB = S == 1 ? 2 : 1;
    const int WB = (Wo + 3) / 4, HB = (Ho + RB - 1) / RB, Q = C / 4;
    const int nchunk = (WB * Q + 63) / 64, nsteps = imgs * HB;
    static const int tgt = fm_tune("FM_DW_WG_BLOCKS", 2048);
    int spb = std::max(8, (int)(((int64_t)nsteps * nchunk + tgt - 1) / tgt));
    spb = (spb + 3) / 4 * 4;
    const int nrg = (nsteps + spb - 1) / spb;
    hipLaunchKernelGGL((dw_rowu_wgrad_kernel<K, S, T>), dim3(nchunk * nrg), dim3(256), 0, s, dy, x, reinterpret_cast<f32x4*>(part),
                       nsteps, Hi, Wi, Ho, Wo, C, nchunk, spb);
    hipLaunchKernelGGL(dw_rowu_wgrad_reduce, dim3((Q + 15) / 16, K * K), dim3(256), 0, s, reinterpret_cast<const f32x4*>(part), out,
                       K * K, Q, WB, nchunk, nrg, 1);
}
// QT = channel quads per block (a divisor of Q, <= 256), P = pixel lanes
static inline void dw_map(int C, int& QT, int& P, int& ytiles)
{
    const int Q = C / 4;
    ytiles = (Q + 255) / 256;
    while (Q % ytiles) ++ytiles;
    QT = Q / ytiles;
    P = std::max(1, 256 / QT);
}
int dw_stats_tiles() { return DW_ST_SPLITS; }
int dw_wgrad_blocks(int npix) { return std::max(1, std::min(2048, npix / 128)); }
template <typename T>
static void dw_wgrad_t(const T* dy, const T* x, float* part, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                       int stride, int pad_t, int pad_l, hipStream_t s);
template <typename T>
static void dw_wgrad_full(const T* dy, const T* x, float* part, float* out, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                          int stride, int pad_t, int pad_l, hipStream_t s)
{
    // Measured per layer (ms, 1024 bf16 / 512 fp32 images): bf16 every layer 1.9-4x faster than the lane = (pixel lane,
    // quad) kernels below (block 3: 1.41 -> 0.41, block 9: 0.89 -> 0.22; sum 10.5 -> 4.1); fp32 gains on the 5x5 layers
    // (0.33 -> 0.17) and loses on the wide 3x3 ones (112x112x32: 0.35 -> 0.63, one row of lookahead is too little in
    // flight for 16-B loads), so fp32 3x3 keeps the older kernels.  FM_DW_WG_ROWU: 0 never / 1 this rule / 2 always.
    static const int mode = fm_tune("FM_DW_WG_ROWU", 1);
    const bool pick = mode == 2 || (mode == 1 && (sizeof(T) == 2 || K == 5));
    if (pick && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l) && Wi == stride * Wo && Hi == stride * Ho) {
        if (K == 3 && stride == 1) dw_rowu_wgrad_launch<3, 1, T>(dy, x, part, out, imgs, Hi, Wi, Ho, Wo, C, s);
        else if (K == 3) dw_rowu_wgrad_launch<3, 2, T>(dy, x, part, out, imgs, Hi, Wi, Ho, Wo, C, s);
        else if (stride == 1) dw_rowu_wgrad_launch<5, 1, T>(dy, x, part, out, imgs, Hi, Wi, Ho, Wo, C, s);
        else dw_rowu_wgrad_launch<5, 2, T>(dy, x, part, out, imgs, Hi, Wi, Ho, Wo, C, s);
        return;
    }
    dw_wgrad_t(dy, x, part, imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, s);
    k_reduce_slabs(part, out, dw_wgrad_blocks(imgs * Ho * Wo), (int64_t)K * K * C, s);
}
template <typename T>
static void dw_wgrad_t(const T* dy, const T* x, float* part, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                       int stride, int pad_t, int pad_l, hipStream_t s)
{
    int QT, P, yt;
    dw_map(C, QT, P, yt);
    const dim3 grid(dw_wgrad_blocks(imgs * Ho * Wo), yt), blk(QT * P);
    // (reached by the fp32 3x3 layers only -- dw_wgrad_full -- and by paddings / sizes the row-uniform kernels do not take)
    if (K == 3 && stride == 1 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l)) {
        hipLaunchKernelGGL((dw_wgrad_blk2_kernel<3, T>), grid, blk, 0, s, dy, x, part, imgs, Hi, Wi, Ho, Wo, C, QT, P);
        return;
    }
    if (K == 3 && dw_blk_ok(K, stride, Hi, Wi, pad_t, pad_l)) {
        hipLaunchKernelGGL((dw_wgrad_blk_kernel<3, 2, T>), grid, blk, 0, s, dy, x, part, imgs, Hi, Wi, Ho, Wo, C, QT, P);
        return;
    }
    if (K == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, T>), grid, blk, 0, s, dy, x, part, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l, QT, P);
    else hipLaunchKernelGGL((dw_wgrad_kernel<5, T>), grid, blk, 0, s, dy, x, part, imgs, Hi, Wi, Ho, Wo, C, stride, pad_t, pad_l, QT, P);
}
void k_dw_wgrad(const void* dy, const void* x, int dt, float* part, float* out, int imgs, int Hi, int Wi, int Ho, int Wo, int C,
                int K, int stride, int pad_t, int pad_l, hipStream_t s)
{
    if (dt == DT_F32) dw_wgrad_full(cp<float>(dy), cp<float>(x), part, out, imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, s);
    else dw_wgrad_full(cp<bf16>(dy), cp<bf16>(x), part, out, imgs, Hi, Wi, Ho, Wo, C, K, stride, pad_t, pad_l, s);
}

// per-image channel sums over a chunk of pixels: part[img][chunk][C] = sum_p a[p][c] (* b[p][c]).
// `tsel` 1 / 2: operand a / b is a raw BN input and is read as swish(v*scale[g]+shift[g]), g = img/ipg
// (the post-BN activation is never materialised).  One 16-B piece per thread and tensor (NV quads).
template <typename T>
__global__ __launch_bounds__(256) void chan_pool_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        float* __restrict__ part, int HW, int C, int QT, int P,
                                                        int tsel, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int ipg)
{
    constexpr int NV = VecOf<T>::NV;
    __shared__ f32x4 red[NV][256];
    const int img = blockIdx.y, nch = gridDim.x;
    const int Q = C / (4 * NV);
    const int cq0 = threadIdx.x % QT, pl = threadIdx.x / QT;
    const int chunk = (HW + nch - 1) / nch;
    const int pb = blockIdx.x * chunk, pe = min(HW, pb + chunk);
    const int g = tsel ? img / ipg : 0;
    for (int cq = cq0; cq < Q; cq += QT) {
        const int c0 = cq * 4 * NV;
        f32x4 s1[NV], sc[NV], sh[NV];
#pragma unroll
        for (int h = 0; h < NV; ++h) { s1[h] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[h] = f32x4{1.f, 1.f, 1.f, 1.f}; sh[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (tsel) {
            ldf<NV>(scale + g * C + c0, sc);
            ldf<NV>(shift + g * C + c0, sh);
        }
#pragma unroll 4
        for (int p = pb + pl; p < pe; p += P) {
            const size_t o = ((size_t)img * HW + p) * C + c0;
            f32x4 v[NV];
            ldv<NV>(a + o, v);
            if (tsel == 1) {
#pragma unroll
                for (int h = 0; h < NV; ++h) v[h] = act_fwd<NV == 2>(v[h] * sc[h] + sh[h], 2);
            }
            if (b) {
                f32x4 w[NV];
                ldv<NV>(b + o, w);
#pragma unroll
                for (int h = 0; h < NV; ++h) {
                    if (tsel == 2) w[h] = act_fwd<NV == 2>(w[h] * sc[h] + sh[h], 2);
                    v[h] = v[h] * w[h];
                }
            }
#pragma unroll
            for (int h = 0; h < NV; ++h) s1[h] += v[h];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < NV; ++h) red[h][threadIdx.x] = s1[h];
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                for (int k = 1; k < P; ++k) s1[h] += red[h][k * QT + cq0];
                st4(part + ((size_t)img * nch + blockIdx.x) * C + c0 + 4 * h, s1[h]);
            }
        }
    }
}
int chan_pool_chunks(int HW) { return std::max(1, std::min(16, HW / 64)); }
void k_chan_pool(const void* a, const void* b, int dt, float* part, int imgs, int HW, int C, int tsel, const float* scale,
                 const float* shift, int ipg, hipStream_t s)
{
    int QT, P, yt;
    dw_map(dt == DT_BF16 ? C / 2 : C, QT, P, yt);          // 16-B pieces per pixel: C/4 (fp32) or C/8 (bf16)
    const dim3 grid(chan_pool_chunks(HW), imgs), blk(QT * P);
    if (dt == DT_F32)
        hipLaunchKernelGGL((chan_pool_kernel<float>), grid, blk, 0, s, cp<float>(a), cp<float>(b), part, HW, C, QT, P, tsel, scale, shift, ipg);
    else
        hipLaunchKernelGGL((chan_pool_kernel<bf16>), grid, blk, 0, s, cp<bf16>(a), cp<bf16>(b), part, HW, C, QT, P, tsel, scale, shift, ipg);
}

// ------------------------------------------------------------ squeeze-excite ---
// one block per image: s = mean_hw(a) (from chan_pool partials [img][nch][C]); r_pre = W1 s + b1;
// g = sigmoid(W2 swish(r_pre) + b2).  W1 [Cs][C], W2 stored transposed [Cs][C] (lanes = channels: coalesced).  Stores s [imgs][C], r_pre [imgs][Cs], g [imgs][C].
__global__ void se_fwd_kernel(const float* __restrict__ pool, int nch, const float* __restrict__ W1,
                              const float* __restrict__ b1, const float* __restrict__ W2,
                              const float* __restrict__ b2, float* __restrict__ sq, float* __restrict__ rpre,
                              float* __restrict__ gate, int HW, int C, int Cs)
{
    extern __shared__ float sm[];            // s[C] then r[Cs]
    float* s_ = sm;
    float* r_ = sm + C;
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.f / (float)HW;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
#pragma unroll 4
        for (int k = 0; k < nch; ++k) t += pool[((size_t)img * nch + k) * C + c];     // (loads of 4 chunks in flight; same order of sums)
        t *= inv;
        s_[c] = t;
        sq[(size_t)img * C + c] = t;
    }
    __syncthreads();
    // four squeezed channels per wave at a time: their weight rows are in flight together and lanes 0-3 finish one each (the
    // one-row-at-a-time form was a chain of up to 12 dependent L2 round trips per image: 43 us per launch, 1.4 ms per step);
    // per channel the sums are formed in the same order as before
    // (round 4: the `j0 + u < Cs` test inside the loop made every one of the four loads its own load -> s_waitcnt vmcnt(0) -> FMA
    // sequence, 4 x C/64 x Cs/16 dependent L2 round trips per image: 44 us per launch.  Rows past Cs are clamped instead -- their
    // sums are never used -- so the loop body is branch-free and the loads of three iterations are in flight together.)
    for (int j0 = 4 * wave; j0 < Cs; j0 += 16) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[u] = W1 + (size_t)min(j0 + u, Cs - 1) * C;
#pragma unroll 3
        for (int c = lane; c < C; c += 64) {
            const float sv = s_[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += wr[u][c] * sv;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t[u] += __shfl_xor(t[u], d);
        const int j = j0 + lane;
        if (lane < 4 && j < Cs) {
            float tv = lane == 0 ? t[0] : (lane == 1 ? t[1] : (lane == 2 ? t[2] : t[3]));
            tv += b1[j];
            rpre[(size_t)img * Cs + j] = tv;
            r_[j] = tv * sigm(tv);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = b2[c];
#pragma unroll 8
        for (int j = 0; j < Cs; ++j) t += W2[(size_t)j * C + c] * r_[j];
        gate[(size_t)img * C + c] = sigm(t);
    }
}
// pooled: pool_ws [imgs][C] already holds the per-image channel sums (left by the eval-mode depthwise forward)
void k_se_fwd(const void* a, int dt, const float* scale, const float* shift, int ipg, float* pool_ws, const float* W1,
              const float* b1, const float* W2, const float* b2, float* sq, float* rpre, float* gate, int imgs, int HW,
              int C, int Cs, hipStream_t s, bool pooled)
{
    if (!pooled) k_chan_pool(a, nullptr, dt, pool_ws, imgs, HW, C, scale ? 1 : 0, scale, shift, ipg, s);
    hipLaunchKernelGGL(se_fwd_kernel, dim3(imgs), dim3(256), (C + Cs) * sizeof(float), s, pool_ws, pooled ? 1 : chan_pool_chunks(HW),
                       W1, b1, W2, b2, sq, rpre, gate, HW, C, Cs);
}

// out = A * gate[img][c], A = a or (scale != null) swish(a*scale[g]+shift[g]) with g = img/ipg
template <typename T>
__global__ void se_scale_kernel(const T* __restrict__ a, const float* __restrict__ gate, T* __restrict__ out,
                                int64_t nq, int HW, int C, const float* __restrict__ scale,
                                const float* __restrict__ shift, int ipg)
{
    constexpr int NV = VecOf<T>::NV;
    const int Q = C / (4 * NV);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int c0 = (int)(i % Q) * 4 * NV;
    const int64_t img = i / Q / HW;
    f32x4 v[NV], gt[NV];
    ldv<NV>(a + i * (4 * NV), v);
    if (scale) {
        const int g = (int)(img / ipg);
        f32x4 sc[NV], sh[NV];
        ldf<NV>(scale + g * C + c0, sc);
        ldf<NV>(shift + g * C + c0, sh);
#pragma unroll
        for (int h = 0; h < NV; ++h) v[h] = act_fwd<NV == 2>(v[h] * sc[h] + sh[h], 2);
    }
    ldf<NV>(gate + img * C + c0, gt);
#pragma unroll
    for (int h = 0; h < NV; ++h) v[h] = v[h] * gt[h];
    stv<NV>(out + i * (4 * NV), v);
}
void k_se_scale(const void* a, int dt, const float* scale, const float* shift, int ipg, const float* gate, void* out, int imgs,
                int HW, int C, hipStream_t s)
{
    const int64_t nq = (int64_t)imgs * HW * (C / (dt == DT_BF16 ? 8 : 4));
    const dim3 grid(cdiv(nq, 256));
    if (dt == DT_F32)
        hipLaunchKernelGGL((se_scale_kernel<float>), grid, dim3(256), 0, s, cp<float>(a), gate, mp<float>(out), nq, HW, C, scale, shift, ipg);
    else
        hipLaunchKernelGGL((se_scale_kernel<bf16>), grid, dim3(256), 0, s, cp<bf16>(a), gate, mp<bf16>(out), nq, HW, C, scale, shift, ipg);
}

// backward, one block per image:  dgs[c] = sum_hw dout*a ; dgp = dgs*g(1-g) ; dr = W2^T dgp ;
// drp = dr*swish'(r_pre) ; ds = W1^T drp.  Stores dgp [imgs][C], drp [imgs][Cs], ds [imgs][C].
__global__ void se_bwd_kernel(const float* __restrict__ pool, int nch, int pstride,
                              const float* __restrict__ gate, const float* __restrict__ rpre,
                              const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ dgp,
                              float* __restrict__ drp, float* __restrict__ ds, int HW, int C, int Cs)
{
    extern __shared__ float sm[];            // dgp[C] then drp[Cs]
    float* g_ = sm;
    float* r_ = sm + C;
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
#pragma unroll 4
        for (int k = 0; k < nch; ++k) t += pool[((size_t)img * nch + k) * pstride + c];
        const float g = gate[(size_t)img * C + c];
        t *= g * (1.f - g);
        g_[c] = t;
        dgp[(size_t)img * C + c] = t;
    }
    __syncthreads();
    for (int j0 = 4 * wave; j0 < Cs; j0 += 16) {          // four squeezed channels per wave at a time, as in se_fwd_kernel
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[u] = W2 + (size_t)min(j0 + u, Cs - 1) * C;     // rows past Cs clamped: branch-free loop body
#pragma unroll 3
        for (int c = lane; c < C; c += 64) {
            const float gv = g_[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += wr[u][c] * gv;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t[u] += __shfl_xor(t[u], d);
        const int j = j0 + lane;
        if (lane < 4 && j < Cs) {
            float tv = lane == 0 ? t[0] : (lane == 1 ? t[1] : (lane == 2 ? t[2] : t[3]));
            tv *= swish_grad(rpre[(size_t)img * Cs + j]);
            r_[j] = tv;
            drp[(size_t)img * Cs + j] = tv;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
#pragma unroll 8
        for (int j = 0; j < Cs; ++j) t += W1[(size_t)j * C + c] * r_[j];
        ds[(size_t)img * C + c] = t;
    }
}
void k_se_bwd(const void* dout, const void* a, int dt, const float* scale, const float* shift, int ipg, float* pool_ws,
              const float* gate, const float* rpre, const float* W1, const float* W2, float* dgp, float* drp, float* ds,
              int imgs, int HW, int C, int Cs, hipStream_t s)
{
    k_chan_pool(dout, a, dt, pool_ws, imgs, HW, C, scale ? 2 : 0, scale, shift, ipg, s);
    hipLaunchKernelGGL(se_bwd_kernel, dim3(imgs), dim3(256), (C + Cs) * sizeof(float), s, pool_ws, chan_pool_chunks(HW), C,
                       gate, rpre, W1, W2, dgp, drp, ds, HW, C, Cs);
}

// ---- squeeze-excite backward + BN1 backward sums from ONE pass over (d a_s, y_d) ------------------------
// The BN1 backward needs S1 = sum dyh and S2 = sum dyh*xh over the batch with dyh = (d*gate + ds/HW)*sg
// (d = d a_s, sg = swish'(v), v = y*scale+shift, xh = (y-mean)*istd), but ds itself comes out of the squeeze-excite
// backward, which needs the per-image sums of d*a_d first.  gate and ds are constant over an image, so
//   S1 = sum_img gate*P1 + ds/HW*Q1,  S2 = sum_img gate*P2 + ds/HW*Q2
// with per-image sums P1 = sum d*sg, P2 = sum d*sg*xh, Q1 = sum sg, Q2 = sum sg*xh.  All five per-image sums
// (R = sum d*a_d for the squeeze-excite backward included) are taken in the single pass below; the separate
// reduction pass of the BN1 backward (two more reads of the block's largest gradient tensors) is gone.
// part layout [img][chunk][5][C].
template <typename T>
__global__ __launch_bounds__(256) void chan_pool5_kernel(const T* __restrict__ d, const T* __restrict__ y,
                                                         float* __restrict__ part, int HW, int C, int QT, int P,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ istd,
                                                         int ipg)
{
    constexpr int NV = VecOf<T>::NV;
    constexpr bool FAST = NV == 2;
    __shared__ f32x4 red[5][NV][256];
    const int img = blockIdx.y, nch = gridDim.x;
    const int Q = C / (4 * NV);
    const int cq0 = threadIdx.x % QT, pl = threadIdx.x / QT;
    const int chunk = (HW + nch - 1) / nch;
    const int pb = blockIdx.x * chunk, pe = min(HW, pb + chunk);
    const int g = img / ipg;
    for (int cq = cq0; cq < Q; cq += QT) {
        const int c0 = cq * 4 * NV;
        f32x4 acc[5][NV], sc[NV], sh[NV], mu[NV], is[NV];
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int h = 0; h < NV; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        ldf<NV>(scale + g * C + c0, sc);
        ldf<NV>(shift + g * C + c0, sh);
        ldf<NV>(mean + g * C + c0, mu);
        ldf<NV>(istd + g * C + c0, is);
#pragma unroll 2
        for (int p = pb + pl; p < pe; p += P) {
            const size_t o = ((size_t)img * HW + p) * C + c0;
            f32x4 dd[NV], yy[NV];
            ldv<NV>(d + o, dd);
            ldv<NV>(y + o, yy);
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                const f32x4 v = yy[h] * sc[h] + sh[h];
                const f32x4 xh = (yy[h] - mu[h]) * is[h];
                f32x4 ad, sg;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sgm = sigm_t<FAST>(v[k]);
                    ad[k] = v[k] * sgm;
                    sg[k] = sgm * (1.f + v[k] * (1.f - sgm));
                }
                const f32x4 dsg = dd[h] * sg;
                acc[0][h] += dd[h] * ad;
                acc[1][h] += dsg;
                acc[2][h] += dsg * xh;
                acc[3][h] += sg;
                acc[4][h] += sg * xh;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int h = 0; h < NV; ++h) red[t][h][threadIdx.x] = acc[t][h];
        __syncthreads();
        // the P pixel lanes of a (sum, piece) pair are folded by ONE thread each: 5 NV QT folds spread over the block
        for (int j = threadIdx.x; j < 5 * NV * QT; j += blockDim.x) {
            const int t = j / (NV * QT), rem = j - t * NV * QT, h = rem / QT, q = rem - h * QT;
            f32x4 v = red[t][h][q];
            for (int k = 1; k < P; ++k) v += red[t][h][k * QT + q];
            st4(part + (((size_t)img * nch + blockIdx.x) * 5 + t) * C + (cq - cq0 + q) * 4 * NV + 4 * h, v);
        }
    }
}
// BN1-backward sums from the per-image partials: part_out[g][split][2][C]; block = (64 channels, group, split of the
// group's images), 4 image lanes.  (One block per (64 channels, group) -- 2 to 36 blocks on 256 CUs -- took 2.8 ms per
// step over the 16 blocks; the splits are folded by the BN-backward finalize like any other per-block partials.)
int se_bwd_bn1_splits(int ipg) { return std::max(1, std::min(32, ipg / 16)); }
__global__ void bn1_sums_kernel(const float* __restrict__ pool5, int nch, const float* __restrict__ gate,
                                const float* __restrict__ ds, float* __restrict__ out, int HW, int C, int ipg)
{
    __shared__ float red[2][4][64];
    const int g = blockIdx.y, cl = threadIdx.x & 63, il = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int nsplit = gridDim.z, sp = blockIdx.z;
    const int ips = (ipg + nsplit - 1) / nsplit;
    const int i0 = sp * ips, i1 = min(ipg, i0 + ips);
    const float inv = 1.f / (float)HW;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
        for (int i = i0 + il; i < i1; i += 4) {
            const size_t img = (size_t)g * ipg + i;
            float p1 = 0.f, p2 = 0.f, q1 = 0.f, q2 = 0.f;
            for (int k = 0; k < nch; ++k) {
                const float* pp = pool5 + ((img * nch + k) * 5) * C + c;
                p1 += pp[C]; p2 += pp[2 * C]; q1 += pp[3 * C]; q2 += pp[4 * C];
            }
            const float gt = gate[img * C + c], dv = ds[img * C + c] * inv;
            s1 += gt * p1 + dv * q1;
            s2 += gt * p2 + dv * q2;
        }
    red[0][il][cl] = s1; red[1][il][cl] = s2;
    __syncthreads();
    if (il == 0 && c < C) {
        float* o = out + ((size_t)g * nsplit + sp) * 2 * C;
        o[c] = ((red[0][0][cl] + red[0][1][cl]) + red[0][2][cl]) + red[0][3][cl];
        o[C + c] = ((red[1][0][cl] + red[1][1][cl]) + red[1][2][cl]) + red[1][3][cl];
    }
}
// pool_ws: [imgs][chunks][5][C]; writes dgp/drp/ds like k_se_bwd and the BN1-backward sums to
// bn_part [groups][se_bwd_bn1_splits(ipg)][2][C]
void k_se_bwd_bn1(const void* dout, const void* y, int dt, const float* scale, const float* shift, const float* mean,
                  const float* istd, int ipg, float* pool_ws, const float* gate, const float* rpre, const float* W1,
                  const float* W2, float* dgp, float* drp, float* ds, float* bn_part, int imgs, int HW, int C, int Cs,
                  hipStream_t s, int nch_ready)
{
    int QT, P, yt;
    dw_map(dt == DT_BF16 ? C / 2 : C, QT, P, yt);
    // fewer, longer pixel chunks than the forward pooling: the five-way fold at the end of a block is amortised over
    // >= 512 pixels, and imgs x nch blocks still fill the chip
    // nch_ready > 0: pool_ws already holds that many records per image (pw_proj_bwd_kernel phase 0 took the sums)
    const int nch = nch_ready > 0 ? nch_ready : std::max(1, std::min(chan_pool_chunks(HW), std::max(HW / 512, imgs >= 512 ? 1 : 2)));
    const dim3 grid(nch, imgs), blk(QT * P);
    if (nch_ready > 0) {}
    else if (dt == DT_F32)
        hipLaunchKernelGGL((chan_pool5_kernel<float>), grid, blk, 0, s, cp<float>(dout), cp<float>(y), pool_ws, HW, C, QT, P, scale,
                           shift, mean, istd, ipg);
    else
        hipLaunchKernelGGL((chan_pool5_kernel<bf16>), grid, blk, 0, s, cp<bf16>(dout), cp<bf16>(y), pool_ws, HW, C, QT, P, scale,
                           shift, mean, istd, ipg);
    hipLaunchKernelGGL(se_bwd_kernel, dim3(imgs), dim3(256), (C + Cs) * sizeof(float), s, pool_ws, nch, 5 * C, gate, rpre, W1,
                       W2, dgp, drp, ds, HW, C, Cs);
    hipLaunchKernelGGL(bn1_sums_kernel, dim3(cdiv(C, 64), imgs / ipg, se_bwd_bn1_splits(ipg)), dim3(256), 0, s, pool_ws, nch, gate,
                       ds, bn_part, HW, C, ipg);
}

// dW2^T[j][c] = sum_img dgp[img][c]*swish(rpre[img][j]); db2[c]; dW1[j][c] = sum_img drp[img][j]*s[img][c]; db1[j]
// Block = 64 channels x 4 lanes of squeezed channels j, one of SE_SPLITS image ranges; the per-image squeezed vectors
// (swish(rpre), drp) of 16 images at a time sit in LDS (a wave reads one address: broadcast), the per-channel operands are
// coalesced 256-B rows.  Each split writes a slab laid out like the four gradient tensors themselves
// ([cs][C] | b1 padded to 4 | [cs][C] (W2 is stored transposed) | [C]: they are contiguous in the arena), summed by k_reduce_slabs in a fixed order.
// (The earlier form -- 16 lanes per (c, j) pair striding over the images -- took 143 us per block, 2.3 ms per step.)
constexpr int SE_SPLITS = 16, SE_TI = 16, SE_NJ = 12;          // squeezed channels <= 48
__global__ __launch_bounds__(256) void se_wgrad_part_kernel(const float* __restrict__ dgp, const float* __restrict__ drp,
                                                            const float* __restrict__ rpre, const float* __restrict__ sq,
                                                            float* __restrict__ part, int imgs, int C, int Cs, int n)
{
    __shared__ float sw[SE_TI][48], dd[SE_TI][48];
    const int t = threadIdx.x, cl = t & 63, jl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cv = c < C;
    const int ips = (imgs + SE_SPLITS - 1) / SE_SPLITS;
    const int i0 = blockIdx.y * ips, i1 = min(imgs, i0 + ips);
    const int cs4 = (Cs + 3) / 4 * 4;
    float acc2[SE_NJ], acc1[SE_NJ], b2 = 0.f;
#pragma unroll
    for (int k = 0; k < SE_NJ; ++k) { acc2[k] = 0.f; acc1[k] = 0.f; }
    for (int it = i0; it < i1; it += SE_TI) {
        __syncthreads();
        for (int idx = t; idx < SE_TI * Cs; idx += 256) {
            const int ii = idx / Cs, j = idx - ii * Cs;
            const int img = it + ii;
            float rp = 0.f, d = 0.f;
            if (img < i1) { rp = rpre[(size_t)img * Cs + j]; d = drp[(size_t)img * Cs + j]; }
            sw[ii][j] = rp * sigm(rp);
            dd[ii][j] = d;
        }
        __syncthreads();
        const int ni = min(SE_TI, i1 - it);
        // four images per pass: their eight loads are issued together (one image at a time was a chain of dependent L2 round
        // trips); images past the range contribute exact zeros (g = q = 0 against the zero-staged sw / dd rows), sums in image order
        for (int ii = 0; ii < ni; ii += 4) {
            float g[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = cv && ii + u < ni;
                const size_t img = (size_t)(it + (ok ? ii + u : 0));
                g[u] = ok ? dgp[img * C + c] : 0.f;
                q[u] = ok ? sq[img * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b2 += g[u];
#pragma unroll
                for (int k = 0; k < SE_NJ; ++k) {
                    const int j = jl + 4 * k;
                    if (j < Cs) {
                        acc2[k] += g[u] * sw[ii + u][j];
                        acc1[k] += dd[ii + u][j] * q[u];
                    }
                }
            }
        }
    }
    float* slab = part + (size_t)blockIdx.y * n;
    const size_t o_b1 = (size_t)Cs * C, o_w2 = o_b1 + cs4, o_b2 = o_w2 + (size_t)C * Cs;
    if (cv) {
#pragma unroll
        for (int k = 0; k < SE_NJ; ++k) {
            const int j = jl + 4 * k;
            if (j < Cs) {
                slab[o_w2 + (size_t)j * C + c] = acc2[k];
                slab[(size_t)j * C + c] = acc1[k];
            }
        }
        if (jl == 0) slab[o_b2 + c] = b2;
    }
    if (blockIdx.x == 0 && t < cs4) {                        // db1 (and its padding, written as zeros)
        float v = 0.f;
        if (t < Cs) {
#pragma unroll 8
            for (int img = i0; img < i1; ++img) v += drp[(size_t)img * Cs + t];
        }
        slab[o_b1 + t] = v;
    }
}
// dW1 .. db2 are the contiguous arena range starting at dW1; part = workspace for SE_SPLITS slabs of that range
void k_se_wgrad(const float* dgp, const float* drp, const float* rpre, const float* sq, float* part, float* dW1, int imgs, int C,
                int Cs, hipStream_t s)
{
    const int n = Cs * C + (Cs + 3) / 4 * 4 + C * Cs + C;
    hipLaunchKernelGGL(se_wgrad_part_kernel, dim3(cdiv(C, 64), SE_SPLITS), dim3(256), 0, s, dgp, drp, rpre, sq, part, imgs, C, Cs, n);
    k_reduce_slabs(part, dW1, SE_SPLITS, n, s);
}

// y = a * b, y += a (elementwise helpers: dropout on the feature, residual-gradient accumulation)
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i];
}
void k_mul(const float* a, const float* b, float* y, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(mul_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, a, b, y, n);
}
template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ y, const T* __restrict__ a, int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st4(y + 4 * i, ld4(y + 4 * i) + ld4(a + 4 * i));
}
void k_add_inplace(void* y, const void* a, int dt, int64_t n, hipStream_t s)
{
    const dim3 grid(cdiv(n / 4, 256));
    if (dt == DT_F32) hipLaunchKernelGGL((add_inplace_kernel<float>), grid, dim3(256), 0, s, mp<float>(y), cp<float>(a), n / 4);
    else hipLaunchKernelGGL((add_inplace_kernel<bf16>), grid, dim3(256), 0, s, mp<bf16>(y), cp<bf16>(a), n / 4);
}
