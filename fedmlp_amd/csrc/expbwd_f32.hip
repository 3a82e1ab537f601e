// Fused backward of an expand (1x1) convolution with its BatchNorm + Swish, fp32 storage (BASELINE configs[3]), gfx950: the
// fp32 twin of pw_exp_bwd_kernel (pwconv_bf16.hip) for the early high-resolution MBConv blocks 1-3 (ce = L in {96, 144},
// cin = S <= 32: 62 % of the network's expanded-tensor bytes).
//
// Reference ops replaced: inside loss.backward() (utils/local_training.py:674, 965, 1191 through efficientnet_pytorch 0.7.1's
// MBConvBlock, model/efficientnet.py:28-33) the backward of  a_e = swish(bn0(expand_conv(x))):  BN0-backward apply, the conv's
// weight gradient and its data gradient (+ the skip connection's gradient).
//
// The unfused order moves the expanded gradient five times (BN-backward apply reads d a_e and y_e and writes d y_e, the weight
// gradient reads d y_e, the data gradient reads d y_e); here d a_e and y_e are read ONCE: a wave owns 16-pixel tiles, forms
//     d y_e = ca * (d a_e * swish'(y_e * sc + sh)) + cb * y_e + cc        (bnact_bwd_apply_kernel's arithmetic)
// in registers, writes it to a wave-private LDS tile and feeds two fp32-MFMA products (v_mfma_f32_16x16x4_f32) from it:
//     dW[l][s] += sum_pix dy[pix][l] x[pix][s]      (k = pixels)         dX[pix][s] = sum_l dy[pix][l] W[l][s] (+ residual)   (k = l)
// LDS rows are 4 x odd floats long, so the 16 rows x 4 k-lanes of a fragment read fall on 64 different banks.
// Roofline: HBM (288 MFMAs of 32 cycles per 18 KB tile pair: a quarter of the time the bytes take at 5 TB/s).
#include <stdlib.h>

#include <algorithm>

#include "pwconv.h"

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct ExpBwdF32Args {
    const float *dA, *Ye, *X, *W, *res;      // [npix][L], [npix][L], [npix][S], conv weight [L][S], [npix][S] or null
    float* dX;                               // [npix][S]
    float* slab;                             // [waves][L][S] partial dW
    const float *ca, *cb, *cc, *sc, *sh;     // [groups][L]
    int npix, pix_per_group, groups;
};

template <int NRT, int CC>
__global__ __launch_bounds__(256, 2) void pw_exp_bwd_f32_kernel(const ExpBwdF32Args p)
{
    extern __shared__ __attribute__((aligned(16))) float sme[];
    constexpr int L = 16 * NRT, S = 16 * CC, TP = 16;
    constexpr int SB = L + 4, SS = S + 4;                    // LDS row lengths in floats (4 x odd)
    constexpr int CPB = L / 4, CPS = S / 4;                  // 16-B chunks per row
    constexpr int NBG = (TP * CPB + 63) / 64, NSM = (TP * CPS + 63) / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    float* wt = sme;                                         // [S][SB]: W^T
    float* vec = sme + S * SB;                               // [groups][5][L]: ca, cb, cc, sc, sh
    float* dyt = vec + (size_t)p.groups * 5 * L + (size_t)wave * TP * (SB + SS);      // [TP][SB]
    float* xt = dyt + TP * SB;                               // [TP][SS]
    for (int i = tid; i < L * S; i += 256) {
        const int l = i / S, s = i - l * S;
        wt[s * SB + l] = p.W[i];
    }
    for (int i = tid; i < p.groups * L; i += 256) {
        const int g = i / L, l = i - g * L;
        float* v = vec + (size_t)g * 5 * L;
        v[l] = p.ca[i]; v[L + l] = p.cb[i]; v[2 * L + l] = p.cc[i]; v[3 * L + l] = p.sc[i]; v[4 * L + l] = p.sh[i];
    }
    __syncthreads();
    const int ws = blockIdx.x * 4 + wave, nws = gridDim.x * 4;
    const int tsteps = p.npix / TP;                          // npix % 16 == 0 (launcher)
    const int nsteps = ws < tsteps ? (tsteps - ws + nws - 1) / nws : 0;
    f32x4 va[NBG], vy[NBG], vs[NSM];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int st) {
        const size_t pb = (size_t)(ws + st * nws) * TP;
        const f32x4* ta = reinterpret_cast<const f32x4*>(p.dA + pb * L) + lane;
        const f32x4* ty = reinterpret_cast<const f32x4*>(p.Ye + pb * L) + lane;
        const f32x4* tx = reinterpret_cast<const f32x4*>(p.X + pb * S) + lane;
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            if (64 * q + 63 < TP * CPB || lane + 64 * q < TP * CPB) { va[q] = ta[64 * q]; vy[q] = ty[64 * q]; }
            else { va[q] = z; vy[q] = z; }
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            if (64 * q + 63 < TP * CPS || lane + 64 * q < TP * CPS) vs[q] = tx[64 * q];
            else vs[q] = z;
        }
    };
    auto lstore = [&](int st) {
        const int pb = (ws + st * nws) * TP;
        const float* vg = vec + (size_t)(pb / p.pix_per_group) * 5 * L;          // a tile lies inside one statistics group
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPB, cb = c - row * CPB;
            if (c < TP * CPB) {
                const float* v = vg + 4 * cb;
                const f32x4 y = vy[q];
                const f32x4 u = y * ld4(v + 3 * L) + ld4(v + 4 * L);
                f32x4 d = va[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-u[k]));          // FM_F32_FAST_SWISH forms (effnet.hip)
                    d[k] *= sg * (1.f + u[k] * (1.f - sg));
                }
                *reinterpret_cast<f32x4*>(dyt + row * SB + 4 * cb) = ld4(v) * d + ld4(v + L) * y + ld4(v + 2 * L);
            }
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPS, cb = c - row * CPS;
            if (c < TP * CPS) *reinterpret_cast<f32x4*>(xt + row * SS + 4 * cb) = vs[q];
        }
    };
    f32x4 acc[NRT][CC];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[r][c] = z;
    if (nsteps > 0) { gload(0); lstore(0); }
    for (int st = 0; st < nsteps; ++st) {
        __builtin_amdgcn_wave_barrier();
        if (st + 1 < nsteps) gload(st + 1);
        // ---- weight gradient: dW[l][s] += sum_pix dy[pix][l] x[pix][s], 4 pixels per MFMA ----
#pragma unroll
        for (int kk = 0; kk < TP / 4; ++kk) {
            float b[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) b[c] = xt[(4 * kk + lg) * SS + 16 * c + li];
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                const float a = dyt[(4 * kk + lg) * SB + 16 * r + li];
#pragma unroll
                for (int c = 0; c < CC; ++c) acc[r][c] = mfma4(a, b[c], acc[r][c]);
            }
        }
        // ---- data gradient: D[s][pix] = sum_l W^T[s][l] dy[pix][l]; lane (li, lg) ends with 4 consecutive s of pixel li ----
        const int pb = (ws + st * nws) * TP;
        f32x4 dx[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) dx[c] = z;
#pragma unroll 4
        for (int kk = 0; kk < L / 4; ++kk) {
            const float b = dyt[li * SB + 4 * kk + lg];
#pragma unroll
            for (int c = 0; c < CC; ++c) dx[c] = mfma4(wt[(16 * c + li) * SB + 4 * kk + lg], b, dx[c]);
        }
        const size_t o0 = ((size_t)pb + li) * S + 4 * lg;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            f32x4 v = dx[c];
            if (p.res) v += ld4(p.res + o0 + 16 * c);
            *reinterpret_cast<f32x4*>(p.dX + o0 + 16 * c) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (st + 1 < nsteps) lstore(st + 1);
    }
    // acc[r][c][q] = dW[l = 16 r + 4 lg + q][s = 16 c + li]; this wave's slab (zeros when it had no tile)
    float* slab = p.slab + (size_t)ws * L * S;
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) slab[(size_t)(16 * r + 4 * lg + q) * S + 16 * c + li] = acc[r][c][q];
}

}  // namespace

// Returns the number of [L][S] slabs written (reduce with k_reduce_slabs), 0 = shape not handled.
int launch_pw_exp_bwd_f32(const PwExpBwdF32Params& w, size_t slab_floats, hipStream_t s)
{
    const int nrt = w.L / 16, cc = w.S / 16;
    const bool shape = (nrt == 6 || nrt == 9) && (cc == 1 || cc == 2);
    if (!shape || (w.L & 15) || (w.S & 15) || w.pix_per_group % 16 != 0 || w.npix % 16 != 0 || w.groups < 1 || w.groups > 2) return 0;
    static const int on = fm_tune("FM_PW_EXP_BWD_F32", 1);
    if (!on) return 0;
    ExpBwdF32Args a{};
    a.dA = w.dA; a.Ye = w.Ye; a.X = w.X; a.W = w.W; a.res = w.res; a.dX = w.dX; a.slab = w.slab;
    a.ca = w.ca; a.cb = w.cb; a.cc = w.cc; a.sc = w.sc; a.sh = w.sh;
    a.npix = w.npix; a.pix_per_group = w.pix_per_group; a.groups = w.groups;
    const int tsteps = w.npix / 16;
    int nblk = std::max(1, std::min(512, tsteps / 32));
    nblk = (int)std::min<size_t>(nblk, std::max<size_t>(1, slab_floats / ((size_t)4 * w.L * w.S)));
    const size_t lds = ((size_t)w.S * (w.L + 4) + (size_t)w.groups * 5 * w.L + (size_t)4 * 16 * (w.L + 4 + w.S + 4)) * sizeof(float);
#define EXP_F32(N, C)                                                                                                   \
    do {                                                                                                                \
        static bool done_ = false;                                                                                      \
        if (!done_) { set_max_dyn_lds(reinterpret_cast<const void*>(&pw_exp_bwd_f32_kernel<N, C>), 96 * 1024, "pw_exp_bwd_f32"); done_ = true; } \
        hipLaunchKernelGGL((pw_exp_bwd_f32_kernel<N, C>), dim3(nblk), dim3(256), lds, s, a);                            \
    } while (0)
    if (nrt == 6) { if (cc == 1) EXP_F32(6, 1); else EXP_F32(6, 2); }
    else { if (cc == 1) EXP_F32(9, 1); else EXP_F32(9, 2); }
#undef EXP_F32
    return 4 * nblk;
}
