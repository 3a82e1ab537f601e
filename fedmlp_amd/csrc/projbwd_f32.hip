// Fused backward of a project (1x1) convolution with the squeeze-excite gate and BatchNorm1 + Swish in front of it, fp32
// storage (BASELINE configs[3]: EfficientNet-B0 as the reference runs it), gfx950.  The fp32 twin of pw_proj_bwd_kernel
// (pwconv_bf16.hip), for the early high-resolution MBConv blocks 0-3 (block 4's 240 x 48 weight and five wave tiles exceed the
// LDS of a block).
//
// Reference ops replaced: inside loss.backward() (utils/local_training.py:674, 965, 1191 through efficientnet_pytorch 0.7.1's
// MBConvBlock, model/efficientnet.py:28-33) the backward of  y_p = project_conv(swish(bn1(y_d)) * se_gate):  the conv's weight
// gradient, its data gradient d a_s, the per-image sums the squeeze-excite backward and the BN1 backward take over (d a_s, y_d),
// and the BN1-backward apply.
//
// The unfused order moves the block's depthwise-resolution tensors seven times (weight gradient reads y_d, data gradient writes
// d a_s, the five-sum pooling pass reads d a_s and y_d, the apply pass reads both and writes d y_d).  d a_s = d y_p W is a
// K = 16-32 product of a tensor six times smaller: it is formed TWICE on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32) instead of
// being stored once --
//   phase 0: d a_s tile -> LDS -> the five per-image sums of chan_pool5_kernel and a_s = swish(bn1(y_d)) * gate from the same
//            registers -> dW_p += d y_p^T a_s; reads y_d once, writes partial sums and slabs only;
//   phase 1 (after the BN1-backward finalize): the same d a_s tile again (same instructions, same bits) -> d y_d with
//            bnact_bwd_apply_kernel's arithmetic; reads y_d once, writes d y_d once.
// Layout as in the bf16 kernel: a wave owns one 48-channel (32 for block 0) slice of runs of consecutive 32-pixel tiles of one
// image; elementwise work is lane = (4-channel piece = 16 B, pixel sub-lane), parameters and running sums in registers for a
// whole run.  Roofline: HBM -- but phase 0 issues 96 fp32 MFMAs (32 cycles each) per 6 KB of y_d, about the time the bytes
// take: the fp32 matrix pipe is 16x slower than the bf16 one, the reason the late blocks keep the unfused passes.
#include <stdlib.h>

#include <algorithm>

#include "pwconv.h"

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct ProjBwdF32Args {
    const float *dYp, *Yd, *W;               // [npix][S], [npix][L], conv weight [S][L]
    float* dYd;                              // phase 1: [npix][L]
    float* slab;                             // phase 0: [blocks x run lanes][S][L] partial dW
    float* pool5;                            // phase 0: [imgs][nch][5][L]
    const float *sc, *sh, *mean, *istd;      // BN1 forward affine and statistics [groups][L]
    const float *ca, *cb, *cc;               // BN1-backward coefficients [groups][L] (phase 1)
    const float *gate, *ds;                  // [imgs][L]
    int L, nsl, imgs, HW, ipg, nch;
    float inv_hw;
};

template <int NLT, int CS, int PHASE>
__global__ __launch_bounds__(256) void pw_proj_bwd_f32_kernel(const ProjBwdF32Args p)
{
    extern __shared__ __attribute__((aligned(16))) float smf[];
    constexpr int LS = 16 * NLT, S = 16 * CS;
    constexpr int NOCT = 4 * NLT, NJ = 64 / NOCT, R = (32 + NJ - 1) / NJ, EVL = NJ * NOCT;
    constexpr int SW = S + 4, SB = LS + 4;                  // LDS row lengths in floats: 4 x odd, 16 rows x 4 k-lanes hit 64 banks
    constexpr int CPR = S / 4, NVS = CPR / 2;               // 16-B chunks per d y_p row; a 32-pixel tile = NVS chunks per lane
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int L = p.L, L4 = L >> 2, nsl = p.nsl;
    const int rw = wave / nsl, slice = wave - rw * nsl, nrw = (int)(blockDim.x >> 6) / nsl;
    const int l0 = slice * LS;
    float* wt = smf;                                        // [L][SW]: W^T
    float* dt = smf + (size_t)L * SW + (size_t)wave * 32 * (SB + SW);      // [32][SB]: d a_s, then a_s
    float* st = dt + 32 * SB;                               // [32][SW]: d y_p
    for (int i = tid; i < S * L; i += blockDim.x) {
        const int s = i / L, l = i - s * L;
        wt[l * SW + s] = p.W[i];
    }
    __syncthreads();
    const int oc = lane % NOCT, jr = lane / NOCT;
    const bool ev = lane < EVL;
    const int tpi = (p.HW + 31) >> 5, nruns = p.imgs * p.nch;
    f32x4 acc[PHASE == 0 ? CS : 1][PHASE == 0 ? NLT : 1];
    if constexpr (PHASE == 0) {
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int r = 0; r < NLT; ++r) acc[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 vy[R], vs[NVS];
    for (int run = blockIdx.x * nrw + rw; run < nruns; run += gridDim.x * nrw) {
        const int img = run / p.nch, ch = run - img * p.nch;
        const int t0 = (int)((long long)ch * tpi / p.nch), t1 = (int)((long long)(ch + 1) * tpi / p.nch);
        const int g = img / p.ipg;
        const f32x4* yimg = reinterpret_cast<const f32x4*>(p.Yd + (size_t)img * p.HW * L) + (l0 >> 2) + oc;
        const f32x4* simg = reinterpret_cast<const f32x4*>(p.dYp + (size_t)img * p.HW * S);
        f32x4* dimg = reinterpret_cast<f32x4*>(p.dYd + (size_t)img * p.HW * L) + (l0 >> 2) + oc;
        f32x4 sc, sh, pa, pb, pc, gt, dv, sm[PHASE == 0 ? 5 : 1];
        {
            const int po = g * L + l0 + 4 * oc, io = img * L + l0 + 4 * oc;
            sc = ld4(p.sc + po); sh = ld4(p.sh + po);
            gt = ld4(p.gate + io);
            if constexpr (PHASE == 0) {
                pa = ld4(p.mean + po); pb = ld4(p.istd + po);
#pragma unroll
                for (int t = 0; t < 5; ++t) sm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                pa = ld4(p.ca + po); pb = ld4(p.cb + po); pc = ld4(p.cc + po);
                dv = ld4(p.ds + io) * p.inv_hw;
            }
        }
        // rows at and beyond nv = HW - 32 t (the ragged last tile of a 28 x 28 image) load zeros and store nothing
        auto gload = [&](int t) {
            const int nv = p.HW - 32 * t;
            const f32x4* ty = yimg + (size_t)t * 32 * L4;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int row = jr + NJ * i;
                vy[i] = (ev && row < 32 && row < nv) ? ty[(size_t)row * L4] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const f32x4* tx = simg + (size_t)t * 32 * CPR + lane;
#pragma unroll
            for (int q = 0; q < NVS; ++q) vs[q] = (lane + 64 * q) / CPR < nv ? tx[64 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        gload(t0);
        for (int t = t0; t < t1; ++t) {
            const int nv = p.HW - 32 * t;
            // ---- d y_p tile -> LDS ----
#pragma unroll
            for (int q = 0; q < NVS; ++q) {
                const int c = lane + 64 * q;
                *reinterpret_cast<f32x4*>(st + (c / CPR) * SW + (c % CPR) * 4) = vs[q];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- data gradient: D[l][pix] = sum_s W^T[l][s] d y_p[pix][s]; lane (li, lg): 4 consecutive l of pixel li ----
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                float b[S / 4];
#pragma unroll
                for (int kk = 0; kk < S / 4; ++kk) b[kk] = st[(16 * pt + li) * SW + 4 * kk + lg];
#pragma unroll
                for (int r = 0; r < NLT; ++r) {
                    const float* wrow = wt + (l0 + 16 * r + li) * SW + lg;
                    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < S / 4; ++kk) d = mfma4(wrow[4 * kk], b[kk], d);
                    *reinterpret_cast<f32x4*>(dt + (16 * pt + li) * SB + 16 * r + 4 * lg) = d;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- elementwise: (d a_s, y_d) -> sums + a_s (phase 0) / d y_d (phase 1) ----
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int row = jr + NJ * i;
                if (ev && row < 32 && row < nv) {
                    float* dp = dt + row * SB + 4 * oc;
                    f32x4 d = *reinterpret_cast<const f32x4*>(dp);
                    const f32x4 y = vy[i];
                    const f32x4 v = y * sc + sh;
                    if constexpr (PHASE == 0) {
                        const f32x4 xh = (y - pa) * pb;
                        f32x4 ad, sg;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-v[k]));      // FM_F32_FAST_SWISH forms (effnet.hip)
                            ad[k] = v[k] * sgm;
                            sg[k] = sgm * (1.f + v[k] * (1.f - sgm));
                        }
                        const f32x4 dsg = d * sg;
                        sm[0] += d * ad;
                        sm[1] += dsg;
                        sm[2] += dsg * xh;
                        sm[3] += sg;
                        sm[4] += sg * xh;
                        *reinterpret_cast<f32x4*>(dp) = ad * gt;
                    } else {
                        d = d * gt + dv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-v[k]));
                            d[k] *= sgm * (1.f + v[k] * (1.f - sgm));
                        }
                        dimg[((size_t)t * 32 + row) * L4] = pa * d + pb * y + pc;
                    }
                }
            }
            if (t + 1 < t1) gload(t + 1);
            if constexpr (PHASE == 0) {
                __builtin_amdgcn_wave_barrier();
                // ---- weight gradient: dW[s][l] += sum_pix d y_p[pix][s] a_s[pix][l]; k = pixels, 4 per MFMA ----
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    float a[CS];
#pragma unroll
                    for (int c = 0; c < CS; ++c) a[c] = st[(4 * kk + lg) * SW + 16 * c + li];
#pragma unroll
                    for (int r = 0; r < NLT; ++r) {
                        const float b = dt[(4 * kk + lg) * SB + 16 * r + li];
#pragma unroll
                        for (int c = 0; c < CS; ++c) acc[c][r] = mfma4(a[c], b, acc[c][r]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PHASE == 0) {
            float* scr = dt;                                // 64 lanes x 16 B
            float* rec = p.pool5 + ((size_t)img * p.nch + ch) * 5 * L + l0;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (ev) *reinterpret_cast<f32x4*>(scr + lane * 4) = sm[t];
                __builtin_amdgcn_wave_barrier();
                if (lane < NOCT) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + lane * 4);
                    for (int j = 1; j < NJ; ++j) v += *reinterpret_cast<const f32x4*>(scr + (j * NOCT + lane) * 4);
                    *reinterpret_cast<f32x4*>(rec + t * L + 4 * lane) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if constexpr (PHASE == 0) {
        // acc[c][r][q] = dW[s = 16 c + 4 lg + q][l = l0 + 16 r + li]
        float* slab = p.slab + (size_t)(blockIdx.x * nrw + rw) * L * S + l0;
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int r = 0; r < NLT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) slab[(size_t)(16 * c + 4 * lg + q) * L + 16 * r + li] = acc[c][r][q];
    }
}

static int proj_bwd_f32_slice(int L) { return L == 32 ? 32 : (L % 48 == 0 ? 48 : 0); }

}  // namespace

// pooling records per image; 0 = shape not handled (the caller runs the separate passes)
int pw_proj_bwd_f32_nch(int L, int S, int imgs, int HW)
{
    static const int on = fm_tune("FM_PW_PROJ_BWD_F32", 1);
    const int ls = proj_bwd_f32_slice(L);
    if (!on || !ls || L / ls > 3 || (S != 16 && S != 32 && S != 48) || imgs < 1 || HW % 16 != 0) return 0;
    const int tpi = (HW + 31) / 32;
    return std::max(1, std::min(std::min(16, tpi), (2048 + imgs - 1) / imgs));
}

int launch_pw_proj_bwd_f32(const PwProjBwdF32Params& w, int phase, size_t slab_floats, hipStream_t s)
{
    const int nch = pw_proj_bwd_f32_nch(w.L, w.S, w.imgs, w.HW);
    if (!nch || nch != w.nch || w.imgs % w.ipg != 0) return 0;
    const int ls = proj_bwd_f32_slice(w.L), nsl = w.L / ls;
    ProjBwdF32Args a{};
    a.dYp = w.dYp; a.Yd = w.Yd; a.W = w.W; a.dYd = w.dYd; a.slab = w.slab; a.pool5 = w.pool5;
    a.sc = w.sc; a.sh = w.sh; a.mean = w.mean; a.istd = w.istd; a.ca = w.ca; a.cb = w.cb; a.cc = w.cc;
    a.gate = w.gate; a.ds = w.ds;
    a.L = w.L; a.nsl = nsl; a.imgs = w.imgs; a.HW = w.HW; a.ipg = w.ipg; a.nch = nch; a.inv_hw = 1.f / (float)w.HW;
    const int nrw = nsl == 1 ? 4 : (nsl == 2 ? 2 : 1);
    const int nwaves = nsl * nrw;
    const int nruns = w.imgs * nch;
    int nblk = std::max(1, std::min(1024, (nruns + nrw - 1) / nrw));
    if (phase == 0) nblk = (int)std::min<size_t>(nblk, std::max<size_t>(1, slab_floats / ((size_t)nrw * w.L * w.S)));
    const size_t lds = ((size_t)w.L * (w.S + 4) + (size_t)nwaves * 32 * (ls + 4 + w.S + 4)) * sizeof(float);
#define PROJ_F32(N, C, PH)                                                                                              \
    do {                                                                                                                \
        static bool done_ = false;                                                                                      \
        if (!done_) { set_max_dyn_lds(reinterpret_cast<const void*>(&pw_proj_bwd_f32_kernel<N, C, PH>), 96 * 1024, "pw_proj_bwd_f32"); done_ = true; } \
        hipLaunchKernelGGL((pw_proj_bwd_f32_kernel<N, C, PH>), dim3(nblk), dim3(64 * nwaves), lds, s, a);               \
    } while (0)
    if (phase == 0) {
        if (ls == 32) { if (w.S == 16) PROJ_F32(2, 1, 0); else if (w.S == 32) PROJ_F32(2, 2, 0); else PROJ_F32(2, 3, 0); }
        else { if (w.S == 16) PROJ_F32(3, 1, 0); else if (w.S == 32) PROJ_F32(3, 2, 0); else PROJ_F32(3, 3, 0); }
        return nrw * nblk;
    }
    if (ls == 32) { if (w.S == 16) PROJ_F32(2, 1, 1); else if (w.S == 32) PROJ_F32(2, 2, 1); else PROJ_F32(2, 3, 1); }
    else { if (w.S == 16) PROJ_F32(3, 1, 1); else if (w.S == 32) PROJ_F32(3, 2, 1); else PROJ_F32(3, 3, 1); }
#undef PROJ_F32
    return 1;
}
