// bf16 pointwise (1x1) convolution kernels of the EfficientNet-B0 bf16 configuration (pwconv_bf16.hip).
#pragma once
#include "common.h"

// D[pix][m] = sum_k Xe[pix][k] * W[m][k]   (forward: W = conv weight [Cout][Cin];
//                                            data gradient: W = its transpose [Cin][Cout], X = dY)
// Xe = X, or with a prologue (squeeze-excite gate fused into the project conv's operand load)
//   Xe = swish(X*psc[g][k] + psh[g][k]) * gate[img][k]     (psc == null: Xe = X * gate[img][k])
struct PwParams {
    const bf16* W;        // [M][K] row-major
    const bf16* X;        // [groups*npix][K]
    bf16* Y;              // [groups*npix][M]
    int M, K;             // multiples of 16
    int npix;             // pixels per BN-statistics group
    int groups;
    int ppb, nblk;        // pixels per block (multiple of 32), blocks per group
    const float* scale;   // optional per-m affine (eval-mode BN folded)
    const float* shift;
    const bf16* res;      // optional residual, indexed like Y
    int act;              // 0 none, 2 swish (after the affine)
    float* stats;         // optional BN partials [groups][nblk][2][M] (from the fp32 accumulators)
    const float* psc;     // prologue: [groups][K] or null
    const float* psh;
    const float* gate;    // prologue: [imgs][K] or null (no prologue at all)
    int HW;               // pixels per image (gate row = global pixel / HW)
    int tiles_m, xcd;     // set by the launcher: M-tiles of the launch; 1 = XCD-aware block order
    const void* zeros;    // >= 16 B of zeros on the device (DMA source of padded chunks in the LDS-tiled form)
};
// nblk the launcher will use (statistics layout); pro = the launch has an operand prologue (PwParams::gate != null),
// HW = pixels per image (the prologue form of the LDS-tiled kernel needs HW >= 32)
int pw_blocks(int npix_per_group, int groups, int M, int K, bool pro = false, int HW = 0);
int pw_tiles_m(int M, int K);       // M-tiles of the launch: the pixel operand (and its prologue) is read once per M-tile
void launch_pw_conv(PwParams p, hipStream_t s);

// dW[m][k] = sum_pix dY[pix][m] * Xe[pix][k]  -> fp32 partial slabs [splits][M][K]
struct PwWgradParams {
    const bf16* dY;       // [npix][M]
    const bf16* X;        // [npix][K]
    float* slab;
    int M, K, npix;
    const float* psc;     // prologue on X as above ([groups][K], groups of pix_per_group pixels)
    const float* psh;
    const float* gate;
    int HW, pix_per_group;
};
// returns the number of splits written (reduce with k_reduce_slabs), 0 = shape not handled
int launch_pw_wgrad(const PwWgradParams& p, size_t slab_floats, hipStream_t s);

// Fused backward of an expand conv + BatchNorm + Swish (early MBConv blocks): reads d a_e and y_e once, forms d y_e on the
// fly, writes the data gradient dX (+ residual) and fp32 partial slabs [slabs][L][S] of the weight gradient.
struct PwExpBwdParams {
    const bf16 *dA, *Ye, *X, *Wt, *res;      // [npix][L], [npix][L], [npix][S], W^T [S][L] (the conv's transposed shadow), [npix][S] or null
    bf16* dX;
    float* slab;
    const float *ca, *cb, *cc, *sc, *sh;     // BN-backward coefficients and the forward's scale / shift, [groups][L]
    int L, S, npix, pix_per_group, groups;
};
// returns the number of slabs written (reduce with k_reduce_slabs), 0 = shape not handled
int launch_pw_exp_bwd(const PwExpBwdParams& p, size_t slab_floats, hipStream_t s);

// Fused backward of a project conv with the squeeze-excite gate and BN1 + Swish in front of it (early MBConv blocks), two
// launches around the squeeze-excite backward: phase 0 = chan_pool5's per-image sums + the weight gradient's slabs from ONE
// read of y_d, phase 1 = d y_d (BN1-backward apply of the re-formed d a_s) from one more read.  d a_s is never stored.
struct PwProjBwdParams {
    const bf16 *dYp, *Yd, *Wt;               // [npix][S], [npix][L], W^T [L][S]
    bf16* dYd;                               // phase 1 result [npix][L]
    float *slab, *pool5;                     // phase 0 results: [slabs][S][L], [imgs][nch][5][L]
    const float *sc, *sh, *mean, *istd, *ca, *cb, *cc;   // [groups][L]
    const float *gate, *ds;                  // [imgs][L]
    int L, S, imgs, HW, ipg, nch;            // ipg = images per statistics group, nch = pw_proj_bwd_nch(...)
};
int pw_proj_bwd_nch(int L, int S, int imgs, int HW);      // pooling records per image; 0 = shape not handled
// phase 0: returns the number of slabs written (0 = not handled); phase 1: 1 = launched, 0 = not handled
int launch_pw_proj_bwd(const PwProjBwdParams& p, int phase, size_t slab_floats, hipStream_t s);

// the fp32-storage twin (projbwd_f32.hip; blocks 0-2): same two phases on the fp32 matrix pipe, W = the conv weight [S][L]
struct PwProjBwdF32Params {
    const float *dYp, *Yd, *W;
    float* dYd;
    float *slab, *pool5;
    const float *sc, *sh, *mean, *istd, *ca, *cb, *cc;
    const float *gate, *ds;
    int L, S, imgs, HW, ipg, nch;
};
int pw_proj_bwd_f32_nch(int L, int S, int imgs, int HW);
int launch_pw_proj_bwd_f32(const PwProjBwdF32Params& p, int phase, size_t slab_floats, hipStream_t s);

// the fp32-storage twin of launch_pw_exp_bwd (expbwd_f32.hip): W = the expand conv's weight [L][S]
struct PwExpBwdF32Params {
    const float *dA, *Ye, *X, *W, *res;
    float* dX;
    float* slab;
    const float *ca, *cb, *cc, *sc, *sh;
    int L, S, npix, pix_per_group, groups;
};
int launch_pw_exp_bwd_f32(const PwExpBwdF32Params& p, size_t slab_floats, hipStream_t s);

// fp32 master weights -> bf16 shadows, all 1x1 convolutions in one launch:
// wb[w_off ...] = bf16(W[m][k]) row-major and wbt[t_off ...] = its transpose [K][M]
struct CastJob { long long src_off, w_off, t_off; int M, K, blk0; };
void launch_cast_weights(const float* state, bf16* shadow, const CastJob* jobs, int njobs, int nblocks, hipStream_t s);
