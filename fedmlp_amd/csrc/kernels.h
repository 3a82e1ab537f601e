// Launchers of the HBM-bound / small kernels of the FedMLP engine (elementwise.hip,
// heads.hip).  All tensors are fp32; activations are NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

#define FM_MAXC 32

struct ClassVec { float v[FM_MAXC]; };
struct TapList { int t[9]; int n; };

// ---- layout -----------------------------------------------------------------
void k_nchw_to_nhwc4(const float* x, float* y, int imgs, int H, int W, hipStream_t s);
// bf16 stem: col [imgs][Ho][Wo][k][4][4] bf16 from the NCHW fp32 batch (kw_p = 4, 3 channels + 1 zero slot)
void k_stem_im2col(const float* x, void* col, int imgs, int H, int W, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                   hipStream_t s);
// dst[o][h][w(Wpad)][i(Ipad)] <- src[o][i][h][w]  (zero padded) and its inverse
void k_oihw_to_ohwi(const float* src, float* dst, int O, int I, int H, int W, int Wpad, int Ipad, hipStream_t s, int Ostride = 0);
void k_ohwi_to_oihw(const float* src, float* dst, int O, int I, int H, int W, int Wpad, int Ipad, hipStream_t s, int Ostride = 0);
// zero-framed NHWC3 input of the packed 7x7 stem (interior only; the frame is zeroed once at allocation)
void k_stem3_mask_grad(float* g, int O, hipStream_t s);     // zero the gradient of the packed stem's zero tap slots
void k_frame_nhwc3(const float* x, float* y, int imgs, int H, int W, int Hp, int Wp, int top, int left, int src_nhwc3,
                   hipStream_t s, unsigned short* yp = nullptr, long long yp_plane_elems = 0);
// dgrad pack: out[ci][j][co] = w[co][taps.t[j]][ci]   (w is [Co][T][Ci])
void k_pack_dgrad(const float* w, float* out, int Co, int T, int Ci, TapList taps, hipStream_t s);
// all data-gradient weight packs of a step in one launch: job j fills jobs[j].out from state + jobs[j].w_off
struct PackJob { long long w_off; float* out; int Co, T, Ci, ntaps; int taps[9]; int blk0; };
void k_pack_dgrad_all(const float* state, const PackJob* jobs, int njobs, int nblocks, hipStream_t s);
int pack_job_blocks(int Co, int Ci, int ntaps);      // blocks of one job (32 x 32 tiles per tap); PackJob.blk0 = running sum
// bf16 planes (split3.h) of [M][32 nkb] fp32 weight matrices for the split-product conv GEMMs, every matrix of a list in ONE
// launch: job j reads src (or src_base + src_off when src is null) and writes dst_base + dst_off (in 2-byte units) as
// [M][nkb][3 planes][32]; inside a 32-k block, 8-value chunk g holds k = 4g..4g+3, 16+4g..16+4g+3 (the igemm lane groups' order)
struct SplitJob { long long src_off; const float* src; long long dst_off; int M, nkb, blk0; };
void k_split_weights(const float* src_base, unsigned short* dst_base, const SplitJob* jobs, int njobs, int nblocks, hipStream_t s);
int split_job_blocks(int M, int nkb);
// block-major planes for pconv.hip: dst[K block s = (channel block, tap), tap-minor][3 planes][M][32] from W[M][ntaps][cib * 32]
struct SplitJobBM { long long src_off; const float* src; long long dst_off; int M, ntaps, cib, blk0; };
void k_split_weights_bm(const float* src_base, unsigned short* dst_base, const SplitJobBM* jobs, int njobs, int nblocks, hipStream_t s);
int split_job_bm_blocks(int M, int nsteps);
// block-major planes [C/32][3][npix][32] of an fp32 NHWC tensor [npix][C] (C % 32 == 0), and back (x = (h + m) + l, exact)
void k_split_planes(const float* x, unsigned short* dst, long long npix, int C, hipStream_t s);
void k_planes_to_f32(const unsigned short* src, float* x, long long npix, int C, hipStream_t s);
// planes_ew.hip: k_bn_apply / k_stem_pool / k_bn_bwd_apply writing their result as block-major planes `...p` [C/32][3][groups*pix][32]
// (C % 32 == 0) and, where the fp32 pointer is not null, as fp32 too
// resp: the residual as planes (res then null); out: null = planes only
void k_bn_apply_planes(const float* y, const float* scale, const float* shift, const float* res, const float* y2, const float* scale2,
                       const float* shift2, float* out, unsigned short* outp, int groups, int pix_per_group, int C, int relu,
                       hipStream_t s, const unsigned short* resp = nullptr);
void k_stem_pool_planes(const float* y, const float* scale, const float* shift, float* pooled, uint8_t* idx, unsigned short* pooledp,
                        int groups, int imgs_per_group, int H, int W, int C, hipStream_t s);
void k_bn_bwd_apply_planes(const float* dz, const float* z, const float* y, const float* ca, const float* cb, const float* cc, float* dy,
                           unsigned short* dyp, float* dyh_out, int groups, int pix_per_group, int C, hipStream_t s,
                           const float* mask_scale = nullptr, const float* mask_shift = nullptr, const unsigned short* zh = nullptr);
void k_scale(float* x, float w, int64_t n, hipStream_t s);
// utils/FedAvg.py:7-14 over K engine-layout states on one GPU: out = ((s0*n0 + s1*n1) + ...) / tot, the reference's
// left-to-right order with separately rounded products, sums and an IEEE division (bit-identical on fp32 entries)
#define FM_FOLD_MAX 16
struct FoldArgs { const float* s[FM_FOLD_MAX]; float n[FM_FOLD_MAX]; };
void k_fedavg_fold(const FoldArgs& a, int K, float tot, float* out, int64_t n, hipStream_t s);
void k_axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t s);   // y = a*y + b*x

// ---- input pipeline: uint8 HBM cache -> augmented, normalised fp32 NCHW batch.
// params[b] = {c0, c1, c2, c3, c4, c5 (Pillow's 16.16 fixed-point inverse affine), flip, unused}; nearest sampling,
// fill 0, then horizontal flip, /255, (v-mean)/std   (dataset/dataset.py:40-53 pipeline)
void k_augment(const uint8_t* cache, const int* idx, const int* params, float* out, int B, int H, int W,
               float m0, float m1, float m2, float s0, float s1, float s2, hipStream_t s);

// ---- batch norm ---------------------------------------------------------------
// stats: [groups][tiles][2][C] partial (sum, sumsq) from the conv epilogue.
// Writes mean/istd/scale/shift [groups][C]; updates running stats group by group
// (momentum 0.1, unbiased variance), like consecutive train-mode forwards.
void k_bn_finalize(const float* stats, int groups, int tiles, int C, int count,
                   const float* gamma, const float* beta, float* run_mean, float* run_var,
                   float* mean, float* istd, float* scale, float* shift, float eps, float momentum,
                   hipStream_t s, const int* skip = nullptr);    // *skip != 0: the running statistics are not updated
// eval-mode folded affine for all BN channels at once
void k_bn_eval_affine(const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                      float* scale, float* shift, int n, float eps, hipStream_t s);
// out = [relu]( y*scale+shift [+ res] [+ y2*scale2+shift2] ), per group
void k_bn_apply(const float* y, const float* scale, const float* shift, const float* res,
                const float* y2, const float* scale2, const float* shift2, float* out,
                int groups, int pix_per_group, int C, int relu, hipStream_t s);
// stem: pooled = maxpool3x3s2p1(relu(y*scale+shift)) (+argmax code) ; scale==null -> plain maxpool
void k_stem_pool(const float* y, const float* scale, const float* shift, float* pooled, uint8_t* idx,
                 int groups, int imgs_per_group, int H, int W, int C, hipStream_t s);
void k_stem_pool_bwd(const float* dpooled, const float* pooled, const uint8_t* idx, float* dy,
                     int imgs, int H, int W, int C, hipStream_t s);
// stem: max-pool backward + BatchNorm backward without the dense 112x112 intermediate.  reduce: the two BN-backward sums over the
// pooled positions (each window's gradient at its argmax, ReLU mask = pooled > 0) -> part[groups][stem_pool_bn_blocks()][2][C];
// apply: dy[dense] = ca * (sum of the window gradients that chose this position) + cb * y + cc
// test hook: bit-packed ReLU mask of the dense stem map, bits[pix][C/8] (bit j of byte b = y*scale+shift > 0 of channel 8b + j)
void k_stem_relu_bits(const float* y, const float* scale, const float* shift, uint8_t* bits, int groups, int64_t pix_per_group, int C,
                      hipStream_t s);
int stem_pool_bn_blocks(int pooled_per_group);
void k_stem_pool_bn_reduce(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y, const float* mean,
                           const float* istd, float* part, int groups, int imgs_per_group, int H, int W, int C, hipStream_t s,
                           const float* gamma = nullptr, const float* beta = nullptr);    // gamma / beta: xhat from the pooled value
void k_stem_pool_bn_apply(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y, const float* ca,
                          const float* cb, const float* cc, float* dy, int groups, int imgs_per_group, int H, int W, int C,
                          hipStream_t s);
// backward: partial sums of dyh = dz*(z>0) and dyh*xhat -> part[groups][nblk][2][C]
int bn_bwd_blocks(int pix_per_group);
// ReLU mask of dz: z > 0 (z = stored post-activation), or, with z null and mask_scale / mask_shift given, y*scale+shift > 0
// zh (planes mode): z exists only as block-major planes [C/32][3][groups*pix][32]: the mask is the sign of its h plane
void k_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean, const float* istd,
                     float* part, int groups, int pix_per_group, int C, hipStream_t s,
                     const float* mask_scale = nullptr, const float* mask_shift = nullptr, const unsigned short* zh = nullptr);
// coefficients ca,cb,cc [groups][C]; dgamma/dbeta written (summed over groups)
void k_bn_bwd_finalize(const float* part, int groups, int nblk, int C, int count, const float* gamma,
                       const float* mean, const float* istd, float* ca, float* cb, float* cc,
                       float* dgamma, float* dbeta, hipStream_t s);
// dy = ca*dyh + cb*y + cc ; optionally store dyh
void k_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* ca, const float* cb,
                    const float* cc, float* dy, float* dyh_out, int groups, int pix_per_group, int C,
                    hipStream_t s, const float* mask_scale = nullptr, const float* mask_shift = nullptr);

// ---- EfficientNet-B0 path (effnet.hip): any C % 4 == 0, act 0 none / 1 relu / 2 swish ---------
// Activation tensors are fp32 or bf16 in HBM (DT_F32 / DT_BF16, common.h); arithmetic is fp32.  `ty` is the
// storage type of raw conv outputs (y, dy), `ta` / `dt` that of activations and their gradients; per-channel
// and per-image vectors (scale, shift, gate, statistics, partial sums) are always fp32.
// out = act(y*scale+shift) * rowscale[img] + res
void k_bnact_apply(const void* y, int ty, const float* scale, const float* shift, const void* res, const float* rowscale,
                   void* out, int ta, int groups, int pix_per_group, int HW, int C, int act, hipStream_t s);
// mode 0: (sum y, sum y^2); mode 1: (sum dyh, sum dyh*xhat), dyh = a*act'(y*scale+shift)*rowscale
// part [groups][bn_bwd_blocks(pix_per_group)][2][C]
void k_chan_reduce(const void* a, int ta, const void* y, int ty, const float* mean, const float* istd, const float* scale,
                   const float* shift, const float* rowscale, float* part, int groups, int pix_per_group, int HW,
                   int C, int mode, int act, const float* gate, const float* dsv, hipStream_t s);
void k_bnact_bwd_apply(const void* dz, int ta, const void* y, int ty, const float* ca, const float* cb, const float* cc,
                       const float* scale, const float* shift, const float* rowscale, void* dy, int groups,
                       int pix_per_group, int HW, int C, int act, const float* gate, const float* dsv, hipStream_t s);
// gate/dsv (optional, [imgs][C]): the incoming gradient is d(a_s); d(a_d) = d(a_s)*gate + dsv/HW is formed on load
// depthwise KxK (K 3 or 5): x [imgs][Hi][Wi][C], w [K*K][C]; optional fused y = act(y*scale+shift)
int dw_stats_tiles();                 // per-group partials k_dw_fwd leaves in stats_out when it serves the request
bool k_dw_fwd(const void* x, const float* w, void* y, int dt, const float* scale, const float* shift, int imgs, int Hi,
              int Wi, int Ho, int Wo, int C, int K, int stride, int pad_t, int pad_l, int act, hipStream_t s,
              float* stats_rec = nullptr, float* stats_out = nullptr, int groups = 1, float* pool_out = nullptr);
bool k_dw_dgrad(const void* dy, const float* w, void* dx, int dt, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                int stride, int pad_t, int pad_l, hipStream_t s, const void* ye = nullptr, const float* mean = nullptr,
                const float* istd = nullptr, const float* scale = nullptr, const float* shift = nullptr,
                float* stats_rec = nullptr, float* stats_out = nullptr, int groups = 1);
int dw_wgrad_blocks(int npix);
// out [K*K][C] = the weight gradient; part = workspace for the per-block partial sums (reduced inside, fixed order)
void k_dw_wgrad(const void* dy, const void* x, int dt, float* part, float* out, int imgs, int Hi, int Wi, int Ho, int Wo, int C, int K,
                int stride, int pad_t, int pad_l, hipStream_t s);
// squeeze-excite: W1 [Cs][C], W2 stored transposed [Cs][C]
// pool_ws: [imgs][16][C] scratch for the per-image channel sums
// scale/shift (optional, [groups][C], ipg images per group): `a` is the raw depthwise output and is read
// as swish(a*scale+shift) -- the post-BN activation is not materialised in the train path
void k_se_fwd(const void* a, int dt, const float* scale, const float* shift, int ipg, float* pool_ws, const float* W1,
              const float* b1, const float* W2, const float* b2, float* sq, float* rpre, float* gate, int imgs, int HW,
              int C, int Cs, hipStream_t s, bool pooled = false);
void k_se_scale(const void* a, int dt, const float* scale, const float* shift, int ipg, const float* gate, void* out, int imgs,
                int HW, int C, hipStream_t s);
void k_se_bwd(const void* dout, const void* a, int dt, const float* scale, const float* shift, int ipg, float* pool_ws,
              const float* gate, const float* rpre, const float* W1, const float* W2, float* dgp, float* drp, float* ds,
              int imgs, int HW, int C, int Cs, hipStream_t s);
// squeeze-excite backward AND the BN1-backward sums from one pass over (dout = d a_s, y = raw depthwise output);
// pool_ws [imgs][16][5][C]; bn_part [groups][1][2][C] is what k_bn_bwd_finalize consumes with nblk = 1
int se_bwd_bn1_splits(int ipg);       // per-group partials k_se_bwd_bn1 leaves in bn_part
void k_se_bwd_bn1(const void* dout, const void* y, int dt, const float* scale, const float* shift, const float* mean,
                  const float* istd, int ipg, float* pool_ws, const float* gate, const float* rpre, const float* W1,
                  const float* W2, float* dgp, float* drp, float* ds, float* bn_part, int imgs, int HW, int C, int Cs,
                  hipStream_t s, int nch_ready = 0);
// dW1 = start of the contiguous [dW1 | db1 (padded to 4) | dW2 | db2] gradient range; part = workspace (16 slabs of it)
void k_se_wgrad(const float* dgp, const float* drp, const float* rpre, const float* sq, float* part, float* dW1, int imgs, int C,
                int Cs, hipStream_t s);
void k_mul(const float* a, const float* b, float* y, int64_t n, hipStream_t s);
void k_add_inplace(void* y, const void* a, int dt, int64_t n, hipStream_t s);

// ---- head ---------------------------------------------------------------------
void k_avgpool(const void* x, int dt, float* feat, int imgs, int HW, int C, hipStream_t s);
void k_fc_fwd(const float* feat, const float* W, const float* b, float* logits, int imgs, int D, int C,
              hipStream_t s);
// dW[k][d], db[k] written; dout[img][hw][d] = (sum_k dz[img][k] W[k][d]) * mask[img][d] / HW  (mask optional)
void k_fc_bwd(const float* dz, const float* feat, const float* W, const float* mask, float* dW, float* db,
              void* dout, int dt, int imgs, int D, int C, int HW, hipStream_t s);

// ---- losses (one block; deterministic) -------------------------------------------
void k_loss_bce(const float* z, const float* y, ClassVec pos_w, int B, int C, float inv_norm,
                float* dz, float* loss, hipStream_t s);
void k_loss_stage1(const float* z, const float* g, const float* y, ClassVec active, int B, int C,
                   float inv_sup, float inv_dis, float* dz, float* loss, hipStream_t s);
void k_loss_stage2(const float* z, const float* y, const float* distill, int B, int C, float* dz,
                   float* loss, hipStream_t s);
void k_loss_fixmatch(const float* z, const float* y, ClassVec pos_w, ClassVec pos_wu, ClassVec active,
                     int B, int C, int n_neg, float inv_sup, int n_cls_minus_ann, float* dz, float* loss,
                     hipStream_t s);

// ---- optimiser -------------------------------------------------------------------
void k_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
            float eps, float wd, float bc1, float bc2_sqrt, hipStream_t s, const int* skip = nullptr);   // *skip != 0: no update
void k_reduce_slabs(const float* slab, float* out, int splits, int64_t n, hipStream_t s);

// ---- prototypes / tagging ----------------------------------------------------------
void k_proto_accumulate(const float* feat, const float* logits, const float* labels, int B, int D, int C,
                        ClassVec active, ClassVec negative, float L, float U, float* psum,
                        int64_t* pcnt, int64_t* tcnt, hipStream_t s);
void k_cos_tag(const float* feat, int64_t N, int D, const float* proto, const int* classes, int ncls,
               float* sim, hipStream_t s);
void k_count_sign(const float* sim, int64_t N, int* counts /*[2]: >=0, <0*/, hipStream_t s);
// stable ranks: top[rank_desc] = pos if rank_desc < ktop ; bot[rank_asc] = pos if rank_asc < kbot
void k_rank_select(const float* sim, int64_t N, int ktop, int kbot, int* top, int* bot, hipStream_t s);
// all classes of a round: counts [ncls][2], top / bot [ncls][cap] (positions inside each class's pool); rows null = whole rows
void k_select_rows(const float* sim, int64_t N, int ncls, const int* rows, const int* pn, int stride, int maxn, double clean_thr,
                   double noise_thr, int cap, int* counts, int* top, int* bot, hipStream_t s);
