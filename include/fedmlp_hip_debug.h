/*
 * fedmlp_hip_debug.h -- kernel-level test hooks of libfedmlp_hip.so (tests/ and tools/ only).
 *
 * NOT part of the drop-in surface of fedmlp_hip.h: nothing behind build_model() / LocalUpdate / FedAvg* calls these.
 * They expose single kernels (one convolution forward / data gradient / weight gradient on caller-supplied tensors),
 * the activations the last train-mode forward kept, and the gradients of the last step, so that the parity tests can
 * compare each kernel with a CPU yardstick.
 */
#ifndef FEDMLP_HIP_DEBUG_H
#define FEDMLP_HIP_DEBUG_H

#include "fedmlp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* info[0..12] = cin, cout, k, stride, pad, hin, win, hout, wout, cin_p (padded
 * input channels of the NHWC operand), Kw (row length of the engine-layout
 * weight matrix [cout_p][Kw] = [cout_p][k][kw_p][cin_p]), kw_p, cout_p (padded output
 * channels of the NHWC result; = cout for ResNet-18); info[13..15] = 0. */
int fm_debug_conv_info(fm_engine* e, int32_t conv, int32_t* info16);
int fm_debug_num_convs(fm_engine* e);
/* op 0: raw forward  x[imgs,hin,win,cin_p] -> out[imgs,hout,wout,cout]; if stats_dev
 *       != NULL also the per-group per-channel (sum, sumsq) [groups][2][cout]
 * op 1: data gradient dy[imgs,hout,wout,cout] -> out[imgs,hin,win,cin]
 * op 2: weight gradient (x, dy) -> out[cout][Kw] (engine layout)
 * All tensors NHWC fp32 on device; weights are the engine's current state. */
int fm_debug_conv(fm_engine* e, int32_t op, int32_t conv, const float* x_dev, const float* dy_dev,
                  float* out_dev, int32_t imgs, int32_t groups, float* stats_dev);

/* bf16 pointwise-convolution kernels of a precision-1 engine (conv must be a 1x1 convolution):
 * op 0: x bf16 [imgs,h,w,cin_p] -> out bf16 [imgs,h,w,cout_p] raw; stats_dev (optional) per-group (sum, sumsq)
 *       [groups][2][cout_p] fp32; gate_dev != NULL applies the operand prologue
 *       x <- swish(x*psc[g]+psh[g]) * gate[img]   (psc_dev NULL: x * gate[img]); psc/psh [groups][cin_p], gate [imgs][cin_p]
 * op 1: dy bf16 [imgs,h,w,cout_p] -> out bf16 [imgs,h,w,cin_p]; x_dev (optional) = residual added to the result
 * op 2: (x, dy) -> out fp32 [cout_p][cin_p], with the same optional prologue on x */
int fm_debug_pw(fm_engine* e, int32_t op, int32_t conv, const void* x_dev, const void* dy_dev, void* out_dev,
                int32_t imgs, int32_t groups, const float* psc_dev, const float* psh_dev, const float* gate_dev,
                float* stats_dev);

/* Fused backward of a project convolution (pw_proj_bwd_kernel: blocks 0-4 of a precision-1 EfficientNet engine, tensors bf16;
 * pw_proj_bwd_f32_kernel: blocks 0-2 of a precision-0 one, every "bf16" below is then fp32):
 * dyp bf16 [imgs,h,w,cout_p] (= d y_p), yd bf16 [imgs,h,w,cin_p] (the depthwise output y_d), bn [7][groups][cin_p] fp32 =
 * BN1's scale, shift, mean, istd and the BN1-backward coefficients ca, cb, cc; gate / ds fp32 [imgs][cin_p].
 * phase 0: out = fp32 dW [cout_p][cin_p] with a_s = swish(yd*scale+shift)*gate, pool5 = fp32 [imgs][5][cin_p], the five
 *          per-image sums of (d a_s, y_d) that the squeeze-excite and BN1 backward need (d a_s = bf16(dyp W));
 * phase 1: out = bf16 d y_d [imgs,h,w,cin_p] = ca*((d a_s*gate + ds/HW)*swish'(v)) + cb*yd + cc. */
int fm_debug_proj_bwd(fm_engine* e, int32_t conv, int32_t phase, const void* dyp_dev, const void* yd_dev, const float* bn_dev,
                      const float* gate_dev, const float* ds_dev, int32_t imgs, int32_t groups, void* out_dev,
                      float* pool5_dev);

/* Fused backward of an expand convolution with its BatchNorm + Swish (pw_exp_bwd_kernel, precision 1: tensors bf16;
 * pw_exp_bwd_f32_kernel, precision 0: fp32; conv = an expand conv of blocks 1-3):  da = d a_e and ye = y_e [imgs,h,w,cout_p],
 * x = the block input [imgs,h,w,cin_p], res (optional) = the skip connection's gradient [imgs,h,w,cin_p], bn [5][groups][cout_p]
 * fp32 = the BN0-backward coefficients ca, cb, cc and the forward's scale, shift.
 * dx = d y_e W (+ res) [imgs,h,w,cin_p] with d y_e = ca*(da*swish'(ye*scale+shift)) + cb*ye + cc; dw = fp32 [cout_p][cin_p]. */
int fm_debug_exp_bwd(fm_engine* e, int32_t conv, const void* da_dev, const void* ye_dev, const void* x_dev, const void* res_dev,
                     const float* bn_dev, int32_t imgs, int32_t groups, void* dx_dev, float* dw_dev);

/* Post-ReLU activations the last train-mode forward kept (ResNet-18): kind 0 = relu(bn1(conv1)) of
 * basic block `block`, kind 1 = the block's output relu(bn2(conv2) + identity); NHWC fp32 for the first
 * `imgs` images, dims4 = {imgs, H, W, C}.  host_nhwc may be NULL to query the dims only.  Parity tests
 * hand the ReLU masks (value > 0) to the oracle's backward pass, so a pre-activation within rounding
 * distance of zero cannot turn a 1e-6 forward difference into a percent-level gradient difference. */
int fm_debug_activation(fm_engine* e, int32_t kind, int32_t block, int32_t imgs, float* host_nhwc,
                        int32_t* dims4);

/* The discrete decisions of the ResNet-18 stem in the last train-mode forward, for the same purpose: the ReLU mask of
 * relu(bn1(conv1(x))) over the dense [imgs][H/2][W/2][64] map, bit-packed (byte b of a pixel = channels 8b .. 8b+7, bit j =
 * channel 8b + j: relu_bits_host [imgs][H/2][W/2][8]), and the 3x3 / stride-2 max-pool's choice per pooled element
 * (argmax_host [imgs][H/4][W/4][64], code kh * 3 + kw of the chosen window position).  groups = the views of that forward
 * (BatchNorm statistics are per view). */
int fm_debug_stem_masks(fm_engine* e, int32_t imgs, int32_t groups, uint8_t* relu_bits_host, uint8_t* argmax_host);

/* Fault injection for the stream-K fix-up of the planes conv GEMMs (pconv.hip): on = 1 makes the owners of partial tiles keep
 * their arrival announcements to themselves (and shortens the finisher's bounded wait), so that every shared tile times out.
 * Expected behaviour, which tests/test_engine_gpu.py checks: the step's optimizer update is skipped on the device (weights,
 * moments unchanged) and the NEXT call that ends a step or synchronises returns FM_ERR_HIP once.  Process-wide; on = 0 restores. */
int fm_debug_lose_part(int32_t on);

/* Gradients of the last step in state_dict order (running-stat slots are 0). */
int fm_debug_get_grads(fm_engine* e, float* host_f32);

#ifdef __cplusplus
}
#endif
#endif /* FEDMLP_HIP_DEBUG_H */
