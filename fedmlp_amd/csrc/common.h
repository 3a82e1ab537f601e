// Shared declarations for the FedMLP HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Swish in the fp32 configuration's streaming kernels and conv epilogues: 1 = hardware v_exp_f32 / v_rcp_f32 (both within ~2 ulp
// of the IEEE sequences; the step parity against the fp32 oracle is unchanged at its 2e-5 / 5e-4 bounds), 0 = expf and 1/x.
#ifndef FM_F32_FAST_SWISH
#define FM_F32_FAST_SWISH 1
#endif
#ifdef __HIPCC__
__device__ __forceinline__ float fm_swish_f32(float v)
{
#if FM_F32_FAST_SWISH
    return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
#else
    return v / (1.f + expf(-v));
#endif
}
#endif

// Tuning knobs of the kernels (tile orders, split counts, variant choices that were measured and settled): the shipped
// library uses the defaults; a `make TUNING=1` build (-DFM_TUNING) reads them from the environment for measurements.
// The few RUNTIME switches the shipped library does read are listed in DESIGN.md section 9, each with the test that
// exercises it (FM_MFMA_SPLIT, FM_PLANES, FM_IGEMM_BLOCKS, FM_STEM_PACKED, FM_BN_MASK_FROM_Y, FM_EW_ROWS / FM_EW_ROWS_F32,
// FM_DW_GENERIC, FM_FUSE_GATE; FM_DEBUG_REUSE_PLANES belongs to the timing probes of the test hooks).
inline int fm_tune(const char* name, int dflt)
{
#ifdef FM_TUNING
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// The ResNet conv GEMMs (igemm.hip, wgrad.hip) form their fp32 products either on the fp32 matrix pipe (0) or as exact bf16
// partial products on the bf16 matrix pipe (9 = all nine, 6 = without the three below 2^-24 of the product; split3.h).  The form is
// a property of the engine: fm_config.reserved[2] (0 = this default, 1 = fp32 pipe, 2 = nine products), fixed at fm_create
// and carried to the launchers in IgemmParams / WgradParams `sp`.  fm_mfma_split() resolves the default; the environment
// variable FM_MFMA_SPLIT overrides it for tests and is read there, once per fm_create, never per launch.
#ifndef FM_MFMA_SPLIT_DEFAULT
#define FM_MFMA_SPLIT_DEFAULT 6
#endif
inline int fm_mfma_split()
{
    const char* v = getenv("FM_MFMA_SPLIT");
    const int s = v ? atoi(v) : FM_MFMA_SPLIT_DEFAULT;
    return (s == 9 || s == 6) ? s : 0;
}

// ---- activation storage types -------------------------------------------------------------------
// The EfficientNet-B0 path stores its NHWC activations either as fp32 (the reference's arithmetic)
// or as bf16 (BASELINE configs[4]); every kernel computes in fp32 registers.  ld4 / st4 move 4
// consecutive channels: 16 B (fp32) or 8 B (bf16) per lane.  The float -> bf16 cast is the
// compiler's (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN).
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { DT_F32 = 0, DT_BF16 = 1 };
#if defined(__HIPCC__)
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16* p)
{
    return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
// Bounds-checked load WITHOUT a branch: the caller clamps the address (p is always readable) and passes the predicate;
// the value is zeroed afterwards.  `cond ? ld4(p) : 0` compiles to a branch + s_waitcnt vmcnt(0) per element for bf16
// (every load of a depthwise window serialised on the previous one's latency); this form keeps the loads in flight.
__device__ __forceinline__ f32x4 ld4z(const float* p, bool ok)
{
    const f32x4 v = ld4(p);
    return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 ld4z(const bf16* p, bool ok)
{
    uint2 r = *reinterpret_cast<const uint2*>(p);
    r.x = ok ? r.x : 0u;
    r.y = ok ? r.y : 0u;
    return __builtin_convertvector(__builtin_bit_cast(bf16x4, r), f32x4);
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }
// NV consecutive channel quads per thread: one 16-B access per lane in either storage type
// (NV = 1 for fp32 = 4 channels, NV = 2 for bf16 = 8 channels)
template <typename T> struct VecOf { static constexpr int NV = 1; };
template <> struct VecOf<bf16> { static constexpr int NV = 2; };
template <int NV, typename T> __device__ __forceinline__ void ldv(const T* p, f32x4 (&v)[NV])
{
    if constexpr (NV == 1) v[0] = ld4(p);
    else {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
        v[0] = __builtin_convertvector(__builtin_shufflevector(r, r, 0, 1, 2, 3), f32x4);
        v[1] = __builtin_convertvector(__builtin_shufflevector(r, r, 4, 5, 6, 7), f32x4);
    }
}
template <int NV, typename T> __device__ __forceinline__ void stv(T* p, const f32x4 (&v)[NV])
{
    if constexpr (NV == 1) st4(p, v[0]);
    else {
        const bf16x4 a = __builtin_convertvector(v[0], bf16x4), b = __builtin_convertvector(v[1], bf16x4);
        *reinterpret_cast<bf16x8*>(p) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
// float vectors (scale, gate ...) read as NV quads
template <int NV> __device__ __forceinline__ void ldf(const float* p, f32x4 (&v)[NV])
{
#pragma unroll
    for (int h = 0; h < NV; ++h) v[h] = ld4(p + 4 * h);
}
#endif

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution (forward and data-gradient), fp32 MFMA.
//   D[m][n] = sum_k Wp[m][k] * Xg[n][k]
//   m: output channel, n: output pixel (per BN-statistics group), k: (tap, ci)
// The K axis is (tap, channel): `ntaps` taps of `Ci` gathered channels each, cut
// in steps of 32 floats = 8 chunks of 16 B.  Tap t reads the input pixel at
// offset (dh[t], dw[t]) from the output pixel's base position; the tap arrays
// sit in the kernel arguments, so the per-step tap lookup is scalar work and the
// main loop has no dependent global load.  The kernel knows nothing else about
// strides or padding, so forward convs and the per-parity-class data gradients
// all run through it.  stem_kw > 0 selects the 7x7-stem form (Ci = 4): one step
// = one kernel row, chunk j = kernel column j (valid while j < stem_kw).
// ---------------------------------------------------------------------------
struct IgemmParams {
    const float* W;       // packed weights [M][nsteps*32]
    const float* X;       // input activations NHWC [imgs][Hi][Wi][Ci]
    float* Y;             // output NHWC [imgs][Ho][Wo][Co]
    int dh[9], dw[9];     // per-tap input offsets
    int ntaps;
    unsigned long long tapcode;   // 4 bits per tap: (dh+1) | (dw+1)<<2  (filled by launch_igemm from dh/dw)
    int tap_minor;        // K walked chunk-major / tap-minor (L2 reuse) instead of tap-major
    int tn_fast;          // tile order: pixel tiles fastest within an M-tile (weight slice stays in L2)
    int stem_kw, stem_pad;
    int stem_h2;          // 1: kernel width padded to 8 (two 16-k steps per kernel row), 0: padded to 4
    int stem3;            // 1: packed 7x7 stem -- X is a zero-framed NHWC3 image (Hi, Wi = framed sizes, Ci = 3), K = kernel
                          // rows of 24 floats (7 taps x 3 channels + 1 zero tap) = 6 chunks each, 16-B chunks at 4-B alignment
    const float* zeros;   // >= 16 B of zeros (source of padded / out-of-range chunks)
    const float* res;     // optional residual, indexed like Y (may alias Y)
    const float* scale;   // optional per-m affine (eval-mode BN folded), else null
    const float* shift;
    float* stats;         // optional BN partials [group][tilesN][2][M]
    float* slab;          // stream-K partial tiles [blocks][2][BM*BN]
    int* counters;        // stream-K arrival counters [tiles], zero between launches
    long long total_steps;   // tiles * nsteps (set by launch_igemm)
    int steps_per_block;     // range length per persistent block (set by launch_igemm)
    int M, nsteps;
    int Hi, Wi, Ci;
    int Hg, Wg, sg;       // virtual output grid per image and its stride into the input
    int Ho, Wo, Co;       // output tensor dims
    int os, oh0, ow0;     // output placement: oh = hg*os + oh0
    int imgs_per_group;   // images per BN-statistics group (grid.y = group)
    int tilesM, tilesN;   // tiles per group
    int relu;
    // conv1x1.hip only: per-image gate on the operand, Xe[pix][k] = X[pix][k] * gate[pix / gate_HW][k] (the squeeze-excite
    // gate of an eval-mode project conv: the gated activation is never written); null = no prologue
    const float* gate = nullptr;
    int gate_HW = 1;
    // + BN1 + Swish in front of the gate (train-mode project conv of the blocks whose backward is fused):
    // Xe = swish(X * psc[group][k] + psh[group][k]) * gate; null = gate only
    const float* psc = nullptr;
    const float* psh = nullptr;
    int sp = 0;           // product form of the launch (fm_engine::products): 0 fp32 matrix pipe, 6 / 9 bf16 partial products
    // split-product form (split3.h), M >= 128: the bf16 planes of W made by k_split_weights, [M][nsteps of 32 k][3 planes][32]
    // (192 B per 32-k block, the k order inside a block as the kernel's lane groups take it); null = split W in the kernel
    const unsigned short* Wsp = nullptr;
    // pconv.hip (both operands as block-major bf16 planes): Wsp then holds [nsteps = (channel block, tap)][3][M][32]; Xp the
    // activation planes [Ci/32][3][xp_pix][32] of the WHOLE input tensor (xp_pix = its pixel count); Y may be null when only planes
    // are wanted; Yp (eval epilogue, dense output): the output's planes [Co/32][3][yp_pix][32]
    const unsigned short* Xp = nullptr;
    long long xp_pix = 0;
    unsigned short* Yp = nullptr;
    long long yp_pix = 0;
    const unsigned short* resp = nullptr;   // the residual as planes [Co/32][3][yp_pix][32] (dense output only), instead of `res`
    // floats per row of W when that is not nsteps * KS (the packed stem in 32-k stages: rows of 176 floats walked as 6 x 32 --
    // the 16 floats past a row's end meet the zero-page chunks of X); 0 = nsteps * KS
    int wrow = 0;
    // pconv.hip scheduling switches (set by launch_pconv): bit 0 = waves 4-7 issue a step's DMA behind its last column's MFMAs
    // instead of in front of them (their SIMD partners 0-3 issue in front: one of a pair computes while the other issues),
    // bit 1 = s_setprio 1 for waves 4-7, bit 2 = s_setprio 1 for waves 0-3
    int pc_flags = 0;
    // pconv.hip: where a stream-K finisher reports a part that never arrived (its wait is bounded): a device word the optimizer
    // kernel reads (a step with a lost part does not update the weights) and its host-mapped twin the next API call checks
    int* err = nullptr;
    int* err_host = nullptr;
};
// test hook (fedmlp_hip_debug.h fm_debug_lose_part): the non-finishers of shared stream-K tiles do not announce their parts
void pconv_debug_lose_part(int on);

// Weight-gradient GEMM: dW[m][n] = sum_p dY[p][m] * Xg[p][n], split over p.
struct WgradParams {
    const float* dY;      // [imgs][Ho][Wo][M]
    const float* X;       // [imgs][Hi][Wi][Ci]
    float* slab;          // [splits][M][Nw]
    const int4* tab;      // [Nw/4]: {dh, dw, ci0, valid} per 4-column chunk
    const float* zeros;   // >= 16 B of zeros
    int M, Nw;
    int Ho, Wo, Hi, Wi, Ci, stride;
    int npix;             // imgs*Ho*Wo
    int pix_per_split;    // multiple of 32
    int tilesM, tilesN;
    int bn;               // N-tile width the caller tiled with (wgrad_tile_n)
    int gather_k, gather_pad, gather_kw_p;   // stem form for launch_wgrad_skinny (kernel size, top/left pad, padded width)
    int sp = 0;           // product form of the launch (fm_engine::products): 0 fp32 matrix pipe, 6 / 9 bf16 partial products
};

// pwgrad.hip: the weight gradient with both operands as block-major bf16 planes (ResNet-18 planes mode)
struct PwgradParams {
    const unsigned short* dYp;   // [M/32][3][npix][32]: planes of the output gradient, npix = imgs*Ho*Wo
    const unsigned short* Xp;    // [Ci/32][3][xpix][32]: planes of the conv input, xpix = imgs*Hi*Wi
    float* slab;                 // [splits][M][Nw]
    int M, Nw;                   // Nw = ksz*ksz*Ci, n = tap*Ci + ci
    int Ho, Wo, Hi, Wi, Ci, stride, pad, ksz;
    long long npix, xpix;
    int pix_per_split;           // multiple of 32 (set by the launcher)
    int tilesM, tilesN, nblk_n;  // set by the launcher
    int sp;                      // product form: 6 / 9
    int pw_flags = 0;            // bit 0: waves 4-7 issue a step's DMA behind the step's last MFMAs (their SIMD partners in front of them)
    // pwgrad_ring.hip (set by its launcher): images, padded positions in all, positions per split (multiple of 32), splits,
    // blocks of a split share an XCD, ceil(2^32 / (Wi + 1)), ceil(2^32 / (Hi + 1))
    int nimg, Qtot, q_per_split, splits, xcd_remap;
    unsigned magW, magH;
};
// stem_rows.hip: ResNet-18's 7x7 stem forward from planes of the framed input (four channels per pixel) and resident weight planes
struct StemRowsParams {
    const unsigned short* Xp;     // [3 planes][imgs][Hp][Wp][4] bf16
    const unsigned short* Wst;    // [7 kh][3 planes][64][32] bf16, k = 4 kw + c
    float* Y;                     // [imgs][Ho][Wo][64]
    const float* scale;           // eval: folded BatchNorm per channel (with relu), else null
    const float* shift;
    float* stats;                 // train: BatchNorm partial sums [group][imgs_per_group * Ho / 4][2][64], else null
    int imgs, imgs_per_group, Hp, Wp, Ho, Wo, relu, sp;
    long long plane_bytes;        // imgs * Hp * Wp * 8
};
bool stem_rows_takes(int k, int stride, int cout, int hout, int wout, long long plane_bytes);
int stem_rows_stats_tiles(int imgs_per_group, int hout);
void launch_stem_rows(StemRowsParams p, hipStream_t s);
void k_stem_weight_planes(const float* w, unsigned short* dst, int Kw, hipStream_t s);
int launch_pwgrad(PwgradParams p, size_t slab_floats, hipStream_t s);     // returns the slabs written (0 = nothing launched)
// pwgrad_ring.hip: 3x3 / stride 1 on the wide maps, X staged once per block through a ring of padded positions
bool pwgrad_ring_takes(const PwgradParams& p);
int launch_pwgrad_ring(PwgradParams p, size_t slab_floats, hipStream_t s);
bool pwgrad_takes(int M, int Ci, int ksz, long long npix, long long xpix, int Wi, int pad);
void launch_igemm(IgemmParams p, int groups, hipStream_t s);
// pconv.hip: the same GEMM with both operands as block-major bf16 planes (p.Wsp, p.Xp); p.nsteps is set by the launcher
void launch_pconv(IgemmParams p, int groups, hipStream_t s);
bool pconv_takes(int M, int Ci, long long xp_pix, int Wi);
bool pconv_uses_ts(const IgemmParams& p);      // will launch_pconv take the tap-row-sharing kernel for these parameters?
int pconv_tile_m(int M);
int pconv_tile_n(int M);
size_t pconv_slab_floats();   // stream-K slab of launch_pconv (p.slab): pconv_max_blocks() x 2 x the largest tile
// small-K (Ci <= 256) 1x1 stride-1 convolutions; false = shape not handled (run igemm)
bool launch_conv1x1_stream(const IgemmParams& p, int groups, hipStream_t s);
bool conv1x1_stream_takes(int Ci, int M, int Co);     // would a stride-1 1x1 conv of this shape stream through conv1x1.hip?
int igemm_max_blocks();
bool launch_wgrad(const WgradParams& p, int splits, hipStream_t s);    // false = split exceeds the 32-bit offset span
int wgrad_tile_n(int M, int Nw);
// 1x1 stride-1 convs with min(M, Nw) <= 128: returns the splits written to p.slab, 0 = not handled
int launch_wgrad_skinny(const WgradParams& p, size_t slab_floats, hipStream_t s);
int igemm_tile_n(int M, bool stem = false);   // pixel-tile width the igemm uses for this M (the stem kernels: 256)
int igemm_tile_m(int M);

// hipFuncSetAttribute with the failure reported (a kernel whose dynamic-LDS limit was not raised fails at its
// first launch; fm_step_* then return FM_ERR_HIP through hipGetLastError)
#include <stdio.h>
inline void set_max_dyn_lds(const void* fn, int bytes, const char* what)
{
    const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc != hipSuccess) fprintf(stderr, "fedmlp_hip: hipFuncSetAttribute(%s, %d B) failed: %s\n", what, bytes, hipGetErrorString(rc));
}
