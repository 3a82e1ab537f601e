// Pointwise (1x1) convolutions of EfficientNet-B0 in bf16 storage on the bf16 matrix pipe, gfx950.
//
// Reference ops replaced: the expand / project / head 1x1 convolutions of efficientnet-pytorch's
// MBConvBlock, their data gradients and their weight gradients, inside net(images) and
// loss.backward() (utils/local_training.py:657, 674, 937-947, 965, 1178, 1191; the model is built at
// model/efficientnet.py:28-33).  This is the bf16 configuration (BASELINE configs[4]); the reference
// itself never enables AMP (utils/local_training.py:14 imports autocast and does not use it), so the
// yardstick is the fp32 oracle with a stated bf16 tolerance.
//
// Roofline: HBM.  EfficientNet-B0 moves ~20 FLOP per byte in bf16 and the bf16 matrix pipe
// is 16x the fp32 one, so these kernels are built to touch every activation byte once:
//  * forward / data gradient (pw_conv_bf16_kernel): the weight slice of an M-tile is staged ONCE per
//    block in LDS, already in MFMA fragment order (each lane's 16 B are contiguous: conflict-free
//    ds_read_b128 with no swizzle); every wave then streams its own pixels: the pixel operand goes
//    global -> VGPR as 16-B fragments (v_mfma_f32_16x16x32_bf16: lane (li, lg) holds x[pixel li][8 lg .. 8 lg+7],
//    i.e. 16 contiguous bytes of an NHWC row), no LDS traffic and no barrier in the loop.  K % 32 == 16
//    (channel counts are padded to 16) ends with one v_mfma_f32_16x16x16_bf16 on 8-B fragments.
//    Output rows are permuted between the two MFMA row tiles of a pair so that a lane ends up with 8
//    consecutive channels of one pixel: one 16-B NHWC store.  Train-mode BN statistics come from the
//    fp32 accumulators (registers, folded once per block, fixed order).  Optional prologue on the pixel
//    operand: BN1 + Swish + squeeze-excite gate applied on load (the gated activation a_s is never
//    written to HBM).
//  * weight gradient (pw_wgrad_bf16_kernel): dW = dY^T X sums over PIXELS, the slow axis of both NHWC
//    operands, so both MFMA operands are "transposed".  Tiles of 32 pixels are staged row-major in LDS
//    and read with ds_read_b64_tr_b16 (hardware transpose: 4 pixels x 16 channels per lane group);
//    row strides are 32 B x odd so that the 8 row segments a half-wave touches fall on distinct banks.
//    Block tile = 64 channels of the larger channel count x ALL of the smaller one (<= 320): the large
//    operand is read exactly once; split over pixels, fp32 slabs, fixed-order reduction (deterministic).
//  * fused backward kernels of the high-resolution MBConv blocks (further down, each with its own header):
//    pw_exp_bwd_kernel -- BN0-backward apply + expand weight gradient + data gradient from one read of (d a_e, y_e);
//    pw_proj_bwd_kernel -- the project conv's backward with the squeeze-excite / BN1 backward around it, in two phases that
//    re-form d a_s = d y_p W on the matrix pipe instead of storing it (7 passes over the depthwise-resolution tensors -> 3).
#include <stdlib.h>

#include <algorithm>

#include "pwconv.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma32(uint4 a, uint4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(uint2 a, uint2 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 lo4(uint4 v) { return __builtin_convertvector(__builtin_bit_cast(bf16x4, make_uint2(v.x, v.y)), f32x4); }
__device__ __forceinline__ f32x4 hi4(uint4 v) { return __builtin_convertvector(__builtin_bit_cast(bf16x4, make_uint2(v.z, v.w)), f32x4); }
__device__ __forceinline__ f32x4 cvt4(uint2 v) { return __builtin_convertvector(__builtin_bit_cast(bf16x4, v), f32x4); }
__device__ __forceinline__ uint2 pack4(f32x4 v) { return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4)); }
__device__ __forceinline__ uint4 pack8(f32x4 a, f32x4 b)
{
    const uint2 x = pack4(a), y = pack4(b);
    return make_uint4(x.x, x.y, y.x, y.y);
}
__device__ __forceinline__ f32x4 swish4(f32x4 v)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] * __builtin_amdgcn_rcpf(1.f + __expf(-v[k]));
    return v;
}
// prologue: Xe = swish(x*sc+sh)*gate (sc == null: x*gate) on 4 channels
__device__ __forceinline__ f32x4 pro4(f32x4 x, const float* sc, const float* sh, const float* gate, bool affine)
{
    if (affine) x = swish4(x * ld4(sc) + ld4(sh));
    return x * ld4(gate);
}

// channel (relative to the M-tile) of MFMA row i of row tile r: tiles of a pair interleave in groups of 4 so
// that accumulator registers of tiles 2u / 2u+1 hold channels 32u + 8 lg + {0..3} / {4..7}
__device__ __forceinline__ int tile_row_to_channel(int r, int i, int npairs)
{
    return r < 2 * npairs ? 32 * (r >> 1) + 8 * (i >> 2) + 4 * (r & 1) + (i & 3) : 16 * r + i;
}

// =====================================================================================================
// P = 16-pixel groups per wave iteration, KB = 32-k chunks requested together.  A wave's loads in flight are
// P x 16 pixels x min(K, 32 KB) x 2 B: with P = 2 a K = 16 layer has ONE KB per wave in flight and the whole kernel
// crawls along its read latency (block 1's expand conv: 2.7 TB/s of a write-dominated 2.9 GB), so small K takes more
// pixels per iteration (accumulators are AGPRs) instead of more k-chunks.
// NW = waves per block (4, or 8 for the K > 640 layers: the weight slice of a 64-channel M-tile is then 86-147 KB of LDS,
// one block per CU, and eight waves share it -- half the M-tiles, i.e. half the re-reads of the pixel operand, of the
// 32-channel tiles two 4-wave blocks would use).
// MODE: 0 plain store (+ residual), 1 train forward (BN partial sums from the accumulators), 2 eval forward (folded BN affine +
// Swish + residual).  A compile-time choice: the running sums (32 registers) and the eval affine (32) were live across the
// pixel loop of EVERY variant (188-244 registers, two waves per SIMD).
template <int RT, bool PRO, int P = 2, int KB = 4, int NW = 4, int MODE = 0>
__global__ __launch_bounds__(64 * NW) void pw_conv_bf16_kernel(const PwParams p)
{
    constexpr int MT = 16 * RT, PPI = 16 * P, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K = p.K, nfull = K >> 5, tail = (K >> 4) & 1;
    uint4* As = reinterpret_cast<uint4*>(smem);                                       // [nfull][RT][64 lanes] x 16 B
    uint2* At = reinterpret_cast<uint2*>(smem + (size_t)nfull * RT * 1024);           // [RT][64 lanes] x 8 B
    float* psc_l = reinterpret_cast<float*>(smem + (size_t)nfull * RT * 1024 + (size_t)RT * 512);   // [K] x 2 (PRO)
    // 1-D grid -> (M-tile bx, pixel range by).  The M-tiles of one pixel range re-read the same pixels: with consecutive
    // block ids dealt round-robin to the 8 XCDs they would land on different L2s, so ids are arranged such that the tiles
    // of a range are congruent mod 8 (same XCD) and close in launch order.
    int bx, by;
    {
        const int id = blockIdx.x, nby = p.nblk * p.groups, tm = p.tiles_m;
        const int full = p.xcd ? (nby >> 3) << 3 : 0;
        if (id < full * tm) {
            const int q = id / (8 * tm), r = id - q * 8 * tm;
            bx = r >> 3;
            by = q * 8 + (r & 7);
        } else {
            const int r = id - full * tm;
            by = full + r / tm;
            bx = r - (r / tm) * tm;
        }
    }
    const int m0 = bx * MT;
    const int grp = by / p.nblk, blk = by - grp * p.nblk;
    const int pb = blk * p.ppb, pe = min(p.npix, pb + p.ppb);
    const size_t gbase = (size_t)grp * p.npix;
    const int nrt = min(RT, (p.M - m0) >> 4);
    const int npairs = nrt >> 1;

    // ---- stage the weight slice once, in fragment order ------------------------------------------
    const int cpr = K >> 3;                                  // 16-B chunks per weight row
    for (int idx = tid; idx < MT * cpr; idx += NT) {
        const int crel = idx / cpr, c = idx - crel * cpr;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m0 + crel < p.M) v = *reinterpret_cast<const uint4*>(p.W + (size_t)(m0 + crel) * K + 8 * c);
        int r, i;
        if (crel < 32 * npairs) {
            const int w = crel & 31;
            r = 2 * (crel >> 5) + ((w >> 2) & 1);
            i = 4 * (w >> 3) + (w & 3);
        } else {
            r = crel >> 4;
            i = crel & 15;
        }
        const int sidx = c >> 2, lgc = c & 3;
        if (sidx < nfull) {
            As[(sidx * RT + r) * 64 + lgc * 16 + i] = v;
        } else {                                             // the 16-k tail: two 4-k halves per chunk
            const int c2 = c - 4 * nfull;
            At[r * 64 + (2 * c2) * 16 + i] = make_uint2(v.x, v.y);
            At[r * 64 + (2 * c2 + 1) * 16 + i] = make_uint2(v.z, v.w);
        }
    }
    if constexpr (PRO) {
        if (p.psc)
            for (int k = tid; k < K; k += NT) {
                psc_l[k] = p.psc[grp * K + k];
                psc_l[K + k] = p.psh[grp * K + k];
            }
    }
    __syncthreads();
    const bool affine = PRO && p.psc != nullptr;

    f32x4 s1[MODE == 1 ? RT : 1], s2[MODE == 1 ? RT : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int r = 0; r < RT; ++r) { s1[r] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[r] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    // eval epilogue: this lane's per-channel affine, loaded once (register r holds the 4 channels of row tile r)
    f32x4 esc[MODE == 2 ? RT : 1], esh[MODE == 2 ? RT : 1];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            esc[r] = f32x4{1.f, 1.f, 1.f, 1.f}; esh[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < nrt) {
                const int m = m0 + tile_row_to_channel(r, 4 * lg, npairs);
                esc[r] = ld4(p.scale + m); esh[r] = ld4(p.shift + m);
            }
        }
    }

    const int n_iter = (pe - pb + PPI - 1) / PPI;
    for (int it = wave; it < n_iter; it += NW) {
        const int pix0 = pb + it * PPI;
        bool pv[P];
        const bf16* xp[P];
        const float* gp[P];
#pragma unroll
        for (int g = 0; g < P; ++g) {
            const int px = pix0 + 16 * g + li;
            pv[g] = px < pe;
            const size_t gpx = gbase + (pv[g] ? px : pb);
            xp[g] = p.X + gpx * K;
            gp[g] = PRO ? p.gate + (gpx / p.HW) * K : nullptr;
        }
        f32x4 acc[RT][P];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int g = 0; g < P; ++g) acc[r][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nfull; kb += KB) {
            uint4 b[P][KB];
#pragma unroll
            for (int g = 0; g < P; ++g)
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    b[g][k] = (pv[g] && kb + k < nfull) ? *reinterpret_cast<const uint4*>(xp[g] + (kb + k) * 32 + 8 * lg)
                                                       : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRO) {
#pragma unroll
                for (int g = 0; g < P; ++g)
#pragma unroll
                    for (int k = 0; k < KB; ++k) {
                        if (!pv[g] || kb + k >= nfull) continue;
                        const int kk = (kb + k) * 32 + 8 * lg;
                        const f32x4 lo = pro4(lo4(b[g][k]), psc_l + kk, psc_l + K + kk, gp[g] + kk, affine);
                        const f32x4 hi = pro4(hi4(b[g][k]), psc_l + kk + 4, psc_l + K + kk + 4, gp[g] + kk + 4, affine);
                        b[g][k] = pack8(lo, hi);
                    }
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (kb + k >= nfull) break;
                const uint4* A = As + (size_t)(kb + k) * RT * 64 + lane;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (r < nrt) {                           // row tiles past M (no `break`: a loop with an early exit is fully
                        const uint4 a = A[r * 64];           // unrolled only up to 8 trips, beyond that acc[] lands in scratch)
#pragma unroll
                        for (int g = 0; g < P; ++g) acc[r][g] = mfma32(a, b[g][k], acc[r][g]);
                    }
                }
            }
        }
        if (tail) {
            uint2 b4[P];
#pragma unroll
            for (int g = 0; g < P; ++g) {
                b4[g] = pv[g] ? *reinterpret_cast<const uint2*>(xp[g] + nfull * 32 + 4 * lg) : make_uint2(0u, 0u);
                if constexpr (PRO) {
                    if (pv[g]) {
                        const int kk = nfull * 32 + 4 * lg;
                        b4[g] = pack4(pro4(cvt4(b4[g]), psc_l + kk, psc_l + K + kk, gp[g] + kk, affine));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (r < nrt) {
                    const uint2 a = At[r * 64 + lane];
#pragma unroll
                    for (int g = 0; g < P; ++g) acc[r][g] = mfma16(a, b4[g], acc[r][g]);
                }
            }
        }
        // ---- epilogue of the iteration: acc[r][g][q] = D[channel(r, 4 lg + q)][pixel pix0 + 16 g + li] -----
        if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int g = 0; g < P; ++g) { s1[r] += acc[r][g]; s2[r] += acc[r][g] * acc[r][g]; }   // padded pixels are exact zeros
        }
#pragma unroll
        for (int g = 0; g < P; ++g) {
            if (!pv[g]) continue;
            const size_t o = (gbase + pix0 + 16 * g + li) * (size_t)p.M;
            auto finish = [&](f32x4 v, int m, int r) {
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int rr = 0; rr < RT; ++rr)
                        if (rr == r) v = v * esc[rr] + esh[rr];          // static register index
                }
                if (p.res) v += ld4(p.res + o + m);
                if constexpr (MODE == 2) { if (p.act == 2) v = swish4(v); }
                return v;
            };
#pragma unroll
            for (int u = 0; u < RT / 2; ++u) {
                if (u < npairs) {
                    const int m = m0 + 32 * u + 8 * lg;
                    *reinterpret_cast<uint4*>(p.Y + o + m) = pack8(finish(acc[2 * u][g], m, 2 * u), finish(acc[2 * u + 1][g], m + 4, 2 * u + 1));
                }
            }
            if (nrt & 1) {
                const int r = nrt - 1;
                const int m = m0 + 16 * r + 4 * lg;
                f32x4 v = acc[0][g];
#pragma unroll
                for (int rr = 1; rr < RT; ++rr)
                    if (rr == r) v = acc[rr][g];             // static register index
                *reinterpret_cast<uint2*>(p.Y + o + m) = pack4(finish(v, m, r));
            }
        }
    }

    // ---- BN statistics: fold the 16 pixel lanes, then the 4 waves (fixed order) -----------------------
    if constexpr (MODE == 1) {
        __syncthreads();                       // every wave is done with the weight slice: reuse the LDS
        float* red = reinterpret_cast<float*>(smem);          // [NW waves][MT][2]
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u = s1[r][q], v = s2[r][q];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    u += __shfl_xor(u, d);
                    v += __shfl_xor(v, d);
                }
                if (li == 0 && r < nrt) {
                    const int ml = tile_row_to_channel(r, 4 * lg + q, npairs);
                    red[(wave * MT + ml) * 2 + 0] = u;
                    red[(wave * MT + ml) * 2 + 1] = v;
                }
            }
        __syncthreads();
        if (tid < 16 * nrt) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                u += red[(w * MT + tid) * 2 + 0];
                v += red[(w * MT + tid) * 2 + 1];
            }
            float* st = p.stats + (size_t)(grp * p.nblk + blk) * 2 * p.M;
            st[m0 + tid] = u;
            st[p.M + m0 + tid] = v;
        }
    }
}

// =====================================================================================================
// LDS-tiled GEMM form of the same convolution, used for the DATA GRADIENTS with K >= 64 (expand / project / head convs of the
// low-resolution blocks: few pixels, long K; launch_pw_gemm says what was measured for the forward forms).  The streaming kernel
// above keeps a whole 64-channel weight slice in LDS (86-147 KB, one block per CU) and every wave waits out the full load
// latency of each of its pixel groups.  Here a block owns a tile of
// BP pixels x BM channels and walks K in 64-k stages: both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4,
// a whole 128-B line per row and stage, no registers), NS stages deep, counted s_waitcnt vmcnt + one raw s_barrier per
// stage (the structure of igemm.hip).  The LDS image is lane-linear; the bank swizzle slot = chunk ^ ((row >> 1) & 7) is
// applied on the DMA's SOURCE address and on the fragment reads (conflict-free ds_read_b128, brute-force checked).  The four
// waves split the PIXELS (each reads its own pixel rows once, all of them read every weight row), so the accumulators
// come out exactly as in the streaming kernel: lane (li, lg) holds 4 (paired tiles: 8) consecutive channels of pixel li,
// one 8- / 16-B NHWC store, BN partial sums from the fp32 accumulators.  Chunks past K and rows past M / past the group's
// pixels are DMA'd from a zero page.
// PRO: operand prologue of the project convs (PwParams::gate): 1 = Xe = X * gate[img][k] (eval forward), 2 = Xe = swish(X * psc[k]
// + psh[k]) * gate[img][k] (train forward: BN1 + Swish + squeeze-excite gate; a_s is never written).  The gate rows of the (at most
// two: HW >= BP) images of the tile ride along as one more DMA per stage ([2][64 k] fp32); psc / psh are staged once per block.  The
// waves split the pixels, so every operand element is transformed exactly ONCE (the streaming kernel redoes it per 64-channel tile).
template <int BM, int BP, int NS, int MODE, int PRO = 0>
__global__ __launch_bounds__(256) void pw_gemm_bf16_kernel(const PwParams p)
{
#if __HIP_DEVICE_COMPILE__
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int FR = BM / 16, FC = BP / 64;               // MFMA tiles per wave: all channel tiles x its pixel tiles
    constexpr int STG_W = BM * 128, STG_X = BP * 128;       // bytes per stage (64 k x 2 B per row)
    constexpr int GW = BM / 32, GX = BP / 32;               // DMA instructions per wave and stage (8 rows each)
    constexpr int PER = GW + GX + (PRO ? 1 : 0), D = NS - 1;   // (+ wave 0..3 each issue the gate DMA: counted alike in every wave)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ws = smem;
    unsigned char* Xs = smem + NS * STG_W;
    float* Gs = reinterpret_cast<float*>(smem + NS * (STG_W + STG_X));                 // [NS][4 waves][1 KB: 2 images x 64 k fp32 + the DMA's unused upper half] (PRO)
    float* Ps = reinterpret_cast<float*>(smem + NS * (STG_W + STG_X) + NS * 4 * 1024); // [2][K] psc, psh (PRO == 2)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int K = p.K;
    int bx, by;                                             // M-tile, (group, pixel tile): the M-tiles of one pixel tile on ONE XCD
    {
        const int id = blockIdx.x, nby = p.nblk * p.groups, tm = p.tiles_m;
        const int full = p.xcd ? (nby >> 3) << 3 : 0;
        if (id < full * tm) {
            const int q = id / (8 * tm), r = id - q * 8 * tm;
            bx = r >> 3;
            by = q * 8 + (r & 7);
        } else {
            const int r = id - full * tm;
            by = full + r / tm;
            bx = r - (r / tm) * tm;
        }
    }
    const int m0 = bx * BM;
    const int grp = by / p.nblk, tn = by - grp * p.nblk;
    const int n0 = tn * BP;
    const size_t gbase = (size_t)grp * p.npix;
    const int nrt = min(FR, (p.M - m0) >> 4);
    const int npairs = nrt >> 1;
    const int nst = (K + 63) >> 6;

    // ---- LDS-DMA source state: lane = (row of an 8-row group, 16-B slot); slot s of row rho holds chunk s ^ ((rho >> 1) & 7)
    const int drow = lane >> 3, dslot = lane & 7;
    const bf16* wsrc[GW];
    const bf16* xsrc[GX];
    int wk[GW], xk[GX];                                     // element offset of this lane's chunk inside a stage, or -1 (zero page)
#pragma unroll
    for (int q = 0; q < GW; ++q) {
        const int rho = 8 * (wave * GW + q) + drow, r = rho >> 4;
        const bool ok = r < nrt;
        wsrc[q] = p.W + (size_t)(ok ? m0 + tile_row_to_channel(r, rho & 15, npairs) : 0) * K;
        wk[q] = ok ? (dslot ^ ((rho >> 1) & 7)) * 8 : -1;
    }
#pragma unroll
    for (int q = 0; q < GX; ++q) {
        const int rho = 8 * (wave * GX + q) + drow;
        const bool ok = n0 + rho < p.npix;
        xsrc[q] = p.X + (gbase + (ok ? n0 + rho : 0)) * (size_t)K;
        xk[q] = ok ? (dslot ^ ((rho >> 1) & 7)) * 8 : -1;
    }
    const bf16* zeros = reinterpret_cast<const bf16*>(p.zeros);
    // PRO: this WAVE's pixels span at most two images (HW >= BP / 4 is all that needs); lanes 0-31 stage [2][64 k] of their gate rows
    const float* gsrc = nullptr;
    int gk = -1, img0w = 0;
    if constexpr (PRO != 0) {
        const int pw0 = n0 + wave * (BP / 4);
        const long long gp0 = (long long)gbase + min(pw0, p.npix - 1), gp1 = (long long)gbase + min(pw0 + BP / 4 - 1, p.npix - 1);
        img0w = (int)(gp0 / p.HW);
        const int img1w = (int)(gp1 / p.HW);
        const int gi = (lane >> 4) & 1;
        const bool ok = lane < 32 && img0w + gi <= img1w;
        gsrc = p.gate + (size_t)(img0w + (ok ? gi : 0)) * K;
        gk = ok ? 4 * (lane & 15) : -1;
        if constexpr (PRO == 2) {
            for (int k = tid; k < K; k += 256) { Ps[k] = p.psc[grp * K + k]; Ps[K + k] = p.psh[grp * K + k]; }
            __syncthreads();
        }
    }
    auto issue = [&](int s) {
        const int slot = s % NS, k0 = s * 64;
        if constexpr (PRO != 0) {
            const int k = k0 + gk;
            __builtin_amdgcn_global_load_lds((gk >= 0 && k < K) ? reinterpret_cast<const bf16*>(gsrc + k) : zeros,
                                             (lds_void*)(Gs + (slot * 4 + wave) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < GW; ++q) {
            const int k = k0 + wk[q];
            __builtin_amdgcn_global_load_lds((wk[q] >= 0 && k < K) ? wsrc[q] + k : zeros,
                                             (lds_void*)(Ws + slot * STG_W + (wave * GW + q) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < GX; ++q) {
            const int k = k0 + xk[q];
            __builtin_amdgcn_global_load_lds((xk[q] >= 0 && k < K) ? xsrc[q] + k : zeros,
                                             (lds_void*)(Xs + slot * STG_X + (wave * GX + q) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[FR][FC];
#pragma unroll
    for (int r = 0; r < FR; ++r)
#pragma unroll
        for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
#pragma unroll
    for (int d = 1; d < D; ++d)
        if (d < nst) issue(d);
    const int fsw = (li >> 1) & 7;
    int pimg[FC];                                           // PRO: image (0 / 1, relative to the wave's first) and validity of
    bool pval[FC];                                          // this lane's pixel of each pixel tile
#pragma unroll
    for (int c = 0; c < FC; ++c) {
        const int px = n0 + wave * (BP / 4) + 16 * c + li;
        pval[c] = px < p.npix;
        pimg[c] = PRO ? (int)(((long long)gbase + min(px, p.npix - 1)) / p.HW) - img0w : 0;
    }
    const unsigned char* arow = Ws + li * 128;
    const unsigned char* brow = Xs + (wave * (BP / 4) + li) * 128;
    for (int s = 0; s < nst; ++s) {
        const int younger = min(D - 1, nst - 1 - s);        // my DMA of stage s has landed once only younger stages are outstanding
        if (D >= 3 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (D >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // everyone's stage-s rows landed; everyone finished reading stage s - 1
        asm volatile("" ::: "memory");
        if (s + D < nst) issue(s + D);                      // refill the slot stage s - 1 just vacated
        const int slot = s % NS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (s * 64 + 32 * h >= K) break;                // K % 64 <= 32: the stage's second half is all zero page
            const int so = ((4 * h + lg) ^ fsw) * 16;
            uint4 b[FC];
#pragma unroll
            for (int c = 0; c < FC; ++c) b[c] = *reinterpret_cast<const uint4*>(brow + slot * STG_X + c * 16 * 128 + so);
            if constexpr (PRO != 0) {
                const int kk = 32 * h + 8 * lg;                                 // k inside the stage of this lane's 8 values
#pragma unroll
                for (int c = 0; c < FC; ++c) {
                    const float* gr = Gs + (slot * 4 + wave) * 256 + pimg[c] * 64 + kk;
                    f32x4 lo = lo4(b[c]), hi = hi4(b[c]);
                    if constexpr (PRO == 2) {
                        const float* ps = Ps + s * 64 + kk;
                        const bool kv = s * 64 + kk < K;                        // (chunks past K: the zero page must stay zero)
                        lo = swish4(lo * ld4(ps) + ld4(ps + K)); hi = swish4(hi * ld4(ps + 4) + ld4(ps + K + 4));
                        if (!kv || !pval[c]) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = lo; }
                    }
                    b[c] = pack8(lo * ld4(gr), hi * ld4(gr + 4));
                }
            }
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                if (r < nrt) {
                    const uint4 a = *reinterpret_cast<const uint4*>(arow + slot * STG_W + r * 16 * 128 + so);
#pragma unroll
                    for (int c = 0; c < FC; ++c) acc[r][c] = mfma32(a, b[c], acc[r][c]);
                }
            }
        }
    }

    // ---- epilogue: acc[r][c][q] = D[channel(r, 4 lg + q)][pixel n0 + wave BP/4 + 16 c + li] --------------------------------
    if constexpr (MODE == 1) {
        __syncthreads();                                    // every wave is done with the stages: reuse the LDS
        float* red = reinterpret_cast<float*>(smem);       // [4 waves][BM][2]
#pragma unroll
        for (int r = 0; r < FR; ++r) {
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < FC; ++c) { s1 += acc[r][c]; s2 += acc[r][c] * acc[r][c]; }     // padded pixels are exact zeros
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u = s1[q], v = s2[q];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    u += __shfl_xor(u, d);
                    v += __shfl_xor(v, d);
                }
                if (li == 0 && r < nrt) {
                    const int ml = tile_row_to_channel(r, 4 * lg + q, npairs);
                    red[(wave * BM + ml) * 2 + 0] = u;
                    red[(wave * BM + ml) * 2 + 1] = v;
                }
            }
        }
        __syncthreads();
        if (tid < 16 * nrt) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                u += red[(w * BM + tid) * 2 + 0];
                v += red[(w * BM + tid) * 2 + 1];
            }
            float* st = p.stats + (size_t)(grp * p.nblk + tn) * 2 * p.M;
            st[m0 + tid] = u;
            st[p.M + m0 + tid] = v;
        }
    }
#pragma unroll
    for (int c = 0; c < FC; ++c) {
        const int px = n0 + wave * (BP / 4) + 16 * c + li;
        if (px >= p.npix) continue;
        const size_t o = (gbase + px) * (size_t)p.M;
        auto finish = [&](f32x4 v, int m) {
            if constexpr (MODE == 2) v = v * ld4(p.scale + m) + ld4(p.shift + m);
            if (p.res) v += ld4(p.res + o + m);
            if constexpr (MODE == 2) { if (p.act == 2) v = swish4(v); }
            return v;
        };
#pragma unroll
        for (int u = 0; u < FR / 2; ++u) {
            if (u < npairs) {
                const int m = m0 + 32 * u + 8 * lg;
                *reinterpret_cast<uint4*>(p.Y + o + m) = pack8(finish(acc[2 * u][c], m), finish(acc[2 * u + 1][c], m + 4));
            }
        }
        if (nrt & 1) {
            const int r = nrt - 1;
            const int m = m0 + 16 * r + 4 * lg;
            f32x4 v = acc[0][c];
#pragma unroll
            for (int rr = 1; rr < FR; ++rr)
                if (rr == r) v = acc[rr][c];                // static register index
            *reinterpret_cast<uint2*>(p.Y + o + m) = pack4(finish(v, m));
        }
    }
#endif
}

// =====================================================================================================
// weight gradient.  P[l][s] = sum_pix Big[pix][l] * Small[pix][s]; `swap` = Big is X (then P = dW^T).
struct WgArgs {
    const bf16* Big;      // [npix][L]
    const bf16* Small;    // [npix][S]
    float* slab;          // [splits][M][K]
    int L, S, M, K, swap, npix;
    int strideS;          // LDS row stride of the Small tile in bytes (32 x odd, >= 2 S)
    const float* psc;     // prologue on X (the Big operand if swap, else the Small one)
    const float* psh;
    const float* gate;
    int HW, pix_per_group;
    int tilesL, nsplit, xcd;   // launch geometry (1-D grid)
};
constexpr int WG_STRIDE_BIG = 160;          // 64 channels x 2 B = 128 B of data per pixel row, padded to 32 B x 5

__device__ __forceinline__ uint2 ds_read_tr16(const unsigned char* lds_ptr)
{
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    // generic -> LDS address space: the low 32 bits of a generic LDS pointer are the LDS offset
    const unsigned off = (unsigned)(size_t)lds_ptr;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)off);
    return __builtin_bit_cast(uint2, v);
}

template <int CC, bool PRO>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(const WgArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // 1-D grid -> (64-channel tile of Big, pixel split).  The tiles of a split all re-read the split's Small rows: block
    // ids are arranged so that they are congruent mod 8 = on ONE XCD's L2 (p.xcd; same scheme as pw_conv_bf16_kernel)
    int tile, split;
    const int nsplit = p.nsplit;
    {
        const int T = p.tilesL, id = blockIdx.x;
        const int full = p.xcd ? (nsplit >> 3) << 3 : 0;
        if (id < full * T) {
            const int q = id >> 3, x = id & 7;
            split = (q / T) * 8 + x;
            tile = q - (q / T) * T;
        } else {
            const int r = id - full * T;
            split = full + r / T;
            tile = r - (r / T) * T;
        }
    }
    const int l0 = tile * 64;
    const int tsteps = (p.npix + 31) >> 5;
    const int nsteps = split < tsteps ? (tsteps - split + nsplit - 1) / nsplit : 0;   // 32-pixel steps dealt round-robin
    const int strideS = p.strideS;
    const int bufbytes = 32 * WG_STRIDE_BIG + 32 * strideS;
    const int cps = p.S >> 3;                                  // 16-B chunks per Small row
    constexpr int NB = (32 * (2 * CC) + 255) / 256;            // Small-tile chunks per thread (S = 16 CC -> 2 CC chunks per row)

    // staging: Big tile = 32 rows x 8 chunks (one per thread), Small tile = 32 rows x cps chunks
    const int ra = tid >> 3, ca = tid & 7;
    const bool a_ok = l0 + 8 * ca < p.L;
    uint4 va, vb[NB];
    auto xform = [&](uint4 v, int pix, int ch) {            // prologue on 8 channels ch.. of pixel pix of X
        const int g = pix / p.pix_per_group;
        const float* gt = p.gate + (size_t)(pix / p.HW) * p.K + ch;
        const bool affine = p.psc != nullptr;
        const float* sc = affine ? p.psc + g * p.K + ch : nullptr;
        const float* sh = affine ? p.psh + g * p.K + ch : nullptr;
        const f32x4 lo = pro4(lo4(v), sc, sh, gt, affine);
        const f32x4 hi = pro4(hi4(v), affine ? sc + 4 : nullptr, affine ? sh + 4 : nullptr, gt + 4, affine);
        return pack8(lo, hi);
    };
    auto gload = [&](int s) {
        const int pb = (split + s * nsplit) * 32;
        {
            const int pix = pb + ra;
            va = make_uint4(0u, 0u, 0u, 0u);
            if (a_ok && pix < p.npix) {
                va = *reinterpret_cast<const uint4*>(p.Big + (size_t)pix * p.L + l0 + 8 * ca);
                if constexpr (PRO) { if (p.swap) va = xform(va, pix, l0 + 8 * ca); }
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / cps, cb = c - row * cps;
            const int pix = pb + row;
            vb[q] = make_uint4(0u, 0u, 0u, 0u);
            if (c < 32 * cps && pix < p.npix) {
                vb[q] = *reinterpret_cast<const uint4*>(p.Small + (size_t)pix * p.S + 8 * cb);
                if constexpr (PRO) { if (!p.swap) vb[q] = xform(vb[q], pix, 8 * cb); }
            }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* base = smem + buf * bufbytes;
        *reinterpret_cast<uint4*>(base + ra * WG_STRIDE_BIG + ca * 16) = va;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / cps, cb = c - row * cps;
            if (c < 32 * cps) *reinterpret_cast<uint4*>(base + 32 * WG_STRIDE_BIG + row * strideS + cb * 16) = vb[q];
        }
    };

    f32x4 acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();
    // transposed fragment reads: lane (group lg, i = li) supplies the address of row (4 lg + (li >> 2)) [+16 for
    // the second half of k], columns 4 (li & 3) .. +3 of the 16-channel block, and receives channel li of
    // rows 4 lg .. 4 lg + 3: element j < 4 of a fragment is pixel 4 lg + j, j >= 4 pixel 16 + 4 lg + (j - 4)
    // (the same permutation of k for both operands)
    const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const unsigned char* base = smem + buf * bufbytes;
        const unsigned char* ab = base + trow * WG_STRIDE_BIG + (16 * wave + tcol) * 2;
        const uint2 a0 = ds_read_tr16(ab), a1 = ds_read_tr16(ab + 16 * WG_STRIDE_BIG);
        const uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y);
        const unsigned char* bb = base + 32 * WG_STRIDE_BIG + trow * strideS + tcol * 2;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const uint2 b0 = ds_read_tr16(bb + 32 * c), b1 = ds_read_tr16(bb + 32 * c + 16 * strideS);
            acc[c] = mfma32(a, make_uint4(b0.x, b0.y, b1.x, b1.y), acc[c]);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    // acc[c][q] = P[l = l0 + 16 wave + 4 lg + q][s = 16 c + li]
    const int l = l0 + 16 * wave + 4 * lg;
    if (l < p.L) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int sc = 16 * c + li;
            if (sc >= p.S) continue;
            if (p.swap) {            // l = k (X channel), s = m: 4 consecutive k -> one 16-B store
                *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + sc) * p.K + l) = acc[c];
            } else {                 // l = m, s = k
#pragma unroll
                for (int q = 0; q < 4; ++q) p.slab[((size_t)split * p.M + l + q) * p.K + sc] = acc[c][q];
            }
        }
    }
}

// ---- weight gradient of the small high-resolution layers: one wave per pixel tile, no block barrier -----------------
// With L <= 144 and S <= 32 channels (blocks 0-3: 16x32 ... 32x144 weights over 3-13 M pixels) the kernel above moves a
// 32-pixel tile per step through a block-wide barrier pair and gives each wave ONE to two MFMAs (half the waves none when
// L = 32): it ran at 1.4-1.7 TB/s.  Here every wave owns its own pixel tiles and ALL L x S of the product: it stages a
// 32-pixel tile of both operands in a wave-private LDS region (no __syncthreads: a wave's LDS operations execute in order,
// so the next tile is written right after the transposed reads of the current one were issued), its global loads for
// tile s+1 fly during the MFMAs of tile s, and it leaves its own slab (4 x blocks slabs, fixed-order reduction as before).
template <int NRT, int CC, bool PRO>
__global__ __launch_bounds__(256) void pw_wgrad_wave_kernel(const WgArgs p, int strideB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int strideS = p.strideS;
    unsigned char* base = smem + wave * 32 * (strideB + strideS);
    unsigned char* sbase = base + 32 * strideB;
    const int ws = blockIdx.x * 4 + wave, nws = gridDim.x * 4;
    const int tsteps = (p.npix + 31) >> 5;
    const int nsteps = ws < tsteps ? (tsteps - ws + nws - 1) / nws : 0;
    constexpr int CPB = 2 * NRT, CPS = 2 * CC;              // 16-B chunks per row of the Big / Small tile
    constexpr int NBG = (32 * CPB + 63) / 64, NSM = (32 * CPS + 63) / 64;
    const int L = 16 * NRT, S = 16 * CC;
    uint4 vb[NBG], vs[NSM];
    auto xform = [&](uint4 v, int pix, int ch) {
        const int g = pix / p.pix_per_group;
        const float* gt = p.gate + (size_t)(pix / p.HW) * p.K + ch;
        const bool affine = p.psc != nullptr;
        const float* sc = affine ? p.psc + g * p.K + ch : nullptr;
        const float* sh = affine ? p.psh + g * p.K + ch : nullptr;
        const f32x4 lo = pro4(lo4(v), sc, sh, gt, affine);
        const f32x4 hi = pro4(hi4(v), affine ? sc + 4 : nullptr, affine ? sh + 4 : nullptr, gt + 4, affine);
        return pack8(lo, hi);
    };
    auto gload = [&](int s) {
        const int pb = (ws + s * nws) * 32;
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPB, cb = c - row * CPB;
            const int pix = pb + row;
            vb[q] = make_uint4(0u, 0u, 0u, 0u);
            if (c < 32 * CPB && pix < p.npix) {
                vb[q] = *reinterpret_cast<const uint4*>(p.Big + (size_t)pix * L + 8 * cb);
                if constexpr (PRO) { if (p.swap) vb[q] = xform(vb[q], pix, 8 * cb); }
            }
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPS, cb = c - row * CPS;
            const int pix = pb + row;
            vs[q] = make_uint4(0u, 0u, 0u, 0u);
            if (c < 32 * CPS && pix < p.npix) {
                vs[q] = *reinterpret_cast<const uint4*>(p.Small + (size_t)pix * S + 8 * cb);
                if constexpr (PRO) { if (!p.swap) vs[q] = xform(vs[q], pix, 8 * cb); }
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPB, cb = c - row * CPB;
            if (c < 32 * CPB) *reinterpret_cast<uint4*>(base + row * strideB + cb * 16) = vb[q];
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPS, cb = c - row * CPS;
            if (c < 32 * CPS) *reinterpret_cast<uint4*>(sbase + row * strideS + cb * 16) = vs[q];
        }
    };
    f32x4 acc[NRT][CC];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) { gload(0); lstore(); }
    const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) gload(s + 1);
        uint4 b[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const unsigned char* bb = sbase + trow * strideS + (16 * c + tcol) * 2;
            const uint2 b0 = ds_read_tr16(bb), b1 = ds_read_tr16(bb + 16 * strideS);
            b[c] = make_uint4(b0.x, b0.y, b1.x, b1.y);
        }
#pragma unroll
        for (int r = 0; r < NRT; ++r) {
            const unsigned char* ab = base + trow * strideB + (16 * r + tcol) * 2;
            const uint2 a0 = ds_read_tr16(ab), a1 = ds_read_tr16(ab + 16 * strideB);
            const uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[r][c] = mfma32(a, b[c], acc[r][c]);
        }
        if (s + 1 < nsteps) lstore();
    }
    // acc[r][c][q] = P[l = 16 r + 4 lg + q][s = 16 c + li]; this wave's slab (zeros when it had no tile)
    float* slab = p.slab + (size_t)ws * p.M * p.K;
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
        const int l = 16 * r + 4 * lg;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int sc = 16 * c + li;
            if (p.swap) {
                *reinterpret_cast<f32x4*>(slab + (size_t)sc * p.K + l) = acc[r][c];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) slab[(size_t)(l + q) * p.K + sc] = acc[r][c][q];
            }
        }
    }
}


// =====================================================================================================
// Fused backward of an expand (1x1) convolution with its BatchNorm + Swish, for the early high-resolution MBConv blocks
// (ce = L <= 144, cin = S <= 32: 62 % of the network's expanded-tensor bytes).  The unfused order moves the expanded gradient
// five times -- BN-backward apply (read d a_e and y_e, write d y_e), weight gradient (read d y_e), data gradient (read d y_e) --
// here d a_e and y_e are read ONCE: a wave owns 32-pixel tiles, forms
//     d y_e = ca * (d a_e * swish'(y_e * sc + sh)) + cb * y_e + cc        (bnact_bwd_apply_kernel's arithmetic)
// in registers, rounds it to bf16 (the value the unfused path would have stored) into a wave-private LDS tile and feeds two
// MFMA products from it:  dW[l][s] += sum_pix dy[pix][l] x[pix][s]  (transposed reads, as pw_wgrad_wave_kernel) and
// dX[pix][s] = sum_l dy[pix][l] W[l][s] (+ residual)  (plain row reads of the tile against W^T rows held in LDS by the block).
struct ExpBwdArgs {
    const bf16 *dA, *Ye, *X, *Wt, *res;      // [npix][L], [npix][L], [npix][S], [S][L], [npix][S] or null
    bf16* dX;                                // [npix][S]
    float* slab;                             // [waves][L][S] fp32 partial dW (conv weight layout [M = L][K = S])
    const float *ca, *cb, *cc, *sc, *sh;     // [groups][L]
    int npix, pix_per_group, groups;
    int strideB, strideS, strideW;           // LDS row strides in bytes (32 x odd)
};
template <int NRT, int CC>
__global__ __launch_bounds__(256, 2) void pw_exp_bwd_kernel(const ExpBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int L = 16 * NRT, S = 16 * CC;
    constexpr int CPB = 2 * NRT, CPS = 2 * CC;              // 16-B chunks per row of the L-wide / S-wide tiles
    constexpr int NBG = (32 * CPB + 63) / 64, NSM = (32 * CPS + 63) / 64;
    constexpr int NFULL = L / 32, TAIL = (L / 16) & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int strideB = p.strideB, strideS = p.strideS, strideW = p.strideW;
    unsigned char* wt = smem;                                           // [S][strideW]
    float* vec = reinterpret_cast<float*>(smem + S * strideW);          // [groups][5][L]: ca, cb, cc, sc, sh
    unsigned char* base = smem + S * strideW + (size_t)p.groups * 5 * L * 4 + (size_t)wave * 32 * (strideB + strideS);
    unsigned char* sbase = base + 32 * strideB;
    for (int c = tid; c < S * CPB; c += 256) {
        const int row = c / CPB, cb = c - row * CPB;
        *reinterpret_cast<uint4*>(wt + row * strideW + cb * 16) = *reinterpret_cast<const uint4*>(p.Wt + (size_t)row * L + 8 * cb);
    }
    for (int i = tid; i < p.groups * L; i += 256) {
        const int g = i / L, l = i - g * L;
        float* v = vec + (size_t)g * 5 * L;
        v[l] = p.ca[i]; v[L + l] = p.cb[i]; v[2 * L + l] = p.cc[i]; v[3 * L + l] = p.sc[i]; v[4 * L + l] = p.sh[i];
    }
    __syncthreads();
    const int ws = blockIdx.x * 4 + wave, nws = gridDim.x * 4;
    const int tsteps = (p.npix + 31) >> 5;
    const int nsteps = ws < tsteps ? (tsteps - ws + nws - 1) / nws : 0;
    uint4 va[NBG], vy[NBG], vs[NSM];
    // a 32-pixel tile of an [npix][L] tensor is one contiguous run of 32 * 2L bytes: chunk c = lane + 64 q of the tile sits at
    // tile base + 16 c (one base address per tensor and step; npix % 32 == 0 is the launcher's precondition)
    auto gload = [&](int st) {
        const size_t pb = (size_t)(ws + st * nws) * 32;
        const uint4* ta = reinterpret_cast<const uint4*>(p.dA + pb * L) + lane;
        const uint4* ty = reinterpret_cast<const uint4*>(p.Ye + pb * L) + lane;
        const uint4* tx = reinterpret_cast<const uint4*>(p.X + pb * S) + lane;
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            if (64 * q + 63 < 32 * CPB || lane + 64 * q < 32 * CPB) { va[q] = ta[64 * q]; vy[q] = ty[64 * q]; }
            else { va[q] = make_uint4(0u, 0u, 0u, 0u); vy[q] = va[q]; }
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            if (64 * q + 63 < 32 * CPS || lane + 64 * q < 32 * CPS) vs[q] = tx[64 * q];
            else vs[q] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // BN-backward apply on 4 channels: d = dA * swish'(y*sc+sh); ca*d + cb*y + cc   (fast exp / rcp like the bf16 passes)
    auto bn4 = [&](f32x4 d, f32x4 y, const float* v) {
        const f32x4 u = y * ld4(v + 3 * L) + ld4(v + 4 * L);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-u[k]));
            d[k] *= sg * (1.f + u[k] * (1.f - sg));
        }
        return ld4(v) * d + ld4(v + L) * y + ld4(v + 2 * L);
    };
    auto lstore = [&](int st) {
        const int pb = (ws + st * nws) * 32;
        const float* vg = vec + (size_t)(pb / p.pix_per_group) * 5 * L;         // a tile lies inside one statistics group
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPB, cb = c - row * CPB;
            if (c < 32 * CPB)
                *reinterpret_cast<uint4*>(base + row * strideB + cb * 16) =
                    pack8(bn4(lo4(va[q]), lo4(vy[q]), vg + 8 * cb), bn4(hi4(va[q]), hi4(vy[q]), vg + 8 * cb + 4));
        }
#pragma unroll
        for (int q = 0; q < NSM; ++q) {
            const int c = lane + 64 * q;
            const int row = c / CPS, cb = c - row * CPS;
            if (c < 32 * CPS) *reinterpret_cast<uint4*>(sbase + row * strideS + cb * 16) = vs[q];
        }
    };
    f32x4 acc[NRT][CC];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) { gload(0); lstore(0); }
    const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
    for (int st = 0; st < nsteps; ++st) {
        if (st + 1 < nsteps) gload(st + 1);
        // ---- weight gradient: P[l][s] += sum_pix dy[pix][l] x[pix][s] (transposed fragments of both tiles) ----
        uint4 b[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const unsigned char* bb = sbase + trow * strideS + (16 * c + tcol) * 2;
            const uint2 b0 = ds_read_tr16(bb), b1 = ds_read_tr16(bb + 16 * strideS);
            b[c] = make_uint4(b0.x, b0.y, b1.x, b1.y);
        }
#pragma unroll
        for (int r = 0; r < NRT; ++r) {
            const unsigned char* ab = base + trow * strideB + (16 * r + tcol) * 2;
            const uint2 a0 = ds_read_tr16(ab), a1 = ds_read_tr16(ab + 16 * strideB);
            const uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[r][c] = mfma32(a, b[c], acc[r][c]);
        }
        // ---- data gradient: D[s][pix] = sum_l W^T[s][l] dy[pix][l]; lane (li, lg) ends with 4 consecutive s of pixel li ----
        const int pb = (ws + st * nws) * 32;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            f32x4 dx[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) dx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* yrow = base + (16 * pt + li) * strideB;
#pragma unroll
            for (int kc = 0; kc < NFULL; ++kc) {
                const uint4 bf = *reinterpret_cast<const uint4*>(yrow + (32 * kc + 8 * lg) * 2);
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    dx[c] = mfma32(*reinterpret_cast<const uint4*>(wt + (16 * c + li) * strideW + (32 * kc + 8 * lg) * 2), bf, dx[c]);
            }
            if (TAIL) {
                const uint2 bf = *reinterpret_cast<const uint2*>(yrow + (32 * NFULL + 4 * lg) * 2);
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    dx[c] = mfma16(*reinterpret_cast<const uint2*>(wt + (16 * c + li) * strideW + (32 * NFULL + 4 * lg) * 2), bf, dx[c]);
            }
            const size_t o0 = ((size_t)pb + 16 * pt + li) * S + 4 * lg;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                f32x4 v = dx[c];
                if (p.res) v += ld4(p.res + o0 + 16 * c);
                *reinterpret_cast<uint2*>(p.dX + o0 + 16 * c) = pack4(v);
            }
        }
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

// =====================================================================================================
// Fused backward of a project (1x1) convolution with the squeeze-excite gate and BatchNorm1 + Swish in front of it, for the
// early high-resolution MBConv blocks (blocks 0-4: S <= 48 project channels, 3/4 of the network's depthwise-output bytes).  The
// unfused order moves the block's depthwise-resolution tensors seven times: weight gradient (reads y_d, re-forms a_s on load),
// data gradient (writes d a_s), the five-sum pooling pass (reads d a_s, y_d), the BN1-backward apply (reads d a_s, y_d, writes
// d y_d).  d a_s = d y_p W is a K = 16-48 product of a tensor six times smaller, so it is cheaper to form it twice on the
// matrix pipe than to store it once:
//   phase 0 (before the squeeze-excite backward): d a_s tile by MFMA -> bf16 (the value the unfused path stores) -> the five
//            per-image sums of chan_pool5_kernel and a_s = swish(bn1(y_d)) * gate from the SAME registers -> dW_p += d y_p^T a_s;
//            reads y_d once, writes only partial sums and slabs;
//   phase 1 (after the BN1-backward finalize): the same d a_s tile again (same instructions, same bits) -> d y_d with
//            bnact_bwd_apply_kernel's arithmetic; reads y_d once, writes d y_d once.
// The expanded channels are cut into slices of 16 NLT (48, or 32 for block 0); a WAVE owns one slice of runs of consecutive
// 32-pixel tiles of one image (a run = one pooling record; the waves of a block take the slices of the same runs, so the block
// reads whole rows).  Elementwise work is laid out as lane = (16-B channel piece, pixel sub-lane): the per-channel parameters
// and the five running sums stay in registers for the whole run; per-slice registers are what lets 3-4 waves per SIMD overlap
// the phases (the elementwise phase is VALU-bound: two transcendentals and ~25 other operations per element).
__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }
struct ProjBwdArgs {
    const bf16 *dYp, *Yd, *Wt;               // [npix][S], [npix][L], W^T [L][S] (the conv's transposed shadow)
    bf16* dYd;                               // phase 1: [npix][L]
    float* slab;                             // phase 0: [blocks x run lanes][S][L] fp32 partial dW (conv weight layout [M = S][K = L])
    float* pool5;                            // phase 0: [imgs][nch][5][L]
    const float *sc, *sh, *mean, *istd;      // BN1 forward affine and statistics [groups][L]
    const float *ca, *cb, *cc;               // BN1-backward coefficients [groups][L] (phase 1)
    const float *gate, *ds;                  // [imgs][L]
    int L, nsl, imgs, HW, ipg, nch;          // nsl = L / (16 NLT) slices = waves per run lane
    float inv_hw;
    int strideB, strideS, strideW;           // LDS row strides in bytes (32 x odd)
};
template <int NLT, int CS, int PHASE>
#ifndef FM_PROJ_WPE
#define FM_PROJ_WPE 2
#endif
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(FM_PROJ_WPE, 4))) void pw_proj_bwd_kernel(const ProjBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LS = 16 * NLT, S = 16 * CS;
    constexpr int NOCT = 4 * NLT, NJ = 64 / NOCT, R = (32 + NJ - 1) / NJ, EVL = NJ * NOCT;    // pieces of 4 channels (8 B)
    constexpr int CPS = 2 * CS;                             // 16-B chunks per d y_p row; a 32-pixel tile = CS chunks per lane
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform loop control
    const int li = lane & 15, lg = lane >> 4;
    const int L = p.L, L4 = L >> 2, nsl = p.nsl;
    const int rw = wave / nsl, slice = wave - rw * nsl, nrw = (int)(blockDim.x >> 6) / nsl;
    const int l0 = slice * LS;
    const int strideB = p.strideB, strideS = p.strideS, strideW = p.strideW;
    unsigned char* wt = smem;                                           // [L][strideW]
    unsigned char* base = smem + L * strideW + (size_t)wave * 32 * (strideB + strideS);
    unsigned char* sbase = base + 32 * strideB;
    for (int c = tid; c < L * CPS; c += blockDim.x) {
        const int row = c / CPS, cb = c - row * CPS;
        *reinterpret_cast<uint4*>(wt + row * strideW + cb * 16) = *reinterpret_cast<const uint4*>(p.Wt + (size_t)row * S + 8 * cb);
    }
    __syncthreads();
    const int oc = lane % NOCT, jr = lane / NOCT;
    const bool ev = lane < EVL;                             // lanes of the elementwise phase: lane = jr * NOCT + oc
    const int tpi = (p.HW + 31) >> 5, nruns = p.imgs * p.nch;
    const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
    f32x4 acc[PHASE == 0 ? CS : 1][PHASE == 0 ? NLT : 1];
    if constexpr (PHASE == 0) {
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int r = 0; r < NLT; ++r) acc[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    uint2 vy[R];
    uint4 vs0, vs1, vs2;
    for (int run = blockIdx.x * nrw + rw; run < nruns; run += gridDim.x * nrw) {
        const int img = run / p.nch, ch = run - img * p.nch;
        const int t0 = (int)((long long)ch * tpi / p.nch), t1 = (int)((long long)(ch + 1) * tpi / p.nch);
        const int g = img / p.ipg;
        const uint2* yimg = reinterpret_cast<const uint2*>(p.Yd + (size_t)img * p.HW * L) + (l0 >> 2) + oc;
        const uint4* simg = reinterpret_cast<const uint4*>(p.dYp + (size_t)img * p.HW * S);
        uint2* dimg = reinterpret_cast<uint2*>(p.dYd + (size_t)img * p.HW * L) + (l0 >> 2) + oc;
        // per-lane parameters of this lane's 4 channels: pa / pb = mean / istd (phase 0) or ca / cb (phase 1)
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
            const uint2* ty = yimg + (size_t)t * 32 * L4;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int row = jr + NJ * i;
                vy[i] = (ev && row < 32 && row < nv) ? ty[(size_t)row * L4] : make_uint2(0u, 0u);
            }
            const uint4* tx = simg + (size_t)t * 32 * CPS + lane;
            vs0 = lane / CPS < nv ? tx[0] : zero4();
            if constexpr (CS >= 2) vs1 = (lane + 64) / CPS < nv ? tx[64] : zero4();
            if constexpr (CS >= 3) vs2 = (lane + 128) / CPS < nv ? tx[128] : zero4();
        };
        gload(t0);
        for (int t = t0; t < t1; ++t) {
            const int nv = p.HW - 32 * t;
            // ---- d y_p tile -> LDS ----
            *reinterpret_cast<uint4*>(sbase + (lane / CPS) * strideS + (lane % CPS) * 16) = vs0;
            if constexpr (CS >= 2) *reinterpret_cast<uint4*>(sbase + ((lane + 64) / CPS) * strideS + ((lane + 64) % CPS) * 16) = vs1;
            if constexpr (CS >= 3) *reinterpret_cast<uint4*>(sbase + ((lane + 128) / CPS) * strideS + ((lane + 128) % CPS) * 16) = vs2;
            __builtin_amdgcn_wave_barrier();
            // ---- data gradient: D[l][pix] = sum_s W^T[l][s] d y_p[pix][s]; lane (li, lg): 4 consecutive l of pixel li -> bf16 tile ----
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const unsigned char* xrow = sbase + (16 * pt + li) * strideS;
                unsigned char* drow = base + (16 * pt + li) * strideB + 8 * lg;
                // K = S in chunks of 32; the last chunk of S = 16 / 48 is half a chunk: lane groups 2, 3 feed zeros on both sides
                // (one opcode for the whole chain: a dependent 16x16x32 -> 16x16x16 pair gave wrong low accumulator halves)
                constexpr int NKC = (S + 31) / 32, HALF = (S & 16) != 0;
                uint4 bf[NKC];
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    bf[kc] = (HALF && kc == NKC - 1 && lg >= 2) ? zero4() : *reinterpret_cast<const uint4*>(xrow + 64 * kc + 16 * lg);
#pragma unroll
                for (int r = 0; r < NLT; ++r) {
                    const unsigned char* wrow = wt + (l0 + 16 * r + li) * strideW;
                    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const uint4 af = (HALF && kc == NKC - 1 && lg >= 2) ? zero4() : *reinterpret_cast<const uint4*>(wrow + 64 * kc + 16 * lg);
                        d = mfma32(af, bf[kc], d);
                    }
                    *reinterpret_cast<uint2*>(drow + 32 * r) = pack4(d);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- elementwise: (d a_s, y_d) -> sums + a_s (phase 0) / d y_d (phase 1) ----
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int row = jr + NJ * i;
                if (ev && row < 32 && row < nv) {
                    unsigned char* dp = base + row * strideB + oc * 8;
                    f32x4 d = cvt4(*reinterpret_cast<const uint2*>(dp));
                    const f32x4 y = cvt4(vy[i]);
                    const f32x4 v = y * sc + sh;
                    f32x4 o;
                    if constexpr (PHASE == 0) {
                        const f32x4 xh = (y - pa) * pb;
                        f32x4 ad, sg;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-v[k]));
                            ad[k] = v[k] * sgm;
                            sg[k] = sgm * (1.f + v[k] * (1.f - sgm));
                        }
                        const f32x4 dsg = d * sg;
                        sm[0] += d * ad;
                        sm[1] += dsg;
                        sm[2] += dsg * xh;
                        sm[3] += sg;
                        sm[4] += sg * xh;
                        o = ad * gt;
                        *reinterpret_cast<uint2*>(dp) = pack4(o);
                    } else {
                        d = d * gt + dv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-v[k]));
                            d[k] *= sgm * (1.f + v[k] * (1.f - sgm));
                        }
                        o = pa * d + pb * y + pc;
                        dimg[((size_t)t * 32 + row) * L4] = pack4(o);
                    }
                }
            }
            if (t + 1 < t1) gload(t + 1);
            if constexpr (PHASE == 0) {
                __builtin_amdgcn_wave_barrier();
                // ---- weight gradient: dW[s][l] += sum_pix d y_p[pix][s] a_s[pix][l] (transposed fragments of both tiles; the rows
                //      of a ragged tile beyond nv hold d y_p = 0 against finite a_s) ----
                uint4 a[CS];
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    const unsigned char* ab = sbase + trow * strideS + (16 * c + tcol) * 2;
                    const uint2 a0 = ds_read_tr16(ab), a1 = ds_read_tr16(ab + 16 * strideS);
                    a[c] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
#pragma unroll
                for (int r = 0; r < NLT; ++r) {
                    const unsigned char* bb = base + trow * strideB + (16 * r + tcol) * 2;
                    const uint2 b0 = ds_read_tr16(bb), b1 = ds_read_tr16(bb + 16 * strideB);
                    const uint4 b = make_uint4(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
                    for (int c = 0; c < CS; ++c) acc[c][r] = mfma32(a[c], b, acc[c][r]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PHASE == 0) {
            // the NJ pixel sub-lanes of a channel piece are folded through LDS in a fixed order: one record per run
            float* scr = reinterpret_cast<float*>(base);
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
        // acc[c][r][q] = dW[s = 16 c + 4 lg + q][l = l0 + 16 r + li]; the run lane's slab (zeros when it had no run), this
        // wave's columns of it
        float* slab = p.slab + (size_t)(blockIdx.x * nrw + rw) * L * S + l0;
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int r = 0; r < NLT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) slab[(size_t)(16 * c + 4 * lg + q) * L + 16 * r + li] = acc[c][r][q];
    }
}

// smallest 32 x odd >= bytes
int odd32(int bytes)
{
    int n = (bytes + 31) / 32;
    if (!(n & 1)) ++n;
    return 32 * n;
}

// =====================================================================================================
__global__ void cast_weights_kernel(const float* __restrict__ state, bf16* __restrict__ shadow,
                                    const CastJob* __restrict__ jobs, int njobs)
{
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
    const CastJob jb = jobs[j];
    const int i = ((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x;
    if (i >= jb.M * jb.K) return;
    const int m = i / jb.K, k = i - m * jb.K;
    const bf16 v = (bf16)state[jb.src_off + i];
    shadow[jb.w_off + i] = v;
    shadow[jb.t_off + (long long)k * jb.M + m] = v;
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------------
// row tiles per block: 4 (64 channels).  K > 640: the slice is 86-147 KB -- one 8-wave block per CU (FM_PW_W8=0: two
// 4-wave blocks of 32-channel tiles, the round-2 first form)
bool pw_gemm_takes(int M, int K, bool pro, bool plain, int HW);
int pw_gemm_blocks(int npix_per_group);
static int pw_w8() { static const int v = fm_tune("FM_PW_W8", 1); return v; }
static int pw_rt(int M, int K)
{
    (void)M;
    return (K <= 640 || (pw_w8() && K <= 1152)) ? 4 : 2;          // K = 1280 (head dgrad): 164 KB, stays 32-channel
}
static int pw_ppb(int npix_per_group, int groups, int M, int K)
{
    const int MT = 16 * pw_rt(M, K);
    const long long tilesM = (M + MT - 1) / MT;
    const long long total = (long long)npix_per_group * groups * tilesM;
    long long ppb = total / 2048;                               // ~2048 blocks when there is enough work
    // staging the weight slice costs MT*K*2 B per block: keep it <= ~6 % of the block's pixel traffic
    ppb = std::max<long long>(ppb, 16LL * MT);
    ppb = std::min<long long>(std::max<long long>(ppb, 128), 4096);
    return (int)((ppb + 31) / 32 * 32);
}
static bool pw_gemm_pro_shape(int M, int K);
// M-tiles of a launch WITH an operand prologue (what the engine asks before it fuses the squeeze-excite gate into a project conv)
int pw_tiles_m(int M, int K)
{
    if (pw_gemm_pro_shape(M, K)) return (M + 127) / 128;
    return (M + 16 * pw_rt(M, K) - 1) / (16 * pw_rt(M, K));
}
int pw_blocks(int npix_per_group, int groups, int M, int K, bool pro, int HW)
{
    if (pro && pw_gemm_takes(M, K, true, false, HW)) return pw_gemm_blocks(npix_per_group);   // train forward of a fused project conv
    const int ppb = pw_ppb(npix_per_group, groups, M, K);
    return (npix_per_group + ppb - 1) / ppb;
}

template <int RT, int P, int KB, int NW, int MODE>
static void pw_launch_m(const PwParams& p, bool pro, dim3 grid, size_t lds, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int cap = NW == 8 ? 156 * 1024 : 96 * 1024;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&pw_conv_bf16_kernel<RT, false, P, KB, NW, MODE>), cap, "pw_conv_bf16_kernel");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pw_conv_bf16_kernel<RT, true, P, KB, NW, MODE>), cap, "pw_conv_bf16_kernel");
        attr_done = true;
    }
    if (pro) hipLaunchKernelGGL((pw_conv_bf16_kernel<RT, true, P, KB, NW, MODE>), grid, dim3(64 * NW), lds, s, p);
    else hipLaunchKernelGGL((pw_conv_bf16_kernel<RT, false, P, KB, NW, MODE>), grid, dim3(64 * NW), lds, s, p);
}
template <int RT, int P, int KB, int NW = 4>
static void pw_launch_t(const PwParams& p, bool pro, dim3 grid, size_t lds, hipStream_t s)
{
    if (p.stats) pw_launch_m<RT, P, KB, NW, 1>(p, pro, grid, lds, s);          // train forward: raw store + BN partial sums
    else if (p.scale) pw_launch_m<RT, P, KB, NW, 2>(p, pro, grid, lds, s);     // eval forward: folded BN (+ Swish, residual)
    else pw_launch_m<RT, P, KB, NW, 0>(p, pro, grid, lds, s);                  // data gradient (+ residual)
}
// ---- LDS-tiled GEMM form (pw_gemm_bf16_kernel) ----------------------------------------------------------------------------
static int pw_gemm_on() { static const int v = fm_tune("FM_PW_GEMM", 1); return v; }
static int pw_gemm_pro_on() { static const int v = fm_tune("FM_PW_GEMM_PRO", 1); return v; }
static int pw_gemm_bm(int M) { return M <= 64 ? 64 : 128; }
// Taken for (1) the plain-store form with K >= 64 (the data gradients: no statistics, no affine).  Measured inside the bf16 bs-512
// step (one stream, tools/op_profile.py): expand / project / head data gradients 2.24 -> 1.80 ms per step; the train (statistics)
// and eval (affine + Swish) forwards of the K = 80 ... 192 expand convs were SLOWER in this form (2.69 -> 3.19 / 2.76 -> 2.96 ms: a
// 128-pixel tile is two stages of K there, so the per-tile statistics fold and epilogue loads weigh as much as its main loop;
// the streaming kernel folds its statistics once per ~1 000 pixels) and the K = 672 / 1152 project forwards did not move;
// (2) the project convs WITH an operand prologue and K >= 240 (blocks 4-10, M <= 128 = one M-tile): the prologue runs once per
// element instead of once per 64-channel tile, and a tile is 4-11 stages long.
static bool pw_gemm_pro_shape(int M, int K)
{
    static const int mink = fm_tune("FM_PW_GEMM_PRO_MINK", 240);
    // (256 = blocks 11-14, M = 192, as two 128-channel tiles with the gate fused instead of a materialised a_s: measured slower,
    // project forward 2.52 -> 2.85 ms and its weight gradient 1.59 -> 1.84 ms per step against 0.16 ms of se_scale saved)
    static const int maxm = fm_tune("FM_PW_GEMM_PRO_MAXM", 128);
    return pw_gemm_pro_on() && K >= mink && M <= maxm;
}
bool pw_gemm_takes(int M, int K, bool pro, bool plain, int HW)
{
    if ((M & 15) || (K & 15) || !pw_gemm_on()) return false;
    return pro ? (pw_gemm_pro_shape(M, K) && HW >= 32) : (plain && K >= 64);     // (a wave's 32 pixels span at most two images)
}
constexpr int PW_GEMM_BP = 128;
int pw_gemm_blocks(int npix_per_group) { return (npix_per_group + PW_GEMM_BP - 1) / PW_GEMM_BP; }
template <int BM, int MODE, int PRO>
static void pw_gemm_launch_m(const PwParams& p, dim3 grid, hipStream_t s)
{
    constexpr int NS = 2;       // two stages (64 / 48 KB: two / three blocks per CU); three = one block per CU was 30-50 % slower
    const int lds = NS * (BM + PW_GEMM_BP) * 128 + (PRO ? NS * 4 * 1024 : 0) + (PRO == 2 ? 2 * p.K * 4 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&pw_gemm_bf16_kernel<BM, PW_GEMM_BP, NS, MODE, PRO>), 96 * 1024, "pw_gemm_bf16_kernel");
        attr_done = true;
    }
    hipLaunchKernelGGL((pw_gemm_bf16_kernel<BM, PW_GEMM_BP, NS, MODE, PRO>), grid, dim3(256), lds, s, p);
}
static bool launch_pw_gemm(PwParams p, hipStream_t s)
{
    const int BM = pw_gemm_bm(p.M);
    p.ppb = PW_GEMM_BP;
    p.nblk = pw_gemm_blocks(p.npix);
    p.tiles_m = (p.M + BM - 1) / BM;
    static const int xcd = fm_tune("FM_PW_XCD", 1);
    p.xcd = xcd;
    const dim3 grid(p.tiles_m * p.nblk * p.groups);
    if (!p.gate) {                                              // data gradients
        if (BM == 64) pw_gemm_launch_m<64, 0, 0>(p, grid, s); else pw_gemm_launch_m<128, 0, 0>(p, grid, s);
        return true;
    }
    // the engine uses <1, 2> (train forward of a fused project conv: statistics + BN1 / Swish / gate prologue) and <2, 1> (eval
    // forward: folded BN2 + gate); the kernel-level tests also run the plain-store forms
#define PW_GEMM_PRO(MODE, PRO)                                                                              \
    do { if (BM == 64) pw_gemm_launch_m<64, MODE, PRO>(p, grid, s); else pw_gemm_launch_m<128, MODE, PRO>(p, grid, s); } while (0)
    if (p.stats && p.scale) return false;
    if (p.psc) { if (p.stats) PW_GEMM_PRO(1, 2); else if (p.scale) PW_GEMM_PRO(2, 2); else PW_GEMM_PRO(0, 2); }
    else { if (p.stats) PW_GEMM_PRO(1, 1); else if (p.scale) PW_GEMM_PRO(2, 1); else PW_GEMM_PRO(0, 1); }
#undef PW_GEMM_PRO
    return true;
}

void launch_pw_conv(PwParams p, hipStream_t s)
{
    if (pw_gemm_takes(p.M, p.K, p.gate != nullptr, !p.stats && !p.scale, p.HW) && launch_pw_gemm(p, s)) return;
    const int RT = pw_rt(p.M, p.K);
    const int MT = 16 * RT;
    p.ppb = pw_ppb(p.npix, p.groups, p.M, p.K);
    p.nblk = (p.npix + p.ppb - 1) / p.ppb;
    const bool pro = p.gate != nullptr;
    size_t lds = (size_t)(p.K >> 5) * RT * 1024 + (size_t)RT * 512 + (pro ? (size_t)2 * p.K * 4 : 0);
    lds = std::max<size_t>(lds, (size_t)8 * MT * 2 * 4);
    static const int xcd = fm_tune("FM_PW_XCD", 1);
    p.tiles_m = (p.M + MT - 1) / MT;
    p.xcd = xcd;
    const dim3 grid(p.tiles_m * p.nblk * p.groups);
    // FM_PW_SMALLK: 0 = P 2 everywhere; 2 (default) = 4 pixel groups x 2 k-chunks for K <= 64 (2 waves per SIMD kept:
    // block 1's expand conv 1.07 -> 0.90 ms, the K <= 64 layers together -1.0 ms per step); 1 = 8 groups for K <= 32
    // (372 registers = 1 wave per SIMD: slower than 2).  Outputs are bit-identical across the settings; the BN partial
    // sums are taken in a different (still fixed) order.
    static const int smallk = fm_tune("FM_PW_SMALLK", 2);
    if (RT == 4 && smallk == 3 && p.K <= 64) pw_launch_t<4, 1, 2>(p, pro, grid, lds, s);
    else if (RT == 4 && smallk == 4 && p.K <= 64) pw_launch_t<4, 2, 2>(p, pro, grid, lds, s);
    else if (RT == 4 && smallk && p.K <= 64) pw_launch_t<4, 4, 2>(p, pro, grid, lds, s);
    else if (RT == 4 && p.K > 640) pw_launch_t<4, 2, 4, 8>(p, pro, grid, lds, s);
    else if (RT == 4) pw_launch_t<4, 2, 4>(p, pro, grid, lds, s);
    else pw_launch_t<2, 2, 4>(p, pro, grid, lds, s);
}

template <int CC>
static void wg_launch(const WgArgs& a, bool pro, dim3 grid, size_t lds, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&pw_wgrad_bf16_kernel<CC, false>), 64 * 1024, "pw_wgrad_bf16_kernel");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pw_wgrad_bf16_kernel<CC, true>), 64 * 1024, "pw_wgrad_bf16_kernel");
        attr_done = true;
    }
    if (pro) hipLaunchKernelGGL((pw_wgrad_bf16_kernel<CC, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((pw_wgrad_bf16_kernel<CC, false>), grid, dim3(256), lds, s, a);
}

int launch_pw_wgrad(const PwWgradParams& w, size_t slab_floats, hipStream_t s)
{
    WgArgs a{};
    a.swap = w.K > w.M;                      // the larger channel count is the tiled ("Big") operand
    a.Big = a.swap ? w.X : w.dY;
    a.Small = a.swap ? w.dY : w.X;
    a.L = std::max(w.M, w.K); a.S = std::min(w.M, w.K);
    if (a.S > 320 || (a.S & 15) || (a.L & 15)) return 0;
    a.slab = w.slab; a.M = w.M; a.K = w.K; a.npix = w.npix;
    a.psc = w.psc; a.psh = w.psh; a.gate = w.gate; a.HW = w.HW; a.pix_per_group = w.pix_per_group;
    const int cc = a.S / 16;
    {   // small high-resolution layers: the barrier-free one-wave-per-tile kernel
        static const int wave_on = fm_tune("FM_PW_WG_WAVE", 1);
        const int nrt = a.L / 16;
        const bool shape = (nrt == 2 && cc == 1) || (nrt == 6 && (cc == 1 || cc == 2)) || (nrt == 9 && cc == 2);
        // measured (ms, 1024 images): without prologue 96x16 0.77 -> 0.55, 144x32 0.35 -> 0.31; with the gate prologue 32x16
        // 0.89 -> 0.65 but 96x32 0.48 -> 0.69 and 144x32 0.68 -> 1.26 (all of a tile's BN / gate operands per lane: 204-276
        // registers), so the prologue variants keep the block kernel except for the smallest shape
        if (wave_on && shape && (w.gate == nullptr || nrt == 2)) {
            a.strideS = odd32(2 * a.S);
            const int strideB = odd32(2 * a.L);
            const int tsteps = (w.npix + 31) / 32;
            int nblk = std::max(1, std::min(768, tsteps / 32));
            nblk = (int)std::min<size_t>(nblk, std::max<size_t>(1, slab_floats / ((size_t)4 * w.M * w.K)));
            const size_t lds = (size_t)4 * 32 * (strideB + a.strideS);
            const bool pro = w.gate != nullptr;
#define WG_WAVE(N, C)                                                                                                      \
    do {                                                                                                                   \
        static bool done_ = false;                                                                                         \
        if (!done_) {                                                                                                      \
            set_max_dyn_lds(reinterpret_cast<const void*>(&pw_wgrad_wave_kernel<N, C, false>), 64 * 1024, "pw_wgrad_wave");  \
            set_max_dyn_lds(reinterpret_cast<const void*>(&pw_wgrad_wave_kernel<N, C, true>), 64 * 1024, "pw_wgrad_wave");   \
            done_ = true;                                                                                                  \
        }                                                                                                                  \
        if (pro) hipLaunchKernelGGL((pw_wgrad_wave_kernel<N, C, true>), dim3(nblk), dim3(256), lds, s, a, strideB);        \
        else hipLaunchKernelGGL((pw_wgrad_wave_kernel<N, C, false>), dim3(nblk), dim3(256), lds, s, a, strideB);           \
    } while (0)
            if (nrt == 2) WG_WAVE(2, 1);
            else if (nrt == 6 && cc == 1) WG_WAVE(6, 1);
            else if (nrt == 6) WG_WAVE(6, 2);
            else WG_WAVE(9, 2);
#undef WG_WAVE
            return 4 * nblk;
        }
    }
    static const int avail[] = {1, 2, 3, 5, 7, 12, 20};
    int CCi = 20;
    for (int v : avail) if (v >= cc) { CCi = v; break; }
    a.strideS = odd32(2 * a.S);
    // the tile the kernel stages is 16*CC wide only through cps = S/8 chunks: rows hold S channels; fragments of
    // column tiles past S read stale LDS but their results are never stored (sc >= S)
    if (odd32(2 * a.S) < 32 * CCi) a.strideS = odd32(32 * CCi);
    const int tilesL = (a.L + 63) / 64;
    const int tsteps = (w.npix + 31) / 32;
    // Splits of the pixel axis.  Every split leaves an fp32 [M][K] partial that reduce_slabs reads back.  The late,
    // channel-heavy layers (1 152 x 192: 885 KB per split) run 512 blocks of >= 32 steps -- at 2 048 blocks they moved
    // more slab bytes than activations (bf16 bs-512 step 59.1 -> 58.3 ms) -- while layers whose partial is small
    // (<= FM_PW_WG_SMALL_KB, default 64: the early high-resolution ones, 24 x 96 = 12 KB) keep 2 048 blocks of >= 8 steps:
    // their gate / BN prologue is issue-bound and needs every SIMD busy (at 512 blocks the two largest ran 0.78 -> 1.9 ms).
    static const int wg_blocks = fm_tune("FM_PW_WG_BLOCKS", 512);
    static const int wg_minsteps = fm_tune("FM_PW_WG_MINSTEPS", 32);
    static const int wg_small_kb = fm_tune("FM_PW_WG_SMALL_KB", 64);
    const bool small_partial = (size_t)w.M * w.K * 4 <= ((size_t)wg_small_kb << 10);
    int splits = std::max(1, (small_partial ? 2048 : wg_blocks) / tilesL);
    splits = std::min(splits, std::max(1, tsteps / (small_partial ? 8 : wg_minsteps)));
    splits = (int)std::min<size_t>(splits, std::max<size_t>(1, slab_floats / ((size_t)w.M * w.K)));
    const size_t lds = (size_t)2 * (32 * WG_STRIDE_BIG + 32 * a.strideS);
    static const int xcd = fm_tune("FM_PW_XCD", 1);
    a.tilesL = tilesL; a.nsplit = splits; a.xcd = xcd;
    const dim3 grid(tilesL * splits);
    const bool pro = w.gate != nullptr;
    switch (CCi) {
    case 1: wg_launch<1>(a, pro, grid, lds, s); break;
    case 2: wg_launch<2>(a, pro, grid, lds, s); break;
    case 3: wg_launch<3>(a, pro, grid, lds, s); break;
    case 5: wg_launch<5>(a, pro, grid, lds, s); break;
    case 7: wg_launch<7>(a, pro, grid, lds, s); break;
    case 12: wg_launch<12>(a, pro, grid, lds, s); break;
    default: wg_launch<20>(a, pro, grid, lds, s); break;
    }
    return splits;
}


// Fused BN0-backward apply + expand-conv weight gradient + data gradient (pw_exp_bwd_kernel).  Returns the number of
// [M][K] slabs written (reduce with k_reduce_slabs), 0 = shape not handled (the caller runs the three separate passes).
int launch_pw_exp_bwd(const PwExpBwdParams& w, size_t slab_floats, hipStream_t s)
{
    const int nrt = w.L / 16, cc = w.S / 16;
    const bool shape = (nrt == 6 && cc == 1) || (nrt == 9 && cc == 2);
    if (!shape || (w.L & 15) || (w.S & 15) || w.pix_per_group % 32 != 0 || w.npix % 32 != 0 || w.groups < 1 || w.groups > 2) return 0;
    static const int on = fm_tune("FM_PW_EXP_BWD", 1);
    if (!on) return 0;
    ExpBwdArgs a{};
    a.dA = w.dA; a.Ye = w.Ye; a.X = w.X; a.Wt = w.Wt; a.res = w.res; a.dX = w.dX; a.slab = w.slab;
    a.ca = w.ca; a.cb = w.cb; a.cc = w.cc; a.sc = w.sc; a.sh = w.sh;
    a.npix = w.npix; a.pix_per_group = w.pix_per_group; a.groups = w.groups;
    a.strideB = odd32(2 * w.L); a.strideS = odd32(2 * w.S); a.strideW = odd32(2 * w.L);
    const int tsteps = (w.npix + 31) / 32;
    int nblk = std::max(1, std::min(768, tsteps / 32));
    nblk = (int)std::min<size_t>(nblk, std::max<size_t>(1, slab_floats / ((size_t)4 * w.L * w.S)));
    const size_t lds = (size_t)w.S * a.strideW + (size_t)w.groups * 5 * w.L * 4 + (size_t)4 * 32 * (a.strideB + a.strideS);
#define EXP_BWD(N, C)                                                                                                   \
    do {                                                                                                                \
        static bool done_ = false;                                                                                      \
        if (!done_) { set_max_dyn_lds(reinterpret_cast<const void*>(&pw_exp_bwd_kernel<N, C>), 96 * 1024, "pw_exp_bwd"); done_ = true; } \
        hipLaunchKernelGGL((pw_exp_bwd_kernel<N, C>), dim3(nblk), dim3(256), lds, s, a);                                \
    } while (0)
    if (nrt == 6) EXP_BWD(6, 1);
    else EXP_BWD(9, 2);
#undef EXP_BWD
    return 4 * nblk;
}

// Fused project-conv backward (pw_proj_bwd_kernel).  pw_proj_bwd_nch: pooling records per image (0 = shape not handled: the
// caller runs the separate passes).  Phase 0 returns the number of [S][L] slabs written (reduce with k_reduce_slabs).
static int proj_bwd_slice(int L) { return L == 32 ? 32 : (L % 48 == 0 ? 48 : 0); }
int pw_proj_bwd_nch(int L, int S, int imgs, int HW)
{
    static const int on = fm_tune("FM_PW_PROJ_BWD", 1), ragged = fm_tune("FM_PW_PROJ_RAGGED", 1);
    const int ls = proj_bwd_slice(L);
    if (!on || !ls || L / ls > 5 || (S != 16 && S != 32 && S != 48) || imgs < 1) return 0;
    if (HW % 32 != 0 && (!ragged || HW % 16 != 0)) return 0;
    const int tpi = (HW + 31) / 32;
    return std::max(1, std::min(std::min(16, tpi), (2048 + imgs - 1) / imgs));
}
int launch_pw_proj_bwd(const PwProjBwdParams& w, int phase, size_t slab_floats, hipStream_t s)
{
    const int nch = pw_proj_bwd_nch(w.L, w.S, w.imgs, w.HW);
    if (!nch || nch != w.nch || w.imgs % w.ipg != 0) return 0;
    const int ls = proj_bwd_slice(w.L), nsl = w.L / ls;
    ProjBwdArgs a{};
    a.dYp = w.dYp; a.Yd = w.Yd; a.Wt = w.Wt; a.dYd = w.dYd; a.slab = w.slab; a.pool5 = w.pool5;
    a.sc = w.sc; a.sh = w.sh; a.mean = w.mean; a.istd = w.istd; a.ca = w.ca; a.cb = w.cb; a.cc = w.cc;
    a.gate = w.gate; a.ds = w.ds;
    a.L = w.L; a.nsl = nsl; a.imgs = w.imgs; a.HW = w.HW; a.ipg = w.ipg; a.nch = nch; a.inv_hw = 1.f / (float)w.HW;
    a.strideB = odd32(2 * ls); a.strideS = odd32(2 * w.S); a.strideW = odd32(2 * w.S);
    const int nrw = nsl == 1 ? 4 : (nsl == 2 ? 2 : 1);               // run lanes per block: 4 x 1, 2 x 2, 1 x 3..5 waves
    const int nwaves = nsl * nrw;
    const int nruns = w.imgs * nch;
    static const int cap = fm_tune("FM_PW_PROJ_BLOCKS", 1024);
    int nblk = std::max(1, std::min(cap, (nruns + nrw - 1) / nrw));
    if (phase == 0) nblk = (int)std::min<size_t>(nblk, std::max<size_t>(1, slab_floats / ((size_t)nrw * w.L * w.S)));
    const size_t lds = (size_t)w.L * a.strideW + (size_t)nwaves * 32 * (a.strideB + a.strideS);
#define PROJ_BWD(N, C, PH)                                                                                              \
    do {                                                                                                                \
        static bool done_ = false;                                                                                      \
        if (!done_) { set_max_dyn_lds(reinterpret_cast<const void*>(&pw_proj_bwd_kernel<N, C, PH>), 96 * 1024, "pw_proj_bwd"); done_ = true; } \
        hipLaunchKernelGGL((pw_proj_bwd_kernel<N, C, PH>), dim3(nblk), dim3(64 * nwaves), lds, s, a);                   \
    } while (0)
#define PROJ_BWD_S(N, PH)                                                                                               \
    do {                                                                                                                \
        if (w.S == 16) PROJ_BWD(N, 1, PH);                                                                              \
        else if (w.S == 32) PROJ_BWD(N, 2, PH);                                                                         \
        else PROJ_BWD(N, 3, PH);                                                                                        \
    } while (0)
    if (phase == 0) {
        if (ls == 32) PROJ_BWD_S(2, 0);
        else PROJ_BWD_S(3, 0);
        return nrw * nblk;
    }
    if (ls == 32) PROJ_BWD_S(2, 1);
    else PROJ_BWD_S(3, 1);
#undef PROJ_BWD_S
#undef PROJ_BWD
    return 1;
}

void launch_cast_weights(const float* state, bf16* shadow, const CastJob* jobs, int njobs, int nblocks, hipStream_t s)
{
    if (njobs) hipLaunchKernelGGL(cast_weights_kernel, dim3(nblocks), dim3(256), 0, s, state, shadow, jobs, njobs);
}
