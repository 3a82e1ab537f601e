// Convolution weight gradient as a split-K GEMM, gfx950: fp32 operands and accumulators, the products as exact bf16 partial
// products on the bf16 matrix pipe (split3.h; template parameter SP = 6 / 9, the shipped form) or on the fp32 pipe (SP = 0).
//
// Reference op replaced: the weight-gradient half of loss.backward()
// (utils/local_training.py:674, 965, 1191) for every nn.Conv2d of the model
// (cuDNN wgrad under torchvision resnet18, model/all_models.py:53-54).
//
//   dW[m][n] = sum_p dY[p][m] * Xg[p][n]      m: out channel, n: (tap, ci),
//                                             p: output pixel over ALL images
// Both operands are pixel-major in NHWC, so the LDS tiles are k-major
// ([32 pixels][BM] and [32 pixels][BN]) and each lane fetches, with ONE
// ds_read_b128 per operand per 4-pixel substep, the A values of 4 row-tiles /
// the B values of 4 column-tiles (rows m = 4i+r and columns n = 4j+c are
// interleaved across the 4x4 MFMA tiles).  That layout is conflict-free without
// a swizzle, and it makes every lane own 4 consecutive n of one m -> 16-B
// stores of the partial slab.  The pixel axis is split over blockIdx.y; slabs
// are summed in a fixed order by reduce_slabs (run-to-run deterministic, like
// the reference's cudnn.deterministic=True, main.py:36-37).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "split3.h"

// SP = 9 / 6: the products as exact bf16 partial products (split3.h).  The staging threads split their 16-B fragments ONCE per
// block on the way into LDS (three bf16 planes per operand, [plane][32 pixels][channels], rows padded to 32 B x odd); the
// MFMA fragments come out of the pixel-major tiles through ds_read_b64_tr_b16 (hardware transpose: a lane supplies the
// address of pixel row 4 lg + (li >> 2), columns 4 (li & 3).. of a 16-channel block and receives channel li of pixels
// 4 lg .. 4 lg + 3; two reads = the 8 k-slots of v_mfma_f32_16x16x32_bf16, the same pixel permutation for both operands).
// X is the row operand of the MFMA, so a lane ends up with 4 consecutive n of one m: one 16-B slab store per tile.
// The planes take 6 B per element where fp32 takes 4: ONE LDS buffer (two barriers per step) keeps two blocks per CU.
#if __HIP_DEVICE_COMPILE__
__device__ __forceinline__ uint2 wg_read_tr16(const unsigned char* lds_ptr)
{
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const unsigned off = (unsigned)(size_t)lds_ptr;          // generic -> LDS: the low 32 bits are the LDS offset
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)off));
}
#endif

// FC = 16-column MFMA tiles per wave: 4 (64x64 wave tile) or 3 (64x48: BN = 192 tiles N = 576 = 9 taps x 64
// channels of ResNet layer 1 exactly, where 256-wide tiles waste a quarter of the MFMA work)
template <int BM, int BN, int WN, int FC = 4, int SP = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int WM = 4 / WN;
    static_assert(BM == WM * 64 && BN == WN * 16 * FC && (FC == 4 || FC == 3), "wave tile is 64 x 16*FC");
    constexpr int BNP = FC == 4 ? BN : BN + 16;  // LDS row stride of the B tile (the pad keeps the FC = 3 reads conflict-free)
    constexpr int CA = BM / 4, CB = BN / 4;      // 16-B chunks per tile row
    constexpr int RPA = 256 / CA, NA = 32 / RPA; // rows per pass / passes
    constexpr int RPB = CB > 32 ? 4 : 8, NB = 32 / RPB;   // B staging: RPB*CB threads take part (192 of 256 for CB = 48)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][32][BM]
    float* Bs = smem + 2 * 32 * BM;   // [2][32][BNP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;

    const int tm = blockIdx.x % p.tilesM, tn = blockIdx.x / p.tilesM;
    const int split = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(p.npix, pbeg + p.pix_per_split);
    const int nsteps = (pend - pbeg + 31) >> 5;
    const int HWo = p.Ho * p.Wo;

    const int ca = tid % CA, ra0 = tid / CA;
    const int cb = tid % CB, rb0 = tid / CB;
    const bool bstage = rb0 < RPB;               // this thread stages B rows
    const int gc = (n0 >> 2) + cb;
    int4 e = {0, 0, 0, 0};
    if (bstage && gc < (p.Nw >> 2)) e = p.tab[gc];

    // Global loads go through two buffer descriptors with 32-bit offsets, and every bounds case (pixels past the
    // split's end, padded taps, halo rows / columns) is an offset the hardware range check answers with zeros: gload
    // has no branch and no 64-bit address arithmetic, so it shares ONE basic block with the MFMAs of the current step
    // and the scheduler interleaves the two (the pointer-select form cost ~250 VALU instructions in 18 basic blocks
    // ahead of every step's first MFMA).
    //   A: base = first pixel of the split, records = the split's rows -> rows past pend are out of range by themselves
    //   B: base = first image the split touches, offsets relative to it (launch_wgrad keeps a split's span < 2 GB)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dY + (size_t)pbeg * p.M), 0, (pend - pbeg) * p.M * 4, 0x00020000);
    const int img0 = pbeg / HWo;
    const int nimg = p.npix / HWo;
    const size_t img_bytes = (size_t)p.Hi * p.Wi * p.Ci * 4;
    const size_t restB = (size_t)(nimg - img0) * img_bytes;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.X + (size_t)img0 * (img_bytes >> 2)), 0, restB > 0x80000000ull ? 0x80000000u : (unsigned)restB, 0x00020000);
    // the offset of every load that must return zeros: both descriptors hold at most 2^31 bytes (launch_wgrad checks the
    // spans), so 2^31 is out of range for either without relying on how the range check treats a 32-bit wrap
    constexpr int OOR = (int)0x80000000u;
    const bool a_ok = m0 + 4 * ca < p.M;
    const int a_off0 = (ra0 * p.M + m0 + 4 * ca) * 4;
    const bool b_ok = bstage && e.w;
    const int ci4 = p.Ci * 4, ez4 = e.z * 4;

    u32x4 ra[NA], rb[NB];
    // Pixel -> (image relative to img0, oh, ow) of every B row this thread stages: decoded once by division,
    // then advanced by 32 pixels per step with carries only (32 = d_img*HWo + dq*Wo + dr).
    int r_img[NB], r_oh[NB], r_ow[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int pix = pbeg + rb0 + RPB * q;
        const int im = pix / HWo;
        const int rem = pix - im * HWo;
        r_img[q] = im - img0;
        r_oh[q] = rem / p.Wo;
        r_ow[q] = rem - r_oh[q] * p.Wo;
    }
    const int d_img = 32 / HWo, rem32 = 32 - d_img * HWo;
    const int dq = rem32 / p.Wo, dr = rem32 - dq * p.Wo;
    // gaddr(s) computes the offsets of step s (call it for s = 0, 1, 2, ... in order: it advances the row state; steps
    // past the split's end get out-of-range offsets = zeros), gissue() issues the loads of the offsets computed last.
    // In the main loop the loads of step s+1 are issued FIRST (their latency hides behind the step's MFMAs), the offsets
    // of step s+2 are computed among the MFMAs, and sched_barriers keep the compiler from sinking the loads down to
    // the LDS stores (which it does to save registers: the whole global latency was exposed before every barrier).
    int offA[NA], offB[NB];
    auto gaddr = [&](int s) {
#pragma unroll
#ifdef FM_WGRAD_FAKE_L2
        for (int q = 0; q < NA; ++q) offA[q] = a_ok ? (a_off0 + ((s & 3) * 32 + RPA * q) * p.M * 4) : OOR;
#else
        for (int q = 0; q < NA; ++q) offA[q] = a_ok ? a_off0 + (s * 32 + RPA * q) * p.M * 4 : OOR;
#endif
        const int pb = pbeg + s * 32;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int pix = pb + rb0 + RPB * q;
            // 24-bit multiplies (full rate; every factor is far below 2^23, launch_wgrad checks the span) and bitwise
            // predicate logic: no 64-bit mad, no short-circuit branch
            const int ih = __mul24(r_oh[q], p.stride) + e.x, iw = __mul24(r_ow[q], p.stride) + e.y;
            const bool ok = b_ok & (pix < pend) & ((unsigned)ih < (unsigned)p.Hi) & ((unsigned)iw < (unsigned)p.Wi);
            const int off = __mul24(__mul24(__mul24(r_img[q], p.Hi) + ih, p.Wi) + iw, ci4) + ez4;
#ifdef FM_WGRAD_FAKE_L2
            offB[q] = ok ? (off & 0x3fff0) : OOR;      // timing experiment only: every B load inside one 256-KB window
#else
            offB[q] = ok ? off : OOR;
#endif
            int ow = r_ow[q] + dr, oh = r_oh[q] + dq, im = r_img[q] + d_img;
            if (ow >= p.Wo) { ow -= p.Wo; ++oh; }
            if (oh >= p.Ho) { oh -= p.Ho; ++im; }
            r_ow[q] = ow; r_oh[q] = oh; r_img[q] = im;
        }
    };
    auto gissue = [&]() {
#pragma unroll
        for (int q = 0; q < NA; ++q) ra[q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[q], 0, 0);
#pragma unroll
        for (int q = 0; q < NB; ++q) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[q], 0, 0);
    };
    // split form: byte strides of a pixel row / a plane of the two tiles
    constexpr int SA = BM * 2 + 32, SB = BN * 2 + 32, PA = 32 * SA, PB = 32 * SB;
    unsigned char* const Ap = reinterpret_cast<unsigned char*>(smem);          // [3][32][SA]
    unsigned char* const Bp = Ap + 3 * PA;                                      // [3][32][SB], then 64 x 8-B dummy slots
    auto lstore_split = [&]() {
        unsigned char* a = Ap + ra0 * SA + 8 * ca;
        unsigned char* b = bstage ? Bp + rb0 * SB + 8 * cb : Bp + 3 * PB + 8 * (tid & 63);
        const int bstep = bstage ? RPB * SB : 0, bplane = bstage ? PB : 0;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, ra[q]);
            uint2 H, M, L;
            split3_pair(sp_f32x2{v[0], v[1]}, H.x, M.x, L.x);
            split3_pair(sp_f32x2{v[2], v[3]}, H.y, M.y, L.y);
            *reinterpret_cast<uint2*>(a + q * RPA * SA) = H;
            *reinterpret_cast<uint2*>(a + q * RPA * SA + PA) = M;
            *reinterpret_cast<uint2*>(a + q * RPA * SA + 2 * PA) = L;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, rb[q]);
            uint2 H, M, L;
            split3_pair(sp_f32x2{v[0], v[1]}, H.x, M.x, L.x);
            split3_pair(sp_f32x2{v[2], v[3]}, H.y, M.y, L.y);
            *reinterpret_cast<uint2*>(b + q * bstep) = H;
            *reinterpret_cast<uint2*>(b + q * bstep + bplane) = M;
            *reinterpret_cast<uint2*>(b + q * bstep + 2 * bplane) = L;
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * 32 * BM + ra0 * BM + 4 * ca;
        // threads that stage no B row (64 of 256 when CB = 48) store their zeros to a private 16-B slot behind the tiles:
        // a branch here would let the compiler sink the B loads below the MFMAs, right in front of these stores
        float* b = bstage ? Bs + buf * 32 * BNP + rb0 * BNP + 4 * cb : Bs + 2 * 32 * BNP + 4 * (tid & 63);
        const int bstep = bstage ? RPB * BNP : 0;
#pragma unroll
        for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x4*>(a + q * RPA * BM) = ra[q];
#pragma unroll
        for (int q = 0; q < NB; ++q) *reinterpret_cast<u32x4*>(b + q * bstep) = rb[q];
    };

    const bool wave_cols = __builtin_amdgcn_readfirstlane(n0 + wn * (16 * FC)) < p.Nw;
    f32x4 acc[4][FC];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (SP != 0) {
        // acc[r][c][q] = dW[m = m0 + wm*64 + 16 r + li][n = n0 + wn*16FC + 16 c + 4 lg + q]
        if (nsteps > 0) {
            gaddr(0);
            gissue();
            lstore_split();
            gaddr(1);
        }
        __syncthreads();
        const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
        const unsigned char* fa = Ap + trow * SA + (wm * 64 + tcol) * 2;
        const unsigned char* fb = Bp + trow * SB + (wn * (16 * FC) + tcol) * 2;
        for (int s = 0; s < nsteps; ++s) {
            gissue();                    // step s+1 (the step after the last loads zeros)
            __builtin_amdgcn_sched_barrier(0);
            gaddr(s + 2);
            if (wave_cols) {
                sp_u32x4 xh[FC], xm[FC], xl[FC];
#pragma unroll
                for (int c = 0; c < FC; ++c) {
                    const unsigned char* q = fb + 32 * c;
                    const uint2 h0 = wg_read_tr16(q), h1 = wg_read_tr16(q + 16 * SB);
                    const uint2 m0 = wg_read_tr16(q + PB), m1 = wg_read_tr16(q + PB + 16 * SB);
                    const uint2 l0 = wg_read_tr16(q + 2 * PB), l1 = wg_read_tr16(q + 2 * PB + 16 * SB);
                    xh[c] = sp_u32x4{h0.x, h0.y, h1.x, h1.y};
                    xm[c] = sp_u32x4{m0.x, m0.y, m1.x, m1.y};
                    xl[c] = sp_u32x4{l0.x, l0.y, l1.x, l1.y};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned char* q = fa + 32 * r;
                    const uint2 h0 = wg_read_tr16(q), h1 = wg_read_tr16(q + 16 * SA);
                    const uint2 m0 = wg_read_tr16(q + PA), m1 = wg_read_tr16(q + PA + 16 * SA);
                    const uint2 l0 = wg_read_tr16(q + 2 * PA), l1 = wg_read_tr16(q + 2 * PA + 16 * SA);
                    const sp_u32x4 yh = {h0.x, h0.y, h1.x, h1.y}, ym = {m0.x, m0.y, m1.x, m1.y}, yl = {l0.x, l0.y, l1.x, l1.y};
#pragma unroll
                    for (int c = 0; c < FC; ++c) acc[r][c] = mfma_split<SP>(xh[c], xm[c], xl[c], yh, ym, yl, acc[r][c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();             // every wave has read step s
            lstore_split();
            __syncthreads();
        }
        const int nb = n0 + wn * (16 * FC) + 4 * lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + 16 * r + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int c = 0; c < FC; ++c) {
                const int n = nb + 16 * c;
                if (n < p.Nw) *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + m) * p.Nw + n) = acc[r][c];
            }
        }
        return;
    }
    if (nsteps > 0) {
        gaddr(0);
        gissue();
        lstore(0);
        gaddr(1);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        gissue();                        // step s+1, unconditional (the step after the last loads zeros): one basic block
        __builtin_amdgcn_sched_barrier(0);
        gaddr(s + 2);
        const float* A = As + buf * 32 * BM + lg * BM + wm * 64 + 4 * li;
        const float* B = Bs + buf * 32 * BNP + lg * BNP + wn * (16 * FC) + FC * li;
        // a wave whose 16*FC columns all lie past Nw (the last N-tile of N = 576 = 2.25 x 256) only stages: its
        // fragment reads and MFMAs are skipped, wave-uniformly
        if (wave_cols)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(A + kk * 4 * BM);
            float b[FC];
            if constexpr (FC == 4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(B + kk * 4 * BNP);
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = bv[c];
            } else {
#pragma unroll
                for (int c = 0; c < FC; ++c) b[c] = B[kk * 4 * BNP + c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < FC; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[c], acc[r][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        lstore(buf ^ 1);
        __syncthreads();
    }

    // acc[r][c][q] = dW[m0 + wm*64 + 16*lg + 4*q + r][n0 + wn*16FC + FC*li + c]
    const int n = n0 + wn * (16 * FC) + FC * li;
    if (n < p.Nw) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm * 64 + 16 * lg + 4 * q + r;
                if (m >= p.M) continue;
                float* dst = p.slab + ((size_t)split * p.M + m) * p.Nw + n;
                if constexpr (FC == 4) {
                    const f32x4 v = {acc[r][0][q], acc[r][1][q], acc[r][2][q], acc[r][3][q]};
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int c = 0; c < FC; ++c)
                        if (n + c < p.Nw) dst[c] = acc[r][c][q];      // Nw need not be a multiple of 3 (stem: 176)
                }
            }
    }
#endif
}

// ---------------------------------------------------------------------------------------
// "Skinny" weight gradient of a 1x1 stride-1 convolution whose smaller channel count S is
// <= 128 (EfficientNet-B0's expand / project convs: 96x16, 32x96, 480x80 ...).  These GEMMs are
// HBM-bound (K = all pixels, output tiny), so the 64x64-per-wave tiles above would spend most
// of their MFMA work and LDS traffic on padding.  Here the block tile is 64 (large channel dim
// L) x 16*CC (the WHOLE small dim): the large operand is read exactly once, each wave owns 16
// rows of L and all CC column tiles, no cross-wave reduction.
//   P[l][s] = sum_p Big[p][l] * Small[p][s]
// `swap` = the large operand is X (then P is dW transposed; the store transposes back).
// ---------------------------------------------------------------------------------------
struct SkinnyParams {
    const float* Big;     // [npix][L]
    const float* Small;   // [npix][S]
    float* slab;          // [splits][M][Nw]
    const float* zeros;
    int L, S, M, Nw, swap;
    int npix, pix_per_split;
    // gather != 0: the X operand is the im2col row of a stem conv (Ci = 4): 16-B chunk c of a row is
    // input pixel (oh*stride + c/kw_p - pad, ow*stride + c%kw_p - pad), valid while c%kw_p < k
    int gather, Hi, Wi, Ho, Wo, stride, pad, k, kw_p;
    int tilesL, nsplit, xcd;   // launch geometry of the 1-D grid
};

template <int CC>
__global__ __launch_bounds__(256) void wgrad_skinny_kernel(const SkinnyParams p)
{
    constexpr int BS = 16 * CC;                   // small-dim tile (covers S)
    constexpr int CB = BS / 4;                    // 16-B chunks per Small row
    constexpr int NB = (32 * CB + 255) / 256;     // staging passes for the Small tile
    __shared__ __attribute__((aligned(16))) float As[2][32][64];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][BS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // 1-D grid -> (64-channel tile, pixel split) with the tiles of a split on ONE XCD (ids congruent mod 8): they all
    // re-read the split's Small rows (measured on the bf16 twin of this kernel: -5 %)
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
    // the 32-pixel steps are dealt round-robin to the splits (step j -> split j % nsplit): the blocks running
    // together stream neighbouring memory instead of walking nsplit far-apart ranges (DRAM locality)
    const int tsteps = (p.npix + 31) >> 5;
    const int nsteps = split < tsteps ? (tsteps - split + nsplit - 1) / nsplit : 0;
    const int pend = p.npix;

    // staging: A tile = 32 rows x 16 chunks (2 per thread), Small tile = 32 rows x CB chunks
    const int ca = tid & 15, ra = tid >> 4;       // rows ra, ra+16
    const bool a_ok = l0 + 4 * ca < p.L;
    f32x4 va[2], vb[NB];
    // address of 16-B chunk `ch` of pixel `pix` of the X operand in gather (stem) mode
    auto xgather = [&](const float* X, int pix, int ch) -> const float* {
        const int HWo = p.Ho * p.Wo;
        const int img = pix / HWo, rem = pix - img * HWo;
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        const int kh = ch / p.kw_p, kw = ch - kh * p.kw_p;
        const int ih = oh * p.stride + kh - p.pad, iw = ow * p.stride + kw - p.pad;
        const bool ok = kw < p.k && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        return ok ? X + ((size_t)(img * p.Hi + ih) * p.Wi + iw) * 4 : p.zeros;
    };
    const bool big_g = p.gather && p.swap, small_g = p.gather && !p.swap;
    auto gload = [&](int s) {
        const int pb = (split + s * nsplit) * 32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pix = pb + ra + 16 * q;
            const float* src = p.zeros;
            if (a_ok && pix < pend) src = big_g ? xgather(p.Big, pix, (l0 >> 2) + ca) : p.Big + (size_t)pix * p.L + l0 + 4 * ca;
            va[q] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / CB, cb = c - row * CB;
            const int pix = pb + row;
            const float* src = p.zeros;
            if (c < 32 * CB && 4 * cb < p.S && pix < pend) src = small_g ? xgather(p.Small, pix, cb) : p.Small + (size_t)pix * p.S + 4 * cb;
            vb[q] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(&As[buf][ra + 16 * q][4 * ca]) = va[q];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / CB, cb = c - row * CB;
            if (c < 32 * CB) *reinterpret_cast<f32x4*>(&Bs[buf][row][4 * cb]) = vb[q];
        }
    };

    f32x4 acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float a = As[buf][4 * kk + lg][16 * wave + li];
#pragma unroll
            for (int c = 0; c < CC; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[buf][4 * kk + lg][16 * c + li], acc[c], 0, 0, 0);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    // acc[c][q] = P[l = l0 + 16*wave + 4*lg + q][s = 16*c + li]
    const int l = l0 + 16 * wave + 4 * lg;
    if (l < p.L) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int sc = 16 * c + li;
            if (sc >= p.S) continue;
            if (p.swap) {            // l = n (X channel), s = m: 4 consecutive n -> one 16-B store
                *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + sc) * p.Nw + l) = acc[c];
            } else {                 // l = m, s = n
#pragma unroll
                for (int q = 0; q < 4; ++q) p.slab[((size_t)split * p.M + l + q) * p.Nw + sc] = acc[c][q];
            }
        }
    }
}

// The same GEMM in the split-product form (split3.h): the staging threads split their 16-B fragments on the way into LDS (three
// bf16 planes per operand, [plane][32 pixels][channels], rows padded to 32 B x odd), fragments by ds_read_b64_tr_b16 -- the
// structure of wgrad_kernel's SP form: one LDS buffer, two barriers per step.  On the fp32 pipe a 32-pixel step of a 64 x 128
// tile is 64 MFMAs of 32 cycles per wave against ~1 200 cycles of bytes; here it is 48 of 16.
template <int CC, int SP>
__global__ __launch_bounds__(256) void wgrad_skinny_split_kernel(const SkinnyParams p)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int BS = 16 * CC;
    constexpr int CB = BS / 4;
    constexpr int NB = (32 * CB + 255) / 256;
    constexpr int SA = 160, SB = 32 * ((CC + 1) | 1);        // row strides in bytes (32 x odd)
    constexpr int PA = 32 * SA, PB = 32 * SB;
    __shared__ __attribute__((aligned(16))) unsigned char Pl[3 * PA + 3 * PB];
    unsigned char* const Ap = Pl;
    unsigned char* const Bp = Pl + 3 * PA;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
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
    const int nsteps = split < tsteps ? (tsteps - split + nsplit - 1) / nsplit : 0;
    const int pend = p.npix;
    const int ca = tid & 15, ra = tid >> 4;
    const bool a_ok = l0 + 4 * ca < p.L;
    f32x4 va[2], vb[NB];
    auto xgather = [&](const float* X, int pix, int ch) -> const float* {
        const int HWo = p.Ho * p.Wo;
        const int img = pix / HWo, rem = pix - img * HWo;
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        const int kh = ch / p.kw_p, kw = ch - kh * p.kw_p;
        const int ih = oh * p.stride + kh - p.pad, iw = ow * p.stride + kw - p.pad;
        const bool ok = kw < p.k && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        return ok ? X + ((size_t)(img * p.Hi + ih) * p.Wi + iw) * 4 : p.zeros;
    };
    const bool big_g = p.gather && p.swap, small_g = p.gather && !p.swap;
    auto gload = [&](int s) {
        const int pb = (split + s * nsplit) * 32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pix = pb + ra + 16 * q;
            const float* src = p.zeros;
            if (a_ok && pix < pend) src = big_g ? xgather(p.Big, pix, (l0 >> 2) + ca) : p.Big + (size_t)pix * p.L + l0 + 4 * ca;
            va[q] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / CB, cb = c - row * CB;
            const int pix = pb + row;
            const float* src = p.zeros;
            if (c < 32 * CB && 4 * cb < p.S && pix < pend) src = small_g ? xgather(p.Small, pix, cb) : p.Small + (size_t)pix * p.S + 4 * cb;
            vb[q] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto put = [&](unsigned char* dst, int plane_bytes, const f32x4 v) {
        uint2 H, M, L;
        split3_pair(sp_f32x2{v[0], v[1]}, H.x, M.x, L.x);
        split3_pair(sp_f32x2{v[2], v[3]}, H.y, M.y, L.y);
        *reinterpret_cast<uint2*>(dst) = H;
        *reinterpret_cast<uint2*>(dst + plane_bytes) = M;
        *reinterpret_cast<uint2*>(dst + 2 * plane_bytes) = L;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) put(Ap + (ra + 16 * q) * SA + 8 * ca, PA, va[q]);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = tid + 256 * q;
            const int row = c / CB, cb = c - row * CB;
            if (c < 32 * CB) put(Bp + row * SB + 8 * cb, PB, vb[q]);
        }
    };
    f32x4 acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) { gload(0); lstore(); }
    __syncthreads();
    const int trow = 4 * lg + (li >> 2), tcol = 4 * (li & 3);
    const unsigned char* fa = Ap + trow * SA + (16 * wave + tcol) * 2;
    const unsigned char* fb = Bp + trow * SB + tcol * 2;
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) gload(s + 1);
        {
            const uint2 h0 = wg_read_tr16(fa), h1 = wg_read_tr16(fa + 16 * SA);
            const uint2 m0 = wg_read_tr16(fa + PA), m1 = wg_read_tr16(fa + PA + 16 * SA);
            const uint2 q0 = wg_read_tr16(fa + 2 * PA), q1 = wg_read_tr16(fa + 2 * PA + 16 * SA);
            const sp_u32x4 ah = {h0.x, h0.y, h1.x, h1.y}, am = {m0.x, m0.y, m1.x, m1.y}, al = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const unsigned char* q = fb + 32 * c;
                const uint2 bh0 = wg_read_tr16(q), bh1 = wg_read_tr16(q + 16 * SB);
                const uint2 bm0 = wg_read_tr16(q + PB), bm1 = wg_read_tr16(q + PB + 16 * SB);
                const uint2 bl0 = wg_read_tr16(q + 2 * PB), bl1 = wg_read_tr16(q + 2 * PB + 16 * SB);
                acc[c] = mfma_split<SP>(ah, am, al, sp_u32x4{bh0.x, bh0.y, bh1.x, bh1.y}, sp_u32x4{bm0.x, bm0.y, bm1.x, bm1.y},
                                        sp_u32x4{bl0.x, bl0.y, bl1.x, bl1.y}, acc[c]);
            }
        }
        __syncthreads();
        if (s + 1 < nsteps) lstore();
        __syncthreads();
    }
    // acc[c][q] = P[l = l0 + 16*wave + 4*lg + q][s = 16*c + li]
    const int l = l0 + 16 * wave + 4 * lg;
    if (l < p.L) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int sc = 16 * c + li;
            if (sc >= p.S) continue;
            if (p.swap) {
                *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + sc) * p.Nw + l) = acc[c];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) p.slab[((size_t)split * p.M + l + q) * p.Nw + sc] = acc[c][q];
            }
        }
    }
#endif
}

// returns the number of splits used, or 0 if the shape is not handled (caller falls back)
int launch_wgrad_skinny(const WgradParams& w, size_t slab_floats, hipStream_t s)
{
    const int S = std::min(w.M, w.Nw), L = std::max(w.M, w.Nw);
    const bool stem = w.Ci == 4 && w.gather_k > 0;
    if (S > 128) return 0;
    if (!stem && (w.stride != 1 || w.Ci != w.Nw || w.Ho != w.Hi || w.Wo != w.Wi)) return 0;
    SkinnyParams p{};
    if (stem) {
        p.gather = 1; p.Hi = w.Hi; p.Wi = w.Wi; p.Ho = w.Ho; p.Wo = w.Wo; p.stride = w.stride;
        p.pad = w.gather_pad; p.k = w.gather_k; p.kw_p = w.gather_kw_p;
    }
    p.swap = w.Nw > w.M;      // large operand is X
    p.Big = p.swap ? w.X : w.dY;
    p.Small = p.swap ? w.dY : w.X;
    p.slab = w.slab; p.zeros = w.zeros;
    p.L = L; p.S = S; p.M = w.M; p.Nw = w.Nw; p.npix = w.npix;
    const int tilesL = (L + 63) / 64;
    int splits = std::max(1, 4096 / tilesL);
    splits = std::min(splits, std::max(1, w.npix / 512));
    splits = (int)std::min<size_t>(splits, std::max<size_t>(1, slab_floats / ((size_t)w.M * w.Nw)));
    p.pix_per_split = (((w.npix + splits - 1) / splits) + 31) & ~31;
    splits = (w.npix + p.pix_per_split - 1) / p.pix_per_split;
    static const int xcd = fm_tune("FM_PW_XCD", 1);
    p.tilesL = tilesL; p.nsplit = splits; p.xcd = xcd;
    dim3 grid(tilesL * splits);
    const int cc = (S + 15) / 16;
    const int sp = w.sp;
#define FM_SK(CC_)                                                                                                    \
    case CC_:                                                                                                         \
        if (sp == 6) hipLaunchKernelGGL((wgrad_skinny_split_kernel<CC_, 6>), grid, dim3(256), 0, s, p);               \
        else if (sp == 9) hipLaunchKernelGGL((wgrad_skinny_split_kernel<CC_, 9>), grid, dim3(256), 0, s, p);          \
        else hipLaunchKernelGGL(wgrad_skinny_kernel<CC_>, grid, dim3(256), 0, s, p);                                  \
        break;
    switch (cc < 8 ? cc : 8) {
        FM_SK(1) FM_SK(2) FM_SK(3) FM_SK(4) FM_SK(5) FM_SK(6) FM_SK(7) FM_SK(8)
    }
#undef FM_SK
    return splits;
}

// N-tile width for an (M, Nw) weight gradient: 128 with the 128-row tiles; 192 when it tiles Nw exactly
// (ResNet layer 1: 576 = 3 x 192) and 256 otherwise for the 64-row tiles
int wgrad_tile_n(int M, int Nw)
{
    static const int t192 = fm_tune("FM_WGRAD192", 1);
    if (M >= 128) return 128;
    return (t192 && (Nw % 192 == 0 || Nw <= 192)) ? 192 : 256;      // Nw = 176: the packed 7x7 stem
}

// false: the split is too large for the kernel's 32-bit buffer offsets (nothing launched; the caller reports the error)
bool launch_wgrad(const WgradParams& p, int splits, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int LDS_L = 2 * 32 * (128 + 128) * 4 + 1024;   // + 64 x 16 B dummy slots
    constexpr int LDS_S = 2 * 32 * (64 + 256) * 4 + 1024;
    constexpr int LDS_T = 2 * 32 * (64 + 192 + 16) * 4 + 1024;
    // split form: three bf16 planes of both tiles, single-buffered, + 64 x 8 B dummy slots
    constexpr int SPL_L = 3 * 32 * ((128 * 2 + 32) + (128 * 2 + 32)) + 512;
    constexpr int SPL_S = 3 * 32 * ((64 * 2 + 32) + (256 * 2 + 32)) + 512;
    constexpr int SPL_T = 3 * 32 * ((64 * 2 + 32) + (192 * 2 + 32)) + 512;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<128, 128, 2, 4, 9>), SPL_L, "wgrad_kernel<128, 128, 2, 4, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 256, 4, 4, 9>), SPL_S, "wgrad_kernel<64, 256, 4, 4, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 192, 4, 3, 9>), SPL_T, "wgrad_kernel<64, 192, 4, 3, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<128, 128, 2, 4, 6>), SPL_L, "wgrad_kernel<128, 128, 2, 4, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 256, 4, 4, 6>), SPL_S, "wgrad_kernel<64, 256, 4, 4, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 192, 4, 3, 6>), SPL_T, "wgrad_kernel<64, 192, 4, 3, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<128, 128, 2>), LDS_L, "wgrad_kernel<128, 128, 2>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 256, 4>), LDS_S, "wgrad_kernel<64, 256, 4>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 192, 4, 3>), LDS_T, "wgrad_kernel<64, 192, 4, 3>");
        attr_done = true;
    }
    {   // 32-bit buffer offsets: one split's dY rows and the images it spans must stay below 2 GB (never close: a split
        // of the largest layer at 1024 images is ~5 MB)
        const size_t spanA = ((size_t)p.pix_per_split + 64) * p.M * 4;
        const size_t spanB = ((size_t)p.pix_per_split / ((size_t)p.Ho * p.Wo) + 3) * p.Hi * p.Wi * p.Ci * 4;
        if (spanA >= (1ull << 31) || spanB >= (1ull << 31) || spanB / ((size_t)p.Ci * 4) >= (1ull << 23)) {
            fprintf(stderr, "fedmlp_hip: wgrad split too large for 32-bit buffer offsets (%zu / %zu bytes)\n", spanA, spanB);
            return false;
        }
    }
    dim3 grid(p.tilesM * p.tilesN, splits);
    const int sp = p.sp;
    if (sp == 9) {
        if (p.M >= 128) hipLaunchKernelGGL((wgrad_kernel<128, 128, 2, 4, 9>), grid, dim3(256), SPL_L, s, p);
        else if (p.bn == 192) hipLaunchKernelGGL((wgrad_kernel<64, 192, 4, 3, 9>), grid, dim3(256), SPL_T, s, p);
        else hipLaunchKernelGGL((wgrad_kernel<64, 256, 4, 4, 9>), grid, dim3(256), SPL_S, s, p);
    } else if (sp == 6) {
        if (p.M >= 128) hipLaunchKernelGGL((wgrad_kernel<128, 128, 2, 4, 6>), grid, dim3(256), SPL_L, s, p);
        else if (p.bn == 192) hipLaunchKernelGGL((wgrad_kernel<64, 192, 4, 3, 6>), grid, dim3(256), SPL_T, s, p);
        else hipLaunchKernelGGL((wgrad_kernel<64, 256, 4, 4, 6>), grid, dim3(256), SPL_S, s, p);
    } else if (p.M >= 128)
        hipLaunchKernelGGL((wgrad_kernel<128, 128, 2>), grid, dim3(256), LDS_L, s, p);
    else if (p.bn == 192)
        hipLaunchKernelGGL((wgrad_kernel<64, 192, 4, 3>), grid, dim3(256), LDS_T, s, p);
    else
        hipLaunchKernelGGL((wgrad_kernel<64, 256, 4>), grid, dim3(256), LDS_S, s, p);
    return true;
}
