// Implicit-GEMM convolution (forward + data gradient), gfx950: fp32 operands and accumulators; the products on the bf16 matrix
// pipe as exact bf16 partial products (split3.h, template parameter SP = 6 / 9: the shipped form) or on the fp32 pipe (SP = 0).
//
// Reference op replaced: every nn.Conv2d forward inside net(images)
// (utils/local_training.py:657, 937-947, 983, 1030, 1178) and its input
// gradient inside loss.backward() (:674, 965, 1191), which the reference gets
// from cuDNN through torchvision's resnet18 (model/all_models.py:53-54).
//
// In planes mode (ResNet-18 with a split product form: the shipped default) the 3x3 / 1x1 convolutions run through pconv.hip, which
// takes BOTH operands as bf16 planes; what still runs here: the 7x7 stem (STEM = 2, activations split in the kernel), every conv
// of a handle on the fp32 matrix pipe (fm_config.reserved[2] = 1) or with FM_PLANES=0, and EfficientNet-B0's wide fp32 1x1 convs.
//
// Roofline: SP = 6: v_mfma_f32_16x16x32_bf16, 2 500 TFLOP/s / 6 products = 416.7 TFLOP/s of fp32 products; SP = 0: fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32, 157 TFLOP/s).  The shipped instantiations and their stage schedule:
//   igemm_kernel<128, 128, 2, 0, NS = 2, KS = 32, SP, WP = 1>   M >= 128: weights as pre-split planes (k_split_weights), pixels fp32
//   igemm_kernel< 64, 192, 4, 0, 2, 32, SP, 1>                  M = 64
//   igemm_kernel< 64, 256, 4, 2, 2, 32, SP, 0>                  packed 7x7 stem: both operands fp32, split in the kernel
//   (SP = 0: the same tiles with fp32 MFMA; KS = 16 / NS = 4 where Ci % 32 != 0)
//  * D = Wp * Xg^T with output CHANNELS on the MFMA row axis, so each lane ends up with 4 consecutive channels of one pixel
//    -> one 16-B NHWC store.  256 threads = 4 waves, each wave owns a 64 x 64 (64 x 48) sub-tile; two blocks per CU.
//  * Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of NS = 2 stages of
//    KS = 32 k: a whole 128-B line per DMA row.  WP = 1: the weight stage is [3 planes][BM rows][64 B] (24 KB for BM = 128), the
//    pixel stage fp32 [BN][32] (16 KB).  The LDS image is lane-linear (DMA destination = base + lane * 16 B); the bank swizzle
//    slot = chunk ^ ((row >> 1) & SM) is applied on the SOURCE address and on the read (conflict-free ds_read_b128).
//  * One K-step (WP form, "LA2"): wait for the step's DMA (counted vmcnt) -> barrier -> read ALL of the step's fragments into
//    registers (12 weight-plane + 8 pixel reads) -> barrier -> the slot is free: issue the DMA of step s + 2 (two stages of
//    lookahead from a two-stage ring) -> 96 MFMAs with the pixel columns' fp32 -> 3 x bf16 split (44 VALU per column) pinned
//    two instructions into the shadow of every MFMA, one column ahead of its use.
//  * K order inside a 32-k block is permuted (lane group g supplies k = 4g..4g+3 and 16+4g..16+4g+3); legal because A and B use
//    the same permutation.  Tap offsets are scalar kernel arguments; padding and tail rows read a 16-B zero page.
//  * STREAM-K scheduling: ResNet's pixel counts are 49*2^k, so a one-tile-per-block grid leaves the last round of the CUs partly
//    empty.  The launch is persistent: the (tile, K-step) space is cut into equal contiguous ranges, one per block.  A block
//    whose range ends inside a tile stores its partial accumulators to a slab, releases them at agent scope and bumps the tile's
//    arrival counter; the LAST arriver re-reads every partial of the tile in segment order (deterministic) and runs the epilogue.
//  * Epilogue variants (runtime-uniform): raw store + per-channel sum / sumsq partials (train-mode BN statistics, fixed reduction
//    order), or folded eval-BN affine + residual + ReLU, or plain residual add (dgrad).
//  * block -> range map is XCD-aware (blocks b, b+8, .. share an XCD and get adjacent ranges: neighbouring tiles share an L2).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "split3.h"

// STEM: 0 = 3x3 / 1x1 convs, 1 = stem with K = [kh][kw padded][4] (EfficientNet's 3x3, ResNet's padded form), 2 = ResNet's
// packed 7x7 stem over the zero-framed NHWC3 input
// SP: 0 = fp32 products on v_mfma_f32_16x16x4_f32; 9 / 6 = fp32 products as exact bf16 partial products on
// v_mfma_f32_16x16x32_bf16 (split3.h; KS = 32 only: a stage is one bf16 k-step)
// WP (with SP): the weight operand arrives as the bf16 planes of k_split_weights (p.Wsp) -- no split arithmetic for A in the
// kernel.  Its LDS stage is [3 planes][BM rows][64 B] (one LDS-DMA instruction = 16 rows of one plane, the 64-B-row swizzle
// slot = chunk ^ ((row>>1)&3)), 24 KB for BM = 128 where the fp32 tile takes 16.
template <int BM, int BN, int WN, int STEM, int NS = 4, int KS = 16, int SP = 0, int WP = 0>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p)
{
#if __HIP_DEVICE_COMPILE__     // the host pass only needs the launch stub (the body uses gfx950-only builtins)
    static_assert(SP == 0 || KS == 32, "the split form takes one 32-k stage per MFMA step");
    static_assert(WP == 0 || (SP != 0 && STEM == 0 && BM % 64 == 0), "weight planes: split form, no stem");
    constexpr int WM = 4 / WN;
    constexpr int FR = BM / WM / 16, FC = BN / WN / 16;   // MFMA tiles per wave (rows, cols)
    static_assert(FR * FC == 12 || FR * FC == 16 || FR * FC == 32, "wave tile is 64x64, 64x48 (or 128x64 / 64x128)");
    static_assert(KS == 16 || (KS == 32 && (STEM == 0 || STEM == 2)), "K per stage is 16, or 32 (a whole 128-B line per DMA row)");
    constexpr int D = NS - 1;                   // LDS ring of NS stages of KS k; DMA runs D steps ahead
    constexpr int STG_A = WP ? 3 * BM * 16 : BM * KS, STG_B = BN * KS;   // floats per stage (planes: 3 x BM x 64 B)
    constexpr int CPR = KS / 4;                 // 16-B chunks per row of a stage (4 or 8)
    constexpr int RPI = 64 / CPR;               // rows one LDS-DMA instruction stages (16 or 8)
    constexpr int SM = CPR - 1;                 // swizzle mask: chunk c of row r lives in slot c ^ ((r>>1)&SM)
    constexpr int GA = WP ? 3 * BM / 64 : BM / (4 * RPI), GB = BN / (4 * RPI);   // DMA instructions each wave issues per step
    constexpr int PER = GA + GB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [NS][BM][KS]
    float* Bs = smem + NS * STG_A;    // [NS][BN][KS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    const int drow = lane / CPR, dpos = lane % CPR;        // LDS-DMA: row inside an RPI-row group, 16-B slot
    // fragment reads: row 16r+i, k-chunk g lives in slot g ^ ((row>>1)&SM)  (conflict-free ds_read_b128);
    // (row>>1)&SM == (li>>1)&SM for every row tile because 16r/2 is a multiple of 8
    const int aoff = (wm * (16 * FR) + li) * KS;
    const int boff = (wn * (16 * FC) + li) * KS;
    const int fsw = (li >> 1) & SM;

    const int nsteps = p.nsteps;                // steps of KS k
    const int Ktot = p.wrow ? p.wrow : nsteps * KS;      // floats per row of W
    const int HWg = p.Hg * p.Wg;
    const int npix = p.imgs_per_group * HWg;
    const int tiles_pg = p.tilesM * p.tilesN;

    // XCD-aware bijective remap of the block id to a range index
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int rbk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const long long S = p.steps_per_block;
    long long w = (long long)rbk * S;
    const long long wend = min(p.total_steps, w + S);

    while (w < wend) {
        const int tile = (int)(w / nsteps);
        const int k0 = (int)(w - (long long)tile * nsteps);
        const int k1 = min(nsteps, k0 + (int)(wend - w));
        const bool seg_first_of_block = (w == (long long)rbk * S);
        w += k1 - k0;
        const int grp = tile / tiles_pg;
        const int tl = tile - grp * tiles_pg;
        // tile order inside a group: pixel tiles fastest (tn_fast) keeps ONE M-tile's weight slice hot in
        // an XCD's L2 while its blocks walk the pixel tiles; channel tiles fastest shares the X tile instead
        const int tm = p.tn_fast ? tl / p.tilesN : tl % p.tilesM;
        const int tn = p.tn_fast ? tl % p.tilesN : tl / p.tilesM;
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- LDS-DMA source state: this lane stages row 16*(wave*G+q)+drow of each operand -------
        // source-side swizzle (floats) of the chunk this lane stages, per row group
        auto csrc_of = [&](int group) { return (dpos ^ (((RPI * group + drow) >> 1) & SM)) << 2; };
        const int csrc = csrc_of(0);                               // KS == 16: the same for every group
        const float* asrc[GA];
        bool av[GA];                                               // rows past M (M % BM != 0) read the zero page
#pragma unroll
        for (int q = 0; q < GA; ++q) {
            if constexpr (WP) {
                // DMA job j = wave*GA + q stages 16 rows (group j % (BM/16)) of plane j / (BM/16); lane = (row lane>>2, slot lane&3)
                const int j = wave * GA + q, plane = j / (BM / 16), group = j % (BM / 16);
                const int m = m0 + 16 * group + (lane >> 2);
                av[q] = m < p.M;
                const int chunk = (lane & 3) ^ ((lane >> 3) & 3);          // source-side swizzle of the 64-B rows
                asrc[q] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.Wsp) +
                                                         (size_t)(av[q] ? m : 0) * nsteps * 192 + plane * 64 + chunk * 16);
            } else {
                const int m = m0 + RPI * (wave * GA + q) + drow;
                av[q] = m < p.M;
                asrc[q] = p.W + (size_t)(av[q] ? m : 0) * Ktot + csrc_of(wave * GA + q);
            }
        }
        int ih0[GB], iw0[GB], xb[GB];
        bool rv[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            const int n = n0 + RPI * (wave * GB + q) + drow;
            rv[q] = n < npix;
            const int nn = rv[q] ? n : 0;
            const int img = nn / HWg;
            const int rem = nn - img * HWg;
            const int hg = rem / p.Wg;
            const int wg = rem - hg * p.Wg;
            ih0[q] = hg * p.sg;
            iw0[q] = wg * p.sg;
            xb[q] = (grp * p.imgs_per_group + img) * (p.Hi * p.Wi * p.Ci);
        }
        // Non-stem convs: everything a K-step's DMA needs is prepared once per tile segment, so that the
        // per-step issue path is a handful of scalar ops and 3 VALU ops per DMA (no division, no scalar
        // memory load, no branch): pixoff = element offset of the lane's (pixel, chunk) at tap (0,0),
        // vmask bit t = tap t stays inside the image for that pixel.  Tap offsets come from p.tapcode
        // (4 bits per tap: dh+1 | (dw+1)<<2, all taps of 3x3/1x1 convs and their dgrad classes are in [-1,1]).
        int pixoff[GB];
        unsigned vmask[GB];
        if constexpr (!STEM) {
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                pixoff[q] = xb[q] + (ih0[q] * p.Wi + iw0[q]) * p.Ci + csrc_of(wave * GB + q);
                unsigned vm = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const unsigned f = (unsigned)(p.tapcode >> (4 * t)) & 15u;
                    const int ih = ih0[q] + (int)(f & 3u) - 1, iw = iw0[q] + (int)(f >> 2) - 1;
                    if (t < p.ntaps && rv[q] && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) vm |= 1u << t;
                }
                vmask[q] = vm;
            }
        }
        // issue cursor (tap, 16-channel chunk) of the next K-step to be issued: K is walked channel-chunk-
        // major, tap-minor, so the shifted re-reads of an input pixel hit L1/L2 instead of coming back
        // from MALL/HBM after the other channels have streamed through.  issue() is called for
        // s = k0, k0+1, ... in order.
        int icc = STEM ? 0 : k0 / p.ntaps;
        int it = STEM ? 0 : k0 - icc * p.ntaps;
        const long long zoff = reinterpret_cast<const char*>(p.zeros) - reinterpret_cast<const char*>(p.X);
        auto issue = [&](int s) {
            const int slot = (s - k0) % NS;
            typedef __attribute__((address_space(3))) void lds_void;
            if constexpr (STEM != 0) {
              if constexpr (STEM == 2) {
                // packed 7x7 stem: chunk c of K = floats 4j .. 4j+3 of kernel row kh's 24-float window, which starts at the
                // framed pixel (2 oh + kh, 2 ow): no bounds to check, chunks past the 42 real ones read the zero page
                const int koff = s * KS;
                // rows of W hold Ktot floats (176): a 32-k stage past that (the sixth) would read the next row's head -- harmless
                // against the zero chunks of X only while those floats are finite, so such chunks come from the zero page
#pragma unroll
                for (int q = 0; q < GA; ++q)
                    __builtin_amdgcn_global_load_lds((av[q] && koff + csrc_of(wave * GA + q) < Ktot) ? asrc[q] + koff : p.zeros,
                                                     (lds_void*)(As + slot * STG_A + (wave * GA + q) * 256), 16, 0, 0);
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const int c = CPR * s + (csrc_of(wave * GB + q) >> 2);       // (KS == 16: the same chunk for every group)
                    const int kh = c / 6, j = c - 6 * kh;
                    const bool cv = c < 42;
                    const float* src = (rv[q] && cv) ? p.X + (size_t)(xb[q] + ((ih0[q] + kh) * p.Wi + iw0[q]) * 3 + 4 * j) : p.zeros;
                    __builtin_amdgcn_global_load_lds(src, (lds_void*)(Bs + slot * STG_B + (wave * GB + q) * 256), 16, 0, 0);
                }
              } else {
                // stem: K = (kh, kw padded to 8 or 4, ci padded to 4); step = (half) a kernel row, chunk = kw
                const int kw = ((s & p.stem_h2) << 2) + (csrc >> 2);
                const int dh = (s >> p.stem_h2) - p.stem_pad, dw = kw - p.stem_pad;
                const bool cv = kw < p.stem_kw;
                const int koff = s * 16;
#pragma unroll
                for (int q = 0; q < GA; ++q)
                    __builtin_amdgcn_global_load_lds(av[q] ? asrc[q] + koff : p.zeros,
                                                     (lds_void*)(As + slot * STG_A + (wave * GA + q) * 256), 16, 0, 0);
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const int ih = ih0[q] + dh, iw = iw0[q] + dw;
                    const bool ok = rv[q] && cv && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                    const float* src = ok ? p.X + (size_t)(xb[q] + (ih * p.Wi + iw) * p.Ci) : p.zeros;
                    __builtin_amdgcn_global_load_lds(src, (lds_void*)(Bs + slot * STG_B + (wave * GB + q) * 256), 16, 0, 0);
                }
              }
            } else {
                const unsigned f = (unsigned)(p.tapcode >> (4 * it)) & 15u;
                const int dh = (int)(f & 3u) - 1, dw = (int)(f >> 2) - 1;
                const int tapoff = (dh * p.Wi + dw) * p.Ci + icc * KS;       // scalar
                // floats from the row start: fp32 rows hold K floats, plane rows 48 floats (192 B) per 32-k block
                const int koff = WP ? (it * (p.Ci >> 5) + icc) * 48 : it * p.Ci + icc * KS;      // scalar
#pragma unroll
                for (int q = 0; q < GA; ++q)
                    __builtin_amdgcn_global_load_lds(av[q] ? asrc[q] + koff : p.zeros,
                                                     (lds_void*)(As + slot * STG_A + (wave * GA + q) * 256), 16, 0, 0);
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const long long xo = (long long)(pixoff[q] + tapoff) * 4;
                    const long long o = ((vmask[q] >> it) & 1u) ? xo : zoff;
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.X) + o),
                                                     (lds_void*)(Bs + slot * STG_B + (wave * GB + q) * 256), 16, 0, 0);
                }
                if (++it == p.ntaps) { it = 0; ++icc; }
            }
        };

        f32x4 acc[FR][FC];
#pragma unroll
        for (int r = 0; r < FR; ++r)
#pragma unroll
            for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        // LA2 (weight-plane form): a wave holds ALL its fragments of a stage in registers before the first MFMA, so the stage's
        // slot is free as soon as everyone has read it -- a second barrier right after the reads lets the DMA of step s+2 start
        // at the TOP of step s: two stages of lookahead from a two-stage ring (a 96-MFMA stage is ~1 us, no longer enough to
        // cover a MALL / HBM round trip with one)
        constexpr bool LA2 = WP != 0 && NS == 2;
        issue(k0);
        if constexpr (LA2) {
            if (k0 + 1 < k1) issue(k0 + 1);
        } else {
#pragma unroll
            for (int d = 1; d < D; ++d)
                if (k0 + d < k1) issue(k0 + d);
        }
        for (int s = k0; s < k1; ++s) {
            // my DMA of step s has landed once at most the younger steps' instructions are outstanding
            const int younger = LA2 ? min(1, k1 - 1 - s) : min(D - 1, k1 - 1 - s);
            if (D >= 4 && younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // everyone's step-s data landed; everyone finished reading step s-1
            asm volatile("" ::: "memory");
            if constexpr (!LA2) {
                if (s + D < k1) issue(s + D);      // refill the slot step s-1 just vacated
            }
            const int slot = (s - k0) % NS;
            if constexpr (SP != 0) {
                // lane group lg supplies k = 4lg..4lg+3 and 16+4lg..16+4lg+3 of the stage as its 8 k-slots (same for A and B)
                const int sl0 = (lg ^ fsw) << 2, sl1 = ((4 + lg) ^ fsw) << 2;
                const float* A = As + slot * STG_A + aoff;
                const float* B = Bs + slot * STG_B + boff;
                if constexpr (WP) {
                    // Weight planes straight from LDS (plane rows are 16 floats = 64 B; chunk lg of row li + 16 r sits in slot
                    // lg ^ ((li>>1)&3)); the pixel columns are split one column AHEAD of the MFMAs that use them, and the
                    // schedule is pinned: two VALU instructions of the next column's split in the shadow of every MFMA (an MFMA
                    // holds the vector issue port for 8 of its 16 cycles: two plain VALU instructions are what fits).  Left to
                    // itself the compiler splits every column up front (6 VALU per MFMA at first, then 77 bare MFMAs).
                    constexpr int SPLIT_VALU = 44;                       // 4 pairs x (3 cvt_pk + 2 shl + 2 and + 4 sub)
                    constexpr int COL_MFMA = FR * SP;
                    static_assert(2 * COL_MFMA >= SPLIT_VALU, "a column's MFMAs cover the next column's split");
                    const float* Ap = As + slot * STG_A + (wm * (16 * FR) + li) * 16 + ((lg ^ ((li >> 1) & 3)) << 2);
                    sp_u32x4 ah[FR], am[FR], al[FR], bh[2], bm[2], bl[2];
                    f32x4 b0[FC], b1[FC];
#pragma unroll
                    for (int c = 0; c < FC; ++c) {
                        b0[c] = *reinterpret_cast<const f32x4*>(B + c * 16 * KS + sl0);
                        b1[c] = *reinterpret_cast<const f32x4*>(B + c * 16 * KS + sl1);
                    }
#pragma unroll
                    for (int r = 0; r < FR; ++r) {
                        ah[r] = *reinterpret_cast<const sp_u32x4*>(Ap + r * 256);
                        am[r] = *reinterpret_cast<const sp_u32x4*>(Ap + r * 256 + BM * 16);
                        al[r] = *reinterpret_cast<const sp_u32x4*>(Ap + r * 256 + 2 * BM * 16);
                    }
                    if constexpr (LA2) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();      // every wave holds its fragments of step s: the slot is free
                        asm volatile("" ::: "memory");
                        if (s + 2 < k1) issue(s + 2);
                    } else
                        __builtin_amdgcn_sched_group_barrier(0x100, 3 * FR + 2 * FC, 0);
                    split3(b0[0], b1[0], bh[0], bm[0], bl[0]);
                    __builtin_amdgcn_sched_group_barrier(0x002, SPLIT_VALU, 0);
#pragma unroll
                    for (int c = 0; c < FC; ++c) {
                        if (c + 1 < FC) split3(b0[c + 1], b1[c + 1], bh[(c + 1) & 1], bm[(c + 1) & 1], bl[(c + 1) & 1]);
#pragma unroll
                        for (int r = 0; r < FR; ++r)
                            acc[r][c] = mfma_split<SP>(ah[r], am[r], al[r], bh[c & 1], bm[c & 1], bl[c & 1], acc[r][c]);
                        if (c + 1 < FC) {
#pragma unroll
                            for (int i = 0; i < SPLIT_VALU / 2; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, COL_MFMA - SPLIT_VALU / 2, 0);
                        } else
                            __builtin_amdgcn_sched_group_barrier(0x008, COL_MFMA, 0);
                    }
                } else {
                sp_u32x4 ah[FR], am[FR], al[FR], bh[FC], bm[FC], bl[FC];
#pragma unroll
                for (int r = 0; r < FR; ++r)
                    split3(*reinterpret_cast<const f32x4*>(A + r * 16 * KS + sl0),
                           *reinterpret_cast<const f32x4*>(A + r * 16 * KS + sl1), ah[r], am[r], al[r]);
#pragma unroll
                for (int c = 0; c < FC; ++c)
                    split3(*reinterpret_cast<const f32x4*>(B + c * 16 * KS + sl0),
                           *reinterpret_cast<const f32x4*>(B + c * 16 * KS + sl1), bh[c], bm[c], bl[c]);
#pragma unroll
                for (int r = 0; r < FR; ++r)
#pragma unroll
                    for (int c = 0; c < FC; ++c)
                        acc[r][c] = mfma_split<SP>(ah[r], am[r], al[r], bh[c], bm[c], bl[c], acc[r][c]);
                }
            } else
#pragma unroll
            for (int h = 0; h < KS / 16; ++h) {
                const int sl = ((4 * h + lg) ^ fsw) << 2;
                const float* A = As + slot * STG_A + aoff + sl;
                const float* B = Bs + slot * STG_B + boff + sl;
                f32x4 a[FR], b[FC];
#pragma unroll
                for (int r = 0; r < FR; ++r) a[r] = *reinterpret_cast<const f32x4*>(A + r * 16 * KS);
#pragma unroll
                for (int c = 0; c < FC; ++c) b[c] = *reinterpret_cast<const f32x4*>(B + c * 16 * KS);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < FR; ++r)
#pragma unroll
                        for (int c = 0; c < FC; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
            }
        }
        __syncthreads();       // all LDS reads done before the fix-up / epilogue reuse the LDS

        // ---- stream-K fix-up: partial tiles meet in the slab ------------------------
        if (k0 != 0 || k1 != nsteps) {
            // slot 0: the block's first segment, slot 1: its last one (middle ones are whole tiles)
            float* mine = p.slab + (size_t)(rbk * 2 + (seg_first_of_block ? 0 : 1)) * (BM * BN);
#pragma unroll
            for (int r = 0; r < FR; ++r)
#pragma unroll
                for (int c = 0; c < FC; ++c)
                    *reinterpret_cast<f32x4*>(mine + ((r * FC + c) * 256 + tid) * 4) = acc[r][c];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
            __syncthreads();
            const long long t0 = (long long)tile * nsteps;
            const int b_first = (int)(t0 / S), b_last = (int)((t0 + nsteps - 1) / S);
            int* flag = reinterpret_cast<int*>(smem);
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // keep: the fence's own wait may be dropped
                const int old = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == b_last - b_first;
                if (last) {
                    __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = last;
            }
            __syncthreads();
            const bool last = *flag != 0;
            __syncthreads();                                          // flag word is reused as LDS below
            if (!last) continue;
            // last arriver: sum every segment of this tile in segment order (incl. its own, from the
            // slab) -> the result is independent of which block arrived last
#pragma unroll
            for (int r = 0; r < FR; ++r)
#pragma unroll
                for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int bb = b_first; bb <= b_last; ++bb) {
                const long long sstart = max(t0, (long long)bb * S);
                const float* src = p.slab + (size_t)(bb * 2 + (sstart == (long long)bb * S ? 0 : 1)) * (BM * BN);
#pragma unroll
                for (int r = 0; r < FR; ++r)
#pragma unroll
                    for (int c = 0; c < FC; ++c)
                        acc[r][c] += *reinterpret_cast<const f32x4*>(src + ((r * FC + c) * 256 + tid) * 4);
            }
        }

        // ---- epilogue -------------------------------------------------------------
        // acc[r][c][q] = D[m = m0 + wm*16FR + 16r + 4*lg + q][n = n0 + wn*16FC + 16c + li]
        const int mbase = m0 + wm * (16 * FR) + 4 * lg;
        if (p.stats) {
            float* red = smem;            // [WN][BM][2]
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < FC; ++c) {
                    s1 += acc[r][c];
                    s2 += acc[r][c] * acc[r][c];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float u = s1[q], v = s2[q];
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) {
                        u += __shfl_xor(u, d);
                        v += __shfl_xor(v, d);
                    }
                    if (li == 0) {
                        const int ml = wm * (16 * FR) + 16 * r + 4 * lg + q;
                        red[(wn * BM + ml) * 2 + 0] = u;
                        red[(wn * BM + ml) * 2 + 1] = v;
                    }
                }
            }
            __syncthreads();
            for (int ch = tid; ch < BM; ch += 256) {
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int ww = 0; ww < WN; ++ww) {
                    u += red[(ww * BM + ch) * 2 + 0];
                    v += red[(ww * BM + ch) * 2 + 1];
                }
                if (m0 + ch >= p.M) continue;
                float* st = p.stats + (size_t)(grp * p.tilesN + tn) * 2 * p.M;
                st[m0 + ch] = u;
                st[p.M + m0 + ch] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const int n = n0 + wn * (16 * FC) + 16 * c + li;
            if (n >= npix) continue;
            const int img = n / HWg;
            const int rem = n - img * HWg;
            const int hg = rem / p.Wg;
            const int wg = rem - hg * p.Wg;
            const size_t o = ((size_t)((grp * p.imgs_per_group + img) * p.Ho + hg * p.os + p.oh0) * p.Wo
                              + (wg * p.os + p.ow0)) * p.Co;
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                const int m = mbase + 16 * r;
                if (m >= p.M) continue;
                f32x4 v = acc[r][c];
                if (p.scale) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + m);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + m);
                    v = v * sc + sh;
                }
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o + m);
                if (p.relu == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (p.relu == 2) {                           // swish (EfficientNet eval epilogue)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fm_swish_f32(v[q]);
                }
                *reinterpret_cast<f32x4*>(p.Y + o + m) = v;
            }
        }
        __syncthreads();       // LDS (stats scratch) is re-staged by the next segment
    }
#endif
}

// bf16 planes of the weight matrices (kernels.h SplitJob): one thread = one 8-value chunk (k = 4g..4g+3, 16+4g..16+4g+3 of
// a 32-k block): two 16-B reads 64 B apart (4 threads = one 128-B line), three 16-B writes (4 threads = 64 B per plane)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ src_base, unsigned short* __restrict__ dst_base,
                                                            const SplitJob* __restrict__ jobs, int njobs)
{
#if __HIP_DEVICE_COMPILE__
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
    const SplitJob jb = jobs[j];
    const long long idx = (long long)((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x;
    if (idx >= (long long)jb.M * jb.nkb * 4) return;
    const int g = (int)(idx & 3);
    const long long blk = idx >> 2;                       // m * nkb + kb
    const float* src = (jb.src ? jb.src : src_base + jb.src_off) + blk * 32;
    sp_u32x4 H, M, L;
    split3(ld4(src + 4 * g), ld4(src + 16 + 4 * g), H, M, L);
    unsigned char* dst = reinterpret_cast<unsigned char*>(dst_base + jb.dst_off) + blk * 192 + g * 16;
    *reinterpret_cast<sp_u32x4*>(dst) = H;
    *reinterpret_cast<sp_u32x4*>(dst + 64) = M;
    *reinterpret_cast<sp_u32x4*>(dst + 128) = L;
#endif
}
int split_job_blocks(int M, int nkb) { return (int)(((long long)M * nkb * 4 + 255) / 256); }
void k_split_weights(const float* src_base, unsigned short* dst_base, const SplitJob* jobs, int njobs, int nblocks, hipStream_t s)
{
    if (njobs > 0) hipLaunchKernelGGL(split_weights_kernel, dim3(nblocks), dim3(256), 0, s, src_base, dst_base, jobs, njobs);
}

// Tile configuration.  Measured on MI355X (conv 3x3, 256 images, tools/probe_conv.py):
//   two 4-wave blocks per CU, 128x128 / 64x256 tiles (this build) : 100 / 116 / 117 / 118 TFLOP/s
//   one block per CU, 256x128 / 128x256 / 64x512 tiles (128x64 wave tiles):  90 / 107 / 116 / 115
// for layer1 / layer2 / layer3 / layer4; the kernel template supports both.
int igemm_tile_m(int M) { return M >= 128 ? 128 : 64; }
// 64-row layers: 192 pixels (wave tile 64 x 48) -- with the weight planes a 64 x 256 tile's two stages (88 KB) no longer fit
// twice into a CU's LDS, 64 x 192 takes 72 KB; the stem kernels keep 64 x 256
int igemm_tile_n(int M, bool stem) { return M >= 128 ? 128 : (stem ? 256 : 192); }
int igemm_max_blocks() { return 512; }    // 2 blocks per CU x 256 CUs (64-80 KB LDS each)

void launch_igemm(IgemmParams p, int groups, hipStream_t s)
{
    if (launch_conv1x1_stream(p, groups, s)) return;      // small-K 1x1 stride-1 convs stream through conv1x1.hip
    static bool attr_done = false;
    constexpr int LDS_L = 4 * (128 + 128) * 16 * 4;     // 64 KB
    constexpr int LDS_S = 4 * (64 + 256) * 16 * 4;      // 80 KB
    constexpr int LDS_P = 2 * (3 * 128 * 64 + 128 * 32 * 4);   // 80 KB: two stages of (weight planes + fp32 pixel rows)
    constexpr int LDS_T = 4 * (64 + 192) * 16 * 4;             // 64 KB: 64 x 192 tiles
    constexpr int LDS_PT = 2 * (3 * 64 * 64 + 192 * 32 * 4);   // 72 KB: 64 x 192 with weight planes
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0>), LDS_L, "igemm_kernel<128, 128, 2, 0>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0>), LDS_T, "igemm_kernel<64, 192, 4, 0>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0, 2, 32>), LDS_T, "igemm_kernel<64, 192, 4, 0, 2, 32>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0, 2, 32, 9>), LDS_T, "igemm_kernel<64, 192, 4, 0, 2, 32, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0, 2, 32, 6>), LDS_T, "igemm_kernel<64, 192, 4, 0, 2, 32, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0, 2, 32, 9, 1>), LDS_PT, "igemm_kernel<64, 192, 4, 0, 2, 32, 9, 1>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 192, 4, 0, 2, 32, 6, 1>), LDS_PT, "igemm_kernel<64, 192, 4, 0, 2, 32, 6, 1>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0, 2, 32>), LDS_L, "igemm_kernel<128, 128, 2, 0, 2, 32>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0, 2, 32, 9, 1>), LDS_P, "igemm_kernel<128, 128, 2, 0, 2, 32, 9, 1>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0, 2, 32, 6, 1>), LDS_P, "igemm_kernel<128, 128, 2, 0, 2, 32, 6, 1>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0, 2, 32, 9>), LDS_L, "igemm_kernel<128, 128, 2, 0, 2, 32, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2, 0, 2, 32, 6>), LDS_L, "igemm_kernel<128, 128, 2, 0, 2, 32, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 256, 4, 1>), LDS_S, "igemm_kernel<64, 256, 4, 1>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 256, 4, 2>), LDS_S, "igemm_kernel<64, 256, 4, 2>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 256, 4, 2, 2, 32, 9>), LDS_S, "igemm_kernel<64, 256, 4, 2, 2, 32, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<64, 256, 4, 2, 2, 32, 6>), LDS_S, "igemm_kernel<64, 256, 4, 2, 2, 32, 6>");
        attr_done = true;
    }
    // K per LDS stage: 32 (two stages) stages a whole 128-B line per DMA row -- the L1 hands out whole
    // lines, so the 64-B rows of the 16-k stages use half of what it moves.  Needs Ci % 32 == 0.
    static const int ks32_mode = fm_tune("FM_KS32", 2);
    const bool ks32 = !p.stem_kw && p.Ci % 32 == 0 && (ks32_mode == 2 || (ks32_mode == 1 && p.M < 128));
    if (ks32) p.nsteps /= 2;           // the caller counts 16-k steps
    const int split = p.sp;
    // the packed 7x7 stem in the split form: 11 steps of 16 k become 6 of 32, the rows of W keep their 176 floats
    const bool stem_split = p.stem_kw && p.stem3 && split != 0;
    if (stem_split) { p.wrow = p.nsteps * 16; p.nsteps = (p.nsteps + 1) / 2; }
    const long long T = (long long)p.tilesM * p.tilesN * groups;
    p.total_steps = T * p.nsteps;
    p.tap_minor = 1;
    p.tapcode = 0;
    for (int t = 0; t < p.ntaps; ++t)      // taps of 3x3 / 1x1 convs and of their dgrad parity classes lie in [-1, 1]
        p.tapcode |= (unsigned long long)(((p.dh[t] + 1) & 3) | (((p.dw[t] + 1) & 3) << 2)) << (4 * t);
    static const int tn_fast = fm_tune("FM_TN_FAST", 2);
    // weights of one M-tile: BM rows x K floats; beyond ~1 MB per M-tile the all-M-tiles working set no longer fits L2
    p.tn_fast = tn_fast == 1 ? 1 : (tn_fast == 2 ? (p.tilesM > 1 && (long long)p.M * p.nsteps * 64 > (2LL << 20)) : 0);
    // persistent grid: every block slot whenever there are >= 4 K-steps for each of them, otherwise
    // one tile per block.  FM_IGEMM_BLOCKS overrides the grid (tests force odd splits so that every
    // fix-up path runs on small shapes).
    static const int forced = getenv("FM_IGEMM_BLOCKS") ? atoi(getenv("FM_IGEMM_BLOCKS")) : 0;
    int nblk = p.total_steps >= 4LL * igemm_max_blocks() ? igemm_max_blocks()
                                                         : (int)std::min<long long>(igemm_max_blocks(), T);
    if (forced > 0) nblk = (int)std::min<long long>(std::min(forced, igemm_max_blocks()), p.total_steps);
    p.steps_per_block = (int)((p.total_steps + nblk - 1) / nblk);
    dim3 grid(nblk);
    if (stem_split && split == 9)
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4, 2, 2, 32, 9>), grid, dim3(256), LDS_S, s, p);
    else if (stem_split)
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4, 2, 2, 32, 6>), grid, dim3(256), LDS_S, s, p);
    else if (p.stem_kw && p.stem3)
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4, 2>), grid, dim3(256), LDS_S, s, p);
    else if (p.stem_kw)
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4, 1>), grid, dim3(256), LDS_S, s, p);
    else if (p.M >= 128) {
        static const int planes = fm_tune("FM_WPLANES", 1);
        if (ks32 && split == 9 && p.Wsp && planes)
            hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0, 2, 32, 9, 1>), grid, dim3(256), LDS_P, s, p);
        else if (ks32 && split == 6 && p.Wsp && planes)
            hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0, 2, 32, 6, 1>), grid, dim3(256), LDS_P, s, p);
        else if (ks32 && split == 9) hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0, 2, 32, 9>), grid, dim3(256), LDS_L, s, p);
        else if (ks32 && split == 6) hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0, 2, 32, 6>), grid, dim3(256), LDS_L, s, p);
        else if (ks32) hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0, 2, 32>), grid, dim3(256), LDS_L, s, p);
        else hipLaunchKernelGGL((igemm_kernel<128, 128, 2, 0>), grid, dim3(256), LDS_L, s, p);
    }
    else {
        static const int planes = fm_tune("FM_WPLANES", 1);
        if (ks32 && split == 9 && p.Wsp && planes)
            hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0, 2, 32, 9, 1>), grid, dim3(256), LDS_PT, s, p);
        else if (ks32 && split == 6 && p.Wsp && planes)
            hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0, 2, 32, 6, 1>), grid, dim3(256), LDS_PT, s, p);
        else if (ks32 && split == 9) hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0, 2, 32, 9>), grid, dim3(256), LDS_T, s, p);
        else if (ks32 && split == 6) hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0, 2, 32, 6>), grid, dim3(256), LDS_T, s, p);
        else if (ks32) hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0, 2, 32>), grid, dim3(256), LDS_T, s, p);
        else hipLaunchKernelGGL((igemm_kernel<64, 192, 4, 0>), grid, dim3(256), LDS_T, s, p);
    }
}
