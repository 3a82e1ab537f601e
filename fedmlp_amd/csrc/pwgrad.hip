// Convolution weight gradient whose BOTH operands arrive as block-major bf16 planes, gfx950 (ResNet-18 planes mode).
//
// Reference op replaced: the weight-gradient half of loss.backward() (utils/local_training.py:674, 965, 1191) for the 3x3
// convolutions of torchvision's resnet18 (model/all_models.py:53-54).
//
//   dW[m][n] = sum_p dY[p][m] * X[p (+) tap][ci]      m: out channel, n = (tap, ci), p: output pixel over ALL images
//
// Same arithmetic as wgrad.hip's split form (split3.h: every fp32 product as SP = 6 / 9 exact bf16 partial products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation), same pipeline as pconv.hip -- what differs from pconv is that the contraction
// runs over PIXELS, which are the rows of both plane tensors:
//  * operands: dYp[M/32][3][npix][32], Xp[Ci/32][3][xpix][32] (the planes the BatchNorm-backward / BatchNorm-apply / pool
//    kernels wrote for the data gradient and the forward); a K-step = 32 consecutive output pixels;
//  * LDS stage = [channel block][plane][32 pixel rows][64 B], filled by LDS-DMA (one instruction = 16 pixel rows of one plane
//    of one 32-channel block); the X rows of a tap are the output pixels' rows shifted by the tap: a wave decodes its 16 pixels
//    once per step (carries, no division), the tap shift and the block / plane are scalar offsets, rows outside the image or
//    past the split's end are an out-of-range offset (zeros);
//  * MFMA fragments come out of the pixel-major tiles transposed by ds_read_b64_tr_b16 (a lane supplies the address of pixel
//    row 8 lg + 4 j + (li >> 2), columns 4 (li & 3) .. of a 16-channel half block and receives channel li of 4 pixels); the
//    64-B rows are swizzled in 32-B halves, half ^= (row >> 3) & 1, on the DMA source side and on the read: conflict-free;
//  * ONE 8-wave block per CU, tile 32 FR (m) x 64 FC (n); the dY fragments are double-buffered in registers across steps,
//    the X fragments roll column by column; one barrier per step before its last column (pconv.hip);
//  * the pixel axis is split over blocks (tiles x splits = one block per CU); slabs are summed in a fixed order by
//    reduce_slabs (run-to-run deterministic, like the reference's cudnn.deterministic=True, main.py:36-37).
// Roofline: bf16 MFMA dense peak / SP = 416.7 TFLOP/s of fp32 products (SP = 6).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split3.h"

#if __HIP_DEVICE_COMPILE__
template <int IMM> __device__ __forceinline__ uint2 pw_read_tr(unsigned addr)
{
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
#endif

// FR: 16-channel m tiles per wave (4 or 2: BM = 128 / 64); FC: 16-column n tiles per wave (4 or 3: BN = 256 / 192)
// f(integral_constant<int, 0>) ... f(<N - 1>): a loop whose index is a template argument inside f (the LDS reads' offsets are immediates)
template <int N, typename F> __device__ __forceinline__ void pw_for(F f)
{
    f(std::integral_constant<int, 0>{});
    if constexpr (N > 1) f(std::integral_constant<int, 1>{});
    if constexpr (N > 2) f(std::integral_constant<int, 2>{});
    if constexpr (N > 3) f(std::integral_constant<int, 3>{});
}
template <int FR, int FC, int SP>
__global__ __launch_bounds__(512, 2) void pwgrad_kernel(const PwgradParams p)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int WN = 4;
    constexpr int BM = 32 * FR, BN = 64 * FC;
    constexpr int NBA = BM / 32, NBB = BN / 32;                // 32-channel blocks per operand tile
    constexpr int SA = NBA * 6144, SB = NBB * 6144;            // bytes per stage ([block][plane][32 rows][64 B])
    constexpr int JA = (NBA * 3 + 3) / 4, JB = (NBB * 3 + 3) / 4;   // DMA jobs per wave per stage (a wave stages ONE 16-row half)
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned As = lds0, Bs = lds0 + 2 * SA;

    const int tile = blockIdx.x, split = blockIdx.y;
    const int tm = tile % p.tilesM, tn = tile / p.tilesM;
    const int m0 = tm * BM;
    const int jb0 = tn * NBB;                                  // first (tap, channel block) column block of the tile
    const int cib = p.Ci >> 5;
    const long long pbeg = (long long)split * p.pix_per_split;
    const long long pend = min(p.npix, pbeg + p.pix_per_split);
    const int nsteps = (int)((pend - pbeg + 31) >> 5);
    const int Wi = p.Wi, Hi = p.Hi, Wo = p.Wo, Ho = p.Ho;

    // ---- LDS-DMA: this wave stages pixel rows 16 h .. 16 h + 15 of every step, jobs (block, plane) = q + 4 i ------------------
    constexpr unsigned OOB = 0x80000000u;
    const int h16 = wave & 1, q4 = wave >> 1;
    const int drow = 16 * h16 + (lane >> 2);                                     // pixel row inside the step
    const int chunk = (lane & 3) ^ (((drow >> 3) & 1) << 1);                     // source-side swizzle of the 32-B halves
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.dYp), 0, (unsigned)((size_t)(p.M >> 5) * 3 * p.npix * 64), 0x00020000);
    // X: the base sits pad * (Wi + 1) pixels BELOW the tensor: window origins (oh*stride - pad, ow*stride - pad) are never negative
    const int shift = p.pad * (Wi + 1);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.Xp) - (size_t)shift * 64), 0,
        (unsigned)(((size_t)cib * 3 * p.xpix + shift) * 64), 0x00020000);
    const unsigned voffA0 = (unsigned)(drow * 64 + chunk * 16);
    // pixel -> (image, oh, ow) of this lane's row, decoded once by division, then advanced by 32 pixels per step with carries
    int r_img, r_oh, r_ow;
    {
        const long long pix = pbeg + drow;
        const int HWo = Ho * Wo;
        r_img = (int)(pix / HWo);
        const int rem = (int)(pix - (long long)r_img * HWo);
        r_oh = rem / Wo;
        r_ow = rem - r_oh * Wo;
    }
    const int d_img = 32 / (Ho * Wo), rem32 = 32 - d_img * (Ho * Wo);
    const int dq = rem32 / Wo, dr = rem32 - dq * Wo;
    int sA = 0, sB = 0;                                        // steps issued so far (A and B advance separately)
    auto issueA = [&](int slot) {
        const long long prow = pbeg + (long long)sA * 32 + drow;
        const unsigned vo = prow < pend ? voffA0 : OOB;
#pragma unroll
        for (int i = 0; i < JA; ++i) {
            const int jj = q4 + 4 * i;
            if ((NBA * 3) % 4 != 0 && jj >= NBA * 3) break;
            const int b = jj / 3, pl = jj - 3 * b;
            const unsigned so = (unsigned)(((size_t)((m0 >> 5) + b) * 3 + pl) * p.npix * 64 + (size_t)(pbeg + (long long)sA * 32) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(size_t)(As + slot * SA + (b * 3 + pl) * 2048 + h16 * 1024), 16, vo, so, 0, 0);
        }
        ++sA;
    };
    auto issueB = [&](int slot) {
        // window origin of this lane's pixel in the (shifted) input, and which taps stay inside the image
        const int ih0 = r_oh * p.stride, iw0 = r_ow * p.stride;              // (+ kh - pad, + kw - pad: the base shift holds -pad)
        const long long prow = pbeg + (long long)sB * 32 + drow;
        unsigned rmask = 0, cmask = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < p.ksz && (unsigned)(ih0 + k - p.pad) < (unsigned)Hi) rmask |= 1u << k;
            if (k < p.ksz && (unsigned)(iw0 + k - p.pad) < (unsigned)Wi) cmask |= 1u << k;
        }
        if (prow >= pend) rmask = 0;
        const unsigned base = (unsigned)(((r_img * Hi + ih0) * Wi + iw0) * 64 + chunk * 16);
#pragma unroll
        for (int i = 0; i < JB; ++i) {
            const int jj = q4 + 4 * i;
            if ((NBB * 3) % 4 != 0 && jj >= NBB * 3) break;
            const int b = jj / 3, pl = jj - 3 * b;
            const int jb = jb0 + b;                                           // column block = (tap, channel block), tap-major
            const int tap = jb / cib, cb = jb - tap * cib;
            const int kh = tap / p.ksz, kw = tap - kh * p.ksz;
            const bool ok = jb < p.nblk_n && ((rmask >> kh) & 1u) && ((cmask >> kw) & 1u);
            const unsigned vo = ok ? base : OOB;
            const unsigned so = (unsigned)(((size_t)(cb * 3 + pl) * p.xpix + kh * Wi + kw) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(size_t)(Bs + slot * SB + (b * 3 + pl) * 2048 + h16 * 1024), 16, vo, so, 0, 0);
        }
        // advance this lane's pixel by 32
        int ow = r_ow + dr, oh = r_oh + dq, im = r_img + d_img;
        if (ow >= Wo) { ow -= Wo; ++oh; }
        if (oh >= Ho) { oh -= Ho; ++im; }
        r_ow = ow; r_oh = oh; r_img = im;
        ++sB;
    };

    // ---- fragment reads (transposed): tile t of an operand = half hh = t & 1 of block t >> 1; the two reads j = 0, 1 of a
    // plane deliver pixels 8 lg .. 8 lg + 7 of channel position 16 hh + li.  Row 8 lg + 4 j + (li >> 2); 16-B chunk
    // 2 hh + ((li & 3) >> 1) sits in slot chunk ^ (2 (lg & 1))
    const unsigned frow = (unsigned)((8 * lg + (li >> 2)) * 64 + (li & 1) * 8);
    const unsigned fh0 = frow + (unsigned)(((0 ^ (2 * (lg & 1))) + ((li & 3) >> 1)) * 16);     // hh = 0
    const unsigned fh1 = frow + (unsigned)(((2 ^ (2 * (lg & 1))) + ((li & 3) >> 1)) * 16);     // hh = 1
    // wave's first tile: A (m) tiles wm * FR + r, B (n) tiles wn * FC + c
    const unsigned Af[2] = {As + fh0, As + fh1}, Bf[2] = {Bs + fh0, Bs + fh1};

    f32x4 acc[FR][FC];
#pragma unroll
    for (int r = 0; r < FR; ++r)
#pragma unroll
        for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // RM (round 6, as pconv.hip): the step's MFMAs run row by row -- all FC columns of X in registers, the dY rows rolling through
    // two buffers -- instead of column by column with both steps' dY fragments resident: 48 registers fewer (room for a wave of a
    // streaming kernel of the other stream beside the two GEMM waves of a SIMD), same reads, MFMAs, stages, one barrier per step
    // (the column-major step of round 5 is in the history of this file)
    sp_u32x4 A0[1][3], A1[1][3], Bb[FC][3];
    // T = tile index inside the block tile (compile-time after unrolling: wm / wn are folded into the base address instead)
    const unsigned a_w = (unsigned)((wm * FR >> 1) * 6144), b_w = (unsigned)((wn * FC >> 1) * 6144);
    // (wm * FR and wn * FC are even for FR = 2, 4 and FC = 4; for FC = 3 the wave's first n tile may be odd: handled by hh0)
    const int b_hh0 = (wn * FC) & 1;
#define PW_READ(DST, BASE_ARR, WOFF, SLOTOFF, T)                                                                  \
    {                                                                                                             \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                         \
            uint2 r0, r1;                                                                                         \
            if (((T) & 1) == 0) {                                                                                 \
                r0 = pw_read_tr<((T) >> 1) * 6144 + 0>(BASE_ARR[0] + (WOFF) + (SLOTOFF) + pl * 2048);             \
                r1 = pw_read_tr<((T) >> 1) * 6144 + 256>(BASE_ARR[0] + (WOFF) + (SLOTOFF) + pl * 2048);           \
            } else {                                                                                              \
                r0 = pw_read_tr<((T) >> 1) * 6144 + 0>(BASE_ARR[1] + (WOFF) + (SLOTOFF) + pl * 2048);             \
                r1 = pw_read_tr<((T) >> 1) * 6144 + 256>(BASE_ARR[1] + (WOFF) + (SLOTOFF) + pl * 2048);           \
            }                                                                                                     \
            DST[pl] = sp_u32x4{r0.x, r0.y, r1.x, r1.y};                                                           \
        }                                                                                                         \
    }
#define PW_READA(SLOT, R, DST) PW_READ(DST, Af, a_w, (SLOT) * SA, R)
    // FC = 3: the wave's n tiles are wn * 3 + c; b_hh0 = 1 shifts the half pattern by one tile
#define PW_READB(SLOT, C, DST)                                                                     \
    {                                                                                              \
        if (FC == 4 || b_hh0 == 0) PW_READ(DST, Bf, b_w, (SLOT) * SB, C)                           \
        else PW_READ(DST, Bf, b_w, (SLOT) * SB, (C) + 1)                                           \
    }
#define PW_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    {
      if (nsteps > 0) {
        issueA(0);
        issueB(0);
        if (1 < nsteps) { issueA(1); issueB(1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PW_READA(0, 0, A0[0]);
        pw_for<FC>([&](auto ic) { constexpr int c = decltype(ic)::value; PW_READB(0, c, Bb[c]); });
        PW_LGKM0();
        // one K-step: at its start row 0 of dY(s) sits in A0[0] and the FC columns of X(s) in Bb (columns 1 .. possibly in flight:
        // counted waits in pass 0; a fragment = six LDS reads, the counter holds 15).  The barrier sits before the LAST pass: every
        // wave holds dY(s), X(s); dY(s + 1), X(s + 1) have landed: the stages take step s + 2, the last pass re-fills the registers
        auto rstep = [&](auto full_c, int s) {
            constexpr bool FULL = decltype(full_c)::value;
            const int sl = s & 1, sl1 = sl ^ 1;
            const bool more = FULL || s + 1 < nsteps;
#define PW_WAIT_ROW(R)                                                                                                          \
            do {                                                                                                                \
                if constexpr (FC == 4)                                                                                          \
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[R][0]), "+v"(acc[R][1]), "+v"(acc[R][2]), "+v"(acc[R][3])::"memory"); \
                else                                                                                                            \
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[R][0]), "+v"(acc[R][1]), "+v"(acc[R][2])::"memory");         \
                __builtin_amdgcn_sched_barrier(0);                                                                              \
            } while (0)
            PW_READA(sl, 1, A1[0]);
            __builtin_amdgcn_sched_barrier(0);
            pw_for<FC>([&](auto ic) {
                constexpr int c = decltype(ic)::value;
                if constexpr (c > 0) {
                    // columns c .. FC - 1 and row 1 may still be in flight (LDS returns in order): all but the youngest are waited for
                    constexpr int N = 6 * (FC - c) > 15 ? 15 : 6 * (FC - c);
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(Bb[c][0]), "+v"(Bb[c][1]), "+v"(Bb[c][2]), "+v"(acc[0][c - 1]) : "n"(N) : "memory");
                }
                acc[0][c] = mfma_split<SP>(Bb[c][0], Bb[c][1], Bb[c][2], A0[0][0], A0[0][1], A0[0][2], acc[0][c]);
            });
            PW_WAIT_ROW(0);
            if constexpr (FR == 4) {
                PW_READA(sl, 2, A0[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < FC; ++c) acc[1][c] = mfma_split<SP>(Bb[c][0], Bb[c][1], Bb[c][2], A1[0][0], A1[0][1], A1[0][2], acc[1][c]);
                PW_WAIT_ROW(1);
                PW_READA(sl, 3, A1[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < FC; ++c) acc[2][c] = mfma_split<SP>(Bb[c][0], Bb[c][1], Bb[c][2], A0[0][0], A0[0][1], A0[0][2], acc[2][c]);
                PW_WAIT_ROW(2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool late = (p.pw_flags & 1) && wave >= 4;        // (as pconv.hip: waves 4-7 issue behind the last pass's MFMAs)
            if (!late && (FULL || s + 2 < nsteps)) { issueA(sl); issueB(sl); }
            if (more) PW_READA(sl1, 0, A0[0]);
            __builtin_amdgcn_sched_barrier(0);
            pw_for<FC>([&](auto ic) {
                constexpr int c = decltype(ic)::value;
                acc[FR - 1][c] = mfma_split<SP>(Bb[c][0], Bb[c][1], Bb[c][2], A1[0][0], A1[0][1], A1[0][2], acc[FR - 1][c]);
                asm volatile("" : "+v"(acc[FR - 1][c]), "+v"(Bb[c][0]), "+v"(Bb[c][1]), "+v"(Bb[c][2]));
                __builtin_amdgcn_sched_barrier(0);
                if (more) PW_READB(sl1, c, Bb[c]);
            });
            __builtin_amdgcn_sched_barrier(0);
            if (late && (FULL || s + 2 < nsteps)) { issueA(sl); issueB(sl); }
            {
                constexpr int N = 6 * (FC - 1) > 15 ? 15 : 6 * (FC - 1);       // row 0 and column 0 of the next step are the oldest twelve
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(A0[0][0]), "+v"(A0[0][1]), "+v"(A0[0][2]), "+v"(Bb[0][0]), "+v"(Bb[0][1]),
                             "+v"(Bb[0][2]) : "n"(N) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#undef PW_WAIT_ROW
        };
        int s = 0;
        for (; s + 3 <= nsteps; ++s) rstep(std::true_type{}, s);
        for (; s < nsteps; ++s) rstep(std::false_type{}, s);
      }
    }
#undef PW_READ
#undef PW_READA
#undef PW_READB
#undef PW_LGKM0

    // ---- epilogue: acc[r][c][q] = dW[m of (m tile wm*FR + r, position li)][n of (n tile wn*FC + c, positions 4 lg + q)]
    // position j of a 32-channel block -> channel: chunk g = j >> 3 holds channels 4g..4g+3 (j & 7 < 4) and 16+4g..16+4g+3
#pragma unroll
    for (int r = 0; r < FR; ++r) {
        const int tM = wm * FR + r;
        const int jm = 16 * (tM & 1) + li, gm = jm >> 3, wi = jm & 7;
        const int m = m0 + 32 * (tM >> 1) + (wi < 4 ? 4 * gm + wi : 16 + 4 * gm + (wi - 4));
        if (m >= p.M) continue;
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const int tN = wn * FC + c;
            const int jb = jb0 + (tN >> 1);
            if (jb >= p.nblk_n) continue;
            const int gn = 2 * (tN & 1) + (lg >> 1);
            const int n = jb * 32 + ((lg & 1) ? 16 + 4 * gn : 4 * gn);
            *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + m) * p.Nw + n) = acc[r][c];
        }
    }
#endif
}

int pwgrad_tile_m(int M) { return M >= 128 ? 128 : 64; }
// column blocks (of 32) per N tile: 6 (BN = 192) when it tiles the k*k*Ci/32 blocks exactly and 8 does not
int pwgrad_tile_nb(int nblk_n) { return (nblk_n % 8 != 0 && nblk_n % 6 == 0) ? 6 : 8; }
bool pwgrad_takes(int M, int Ci, int ksz, long long npix, long long xpix, int Wi, int pad)
{
    if (M % 64 != 0 || (M > 64 && M % 128 != 0) || Ci % 32 != 0 || !((ksz == 3 && pad == 1) || (ksz == 1 && pad == 0))) return false;
    return ((long long)(Ci >> 5) * 3 * xpix + 2 * (Wi + 1)) * 64 < 0x7ff00000LL && (long long)(M >> 5) * 3 * npix * 64 < 0x7ff00000LL;
}

// returns the number of slabs written ([splits][M][Nw] in p.slab), 0 = nothing launched
int launch_pwgrad(PwgradParams p, size_t slab_floats, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int L44 = 2 * (4 + 8) * 6144, L43 = 2 * (4 + 6) * 6144, L24 = 2 * (2 + 8) * 6144, L23 = 2 * (2 + 6) * 6144;
    if (!attr_done) {
#define PW_ATTR(FR_, FC_, SP_, L_) set_max_dyn_lds(reinterpret_cast<const void*>(&pwgrad_kernel<FR_, FC_, SP_>), L_, "pwgrad_kernel")
        PW_ATTR(4, 4, 6, L44); PW_ATTR(4, 3, 6, L43); PW_ATTR(2, 4, 6, L24); PW_ATTR(2, 3, 6, L23);
        PW_ATTR(4, 4, 9, L44); PW_ATTR(4, 3, 9, L43); PW_ATTR(2, 4, 9, L24); PW_ATTR(2, 3, 9, L23);
#undef PW_ATTR
        attr_done = true;
    }
    static const int flags = fm_tune("FM_PWGRAD_FLAGS", 1);      // (measured: -0.2 ms per step)
    p.pw_flags = flags;
    p.nblk_n = p.ksz * p.ksz * (p.Ci >> 5);
    const int nb = pwgrad_tile_nb(p.nblk_n), bm = pwgrad_tile_m(p.M);
    p.tilesM = p.M / bm;
    p.tilesN = (p.nblk_n + nb - 1) / nb;
    const int tiles = p.tilesM * p.tilesN;
    // tiles x splits = ONE round of the 256 CUs (one block each); a split is at least 8 steps
    int splits = std::max(1, 256 / tiles);
    splits = (int)std::min<long long>(splits, std::max<long long>(1, p.npix / 256));
    splits = (int)std::min<size_t>(splits, std::max<size_t>(1, slab_floats / ((size_t)p.M * p.Nw)));
    p.pix_per_split = (int)((((p.npix + splits - 1) / splits) + 31) & ~31LL);
    splits = (int)((p.npix + p.pix_per_split - 1) / p.pix_per_split);
    dim3 grid(tiles, splits);
#define PW_LAUNCH(FR_, FC_, L_)                                                                              \
    do {                                                                                                     \
        if (p.sp == 9) hipLaunchKernelGGL((pwgrad_kernel<FR_, FC_, 9>), grid, dim3(512), L_, s, p);          \
        else hipLaunchKernelGGL((pwgrad_kernel<FR_, FC_, 6>), grid, dim3(512), L_, s, p);                    \
    } while (0)
    if (bm == 128) { if (nb == 8) PW_LAUNCH(4, 4, L44); else PW_LAUNCH(4, 3, L43); }
    else           { if (nb == 8) PW_LAUNCH(2, 4, L24); else PW_LAUNCH(2, 3, L23); }
#undef PW_LAUNCH
    return splits;
}
