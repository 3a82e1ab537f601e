// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions from block-major bf16 planes, "ring" form, gfx950 (ResNet-18 planes mode).
//
// Reference op replaced: the weight-gradient half of loss.backward() (utils/local_training.py:674, 965, 1191) for the 3x3
// stride-1 convolutions of torchvision's resnet18 (model/all_models.py:53-54) on the 56 x 56 and 28 x 28 maps.
//
//   dW[m][tap][ci] = sum_p dY[p][m] * X[p (+) tap][ci]          p: output pixel over ALL images
//
// Same arithmetic as pwgrad.hip (split3.h: every fp32 product as SP = 6 / 9 exact bf16 partial products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation, slabs summed in a fixed order).  What differs is how X reaches the matrix cores.
// pwgrad.hip stages X once per TAP (nine shifted copies of the same pixels per K-step: 48 KB of LDS-DMA for 36 MFMAs per wave,
// which is what bounded it on the 64-channel layers: 0.38 of the roofline).  Here the contraction runs over PADDED pixel
// positions: the image rows are laid out with one zero position after each row and one zero row after each image
// (q = Z0 + (img (H + 1) + oh)(W + 1) + ow), so that every tap is a CONSTANT shift (kh - 1)(W + 1) + (kw - 1) of q and the zero
// padding of the convolution is part of the sequence.  A block keeps the last 256 positions of X in a ring in LDS
// ([channel block][plane][256 rows][64 B]); a K-step of 32 positions appends 32 rows to it (each X row is fetched ONCE per
// block, padding rows are out-of-range offsets = zeros, no HBM traffic) and reads the nine taps' fragments at nine row offsets.
//  * block tile: 64 output channels x (9 taps x 64 input channels) = 64 x 576, 8 waves as 2 (m) x 4 (n): wave (wm, wn) holds the
//    output channels 32 wm .. + 31 and, for EVERY tap, the 16 input channels of quarter wn: 2 x 9 MFMA tiles, 72 accumulator
//    registers, 108 MFMAs per K-step against 12 KB (dY) + 12 KB (X) of LDS-DMA;
//  * fragments: ds_read_b64_tr_b16 out of pixel-major rows as in pwgrad.hip; the ring rows are swizzled in 32-B halves by
//    (row >> 3) & 1, which a step of 32 rows preserves: a fragment address advances by 2048 B modulo the 16-KB region
//    (one add, one and-or); conflict-free at every shift (rows r and r + 8 of a read always differ in that bit);
//  * padding positions cost MFMA work: (W + 1)(H + 1) / (W H) = 1.036 at 56 x 56, 1.07 at 28 x 28 -- the form is used where that
//    is small (pwgrad_ring_takes);
//  * pipeline: three dY stages and five ring units of lookahead; one barrier per step (before column 4 of 9); a wave issues
//    exactly three LDS-DMA instructions per step, and the barrier's wait leaves the youngest group in flight (vmcnt(3)): the DMA
//    has two steps to land;
//  * the position axis is split over blocks (tiles x splits = one block per CU); blocks of one split share an XCD (they read
//    the same dY / X rows through one L2); slabs are summed by reduce_slabs in a fixed order.
// Roofline: bf16 MFMA dense peak / SP = 416.7 TFLOP/s of fp32 products (SP = 6).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split3.h"

#if __HIP_DEVICE_COMPILE__
template <int IMM> __device__ __forceinline__ uint2 pr_read_tr(unsigned addr)
{
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
#endif

namespace {
constexpr int PR_RING = 256;                  // ring rows (padded positions) per (channel block, plane)
constexpr int PR_REG = PR_RING * 64;          // bytes of one region
constexpr int PR_SA = 2 * 3 * 2048;           // one dY stage: [2 channel blocks][3 planes][32 rows][64 B]
constexpr int PR_NSA = 3;
constexpr int PR_LDS = PR_REG + 6 * PR_REG + PR_NSA * PR_SA;      // 148 KB (the first region is alignment slack)
}  // namespace

template <int SP>
__global__ __launch_bounds__(512, 2) void pwgrad_ring_kernel(const PwgradParams p)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int FR = 2, FC = 9;
    constexpr int REG = PR_REG, SA = PR_SA;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned ring0 = (lds0 + (unsigned)(REG - 1)) & ~(unsigned)(REG - 1);      // regions aligned to their size: wrap = and-or
    const unsigned As = ring0 + 6 * REG;

    // block -> (tile, split): the blocks of a split sit on one XCD (block ids go round-robin over the eight XCDs)
    const int tiles = p.tilesM * p.tilesN;
    int tile, split;
    {
        const int bid = blockIdx.x;
        if (p.xcd_remap) {
            const int xcd = bid & 7, l = bid >> 3;
            tile = l % tiles;
            split = xcd * (p.splits >> 3) + l / tiles;
        } else {
            tile = bid % tiles;
            split = bid / tiles;
        }
    }
    const int tm = tile % p.tilesM, tc = tile / p.tilesM;
    const int m0 = tm * 64, cb0 = tc * 2;
    const int W = p.Wi, H = p.Hi, Wp = W + 1, Hp = H + 1, Z0 = Wp + 1;
    const int qbeg = split * p.q_per_split;
    const int qlim = qbeg + p.q_per_split;                     // positions from here on belong to the next split
    const int nsteps = (min(p.Qtot, qlim) - qbeg + 31) >> 5;   // (<= 0: a split past the end writes a zero slab)

    // padded position -> NHWC pixel index, -1 = padding (umulhi by ceil(2^32 / d) divides exactly below 2^32 / d)
    auto pix_of = [&](int q) -> int {
        const int t = q - Z0;
        const unsigned row = __umulhi((unsigned)t, p.magW);
        const unsigned col = (unsigned)t - row * (unsigned)Wp;
        const unsigned img = __umulhi(row, p.magH);
        const unsigned oh = row - img * (unsigned)Hp;
        const bool ok = t >= 0 && col < (unsigned)W && oh < (unsigned)H && img < (unsigned)p.nimg;
        return ok ? (int)((img * (unsigned)H + oh) * (unsigned)W + col) : -1;
    };

    // ---- LDS-DMA: this wave stages rows 16 h16 .. + 15 of every unit of 32 rows; the six (block, plane) jobs of dY and the six of X
    // are dealt so that every wave issues three instructions per step: waves 0 - 3 two of dY and one of X, waves 4 - 7 one and two
    constexpr unsigned OOB = 0x80000000u;
    const int h16 = wave & 1, q4 = wave >> 1;
    const int drow = 16 * h16 + (lane >> 2);
    const unsigned chunkoff = (unsigned)(((lane & 3) ^ (((drow >> 3) & 1) << 1)) * 16);      // source-side swizzle of the 32-B halves
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.dYp), 0, (unsigned)((size_t)(p.M >> 5) * 3 * p.npix * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.Xp), 0, (unsigned)((size_t)(p.Ci >> 5) * 3 * p.xpix * 64), 0x00020000);
    const int jA0 = q4, jA1 = q4 + 4, jB0 = (q4 + 2) & 3, jB1 = jB0 + 4;       // jA1 exists for q4 < 2, jB1 for q4 >= 2
    auto so_a = [&](int j) { return (unsigned)((size_t)(((m0 >> 5) + j / 3) * 3 + j % 3) * p.npix * 64); };
    auto so_b = [&](int j) { return (unsigned)((size_t)((cb0 + j / 3) * 3 + j % 3) * p.xpix * 64); };
    const unsigned soA0 = so_a(jA0), soA1 = so_a(jA1 < 6 ? jA1 : 0), soB0 = so_b(jB0), soB1 = so_b(jB1 < 6 ? jB1 : 0);
    auto voff = [&](int q, bool live) {
        const int px = live ? pix_of(q) : -1;
        return px >= 0 ? (unsigned)px * 64u + chunkoff : OOB;
    };
    auto issueA = [&](int sa, int slot) {                      // dY rows of step sa -> stage slot
        const int q = qbeg + 32 * sa + drow;
        const unsigned vo = voff(q, q < qlim);
        const unsigned dst = As + slot * SA + h16 * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(size_t)(dst + jA0 * 2048), 16, vo, soA0, 0, 0);
        if (q4 < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(size_t)(dst + jA1 * 2048), 16, vo, soA1, 0, 0);
    };
    auto issueB = [&](int u) {                                 // ring unit u = positions qbeg - 64 + 32 u .. + 31
        const unsigned vo = voff(qbeg - 64 + 32 * u + drow, true);
        const unsigned dst = ring0 + ((u & 7) * 32 + 16 * h16) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(size_t)(dst + jB0 * REG), 16, vo, soB0, 0, 0);
        if (q4 >= 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(size_t)(dst + jB1 * REG), 16, vo, soB1, 0, 0);
    };

    // ---- fragment reads (transposed, pwgrad.hip): a lane supplies the address of pixel row 8 lg + 4 j + (li >> 2), columns
    // 4 (li & 3) .. of a 16-channel half, and receives channel li of 4 pixels.  dY: tile r of the wave = half r of block wm
    const unsigned frow = (unsigned)((8 * lg + (li >> 2)) * 64 + (li & 1) * 8);
    const unsigned Af[2] = {As + wm * 6144 + frow + (unsigned)(((0 ^ (2 * (lg & 1))) + ((li & 3) >> 1)) * 16),
                            As + wm * 6144 + frow + (unsigned)(((2 ^ (2 * (lg & 1))) + ((li & 3) >> 1)) * 16)};
    // X: column c of the wave = tap c, half hhw of channel block cbw; row of position (step s, k) under the tap's shift:
    // (64 + 32 s + k + shift) mod 256 (the ring starts 64 positions before the split)
    const int hhw = wn & 1, cbw = wn >> 1;
    const unsigned cbbase = ring0 + cbw * 3 * REG;
    unsigned low[FC][2];
#pragma unroll
    for (int c = 0; c < FC; ++c) {
        const int shift = (c / 3 - 1) * Wp + (c % 3 - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (64 + 8 * lg + 4 * j + (li >> 2) + shift) & (PR_RING - 1);
            const int half = hhw ^ ((row >> 3) & 1);
            low[c][j] = (unsigned)(row * 64 + (2 * half + ((li & 3) >> 1)) * 16 + (li & 1) * 8);
        }
    }

    f32x4 acc[FR][FC];
#pragma unroll
    for (int r = 0; r < FR; ++r)
#pragma unroll
        for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    sp_u32x4 A0[FR][3], A1[FR][3], Bb[2][3];
#define PR_READA(SLOTOFF, R, DST)                                                                  \
    {                                                                                              \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                         \
            uint2 r0, r1;                                                                          \
            if (pl == 0) { r0 = pr_read_tr<0>(Af[R] + (SLOTOFF)); r1 = pr_read_tr<256>(Af[R] + (SLOTOFF)); }                \
            else if (pl == 1) { r0 = pr_read_tr<2048>(Af[R] + (SLOTOFF)); r1 = pr_read_tr<2048 + 256>(Af[R] + (SLOTOFF)); } \
            else { r0 = pr_read_tr<4096>(Af[R] + (SLOTOFF)); r1 = pr_read_tr<4096 + 256>(Af[R] + (SLOTOFF)); }              \
            DST[pl] = sp_u32x4{r0.x, r0.y, r1.x, r1.y};                                            \
        }                                                                                          \
    }
    // the tap's fragment of the CURRENT position of its addresses, which then advance by one step (32 rows, modulo the ring)
#define PR_READB(C, DST)                                                                           \
    {                                                                                              \
        const unsigned a0 = (low[C][0] & (unsigned)(REG - 1)) | cbbase, a1 = (low[C][1] & (unsigned)(REG - 1)) | cbbase;   \
        low[C][0] += 2048; low[C][1] += 2048;                                                      \
        const uint2 h0 = pr_read_tr<0>(a0), h1 = pr_read_tr<0>(a1);                                \
        const uint2 m0_ = pr_read_tr<REG>(a0), m1_ = pr_read_tr<REG>(a1);                          \
        const uint2 l0 = pr_read_tr<2 * REG>(a0), l1 = pr_read_tr<2 * REG>(a1);                    \
        DST[0] = sp_u32x4{h0.x, h0.y, h1.x, h1.y};                                                 \
        DST[1] = sp_u32x4{m0_.x, m0_.y, m1_.x, m1_.y};                                             \
        DST[2] = sp_u32x4{l0.x, l0.y, l1.x, l1.y};                                                 \
    }
#define PR_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    // the wait at the END of column C's MFMAs: the accumulators are operands, so that the machine scheduler cannot move the bare
    // wait up behind the column's first MFMA (which is where it put it: pconv.hip PC_LGKM0_COL)
#define PR_LGKM0_COL(C)                                                                            \
    do {                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][C]), "+v"(acc[1][C])::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define PR_MFMA(C, AC, BI_)                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < FR; ++r)                                                                            \
        acc[r][C] = mfma_split<SP>(Bb[BI_][0], Bb[BI_][1], Bb[BI_][2], AC[r][0], AC[r][1], AC[r][2], acc[r][C])

    if (nsteps > 0) {
        // ---- prologue: dY of steps 0, 1, 2; ring units 0 .. 6 = positions qbeg - 64 .. qbeg + 159 -------------------------------
        issueA(0, 0);
        issueA(1, 1);
        issueA(2, 2);
#pragma unroll 1
        for (int u = 0; u < 7; ++u) issueB(u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PR_READA(0, 0, A0[0]);
        PR_READA(0, 1, A0[1]);
        PR_READB(0, Bb[0]);
        PR_LGKM0();
        int slot0 = 0, slot1 = 1;              // dY stage of the current / the next step
        // One K-step (register parity PAR): column c's fragment sits in Bb[(c + PAR) & 1], read one column ahead.
        // Invariant at the barrier of step s (before column 4): every wave has waited for all but its youngest three DMA
        // instructions, i.e. for everything issued up to step s - 2: dY(s + 1) and the ring units up to s + 5, which covers the
        // rows the rest of step s and the whole of step s + 1 up to ITS barrier read (positions below qbeg + 32 s + 96 + 32).
        // Behind the barrier the stage of dY(s) (read during step s - 1) takes dY(s + 3), and ring unit s + 7 overwrites unit
        // s - 1 (positions up to qbeg + 32 s - 65; the lowest row still read is qbeg + 32 s - W - 2).
        auto step = [&](auto par_c, int s, sp_u32x4 (&Ac)[FR][3], sp_u32x4 (&An)[FR][3]) {
            constexpr int PAR = decltype(par_c)::value;
#define PR_COLUMN(C, EXTRA)                                                                        \
            {                                                                                      \
                PR_READB((C) + 1, Bb[((C) + 1 + PAR) & 1]);                                        \
                EXTRA;                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                PR_MFMA(C, Ac, ((C) + PAR) & 1);                                                   \
                PR_LGKM0_COL(C);                                                                   \
            }
            PR_COLUMN(0, (void)0)
            PR_COLUMN(1, (void)0)
            PR_COLUMN(2, (void)0)
            PR_COLUMN(3, (void)0)
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // (as pconv.hip: the two waves of a SIMD would issue their DMA at the same point; waves 4-7 issue theirs behind the step's
            // last MFMAs -- still three instructions per wave and step, which is what the counted wait above relies on)
            const bool late = (p.pw_flags & 1) && wave >= 4;
            if (!late) { issueA(s + 3, slot0); issueB(s + 7); }
            const unsigned a_nxt = (unsigned)(slot1 * SA);
            PR_COLUMN(4, PR_READA(a_nxt, 0, An[0]))
            PR_COLUMN(5, PR_READA(a_nxt, 1, An[1]))
            PR_COLUMN(6, (void)0)
            PR_COLUMN(7, (void)0)
#undef PR_COLUMN
            PR_READB(0, Bb[(0 + (PAR ^ 1)) & 1]);          // the next step's column 0
            __builtin_amdgcn_sched_barrier(0);
            PR_MFMA(8, Ac, (8 + PAR) & 1);
            PR_LGKM0_COL(8);
            if (late) { issueA(s + 3, slot0); issueB(s + 7); }
            slot0 = slot1;
            slot1 = slot1 == PR_NSA - 1 ? 0 : slot1 + 1;
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        for (int s = 0; s < nsteps; s += 2) {
            step(I0{}, s, A0, A1);
            if (s + 1 < nsteps) step(I1{}, s + 1, A1, A0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef PR_READA
#undef PR_READB
#undef PR_LGKM0
#undef PR_LGKM0_COL
#undef PR_MFMA

    // ---- epilogue: acc[r][c][q] = dW[m of (tile 2 wm + r, position li)][n of (tap c, block cb0 + cbw, positions 16 hhw + 4 lg + q)]
    // position j of a 32-channel block -> channel: chunk g = j >> 3 holds channels 4g..4g+3 (j & 7 < 4) and 16+4g..16+4g+3
    const int cib = p.Ci >> 5;
#pragma unroll
    for (int r = 0; r < FR; ++r) {
        const int jm = 16 * r + li, gm = jm >> 3, wi = jm & 7;
        const int m = m0 + 32 * wm + (wi < 4 ? 4 * gm + wi : 16 + 4 * gm + (wi - 4));
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const int jb = c * cib + cb0 + cbw;                // column block (tap, channel block), tap-major
            const int gn = 2 * hhw + (lg >> 1);
            const int n = jb * 32 + ((lg & 1) ? 16 + 4 * gn : 4 * gn);
            *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + m) * p.Nw + n) = acc[r][c];
        }
    }
#endif
}

// the ring form takes: 3x3, stride 1, pad 1; whole 64-channel tiles on both sides; rows short enough for the ring (a tap shift of
// W + 2 positions within the 64 the ring keeps behind the current step); maps from 28 pixels wide up -- below that the padding
// positions cost more MFMA work than the form saves (measured with the tuning build's FM_PWGRAD_RING_MINW=7: 14 x 14 maps 369 vs
// 346 us, 7 x 7 maps 366 vs 342 us per launch at 256 images; FM_PWGRAD_RING=0 there keeps pwgrad.hip everywhere).
bool pwgrad_ring_takes(const PwgradParams& p)
{
    static const int on = fm_tune("FM_PWGRAD_RING", 1), minw = fm_tune("FM_PWGRAD_RING_MINW", 28);
    if (!on || p.ksz != 3 || p.pad != 1 || p.stride != 1 || p.Ho != p.Hi || p.Wo != p.Wi) return false;
    if (p.M % 64 != 0 || p.Ci % 64 != 0 || p.Wi + 2 > 64 || p.Wi < minw) return false;
    const long long nimg = p.npix / ((long long)p.Ho * p.Wo);
    return nimg * (p.Hi + 1) * (p.Wi + 1) + p.Wi + 2 < (1LL << 24);
}

// returns the number of slabs written ([splits][M][Nw] in p.slab), 0 = nothing launched
int launch_pwgrad_ring(PwgradParams p, size_t slab_floats, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&pwgrad_ring_kernel<6>), PR_LDS, "pwgrad_ring_kernel<6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pwgrad_ring_kernel<9>), PR_LDS, "pwgrad_ring_kernel<9>");
        attr_done = true;
    }
    const int Wp = p.Wi + 1, Hp = p.Hi + 1;
    p.nimg = (int)(p.npix / ((long long)p.Ho * p.Wo));
    p.Qtot = Wp + 1 + p.nimg * Hp * Wp;
    p.magW = (unsigned)((1ULL << 32) / (unsigned)Wp) + 1u;
    p.magH = (unsigned)((1ULL << 32) / (unsigned)Hp) + 1u;
    p.tilesM = p.M / 64;
    p.tilesN = p.Ci / 64;
    const int tiles = p.tilesM * p.tilesN;
    // tiles x splits = ONE round of the 256 CUs; a split is at least 4 steps; FM_IGEMM_BLOCKS (tests) forces odd grids
    static const int forced = getenv("FM_IGEMM_BLOCKS") ? atoi(getenv("FM_IGEMM_BLOCKS")) : 0;
    int splits = std::max(1, (forced > 0 ? std::min(forced, 256) : 256) / tiles);
    splits = std::min(splits, std::max(1, p.Qtot / 128));
    splits = (int)std::min<size_t>((size_t)splits, std::max<size_t>(1, slab_floats / ((size_t)p.M * p.Nw)));
    if (splits >= 8) splits -= splits % 8;
    p.xcd_remap = splits % 8 == 0 ? 1 : 0;
    static const int flags = fm_tune("FM_PWGRAD_FLAGS", 1);      // (measured: -0.2 ms per step)
    p.pw_flags = flags;
    p.q_per_split = (((p.Qtot + splits - 1) / splits) + 31) & ~31;
    p.splits = splits;
    if (p.sp == 9) hipLaunchKernelGGL((pwgrad_ring_kernel<9>), dim3(tiles * splits), dim3(512), PR_LDS, s, p);
    else hipLaunchKernelGGL((pwgrad_ring_kernel<6>), dim3(tiles * splits), dim3(512), PR_LDS, s, p);
    return splits;
}
