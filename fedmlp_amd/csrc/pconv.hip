// Implicit-GEMM convolution (forward + data gradient) whose BOTH operands arrive as bf16 planes, gfx950.
//
// Reference op replaced: every nn.Conv2d forward inside net(images) (utils/local_training.py:657, 937-947, 983, 1030, 1178) and
// its input gradient inside loss.backward() (:674, 965, 1191) for the 3x3 / 1x1 convolutions of torchvision's resnet18
// (model/all_models.py:53-54).  Same arithmetic as igemm.hip's split form (split3.h): operands, accumulators and stored
// tensors are fp32 values; every fp32 product is formed as six (SP = 6) or nine exact bf16 partial products on
// v_mfma_f32_16x16x32_bf16 and accumulated in fp32.  What differs is WHO splits: here nobody does in this kernel.  The weight
// planes are made once per optimizer step (k_split_weights_bm), the activation planes by the kernel that PRODUCES the tensor
// (BatchNorm apply, max-pool, BatchNorm-backward apply, this kernel's own eval epilogue): x = h + m + l exactly, three bf16.
//
// Operand layout, "block-major planes":  P[K block of 32 channels][plane h|m|l][row][32] bf16 -- 64-B rows, 16 rows = 1 KB
// contiguous.  row = output channel for the weights (K block = (channel block, tap), tap-minor) and = NHWC pixel for the
// activations.  Inside a 32-channel block the 16-B chunk g holds channels 4g..4g+3 and 16+4g..16+4g+3 (both operands, so
// the contraction does not care; it is the order in which an MFMA output lane holds two row tiles, so this kernel's epilogue
// writes planes without a shuffle).
//
// Roofline: bf16 MFMA dense peak 2 500 TFLOP/s / SP products = 416.7 TFLOP/s of fp32 products (SP = 6).
// Structure:
//  * ONE 8-wave block per CU (two waves per SIMD), block tile (32 FR) x 256: 128 x 256 for M >= 128, 64 x 256 for M = 64;
//    wave tile (16 FR) x 64.
//  * LDS: two stages of [A planes | B planes] = 2 x 3 x (BM + 256) x 64 B (144 KB / 120 KB), filled by LDS-DMA
//    (buffer_load_dwordx4 ... lds, one instruction = 16 rows of one plane) through buffer descriptors: the per-lane part of an
//    address is ONE 32-bit offset per 16-row group, computed once per tile; everything that changes per step is a scalar
//    offset; padding / tail rows are an out-of-range offset, which the hardware answers with zeros.  Bank swizzle
//    slot = chunk ^ ((row>>1)&3) on the SOURCE side and on the read (conflict-free ds_read_b128 on 64-B rows).
//  * Row-major K-step (round 6, RM below): a step's MFMAs run row by row -- the step's four B columns sit in registers, the A rows
//    roll through two buffers, the next step's B columns are read behind the last row's MFMAs -- with ONE barrier per step, before
//    the last row.  (Rounds 4-5 ran column by column with both steps' A fragments resident: 48 registers more, the whole file
//    at two waves per SIMD; the row-major form leaves a wave of a streaming kernel of the other stream room beside the GEMM.)
//  * LDS reads are inline asm with explicit waits; every wait carries the registers it guards as operands (a bare wait between
//    two scheduling barriers gets moved by the machine scheduler: PC_LGKM0_COL).
//  * Stream-K (as igemm.hip): the (tile, K-step) space is cut into equal contiguous ranges, one per block; partial tiles meet
//    in a slab: the owner of a tile's head (the designated finisher) sums them in segment order (deterministic) and runs the
//    epilogue; the others store their part (one slab slot per block: a range starts inside a tile at most once).  ResNet's pixel counts are
//    49 * 2^k: one-tile-per-block grids leave 23 % of the CUs idle on three of the four layers.
//  * Epilogues: raw fp32 + BatchNorm partial sums (train forward), folded eval BatchNorm + residual + ReLU with fp32 and /
//    or plane output (eval forward), raw + residual with strided placement (data gradient, parity classes of stride 2).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split3.h"

#if __HIP_DEVICE_COMPILE__
template <int IMM> __device__ __forceinline__ sp_u32x4 pc_lds_read128(unsigned addr)
{
    sp_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
// x = q d + r for x < 2^24 (pixel indices): the float estimate of the quotient is off by at most one; a hardware-less integer
// division costs ~40 instructions, and a tile decodes / places a dozen pixel indices per lane
__device__ __forceinline__ void pc_divmod(int x, int d, float rcp, int& q, int& r)
{
    q = (int)((float)x * rcp);
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}
// v of the lane a DPP control selects (quad_perm / row_ror: lanes of the same row of 16): a cross-lane move on the VALU, where
// __shfl_xor is an LDS instruction (ds_bpermute_b32)
template <int CTRL> __device__ __forceinline__ float pc_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// f(integral_constant<int, 0>) ... f(<3>): a loop whose index is a template argument inside f (the LDS reads' offsets are immediates)
template <typename F> __device__ __forceinline__ void pc_for4(F f)
{
    f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{});
    f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
}
template <int SP> __device__ __forceinline__ f32x4 pc_mfma(const sp_u32x4 (&a)[3], const sp_u32x4 (&b)[3], f32x4 v)
{
    return mfma_split<SP>(a[0], a[1], a[2], b[0], b[1], b[2], v);
}
#endif

// FR x FC: 16 x 16 MFMA tiles per wave (wave tile 16 FR x 16 FC); block tile 32 FR x 64 FC.  NS: LDS stages (the DMA runs NS - 1
// steps ahead).  Instantiated: <4, 4, 2> = 128 x 256 (M >= 128) and <2, 4, 2> = 64 x 256 for the 64-channel layers.  Measured and not
// kept for those: <2, 3, 3> = 64 x 192 with three stages (two steps of DMA lookahead): 315 vs 294 us per 56 x 56 x 64 layer at 256
// images -- the layer is bound by LDS traffic per MFMA (every B row serves 64 output channels only), not by DMA latency
// TS ("tap rows shared", 3x3 / stride 1): the three taps of a kernel ROW read the same input pixels shifted by one, so the B stage is
// loaded ONCE per (channel block, kernel row) as the tile's pixels plus one halo pixel on either side ([3 planes][BN + 2 -> 272
// rows][64 B], two slots), and the tap's column shift is a row offset of the fragment reads (conflict-free at every offset,
// brute-force checked); a pixel whose shifted neighbour lies in another image row reads the stage's ZERO rows (rows BN + 2 .. 271,
// filled by out-of-range DMA offsets): one select on the fragment ADDRESS per column.  B traffic into LDS per K-step: 48 KB -> 17 KB.
// (the 64-row tiles are bound by a roughly constant per-step issue + synchronisation overhead over only 48 MFMAs per wave and by
// their per-tile fixed cost at 18 K-steps per tile, tools/probe_phases.py -- not by LDS traffic or DMA latency, which the 64 x 192
// arm above was built against)
// -DPC_PHASES (tools/probe_phases.py, a timing-only build): wave 0 of every block sums the shader-clock cycles it spends in each
// phase of a segment (lead wait | barrier | prologue reads | main loop | next segment's decode + lead | fix-up | epilogue | gap)
#ifdef PC_PHASES
__device__ unsigned long long pc_prof[256 * 8];
#define PC_T(i) do { const unsigned long long t_ = clock64(); pc_acc[i] += t_ - pc_t; pc_t = t_; } while (0)
#else
#define PC_T(i) do { } while (0)
#endif
#ifndef PC_WAVES_PER_EU
#define PC_WAVES_PER_EU 2
#endif
template <int FR, int FC, int NS, int SP, bool TS = false>
__global__ __launch_bounds__(512, PC_WAVES_PER_EU) void pconv_kernel(const IgemmParams p)
{
#if __HIP_DEVICE_COMPILE__
#ifdef PC_PHASES
    unsigned long long pc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pc_t = clock64();
#endif
    constexpr int WN = 4;
    constexpr int BM = 32 * FR, BN = 64 * FC;
    constexpr int GA = BM / 16, GB = TS ? (BN + 2 + 15) / 16 : BN / 16;       // 16-row groups per operand stage (TS: + the two halo rows)
    constexpr int BROWS = 16 * GB;                             // rows of a B stage plane
    constexpr int SA = 3 * BM * 64, SB = 3 * BROWS * 64;       // bytes per stage
    constexpr int NSB = TS ? 2 : NS;                           // B slots (TS: one per kernel row, two in flight)
    static_assert(!TS || FC == 4, "tap-row sharing: 256-pixel tiles");
    constexpr int JA = (3 * GA + 7) / 8;                       // A DMA instructions per wave per stage (waves past 3 GA - 8 (JA - 1): one fewer)
    constexpr int RGB = (GB + 7) / 8;                          // B row groups per wave (wave + 8 i < GB), each x 3 planes
    static_assert(NS == 2 && FC == 4 && (FR == 2 || FR == 4), "two LDS stages, 256-pixel tiles, 64 or 128 rows");
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)smem;              // LDS byte address of the dynamic segment
    const unsigned As = lds0, Bs = lds0 + NS * SA;             // [NS][3][BM][64], [NSB][3][BROWS][64]

    const int nsteps = p.nsteps;                               // K-steps of 32: (channel block, tap), tap-minor
    const int ntaps = p.ntaps;
    const int HWg = p.Hg * p.Wg;
    const int npix = p.imgs_per_group * HWg;
    const int tiles_pg = p.tilesM * p.tilesN;
    const int Wi = p.Wi;
    const float rcpHW = 1.0f / (float)HWg, rcpW = 1.0f / (float)p.Wg;

    // XCD-aware bijective remap of the block id to a range index
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int rbk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const long long S = p.steps_per_block;
    long long w = (long long)rbk * S;
    const long long wend = min(p.total_steps, w + S);

    // buffer descriptors.  X: the base sits (Wi + 1) pixels BELOW the tensor, so that the tap offsets (dh + 1) * Wi + (dw + 1)
    // are never negative (the launcher keeps the planes below 2 GB: 0x80000000 is out of range for every descriptor)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.Wsp), 0, (unsigned)((size_t)nsteps * 3 * p.M * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.Xp) - (size_t)(Wi + 1) * 64), 0,
        (unsigned)(((size_t)(p.Ci >> 5) * 3 * p.xp_pix + (size_t)(Wi + 1)) * 64), 0x00020000);
    const int drow = lane >> 2, chunk = (lane & 3) ^ ((lane >> 3) & 3);
    // fragment reads: row 16r + li of the wave's rows, chunk lg sits in slot lg ^ ((li>>1)&3)
    const unsigned fro = li * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    const unsigned Af0 = As + wm * (16 * FR) * 64 + fro;
    const unsigned Bf0 = Bs + wn * (16 * FC) * 64 + fro;
    // TS: the fragment row of output pixel (c, li) under column shift dw = d - 1 is stage row 1 + dw + 16 c + li (+ 64 wn)
    unsigned Bfd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) Bfd[d] = Bs + wn * (16 * FC) * 64 + (li + d) * 64 + ((lg ^ (((li + d) >> 1) & 3)) << 4);
    // TS: a lane whose pixel has no neighbour under the tap's column shift (the neighbour lies in another image row) reads its
    // fragment from the stage's zero rows -- ONE select on the address per column where masking the fragment took twelve on
    // the registers.  Zc[c] + the read's own offset (plane, 16 c rows) = row BN + 2 + (li & 7), chunk lg of the plane
    unsigned Zc[FC];
#pragma unroll
    for (int c = 0; c < FC; ++c) Zc[c] = Bs + (unsigned)((BN + 2 + (li & 7) - 16 * c) * 64 + lg * 16);
    // B row groups this wave issues (wave + 8 i < GB)
    const int ngrpB = (wave + 8 * (RGB - 1) < GB) ? RGB : RGB - 1;
    // TS: kernel-row offset and column shifts from the tap table: taps 3 j .. 3 j + 2 share dh; d = dw + 1 of tap t
    auto tap_d = [&](int t) { return (int)((unsigned)(p.tapcode >> (4 * t + 2)) & 3u); };

    // A segment = the K-steps [k0, k1) of one tile that this block's range covers, with the per-lane DMA state of the tile
    struct Seg {
        int tile, k0, k1, grp, tn, m0, n0;
        unsigned voffA, voffB[RGB], vmask[RGB];
        int icc_b, it_b;                         // (channel block, tap) cursor of the next B step to be issued (TS: it_b = kernel row)
        // TS: lane masks (wave-uniform; the four 16-lane rows of a wave hold the same pixels): bit li = the pixel of lane row li in
        // column tile c has a left neighbour in its image row, bit 16 + li = a right one
        unsigned mLR[FC];
    };
    auto decode = [&](Seg& g) {                  // takes the next segment off the block's range [w, wend)
        g.tile = (int)((unsigned)w / (unsigned)nsteps);              // (total_steps < 2^31: the launcher checks)
        g.k0 = (int)(w - (long long)g.tile * nsteps);
        g.k1 = min(nsteps, g.k0 + (int)(wend - w));
        w += g.k1 - g.k0;
        g.grp = g.tile / tiles_pg;
        const int tl = g.tile - g.grp * tiles_pg;
        const int tm = p.tn_fast ? tl / p.tilesN : tl % p.tilesM;
        g.tn = p.tn_fast ? tl % p.tilesN : tl / p.tilesM;
        g.m0 = tm * BM; g.n0 = g.tn * BN;
        // A job j = plane j / GA, row group j % GA; wave takes j = wave + 8 q.  Rows past M never occur (M % BM == 0).
        g.voffA = (unsigned)((g.m0 + drow) * 64 + chunk * 16);
        // B: row group gi = wave + 8 i holds output pixels n0 + 16 gi + drow of the group's virtual grid
#pragma unroll
        for (int i = 0; i < RGB; ++i) {
            // TS: stage row r holds pixel n0 - 1 + r (one halo pixel on either side of the tile)
            const int n = g.n0 + 16 * (wave + 8 * i) + drow - (TS ? 1 : 0);
            // (TS: the rows past the halo, BN + 2 .. 16 GB - 1, are the stage's ZERO rows)
            const bool rv = n >= 0 && n < npix && i < ngrpB && (!TS || 16 * (wave + 8 * i) + drow < BN + 2);
            const int nn = rv ? n : 0;
            int img, rem, hg, wg;
            pc_divmod(nn, HWg, rcpHW, img, rem);
            pc_divmod(rem, p.Wg, rcpW, hg, wg);
            const int ih0 = hg * p.sg, iw0 = wg * p.sg;
            unsigned vm = 0;
            if constexpr (TS) {
                // bit j: the pixel's row under kernel row j (dh of tap 3 j) stays inside the image
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ih = ih0 + (int)((unsigned)(p.tapcode >> (12 * j)) & 3u) - 1;
                    if (rv && (unsigned)ih < (unsigned)p.Hi) vm |= 1u << j;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const unsigned f = (unsigned)(p.tapcode >> (4 * t)) & 15u;
                    const int ih = ih0 + (int)(f & 3u) - 1, iw = iw0 + (int)(f >> 2) - 1;
                    if (t < ntaps && rv && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)Wi) vm |= 1u << t;
                }
            }
            g.vmask[i] = vm;
            g.voffB[i] = (unsigned)((((g.grp * p.imgs_per_group + img) * p.Hi + ih0) * Wi + iw0) * 64 + chunk * 16);
        }
        if constexpr (TS) {
            g.icc_b = g.k0 / 9;
            g.it_b = (g.k0 - g.icc_b * 9) / 3;
#pragma unroll
            for (int c = 0; c < FC; ++c) {
                const int n = g.n0 + wn * (16 * FC) + 16 * c + li;
                int img, rem, hg, wg;
                pc_divmod(n, HWg, rcpHW, img, rem);
                pc_divmod(rem, p.Wg, rcpW, hg, wg);
                const unsigned bl = (unsigned)__builtin_amdgcn_ballot_w64(wg > 0) & 0xffffu;
                const unsigned br = (unsigned)__builtin_amdgcn_ballot_w64(wg < p.Wg - 1) & 0xffffu;
                g.mLR[c] = (unsigned)__builtin_amdgcn_readfirstlane((int)(bl | (br << 16)));
            }
        } else {
            g.icc_b = g.k0 / ntaps;
            g.it_b = g.k0 - g.icc_b * ntaps;
        }
    };
    auto issueA = [&](const Seg& g, int s, int slot) {
#pragma unroll
        for (int q = 0; q < JA; ++q) {
            const int j = wave + 8 * q;
            if ((3 * GA) % 8 != 0 && j >= 3 * GA) break;
            const int plane = j / GA, gr = j % GA;
            const unsigned so = (unsigned)(((s * 3 + plane) * p.M + 16 * gr) * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(size_t)(As + slot * SA + j * 1024), 16, g.voffA, so, 0, 0);
        }
    };
    // B steps are issued strictly in order k0, k0 + 1, ...: the (channel block, tap) cursor advances with them
    auto issueB = [&](Seg& g, int slot) {
        if constexpr (TS) {
            // one kernel row (channel block icc_b, row it_b) of the tile: pixels n0 - 1 .. n0 + BN shifted by dh rows
            const unsigned dh1 = (unsigned)(p.tapcode >> (12 * g.it_b)) & 3u;          // dh + 1
            const unsigned tapo = (dh1 * (unsigned)Wi + 1u) * 64u;                      // (dh + 1) * Wi + (0 + 1) pixels
#pragma unroll
            for (int i = 0; i < RGB; ++i) {
                if (GB % 8 != 0 && i >= ngrpB) break;
                const unsigned vo = ((g.vmask[i] >> g.it_b) & 1u) ? g.voffB[i] : OOB;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned so = (unsigned)((size_t)(g.icc_b * 3 + pl) * p.xp_pix * 64) + tapo;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(size_t)(Bs + slot * SB + (pl * GB + wave + 8 * i) * 1024), 16,
                                                             vo, so, 0, 0);
                }
            }
            if (++g.it_b == 3) { g.it_b = 0; ++g.icc_b; }
            return;
        }
        const unsigned f = (unsigned)(p.tapcode >> (4 * g.it_b)) & 15u;
        const unsigned tapo = ((f & 3u) * (unsigned)Wi + (f >> 2)) * 64u;           // (dh + 1) * Wi + (dw + 1) pixels
#pragma unroll
        for (int i = 0; i < RGB; ++i) {
            if (GB % 8 != 0 && i >= ngrpB) break;
            const unsigned vo = ((g.vmask[i] >> g.it_b) & 1u) ? g.voffB[i] : OOB;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned so = (unsigned)((size_t)(g.icc_b * 3 + pl) * p.xp_pix * 64) + tapo;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(size_t)(Bs + slot * SB + (pl * GB + wave + 8 * i) * 1024), 16,
                                                         vo, so, 0, 0);
            }
        }
        if (++g.it_b == ntaps) { g.it_b = 0; ++g.icc_b; }
    };
    // the DMA a segment starts with: steps k0 (both operands) and k0 + 1 (A).  Issued for the NEXT segment before the current
    // one's fix-up / epilogue runs: a tile's pipeline fill hides behind its predecessor's stores (tiles of 2 - 18 K-steps --
    // the 64-channel layers, the 1x1 convs -- spent a third of their time in it)
    auto lead = [&](Seg& g) {
        issueA(g, g.k0, 0);
        issueB(g, 0);
        if (g.k0 + 1 < g.k1) issueA(g, g.k0 + 1, 1);
    };

    if (w >= wend) return;                       // (a block whose range is empty: the grid was rounded up)
    if ((p.pc_flags & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if ((p.pc_flags & 4) && wave < 4) __builtin_amdgcn_s_setprio(1);
    Seg cur;
    decode(cur);
    lead(cur);
    int pend_tile = -1;                          // a stored part of a shared tile that has not been announced yet
    for (;;) {
        const int k0 = cur.k0, k1 = cur.k1;
        f32x4 acc[FR][FC];
#pragma unroll
        for (int r = 0; r < FR; ++r)
#pragma unroll
            for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Round 6: the step's MFMAs run ROW by row -- all four B columns of the step sit in registers (48), the
        // A rows roll through two buffers (24) -- instead of column by column with both steps' A fragments resident (96 + 24).  Same
        // reads, MFMAs, stages and one barrier per step; 48 registers fewer, which is what lets a wave of a streaming kernel of the
        // other stream (BatchNorm apply: 32 registers) live beside the two GEMM waves of a SIMD (512 registers per lane and SIMD)
        // (the column-major step of rounds 4-5 -- both steps' A fragments resident, B rolling through two buffers -- is in the history
        // of this file and in profiles/HISTORY.md)
        sp_u32x4 A0[1][3], A1[1][3], Bb[FC][3];
    // BASE = the fragment base address of the stage slot the read takes (Af0 + slot * SA, Bf0 + slot * SB)
#define PC_READA(BASE, R, DST)                                                                  \
    {                                                                                           \
        DST[0] = pc_lds_read128<(0 * BM + 16 * (R)) * 64>(BASE);                                \
        DST[1] = pc_lds_read128<(1 * BM + 16 * (R)) * 64>(BASE);                                \
        DST[2] = pc_lds_read128<(2 * BM + 16 * (R)) * 64>(BASE);                                \
    }
#define PC_READB(BASE, C, DST)                                                                  \
    {                                                                                           \
        const unsigned base_ = (BASE);                                                          \
        DST[0] = pc_lds_read128<(0 * BROWS + 16 * (C)) * 64>(base_);                            \
        DST[1] = pc_lds_read128<(1 * BROWS + 16 * (C)) * 64>(base_);                            \
        DST[2] = pc_lds_read128<(2 * BROWS + 16 * (C)) * 64>(base_);                            \
    }
#define PC_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    // the wait at the END of column C's MFMAs.  The accumulators are operands of the statement: without that dependence the machine
    // scheduler is free to move the bare wait anywhere between the two scheduling barriers, and it put it behind the column's FIRST
    // MFMA (rounds 4-5 shipped that: every column waited for its reads before 23 of its 24 MFMAs)
#define PC_LGKM0_COL(C)                                                                                                           \
    do {                                                                                                                          \
        if constexpr (FR == 4)                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][C]), "+v"(acc[1][C]), "+v"(acc[2][C]), "+v"(acc[3][C])::"memory"); \
        else                                                                                                                      \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][C]), "+v"(acc[1][C])::"memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    } while (0)
    // workgroup barrier that orders LDS accesses only: a __syncthreads() would also wait for the LDS-DMA in flight
#define PC_SYNC_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

        // ---- prologue.  Stage slot of step s = (s - k0) % NS.  Invariant at the start of step s: B(s), A(s + 1) have landed
        // (A(s) is in registers), B(s + 1 .. s + NS - 1) and A(s + 2 .. s + NS) are issued ----------------------------------------
        PC_T(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the segment's lead DMA: A(k0), B(k0), A(k0 + 1)
        PC_T(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (pend_tile >= 0) {
            // the previous segment's part of a shared tile has reached memory (every wave waited above): announce it
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // keep: the fence's own wait may be dropped
                if (!(p.pc_flags & 8)) __hip_atomic_fetch_add(p.counters + pend_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pend_tile = -1;
        }
        PC_T(1);
        // TS: kernel rows (super-steps of three taps) U = s / 3; this segment covers U0 .. U1; B slot of U = (U - U0) & 1
        int it_cur = 0, Ucur = 0, ub = 0;
        int b_age = 3;                         // TS: barriers since the last B row was issued (it goes out AFTER that barrier's A stage)
        // The step's barrier needs A(s + 1) and B(s + 1) landed.  Vector-memory operations complete in issue order, and a TS row goes
        // out AFTER its barrier's A stage: a row issued one barrier ago may stay in flight.  Every wave waits with the count of the
        // waves that hold the SMALLER share of a row (the others wait for three of their pieces more): ONE branch -- the count as a
        // run-time switch over vmcnt immediates had compiled into a chain of ~25 scalar branches per step
        auto wait_dma = [&](bool full) {
            if constexpr (TS) {
                constexpr int NROW = 3 * (GB % 8 ? RGB - 1 : RGB);
                if (full && b_age == 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NROW) : "memory"); return; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const int U1 = (k1 - 1) / 3;
        // c ? b : a for a wave-uniform c, as ONE v_cndmask (the compiler's own form of a uniform select between vector
        // registers is a branch: three per fragment column in this loop)
        auto usel = [&](unsigned a, unsigned b, bool c) -> unsigned {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(c);      // (every lane is active: all ones or zero, a scalar pair)
            unsigned r;
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
            return r;
        };
        // the shifted fragment rows under column shift d - 1, slot byte offset so
        auto bsel = [&](int d, unsigned so) { return usel(usel(Bfd[2], Bfd[1], d == 1), Bfd[0], d == 0) + so; };
        // fragment base of column tile c: the shifted rows (real), or the zero rows where the lane's pixel has no neighbour
        auto colbase = [&](int c, int d, unsigned so, unsigned real) -> unsigned {
            const unsigned h = d == 0 ? (cur.mLR[c] & 0xffffu) : (cur.mLR[c] >> 16);
            const unsigned w32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(d == 1 ? ~0u : (h | (h << 16))));
            const unsigned long long m = (unsigned long long)w32 | ((unsigned long long)w32 << 32);
            unsigned r;
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(Zc[c] + so), "v"(real), "s"(m));
            return r;
        };
        if constexpr (TS) {
            it_cur = k0 % 9; Ucur = k0 / 3;
            if (Ucur + 1 <= U1) issueB(cur, 1);
        } else {
            if (k0 + 1 < k1) issueB(cur, 1);
        }
        {
            // row 0 of A(k0) and all four columns of B(k0); the other rows of A(k0) are read while the step runs, so its stage is
            // not free yet: A(k0 + 2) goes out behind step k0's barrier like every other A stage
            PC_READA(Af0, 0, A0[0]);
            pc_for4([&](auto ic) {
                constexpr int c = decltype(ic)::value;
                if constexpr (TS) { PC_READB(colbase(c, tap_d(it_cur), 0u, bsel(tap_d(it_cur), 0u)), c, Bb[c]); }
                else PC_READB(Bf0, c, Bb[c]);
            });
            PC_LGKM0();
        }

        PC_T(2);
        int ib = 0;                            // stage slot of the current step
        // the two waves of a SIMD (w, w + 4) run this loop in lockstep between barriers: when both issue a step's DMA at the same
        // point the matrix pipe idles behind them.  `late`: waves 4-7 issue theirs BEHIND the last column's MFMAs
        const bool late = (p.pc_flags & 1) && wave >= 4;
        {
            // ---- one K-step, row-major.  At its start: row 0 of A(s) in A0[0], B(s) columns 0 .. 3 in Bb (columns 1 .. 3 possibly
            // still in flight: counted waits in pass 0).  Pass r = row r's 24 MFMAs with row r + 1 read behind them.  The barrier
            // sits before the LAST pass (3; 1 for the 64-row tiles): every wave holds all of A(s) and B(s), and A(s + 1), B(s + 1) have landed -- the stage of A(s) takes
            // A(s + 2); pass 3 re-fills the registers column by column with B(s + 1) behind each column's last use ---------------
            auto rstep = [&](auto full_c, int s) {
                constexpr bool FULL = decltype(full_c)::value;
                const int ib1 = ib ^ 1;
                const unsigned a_cur = Af0 + ib * SA, a_nxt = Af0 + ib1 * SA;
                unsigned b_nxt = 0, so_nxt = 0;
                int d_nxt = 1;
                bool row_end = false;
                if constexpr (TS) {
                    const int it1 = it_cur == 8 ? 0 : it_cur + 1;
                    d_nxt = tap_d(it1);
                    row_end = it_cur == 2 || it_cur == 5 || it_cur == 8;
                    so_nxt = (unsigned)((row_end ? ub ^ 1 : ub) * SB);
                    b_nxt = bsel(d_nxt, so_nxt);
                } else
                    b_nxt = Bf0 + ib1 * SB;
                auto bn = [&](int c) { if constexpr (TS) return colbase(c, d_nxt, so_nxt, b_nxt); else return b_nxt; };
                const bool more = FULL || s + 1 < k1;
    // the wait in FRONT of column C of pass 0: columns C .. 3 of this step's B (read behind the previous step's pass 3) and
    // row 1 of A (read at this pass's head) may still be in flight: LDS returns in order, so all but the youngest 3 (4 - C) reads
    // are waited for.  The column's registers are operands: the MFMAs that read them stay behind the wait; so is the previous
    // column's accumulator: the wait stays behind that column's MFMAs (three bare waits moved to the head of the pass together)
#define PC_WAIT_BCOL(C)                                                                                                         \
                asm volatile("s_waitcnt lgkmcnt(%4)"                                                                           \
                             : "+v"(Bb[C][0]), "+v"(Bb[C][1]), "+v"(Bb[C][2]), "+v"(acc[0][(C) - 1])                            \
                             : "n"(3 * (4 - (C)))                                                                               \
                             : "memory")
    // the wait at the END of pass R (row R + 1 has arrived), tied to the pass's accumulators (PC_LGKM0_COL)
#define PC_WAIT_ROW(R)                                                                                                          \
                do {                                                                                                            \
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[R][0]), "+v"(acc[R][1]), "+v"(acc[R][2]), "+v"(acc[R][3])::"memory"); \
                    __builtin_amdgcn_sched_barrier(0);                                                                          \
                } while (0)
                // pass 0
                PC_READA(a_cur, 1, A1[0]);
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = pc_mfma<SP>(A0[0], Bb[0], acc[0][0]);
                PC_WAIT_BCOL(1);
                acc[0][1] = pc_mfma<SP>(A0[0], Bb[1], acc[0][1]);
                PC_WAIT_BCOL(2);
                acc[0][2] = pc_mfma<SP>(A0[0], Bb[2], acc[0][2]);
                PC_WAIT_BCOL(3);
                acc[0][3] = pc_mfma<SP>(A0[0], Bb[3], acc[0][3]);
                PC_WAIT_ROW(0);
                if constexpr (FR == 4) {
                    // pass 1
                    PC_READA(a_cur, 2, A0[0]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < FC; ++c) acc[1][c] = pc_mfma<SP>(A1[0], Bb[c], acc[1][c]);
                    PC_WAIT_ROW(1);
                    // pass 2
                    PC_READA(a_cur, 3, A1[0]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < FC; ++c) acc[2][c] = pc_mfma<SP>(A0[0], Bb[c], acc[2][c]);
                    PC_WAIT_ROW(2);
                }
                // every wave holds A(s) and B(s): the barrier, then the DMA into the stages they came from
                wait_dma(FULL);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                auto issue_dma = [&]() {
                    if (FULL || s + 2 < k1) issueA(cur, s + 2, ib);
                    if constexpr (TS) {
                        if (row_end && Ucur + 2 <= U1) { issueB(cur, ub); b_age = 0; }
                    } else {
                        if (FULL || s + 2 < k1) issueB(cur, ib);
                    }
                };
                if (!late) issue_dma();
                // the last pass: row 0 of A(s + 1) at its head; column c of B(s + 1) behind column c's MFMAs
                if (more) PC_READA(a_nxt, 0, A0[0]);
                __builtin_amdgcn_sched_barrier(0);
                pc_for4([&](auto ic) {
                    constexpr int c = decltype(ic)::value;
                    acc[FR - 1][c] = pc_mfma<SP>(A1[0], Bb[c], acc[FR - 1][c]);
                    // (the accumulator is an operand: the read of the next B(c) stays behind the MFMAs that read this one)
                    asm volatile("" : "+v"(acc[FR - 1][c]), "+v"(Bb[c][0]), "+v"(Bb[c][1]), "+v"(Bb[c][2]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) PC_READB(bn(c), c, Bb[c]);
                });
                __builtin_amdgcn_sched_barrier(0);
                if (late) issue_dma();
                // row 0 of A(s + 1) and column 0 of B(s + 1) are the oldest six of the fifteen reads in flight
                asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(A0[0][0]), "+v"(A0[0][1]), "+v"(A0[0][2]), "+v"(Bb[0][0]), "+v"(Bb[0][1]),
                             "+v"(Bb[0][2])::"memory");
                __builtin_amdgcn_sched_barrier(0);
#undef PC_WAIT_BCOL
#undef PC_WAIT_ROW
                if constexpr (TS) {
                    if (row_end) { ub ^= 1; ++Ucur; }
                    it_cur = it_cur == 8 ? 0 : it_cur + 1;
                    if (b_age < 3) ++b_age;
                }
                ib = ib1;
            };
            int s = k0;
            for (; s + 3 <= k1; ++s) rstep(std::true_type{}, s);       // FULL: steps s + 1 and s + 2 exist
            for (; s < k1; ++s) rstep(std::false_type{}, s);
        }
#undef PC_READA
#undef PC_READB
#undef PC_LGKM0
#undef PC_LGKM0_COL
        PC_SYNC_LDS();         // every wave is done reading both stages: they can take the next segment's lead DMA
        PC_T(3);

        // ---- the next segment's pipeline fill starts now, behind this segment's fix-up / epilogue --------------------------------
        const bool more = w < wend;
        Seg nx;
        if (more) {
            decode(nx);
            lead(nx);
        }
        PC_T(4);
        const int tile = cur.tile, grp = cur.grp, tn = cur.tn, m0 = cur.m0, n0 = cur.n0;
        // the fix-up / epilogue scratch sits BEHIND the stages (they are being refilled)
        float* fsmem = reinterpret_cast<float*>(smem + NS * SA + NSB * SB);
        bool do_epilogue = true;

        // ---- stream-K fix-up: partial tiles meet in the slab ------------------------
        // A block's range starts inside a tile at most once (its FIRST segment: the tile's tail, run at the START of the range)
        // and ends inside one at most once (its LAST segment: the tile's head, run at the END).  The head's owner (b_first) is the
        // tile's finisher: by the time it gets there -- a whole range later -- the owners of the other parts have long stored theirs,
        // so it waits on the tile's counter (normally not at all), keeps its own part in registers and sums the parts in segment
        // order (the result does not depend on timing).  The others store their part and announce it AFTER the next segment's
        // pipeline fill has been waited for (the drain of the stores hides behind that wait).
        if (k0 != 0 || k1 != nsteps) {
            const unsigned t0 = (unsigned)tile * (unsigned)nsteps;
            const int b_first = (int)(t0 / (unsigned)S), b_last = (int)((t0 + (unsigned)nsteps - 1u) / (unsigned)S);
            if (rbk != b_first) {
                float* mine = p.slab + (size_t)rbk * 2 * (BM * BN);
#pragma unroll
                for (int r = 0; r < FR; ++r)
#pragma unroll
                    for (int c = 0; c < FC; ++c)
                        *reinterpret_cast<f32x4*>(mine + ((r * FC + c) * 512 + tid) * 4) = acc[r][c];
                pend_tile = tile;
                do_epilogue = false;
            } else {
                int* arrived = reinterpret_cast<int*>(fsmem);
                if (tid == 0) {
                    const int need = b_last - b_first;
                    // bounded (a few seconds: the parts were stored a whole range ago; only a block that was never scheduled can be
                    // missing): a lost part must not hang the GPU -- it POISONS the tile instead (NaN: the step's loss says so)
                    int got = 0;
                    const int spins = (p.pc_flags & 8) ? (1 << 8) : (1 << 22);        // (bit 3: the test hook's short wait)
                    for (int it = 0; it < spins && !got; ++it) {
                        got = __hip_atomic_load(p.counters + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == need;
                        if (!got) __builtin_amdgcn_s_sleep(4);
                    }
                    if (!got) {
                        // the step must not reach the weights: the optimizer kernel reads p.err, the host reads its mapped twin
                        if (p.err) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    *arrived = got;
                }
                __syncthreads();
                const float first = *arrived ? 0.f : __builtin_nanf("");
                PC_SYNC_LDS();                  // (the word is scratch of the epilogue below)
#pragma unroll
                for (int r = 0; r < FR; ++r)
#pragma unroll
                    for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{first, first, first, first} + acc[r][c];
                for (int bb = b_first + 1; bb <= b_last; ++bb) {
                    const float* src = p.slab + (size_t)bb * 2 * (BM * BN);
                    // eight loads in flight at a time (two row tiles x four columns), then their sums: load by load, each was waited
                    // for alone
#pragma unroll
                    for (int r = 0; r < FR; r += 2) {
                        f32x4 part[2][FC];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int c = 0; c < FC; ++c)
                                part[q][c] = *reinterpret_cast<const f32x4*>(src + (((r + q) * FC + c) * 512 + tid) * 4);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int c = 0; c < FC; ++c) acc[r + q][c] += part[q][c];
                    }
                }
            }
        }

        PC_T(5);
        if (do_epilogue) {
        // ---- epilogue -------------------------------------------------------------
        // acc[r][c][q] = D[m = m0 + wm*16FR + 16r + 4*lg + q][n = n0 + wn*64 + 16c + li]
        const int mbase = m0 + wm * (16 * FR) + 4 * lg;
        if (p.stats) {
            float* red = fsmem;            // [WN][BM][2]
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
                    // sum over the 16 pixels of the row of lanes; lane li = 0 receives, level by level, the partners the xor
                    // butterfly 1, 2, 4, 8 would hand it: quad_perm [1,0,3,2], [2,3,0,1], row_ror:12 (lane i <- i + 4), row_ror:8
                    float u = s1[q], v = s2[q];
                    u += pc_dpp<0xB1>(u); v += pc_dpp<0xB1>(v);
                    u += pc_dpp<0x4E>(u); v += pc_dpp<0x4E>(v);
                    u += pc_dpp<0x12C>(u); v += pc_dpp<0x12C>(v);
                    u += pc_dpp<0x128>(u); v += pc_dpp<0x128>(v);
                    if (li == 0) {
                        const int ml = wm * (16 * FR) + 16 * r + 4 * lg + q;
                        red[(wn * BM + ml) * 2 + 0] = u;
                        red[(wn * BM + ml) * 2 + 1] = v;
                    }
                }
            }
            PC_SYNC_LDS();
            for (int ch = tid; ch < BM; ch += 512) {
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int ww = 0; ww < WN; ++ww) {
                    u += red[(ww * BM + ch) * 2 + 0];
                    v += red[(ww * BM + ch) * 2 + 1];
                }
                float* st = p.stats + (size_t)(grp * p.tilesN + tn) * 2 * p.M;
                st[m0 + ch] = u;
                st[p.M + m0 + ch] = v;
            }
        }
        // Round 6: the output loop runs PAIR of row tiles by pair (a pair = one 32-channel block: one plane chunk), columns inside: the
        // folded BatchNorm of a pair is loaded once, and a column's residual loads are issued together.  (Column by column with one
        // branch per load, every load -- seven per column in the eval / data-gradient forms -- was waited for alone.)  The register
        // count of the kernel is set here as much as in the main loop: nothing is kept across pairs or columns beyond opix.
        size_t opix[FC];
        bool ok[FC];
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const int n = n0 + wn * (16 * FC) + 16 * c + li;
            ok[c] = n < npix;
            int img, rem, hg, wg;
            pc_divmod(ok[c] ? n : npix - 1, HWg, rcpHW, img, rem);          // (a pixel past the end reads the last one's, stores nothing)
            pc_divmod(rem, p.Wg, rcpW, hg, wg);
            opix[c] = (size_t)((grp * p.imgs_per_group + img) * p.Ho + hg * p.os + p.oh0) * p.Wo + (wg * p.os + p.ow0);
        }
#pragma unroll
        for (int r = 0; r < FR; r += 2) {
            __builtin_amdgcn_sched_barrier(0);
            const int m = mbase + 16 * r;
            const size_t cb = (size_t)(m0 + wm * (16 * FR) + 16 * r) >> 5;
            f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
            if (p.scale) {
                sc0 = *reinterpret_cast<const f32x4*>(p.scale + m); sc1 = *reinterpret_cast<const f32x4*>(p.scale + m + 16);
                sh0 = *reinterpret_cast<const f32x4*>(p.shift + m); sh1 = *reinterpret_cast<const f32x4*>(p.shift + m + 16);
            }
#pragma unroll
            for (int c = 0; c < FC; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                const size_t o = opix[c] * p.Co;
                f32x4 v0 = acc[r][c], v1 = acc[r + 1][c];
                if (p.scale) { v0 = v0 * sc0 + sh0; v1 = v1 * sc1 + sh1; }
                if (p.res) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.res + o + m), b = *reinterpret_cast<const f32x4*>(p.res + o + m + 16);
                    v0 += a; v1 += b;
                }
                if (p.resp) {
                    // residual kept only as planes: this lane's two row tiles are chunk lg of block cb; x = (h + m) + l exactly
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.resp) + ((cb * 3) * (size_t)p.yp_pix + opix[c]) * 64 + lg * 16;
                    const sp_u32x4 H = *reinterpret_cast<const sp_u32x4*>(src);
                    const sp_u32x4 M = *reinterpret_cast<const sp_u32x4*>(src + (size_t)p.yp_pix * 64);
                    const sp_u32x4 L = *reinterpret_cast<const sp_u32x4*>(src + (size_t)p.yp_pix * 128);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lo = (__builtin_bit_cast(float, H[i] << 16) + __builtin_bit_cast(float, M[i] << 16)) +
                                         __builtin_bit_cast(float, L[i] << 16);
                        const float hi = (__builtin_bit_cast(float, H[i] & 0xffff0000u) + __builtin_bit_cast(float, M[i] & 0xffff0000u)) +
                                         __builtin_bit_cast(float, L[i] & 0xffff0000u);
                        // words 0, 1 = the first row tile's channels 4 lg .. 4 lg + 3, words 2, 3 = the second's
                        if (i < 2) { v0[2 * i] += lo; v0[2 * i + 1] += hi; }
                        else { v1[2 * (i - 2)] += lo; v1[2 * (i - 2) + 1] += hi; }
                    }
                }
                if (p.relu == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                }
                if (!ok[c]) continue;
                if (p.Y) {
                    *reinterpret_cast<f32x4*>(p.Y + o + m) = v0;
                    *reinterpret_cast<f32x4*>(p.Y + o + m + 16) = v1;
                }
                if (p.Yp) {
                    // planes of the output: the pair is chunk lg of the 32-channel block cb
                    sp_u32x4 H, M, L;
                    split3(v0, v1, H, M, L);
                    unsigned char* dst = reinterpret_cast<unsigned char*>(p.Yp) + ((cb * 3) * (size_t)p.yp_pix + opix[c]) * 64 + lg * 16;
                    *reinterpret_cast<sp_u32x4*>(dst) = H;
                    *reinterpret_cast<sp_u32x4*>(dst + (size_t)p.yp_pix * 64) = M;
                    *reinterpret_cast<sp_u32x4*>(dst + (size_t)p.yp_pix * 128) = L;
                }
            }
        }
        }
        PC_T(6);
        if (!more) break;
        PC_SYNC_LDS();         // the scratch (statistics, flag) is reused by the next segment's epilogue
        cur = nx;
    }
    if (pend_tile >= 0) {                        // (a block whose whole range lies inside one tile)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(p.pc_flags & 8)) __hip_atomic_fetch_add(p.counters + pend_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef PC_SYNC_LDS
#ifdef PC_PHASES
    if (tid == 0)
        for (int i = 0; i < 8; ++i) pc_prof[blockIdx.x * 8 + i] = pc_acc[i];
#endif
#endif
}
#ifdef PC_PHASES
extern "C" int fm_debug_pconv_prof(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_prof), sizeof(unsigned long long) * 256 * 8);
}
#endif

// ---- weight planes, block-major: dst[K block s = (channel block, tap)][3][M][32] from fp32 W[M][ntaps][Ci] ------------------
// One thread = one 8-value chunk (channels 4g..4g+3, 16+4g..16+4g+3 of a 32-channel block): two 16-B reads 64 B apart, three
// 16-B writes (4 threads = 64 B of one row of one plane).  Thread order: (s, m, g) with g fastest, then m: writes are contiguous.
__global__ __launch_bounds__(256) void split_weights_bm_kernel(const float* __restrict__ src_base, unsigned short* __restrict__ dst_base,
                                                               const SplitJobBM* __restrict__ jobs, int njobs)
{
#if __HIP_DEVICE_COMPILE__
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
    const SplitJobBM jb = jobs[j];
    const long long idx = (long long)((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x;
    const int nsteps = jb.ntaps * jb.cib;
    if (idx >= (long long)nsteps * jb.M * 4) return;
    const int g = (int)(idx & 3);
    const int m = (int)((idx >> 2) % jb.M);
    const int s = (int)((idx >> 2) / jb.M);
    const int icc = s / jb.ntaps, it = s - icc * jb.ntaps;
    const float* src = (jb.src ? jb.src : src_base + jb.src_off) + ((size_t)m * jb.ntaps + it) * (jb.cib * 32) + icc * 32;
    sp_u32x4 H, M, L;
    split3(ld4(src + 4 * g), ld4(src + 16 + 4 * g), H, M, L);
    unsigned char* dst = reinterpret_cast<unsigned char*>(dst_base + jb.dst_off) + (((size_t)s * 3) * jb.M + m) * 64 + g * 16;
    *reinterpret_cast<sp_u32x4*>(dst) = H;
    *reinterpret_cast<sp_u32x4*>(dst + (size_t)jb.M * 64) = M;
    *reinterpret_cast<sp_u32x4*>(dst + (size_t)jb.M * 128) = L;
#endif
}
int split_job_bm_blocks(int M, int nsteps) { return (int)(((long long)M * nsteps * 4 + 255) / 256); }
void k_split_weights_bm(const float* src_base, unsigned short* dst_base, const SplitJobBM* jobs, int njobs, int nblocks, hipStream_t s)
{
    if (njobs > 0) hipLaunchKernelGGL(split_weights_bm_kernel, dim3(nblocks), dim3(256), 0, s, src_base, dst_base, jobs, njobs);
}

// ---- activation planes of an fp32 NHWC tensor (test hooks, the stem's pooled output when no producer made them) ----------
// dst[C/32][3][npix][32]; one thread = one 8-value chunk of one pixel
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ dst, long long npix,
                                                           int C)
{
#if __HIP_DEVICE_COMPILE__
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cpp = C >> 3;                    // chunks per pixel
    if (idx >= npix * cpp) return;
    const long long pix = idx / cpp;
    const int q = (int)(idx - pix * cpp), cb = q >> 2, g = q & 3;
    const float* src = x + pix * C + cb * 32;
    sp_u32x4 H, M, L;
    split3(ld4(src + 4 * g), ld4(src + 16 + 4 * g), H, M, L);
    unsigned char* d = reinterpret_cast<unsigned char*>(dst) + (((size_t)cb * 3) * npix + pix) * 64 + g * 16;
    *reinterpret_cast<sp_u32x4*>(d) = H;
    *reinterpret_cast<sp_u32x4*>(d + (size_t)npix * 64) = M;
    *reinterpret_cast<sp_u32x4*>(d + (size_t)npix * 128) = L;
#endif
}
void k_split_planes(const float* x, unsigned short* dst, long long npix, int C, hipStream_t s)
{
    const long long n = npix * (C >> 3);
    if (n > 0) hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, dst, npix, C);
}

int pconv_max_blocks() { return 256; }     // ONE block per CU (120 - 144 KB of LDS)
static int g_lose_part = 0;
void pconv_debug_lose_part(int on) { g_lose_part = on; }
// floats of the stream-K slab: one slot of two tiles' worth per block (a range starts inside a tile at most once)
size_t pconv_slab_floats() { return (size_t)pconv_max_blocks() * 2 * 128 * 256; }
int pconv_tile_m(int M) { return M >= 128 ? 128 : 64; }
int pconv_tile_n(int M) { (void)M; return 256; }
// does the planes kernel take this GEMM?  whole M tiles, whole 32-channel blocks, taps within [-1, 1], planes below 2 GB
bool pconv_takes(int M, int Ci, long long xp_pix, int Wi)
{
    if (M % 64 != 0 || (M > 64 && M % 128 != 0) || Ci % 32 != 0) return false;
    // (pixel indices below 2^22: the kernel divides them through a float reciprocal; the parity grids of a stride-2 data gradient
    // hold up to four times the pixels of its input)
    return ((long long)(Ci >> 5) * 3 * xp_pix + Wi + 1) * 64 < 0x7ff00000LL && xp_pix < (1LL << 22);
}

// tap-row sharing applies to: 3x3, stride 1, the taps in three groups of one kernel row each whose column shifts cover -1, 0, +1
// (forward convs and their data gradients alike); FM_PCONV_TS=0 (tuning builds) keeps the per-tap stages
bool pconv_uses_ts(const IgemmParams& p)
{
    static const int ts_on = fm_tune("FM_PCONV_TS", 1);
    bool ts = ts_on && p.ntaps == 9 && p.sg == 1 && p.Hg == p.Hi && p.Wg == p.Wi;
    for (int j = 0; ts && j < 3; ++j) {
        ts = p.dh[3 * j] == p.dh[3 * j + 1] && p.dh[3 * j] == p.dh[3 * j + 2];
        int seen = 0;
        for (int k = 0; k < 3; ++k) seen |= 1 << (p.dw[3 * j + k] + 1);
        ts = ts && seen == 7;
    }
    return ts;
}

void launch_pconv(IgemmParams p, int groups, hipStream_t s)
{
    static bool attr_done = false;
    // two stages + the fix-up / epilogue scratch behind them (statistics partials [4][BM][2] floats, the last-arriver flag)
    constexpr int LDS_L = 2 * 3 * (128 + 256) * 64 + 4 * 128 * 2 * 4;      // 148 KB: 128 x 256, two stages
    constexpr int LDS_S = 2 * 3 * (64 + 256) * 64 + 4 * 128 * 2 * 4;       // 124 KB: 64 x 256, two stages
    // tap-row sharing: two A stages + two B row stages of 272 rows + the scratch
    constexpr int LDS_LT = 2 * 3 * 128 * 64 + 2 * 3 * 272 * 64 + 4 * 128 * 2 * 4;      // 154 KB
    // (three A stages for the 64-row tiles -- 142 KB, two steps of A lookahead, counted waits -- were measured: 377-383 vs 368-378 us on
    // the 56 x 56 x 64 layers, not kept: those tiles are not bound by DMA latency)
    constexpr int LDS_ST = 2 * 3 * 64 * 64 + 2 * 3 * 272 * 64 + 4 * 128 * 2 * 4;       // 130 KB
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<4, 4, 2, 6, true>), LDS_LT, "pconv_kernel<4, 4, 2, 6, true>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<4, 4, 2, 9, true>), LDS_LT, "pconv_kernel<4, 4, 2, 9, true>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<2, 4, 2, 6, true>), LDS_ST, "pconv_kernel<2, 4, 2, 6, true>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<2, 4, 2, 9, true>), LDS_ST, "pconv_kernel<2, 4, 2, 9, true>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<4, 4, 2, 6>), LDS_L, "pconv_kernel<4, 4, 2, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<4, 4, 2, 9>), LDS_L, "pconv_kernel<4, 4, 2, 9>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<2, 4, 2, 6>), LDS_S, "pconv_kernel<2, 4, 2, 6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&pconv_kernel<2, 4, 2, 9>), LDS_S, "pconv_kernel<2, 4, 2, 9>");
        attr_done = true;
    }
    p.nsteps = p.ntaps * (p.Ci >> 5);
    const long long T = (long long)p.tilesM * p.tilesN * groups;
    p.total_steps = T * p.nsteps;
    p.tapcode = 0;
    for (int t = 0; t < p.ntaps; ++t)      // taps of 3x3 / 1x1 convs and of their dgrad parity classes lie in [-1, 1]
        p.tapcode |= (unsigned long long)(((p.dh[t] + 1) & 3) | (((p.dw[t] + 1) & 3) << 2)) << (4 * t);
    // weights of one M-tile: BM rows x K x 6 B; beyond ~1 MB per M-tile the all-M-tiles working set no longer fits L2
    static const int tn_fast = fm_tune("FM_TN_FAST", 2);
    p.tn_fast = tn_fast == 1 ? 1 : (tn_fast == 2 ? (p.tilesM > 1 && (long long)p.M * p.nsteps * 192 > (3LL << 20)) : 0);
    // persistent grid: every CU whenever there are >= 4 K-steps for each of them, otherwise one tile per block.
    // FM_IGEMM_BLOCKS overrides the grid (tests force odd splits so that every fix-up path runs on small shapes).
    static const int forced = getenv("FM_IGEMM_BLOCKS") ? atoi(getenv("FM_IGEMM_BLOCKS")) : 0;
    int nblk = p.total_steps >= 4LL * pconv_max_blocks() ? pconv_max_blocks() : (int)std::min<long long>(pconv_max_blocks(), T);
    if (forced > 0) nblk = (int)std::min<long long>(std::min(forced, pconv_max_blocks()), p.total_steps);
    p.steps_per_block = (int)((p.total_steps + nblk - 1) / nblk);
    // Quantized ranges (round 6; PMC per layer: tools/pmc_convs.sh).  A block's K phase is (range start) mod nsteps; 256 ranges of
    // 111 steps give the 32 blocks of an XCD 32 different phases, and the 512-channel layers -- a 3.5 MB weight slice per M-tile, all
    // of it live at once beside the streaming activation rows in a 4 MB L2 -- fetched the slice about once per TILE: 662 MB per
    // launch for 52 MB of operands.  With the range length a multiple of nsteps / P only P phases exist, and the blocks that share
    // one read the same weight rows at the same time: 112 = 7 x 16 steps, nine phases, 420 MB.  Taken when it lengthens the ranges
    // by <= 2 %; only for slices that do not fit beside the rest (the 256-channel layers' 1.8 MB: 278 -> 326 MB, not taken).
    // Measured and not kept: the M-tiles of a pixel tile as sibling blocks on one XCD (every slice live on every XCD: 770 MB);
    // the non-temporal hint on the activation rows (their kernel-row re-reads miss: step +1.1 ms).
    static const int quant_on = fm_tune("FM_PCONV_QUANT", 1);
    if (quant_on && forced <= 0 && p.steps_per_block > p.nsteps / 2 && (long long)p.nsteps * 128 * 192 > (2LL << 20)) {
        for (int P = 1; P <= 12; ++P) {
            if (p.nsteps % P) continue;
            const int q = p.nsteps / P;
            const int sq = (p.steps_per_block + q - 1) / q * q;
            if ((long long)sq * 100 <= (long long)p.steps_per_block * 102) { p.steps_per_block = sq; break; }
        }
    }
    dim3 grid(nblk);
    // tap-row sharing: 3x3, stride 1, the taps in three groups of one kernel row each whose column shifts cover -1, 0, +1
    // (forward convs and their data gradients alike); FM_PCONV_TS=0 (tuning builds) keeps the per-tap stages
    const bool ts = pconv_uses_ts(p);
    static const int pc_flags = fm_tune("FM_PCONV_FLAGS", 1);      // (measured: 1 = -0.6 % of a step, 2 / 4 = +0.3 %)
    p.pc_flags = (pc_flags & 7) | (g_lose_part ? 8 : 0);
    if (ts) {
        if (p.M >= 128) {
            if (p.sp == 9) hipLaunchKernelGGL((pconv_kernel<4, 4, 2, 9, true>), grid, dim3(512), LDS_LT, s, p);
            else hipLaunchKernelGGL((pconv_kernel<4, 4, 2, 6, true>), grid, dim3(512), LDS_LT, s, p);
        } else {
            if (p.sp == 9) hipLaunchKernelGGL((pconv_kernel<2, 4, 2, 9, true>), grid, dim3(512), LDS_ST, s, p);
            else hipLaunchKernelGGL((pconv_kernel<2, 4, 2, 6, true>), grid, dim3(512), LDS_ST, s, p);
        }
        return;
    }
    if (p.M >= 128) {
        if (p.sp == 9) hipLaunchKernelGGL((pconv_kernel<4, 4, 2, 9>), grid, dim3(512), LDS_L, s, p);
        else hipLaunchKernelGGL((pconv_kernel<4, 4, 2, 6>), grid, dim3(512), LDS_L, s, p);
    } else {
        if (p.sp == 9) hipLaunchKernelGGL((pconv_kernel<2, 4, 2, 9>), grid, dim3(512), LDS_S, s, p);
        else hipLaunchKernelGGL((pconv_kernel<2, 4, 2, 6>), grid, dim3(512), LDS_S, s, p);
    }
}

// fp32 values back from their planes: x = (h + m) + l, both additions exact (test hooks read activations that are only kept as planes)
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ x, long long npix, int C)
{
#if __HIP_DEVICE_COMPILE__
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cpp = C >> 3;
    if (idx >= npix * cpp) return;
    const long long pix = idx / cpp;
    const int q = (int)(idx - pix * cpp), cb = q >> 2, g = q & 3;
    const unsigned char* s = reinterpret_cast<const unsigned char*>(src) + (((size_t)cb * 3) * npix + pix) * 64 + g * 16;
    const sp_u32x4 H = *reinterpret_cast<const sp_u32x4*>(s);
    const sp_u32x4 M = *reinterpret_cast<const sp_u32x4*>(s + (size_t)npix * 64);
    const sp_u32x4 L = *reinterpret_cast<const sp_u32x4*>(s + (size_t)npix * 128);
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = (__builtin_bit_cast(float, H[i] << 16) + __builtin_bit_cast(float, M[i] << 16)) + __builtin_bit_cast(float, L[i] << 16);
        v[2 * i + 1] = (__builtin_bit_cast(float, H[i] & 0xffff0000u) + __builtin_bit_cast(float, M[i] & 0xffff0000u)) +
                       __builtin_bit_cast(float, L[i] & 0xffff0000u);
    }
    float* d = x + pix * C + cb * 32;
    *reinterpret_cast<f32x4*>(d + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(d + 16 + 4 * g) = f32x4{v[4], v[5], v[6], v[7]};
#endif
}
void k_planes_to_f32(const unsigned short* src, float* x, long long npix, int C, hipStream_t s)
{
    const long long n = npix * (C >> 3);
    if (n > 0) hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, x, npix, C);
}
