// Convolution weight gradient as a split-K fp32-MFMA GEMM, gfx950.
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

// FC = 16-column MFMA tiles per wave: 4 (64x64 wave tile) or 3 (64x48: BN = 192 tiles N = 576 = 9 taps x 64
// channels of ResNet layer 1 exactly, where 256-wide tiles waste a quarter of the MFMA work)
template <int BM, int BN, int WN, int FC = 4>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p)
{
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

    f32x4 ra[NA], rb[NB];
    // Pixel -> (image, oh, ow) of every B row this thread stages: decoded once by division,
    // then advanced by 32 pixels per step with carries only (32 = d_img*HWo + dq*Wo + dr).
    int r_img[NB], r_oh[NB], r_ow[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int pix = pbeg + rb0 + RPB * q;
        r_img[q] = pix / HWo;
        const int rem = pix - r_img[q] * HWo;
        r_oh[q] = rem / p.Wo;
        r_ow[q] = rem - r_oh[q] * p.Wo;
    }
    const int d_img = 32 / HWo, rem32 = 32 - d_img * HWo;
    const int dq = rem32 / p.Wo, dr = rem32 - dq * p.Wo;
    // Out-of-range pixels / padded taps read a 16-B block of zeros: no select on loaded data.
    // gload(s) must be called for s = 0, 1, 2, ... in order (it advances the row state).
    auto gload = [&](int s) {
        const int pb = pbeg + s * 32;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int pix = pb + ra0 + RPA * q;
            const float* src = (pix < pend && m0 + 4 * ca < p.M) ? p.dY + (size_t)pix * p.M + m0 + 4 * ca : p.zeros;
            ra[q] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int pix = pb + rb0 + RPB * q;
            const int ih = r_oh[q] * p.stride + e.x, iw = r_ow[q] * p.stride + e.y;
            const bool ok = bstage && pix < pend && e.w && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
            const float* src = ok ? p.X + ((size_t)(r_img[q] * p.Hi + ih) * p.Wi + iw) * p.Ci + e.z : p.zeros;
            rb[q] = *reinterpret_cast<const f32x4*>(src);
            int ow = r_ow[q] + dr, oh = r_oh[q] + dq, im = r_img[q] + d_img;
            if (ow >= p.Wo) { ow -= p.Wo; ++oh; }
            if (oh >= p.Ho) { oh -= p.Ho; ++im; }
            r_ow[q] = ow; r_oh[q] = oh; r_img[q] = im;
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * 32 * BM + ra0 * BM + 4 * ca;
        float* b = Bs + buf * 32 * BNP + rb0 * BNP + 4 * cb;
#pragma unroll
        for (int q = 0; q < NA; ++q) *reinterpret_cast<f32x4*>(a + q * RPA * BM) = ra[q];
        if (bstage) {
#pragma unroll
            for (int q = 0; q < NB; ++q) *reinterpret_cast<f32x4*>(b + q * RPB * BNP) = rb[q];
        }
    };

    f32x4 acc[4][FC];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const float* A = As + buf * 32 * BM + lg * BM + wm * 64 + 4 * li;
        const float* B = Bs + buf * 32 * BNP + lg * BNP + wn * (16 * FC) + FC * li;
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
        if (s + 1 < nsteps) lstore(buf ^ 1);
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
                    for (int c = 0; c < FC; ++c) dst[c] = acc[r][c][q];
                }
            }
    }
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
    static const int xcd = getenv("FM_PW_XCD") ? atoi(getenv("FM_PW_XCD")) : 1;
    p.tilesL = tilesL; p.nsplit = splits; p.xcd = xcd;
    dim3 grid(tilesL * splits);
    const int cc = (S + 15) / 16;
    switch (cc) {
    case 1: hipLaunchKernelGGL(wgrad_skinny_kernel<1>, grid, dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(wgrad_skinny_kernel<2>, grid, dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL(wgrad_skinny_kernel<3>, grid, dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL(wgrad_skinny_kernel<4>, grid, dim3(256), 0, s, p); break;
    case 5: hipLaunchKernelGGL(wgrad_skinny_kernel<5>, grid, dim3(256), 0, s, p); break;
    case 6: hipLaunchKernelGGL(wgrad_skinny_kernel<6>, grid, dim3(256), 0, s, p); break;
    case 7: hipLaunchKernelGGL(wgrad_skinny_kernel<7>, grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(wgrad_skinny_kernel<8>, grid, dim3(256), 0, s, p); break;
    }
    return splits;
}

// N-tile width for an (M, Nw) weight gradient: 128 with the 128-row tiles; 192 when it tiles Nw exactly
// (ResNet layer 1: 576 = 3 x 192) and 256 otherwise for the 64-row tiles
int wgrad_tile_n(int M, int Nw)
{
    static const int t192 = getenv("FM_WGRAD192") ? atoi(getenv("FM_WGRAD192")) : 1;
    if (M >= 128) return 128;
    return (t192 && Nw % 192 == 0) ? 192 : 256;
}

void launch_wgrad(const WgradParams& p, int splits, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int LDS_L = 2 * 32 * (128 + 128) * 4;
    constexpr int LDS_S = 2 * 32 * (64 + 256) * 4;
    constexpr int LDS_T = 2 * 32 * (64 + 192 + 16) * 4;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<128, 128, 2>), LDS_L, "wgrad_kernel<128, 128, 2>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 256, 4>), LDS_S, "wgrad_kernel<64, 256, 4>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&wgrad_kernel<64, 192, 4, 3>), LDS_T, "wgrad_kernel<64, 192, 4, 3>");
        attr_done = true;
    }
    dim3 grid(p.tilesM * p.tilesN, splits);
    if (p.M >= 128)
        hipLaunchKernelGGL((wgrad_kernel<128, 128, 2>), grid, dim3(256), LDS_L, s, p);
    else if (p.bn == 192)
        hipLaunchKernelGGL((wgrad_kernel<64, 192, 4, 3>), grid, dim3(256), LDS_T, s, p);
    else
        hipLaunchKernelGGL((wgrad_kernel<64, 256, 4>), grid, dim3(256), LDS_S, s, p);
}
