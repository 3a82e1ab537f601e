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
#include "common.h"

template <int BM, int BN, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p)
{
    constexpr int WM = 4 / WN;
    static_assert(BM == WM * 64 && BN == WN * 64, "wave tile is 64x64");
    constexpr int CA = BM / 4, CB = BN / 4;      // 16-B chunks per tile row
    constexpr int RPA = 256 / CA, NA = 32 / RPA; // rows per pass / passes
    constexpr int RPB = 256 / CB, NB = 32 / RPB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][32][BM]
    float* Bs = smem + 2 * 32 * BM;   // [2][32][BN]

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
    const int gc = (n0 >> 2) + cb;
    int4 e = {0, 0, 0, 0};
    if (gc < (p.Nw >> 2)) e = p.tab[gc];

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
            const bool ok = pix < pend && e.w && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
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
        float* b = Bs + buf * 32 * BN + rb0 * BN + 4 * cb;
#pragma unroll
        for (int q = 0; q < NA; ++q) *reinterpret_cast<f32x4*>(a + q * RPA * BM) = ra[q];
#pragma unroll
        for (int q = 0; q < NB; ++q) *reinterpret_cast<f32x4*>(b + q * RPB * BN) = rb[q];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const float* A = As + buf * 32 * BM + lg * BM + wm * 64 + 4 * li;
        const float* B = Bs + buf * 32 * BN + lg * BN + wn * 64 + 4 * li;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(A + kk * 4 * BM);
            const f32x4 b = *reinterpret_cast<const f32x4*>(B + kk * 4 * BN);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[c], acc[r][c], 0, 0, 0);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    // acc[r][c][q] = dW[m0 + wm*64 + 16*lg + 4*q + r][n0 + wn*64 + 4*li + c]
    const int n = n0 + wn * 64 + 4 * li;
    if (n < p.Nw) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm * 64 + 16 * lg + 4 * q + r;
                if (m >= p.M) continue;
                const f32x4 v = {acc[r][0][q], acc[r][1][q], acc[r][2][q], acc[r][3][q]};
                *reinterpret_cast<f32x4*>(p.slab + ((size_t)split * p.M + m) * p.Nw + n) = v;
            }
    }
}

void launch_wgrad(const WgradParams& p, int splits, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int LDS_L = 2 * 32 * (128 + 128) * 4;
    constexpr int LDS_S = 2 * 32 * (64 + 256) * 4;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<128, 128, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_L);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<64, 256, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_S);
        attr_done = true;
    }
    dim3 grid(p.tilesM * p.tilesN, splits);
    if (p.M >= 128)
        hipLaunchKernelGGL((wgrad_kernel<128, 128, 2>), grid, dim3(256), LDS_L, s, p);
    else
        hipLaunchKernelGGL((wgrad_kernel<64, 256, 4>), grid, dim3(256), LDS_S, s, p);
}
