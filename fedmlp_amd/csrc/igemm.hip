// Implicit-GEMM convolution (forward + data gradient) on fp32 MFMA, gfx950.
//
// Reference op replaced: every nn.Conv2d forward inside net(images)
// (utils/local_training.py:657, 937-947, 983, 1030, 1178) and its input
// gradient inside loss.backward() (:674, 965, 1191), which the reference gets
// from cuDNN through torchvision's resnet18 (model/all_models.py:53-54).
//
// Roofline: fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 157 TFLOP/s).  Design:
//  * D = Wp * Xg^T with output CHANNELS on the MFMA row axis, so each lane ends
//    up with 4 consecutive channels of one pixel -> one 16-B NHWC store.
//  * 256 threads = 4 waves, each wave owns a 64x64 sub-tile (4x4 MFMA tiles,
//    64 accumulator VGPRs); block tile 128x128 (M>=128) or 64x256 (M==64).
//  * K advances 32 floats per step through double-buffered LDS; both operand
//    tiles are k-contiguous rows of 128 B whose 16-B chunks are XOR-swizzled by
//    (row & 7): conflict-free ds_read_b128 / ds_write_b128 on the 64-bank LDS.
//  * K order inside a step is permuted (lane group g reads k = 4g..4g+3 with
//    ONE ds_read_b128 and feeds 4 MFMAs); legal because A and B use the same
//    permutation.
//  * Global->LDS goes through registers (next step's loads are issued before
//    the current step's MFMAs) because the pixel operand is a bounds-checked
//    gather; zero-fill implements padding.
//  * Epilogue variants (runtime-uniform): raw store + per-channel sum/sumsq
//    partials (train-mode BN statistics, fixed reduction order), or folded
//    eval-BN affine + residual + ReLU, or plain residual add (dgrad).
//  * blockIdx -> tile map is XCD-aware: the 8 XCDs each take a contiguous run
//    of logical tiles (m fastest) so blocks sharing a pixel tile share an L2.
#include "common.h"

template <int BM, int BN, int WN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p)
{
    constexpr int WM = 4 / WN;
    static_assert(BM == WM * 64 && BN == WN * 64, "wave tile is 64x64");
    constexpr int RA = BM / 32;
    constexpr int RB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][32]
    float* Bs = smem + 2 * BM * 32;   // [2][BN][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;

    // XCD-aware bijective remap (blocks b and b+8 share an XCD)
    const int ntile = p.tilesM * p.tilesN;
    const int bid = blockIdx.x;
    const int q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = logical % p.tilesM, tn = logical / p.tilesM;
    const int grp = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;
    const int HWg = p.Hg * p.Wg;
    const int npix = p.imgs_per_group * HWg;
    const int Ktot = p.nsteps * 32;

    // ---- loader state: thread -> (row lr + 32q, chunk lc) --------------------
    const int lc = tid & 7, lr = tid >> 3;
    const int sc4 = (lc ^ (lr & 7)) << 2;          // swizzled float offset inside the row
    int ih0[RB], iw0[RB], xb[RB];
    bool rv[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int n = n0 + lr + 32 * q;
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
    const float* wrow = p.W + (size_t)(m0 + lr) * Ktot + lc * 4;

    f32x4 ra[RA], rb[RB];
    auto gload = [&](int s) {
#pragma unroll
        for (int q = 0; q < RA; ++q)
            ra[q] = *reinterpret_cast<const f32x4*>(wrow + (size_t)q * 32 * Ktot + s * 32);
        // where this step's chunk comes from: scalar tap lookup (kernel arguments), or for
        // the stem one kernel row per step and one kernel column per chunk
        int dh, dw, c0;
        bool cv = true;
        if (p.stem_kw) {
            dh = s - p.stem_pad; dw = lc - p.stem_pad; c0 = 0; cv = lc < p.stem_kw;
        } else {
            const int k0 = s * 32;
            const int t = k0 / p.Ci;
            dh = p.dh[t]; dw = p.dw[t]; c0 = k0 - t * p.Ci + lc * 4;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int ih = ih0[q] + dh, iw = iw0[q] + dw;
            const bool ok = rv[q] && cv && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
            // branch-free padding: out-of-image taps read a 16-B block of zeros, so no select
            // ever touches the loaded data (a select would drag the vmcnt wait ahead of the MFMAs)
            const float* src = ok ? p.X + (size_t)(xb[q] + (ih * p.Wi + iw) * p.Ci + c0) : p.zeros;
            rb[q] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * BM * 32 + lr * 32 + sc4;
        float* b = Bs + buf * BN * 32 + lr * 32 + sc4;
#pragma unroll
        for (int q = 0; q < RA; ++q) *reinterpret_cast<f32x4*>(a + q * 32 * 32) = ra[q];
#pragma unroll
        for (int q = 0; q < RB; ++q) *reinterpret_cast<f32x4*>(b + q * 32 * 32) = rb[q];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (floats) inside one buffer; row & 7 == li & 7
    const int sw = li & 7;
    const int aoff = (wm * 64 + li) * 32;
    const int boff = (wn * 64 + li) * 32;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < p.nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < p.nsteps) gload(s + 1);
        const float* A = As + buf * BM * 32 + aoff;
        const float* B = Bs + buf * BN * 32 + boff;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ch = ((4 * s2 + lg) ^ sw) << 2;
            f32x4 a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const f32x4*>(A + r * 16 * 32 + ch);
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const f32x4*>(B + c * 16 * 32 + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
        }
        if (s + 1 < p.nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------
    // acc[r][c][q] = D[m = m0 + wm*64 + 16r + 4*lg + q][n = n0 + wn*64 + 16c + li]
    const int mbase = m0 + wm * 64 + 4 * lg;
    if (p.stats) {
        float* red = smem;            // [WN][BM][2]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
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
                    const int ml = wm * 64 + 16 * r + 4 * lg + q;
                    red[(wn * BM + ml) * 2 + 0] = u;
                    red[(wn * BM + ml) * 2 + 1] = v;
                }
            }
        }
        __syncthreads();
        if (tid < BM) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                u += red[(w * BM + tid) * 2 + 0];
                v += red[(w * BM + tid) * 2 + 1];
            }
            float* st = p.stats + (size_t)(grp * p.tilesN + tn) * 2 * p.M;
            st[m0 + tid] = u;
            st[p.M + m0 + tid] = v;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + wn * 64 + 16 * c + li;
        if (n >= npix) continue;
        const int img = n / HWg;
        const int rem = n - img * HWg;
        const int hg = rem / p.Wg;
        const int wg = rem - hg * p.Wg;
        const size_t o = ((size_t)((grp * p.imgs_per_group + img) * p.Ho + hg * p.os + p.oh0) * p.Wo
                          + (wg * p.os + p.ow0)) * p.Co;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mbase + 16 * r;
            f32x4 v = acc[r][c];
            if (p.scale) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + m);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + m);
                v = v * sc + sh;
            }
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o + m);
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            *reinterpret_cast<f32x4*>(p.Y + o + m) = v;
        }
    }
}

int igemm_tile_m(int M) { return M >= 128 ? 128 : 64; }
int igemm_tile_n(int M) { return M >= 128 ? 128 : 256; }

void launch_igemm(const IgemmParams& p, int groups, hipStream_t s)
{
    static bool attr_done = false;
    constexpr int LDS_L = 2 * (128 + 128) * 32 * 4;
    constexpr int LDS_S = 2 * (64 + 256) * 32 * 4;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<128, 128, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_L);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<64, 256, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_S);
        attr_done = true;
    }
    dim3 grid(p.tilesM * p.tilesN, groups);
    if (p.M >= 128)
        hipLaunchKernelGGL((igemm_kernel<128, 128, 2>), grid, dim3(256), LDS_L, s, p);
    else
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4>), grid, dim3(256), LDS_S, s, p);
}
