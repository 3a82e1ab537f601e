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
//    gather; padding reads a zero page, tap offsets are scalar kernel arguments,
//    so nothing in the loop waits on a dependent load.
//  * STREAM-K scheduling: ResNet's pixel counts are 49*2^k, so a one-tile-per-
//    block grid leaves the last round of the 256 CUs ~23 % empty (392 / 784 /
//    1568 tiles on 512 block slots).  Instead the launch is persistent: the
//    (tile, K-step) space is cut into equal contiguous ranges, one per block.
//    A block whose range ends inside a tile stores its partial accumulators to a
//    slab, releases them at agent scope and bumps the tile's arrival counter; the
//    LAST arriver re-reads every partial of the tile in segment order (so the
//    sum does not depend on arrival order: deterministic) and runs the epilogue.
//    Nobody waits on anybody, so no residency assumption and no deadlock.
//  * Epilogue variants (runtime-uniform): raw store + per-channel sum/sumsq
//    partials (train-mode BN statistics, fixed reduction order), or folded
//    eval-BN affine + residual + ReLU, or plain residual add (dgrad).
//  * block -> range map is XCD-aware (blocks b, b+8, .. share an XCD and get
//    adjacent ranges, i.e. neighbouring tiles share an L2).
#include <stdlib.h>

#include <algorithm>

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
    const int lc = tid & 7, lr = tid >> 3;         // loader: chunk, row
    const int sc4 = (lc ^ (lr & 7)) << 2;          // swizzled float offset inside the row
    const int sw = li & 7;
    const int aoff = (wm * 64 + li) * 32;
    const int boff = (wn * 64 + li) * 32;

    const int nsteps = p.nsteps;
    const int Ktot = nsteps * 32;
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
        const int tm = tl % p.tilesM, tn = tl / p.tilesM;
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- loader state: thread -> (row lr + 32q, chunk lc) ----------------
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
                const int kk = s * 32;
                const int t = kk / p.Ci;
                dh = p.dh[t]; dw = p.dw[t]; c0 = kk - t * p.Ci + lc * 4;
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

        gload(k0);
        lstore(0);
        __syncthreads();
        for (int s = k0; s < k1; ++s) {
            const int buf = (s - k0) & 1;
            if (s + 1 < k1) gload(s + 1);
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
            if (s + 1 < k1) lstore(buf ^ 1);
            __syncthreads();
        }

        // ---- stream-K fix-up: partial tiles meet in the slab ------------------------
        if (k0 != 0 || k1 != nsteps) {
            // slot 0: the block's first segment, slot 1: its last one (middle ones are whole tiles)
            float* mine = p.slab + ((size_t)(rbk * 2 + (seg_first_of_block ? 0 : 1)) * 16) * 1024;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<f32x4*>(mine + ((r * 4 + c) * 256 + tid) * 4) = acc[r][c];
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
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int bb = b_first; bb <= b_last; ++bb) {
                const long long sstart = max(t0, (long long)bb * S);
                const float* src = p.slab + ((size_t)(bb * 2 + (sstart == (long long)bb * S ? 0 : 1)) * 16) * 1024;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[r][c] += *reinterpret_cast<const f32x4*>(src + ((r * 4 + c) * 256 + tid) * 4);
            }
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
                for (int ww = 0; ww < WN; ++ww) {
                    u += red[(ww * BM + tid) * 2 + 0];
                    v += red[(ww * BM + tid) * 2 + 1];
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
        __syncthreads();       // LDS (stats scratch) is re-staged by the next segment
    }
}

int igemm_tile_m(int M) { return M >= 128 ? 128 : 64; }
int igemm_tile_n(int M) { return M >= 128 ? 128 : 256; }
int igemm_max_blocks() { return 512; }    // 2 blocks per CU x 256 CUs (64-80 KB LDS, <=256 VGPRs)

void launch_igemm(IgemmParams p, int groups, hipStream_t s)
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
    const long long T = (long long)p.tilesM * p.tilesN * groups;
    p.total_steps = T * p.nsteps;
    // persistent grid: all 512 block slots whenever there are >= 2 steps for each of them,
    // otherwise one tile per block.  FM_IGEMM_BLOCKS overrides the grid (tests force odd
    // splits so that every fix-up path runs on small shapes).
    static const int forced = getenv("FM_IGEMM_BLOCKS") ? atoi(getenv("FM_IGEMM_BLOCKS")) : 0;
    int nblk = p.total_steps >= 2LL * igemm_max_blocks() ? igemm_max_blocks()
                                                         : (int)std::min<long long>(igemm_max_blocks(), T);
    if (forced > 0) nblk = (int)std::min<long long>(std::min(forced, igemm_max_blocks()), p.total_steps);
    p.steps_per_block = (int)((p.total_steps + nblk - 1) / nblk);
    dim3 grid(nblk);
    if (p.M >= 128)
        hipLaunchKernelGGL((igemm_kernel<128, 128, 2>), grid, dim3(256), LDS_L, s, p);
    else
        hipLaunchKernelGGL((igemm_kernel<64, 256, 4>), grid, dim3(256), LDS_S, s, p);
}
