// Streaming 1x1 stride-1 convolution on fp32 MFMA for small K (K = input channels <= 256), gfx950.
//
// Reference op replaced: the expand / project 1x1 convolutions of EfficientNet-B0's MBConv blocks (and
// their data gradients, which are 1x1 convolutions with the transposed weights) inside net(images) /
// loss.backward() (utils/local_training.py:657, 674, 937-947, 965, 1178, 1191 through
// efficientnet_pytorch 0.7.1, model/efficientnet.py:28-33).
//
// Why not igemm.hip: with K = 16..240 a 128x128 tile is 1-8 K-steps long, so the tiled kernel spends its
// time on per-tile prologues, barriers and epilogues, and these layers are HBM-bound anyway (output 3-6x
// larger than input).  Here the whole weight slice of an M-tile ([64][K] floats, <= 64 KB) is staged ONCE per
// block in LDS; every wave then streams its own pixels: B fragments go global -> VGPR directly (16 B per
// lane, the same K permutation as the A fragments so one b128 feeds 4 MFMAs), 32 pixels per iteration,
// no barrier and no LDS traffic for the pixel operand in the loop, 16-B NHWC stores straight from the
// accumulators.  BN statistics are accumulated in registers over the wave's whole pixel range and folded
// once per block; the partials land in the [group][tilesN][2][M] layout igemm's epilogue uses (a block that
// covers several virtual tiles writes zeros for the others), so the BN finalize kernel is unchanged.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "split3.h"

namespace {

[[maybe_unused]] constexpr int KB = 4;             // 16-k chunks preloaded per pass (64 k)

// RT = 16-row MFMA tiles per block (MT = 16*RT output channels), P = 16-pixel groups per wave iteration:
// the shipped instantiation is <4,2> = 64 channels x 32 pixels
// STATS: train forward (BN partial sums kept in registers across the pixel loop) -- a template parameter so that the eval /
// data-gradient launches do not carry the 32 sum registers (3 -> 4 waves per SIMD)
// GATE: 1 = the operand is multiplied by a per-image, per-channel gate on load (IgemmParams::gate); 2 = BN affine + Swish of
// the operand first (IgemmParams::psc / psh, per statistics group): a_s = swish(bn1(y_d)) * gate is never written (train forward)
// SP = 6 / 9: the fp32 products as exact bf16 partial products (split3.h): the weight slice is split ONCE per block on its way
// into LDS ([K/32][3 planes][MT rows][64 B], chunk g of a row = k 4g..4g+3, 16+4g..16+4g+3 of the 32-k block, slot g ^ ((row>>1)&3)),
// a lane splits its two 16-B pixel pieces of a 32-k block in registers (after the gate / BN + Swish prologue).  K <= 192
// (the planes take 1.5x the fp32 bytes: 72 KB there).
template <int RT, int P, bool STATS, int GATE = 0, int SP = 0>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const IgemmParams p, int vt_per_block, int blocks_per_group)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int MT = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float As[];     // [K/16][MT][16], chunk c of row r in slot c ^ ((r>>1)&3)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int K = p.Ci, nkk = K >> 4;
    const int m0 = blockIdx.x * MT;
    const int grp = blockIdx.y / blocks_per_group, pb_idx = blockIdx.y - grp * blocks_per_group;
    const int BNv = p.M >= 128 ? 128 : 192;                         // igemm's pixel-tile width for this M (statistics layout)
    const int npix = p.imgs_per_group * p.Hg * p.Wg;                // pixels per group
    const int vt0 = pb_idx * vt_per_block;
    const int pb = vt0 * BNv, pe = min(npix, pb + vt_per_block * BNv);
    const size_t gbase = (size_t)grp * npix;

    // ---- stage the weight slice once ---------------------------------------------------------------
    const int cpr = K >> 2;                                         // 16-B chunks per weight row
    const int nk32 = (K + 31) >> 5;
    unsigned char* const Ap = reinterpret_cast<unsigned char*>(As);
    if constexpr (SP != 0) {
        for (int idx = tid; idx < MT * nk32 * 4; idx += 256) {
            const int row = idx / (nk32 * 4), rem = idx - row * (nk32 * 4);
            const int kb32 = rem >> 2, g = rem & 3;
            const int ka = 32 * kb32 + 4 * g, kc = ka + 16;
            const bool rv = m0 + row < p.M;
            const float* wr = p.W + (size_t)(rv ? m0 + row : 0) * K;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 x0 = (rv && ka < K) ? *reinterpret_cast<const f32x4*>(wr + ka) : z;
            const f32x4 x1 = (rv && kc < K) ? *reinterpret_cast<const f32x4*>(wr + kc) : z;
            sp_u32x4 H, M, L;
            split3(x0, x1, H, M, L);
            unsigned char* dst = Ap + ((size_t)(kb32 * 3) * MT + row) * 64 + ((g ^ ((row >> 1) & 3)) << 4);
            *reinterpret_cast<sp_u32x4*>(dst) = H;
            *reinterpret_cast<sp_u32x4*>(dst + MT * 64) = M;
            *reinterpret_cast<sp_u32x4*>(dst + 2 * MT * 64) = L;
        }
    } else
    for (int idx = tid; idx < MT * cpr; idx += 256) {
        const int row = idx / cpr, ch = idx - row * cpr;
        const int kk = ch >> 2, c4 = ch & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + row < p.M) v = *reinterpret_cast<const f32x4*>(p.W + (size_t)(m0 + row) * K + ch * 4);
        *reinterpret_cast<f32x4*>(As + (kk * MT + row) * 16 + ((c4 ^ ((row >> 1) & 3)) << 2)) = v;
    }
    __syncthreads();

    const int aoff = li * 16 + ((lg ^ ((li >> 1) & 3)) << 2);       // fragment read offset inside a 16-row tile
    const int nrt = min(RT, (p.M - m0 + 15) >> 4);                   // 16-row tiles of this M-tile that hold real rows
    f32x4 s1[STATS ? RT : 1], s2[STATS ? RT : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int r = 0; r < RT; ++r) { s1[r] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[r] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    constexpr int PPI = 16 * P;                                     // pixels per wave iteration
    const int n_iter = (pe - pb + PPI - 1) / PPI;
    // B fragments of KB chunks of one iteration (no explicit prefetch of the next iteration: it measured
    // slower than the third wave per SIMD that its registers cost)
    auto load_b = [&](int it, f32x4 (&b)[P][KB], int kb) {
#pragma unroll
        for (int g = 0; g < P; ++g) {
            const int px = pb + it * PPI + 16 * g + li;
            const bool v = px < pe;
            const float* xp = p.X + (gbase + (v ? px : pb)) * K + 4 * lg;
            const float* gp = GATE ? p.gate + ((gbase + (v ? px : pb)) / p.gate_HW) * K + 4 * lg : nullptr;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                b[g][k] = (v && kb + k < nkk) ? *reinterpret_cast<const f32x4*>(xp + (kb + k) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (GATE == 2) {
                    if (v && kb + k < nkk) {
                        const int ko = grp * K + 4 * lg + (kb + k) * 16;
                        f32x4 u = b[g][k] * *reinterpret_cast<const f32x4*>(p.psc + ko) + *reinterpret_cast<const f32x4*>(p.psh + ko);
#pragma unroll
                        for (int j = 0; j < 4; ++j) u[j] = u[j] * __builtin_amdgcn_rcpf(1.f + __expf(-u[j]));     // act_fwd's fast form (effnet.hip)
                        b[g][k] = u;
                    }
                }
                if constexpr (GATE != 0) { if (v && kb + k < nkk) b[g][k] *= *reinterpret_cast<const f32x4*>(gp + (kb + k) * 16); }
            }
        }
    };
    for (int it = wave; it < n_iter; it += 4) {
        const int pix0 = pb + it * PPI;
        f32x4 acc[RT][P];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int g = 0; g < P; ++g) acc[r][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nkk; kb += KB) {
            f32x4 b[P][KB];
            load_b(it, b, kb);
            if constexpr (SP != 0) {
#pragma unroll
                for (int pp = 0; pp < KB / 2; ++pp) {
                    if (kb + 2 * pp < nkk) {
                        sp_u32x4 bh[P], bm[P], bl[P];
#pragma unroll
                        for (int g = 0; g < P; ++g) split3(b[g][2 * pp], b[g][2 * pp + 1], bh[g], bm[g], bl[g]);
                        const unsigned char* A3 = Ap + ((size_t)(((kb >> 1) + pp) * 3) * MT + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            if (r < nrt) {
                                const sp_u32x4 ah = *reinterpret_cast<const sp_u32x4*>(A3 + r * 1024);
                                const sp_u32x4 am = *reinterpret_cast<const sp_u32x4*>(A3 + r * 1024 + MT * 64);
                                const sp_u32x4 al = *reinterpret_cast<const sp_u32x4*>(A3 + r * 1024 + 2 * MT * 64);
#pragma unroll
                                for (int g = 0; g < P; ++g) acc[r][g] = mfma_split<SP>(ah, am, al, bh[g], bm[g], bl[g], acc[r][g]);
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (kb + k >= nkk) break;
                const float* A = As + (kb + k) * MT * 16 + aoff;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (r >= nrt) break;                  // row tiles past M (M = 16, 32, 48 or a partial last M-tile)
                    const f32x4 a = *reinterpret_cast<const f32x4*>(A + r * 256);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int g = 0; g < P; ++g)
                            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[g][k][j], acc[r][g], 0, 0, 0);
                }
            }
        }
        // ---- epilogue of the iteration: acc[r][g][q] = D[m0 + 16r + 4lg + q][pix0 + 16g + li] ------------
#pragma unroll
        for (int g = 0; g < P; ++g) {
            const int px = pix0 + 16 * g + li;
            const bool pv = px < pe;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int m = m0 + 16 * r + 4 * lg;
                f32x4 v = acc[r][g];
                if constexpr (STATS) { s1[r] += v; s2[r] += v * v; }     // rows of padded pixels are exact zeros
                if (!pv || m >= p.M) continue;
                const size_t o = (gbase + px) * p.Co + m;
                if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + m) + *reinterpret_cast<const f32x4*>(p.shift + m);
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
                if (p.relu == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (p.relu == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fm_swish_f32(v[q]);
                }
                *reinterpret_cast<f32x4*>(p.Y + o) = v;
            }
        }
    }

    // ---- BN statistics: fold the 16 pixel lanes, then the 4 waves (fixed order) -----------------------
    if constexpr (STATS) {
        __syncthreads();                       // every wave is done with the weight slice: reuse the LDS
        float* red = As;                       // [4 waves][MT][2]
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
                if (li == 0) {
                    const int ml = 16 * r + 4 * lg + q;
                    red[(wave * MT + ml) * 2 + 0] = u;
                    red[(wave * MT + ml) * 2 + 1] = v;
                }
            }
        __syncthreads();
        if (tid < MT && m0 + tid < p.M) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                u += red[(w * MT + tid) * 2 + 0];
                v += red[(w * MT + tid) * 2 + 1];
            }
            // this block's sums go to its first virtual tile, zeros to the others it covers
            for (int t = 0; t < vt_per_block && vt0 + t < p.tilesN; ++t) {
                float* st = p.stats + (size_t)(grp * p.tilesN + vt0 + t) * 2 * p.M;
                st[m0 + tid] = t == 0 ? u : 0.f;
                st[p.M + m0 + tid] = t == 0 ? v : 0.f;
            }
        }
    }
#endif
}

}  // namespace

// Returns false when the conv is not a plain small-K 1x1 stride-1 GEMM (the caller then runs igemm).
// shapes the streaming kernel takes (also asked by the engine before it hands over a gate prologue)
bool conv1x1_stream_takes(int Ci, int M, int Co)
{
    static const int enabled = fm_tune("FM_STREAM1X1", 1);
    return enabled && Ci <= 256 && Ci % 16 == 0 && Co == M;
}
bool launch_conv1x1_stream(const IgemmParams& p, int groups, hipStream_t s)
{
    static const int enabled = fm_tune("FM_STREAM1X1", 1);
    if (!enabled || p.stem_kw || p.ntaps != 1 || p.dh[0] != 0 || p.dw[0] != 0) return false;
    if (p.sg != 1 || p.os != 1 || p.oh0 != 0 || p.ow0 != 0) return false;
    if (p.Hg != p.Ho || p.Wg != p.Wo || p.Hi != p.Ho || p.Wi != p.Wo) return false;
    if (p.Ci > 256 || p.Ci % 16 != 0 || p.Co != p.M) return false;
    // (128-channel instantiations for wide outputs measured 2 % slower end to end, both <8,1> -- shorter pixel
    // iterations -- and <8,2> -- 252 VGPRs, two waves per SIMD: reading x once per 128 instead of once per 64
    // channels does not pay for either)
    const int MT = 64;
    const int tilesM = (p.M + MT - 1) / MT;
    // virtual tiles per block: enough pixels that staging the weight slice (MT*K*4 B) stays ~10 % of the block's
    // traffic, but no more -- small blocks launched in order keep the concurrently running ones on
    // neighbouring memory (DRAM locality, tools/ew_bw.hip)
    static const int vt_env = fm_tune("FM_STREAM_VT", 0);
    const int bnv = igemm_tile_n(p.M);
    int vt = std::max(1, (int)((10LL * MT * p.Ci + (long long)bnv * (p.Ci + MT) - 1) / ((long long)bnv * (p.Ci + MT))));
    if (vt_env > 0) vt = vt_env;
    const int bpg = (p.tilesN + vt - 1) / vt;
    // split-product form (split3.h) while its weight planes fit two blocks per CU
    const int sp = p.Ci <= 192 ? p.sp : 0;
    const size_t lds = sp ? std::max<size_t>((size_t)((p.Ci + 31) / 32) * 3 * MT * 64, (size_t)4 * MT * 2 * 4)
                          : std::max<size_t>((size_t)MT * p.Ci * 4, (size_t)4 * MT * 2 * 4);
    const dim3 grid(tilesM, bpg * groups);
    constexpr int LDS_MAX = 6 * 3 * 64 * 64;        // 72 KB (K = 192 planes) >= the fp32 slice of K = 256 (64 KB)
#define FM_C1_LAUNCH(STATS_, GATE_)                                                                                              \
    do {                                                                                                                         \
        static bool done_ = false;                                                                                               \
        if (!done_) {                                                                                                            \
            set_max_dyn_lds(reinterpret_cast<const void*>(&conv1x1_stream_kernel<4, 2, STATS_, GATE_, 0>), LDS_MAX, "conv1x1_stream_kernel"); \
            set_max_dyn_lds(reinterpret_cast<const void*>(&conv1x1_stream_kernel<4, 2, STATS_, GATE_, 6>), LDS_MAX, "conv1x1_stream_kernel<6>"); \
            set_max_dyn_lds(reinterpret_cast<const void*>(&conv1x1_stream_kernel<4, 2, STATS_, GATE_, 9>), LDS_MAX, "conv1x1_stream_kernel<9>"); \
            done_ = true;                                                                                                        \
        }                                                                                                                        \
        if (sp == 6) hipLaunchKernelGGL((conv1x1_stream_kernel<4, 2, STATS_, GATE_, 6>), grid, dim3(256), lds, s, p, vt, bpg);   \
        else if (sp == 9) hipLaunchKernelGGL((conv1x1_stream_kernel<4, 2, STATS_, GATE_, 9>), grid, dim3(256), lds, s, p, vt, bpg); \
        else hipLaunchKernelGGL((conv1x1_stream_kernel<4, 2, STATS_, GATE_, 0>), grid, dim3(256), lds, s, p, vt, bpg);           \
    } while (0)
    if (p.gate && p.psc) {
        if (!p.stats) return false;                          // the affine prologue exists for the train forward only
        FM_C1_LAUNCH(true, 2);
    } else if (p.gate) {
        FM_C1_LAUNCH(false, 1);
    } else if (p.stats) {
        FM_C1_LAUNCH(true, 0);
    } else {
        FM_C1_LAUNCH(false, 0);
    }
#undef FM_C1_LAUNCH
    return true;
}
