// Forward of ResNet-18's 7x7 / stride-2 stem from bf16 planes with the input ROWS and the whole weight resident in LDS, gfx950.
//
// Reference op replaced: net.conv1 inside net(images) (utils/local_training.py:657, 937-947, 983, 1030, 1178; torchvision resnet18,
// model/all_models.py:53-54) at the 224 x 224 input the benchmark is quoted on (112-pixel output rows).
//
// Same arithmetic as every other conv GEMM of the engine (split3.h: fp32 operands as three exact bf16 planes, SP = 6 / 9 partial
// products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).  igemm.hip's stem form gathers a 256-pixel im2col tile per 32-k stage
// and splits BOTH operands in registers in every stage; with K = 147 a tile is six stages of mostly fixed cost: 0.23 of the
// roofline.  Here nothing is gathered and nothing is split:
//  * the input arrives as planes of the zero-framed image, four channels per pixel (the fourth zero): [3][img][H + 6][W + 8][4]
//    bf16, 8 B per pixel (frame_nhwc3_kernel writes them beside the fp32 frame).  The 32-k fragment of output pixel (oh, ow) under
//    kernel row kh is then 64 CONTIGUOUS bytes -- framed pixels (2 oh + kh, 2 ow .. 2 ow + 7), k = 4 kw + c -- at a 16-B aligned
//    address; the eighth tap and the fourth channel meet zero weights.  K = 7 kernel rows x 32 = 224 (147 useful);
//  * a block owns FOUR output rows of one image: the 13 input rows they read are copied into LDS once (3 x 24 KB, a linear
//    LDS-DMA copy), next to ALL weight planes ([7][3][64][32] bf16 = 84 KB, staged once per block).  A wave (wm, wn) computes output
//    row wn, channels 32 wm .. + 31: 2 x 7 MFMA tiles; its B fragments are ds_read_b128 at stride 16 B straight out of the image
//    rows (neighbouring pixels' fragments overlap: lanes with equal li + lg read the same 16 B);
//  * no barrier, no DMA and no address arithmetic inside a tile's 588 MFMAs per wave: two barriers per tile (rows landed / rows
//    free).  Epilogues: raw fp32 + BatchNorm partial sums per tile (train), or folded eval BatchNorm + ReLU (teacher).
// Roofline: bf16 MFMA dense peak / SP = 416.7 TFLOP/s of fp32 products; the K padding costs 224 / 147 of the MFMA work.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split3.h"

#if __HIP_DEVICE_COMPILE__
template <int IMM> __device__ __forceinline__ sp_u32x4 sr_lds_read128(unsigned addr)
{
    sp_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int CTRL> __device__ __forceinline__ float sr_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
#endif

namespace {
constexpr int SR_WBYTES = 7 * 3 * 64 * 64;          // weight planes in LDS: [kh][plane][64 rows][64 B]
constexpr int SR_PLANE = 24 * 1024;                 // one plane of 13 framed rows (13 x 232 x 8 B = 24 128 B)
constexpr int SR_LDS = SR_WBYTES + 3 * SR_PLANE + 4 * 64 * 2 * 4;      // + statistics scratch [4][64][2] floats = 158 KB
}  // namespace

template <int SP>
__global__ __launch_bounds__(512, 2) void stem_rows_kernel(const StemRowsParams p)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int FR = 2, FC = 7;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned Wl = lds0, Xl = lds0 + SR_WBYTES;
    float* red = reinterpret_cast<float*>(smem + SR_WBYTES + 3 * SR_PLANE);       // [4][64][2]
    constexpr unsigned OOB = 0x80000000u;
    const int Wp = p.Wp, Ho = p.Ho, Wo = p.Wo;
    const int rows_bytes = 13 * Wp * 8;                       // bytes of the 13 framed rows of a tile, per plane
    const int tiles_per_img = Ho >> 2, ntiles = p.imgs * tiles_per_img;

    // ---- weights: all planes, once per block (job j = (kh, plane, 16-row group), 84 of them) -----------------------------------
    {
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wst), 0, (unsigned)SR_WBYTES, 0x00020000);
        const unsigned chunk = (unsigned)((lane & 3) ^ ((lane >> 3) & 3));        // source-side swizzle of the 64-B rows
        const unsigned vo = (unsigned)((lane >> 2) * 64) + chunk * 16;
        for (int j = wave; j < 84; j += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(size_t)(Wl + j * 1024), 16, vo, (unsigned)(j * 1024), 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Xp), 0, (unsigned)(3 * p.plane_bytes), 0x00020000);
    // fragment addresses: A (weights) row 32 wm + 16 r + li, chunk lg in slot lg ^ ((li >> 1) & 3); B (pixels) row 2 wn + kh of the
    // tile's 13, pixel 2 (16 c + li), chunk lg: 16 (li + lg) bytes into the column tile
    const unsigned Af = Wl + (unsigned)((32 * wm + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4));
    const unsigned Bf = Xl + (unsigned)(2 * wn * Wp * 8 + 16 * (li + lg));
    const unsigned krow = (unsigned)(Wp * 8);                 // one framed row

    // a tile's 13 framed rows: a linear copy per plane, 16 B per lane
    auto load_rows = [&](int tile) {
        const int img = tile / tiles_per_img, t4 = tile - img * tiles_per_img;
        const unsigned src0 = (unsigned)(((size_t)img * p.Hp + 8 * t4) * Wp * 8);
        for (int j = wave; j < 72; j += 8) {
            const int pl = j / 24, jj = j - 24 * pl;
            const unsigned off = (unsigned)((jj * 64 + lane) * 16);
            const unsigned vo = off < (unsigned)rows_bytes ? off : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(size_t)(Xl + pl * SR_PLANE + jj * 1024), 16, vo,
                                                     (unsigned)(pl * p.plane_bytes) + src0, 0, 0);
        }
    };
    // (measured and not kept: the epilogue deferred to the start of the next iteration, from a copy of the accumulators, so that its
    // stores drain beside the next tile's MFMAs: 487 vs 473 us -- the wait at the loop's top is not what a tile's time goes to; four
    // waves per block, each holding all 64 channels of an output row (half the B-fragment reads per MFMA, one wave per SIMD): the
    // same time as the eight-wave form)
    auto epilogue = [&](const f32x4 (&a)[FR][FC], int tile) {
        const int img = tile / tiles_per_img, t4 = tile - img * tiles_per_img;
        // ---- epilogue: a[r][c][q] = D[channel 32 wm + 16 r + 4 lg + q][pixel (4 t4 + wn, 16 c + li)] ----------------------------
        const int oh = 4 * t4 + wn;
        const int mbase = 32 * wm + 4 * lg;
        if (p.stats) {
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < FC; ++c) {
                    s1 += a[r][c];
                    s2 += a[r][c] * a[r][c];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float u = s1[q], v = s2[q];
                    u += sr_dpp<0xB1>(u); v += sr_dpp<0xB1>(v);
                    u += sr_dpp<0x4E>(u); v += sr_dpp<0x4E>(v);
                    u += sr_dpp<0x12C>(u); v += sr_dpp<0x12C>(v);
                    u += sr_dpp<0x128>(u); v += sr_dpp<0x128>(v);
                    if (li == 0) {
                        red[(wn * 64 + mbase + 16 * r + q) * 2 + 0] = u;
                        red[(wn * 64 + mbase + 16 * r + q) * 2 + 1] = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tid < 64) {
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    u += red[(ww * 64 + tid) * 2 + 0];
                    v += red[(ww * 64 + tid) * 2 + 1];
                }
                const int grp = img / p.imgs_per_group;
                const int tig = (img - grp * p.imgs_per_group) * tiles_per_img + t4;
                float* st = p.stats + (size_t)(grp * (p.imgs_per_group * tiles_per_img) + tig) * 2 * 64;
                st[tid] = u;
                st[64 + tid] = v;
            }
        }
        // the folded BatchNorm of the wave's two row tiles: loaded once per tile (per column and row tile, behind a branch each, every
        // load was waited for alone: 28 dependent round trips per tile in the eval form)
        f32x4 scv[FR], shv[FR];
        if (p.scale) {
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                scv[r] = *reinterpret_cast<const f32x4*>(p.scale + mbase + 16 * r);
                shv[r] = *reinterpret_cast<const f32x4*>(p.shift + mbase + 16 * r);
            }
        }
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            float* o = p.Y + (((size_t)img * Ho + oh) * Wo + 16 * c + li) * 64;
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                const int m = mbase + 16 * r;
                f32x4 v = a[r][c];
                if (p.scale) v = v * scv[r] + shv[r];
                if (p.relu == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                *reinterpret_cast<f32x4*>(o + m) = v;
            }
        }
    };
    if ((int)blockIdx.x < ntiles) load_rows(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's rows (issued before the previous tile's epilogue) and its stores
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        f32x4 acc[FR][FC];
#pragma unroll
        for (int r = 0; r < FR; ++r)
#pragma unroll
            for (int c = 0; c < FC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        sp_u32x4 A0[FR][3], A1[FR][3], Bb[3][3];
#define SR_READA(KH, DST)                                                                        \
    {                                                                                            \
        const unsigned a_ = Af + (unsigned)((KH) * 3 * 4096);      /* (LDS offsets are 16-bit immediates) */ \
        DST[0][0] = sr_lds_read128<0 * 4096>(a_);                                                 \
        DST[0][1] = sr_lds_read128<1 * 4096>(a_);                                                 \
        DST[0][2] = sr_lds_read128<2 * 4096>(a_);                                                 \
        DST[1][0] = sr_lds_read128<0 * 4096 + 1024>(a_);                                          \
        DST[1][1] = sr_lds_read128<1 * 4096 + 1024>(a_);                                          \
        DST[1][2] = sr_lds_read128<2 * 4096 + 1024>(a_);                                          \
    }
#define SR_READB(BASE, C, DST)                                                                   \
    {                                                                                            \
        DST[0] = sr_lds_read128<0 * SR_PLANE + 256 * (C)>(BASE);                                  \
        DST[1] = sr_lds_read128<1 * SR_PLANE + 256 * (C)>(BASE);                                  \
        DST[2] = sr_lds_read128<2 * SR_PLANE + 256 * (C)>(BASE);                                  \
    }
#define SR_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SR_MFMA(C, AC, BI)                                                                                           \
    _Pragma("unroll") for (int r = 0; r < FR; ++r)                                                                    \
        acc[r][C] = mfma_split<SP>(AC[r][0], AC[r][1], AC[r][2], Bb[BI][0], Bb[BI][1], Bb[BI][2], acc[r][C])
        // column G = 7 kh + c of the tile's 49: its fragment sits in Bb[G % 3], read TWO columns ahead (a column is only 12 MFMAs);
        // a kernel row's weight fragments travel with its first column.  The wait at a column's end leaves the newest reads in flight
        auto column = [&](auto g_c) {
            constexpr int G = decltype(g_c)::value, KH = G / 7, C = G % 7, G2 = G + 2, KH2 = G2 / 7, C2 = G2 % 7;
            if constexpr (G2 <= 48) {
                if constexpr (C2 == 0) {
                    if constexpr ((KH2 & 1) == 0) { SR_READA(KH2, A0); } else { SR_READA(KH2, A1); }
                }
                SR_READB(Bf + KH2 * krow, C2, Bb[G2 % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((KH & 1) == 0) { SR_MFMA(C, A0, G % 3); } else { SR_MFMA(C, A1, G % 3); }
            // (the column's accumulators are operands of the wait: the machine scheduler moved a bare one behind the column's FIRST MFMA)
            if constexpr (G2 > 48) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][C]), "+v"(acc[1][C])::"memory");
            else if constexpr (C2 == 0) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(acc[0][C]), "+v"(acc[1][C])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(acc[0][C]), "+v"(acc[1][C])::"memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        SR_READA(0, A0);
        SR_READB(Bf, 0, Bb[0]);
        SR_READB(Bf, 1, Bb[1]);
        SR_LGKM0();
#define SR_G(N) column(std::integral_constant<int, N>{});
#define SR_G7(N) SR_G(N) SR_G(N + 1) SR_G(N + 2) SR_G(N + 3) SR_G(N + 4) SR_G(N + 5) SR_G(N + 6)
        SR_G7(0) SR_G7(7) SR_G7(14) SR_G7(21) SR_G7(28) SR_G7(35) SR_G7(42)
#undef SR_G7
#undef SR_G
#undef SR_MFMA
#undef SR_READB
#undef SR_READA
        // every wave is done with the rows: the next tile's copy overwrites them while the epilogue below runs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tile + (int)gridDim.x < ntiles) load_rows(tile + gridDim.x);

        epilogue(acc, tile);
    }
#undef SR_LGKM0
#endif
}

// ---- the stem's weight planes: dst[kh][plane][64][32] bf16, k = 4 kw + c (kw < 7, c < 3; the rest zero), from the packed fp32
// rows [64][7 x 24 + 8] (k' = 24 kh + 3 kw + c).  One thread = one 16-B chunk (8 k) of one row of one kernel row
__global__ __launch_bounds__(256) void stem_weight_planes_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Kw)
{
#if __HIP_DEVICE_COMPILE__
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (kh, m, g)
    if (idx >= 7 * 64 * 4) return;
    const int g = idx & 3, m = (idx >> 2) & 63, kh = idx >> 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kw = 2 * g + (e >> 2), c = e & 3;
        v[e] = (kw < 7 && c < 3) ? w[(size_t)m * Kw + 24 * kh + 3 * kw + c] : 0.f;
    }
    sp_u32x4 H, M, L;
    split3(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, H, M, L);
    unsigned char* d = reinterpret_cast<unsigned char*>(dst) + ((size_t)(kh * 3) * 64 + m) * 64 + g * 16;
    *reinterpret_cast<sp_u32x4*>(d) = H;
    *reinterpret_cast<sp_u32x4*>(d + 64 * 64) = M;
    *reinterpret_cast<sp_u32x4*>(d + 2 * 64 * 64) = L;
#endif
}
void k_stem_weight_planes(const float* w, unsigned short* dst, int Kw, hipStream_t s)
{
    hipLaunchKernelGGL(stem_weight_planes_kernel, dim3(7), dim3(256), 0, s, w, dst, Kw);
}

// 7x7 / stride 2 / pad 3 on an input whose output rows are 112 pixels (7 column tiles of 16) in multiples of four rows, 64 output
// channels, planes of the whole framed batch below 2 GB (FM_STEM_ROWS=0 in a tuning build keeps igemm.hip's stem form)
bool stem_rows_takes(int k, int stride, int cout, int hout, int wout, long long plane_bytes)
{
    static const int on = fm_tune("FM_STEM_ROWS", 1);
    return on && k == 7 && stride == 2 && cout == 64 && wout == 112 && hout % 4 == 0 && 3 * plane_bytes < 0x7ff00000LL;
}
int stem_rows_stats_tiles(int imgs_per_group, int hout) { return imgs_per_group * (hout / 4); }

void launch_stem_rows(StemRowsParams p, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        set_max_dyn_lds(reinterpret_cast<const void*>(&stem_rows_kernel<6>), SR_LDS, "stem_rows_kernel<6>");
        set_max_dyn_lds(reinterpret_cast<const void*>(&stem_rows_kernel<9>), SR_LDS, "stem_rows_kernel<9>");
        attr_done = true;
    }
    const int ntiles = p.imgs * (p.Ho / 4);
    const int nblk = std::min(256, ntiles);
    if (p.sp == 9) hipLaunchKernelGGL((stem_rows_kernel<9>), dim3(nblk), dim3(512), SR_LDS, s, p);
    else hipLaunchKernelGGL((stem_rows_kernel<6>), dim3(nblk), dim3(512), SR_LDS, s, p);
}
