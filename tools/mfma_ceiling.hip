// Micro-benchmark: what the fp32 MFMA pipe sustains for the igemm inner-loop shape
// (64 x v_mfma_f32_16x16x4_f32 per 16-k step per wave) with and without its LDS fragment
// reads and the per-step workgroup barrier.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: registers only; 1: + 8 ds_read_b128 per step; 2: + s_barrier per step; 3: + 4 LDS-DMA/wave/step
__global__ __launch_bounds__(256, 2) void k(float* out, int steps, const float* src = nullptr, size_t srcn = 0)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) { unsigned h = (i + blockIdx.x * 16384) * 2654435761u; h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13; smem[i] = ((float)(h & 0xffffff) / 8388608.0f) - 1.0f; }
    __syncthreads();
    f32x4 acc[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0, 0, 0, 0};
    f32x4 a[4], b[4];
    for (int r = 0; r < 4; ++r) { a[r] = *(f32x4*)(smem + (lane * 4 + r * 256) % 16000); b[r] = *(f32x4*)(smem + (lane * 4 + r * 256 + 1024) % 16000); }
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 3) {
            // LDS-DMA refill of the ring slot two steps behind, 4 x 1 KiB per wave, streaming a big buffer
            typedef __attribute__((address_space(3))) void lds_void;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const size_t base = (((size_t)blockIdx.x * 977 + (size_t)s * 131) * 4096 + (size_t)wave * 1024) % (srcn - 65536);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds(src + base + q * 256 + (MODE == 4 ? (lane >> 2) * 2048 + (lane & 3) * 4 : lane * 4),
                                                 (lds_void*)(smem + (((s + 2) & 3) * 4096) + wave * 1024 + q * 256), 16, 0, 0);
        } else
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
        if (MODE >= 1) {
            const float* A = smem + ((s & 3) * 4096) + (lane & 15) * 16 + ((lane >> 4) << 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = *(const f32x4*)(A + r * 256); b[r] = *(const f32x4*)(A + 2048 + r * 256); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
    }
    f32x4 t = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t += acc[r][c];
    out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}

__global__ void fill_random(float* x, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13); h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        x[i] = ((float)(h & 0xffffff) / 8388608.0f) - 1.0f;      // uniform [-1, 1)
    }
}
// producer/consumer split: waves 0-3 only read fragments + MFMA, waves 4.. issue all the LDS-DMA
template <int NL>
__global__ __launch_bounds__(256 + 64 * NL, 2) void ks(float* out, int steps, const float* src, size_t srcn)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 256 + 64 * NL) { unsigned h = (i + blockIdx.x * 16384) * 2654435761u; h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13; smem[i] = ((float)(h & 0xffffff) / 8388608.0f) - 1.0f; }
    __syncthreads();
    if (wave >= 4) {                      // loader: 16 KiB per step = 16 DMA, split over NL waves
        typedef __attribute__((address_space(3))) void lds_void;
        const int lw = wave - 4;
        for (int s = 0; s < steps; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 16 / NL) : "memory");
            __builtin_amdgcn_s_barrier();
            const size_t base = (((size_t)blockIdx.x * 977 + (size_t)s * 131) * 4096) % (srcn - 65536);
#pragma unroll
            for (int q = 0; q < 16 / NL; ++q) {
                const int g = lw * (16 / NL) + q;
                __builtin_amdgcn_global_load_lds(src + base + g * 256 + lane * 4,
                                                 (lds_void*)(smem + (((s + 2) & 3) * 4096) + g * 256), 16, 0, 0);
            }
        }
        return;
    }
    f32x4 acc[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0, 0, 0, 0};
    f32x4 a[4], b[4];
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_s_barrier();
        const float* A = smem + ((s & 3) * 4096) + (lane & 15) * 16 + ((lane >> 4) << 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = *(const f32x4*)(A + r * 256); b[r] = *(const f32x4*)(A + 2048 + r * 256); }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], b[c][j], acc[r][c], 0, 0, 0);
    }
    f32x4 t = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t += acc[r][c];
    out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}

template <int NL> void runs(const char* name, int blocks, size_t foot = 0)
{
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    static float* src = nullptr; const size_t srcn = (size_t)512 << 20; if (!src) { hipMalloc(&src, srcn * 4); hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, srcn); hipDeviceSynchronize(); }
    const int steps = 20000;
    hipFuncSetAttribute((const void*)&ks<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const size_t use = foot ? foot : srcn;
    hipLaunchKernelGGL(ks<NL>, dim3(blocks), dim3(256 + 64 * NL), 65536, 0, out, 100, src, use);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ks<NL>, dim3(blocks), dim3(256 + 64 * NL), 65536, 0, out, steps, src, use);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * steps * 64 * 2048.0;
    printf("%-44s blocks %4d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

template <int MODE> void run(const char* name, int blocks)
{
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    static float* src = nullptr; const size_t srcn = (size_t)512 << 20; if (!src) { hipMalloc(&src, srcn * 4); hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, srcn); hipDeviceSynchronize(); }
    const int steps = 20000;
    hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, out, 100, src, srcn);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, out, steps, src, srcn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * steps * 64 * 2048.0;
    printf("%-44s blocks %4d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    for (int blocks : {256, 512}) {
        run<0>("mfma only (operands in registers)", blocks);
        run<1>("mfma + 8 ds_read_b128 / step", blocks);
        run<2>("mfma + ds_read + s_barrier / step", blocks);
        run<3>("  + 4 LDS-DMA (1 KiB contiguous each)/wave/step", blocks);
        run<4>("  + 4 LDS-DMA (16 rows x 64 B gather)/wave/step", blocks);
        runs<1>("4 consumer waves + 1 loader wave (16 DMA/step)", blocks);
        runs<2>("4 consumer waves + 2 loader waves (8 DMA/step)", blocks);
        runs<4>("4 consumer waves + 4 loader waves (4 DMA/step)", blocks);
        runs<2>("4 cons + 2 loaders, 16 MiB footprint (L2/MALL)", blocks, (size_t)4 << 20);
        runs<2>("4 cons + 2 loaders, 1 MiB footprint (L2)", blocks, (size_t)1 << 18);
    }
    return 0;
}
