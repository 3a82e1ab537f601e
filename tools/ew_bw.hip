// HBM bandwidth of the elementwise access patterns used by elementwise.hip / effnet.hip (timing only).
// hipcc --offload-arch=gfx950 -O3 tools/ew_bw.hip -o build/ew_bw && build/ew_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_gridstride(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) y[i] = x[i] * 2.f;
}
__global__ void k_flat(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) y[i] = x[i] * 2.f;
}
template <int U>
__global__ void k_unroll(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4)
{
    const long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = base + (long)u * blockDim.x; if (i < n4) v[u] = x[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = base + (long)u * blockDim.x; if (i < n4) y[i] = v[u] * 2.f; }
}
template <int U>
__global__ void k_gs_unroll(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4)
{
    const long stride = (long)gridDim.x * blockDim.x * U;
    for (long b = (long)blockIdx.x * blockDim.x * U + threadIdx.x; b < n4; b += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b + (long)u * blockDim.x; if (i < n4) v[u] = x[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b + (long)u * blockDim.x; if (i < n4) y[i] = v[u] * 2.f; }
    }
}
template <typename F>
float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
    const long n4 = 154L * 1024 * 1024;   // 2.46 GB per tensor
    f32x4 *x, *y; hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16); hipMemset(x, 0, n4 * 16);
    auto rep = [&](const char* name, float ms) { printf("%-28s %8.1f us  %.2f TB/s\n", name, ms * 1e3, 2.0 * n4 * 16 / ms / 1e9); };
    for (int nb : {1024, 2048, 4096, 8192, 16384})
        { char s[64]; sprintf(s, "gridstride %d blocks", nb); rep(s, timeit([&] { hipLaunchKernelGGL(k_gridstride, dim3(nb), dim3(256), 0, 0, x, y, n4); })); }
    rep("flat (1 elem/thread)", timeit([&] { hipLaunchKernelGGL(k_flat, dim3((n4 + 255) / 256), dim3(256), 0, 0, x, y, n4); }));
    rep("unroll 4 (flat)", timeit([&] { hipLaunchKernelGGL(k_unroll<4>, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, x, y, n4); }));
    rep("unroll 8 (flat)", timeit([&] { hipLaunchKernelGGL(k_unroll<8>, dim3((n4 + 2047) / 2048), dim3(256), 0, 0, x, y, n4); }));
    for (int nb : {2048, 4096, 8192})
        { char s[64]; sprintf(s, "gridstride unroll4 %d", nb); rep(s, timeit([&] { hipLaunchKernelGGL(k_gs_unroll<4>, dim3(nb), dim3(256), 0, 0, x, y, n4); })); }
    return 0;
}
