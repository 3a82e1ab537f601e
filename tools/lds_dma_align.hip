// Does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept a global address that is only 8- or 4-byte aligned?
// (needed to know whether the 7x7 stem's K axis can be packed as 7 x (7*3 -> 24) floats over an NHWC3 row)
// build + run: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_align.hip -o build/lds_dma_align && build/lds_dma_align
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* dst, int shift_floats)
{
    __shared__ __attribute__((aligned(16))) float s[64 * 4];
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x;
    // lane l copies 16 B from src + shift + 6*l floats (24-B lane stride: 8-B aligned for even shift, else 4-B)
    __builtin_amdgcn_global_load_lds(src + shift_floats + 6 * lane, (lds_void*)s, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) dst[lane * 4 + i] = s[lane * 4 + i];
}
int main()
{
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 64 * 4 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int shift : {0, 2, 1, 3}) {
        hipMemset(o, 0, 64 * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        const hipError_t rc = hipDeviceSynchronize();
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) bad += r[l * 4 + i] != (float)(shift + 6 * l + i);
        printf("shift %d floats (address %% 16 = %d): rc=%d mismatches=%d first lane got %g %g %g %g\n", shift, (shift * 4) % 16, (int)rc, bad,
               r[4], r[5], r[6], r[7]);
    }
    return 0;
}
