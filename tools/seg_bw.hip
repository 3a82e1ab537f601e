// Microbenchmark: does the access GRANULE of a wave instruction matter for streaming bandwidth on MI355X?
// Every wave instruction moves 1 KB (64 lanes x 16 B) as 16 rows x 64-B segments (the pointwise conv kernels' fragment
// pattern: lane (li, lg) -> row li, bytes 16 lg of a row of ROWB bytes) or as one contiguous 1-KB run.
// usage: seg_bw   (prints GB/s for read / write / copy in both patterns and several row sizes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0 read-only (sum kept), 1 write-only, 2 copy.  SEG: 1 = 16 rows x 64 B per instruction, 0 = contiguous.
template <int MODE, int SEG>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nrows, int rowchunks, u32x4* sink)
{
    // a wave owns groups of 16 rows; per group it walks the row in steps of 4 chunks (64 B)
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    u32x4 acc = {0, 0, 0, 0};
    const size_t ngroups = nrows / 16;
    for (size_t g = wave; g < ngroups; g += nw) {
        for (int c = 0; c < rowchunks; c += 4) {
            const size_t iseg = (g * 16 + li) * rowchunks + c + lg;             // row li, chunk c+lg
            const size_t icon = g * 16 * rowchunks + (size_t)(c / 4) * 64 + lane;  // the same 1 KB of the group, contiguous per instruction
            // SEG: 0 both contiguous, 1 both segmented, 2 segmented load + contiguous store, 3 contiguous load + segmented store
            const size_t il = (SEG == 1 || SEG == 2) ? iseg : icon, is = (SEG == 1 || SEG == 3) ? iseg : icon;
            u32x4 v = {1, 2, 3, 4};
            if (MODE != 1) v = src[il];
            if (MODE == 0) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
            else dst[is] = v;
        }
    }
    if (MODE == 0 && acc.x == 0x12345678u) *sink = acc;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;
    u32x4 *a, *b, *sink;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int rowbytes[] = {64, 192, 288, 480, 2304};
    for (int rb : rowbytes) {
        const int rowchunks = rb / 16;
        if (rowchunks % 4) continue;
        const size_t nrows = bytes / rb / 16 * 16;
        for (int seg = 0; seg < 4; ++seg)
            for (int mode = 0; mode < 3; ++mode) {
                if (seg >= 2 && mode != 2) continue;
                auto launch = [&]() {
                    dim3 grid(8192), blk(256);
#define L(M, S) hipLaunchKernelGGL((k<M, S>), grid, blk, 0, 0, a, b, nrows, rowchunks, sink)
                    if (seg == 1) { if (mode == 0) L(0, 1); else if (mode == 1) L(1, 1); else L(2, 1); }
                    else if (seg == 0) { if (mode == 0) L(0, 0); else if (mode == 1) L(1, 0); else L(2, 0); }
                    else if (seg == 2) L(2, 2);
                    else L(2, 3);
                };
                launch(); CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                const double moved = (double)nrows * rb * (mode == 2 ? 2 : 1);
                printf("row %4d B  %s  %-5s  %7.0f GB/s\n", rb, seg == 1 ? "16 rows x 64 B" : seg == 0 ? "contiguous 1 KB" : seg == 2 ? "seg load/contig store" : "contig load/seg store", mode == 0 ? "read" : mode == 1 ? "write" : "copy", moved / ms / 1e6);
            }
    }
    return 0;
}
