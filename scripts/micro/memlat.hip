// Micro-benchmark (dev tool): how long does a wave take to fetch the rows an LN prologue needs?
// Each wave loads ROWS rows of 1536 B (24 x 8 B per lane per row-triple) from a 98 MB buffer, all loads issued
// before the first use, and reports s_memtime ticks.  Cases: 1 workgroup alone vs every CU at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int ROWS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, long nrows, unsigned long long* out, float* sink, int stride_rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * 4 + w) * stride_rows % (nrows - ROWS);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x2 v[ROWS][3];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const f32x2* xr = reinterpret_cast<const f32x2*>(x + (row0 + r) * 384);
#pragma unroll
        for (int i = 0; i < 3; ++i) v[r][i] = xr[lane + 64 * i];
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) s += v[r][i][0] + v[r][i][1];
    asm volatile("" :: "v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + w] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}
int main() {
    const long nrows = 64000;
    float* x; unsigned long long* out; float* sink;
    hipMalloc(&x, nrows * 384 * 4); hipMalloc(&out, 8192 * 8); hipMalloc(&sink, 4);
    hipMemset(x, 1, nrows * 384 * 4);
    float* junk; hipMalloc(&junk, 512l << 20);
    auto run = [&](int grid, const char* name) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(junk, rep, 512l << 20);   // evict x from L2 / MALL
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, x, nrows, out, sink, 61);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid * 4);
            hipMemcpy(h.data(), out, grid * 4 * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("%s rep %d: waves %d  ticks min %llu  median %llu  p90 %llu  max %llu\n", name, rep, grid * 4, h[0], h[h.size() / 2], h[h.size() * 9 / 10], h.back());
        }
    };
    run(1, "1 WG, 8 rows/wave (12 KB), cold");
    run(256, "256 WGs, 8 rows/wave, cold");
    run(512, "512 WGs, 8 rows/wave, cold");
    // warm (L2/MALL resident): repeat without eviction
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, x, nrows, out, sink, 61);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1024);
        hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("256 WGs warm rep %d: median %llu p90 %llu max %llu\n", rep, h[512], h[921], h.back());
    }
    return 0;
}
