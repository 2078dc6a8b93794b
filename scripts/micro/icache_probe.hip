// Dev probe (GPU): what does a COLD instruction stream cost a workgroup?  (round 6: the small launches' kernels are 54-72 KB of
// straight-line code that every CU executes once per launch.)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
// Kernel<KB>: KB KiB of straight-line independent VALU instructions (8-byte v_fma_f32, eight rotating chains) executed TWICE in a
// rolled loop by every wave; s_memtime per pass per wave.  Pass 1 fetches the code cold (each launch is preceded by a different kernel
// that sweeps the instruction cache), pass 2 finds it wherever it still is.  Grids: 63 workgroups (B = 1: a CU and its
// instruction-cache neighbour mostly alone), 256, 1024 (two rounds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REPT8 \
    "v_fma_f32 v10, v10, v18, v19\n\tv_fma_f32 v11, v11, v18, v19\n\tv_fma_f32 v12, v12, v18, v19\n\tv_fma_f32 v13, v13, v18, v19\n\t" \
    "v_fma_f32 v14, v14, v18, v19\n\tv_fma_f32 v15, v15, v18, v19\n\tv_fma_f32 v16, v16, v18, v19\n\tv_fma_f32 v17, v17, v18, v19\n\t"

template <int KB>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int nrep, float seed) {
    unsigned long long t[3] = {0, 0, 0};
    float acc = seed;
    asm volatile("v_mov_b32 v18, 0x3f7fff00\n\tv_mov_b32 v19, 0x33000000\n\t"
                 "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\t"
                 "v_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0" ::"v"(acc)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
#pragma unroll 1
    for (int rep = 0; rep < nrep; ++rep) {
        t[rep] = __builtin_amdgcn_s_memtime();
        // KB KiB = KB * 128 instructions of 8 bytes = KB * 16 groups of eight
        asm volatile(".rept %0\n\t" REPT8 ".endr" ::"n"(KB * 16) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    }
    t[2] = __builtin_amdgcn_s_memtime();
    asm volatile("v_add_f32 %0, v10, v17" : "=v"(acc)::"v10", "v17");
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = t[1] - t[0];
        o[1] = t[2] - t[1];
        o[2] = (unsigned long long)(acc != 12345.f);
    }
}
// a different, large kernel in between: sweeps the instruction caches (128 KiB of other code on every CU)
__global__ __launch_bounds__(256) void sweep(float* o) {
    float acc = (float)threadIdx.x;
    asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v18, 0x3f7fff00\n\tv_mov_b32 v19, 0x33000000" ::"v"(acc) : "v10", "v18", "v19");
    asm volatile(".rept 16384\n\tv_fma_f32 v10, v10, v18, v19\n\t.endr" ::: "v10");
    asm volatile("v_mov_b32 %0, v10" : "=v"(acc)::"v10");
    if (acc == 12345.f) o[0] = acc;
}

template <int KB>
void run(int nwg) {
    unsigned long long* d;
    float* f;
    hipMalloc(&d, sizeof(unsigned long long) * nwg * 16);
    hipMalloc(&f, 4);
    double c1 = 0, c2 = 0;
    const int trials = 5;
    for (int tr = 0; tr < trials; ++tr) {
        hipLaunchKernelGGL(sweep, dim3(1024), dim3(256), 0, 0, f);
        hipLaunchKernelGGL(probe<KB>, dim3(nwg), dim3(256), 0, 0, d, 2, 1.0f + tr);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)nwg * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < nwg * 4; ++w) { a += (double)h[(size_t)w * 4]; b += (double)h[(size_t)w * 4 + 1]; }
        c1 += a / (nwg * 4);
        c2 += b / (nwg * 4);
    }
    const double n = KB * 128.0;
    printf("code %4d KiB  grid %5d : pass 1 %8.0f cycles (%.2f per instruction)   pass 2 %8.0f (%.2f)\n", KB, nwg, c1 / trials, c1 / trials / n,
           c2 / trials, c2 / trials / n);
    hipFree(d);
    hipFree(f);
}

int main() {
    for (int nwg : {63, 256, 1024}) {
        run<4>(nwg);
        run<16>(nwg);
        run<32>(nwg);
        run<48>(nwg);
        run<64>(nwg);
        run<96>(nwg);
    }
    return 0;
}
