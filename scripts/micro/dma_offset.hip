// micro-test: does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global
// source?  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 scripts/micro/dma_offset.hip -o /tmp/dma_offset && /tmp/dma_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) unsigned smem[4096];   // 16 KiB
    for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(base + 4096) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = smem[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *src, *out;
    hipMalloc(&src, 16384); hipMalloc(&out, 16384);
    hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    std::vector<unsigned> o(4096);
    hipMemcpy(o.data(), out, 16384, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4096; ++i)
        if (o[i] != 0xdeadbeefu) { printf("first written LDS dword %d (byte %d) holds source dword %u (byte %u); M0 pointed at byte 4096, inst offset 1024\n", i, 4 * i, o[i], 4 * o[i]); break; }
    return 0;
}
