// Dev probe (GPU): how many workgroups of a given shape does the dispatcher keep resident per CU?
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/occ_probe.hip -o /tmp/occ_probe && /tmp/occ_probe
// Each workgroup records HW_ID / XCC_ID and its start / end s_memrealtime and spins ~30 us; the host counts the workgroups whose
// lifetimes overlap on one CU.  Shapes: threads per workgroup x VGPRs (forced by touching the highest register) x LDS bytes x
// scratch (a dynamically indexed private array).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <tuple>
#include <algorithm>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

template <int THREADS, int WAVES_EU, int VG, int LDS, int SCR>
__global__ __launch_bounds__(THREADS, WAVES_EU) void probe(Rec* out, int spin, int idx) {
    __shared__ unsigned char lds[LDS];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (VG >= 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if (VG >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    int acc = 0;
    if (SCR) {
        volatile int priv[SCR / 4 > 0 ? SCR / 4 : 1];
        for (int i = 0; i < SCR / 4; ++i) priv[i] = i + idx;
        acc = priv[(idx + threadIdx.x) % (SCR / 4 > 0 ? SCR / 4 : 1)];
    }
    lds[threadIdx.x] = (unsigned char)acc;
    __syncthreads();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) acc += lds[(threadIdx.x + acc) % LDS];
    if (threadIdx.x == 0) {
        Rec r;
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r.hw ^= (acc & 0);
        out[blockIdx.x] = r;
    }
}

template <int THREADS, int WAVES_EU, int VG, int LDS, int SCR>
void run(const char* name, int nwg) {
    Rec* d;
    hipMalloc(&d, sizeof(Rec) * nwg);
    hipMemset(d, 0, sizeof(Rec) * nwg);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<THREADS, WAVES_EU, VG, LDS, SCR>), dim3(nwg), dim3(THREADS), 0, 0, d, 3000, 1);
    hipDeviceSynchronize();
    std::vector<Rec> h(nwg);
    hipMemcpy(h.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)probe<THREADS, WAVES_EU, VG, LDS, SCR>);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
    for (auto& r : h) cu[{r.xcc & 15, (r.hw >> 13) & 7, (r.hw >> 12) & 1, (r.hw >> 8) & 15}].push_back({r.t0, r.t1});
    std::map<int, int> hist;
    for (auto& kv : cu) {
        int best = 0;
        for (auto& a : kv.second) {
            int n = 0;
            for (auto& b : kv.second) n += (b.first <= a.first && a.first < b.second);
            best = std::max(best, n);
        }
        hist[best]++;
    }
    printf("%-44s regs %3d scratch %3zu B lds %6zu B: %zu CUs; max co-resident workgroups per CU:", name, fa.numRegs, fa.localSizeBytes,
           fa.sharedSizeBytes, cu.size());
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
    hipFree(d);
}

int main() {
    run<384, 3, 168, 49680, 0>("6 waves, 168 regs, 49.7 KB", 1024);
    run<384, 3, 168, 49680, 64>("6 waves, 168 regs, 49.7 KB, scratch", 1024);
    run<384, 3, 128, 49680, 0>("6 waves, 128 regs, 49.7 KB", 1024);
    run<384, 3, 168, 16384, 0>("6 waves, 168 regs, 16 KB", 1024);
    run<384, 3, 0, 16384, 0>("6 waves, few regs, 16 KB", 2048);
    run<256, 3, 168, 16384, 0>("4 waves, 168 regs, 16 KB (k_flash's shape)", 2048);
    run<256, 3, 168, 16384, 64>("4 waves, 168 regs, 16 KB, scratch", 2048);
    run<256, 3, 168, 49680, 0>("4 waves, 168 regs, 49.7 KB", 2048);
    run<256, 2, 232, 49680, 0>("4 waves, 168+ regs (2/EU), 49.7 KB", 2048);
    run<768, 3, 168, 98304, 0>("12 waves, 168 regs, 96 KB", 512);
    run<768, 3, 168, 98304, 64>("12 waves, 168 regs, 96 KB, scratch", 512);
    return 0;
}
