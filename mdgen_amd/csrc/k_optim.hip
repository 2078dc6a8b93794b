// k_optim.hip -- the optimiser side of the training step, as flat-buffer fp32 kernels:
//   gradient-norm clipping  (train.py:56 `gradient_clip_val=args.grad_clip` -> torch.nn.utils.clip_grad_norm_, 2-norm)
//   Adam / AdamW            (wrapper.py:167-172 `configure_optimizers`: torch.optim.Adam(lr) or AdamW)
//   EMA of the weights      (ema.py:41-58: stored -= (stored - param) * (1 - decay))
// All 34.15 M parameters live in ONE contiguous fp32 buffer (and so do grads and the two moments), so each step is one
// HBM-bound launch over 137 MB per stream instead of ~500 per-tensor launches: Adam reads p, g, m, v and writes p, m, v
// (956 MB per step, ~0.2 ms at 5 TB/s).  The clip coefficient never visits the host: the norm kernel leaves
// sum(g^2) in device memory and the Adam kernel derives the coefficient from it.
#include "kernels.h"

namespace mdg {

// partial[b] = sum over the block's slice of (g * scale)^2 ; deterministic (fixed slice per block, tree inside) and
// accumulated in fp64 (the kernel is HBM-bound; an fp32 sum of 3e7 squares would carry ~1e-5 relative error into the
// clip coefficient)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ g, long n, float scale,
                                                       double* __restrict__ partial) {
    __shared__ double red[4];
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double s = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = g[i] * scale;
        s += (double)v * (double)v;
    }
    s = wave_sum_f64(s);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_sumsq_final(const double* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_sum_f64(s);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

struct AdamParams {
    float* p; const float* g; float* m; float* v;
    long n;
    float lr, beta1, beta2, eps, weight_decay;
    int adamw;
    float bc1, bc2_sqrt;          // 1 - beta1^t, sqrt(1 - beta2^t)
    float grad_scale;             // e.g. 1 / world_size (DDP averages, train.py + Lightning DDP)
    const float* sumsq;           // device: sum((g * grad_scale)^2), or null = no clipping
    float max_norm;               // clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))
};

__global__ __launch_bounds__(256) void k_adam(const AdamParams a) {
    float coef = a.grad_scale;
    if (a.sumsq) {
        const float norm = sqrtf(a.sumsq[0]);
        const float c = a.max_norm / (norm + 1e-6f);
        coef *= c < 1.0f ? c : 1.0f;
    }
    const float step = a.lr / a.bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
        float p = a.p[i];
        const float g = a.g[i] * coef;
        float m = a.m[i], v = a.v[i];
        if (a.adamw) p *= 1.0f - a.lr * a.weight_decay;                  // torch.optim.AdamW: decoupled decay first
        m = m + (g - m) * (1.0f - a.beta1);                              // exp_avg.lerp_(grad, 1 - beta1)
        v = v * a.beta2 + (1.0f - a.beta2) * g * g;                      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        p = p - step * (m / denom);                                      // param.addcdiv_(exp_avg, denom, -step_size)
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_ema(float* __restrict__ ema, const float* __restrict__ p, long n, float one_minus_decay) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float s = ema[i];
        ema[i] = s - (s - p[i]) * one_minus_decay;                        // ema.py:49-51
    }
}

void launch_sumsq(const float* g, long n, float scale, float* partial, int nblocks, float* out, hipStream_t s) {
    double* pd = reinterpret_cast<double*>(partial);   // scratch is used as fp64 partials: nblocks <= floats / 2
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nblocks), dim3(256), 0, s, g, n, scale, pd);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, pd, nblocks, out);
}
void launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int adamw, float bc1, float bc2_sqrt, float grad_scale, const float* sumsq,
                 float max_norm, hipStream_t s) {
    AdamParams a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, adamw, bc1, bc2_sqrt, grad_scale, sumsq, max_norm};
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, a);
}
void launch_ema(float* ema, const float* p, long n, float one_minus_decay, hipStream_t s) {
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_ema, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, ema, p, n, one_minus_decay);
}

}  // namespace mdg
