// optim.hip -- the optimiser step of the training loop as ONE launch (gfx950).
//
// Replaces, per iteration of train.py:759-788: optimizer.step() + optimizer_mesh.step() (torch.optim.Adam, eps 1e-8, no weight decay:
// train.py:593-620), the `encoder.params.grad /= 8.0` scaling in front of them (train.py:747-748) and geometry.clamp_deform() behind them
// (hmsdf.py:398-405: deform to [-1, 1], msdf to [-2, 2]).  The reference issues ~8 multi-tensor Adam launches plus the scale and clamp
// kernels from Python on the launch-bound tail of the step, right where the GPU has run dry; here every tensor of both optimisers is one
// row of a by-value table and a single kernel walks them (blockIdx.y = tensor).  HBM-bound: 28 B per element (p, g, m, v in; p, m, v out).
//
// Update (torch/optim/adam.py, amsgrad = False, maximize = False, weight_decay = 0), per element and in this order of operations:
//   g' = g * gscale;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//   p = clamp(p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), lo, hi)
#include "d3h_common.h"

namespace {

constexpr int ADAM_MAX = 32;

struct AdamRow {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    float step_size, inv_sqrt_bc2, gscale, lo, hi;
};

struct AdamBatch {
    AdamRow t[ADAM_MAX];
    float beta1, beta2, eps;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamBatch b) {
    const AdamRow r = b.t[blockIdx.y];
    const float b1 = b.beta1, b2 = b.beta2, eps = b.eps;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < r.n; i += (long long)gridDim.x * 256) {
        const float g = r.g[i] * r.gscale;
        const float m = b1 * r.m[i] + (1.f - b1) * g;
        const float v = b2 * r.v[i] + (1.f - b2) * g * g;
        const float denom = sqrtf(v) * r.inv_sqrt_bc2 + eps;
        float p = r.p[i] - r.step_size * (m / denom);
        p = (p < r.lo) ? r.lo : ((p > r.hi) ? r.hi : p);        // torch.clamp semantics: NaN propagates (fminf / fmaxf would return the bound)
        r.m[i] = m;
        r.v[i] = v;
        r.p[i] = p;
    }
}

}  // namespace

// One Adam step over `nt` tensors.  p / g / m / v: HOST arrays of nt device pointers (parameter, gradient, first and second moment,
// each n[i] floats; m and v are updated in place).  n, lr, step, gscale, lo, hi: HOST arrays of nt entries -- element count, learning
// rate, 1-based step count of the tensor's optimiser state, gradient scale, clamp range applied to the updated parameter
// (-inf / +inf = none).
extern "C" int d3h_adam_multi(void* const* p, const void* const* g, void* const* m, void* const* v, const int64_t* n, const float* lr,
                              const int64_t* step, const float* gscale, const float* lo, const float* hi, int nt, float beta1, float beta2,
                              float eps, void* stream) {
    if (nt < 0 || (nt > 0 && (!p || !g || !m || !v || !n || !lr || !step))) return D3H_ERR_ARG;
    for (int base = 0; base < nt; base += ADAM_MAX) {
        AdamBatch b;
        const int cnt = nt - base < ADAM_MAX ? nt - base : ADAM_MAX;
        long long nmax = 0;
        for (int k = 0; k < cnt; ++k) {
            const int i = base + k;
            if (n[i] < 0 || step[i] < 1 || (n[i] > 0 && (!p[i] || !g[i] || !m[i] || !v[i]))) return D3H_ERR_ARG;
            AdamRow& r = b.t[k];
            r.p = (float*)p[i]; r.g = (const float*)g[i]; r.m = (float*)m[i]; r.v = (float*)v[i];
            r.n = n[i];
            const double bc1 = 1.0 - pow((double)beta1, (double)step[i]);
            const double bc2 = 1.0 - pow((double)beta2, (double)step[i]);
            r.step_size = (float)((double)lr[i] / bc1);
            r.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
            r.gscale = gscale ? gscale[i] : 1.f;
            r.lo = lo ? lo[i] : -INFINITY;
            r.hi = hi ? hi[i] : INFINITY;
            if (r.n > nmax) nmax = r.n;
        }
        b.beta1 = beta1; b.beta2 = beta2; b.eps = eps;
        if (nmax == 0) continue;
        int gx = (int)((nmax + 256 * 8 - 1) / (256 * 8));          // ~8 elements per thread for the largest tensor
        if (gx < 1) gx = 1;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, b);
        D3H_LAUNCH_CHECK();
    }
    return D3H_OK;
}
