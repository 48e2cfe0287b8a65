// lpips_head.hip -- the per-layer head of the LPIPS distance (third_parties/lpips/lpips.py:112-134 with lpips/__init__.py:13-15):
//   n0 = f0 / (||f0||_c + 1e-10);  d = (n0 - n1)^2;  lin (1x1 convolution to one channel, weights w);  spatial mean
// as ONE pass over the prediction's feature map instead of ~9 elementwise / reduction launches (and ~12 in the backward) over 16-66 MB
// tensors per layer.  n1 = the already unit-normalised features of the reference image (no gradient).  HBM-streaming: one thread per
// pixel, channels strided by H*W (NCHW), so every wave instruction reads 256 contiguous bytes of one channel plane.
#include <hip/hip_runtime.h>

#include "d3h_common.h"

namespace {

// out[b] += (1 / HW) * sum_p sum_c w_c (f0[b,c,p] / (r_p + eps) - n1[b,c,p])^2
__global__ __launch_bounds__(256) void lpips_head_fwd_kernel(const float* __restrict__ f0, const float* __restrict__ n1,
                                                             const float* __restrict__ w, int C, int HW, float inv_hw, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (p < HW) {
        const float* x = f0 + (size_t)b * C * HW + p;
        const float* y = n1 + (size_t)b * C * HW + p;
        float ss = 0.f;
        for (int c = 0; c < C; ++c) { const float v = x[(size_t)c * HW]; ss += v * v; }
        const float den = sqrtf(ss) + 1e-10f;
        for (int c = 0; c < C; ++c) {
            const float e = x[(size_t)c * HW] / den - y[(size_t)c * HW];
            term += w[c] * (e * e);
        }
    }
    // wave sum, then one atomic per wave
    for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if ((threadIdx.x & 63) == 0 && term != 0.f) atomicAdd(out + b, term * inv_hw);
}

// d f0[b,k,p] = g[b] / HW * ( q_k / den - x_k * (sum_c q_c x_c) / (r * den^2) ),  q_c = 2 w_c e_c  -- the chain through
// x / (sqrt(sum x^2) + eps) exactly as autograd forms it (r = 0 gives the same 0 * inf = NaN)
__global__ __launch_bounds__(256) void lpips_head_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ n1,
                                                             const float* __restrict__ w, int C, int HW, float inv_hw,
                                                             const float* __restrict__ g, float* __restrict__ d_f0) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* x = f0 + (size_t)b * C * HW + p;
    const float* y = n1 + (size_t)b * C * HW + p;
    float* dx = d_f0 + (size_t)b * C * HW + p;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = x[(size_t)c * HW]; ss += v * v; }
    const float r = sqrtf(ss);
    const float den = r + 1e-10f;
    const float up = g[b] * inv_hw;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = x[(size_t)c * HW];
        const float e = v / den - y[(size_t)c * HW];
        s += (2.f * w[c] * e) * v;
    }
    // d(den)/d(x_k) = x_k / r  (sqrt'(ss) * 2 x_k); the quotient rule gives  -x_c / den^2  per channel, summed with q_c
    const float coef = (-s * up / (den * den)) / r;          // r == 0: -0 / 0 = NaN, as torch.autograd
    for (int c = 0; c < C; ++c) {
        const float v = x[(size_t)c * HW];
        const float e = v / den - y[(size_t)c * HW];
        dx[(size_t)c * HW] = (2.f * w[c] * e) * up / den + coef * v;
    }
}

// ---- channels-last feature maps ([B][HW][C], what the convolutions produce for an NHWC-strided input): 16 lanes per pixel, each
// lane float4 loads at channel 4 (j + 16 i) -- 256 contiguous bytes per pixel and step; the channel sums are 4-step shuffle reductions
// inside the 16-lane group.  C must be a multiple of 64.
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

__global__ __launch_bounds__(256) void lpips_head_fwd_nhwc_kernel(const float* __restrict__ f0, const float* __restrict__ n1,
                                                                  const float* __restrict__ w, int C, int HW, float inv_hw, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int j = threadIdx.x & 15;
    const int nsteps = C >> 6;
    float term = 0.f;
    for (long long p0 = (long long)blockIdx.x * 16; p0 < HW; p0 += (long long)gridDim.x * 16) {      // wave-uniform trip count: every lane
        const long long pp = p0 + (threadIdx.x >> 4);                                                  // takes part in the shuffles
        const bool valid = pp < HW;
        const long long p = valid ? pp : (long long)HW - 1;
        const float* x = f0 + ((size_t)b * HW + p) * C + 4 * j;
        const float* y = n1 + ((size_t)b * HW + p) * C + 4 * j;
        float ss = 0.f;
        for (int i = 0; i < nsteps; ++i) {
            const float4 v = *(const float4*)(x + 64 * i);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        const float den = sqrtf(group16_sum(ss)) + 1e-10f;
        for (int i = 0; i < nsteps; ++i) {
            const float4 v = *(const float4*)(x + 64 * i);
            const float4 r = *(const float4*)(y + 64 * i);
            const float4 ww = *(const float4*)(w + 4 * j + 64 * i);
            const float e0 = v.x / den - r.x, e1 = v.y / den - r.y, e2 = v.z / den - r.z, e3 = v.w / den - r.w;
            if (valid) term += (ww.x * (e0 * e0) + ww.y * (e1 * e1)) + (ww.z * (e2 * e2) + ww.w * (e3 * e3));
        }
    }
    for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if ((threadIdx.x & 63) == 0 && term != 0.f) atomicAdd(out + b, term * inv_hw);
}

__global__ __launch_bounds__(256) void lpips_head_bwd_nhwc_kernel(const float* __restrict__ f0, const float* __restrict__ n1,
                                                                  const float* __restrict__ w, int C, int HW, float inv_hw,
                                                                  const float* __restrict__ g, float* __restrict__ d_f0) {
    const int b = blockIdx.y;
    const int j = threadIdx.x & 15;
    const int nsteps = C >> 6;
    const float up = g[b] * inv_hw;
    for (long long p0 = (long long)blockIdx.x * 16; p0 < HW; p0 += (long long)gridDim.x * 16) {
        const long long pp = p0 + (threadIdx.x >> 4);
        const bool valid = pp < HW;
        const long long p = valid ? pp : (long long)HW - 1;
        const size_t base = ((size_t)b * HW + p) * C + 4 * j;
        const float* x = f0 + base;
        const float* y = n1 + base;
        float ss = 0.f;
        for (int i = 0; i < nsteps; ++i) {
            const float4 v = *(const float4*)(x + 64 * i);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        const float r_ = sqrtf(group16_sum(ss));
        const float den = r_ + 1e-10f;
        float s = 0.f;
        for (int i = 0; i < nsteps; ++i) {
            const float4 v = *(const float4*)(x + 64 * i);
            const float4 r = *(const float4*)(y + 64 * i);
            const float4 ww = *(const float4*)(w + 4 * j + 64 * i);
            s += (2.f * ww.x * (v.x / den - r.x)) * v.x + (2.f * ww.y * (v.y / den - r.y)) * v.y +
                 (2.f * ww.z * (v.z / den - r.z)) * v.z + (2.f * ww.w * (v.w / den - r.w)) * v.w;
        }
        s = group16_sum(s);
        const float coef = (-s * up / (den * den)) / r_;          // r == 0: NaN, as torch.autograd
        for (int i = 0; i < nsteps; ++i) {
            const float4 v = *(const float4*)(x + 64 * i);
            const float4 r = *(const float4*)(y + 64 * i);
            const float4 ww = *(const float4*)(w + 4 * j + 64 * i);
            float4 o;
            o.x = (2.f * ww.x * (v.x / den - r.x)) * up / den + coef * v.x;
            o.y = (2.f * ww.y * (v.y / den - r.y)) * up / den + coef * v.y;
            o.z = (2.f * ww.z * (v.z / den - r.z)) * up / den + coef * v.z;
            o.w = (2.f * ww.w * (v.w / den - r.w)) * up / den + coef * v.w;
            if (valid) *(float4*)(d_f0 + base + 64 * i) = o;
        }
    }
}

}  // namespace

// f0, n1: [B][C][HW] (NCHW feature maps, n1 unit-normalised along C); w [C]; out [B] is zeroed here and receives the layer's distance
// channels_last != 0: both maps are [B][HW][C] instead (C a multiple of 64)
extern "C" int d3h_lpips_head_fwd(const float* f0, const float* n1, const float* w, int B, int C, int HW, int channels_last, float* out,
                                  void* stream) {
    if (B < 0 || C <= 0 || HW <= 0 || !out || (B > 0 && (!f0 || !n1 || !w)) || (channels_last && (C & 63))) return D3H_ERR_ARG;
    if (B == 0) return D3H_OK;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(out, 0, (size_t)B * sizeof(float), s);
    if (channels_last) {
        int gx = d3h_cdiv(HW, 16) < 2048 ? d3h_cdiv(HW, 16) : 2048;
        hipLaunchKernelGGL(lpips_head_fwd_nhwc_kernel, dim3(gx, B), dim3(256), 0, s, f0, n1, w, C, HW, 1.0f / (float)HW, out);
    } else
        hipLaunchKernelGGL(lpips_head_fwd_kernel, dim3(d3h_cdiv(HW, 256), B), dim3(256), 0, s, f0, n1, w, C, HW, 1.0f / (float)HW, out);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// g [B]: upstream gradient of out; d_f0 [B][C][HW] is overwritten
extern "C" int d3h_lpips_head_bwd(const float* f0, const float* n1, const float* w, int B, int C, int HW, int channels_last, const float* g,
                                  float* d_f0, void* stream) {
    if (B < 0 || C <= 0 || HW <= 0 || (B > 0 && (!f0 || !n1 || !w || !g || !d_f0)) || (channels_last && (C & 63))) return D3H_ERR_ARG;
    if (B == 0) return D3H_OK;
    if (channels_last) {
        int gx = d3h_cdiv(HW, 16) < 2048 ? d3h_cdiv(HW, 16) : 2048;
        hipLaunchKernelGGL(lpips_head_bwd_nhwc_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, f0, n1, w, C, HW, 1.0f / (float)HW, g, d_f0);
    } else
        hipLaunchKernelGGL(lpips_head_bwd_kernel, dim3(d3h_cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, f0, n1, w, C, HW, 1.0f / (float)HW, g,
                           d_f0);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
