// material_grads.hip -- the per-pixel smoothness buffers of shade() in one pass each way (gfx950).
//
// Replaces render/render.py:72-74,88-91,104-105 (shade): from the two texture-MLP lookups (at the surface point and at its jittered twin) and the
// interpolated normal with its jittered tap
//     kd        = all_tex[..., 0:3]
//     kd_grad   = |all_tex_jitter[..., 0:3] - kd|
//     ks_grad   = |all_tex_jitter[..., 3:6] - ks| * (0, 1, 1)
//     nrm_grad  = |nrm_jitter - gb_normal| * (mask * mask_tap)
// As torch ops that is 3 slices, 3 subtractions, 3 abs, 3 multiplies forward and, backward, 3 sgn / mul / neg chains plus four
// slice_backward (a zero fill of the 6-channel image and a copy each) and the gradient sums of the 6-channel tensors: ~25 launches over
// 4 x 1024^2 pixels per tick_split.  Here: one launch forward (24 or 36 B in, 36 or 48 B out per pixel), one backward that writes every
// input gradient completely (no zero fill, no accumulation).  HBM-bound streaming kernels, one thread per pixel.
// d|x|/dx = sgn(x) with sgn(0) = 0, as torch.abs.
#include "d3h_common.h"

namespace {

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void material_grads_fwd_kernel(const float* __restrict__ tex, const float* __restrict__ texj, const float* __restrict__ nrm,
                                                                 const float* __restrict__ nrmj, const float* __restrict__ mask,
                                                                 const float* __restrict__ mask_tap, int64_t n, float* __restrict__ kd,
                                                                 float* __restrict__ kdg, float* __restrict__ ksg, float* __restrict__ ng) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        float t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) t[c] = tex[6 * p + c];
        if (kd) {
#pragma unroll
            for (int c = 0; c < 3; ++c) kd[3 * p + c] = t[c];
        }
        if (texj) {
#pragma unroll
            for (int c = 0; c < 3; ++c) kdg[3 * p + c] = fabsf(texj[6 * p + c] - t[c]);
            ksg[3 * p + 0] = fabsf(texj[6 * p + 3] - t[3]) * 0.f;          // "omit o-component" (render.py:91); x * 0 keeps a NaN visible
            ksg[3 * p + 1] = fabsf(texj[6 * p + 4] - t[4]);
            ksg[3 * p + 2] = fabsf(texj[6 * p + 5] - t[5]);
        }
        if (nrm) {
            const float w = mask[p] * mask_tap[p];
#pragma unroll
            for (int c = 0; c < 3; ++c) ng[3 * p + c] = fabsf(nrmj[3 * p + c] - nrm[3 * p + c]) * w;
        }
    }
}

__global__ __launch_bounds__(256) void material_grads_bwd_kernel(const float* __restrict__ tex, const float* __restrict__ texj, const float* __restrict__ nrm,
                                                                 const float* __restrict__ nrmj, const float* __restrict__ mask,
                                                                 const float* __restrict__ mask_tap, int64_t n, const float* __restrict__ g_kd,
                                                                 const float* __restrict__ g_kdg, const float* __restrict__ g_ksg,
                                                                 const float* __restrict__ g_ng, float* __restrict__ d_tex, float* __restrict__ d_texj,
                                                                 float* __restrict__ d_nrm, float* __restrict__ d_nrmj) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        if (d_tex) {
            float dt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dj[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (g_kd) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dt[c] = g_kd[3 * p + c];
            }
            if (texj) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float* g = c < 3 ? g_kdg : g_ksg;
                    float gv = g ? g[3 * p + (c % 3)] : 0.f;
                    if (c == 3) gv *= 0.f;
                    const float v = gv * sgnf(texj[6 * p + c] - tex[6 * p + c]);
                    dj[c] = v;
                    dt[c] -= v;
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) d_tex[6 * p + c] = dt[c];
            if (d_texj) {
#pragma unroll
                for (int c = 0; c < 6; ++c) d_texj[6 * p + c] = dj[c];
            }
        }
        if (d_nrm) {
            const float w = mask[p] * mask_tap[p];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (g_ng ? g_ng[3 * p + c] : 0.f) * w * sgnf(nrmj[3 * p + c] - nrm[3 * p + c]);
                d_nrmj[3 * p + c] = v;
                d_nrm[3 * p + c] = -v;
            }
        }
    }
}

}  // namespace

// tex [n][6]; texj [n][6] or NULL (then kd_grad / ks_grad are not produced); nrm, nrmj [n][3] with mask, mask_tap [n], or all four NULL
// (then nrm_grad is not produced); kd (may be NULL), kd_grad, ks_grad, nrm_grad: [n][3], overwritten
extern "C" int d3h_material_grads_fwd(const float* tex, const float* texj, const float* nrm, const float* nrmj, const float* mask, const float* mask_tap,
                                      int64_t n, float* kd, float* kd_grad, float* ks_grad, float* nrm_grad, void* stream) {
    if (n < 0) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    if (!tex || (texj && (!kd_grad || !ks_grad)) || (nrm && (!nrmj || !mask || !mask_tap || !nrm_grad))) return D3H_ERR_ARG;
    hipLaunchKernelGGL(material_grads_fwd_kernel, dim3(d3h_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, tex, texj, nrm, nrmj, mask, mask_tap, n, kd,
                       kd_grad, ks_grad, nrm_grad);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// g_* [n][3] upstream gradients of the four outputs (each may be NULL = zero); d_tex, d_texj [n][6] and d_nrm, d_nrmj [n][3] are OVERWRITTEN
// (d_tex / d_texj as a pair may be NULL, d_texj alone may be NULL when texj is; d_nrm / d_nrmj as a pair may be NULL)
extern "C" int d3h_material_grads_bwd(const float* tex, const float* texj, const float* nrm, const float* nrmj, const float* mask, const float* mask_tap,
                                      int64_t n, const float* g_kd, const float* g_kd_grad, const float* g_ks_grad, const float* g_nrm_grad,
                                      float* d_tex, float* d_texj, float* d_nrm, float* d_nrmj, void* stream) {
    if (n < 0) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    if (!tex || (d_tex && texj && !d_texj) || (d_nrm && (!nrm || !nrmj || !mask || !mask_tap || !d_nrmj))) return D3H_ERR_ARG;
    hipLaunchKernelGGL(material_grads_bwd_kernel, dim3(d3h_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, tex, texj, nrm, nrmj, mask, mask_tap, n, g_kd,
                       g_kd_grad, g_ks_grad, g_nrm_grad, d_tex, d_texj, d_nrm, d_nrmj);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
