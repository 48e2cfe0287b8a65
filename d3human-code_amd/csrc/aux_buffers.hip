// Forward-only auxiliary buffers of a render layer: z / z-gradient, depth, inverse depth (render/render.py:291-299,197-199).
// A translation unit of its own: added to raster.hip it changed the code generated for that file's kernels (raster_tris +20 %).
#include <hip/hip_runtime.h>

#include "d3h_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// forward-only auxiliary buffers of a render layer (render/render.py:291-299 z / z-gradient, :197-199 depth / inverse depth): the
// reference builds them from a clip-space dr.interpolate with derivatives plus a dozen elementwise ops per buffer; no gradient flows
// through z_grad (torch.no_grad in the reference) and, unless FLAGS.use_depth, none is asked of depth / invdepth
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aux_buffers_fwd_kernel(const float* __restrict__ clip, int clip_bstride, const float* __restrict__ rast,
                                                              const float* __restrict__ db, const int* __restrict__ tri,
                                                              const float* __restrict__ gb_pos, const float* __restrict__ view_pos,
                                                              int view_bstride, size_t npix_total, size_t npix_per_b,
                                                              float* __restrict__ z_grad, float* __restrict__ depth, float* __restrict__ invdepth) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix_total) return;
    const int b = (int)(i / npix_per_b);
    if (z_grad) {
        const float4 r = *(const float4*)(rast + 4 * i);
        const int id = (int)r.w;
        float y_dx = 0.f, y_dy = 0.f, z = 0.f, w = 0.f;          // uncovered: dr.interpolate returns zeros
        if (id > 0) {
            const float* cb = clip + (size_t)b * clip_bstride;
            const size_t f = (size_t)(id - 1);
            const float4 p0 = *(const float4*)(cb + 4 * (size_t)tri[3 * f]);
            const float4 p1 = *(const float4*)(cb + 4 * (size_t)tri[3 * f + 1]);
            const float4 p2 = *(const float4*)(cb + 4 * (size_t)tri[3 * f + 2]);
            const float u = r.x, v = r.y, t = 1.0f - u - v;
            const float4 d = *(const float4*)(db + 4 * i);
            z = u * p0.z + v * p1.z + t * p2.z;
            w = u * p0.w + v * p1.w + t * p2.w;
            // channels 2 and 3 of the [.., 8] derivative image (x: dX dY, y: dX dY, ...) -- render.py:296-297 indexes `[..., 2:3]` and
            // `[..., 3:4]`, i.e. d(clip y)/dX and d(clip y)/dY (reference quirk, reproduced)
            const float e0 = p0.y - p2.y, e1 = p1.y - p2.y;
            y_dx = d.x * e0 + d.z * e1;
            y_dy = d.y * e0 + d.w * e1;
        }
        const float eps = 0.00001f;
        const float z0 = fmaxf(z, eps) / fmaxf(w, eps);
        const float z1 = fmaxf(z + fabsf(y_dx), eps) / fmaxf(w + fabsf(y_dy), eps);
        float* o = z_grad + 3 * i;
        o[0] = z0; o[1] = fabsf(z1 - z0); o[2] = 0.f;
    }
    if (depth || invdepth) {
        const float* vp = view_pos + (size_t)b * view_bstride;
        const float* g = gb_pos + 3 * i;
        const float d0 = g[0] - vp[0], d1 = g[1] - vp[1], d2 = g[2] - vp[2];
        const float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
        if (depth) depth[i] = sqrtf((q0 + q1) + q2);
        if (invdepth) invdepth[i] = 1.0f / sqrtf(((q0 + 1e-8f) + (q1 + 1e-8f)) + (q2 + 1e-8f));
    }
}


}  // namespace

// clip [nb][nv][4] (clip_bstride = nv*4), rast / db [nb][H][W][4], gb_pos [nb][H][W][3] (needed for depth / invdepth), view_pos
// [nb or 1][3] (view_bstride = 3 or 0).  z_grad [nb][H][W][3], depth / invdepth [nb][H][W]: each may be NULL (not produced).
extern "C" int d3h_aux_buffers_fwd(const float* clip, int clip_bstride, const float* rast, const float* db, const int* tri, const float* gb_pos,
                                   const float* view_pos, int view_bstride, int nb, int H, int W, float* z_grad, float* depth, float* invdepth,
                                   void* stream) {
    size_t npb = (size_t)H * W, n = npb * nb;
    if (n == 0) return D3H_OK;
    // clip / tri may be NULL for an empty mesh (no vertex, no face): every pixel is uncovered and neither is dereferenced
    if (!rast || (z_grad && !db) || ((depth || invdepth) && (!gb_pos || !view_pos))) return D3H_ERR_ARG;
    hipLaunchKernelGGL(aux_buffers_fwd_kernel, dim3(d3h_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, clip, clip_bstride, rast, db, tri, gb_pos,
                       view_pos, view_bstride, n, npb, z_grad, depth, invdepth);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

