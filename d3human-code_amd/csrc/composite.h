// composite.h -- shared by image_ops.hip (composite fwd / bwd) and raster.hip (the fused composite + antialias forward):
// the per-buffer composite of a render layer (render/render.py:375-382,430-449) as a per-pixel function, and the LDS-staged row copies.
#pragma once
#include "d3h_common.h"

namespace {

// A kernel that produces C consecutive floats per pixel with scalar stores writes 64 partial cache lines per instruction and the
// same lines again for every channel; rocprofv3 WRITE_SIZE showed 4-7x the algorithmic bytes for such outputs.  Instead every thread
// parks its C values in LDS and the workgroup copies the contiguous 256 x C block to HBM with 16-byte stores.
__device__ __forceinline__ void block_store_rows(float* __restrict__ dst, const float* __restrict__ lds, size_t first_pix, size_t npix, int C) {
    __syncthreads();
    const size_t base = first_pix * (size_t)C;
    const size_t remaining = (npix - first_pix) * (size_t)C;
    const int nflt = (int)(remaining < (size_t)(256 * C) ? remaining : (size_t)(256 * C));
    const int n4 = nflt >> 2;                                    // base is a multiple of 256*C floats: 16-byte aligned
    for (int j = threadIdx.x; j < n4; j += 256) *(float4*)(dst + base + 4 * (size_t)j) = *(const float4*)(lds + 4 * j);
    for (int j = 4 * n4 + threadIdx.x; j < nflt; j += 256) dst[base + j] = lds[j];
    __syncthreads();
}

// The read side of the same problem: a lane that reads the C consecutive floats of its pixel with scalar loads touches C x 64 partial
// lines per wave.  The workgroup copies the contiguous 256 x C block into LDS with 16-byte loads; lanes then read their row from LDS
// (row stride C floats: conflict-free for odd C).  Call with the whole workgroup; rows beyond npix are not written.
__device__ __forceinline__ void block_load_rows(float* __restrict__ lds, const float* __restrict__ src, size_t first_pix, size_t npix, int C) {
    __syncthreads();
    if (first_pix < npix) {
        const size_t base = first_pix * (size_t)C;
        const size_t remaining = (npix - first_pix) * (size_t)C;
        const int nflt = (int)(remaining < (size_t)(256 * C) ? remaining : (size_t)(256 * C));
        const int n4 = nflt >> 2;
        for (int j = threadIdx.x; j < n4; j += 256) *(float4*)(lds + 4 * j) = *(const float4*)(src + base + 4 * (size_t)j);
        for (int j = 4 * n4 + threadIdx.x; j < nflt; j += 256) lds[j] = src[base + j];
    }
    __syncthreads();
}

// ---- composite of the shaded layer against per-buffer backgrounds (render/render.py:375-382,430-449) --------------------------------
// Every buffer of the reference's single layer is [value channels, alpha = 1]; render_mesh lerps it against its background with
// weight coverage * alpha and antialiases each result separately.  Here all buffers are written, channel-concatenated, by one pass:
// covered pixel -> [src, 1], uncovered -> background (torch.lerp is exact at weights 0 and 1).  kind 0: zero background; 1: image
// background `bg` [Bbg][H][W][3] with alpha 0 ('shaded'); 2: constant 20 in every channel ('depth'); 3: alpha-only source
// ('msdf_image': lerp(0, 1, coverage * value) -> one channel coverage * value).
constexpr int COMP_MAX = 12;
struct CompSrc { const float* p; float* d; const float* bg; int stride, nch, kind, bg_batched, dch; };      // dch: floats per pixel of `d` (>= nch; the fused backward zero-fills the rest)
struct CompArgs { CompSrc s[COMP_MAX]; int n, C; };

// value of channel j (0 .. nch; nch = the alpha channel, kind 3: j = 0 only) of source c at global pixel i whose coverage is `cov`
__device__ __forceinline__ float comp_value(const CompSrc& c, int j, size_t i, bool cov, size_t hw) {
    if (c.kind == 3) return cov ? c.p[i * c.stride] : 0.f;
    if (cov) return j < c.nch ? c.p[i * c.stride + j] : 1.0f;
    if (c.kind == 1) return (j < 3 && j < c.nch) ? c.bg[(c.bg_batched ? i : i % hw) * 3 + j] : 0.f;
    return c.kind == 2 ? 20.0f : 0.f;
}
// the whole composited row of pixel i (a.C floats) into o
__device__ __forceinline__ void comp_row(const CompArgs& a, const float* __restrict__ rast, size_t i, size_t hw, float* __restrict__ o) {
    const bool cov = rast[4 * i + 3] > 0.f;
    for (int k = 0; k < a.n; ++k) {
        const CompSrc& c = a.s[k];
        const float* sp = c.p + i * c.stride;
        if (c.kind == 3) { *o++ = cov ? sp[0] : 0.f; continue; }
        if (cov) {
            for (int j = 0; j < c.nch; ++j) o[j] = sp[j];
            o[c.nch] = 1.0f;
        } else if (c.kind == 1) {
            const float* b = c.bg + (c.bg_batched ? i : i % hw) * 3;
            for (int j = 0; j < c.nch; ++j) o[j] = j < 3 ? b[j] : 0.f;
            o[c.nch] = 0.f;
        } else {
            float v = c.kind == 2 ? 20.0f : 0.f;
            for (int j = 0; j <= c.nch; ++j) o[j] = v;
        }
        o += c.nch + 1;
    }
}

static int comp_args(CompArgs& a, int nsrc, const float* const* src, float* const* dsrc, const int* stride, const int* nch, const int* kind,
                     const float* const* bg, const int* bg_batched, bool need_bg, const int* dch = nullptr) {
    if (nsrc <= 0 || nsrc > COMP_MAX || !stride || !nch || !kind) return D3H_ERR_ARG;
    a.n = nsrc;
    a.C = 0;
    for (int k = 0; k < nsrc; ++k) {
        if (kind[k] < 0 || kind[k] > 3 || nch[k] <= 0 || (kind[k] == 3 && nch[k] != 1) || (need_bg && kind[k] == 1 && !(bg && bg[k]))) return D3H_ERR_ARG;
        a.s[k].p = src ? src[k] : nullptr;
        a.s[k].d = dsrc ? dsrc[k] : nullptr;
        a.s[k].bg = bg ? bg[k] : nullptr;
        a.s[k].stride = stride[k]; a.s[k].nch = nch[k]; a.s[k].kind = kind[k];
        a.s[k].bg_batched = bg_batched ? bg_batched[k] : 0;
        a.s[k].dch = dch ? dch[k] : nch[k];
        if (a.s[k].dch < nch[k]) return D3H_ERR_ARG;
        a.C += kind[k] == 3 ? 1 : nch[k] + 1;
    }
    return D3H_OK;
}

}  // namespace
