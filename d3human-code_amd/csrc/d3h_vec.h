// d3h_vec.h -- small float3 algebra, normalisation helpers with the reference's epsilon conventions, and workgroup reductions shared by
// the image-space and mesh kernels (image_ops.hip, mesh_ops.hip).
#pragma once
#include "d3h_common.h"

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ void atomic_add3(float* p, V3 v) { atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); }

// render/util.py:25-29 safe_normalize: x / sqrt(clamp(dot(x,x), min=eps))
__device__ __forceinline__ V3 safe_normalize(V3 x, float eps = 1e-20f) { return x * (1.0f / sqrtf(fmaxf(dot(x, x), eps))); }
__device__ __forceinline__ V3 safe_normalize_bwd(V3 x, V3 g, float eps = 1e-20f) {
    float d = dot(x, x);
    if (d > eps) {
        float il = 1.0f / sqrtf(d);
        V3 n = x * il;
        return (g - n * dot(n, g)) * il;
    }
    return g * (1.0f / sqrtf(eps));
}
// torch.nn.functional.normalize: x / max(|x|, 1e-12)   (renderutils/bsdf.py:25-26)
__device__ __forceinline__ V3 fnormalize(V3 x) { return x * (1.0f / fmaxf(sqrtf(dot(x, x)), 1e-12f)); }
__device__ __forceinline__ V3 fnormalize_bwd(V3 x, V3 g) {
    float l = sqrtf(dot(x, x));
    if (l > 1e-12f) {
        float il = 1.0f / l;
        V3 n = x * il;
        return (g - n * dot(n, g)) * il;
    }
    return g * 1e12f;
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// sum over a 256-thread workgroup, result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* s4) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = s4[0] + s4[1] + s4[2] + s4[3];
    __syncthreads();
    return r;
}

// y = v / max(|v|, eps) (torch's F.normalize / cosine_similarity clamping); n = |v|
__device__ __forceinline__ V3 normalize_eps(V3 v, float eps, float& n) {
    n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return v * (1.0f / fmaxf(n, eps));
}
// gradient of y = v / max(|v|, eps) given dL/dy (torch: the clamped denominator is a constant below eps)
__device__ __forceinline__ V3 normalize_eps_bwd(V3 y, float n, float eps, V3 gy) {
    if (n > eps) return (gy - y * dot(y, gy)) * (1.0f / n);
    return gy * (1.0f / eps);
}

}  // namespace
