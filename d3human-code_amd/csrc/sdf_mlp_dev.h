// sdf_mlp_dev.h -- device helpers shared by the forward and backward SDF MLP kernels.
#pragma once
#include "d3h_common.h"
#include "sdf_mlp_layout.h"

namespace d3h_mlp {

// torch.nn.Softplus(beta=100, threshold=20): x*beta > threshold ? x : log1p(exp(x*beta))/beta
__device__ __forceinline__ float softplus100(float z) {
    float t = z * 100.0f;
    return (t > 20.0f) ? z : (log1pf(expf(t)) / 100.0f);
}

// geometry/embedding.py:33-38: out = [x] + [sin(f x), cos(f x) for f in 2^0..2^5]; index 39 is padding.
__device__ __forceinline__ float emb_feature(int e, float x0, float x1, float x2) {
    if (e >= EMB_DIM) return 0.f;
    if (e < 3) return e == 0 ? x0 : (e == 1 ? x1 : x2);
    int ep = e - 3;
    int fr = ep / 6, fn = (ep % 6) / 3, c = ep % 3;
    float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
    float v = xc * (float)(1 << fr);
    return fn ? cosf(v) : sinf(v);
}

struct Stage {
    f32x4 r[STAGE_F4];
};

// issue the global loads of one weight chunk (n4 float4, contiguous) into registers
__device__ __forceinline__ void stage_issue(Stage& s, const float* __restrict__ src, int n4, int tid) {
#pragma unroll
    for (int i = 0; i < STAGE_F4; ++i) {
        int j = tid + i * 256;
        if (j < n4) s.r[i] = *(const f32x4*)(src + 4 * (size_t)j);
    }
}
// write the staged chunk to an LDS buffer and publish it to the workgroup
__device__ __forceinline__ void stage_commit(const Stage& s, float* dst, int n4, int tid) {
#pragma unroll
    for (int i = 0; i < STAGE_F4; ++i) {
        int j = tid + i * 256;
        if (j < n4) *(f32x4*)(dst + 4 * j) = s.r[i];
    }
    __syncthreads();
}

// acc(32 out-features x 32 points) += W_chunk[:, 0:256] * SRC   (SRC = previous layer, in registers)
__device__ __forceinline__ void mac_hidden(f32x16& acc, const f32x16 (&src)[8], const float* wl, int lane) {
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        f32x4 a = *(const f32x4*)(wl + (g * 64 + lane) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], src[g >> 2][4 * (g & 3) + k], acc, 0, 0, 0);
    }
}

}  // namespace d3h_mlp
