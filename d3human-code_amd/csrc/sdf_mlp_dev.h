// sdf_mlp_dev.h -- device helpers shared by the forward and backward SDF MLP kernels.
#pragma once
#include "d3h_common.h"
#include "sdf_mlp_layout.h"

namespace D3H_MLP_NS {

// torch.nn.Softplus(beta=100, threshold=20): x*beta > threshold ? x : log1p(exp(x*beta))/beta
// Evaluated with the hardware exp/log (v_exp_f32 / v_log_f32): |error| <= ~3e-9 absolute on h (1+e rounds at 6e-8, /100), i.e. at or
// below one ulp of h for every h that matters (h >= 0.02 has ulp >= 1.8e-9); the libm log1pf(expf()) pair costs ~6x the VALU issue
// slots of the MFMA epilogue for no gain in the sdf (parity test: 2e-7 on the network output, signs equal).
__device__ __forceinline__ float softplus100(float z) {
    float t = z * 100.0f;
    return (t > 20.0f) ? z : (__logf(1.0f + __expf(t)) * 0.01f);
}

// geometry/embedding.py:33-38: out = [x] + [sin(f x), cos(f x) for f in 2^0..2^5]; indices >= 39 are padding.
__device__ __forceinline__ float emb_feature(int e, float x0, float x1, float x2) {
    if (e >= EMB_DIM) return 0.f;
    if (e < 3) return e == 0 ? x0 : (e == 1 ? x1 : x2);
    int ep = e - 3;
    int fr = ep / 6, fn = (ep % 6) / 3, c = ep % 3;
    float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
    float v = xc * (float)(1 << fr);
    return fn ? cosf(v) : sinf(v);
}

// d/dx of the positional encoding applied to a direction u (JVP of geometry/embedding.py:33-38): feature e of J_emb(x) u
__device__ __forceinline__ float emb_tangent(int e, float x0, float x1, float x2, float u0, float u1, float u2) {
    if (e >= EMB_DIM) return 0.f;
    if (e < 3) return e == 0 ? u0 : (e == 1 ? u1 : u2);
    int ep = e - 3;
    int fr = ep / 6, fn = (ep % 6) / 3, c = ep % 3;
    float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
    float uc = c == 0 ? u0 : (c == 1 ? u1 : u2);
    float f = (float)(1 << fr);
    return fn ? (-f * sinf(f * xc) * uc) : (f * cosf(f * xc) * uc);
}

struct Stage {
    f32x4 r[STAGE_F4];
};

// issue the global loads of one weight chunk (n4 float4, contiguous) into registers
__device__ __forceinline__ void stage_issue(Stage& s, const float* __restrict__ src, int n4, int tid) {
#pragma unroll
    for (int i = 0; i < STAGE_F4; ++i) {
        int j = tid + i * NTHREADS;
        if (j < n4) s.r[i] = *(const f32x4*)(src + 4 * (size_t)j);
    }
}
// write the staged chunk to an LDS buffer and publish it to the workgroup
__device__ __forceinline__ void stage_commit(const Stage& s, float* dst, int n4, int tid) {
#pragma unroll
    for (int i = 0; i < STAGE_F4; ++i) {
        int j = tid + i * NTHREADS;
        if (j < n4) *(f32x4*)(dst + 4 * j) = s.r[i];
    }
    __syncthreads();
}

// Direct-to-LDS variant of the two calls above (no staging registers: 20 VGPRs less live across the MFMA loop).  A wave's 64 lanes
// write one contiguous KiB, which is exactly the chunk layout (float4 index = tid + i * NTHREADS).  The loads must be issued after the
// barrier that retired the previous readers of `dst` and are drained by glds_commit before the barrier that publishes them.
__device__ __forceinline__ void glds_issue(const float* __restrict__ src, float* dst, int n4, int tid) {
    const int wave_base = tid & ~63;
#pragma unroll
    for (int i = 0; i < STAGE_F4; ++i) {
        int j = tid + i * NTHREADS;
        if (j < n4) D3H_GLDS16(src + 4 * (size_t)j, dst + 4 * (wave_base + i * NTHREADS));
    }
}
__device__ __forceinline__ void glds_commit() {
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0) only (gfx9 encoding: lgkmcnt = 15, expcnt = 7 untouched)
    __syncthreads();
}

// two independent 16x16 accumulators (row blocks rbl = 0, 1 of a chunk) advance together: the 16x16x4 f32 MFMA has a 40-cycle
// dependent-accumulator latency against a 32-cycle issue interval, so alternating two chains keeps the matrix pipe paced, and the
// B operand (previous layer, in registers) is shared.  wl -> [rbl 2][blk NB][lane 64][4]
__device__ __forceinline__ void mac_hidden2(f32x4& acc0, f32x4& acc1, const f32x4 (&src)[16], const float* wl, int rstride, int lane) {
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        f32x4 a0 = *(const f32x4*)(wl + (blk * 64 + lane) * 4);
        f32x4 a1 = *(const f32x4*)(wl + rstride + (blk * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], src[blk][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], src[blk][r], acc1, 0, 0, 0);
        }
    }
}

// mac_hidden2 with a hook after the first half of the k-loop: the SIMD partner waves 4..7 run their per-chunk epilogue there (see the
// stagger note in sdf_mlp.hip).  `mid` may modify src[14], src[15] (they are only read at blk 14, 15).
template <class F>
__device__ __forceinline__ void mac_hidden2_mid(f32x4& acc0, f32x4& acc1, f32x4 (&src)[16], const float* wl, int rstride, int lane, F&& mid) {
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        if (blk == 8) mid();
        f32x4 a0 = *(const f32x4*)(wl + (blk * 64 + lane) * 4);
        f32x4 a1 = *(const f32x4*)(wl + rstride + (blk * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], src[blk][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], src[blk][r], acc1, 0, 0, 0);
        }
    }
}

// single chain (embedding blocks of the backward pass)
__device__ __forceinline__ void mac_hidden(f32x4& acc, const f32x4 (&src)[16], const float* wl, int lane) {
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        f32x4 a = *(const f32x4*)(wl + (blk * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], src[blk][r], acc, 0, 0, 0);
    }
}

}  // namespace D3H_MLP_NS
