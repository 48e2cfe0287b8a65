// sdf_mlp_dev.h -- device helpers shared by the forward and backward SDF MLP kernels.
#pragma once
#include "d3h_common.h"
#include "sdf_mlp_layout.h"

namespace D3H_MLP_NS {

// torch.nn.Softplus(beta=100, threshold=20): x*beta > threshold ? x : log1p(exp(x*beta))/beta
// Evaluated with the hardware base-2 exp/log (v_exp_f32 / v_log_f32, 1 ulp each): |error| <= ~3e-9 absolute on h (1+e rounds at 6e-8,
// /100), i.e. at or below one ulp of h for every h that matters (h >= 0.02 has ulp >= 1.8e-9); parity test: 2e-7 on the network output,
// signs equal.  Instruction count matters here more than anywhere else: on gfx950 a VALU instruction of one wave does NOT overlap the
// fp32 MFMAs of the other wave on its SIMD (tools/probe/mfma_valu_overlap.hip: 4.31 ms of MFMAs + 2.52 ms of FMAs on the partner
// wave = 7.02 ms), so every epilogue instruction is paid in matrix-pipe time.  Six instructions per value: the raw builtins instead of
// __logf (whose extended-precision log2 -> ln conversion and denormal guard were 13 instructions) and one folded scale on each side.
constexpr float SP_LOG2E_100 = 144.26950408889634f;      // 100 * log2(e)
constexpr float SP_LN2_BY_100 = 0.0069314718055994531f;  // ln(2) / 100
__device__ __forceinline__ float softplus100(float z) {
    const float e = __builtin_amdgcn_exp2f(z * SP_LOG2E_100);
    const float l = __builtin_amdgcn_logf(1.0f + e) * SP_LN2_BY_100;
    return (z > 0.2f) ? z : l;
}
// softplus'(z) recovered from h = softplus(z): 1 - exp(-100 h)  (exactly 1 above the threshold)
__device__ __forceinline__ float dsoftplus_from_h(float h) {
    return (h > 0.2f) ? 1.0f : (1.0f - __builtin_amdgcn_exp2f(h * -SP_LOG2E_100));
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

// D3H_SDF_GLDS = 1: weight chunks travel global -> LDS directly; 0: through the Stage registers.  D3H_SDF_PIPE = 1: software-pipelined
// fragment reads.  tools/build_variant.sh + tools/ab_kernels.sh compare the four combinations on one box: 1 / 1 is best or equal for
// every kernel since the epilogue operands are prefetched through LDS (before that the reverse sweeps preferred 0 / 0).
#ifndef D3H_SDF_GLDS
#define D3H_SDF_GLDS 1
#endif
#if defined(D3H_PROBE_NO_STAGE)          // tools/probe/sdf_variants.sh: timing experiments only (results are garbage)
#define SDF_STAGE_ISSUE(st, src, dst, n4, tid) ((void)0)
#if defined(D3H_PROBE_NO_BARRIER)
#define SDF_STAGE_COMMIT(st, dst, n4, tid) ((void)0)
#else
#define SDF_STAGE_COMMIT(st, dst, n4, tid) __syncthreads()
#endif
#elif D3H_SDF_GLDS
#define SDF_STAGE_ISSUE(st, src, dst, n4, tid) glds_issue(src, dst, n4, tid)
#define SDF_STAGE_COMMIT(st, dst, n4, tid) glds_commit()
#else
#define SDF_STAGE_ISSUE(st, src, dst, n4, tid) stage_issue(st, src, n4, tid)
#define SDF_STAGE_COMMIT(st, dst, n4, tid) stage_commit(st, dst, n4, tid)
#endif

// two independent 16x16 accumulators (row blocks rbl = 0, 1 of a chunk) advance together: the 16x16x4 f32 MFMA has a 40-cycle
// dependent-accumulator latency against a 32-cycle issue interval, so alternating two chains keeps the matrix pipe paced, and the
// B operand (previous layer, in registers) is shared.  wl -> [rbl 2][blk NB][lane 64][4]
// D3H_SDF_PRIO = 1: waves 4..7 of a workgroup (the second-dispatched half: the arbitration loser against its SIMD partner on every
// segment, MI355X_MICROARCH.md 'Two waves per SIMD' item 4) raise their priority once, before the tile loop.
#ifndef D3H_SDF_PRIO
#define D3H_SDF_PRIO 0
#endif
#ifndef D3H_SDF_PIPE
#define D3H_SDF_PIPE 1
#endif
__device__ __forceinline__ void mac_hidden2(f32x4& acc0, f32x4& acc1, const f32x4 (&src)[16], const float* wl, int rstride, int lane) {
#if !D3H_SDF_PIPE
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
    return;
#endif
    // The A fragments of k-block blk + 1 are requested BEFORE the eight MFMAs of k-block blk issue (D3H_SCHED_FENCE keeps the compiler
    // from sinking the ds_read back to its first use): with a single fragment buffer the wave sat out one LDS round trip per 8 MFMAs.
    f32x4 a0 = *(const f32x4*)(wl + lane * 4);
    f32x4 a1 = *(const f32x4*)(wl + rstride + lane * 4);
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        f32x4 n0 = a0, n1 = a1;
        if (blk < 15) {
            n0 = *(const f32x4*)(wl + ((blk + 1) * 64 + lane) * 4);
            n1 = *(const f32x4*)(wl + rstride + ((blk + 1) * 64 + lane) * 4);
        }
        D3H_SCHED_FENCE();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], src[blk][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], src[blk][r], acc1, 0, 0, 0);
        }
        D3H_SCHED_FENCE();
        a0 = n0;
        a1 = n1;
    }
}

// mac_hidden2 with a hook after the first half of the k-loop: the SIMD partner waves 4..7 run their per-chunk epilogue there (see the
// stagger note in sdf_mlp.hip).  `mid` may modify src[14], src[15] (they are only read at blk 14, 15).
template <class F>
__device__ __forceinline__ void mac_hidden2_mid(f32x4& acc0, f32x4& acc1, f32x4 (&src)[16], const float* wl, int rstride, int lane, F&& mid) {
#if !D3H_SDF_PIPE
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
    return;
#endif
    f32x4 a0 = *(const f32x4*)(wl + lane * 4);
    f32x4 a1 = *(const f32x4*)(wl + rstride + lane * 4);
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        if (blk == 8) mid();
        f32x4 n0 = a0, n1 = a1;
        if (blk < 15) {
            n0 = *(const f32x4*)(wl + ((blk + 1) * 64 + lane) * 4);
            n1 = *(const f32x4*)(wl + rstride + ((blk + 1) * 64 + lane) * 4);
        }
        D3H_SCHED_FENCE();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], src[blk][r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], src[blk][r], acc1, 0, 0, 0);
        }
        D3H_SCHED_FENCE();
        a0 = n0;
        a1 = n1;
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
