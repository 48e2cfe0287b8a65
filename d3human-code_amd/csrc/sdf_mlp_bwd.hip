// sdf_mlp_bwd.hip -- backward of the fused PE + SDF MLP (csrc/sdf_mlp.hip) on gfx950.
//
// Autograd of geometry/mlp.py:34-45 + geometry/embedding.py:21-38, i.e. what loss.backward() does for
// HmSDFTetsGeometry.sdf_net in the reference (train.py:742).  Three kernels, all on the exact-f32 matrix pipe:
//
//  1. sdf_mlp_bwd_data   dH_{l-1}^T = W_l^T * dZ_l^T, layer 6 -> 0, with dZ_l = dH_l * softplus'(h_l).  Same
//                        register-resident structure as the forward: one wave owns 16 points x 256 features (16x16x4 MFMA, 8 waves),
//                        the MFMA output of layer l is the B operand of layer l-1; h_l comes back from the
//                        tile-packed `act` buffer in exactly the accumulator layout (no shuffles, no LDS).
//                        Writes dZ_l (tile-packed) for the weight-gradient kernel and d(x) through the encoding.
//  2. sdf_mlp_bwd_dw     dW_l += dZ_l^T[256 x pts] * H_{l-1}[pts x K]: split over points across workgroups; the
//                        two tile-packed operands are transposed through LDS (pitch 33: conflict-free), 256x128
//                        output block per workgroup in accumulators, one fp32 atomic add pass at the end
//                        (each wave-instruction adds two 128-B row segments: the full-rate shape).
//  3. sdf_mlp_bwd_last   dW_7, db_7 (the 256 -> 1 head).
//
// softplus'(z) from the stored h = softplus(z): sigmoid(100 z) = 1 - exp(-100 h)  (exact identity; torch's
// threshold branch 100 z > 20 gives 1, which 1 - exp(-100 h) equals in fp32).
#include "sdf_mlp_dev.h"
#include "sdf_mlp_x3.h"
#include <type_traits>

using namespace D3H_MLP_NS;

namespace {

// ------------------------------------------------------------------------------------------------
// transposed pack
// ------------------------------------------------------------------------------------------------
__global__ void sdf_mlp_pack_t_kernel(const float* __restrict__ w0, const float* __restrict__ wh, const float* __restrict__ w4,
                                      float* __restrict__ wpackT) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= WPACKT_FLOATS) return;
    int l = t_layer_of_offset(idx);
    int local = idx - t_layer_offset(l);
    int r = local & 3, lane = (local >> 2) & 63, rest = local >> 8;      // rest = flat (chunk, rbl, blk), 16 blks per in-row-block
    int i = lane & 15, q = lane >> 4;
    int blk = rest & 15, rbg = rest >> 4;                                  // rbg = global 16-row block of INPUT features
    int out = 16 * blk + 4 * q + r;
    int in = 16 * rbg + i;
    float v = 0.f;
    if (l == 0) {
        if (in < EMB_DIM) v = w0[out * EMB_DIM + in];
    } else if (l == 4) {
        if (rbg < 16) v = w4[out * (256 + EMB_DIM) + in];
        else {
            int e = in - 256;
            if (e < EMB_DIM) v = w4[out * (256 + EMB_DIM) + 256 + e];
        }
    } else {
        int hi = (l < 4) ? (l - 1) : (l - 2);
        v = wh[(size_t)hi * 65536 + out * 256 + in];
    }
    wpackT[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// 0. active 16-point tiles
// ------------------------------------------------------------------------------------------------
// d(loss)/d(sdf) of a training sweep is non-zero only at grid vertices that touch the extracted surface (marching tets and the
// sign-change regulariser read the sdf of sign-changing edges only): ~10-20 % of the 16-point tiles.  A point with zero upstream
// gradient contributes exactly zero to dX, every dW and db, so the backward runs over the compacted list of active tiles -- the
// same sums, fewer terms.  list[0 .. count) = ids of the 16-point tiles with any non-zero gout (arbitrary order).
__global__ __launch_bounds__(256) void sdf_mlp_active_tiles_kernel(const float* __restrict__ gout, int64_t n, int ntiles16, int* __restrict__ list,
                                                                   int* __restrict__ count) {
    int t = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    if (t < ntiles16) {
        int64_t p0 = (int64_t)t * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (p0 + k < n) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
                    if (gout[(p0 + k) * NOUT + o] != 0.f) act = true;
            }
    }
    unsigned long long m = __ballot(act);
    int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0);
    if (act) list[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
}

#if D3H_MLP_NOUT == 1
// ---- the compact form (round 5; used when the forward ran without the activation save) -----------------------------------------------------
// The active set is taken per POINT, not per 16-point tile of consecutive grid vertices: a surface crossing touches a few vertices of a run of 16,
// so gathering the active points into dense tiles leaves ~4x fewer tiles than marking position tiles (tet-res 128, 8.7 k faces: ~9 k points with a
// non-zero d(sdf) = ~550 gathered tiles against ~2 450 position tiles).  The gathered problem -- x_g, gout_g, and the activations RECOMPUTED for it
// by the forward kernel -- is then an ordinary small dense backward in "list" mode with the identity list and its tile count on the device.
//   scratch (ints unless noted), d3h_sdf_mlp_bwd_scratch_ints(n) in total:
//     counts[8]: [0] active points, [1] gathered 16-point tiles | plist[n] | ident[n / 16 + 2] | x_g[3 n16] f32 | gout_g[n16] f32 | dx_g[3 n16] f32
//     (n16 = n rounded up to whole tiles; every sub-array 16-byte aligned)
// counts[2] (zeroed by the caller): the largest |gout| as float bits (non-negative floats order as unsigned integers) -- the operand scale of the
// h2 sweeps (h2_grad_scale)
__global__ __launch_bounds__(256) void sdf_mlp_active_points_kernel(const float* __restrict__ gout, int64_t n, int* __restrict__ plist, int* __restrict__ counts) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float gv = p < n ? gout[p] : 0.f;
    const bool act = p < n && gv != 0.f;
    {
        float m = fabsf(gv);
        if (!(m < 3.0e38f)) m = 0.f;                  // inf / NaN upstream: no scale can help, the sweep's outputs will show it
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)counts + 2, __float_as_uint(m));
    }
    const unsigned long long m = __ballot(act);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(counts, __popcll(m));
    base = __shfl(base, 0);
    if (act) plist[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)p;
}
// x_g = x (+ disp * deform: the two roundings of hmsdf.py:433, as the sweep kernel) and gout_g of the listed points, zero-padded to whole tiles;
// ident[t] = t; counts[1] = tiles.  Grid-stride over the (device-side) count.
__global__ __launch_bounds__(256) void sdf_mlp_gather_points_kernel(const float* __restrict__ x, const float* __restrict__ deform, float disp,
                                                                    const float* __restrict__ gout, const int* __restrict__ plist, int* __restrict__ counts,
                                                                    int* __restrict__ ident, float* __restrict__ xg, float* __restrict__ gg) {
    const int cnt = counts[0];
    const int ntile = (cnt + 15) >> 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[1] = ntile;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ntile * 16; g += gridDim.x * 256) {
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, gv = 0.f;
        if (g < cnt) {
            const int64_t p = plist[g];
            x0 = x[3 * p + 0]; x1 = x[3 * p + 1]; x2 = x[3 * p + 2];
            if (deform) {
                x0 = __fadd_rn(x0, __fmul_rn(disp, deform[3 * p + 0]));
                x1 = __fadd_rn(x1, __fmul_rn(disp, deform[3 * p + 1]));
                x2 = __fadd_rn(x2, __fmul_rn(disp, deform[3 * p + 2]));
            }
            gv = gout[p];
        }
        xg[3 * (size_t)g + 0] = x0; xg[3 * (size_t)g + 1] = x1; xg[3 * (size_t)g + 2] = x2;
        gg[g] = gv;
        if ((g & 15) == 0) ident[g >> 4] = g >> 4;
    }
}
// gout_g of a list built BEFORE the upstream gradient existed (d3h_sdf_mlp_bwd_prepare), zero-padded to whole tiles, and the largest |gout| of the
// list as float bits into counts[6] (zeroed by the prepare call)
__global__ __launch_bounds__(256) void sdf_mlp_gather_gout_kernel(const float* __restrict__ gout, const int* __restrict__ plist, int* __restrict__ counts,
                                                                  float* __restrict__ gg) {
    const int cnt = counts[0];
    const int ntile = (cnt + 15) >> 4;
    float m = 0.f;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ntile * 16; g += gridDim.x * 256) {
        const float gv = g < cnt ? gout[plist[g]] : 0.f;
        gg[g] = gv;
        const float a = fabsf(gv);
        if (a < 3.0e38f) m = fmaxf(m, a);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)counts + 6, __float_as_uint(m));
}
// dx[plist[g]] = dx_g[g]: every listed point once (dx was zero-filled)
__global__ __launch_bounds__(256) void sdf_mlp_scatter_dx_kernel(const float* __restrict__ dxg, const int* __restrict__ plist, const int* __restrict__ counts,
                                                                 float* __restrict__ dx) {
    const int cnt = counts[0];
    for (int g = blockIdx.x * 256 + threadIdx.x; g < cnt; g += gridDim.x * 256) {
        const int64_t p = plist[g];
        dx[3 * p + 0] = dxg[3 * (size_t)g + 0]; dx[3 * p + 1] = dxg[3 * (size_t)g + 1]; dx[3 * p + 2] = dxg[3 * (size_t)g + 2];
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// 1. backward data
// ------------------------------------------------------------------------------------------------
// in place: v (dH block) *= softplus'(h) with h from the saved activations; store dZ (tile-packed)
__device__ __forceinline__ void dz_block(f32x4& v, const float* act_l, float* dz_l, int rb, int lane) {
    size_t off = (size_t)(rb * 64 + lane) * 4;
    f32x4 hh = *(const f32x4*)(act_l + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = v[r] * dsoftplus_from_h(hh[r]);
        v[r] = o[r];
    }
    *(f32x4*)(dz_l + off) = o;      // (a non-temporal store here is slower: the dW pass re-reads dZ while part of it is still cached)
}

// eikonal second-order pass: dZ^_l = dH^_l * softplus'(h_l) + e_l, with e_l (written by the tangent pass) replaced IN PLACE by dZ^_l
__device__ __forceinline__ void dz_block_inject(f32x4& v, const float* act_l, float* e_l, int rb, int lane) {
    size_t off = (size_t)(rb * 64 + lane) * 4;
    f32x4 hh = *(const f32x4*)(act_l + off);
    f32x4 ee = *(const f32x4*)(e_l + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = v[r] * dsoftplus_from_h(hh[r]) + ee[r];
        v[r] = o[r];
    }
    *(f32x4*)(e_l + off) = o;
}

// the two above with h (and e) already fetched: the chunk loops prefetch them into LDS while the chunk's MFMAs run
__device__ __forceinline__ void dz_block_pre(f32x4& v, const f32x4 hh, float* dz_l, int rb, int lane) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = v[r] * dsoftplus_from_h(hh[r]);
        v[r] = o[r];
    }
    *(f32x4*)(dz_l + (size_t)(rb * 64 + lane) * 4) = o;
}
__device__ __forceinline__ void dz_block_inject_pre(f32x4& v, const f32x4 hh, const f32x4 ee, float* e_l, int rb, int lane) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = v[r] * dsoftplus_from_h(hh[r]) + ee[r];
        v[r] = o[r];
    }
    *(f32x4*)(e_l + (size_t)(rb * 64 + lane) * 4) = o;
}


// INJECT = false: the first-order backward (gout == nullptr means d(sdf) = 1 for every point: the gradient pass of the eikonal term).
// INJECT = true: the reverse sweep of the eikonal second-order pass: starts from dH^_6 = 0 (the loss does not see f), adds the
// curvature term e_l at every layer; `dz` is then the e / dZ^ buffer (updated in place); no dx.
template <bool INJECT>
__global__ __launch_bounds__(NTHREADS, 2) void sdf_mlp_bwd_data_kernel(const float* __restrict__ x, const float* __restrict__ deform, float disp,
                                                                      const float* __restrict__ gout, const float* __restrict__ w7,
                                                                      const float* __restrict__ wpackT, const float* __restrict__ act,
                                                                      float* __restrict__ dz, float* __restrict__ dx, int64_t n, int ntiles,
                                                                      const int* __restrict__ tile_list, const int* __restrict__ tile_count) {
    __shared__ __attribute__((aligned(16))) float wbuf[2][T_CHUNK_FLOATS];
    __shared__ __attribute__((aligned(16))) float w7s[NOUT * 256];
    // h (and, INJECT, e) of the two row blocks a chunk produces, fetched global -> LDS while the chunk's MFMAs run: the dZ epilogue of
    // layer l - 1 runs on each pair right after the chunk of layer l that produced it (as the forward does), not as 16 load-wait-store
    // round trips at the start of the next layer
    __shared__ __attribute__((aligned(16))) float pfb[NWAVES * (INJECT ? 4 : 2) * 256];
    // Balanced assignment of 16-point wave tiles (see sdf_mlp_fwd_kernel): in round r, wave w of workgroup b owns entry r * 8G + w * G + b
    // of the tile sequence -- the active list in sparse mode, 0 .. n16-1 otherwise; waves past the end skip the arithmetic (wave-uniform
    // `on`) but keep staging weights.  Every launch of this kernel is a "small" one (50 000 eikonal samples, or the active tiles of
    // the grid sweep), so the ragged last round matters: 391 whole 128-point tiles took two full rounds on 256 CUs.
    const int n_active = tile_list ? *tile_count : 0;
    const int64_t n16 = tile_list ? (int64_t)n_active : (int64_t)ntiles * 8;     // dense: incl. the padding wave tiles (zeros the dW pass reads)
    const int G = (int)gridDim.x;
    const int nrounds = (int)((n16 + NWAVES * (int64_t)G - 1) / (NWAVES * (int64_t)G));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;
    constexpr int N4 = T_CHUNK_FLOATS / 4;

    for (int j = tid; j < NOUT * 256; j += NTHREADS) w7s[j] = w7[j];
    Stage st;
    int pb = 0;
    SDF_STAGE_ISSUE(st, wpackT, wbuf[0], N4, tid);
    SDF_STAGE_COMMIT(st, wbuf[0], N4, tid);

    f32x4 X[16], Y[16];
#if D3H_SDF_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t seq = (int64_t)rnd * NWAVES * G + (int64_t)wave * G + blockIdx.x;
        const bool on = seq < n16;                       // wave-uniform
        const int64_t t16 = on ? (tile_list ? (int64_t)tile_list[seq] : seq) : 0;
        const int64_t p = t16 * 16 + (lane & 15);
        const bool valid = on && p < n;
        const float* act_tile = act + t16 * ACT_TILE_FLOATS;
        float* dz_tile = dz + t16 * ACT_TILE_FLOATS;
        float g[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) g[o] = INJECT ? 0.f : (valid ? (gout ? gout[p * NOUT + o] : 1.f) : 0.f);

        // dH_6 = sum_o g_o * W7[o]   (net.14: out_o = W7[o] . h_6 + b7[o])
#pragma unroll
        for (int rb = 0; rb < 16; ++rb) {
            X[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                f32x4 w = *(const f32x4*)(w7s + 256 * o + 16 * rb + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) X[rb][r] = fmaf(g[o], w[r], X[rb][r]);
            }
        }

        f32x4 E[EMB_BLKS];
#pragma unroll
        for (int b = 0; b < EMB_BLKS; ++b) E[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        float* pf = pfb + wave * ((INJECT ? 4 : 2) * 256);
        auto prefetch = [&](int lp, int c) {           // h (and e) of row blocks 2c, 2c + 1 of layer lp: one KiB per wave-instruction
            const size_t o0 = (size_t)lp * ACT_LAYER_FLOATS + (size_t)((2 * c) * 64 + lane) * 4;
            D3H_GLDS16(act_tile + o0, pf);
            D3H_GLDS16(act_tile + o0 + 256, pf + 256);
            if (INJECT) {
                D3H_GLDS16(dz_tile + o0, pf + 512);
                D3H_GLDS16(dz_tile + o0 + 256, pf + 768);
            }
        };
        auto dz_pair = [&](f32x4& v0, f32x4& v1, int lp, int c) {
            const float* pl = pf + lane * 4;
            float* dzl = dz_tile + lp * ACT_LAYER_FLOATS;
            if (INJECT) {
                dz_block_inject_pre(v0, *(const f32x4*)pl, *(const f32x4*)(pl + 512), dzl, 2 * c, lane);
                dz_block_inject_pre(v1, *(const f32x4*)(pl + 256), *(const f32x4*)(pl + 768), dzl, 2 * c + 1, lane);
            } else {
                dz_block_pre(v0, *(const f32x4*)pl, dzl, 2 * c, lane);
                dz_block_pre(v1, *(const f32x4*)(pl + 256), dzl, 2 * c + 1, lane);
            }
        };
        const float* next = wpackT + T_CHUNK_FLOATS;   // chunk stream pointer (next chunk to prefetch)
        for (int it = 0; it < 3; ++it) {
            {   // layer l = 6, 4, 2 : X (= dZ_l) -> Y (= dH_{l-1}, turned into dZ_{l-1} pair by pair)
                const int l = 6 - 2 * it;
                if (on && it == 0) {        // dZ_6 from the head's dH_6; every later dZ is made in the chunk loop that produced its dH
#pragma unroll
                    for (int rb = 0; rb < 16; ++rb) {
                        if (INJECT) dz_block_inject(X[rb], act_tile + l * ACT_LAYER_FLOATS, dz_tile + l * ACT_LAYER_FLOATS, rb, lane);
                        else dz_block(X[rb], act_tile + l * ACT_LAYER_FLOATS, dz_tile + l * ACT_LAYER_FLOATS, rb, lane);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    SDF_STAGE_ISSUE(st, next, wbuf[pb ^ 1], N4, tid);
                    next += T_CHUNK_FLOATS;
                    if (on) prefetch(l - 1, c);
                    if (on) {
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        mac_hidden2(acc0, acc1, X, wbuf[pb], 16 * 256, lane);
                        Y[2 * c] = acc0;
                        Y[2 * c + 1] = acc1;
                    }
                    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the prefetch has landed
                    SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], N4, tid);
                    pb ^= 1;
                    if (on) dz_pair(Y[2 * c], Y[2 * c + 1], l - 1, c);
                }
                if (l == 4) {   // skip layer: the embedding columns of net.8 (mlp.py:40-41): embedding in-blocks 0,1 | 2,(pad)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        SDF_STAGE_ISSUE(st, next, wbuf[pb ^ 1], N4, tid);
                        next += T_CHUNK_FLOATS;
                        if (on) {
                            mac_hidden(E[2 * c], X, wbuf[pb], lane);
                            if (2 * c + 1 < EMB_BLKS) mac_hidden(E[2 * c + 1], X, wbuf[pb] + 16 * 256, lane);
                        }
                        SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], N4, tid);
                        pb ^= 1;
                    }
                }
            }
            {   // layer l = 5, 3, 1 : Y (= dZ_l) -> X (= dH_{l-1} -> dZ_{l-1})
                const int l = 5 - 2 * it;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    SDF_STAGE_ISSUE(st, next, wbuf[pb ^ 1], N4, tid);
                    next += T_CHUNK_FLOATS;
                    if (on) prefetch(l - 1, c);
                    if (on) {
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        mac_hidden2(acc0, acc1, Y, wbuf[pb], 16 * 256, lane);
                        X[2 * c] = acc0;
                        X[2 * c + 1] = acc1;
                    }
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], N4, tid);
                    pb ^= 1;
                    if (on) dz_pair(X[2 * c], X[2 * c + 1], l - 1, c);
                }
            }
        }
        // layer 0: X = dZ_0 (made in the last chunk loop);  dEmb += W0^T dZ_0
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // after the last chunk of the stream comes chunk 0 of the next tile
            SDF_STAGE_ISSUE(st, (c == 0) ? next : wpackT, wbuf[pb ^ 1], N4, tid);
            next += T_CHUNK_FLOATS;
            if (on) {
                mac_hidden(E[2 * c], X, wbuf[pb], lane);
                if (2 * c + 1 < EMB_BLKS) mac_hidden(E[2 * c + 1], X, wbuf[pb] + 16 * 256, lane);
            }
            SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], N4, tid);
            pb ^= 1;
        }

        // d(x) through the positional encoding (embedding.py:33-38)
        if (!INJECT && dx && on) {
            float x0 = 0.f, x1 = 0.f, x2 = 0.f;
            if (valid) {
                x0 = x[3 * p + 0]; x1 = x[3 * p + 1]; x2 = x[3 * p + 2];
                if (deform) {
                    x0 = __fadd_rn(x0, __fmul_rn(disp, deform[3 * p + 0]));
                    x1 = __fadd_rn(x1, __fmul_rn(disp, deform[3 * p + 1]));
                    x2 = __fadd_rn(x2, __fmul_rn(disp, deform[3 * p + 2]));
                }
            }
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int b = 0; b < EMB_BLKS; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int e = 16 * b + 4 * q + r;
                    if (e < EMB_DIM) {
                        float ge = E[b][r];
                        int c;
                        float coef;
                        if (e < 3) { c = e; coef = 1.f; }
                        else {
                            int ep = e - 3;
                            int fr = ep / 6, fn = (ep % 6) / 3;
                            c = ep % 3;
                            float f = (float)(1 << fr);
                            float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
                            coef = fn ? (-f * sinf(f * xc)) : (f * cosf(f * xc));
                        }
                        float t = ge * coef;
                        if (c == 0) d0 += t; else if (c == 1) d1 += t; else d2 += t;
                    }
                }
            d0 += __shfl_xor(d0, 16); d0 += __shfl_xor(d0, 32);
            d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
            d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
            if (valid && q == 0) { dx[3 * p + 0] = d0; dx[3 * p + 1] = d1; dx[3 * p + 2] = d2; }
        }
    }
#if D3H_SDF_GLDS
    __builtin_amdgcn_s_waitcnt(0x0f70);      // drain the dangling weight prefetch (an LDS write) before the LDS is released
#endif
}

#if D3H_MLP_NOUT == 1
// ------------------------------------------------------------------------------------------------
// 1b. backward data on the bf16 matrix pipe (sdf_mlp_x3.h): the same sweep with dZ_l travelling as three bf16 planes and W_l^T pre-split
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ void sdf_mlp_pack_t3_kernel(const float* __restrict__ w0, const float* __restrict__ wh, const float* __restrict__ w4,
                                       unsigned* __restrict__ wpackT3) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= XPT<NP>::WPACKT_DWORDS) return;
    const int chunk = idx / XPT<NP>::HID_CHUNK;
    const int l = t_layer_of_offset(chunk * T_CHUNK_FLOATS);
    const int local = idx - (t_layer_offset(l) / T_CHUNK_FLOATS) * XPT<NP>::HID_CHUNK;
    const int d = local & 3, lane = (local >> 2) & 63;
    int rest = local >> 8;                                       // flat (rbg, kb, part)
    const int part = rest % NP;
    rest /= NP;
    const int kb = rest & 7, rbg = rest >> 3;                    // rbg = global 16-row block of INPUT features
    const int i = lane & 15, q = lane >> 4;
    const int in = 16 * rbg + i;
    unsigned bits[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int out = x3_feature(kb, q, 2 * d + e);
        float v = 0.f;
        if (l == 0) {
            if (in < EMB_DIM) v = w0[out * EMB_DIM + in];
        } else if (l == 4) {
            if (rbg < 16) v = w4[out * (256 + EMB_DIM) + in];
            else if (in - 256 < EMB_DIM) v = w4[out * (256 + EMB_DIM) + in];
        } else {
            const int hi = (l < 4) ? (l - 1) : (l - 2);
            v = wh[(size_t)hi * 65536 + out * 256 + in];
        }
        unsigned h, m, lo = 0;
        if constexpr (NP == 3) x3_split_pair(v, 0.f, h, m, lo);
        else {
            h2_split_pair(v, 0.f, h, m);
            if (!(fabsf(v) <= 16384.0f)) h = m = 0x7e00u;       // outside the fp16 working range (or NaN): poison, as d3h_sdf_mlp_pack_h2
        }
        bits[e] = (part == 0 ? h : (part == 1 ? m : lo)) & 0xffffu;
    }
    wpackT3[idx] = bits[0] | (bits[1] << 16);
}

// sc[0] = s, sc[1] = 1 / s from the largest |v[i]| (h2_grad_scale); one workgroup, grid-stride.  `bits` (optional): the largest |v| is
// already there as float bits (collected by another kernel's atomicMax)
__global__ __launch_bounds__(256) void h2_scale_from_absmax_kernel(const float* __restrict__ v, int64_t n, const unsigned* __restrict__ bits, float* __restrict__ sc) {
    __shared__ float red[256];
    float m = 0.f;
    if (bits) m = __uint_as_float(*bits);
    else
        for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(v[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float s = h2_grad_scale(red[0]);
        sc[0] = s;
        sc[1] = 1.0f / s;
    }
}

// dz_block_pre / dz_block_inject_pre for a sweep that works on scaled gradients (SCALED: the h2 form): the register copy stays scaled for
// the next layer's product, what goes to memory is un-scaled; the injected term arrives true-scaled
template <bool SCALED>
__device__ __forceinline__ void dzs_plain(f32x4& v, const f32x4 hh, float* dz_l, int rb, int lane, float gsi) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = v[r] * dsoftplus_from_h(hh[r]);
        o[r] = SCALED ? v[r] * gsi : v[r];
    }
    *(f32x4*)(dz_l + (size_t)(rb * 64 + lane) * 4) = o;
}
template <bool SCALED>
__device__ __forceinline__ void dzs_inject(f32x4& v, const f32x4 hh, const f32x4 ee, float* e_l, int rb, int lane, float gs, float gsi) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = SCALED ? fmaf(ee[r], gs, v[r] * dsoftplus_from_h(hh[r])) : (v[r] * dsoftplus_from_h(hh[r]) + ee[r]);
        o[r] = SCALED ? v[r] * gsi : v[r];
    }
    *(f32x4*)(e_l + (size_t)(rb * 64 + lane) * 4) = o;
}

// NP: operand planes (sdf_mlp_x3.h): 3 = bf16 x 3, six products; 2 = fp16 x 2 ("h2"), three products -- wpackT3 is then a pack of
// d3h_sdf_mlp_pack_t_h2 and the sweep works on SCALED gradients: every input gradient (gout, or the injected e) is multiplied by
// s = sc_dev ? sc_dev[0] : sc_imm as it is loaded, everything the sweep stores (dz, dx) by 1 / s (h2_grad_scale; memory stays true-scaled).
template <bool INJECT, int NP>
__global__ __launch_bounds__(NTHREADS, 2) void sdf_mlp_bwd_data_x3_kernel(const float* __restrict__ x, const float* __restrict__ deform, float disp,
                                                                         const float* __restrict__ gout, const float* __restrict__ w7,
                                                                         const unsigned* __restrict__ wpackT3, const float* __restrict__ act,
                                                                         float* __restrict__ dz, float* __restrict__ dx, int64_t n, int ntiles,
                                                                         const int* __restrict__ tile_list, const int* __restrict__ tile_count,
                                                                         const float* __restrict__ sc_dev, float sc_imm) {
    constexpr int X3_HID_CHUNK = XPT<NP>::HID_CHUNK;          // (shadows the bf16 x 3 constant: every chunk offset below is in THIS pack's units)
    const float gs = (NP == 2) ? (sc_dev ? sc_dev[0] : sc_imm) : 1.0f;
    const float gsi = (NP == 2) ? (sc_dev ? sc_dev[1] : 1.0f / sc_imm) : 1.0f;
    __shared__ __attribute__((aligned(16))) unsigned wbuf[2][X3_HID_CHUNK];
    __shared__ __attribute__((aligned(16))) float w7s[NOUT * 256];
    __shared__ __attribute__((aligned(16))) float pfb[NWAVES * (INJECT ? 4 : 2) * 256];
    D3H_X3_CLAIM_SIMD();
    const int n_active = tile_list ? *tile_count : 0;
    const int64_t n16 = tile_list ? (int64_t)n_active : (int64_t)ntiles * 8;
    const int G = (int)gridDim.x;
    const int nrounds = (int)((n16 + NWAVES * (int64_t)G - 1) / (NWAVES * (int64_t)G));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile pointers derived from it live in SGPRs
    const int q = lane >> 4;
    constexpr int N4 = X3_HID_CHUNK / 4;
    constexpr int RS = 8 * NP * X3_FRAG;            // dwords between the two row blocks of a chunk

    for (int j = tid; j < NOUT * 256; j += NTHREADS) w7s[j] = w7[j];
    int pb = 0;
    x3_issue(wpackT3, wbuf[0], N4, tid);
    glds_commit();

    u32x4 Xs[8][NP];
    f32x4 Y[16];

    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t seq = (int64_t)rnd * NWAVES * G + (int64_t)wave * G + blockIdx.x;
        const bool on = seq < n16;                       // wave-uniform
        const int64_t t16 = on ? (tile_list ? (int64_t)tile_list[seq] : seq) : 0;
        const int64_t p = t16 * 16 + (lane & 15);
        const bool valid = on && p < n;
        const float* act_tile = act + t16 * ACT_TILE_FLOATS;
        float* dz_tile = dz + t16 * ACT_TILE_FLOATS;
        float g[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) g[o] = INJECT ? 0.f : (valid ? (gout ? gout[p * NOUT + o] : 1.f) * gs : 0.f);

        // dH_6 = sum_o g_o * W7[o], dZ_6 = dH_6 * softplus'(h_6) (+ e_6), split into the B planes of layer 6's product
#pragma unroll
        for (int rb = 0; rb < 16; ++rb) {
            Y[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                f32x4 w = *(const f32x4*)(w7s + 256 * o + 16 * rb + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) Y[rb][r] = fmaf(g[o], w[r], Y[rb][r]);
            }
            if (on) {
                const size_t off6 = (size_t)6 * ACT_LAYER_FLOATS + (size_t)(rb * 64 + lane) * 4;
                const f32x4 hh6 = *(const f32x4*)(act_tile + off6);
                if (INJECT) dzs_inject<NP == 2>(Y[rb], hh6, *(const f32x4*)(dz_tile + off6), dz_tile + 6 * ACT_LAYER_FLOATS, rb, lane, gs, gsi);
                else dzs_plain<NP == 2>(Y[rb], hh6, dz_tile + 6 * ACT_LAYER_FLOATS, rb, lane, gsi);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xp_split_blocks<NP>(Y[2 * kb], Y[2 * kb + 1], Xs[kb]);

        f32x4 E[4], Elo[4];          // (Elo: the cross-product accumulators of the h2 split, folded in before d(x))
#pragma unroll
        for (int b = 0; b < 4; ++b) E[b] = Elo[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        float* pf = pfb + wave * ((INJECT ? 4 : 2) * 256);
        auto prefetch = [&](int lp, int c) {           // h (and e) of row blocks 2c, 2c + 1 of layer lp: one KiB per wave-instruction
            const size_t o0 = (size_t)lp * ACT_LAYER_FLOATS + (size_t)((2 * c) * 64 + lane) * 4;
            D3H_GLDS16(act_tile + o0, pf);
            D3H_GLDS16(act_tile + o0 + 256, pf + 256);
            if (INJECT) {
                D3H_GLDS16(dz_tile + o0, pf + 512);
                D3H_GLDS16(dz_tile + o0 + 256, pf + 768);
            }
        };
        auto dz_pair = [&](f32x4& v0, f32x4& v1, int lp, int c) {
            const float* pl = pf + lane * 4;
            float* dzl = dz_tile + lp * ACT_LAYER_FLOATS;
            if (INJECT) {
                dzs_inject<NP == 2>(v0, *(const f32x4*)pl, *(const f32x4*)(pl + 512), dzl, 2 * c, lane, gs, gsi);
                dzs_inject<NP == 2>(v1, *(const f32x4*)(pl + 256), *(const f32x4*)(pl + 768), dzl, 2 * c + 1, lane, gs, gsi);
            } else {
                dzs_plain<NP == 2>(v0, *(const f32x4*)pl, dzl, 2 * c, lane, gsi);
                dzs_plain<NP == 2>(v1, *(const f32x4*)(pl + 256), dzl, 2 * c + 1, lane, gsi);
            }
        };
        auto emb_chunks = [&](const unsigned* last) {    // E += (embedding rows of W^T) dZ over two chunks: in-blocks 0,1 | 2,(pad); `last`: the chunk after them
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                x3_issue(c == 0 ? last : last + X3_HID_CHUNK, wbuf[pb ^ 1], N4, tid);
                if (on) {
                    xp_mac_blocks<NP, 8, -1>(E[2 * c], Elo[2 * c], Xs, wbuf[pb], lane, X3None());
                    if (2 * c + 1 < EMB_BLKS) xp_mac_blocks<NP, 8, -1>(E[2 * c + 1], Elo[2 * c + 1], Xs, wbuf[pb] + RS, lane, X3None());
                }
                glds_commit();
                pb ^= 1;
            }
        };
        const unsigned* next = wpackT3 + X3_HID_CHUNK;   // chunk stream pointer (next chunk to prefetch)
#pragma unroll 1
        for (int l = 6; l >= 1; --l) {                   // Xs (= dZ_l) -> Y (= dH_{l-1}, turned into dZ_{l-1} pair by pair)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                x3_issue(next, wbuf[pb ^ 1], N4, tid);
                next += X3_HID_CHUNK;
                if (on) {
                    prefetch(l - 1, c);
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, lo0 = {0.f, 0.f, 0.f, 0.f}, lo1 = {0.f, 0.f, 0.f, 0.f};
                    xp_mac_blocks<NP, 8, -1>(acc0, lo0, Xs, wbuf[pb], lane, X3None());
                    xp_mac_blocks<NP, 8, -1>(acc1, lo1, Xs, wbuf[pb] + RS, lane, X3None());
                    Y[2 * c] = xp_fold<NP>(acc0, lo0);
                    Y[2 * c + 1] = xp_fold<NP>(acc1, lo1);
                }
                glds_commit();                           // vmcnt(0): the weight chunk and the h / e prefetch have landed
                pb ^= 1;
                if (on) dz_pair(Y[2 * c], Y[2 * c + 1], l - 1, c);
            }
            if (l == 4) {                                // skip layer: the embedding columns of net.8 (mlp.py:40-41)
                emb_chunks(next);
                next += 2 * X3_HID_CHUNK;
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) xp_split_blocks<NP>(Y[2 * kb], Y[2 * kb + 1], Xs[kb]);
        }
        // layer 0: Xs = dZ_0;  dEmb += W0^T dZ_0.  After the last chunk of the stream comes chunk 0 of the next tile.
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            x3_issue((c == 0) ? next : wpackT3, wbuf[pb ^ 1], N4, tid);
            if (on) {
                xp_mac_blocks<NP, 8, -1>(E[2 * c], Elo[2 * c], Xs, wbuf[pb], lane, X3None());
                if (2 * c + 1 < EMB_BLKS) xp_mac_blocks<NP, 8, -1>(E[2 * c + 1], Elo[2 * c + 1], Xs, wbuf[pb] + RS, lane, X3None());
            }
            glds_commit();
            pb ^= 1;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) E[b] = xp_fold<NP>(E[b], Elo[b]);

        // d(x) through the positional encoding (embedding.py:33-38)
        if (!INJECT && dx && on) {
            float x0 = 0.f, x1 = 0.f, x2 = 0.f;
            if (valid) {
                x0 = x[3 * p + 0]; x1 = x[3 * p + 1]; x2 = x[3 * p + 2];
                if (deform) {
                    x0 = __fadd_rn(x0, __fmul_rn(disp, deform[3 * p + 0]));
                    x1 = __fadd_rn(x1, __fmul_rn(disp, deform[3 * p + 1]));
                    x2 = __fadd_rn(x2, __fmul_rn(disp, deform[3 * p + 2]));
                }
            }
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int b = 0; b < EMB_BLKS; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int e = 16 * b + 4 * q + r;
                    if (e < EMB_DIM) {
                        float ge = E[b][r];
                        int c;
                        float coef;
                        if (e < 3) { c = e; coef = 1.f; }
                        else {
                            int ep = e - 3;
                            int fr = ep / 6, fn = (ep % 6) / 3;
                            c = ep % 3;
                            float f = (float)(1 << fr);
                            float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
                            coef = fn ? (-f * sinf(f * xc)) : (f * cosf(f * xc));
                        }
                        float t = ge * coef;
                        if (c == 0) d0 += t; else if (c == 1) d1 += t; else d2 += t;
                    }
                }
            d0 += __shfl_xor(d0, 16); d0 += __shfl_xor(d0, 32);
            d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
            d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
            if (valid && q == 0) { dx[3 * p + 0] = d0 * gsi; dx[3 * p + 1] = d1 * gsi; dx[3 * p + 2] = d2 * gsi; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);      // drain the dangling weight prefetch (an LDS write) before the LDS is released
}
// host-side dispatch on the plane count of the transposed pack (3: d3h_sdf_mlp_pack_t3, 2: d3h_sdf_mlp_pack_t_h2)
template <bool INJECT>
static inline void bwd_data_x_launch(int planes, int grid, hipStream_t s, const float* x, const float* deform, float disp, const float* gout, const float* w7,
                                     const unsigned* wpackT, const float* act, float* dz, float* dx, int64_t n, int ntiles, const int* list, const int* cnt,
                                     const float* sc_dev, float sc_imm) {
    if (planes == 2)
        hipLaunchKernelGGL((sdf_mlp_bwd_data_x3_kernel<INJECT, 2>), dim3(grid), dim3(NTHREADS), 0, s, x, deform, disp, gout, w7, wpackT, act, dz, dx, n, ntiles,
                           list, cnt, sc_dev, sc_imm);
    else
        hipLaunchKernelGGL((sdf_mlp_bwd_data_x3_kernel<INJECT, 3>), dim3(grid), dim3(NTHREADS), 0, s, x, deform, disp, gout, w7, wpackT, act, dz, dx, n, ntiles,
                           list, cnt, (const float*)nullptr, 1.0f);
}
#endif  // D3H_MLP_NOUT == 1

// ------------------------------------------------------------------------------------------------
// 2. weight gradients
// ------------------------------------------------------------------------------------------------
constexpr int PITCH = 33;

// NCB = number of 32-wide column blocks handled by the workgroup (4: hidden inputs, 2: the 40 embedding inputs)
// sdf_mlp_bwd_last_kernel: tiles per trip, workgroups.  Atomics that land in one memory channel retire at ~2.6 G/s on this part (the
// 1 KiB of dW7 is one channel): 1024 workgroups x 256 atomics took 100 us; 128 workgroups with 4 tiles in flight: 33 us in total
constexpr int LAST_UNROLL = 4, LAST_GRID = 128;
#ifndef D3H_DW_SPLIT
#define D3H_DW_SPLIT 128
#endif
constexpr int DW_SPLIT = D3H_DW_SPLIT;   // workgroups along the point dimension of sdf_mlp_bwd_dw_kernel (see d3h_sdf_mlp_bwd)
#ifndef D3H_DW_SPLIT_EMB
#define D3H_DW_SPLIT_EMB 128
#endif
#ifndef D3H_DW_SPLIT_SPARSE
#define D3H_DW_SPLIT_SPARSE 32
#endif
// Split-K width of a weight-gradient launch over nt32 32-point groups: every workgroup ends with an atomic flush of its whole partial
// (256 x 128 floats per column chunk and layer), so the flush traffic grows with the split while the MFMA work per workgroup shrinks.
// `cap` (128 dense / embedding, 32 sparse) is what the 50 000-sample and grid-sweep launches were tuned to; below 8 groups per workgroup
// the flush IS the kernel (6 250 samples, one rank's share of 50 000 on 8 GPUs: 236 us at a 128-way split), so small launches split less.
// Which launches of the hidden-layer weight-gradient GEMMs use the bf16-pipe kernel when the bf16-plane packs are given (diagnostic switch
// D3H_DW_X3: 0 none, 1 both (default), 2 only the sweep backward's launch, 3 only the eikonal term's dual-source launch).
// D3H_DW_H2=0: the weight-gradient GEMMs stay on the bf16 x 3 split when the data-backward sweeps run h2 (A/B)
static inline bool dw_h2_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("D3H_DW_H2"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static inline bool dw_x3_enabled(int which) {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("D3H_DW_X3");
        v = e ? atoi(e) : 1;
    }
    return v == 1 || v == which;
}
static inline int dw_split(int nt32, int cap) {
    int s = nt32 / 8;
    if (s < 16) s = 16;
    if (s > cap) s = cap;
    if (s > nt32) s = nt32;
    return s < 1 ? 1 : s;
}
template <int NCB, bool EMB>
__device__ __forceinline__ void sdf_mlp_bwd_dw_body(const float* __restrict__ dz_l /* dz + l*ACT_LAYER */, const float* __restrict__ hsrc /* act + (l-1)*ACT_LAYER */,
                                                    const float* __restrict__ x, const float* __restrict__ deform, float disp,
                                                    int64_t n, int ntiles32, float* __restrict__ dW, int ld, int coloff,
                                                    int ncols, float* __restrict__ db, const float* __restrict__ udir,
                                                    const int* __restrict__ tile_list, const int* __restrict__ tile_count,
                                                    const float* __restrict__ dz2, const float* __restrict__ hsrc2) {
    // dual mode (dz2 != nullptr; eikonal second-order pass): dW += dz_l (x) B1 + dz2 (x) hsrc2 in ONE launch / one atomic flush, where
    // B1 = hsrc (or the tangent embedding when udir is given) and the second pair uses the plain embedding; db sums the second A only
    __shared__ float TA[256 * PITCH];
    __shared__ float TB[NCB * 32 * PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rg = wave & 3;                 // row group: rows rg*64 .. +63 (2 row blocks)
    const int cg = wave >> 2;                // column group
    constexpr int CBW = NCB / 2;             // column blocks per wave
    const int cchunk = blockIdx.y;           // which NCB*32-column chunk of the input features

    f32x16 acc[2][CBW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CBW; ++b) acc[a][b] = (f32x16){0};
    float dbsum = 0.f;

    // sparse mode (see sdf_mlp_active_tiles_kernel): 32-point group t = entries 2t, 2t+1 of the active list; the contraction over
    // points does not care which tiles are paired; an odd tail pairs with zeros
    const int n_active = tile_list ? *tile_count : 0;
    if (tile_list) ntiles32 = (n_active + 1) >> 1;
    auto tile_of = [&](int g16) -> int64_t { return tile_list ? (g16 < n_active ? (int64_t)tile_list[g16] : (int64_t)-1) : (int64_t)g16; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // register-staged pipeline: the global loads of the next 32-point group are in flight during the MFMA phase of the current one
    f32x4 ra[4], rb_[NCB / 2];
    float re[4];
    const int ngroups = dz2 ? 2 * ntiles32 : ntiles32;
    auto issue = [&](int tv) {
        const bool second = dz2 && tv >= ntiles32;
        const int t = second ? tv - ntiles32 : tv;
        const float* __restrict__ asrc = second ? dz2 : dz_l;
        const float* __restrict__ bsrc = second ? hsrc2 : hsrc;
        const float* __restrict__ ud = second ? nullptr : udir;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int u = tid + 512 * r;
            int t2 = u >> 10, rem = u & 1023;
            int64_t tl = tile_of(2 * t + t2);
            ra[r] = tl >= 0 ? *(const f32x4*)(asrc + (size_t)tl * ACT_TILE_FLOATS + 4 * (size_t)rem) : zero4;
        }
        if (EMB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int u = tid + 512 * r;      // 0..2047: 64 (40 real) embedding features x 32 points, recomputed from x
                int e = u >> 5, pt = u & 31;
                int64_t tl = tile_of(2 * t + (pt >> 4));
                int64_t p = tl * 16 + (pt & 15);
                float v = 0.f;
                if (tl >= 0 && p < n && e < EMB_DIM) {
                    float x0 = x[3 * p + 0], x1 = x[3 * p + 1], x2 = x[3 * p + 2];
                    if (deform) {
                        x0 = __fadd_rn(x0, __fmul_rn(disp, deform[3 * p + 0]));
                        x1 = __fadd_rn(x1, __fmul_rn(disp, deform[3 * p + 1]));
                        x2 = __fadd_rn(x2, __fmul_rn(disp, deform[3 * p + 2]));
                    }
                    // udir: the B operand is the tangent embedding J_emb(x) u of the eikonal second-order pass
                    v = ud ? emb_tangent(e, x0, x1, x2, ud[3 * p + 0], ud[3 * p + 1], ud[3 * p + 2]) : emb_feature(e, x0, x1, x2);
                }
                re[r] = v;
            }
        } else {
            const int rb0 = cchunk * NCB * 2;
#pragma unroll
            for (int r = 0; r < NCB / 2; ++r) {
                int u = tid + 512 * r;                       // 0 .. NCB*256-1
                int t2 = u / (NCB * 128), rem = u % (NCB * 128);
                int rbl = rem >> 6, ln = rem & 63;
                int64_t tl = tile_of(2 * t + t2);
                rb_[r] = tl >= 0 ? *(const f32x4*)(bsrc + (size_t)tl * ACT_TILE_FLOATS + 4 * (size_t)((rb0 + rbl) * 64 + ln)) : zero4;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int u = tid + 512 * r;
            int t2 = u >> 10, rem = u & 1023;
            int ln = rem & 63, rbk = rem >> 6;
            int f = 16 * rbk + 4 * (ln >> 4), pt = 16 * t2 + (ln & 15);
#pragma unroll
            for (int k = 0; k < 4; ++k) TA[(f + k) * PITCH + pt] = ra[r][k];
        }
        if (EMB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int u = tid + 512 * r;
                TB[(u >> 5) * PITCH + (u & 31)] = re[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NCB / 2; ++r) {
                int u = tid + 512 * r;
                int t2 = u / (NCB * 128), rem = u % (NCB * 128);
                int rbl = rem >> 6, ln = rem & 63;
                int f = 16 * rbl + 4 * (ln >> 4), pt = 16 * t2 + (ln & 15);
#pragma unroll
                for (int k = 0; k < 4; ++k) TB[(f + k) * PITCH + pt] = rb_[r][k];
            }
        }
    };
    int t = blockIdx.x;
    if (t >= ngroups) return;             // (block-uniform) nothing to add: skip the zero-valued atomic flush
    issue(t);
    for (; t < ngroups; t += gridDim.x) {
        commit();
        __syncthreads();
        if (t + (int)gridDim.x < ngroups) issue(t + gridDim.x);
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            float a0 = TA[((rg * 2 + 0) * 32 + i) * PITCH + 2 * s + h];
            float a1 = TA[((rg * 2 + 1) * 32 + i) * PITCH + 2 * s + h];
#pragma unroll
            for (int b = 0; b < CBW; ++b) {
                float bv = TB[((cg * CBW + b) * 32 + i) * PITCH + 2 * s + h];
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][b], 0, 0, 0);
            }
        }
        if (db && cchunk == 0 && tid < 256 && (!dz2 || t >= ntiles32)) {
            float sm = 0.f;
#pragma unroll 8
            for (int pt = 0; pt < 32; ++pt) sm += TA[tid * PITCH + pt];
            dbsum += sm;
        }
        __syncthreads();
    }
    // accumulate into dW[out][coloff + col]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CBW; ++b) {
            int col = cchunk * NCB * 32 + (cg * CBW + b) * 32 + i;
            if (col < ncols) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = (rg * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    atomicAdd(&dW[(size_t)row * ld + coloff + col], acc[a][b][r]);
#ifdef D3H_DW_DOUBLEFLUSH      // timing experiment: the flush a second time with zeros (same results) -- the difference is what it costs
                    atomicAdd(&dW[(size_t)row * ld + coloff + col], acc[a][b][r] * 0.f);
#endif
                }
            }
        }
    if (db && cchunk == 0 && tid < 256) atomicAdd(&db[tid], dbsum);
}

// The two embedding weight gradients of one backward -- layer 0 (39 inputs) and the skip layer's embedding columns -- as ONE launch: blockIdx.z
// picks the argument set.  Each of them alone fills half of the chip (<= 128 workgroups of 512 threads), back to back they took 2 x 70 us on the
// eikonal term's 50 000 samples ("fold the embedding weight-gradient launches").
struct DwEmbArgs {
    const float* dz_l; const float* hsrc; float* dW; int ld, coloff, ncols; float* db; const float* udir; const float* dz2; const float* hsrc2;
};
__global__ __launch_bounds__(512) void sdf_mlp_bwd_dw_emb_pair_kernel(DwEmbArgs a0, DwEmbArgs a1, const float* __restrict__ x, const float* __restrict__ deform,
                                                                       float disp, int64_t n, int ntiles32, const int* __restrict__ tile_list,
                                                                       const int* __restrict__ tile_count) {
    const DwEmbArgs a = blockIdx.z ? a1 : a0;
    sdf_mlp_bwd_dw_body<2, true>(a.dz_l, a.hsrc, x, deform, disp, n, ntiles32, a.dW, a.ld, a.coloff, a.ncols, a.db, a.udir, tile_list, tile_count, a.dz2,
                                 a.hsrc2);
}

template <int NCB, bool EMB>
__global__ __launch_bounds__(512) void sdf_mlp_bwd_dw_kernel(const float* __restrict__ dz_l, const float* __restrict__ hsrc, const float* __restrict__ x,
                                                             const float* __restrict__ deform, float disp, int64_t n, int ntiles32,
                                                             float* __restrict__ dW, int ld, int coloff, int ncols, float* __restrict__ db,
                                                             const float* __restrict__ udir, const int* __restrict__ tile_list,
                                                             const int* __restrict__ tile_count, const float* __restrict__ dz2,
                                                             const float* __restrict__ hsrc2) {
    sdf_mlp_bwd_dw_body<NCB, EMB>(dz_l, hsrc, x, deform, disp, n, ntiles32, dW, ld, coloff, ncols, db, udir, tile_list, tile_count, dz2, hsrc2);
}

// The six hidden-layer weight gradients (layers 1..6: 256 x 256, layer 4's first 256 columns) in ONE launch, blockIdx.z = layer - 1.
// Per layer the grid is S x 2 workgroups = one per CU; six layers together keep two workgroups resident per CU, so the load /
// transpose phase of one overlaps the MFMA phase of another without the doubled atomic flush that a wider split-K costs.
// a_base / b_base: layer-major activation-shaped buffers; A_l = a_base + l * ACT_LAYER, B_l = b_base + (l - 1) * ACT_LAYER; the optional
// second pair (a2_base, b2_base) likewise (dual mode of the eikonal pass).
__global__ __launch_bounds__(512) void sdf_mlp_bwd_dw_layers_kernel(const float* __restrict__ a_base, const float* __restrict__ b_base,
                                                                    const float* __restrict__ x, int64_t n, int ntiles32,
                                                                    float* __restrict__ dwh, float* __restrict__ dbh, float* __restrict__ dw4,
                                                                    float* __restrict__ db4, const int* __restrict__ tile_list,
                                                                    const int* __restrict__ tile_count, const float* __restrict__ a2_base,
                                                                    const float* __restrict__ b2_base) {
    const int l = blockIdx.z + 1;
    const int hi = (l < 4) ? (l - 1) : (l - 2);
    float* dW = (l == 4) ? dw4 : dwh + (size_t)hi * 65536;
    float* db = (l == 4) ? db4 : dbh + hi * 256;
    const int ld = (l == 4) ? 256 + EMB_DIM : 256;
    sdf_mlp_bwd_dw_body<4, false>(a_base + (size_t)l * ACT_LAYER_FLOATS, b_base + (size_t)(l - 1) * ACT_LAYER_FLOATS, x, nullptr, 0.f, n, ntiles32,
                                  dW, ld, 0, 256, db, nullptr, tile_list, tile_count,
                                  a2_base ? a2_base + (size_t)l * ACT_LAYER_FLOATS : nullptr,
                                  b2_base ? b2_base + (size_t)(l - 1) * ACT_LAYER_FLOATS : nullptr);
}

#if D3H_MLP_NOUT == 1
// ------------------------------------------------------------------------------------------------
// 2b. the six hidden-layer weight gradients on the bf16 matrix pipe (sdf_mlp_x3.h)
// ------------------------------------------------------------------------------------------------
// dW_l[256 x 256] += dZ_l^T[256 x pts] H_{l-1}[pts x 256] with the contraction over the POINTS on v_mfma_f32_32x32x16_bf16: one MFMA takes a
// whole 16-point tile as its k-steps (lane i + 32 h holds points 8 h .. 8 h + 7 of feature row / column i), six products per 32 x 32 block as in
// the sweeps.  The tile-packed sources hold four FEATURES of one point per lane, the operands need eight POINTS of one feature: the
// transposition goes through LDS as before, but as bf16 planes -- every lane swaps two values with its neighbour point (lane ^ 1), splits
// two (point j, point j + 1) pairs into the three planes and writes six dwords; rows of 8 pair-words are padded to 12 (b128 fragment reads
// of 16 rows hit 16 disjoint bank quads).  54 KB of LDS: two workgroups stay resident per CU, one transposing while the other multiplies,
// as with the f32 kernel.  db_l comes from the same operands: one more MFMA per plane against a column of ones.
constexpr int DWX_PITCH = 12;
#ifndef D3H_EMULATED
#define D3H_MFMA32_BF16X8(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(d3h_bf16x8, a), __builtin_bit_cast(d3h_bf16x8, b), c, 0, 0, 0)
#define D3H_MFMA32_F16X8(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(d3h_f16x8, a), __builtin_bit_cast(d3h_f16x8, b), c, 0, 0, 0)
#else
#define D3H_MFMA32_BF16X8(a, b, c) emul::mfma_32x32x16bf16(a, b, c)
#define D3H_MFMA32_F16X8(a, b, c) emul::mfma_32x32x16f16(a, b, c)
#endif

// four features f0 .. f0 + 3 of point j (one tile-packed f32x4) -> the three bf16 planes of T[plane][feature][pair word j / 2]
__device__ __forceinline__ void dwx_put(unsigned* T, int nrows, int f0, int j, const f32x4 v) {
    const bool odd = j & 1;
    const float ma = odd ? v[2] : v[0], mb = odd ? v[3] : v[1];          // the two features this lane writes (even lane: 0, 1; odd lane: 2, 3)
    const float xa = __shfl_xor(odd ? v[0] : v[2], 1), xb = __shfl_xor(odd ? v[1] : v[3], 1);      // the same features of the neighbour point
    const int f = f0 + (odd ? 2 : 0), w = j >> 1;
    unsigned h, m, lo;
    x3_split_pair(odd ? xa : ma, odd ? ma : xa, h, m, lo);                // (earlier point, later point)
    T[(0 * nrows + f) * DWX_PITCH + w] = h;
    T[(1 * nrows + f) * DWX_PITCH + w] = m;
    T[(2 * nrows + f) * DWX_PITCH + w] = lo;
    x3_split_pair(odd ? xb : mb, odd ? mb : xb, h, m, lo);
    T[(0 * nrows + f + 1) * DWX_PITCH + w] = h;
    T[(1 * nrows + f + 1) * DWX_PITCH + w] = m;
    T[(2 * nrows + f + 1) * DWX_PITCH + w] = lo;
}

// grid (S, 2, 6): blockIdx.z = layer - 1, blockIdx.y = 128-column chunk of the input features, blockIdx.x strides over the 16-point tiles
// (the active list in sparse mode; both sources one after the other in dual mode: dz (x) b, then dz2 (x) b2, db from the second only)
__global__ __launch_bounds__(512) void sdf_mlp_bwd_dw_layers_x3_kernel(const float* __restrict__ a_base, const float* __restrict__ b_base, int ntiles16,
                                                                       float* __restrict__ dwh, float* __restrict__ dbh, float* __restrict__ dw4,
                                                                       float* __restrict__ db4, const int* __restrict__ tile_list,
                                                                       const int* __restrict__ tile_count, const float* __restrict__ a2_base,
                                                                       const float* __restrict__ b2_base) {
    constexpr int NP = 3;
#ifndef D3H_DWX_PIPE
#define D3H_DWX_PIPE 0
#endif
    // D3H_DWX_PIPE = 1: TWO LDS images; the transposition of tile t + 1 is interleaved, instruction by instruction, with the MFMAs of tile t (one
    // barrier per tile instead of two; sched_group_barrier: one MFMA, then four VALU and one LDS instruction of the put) -- 108 KB of LDS
    constexpr int NIMG = D3H_DWX_PIPE ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned TA3s[NIMG][NP * 256 * DWX_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned TB3s[NIMG][NP * 128 * DWX_PITCH];
    unsigned* TA3 = TA3s[0];
    unsigned* TB3 = TB3s[0];
#ifdef D3H_DWX_PROBE_LDSPAD      // (diagnostic: a larger LDS footprint, so that fewer / no other workgroups share the CU)
    __shared__ volatile unsigned ldspad[D3H_DWX_PROBE_LDSPAD / 4];
    ldspad[threadIdx.x * 16 % (D3H_DWX_PROBE_LDSPAD / 4)] = threadIdx.x;
#endif
    // The kernel needs 180 VGPRs and claims all 256: see THE CO-RESIDENCY RULE in sdf_mlp_x3.h -- a foreign wave sharing a SIMD with a wave that issues
    // bf16 MFMAs gets wrong packed-f32 results (this kernel, on the eikonal side stream, was where it was first seen: lbs_bwd_kernel's 3x3 algebra
    // on the main stream).  The __syncthreads() that closes every tile below is part of the rule (no wave ends while a sibling still multiplies).
    D3H_X3_CLAIM_SIMD();
    const int l = blockIdx.z + 1;
    const int hi = (l < 4) ? (l - 1) : (l - 2);
    float* dW = (l == 4) ? dw4 : dwh + (size_t)hi * 65536;
    float* db = (l == 4) ? db4 : dbh + hi * 256;
    const int ld = (l == 4) ? 256 + EMB_DIM : 256;
    const float* dz_l = a_base + (size_t)l * ACT_LAYER_FLOATS;
    const float* hsrc = b_base + (size_t)(l - 1) * ACT_LAYER_FLOATS;
    const float* dz2 = a2_base ? a2_base + (size_t)l * ACT_LAYER_FLOATS : nullptr;
    const float* hsrc2 = b2_base ? b2_base + (size_t)(l - 1) * ACT_LAYER_FLOATS : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int rg = wave & 3, cg = wave >> 2;      // rows rg * 64 .. + 63 (two 32-row blocks), columns cg * 64 .. + 63 of this workgroup's 128
    const int cchunk = blockIdx.y;
    const bool want_db = cchunk == 0 && cg == 0;  // (wave-uniform) these four waves cover the 256 rows once

    f32x16 acc[2][2], accdb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        accdb[a] = (f32x16){0};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16){0};
    }
    const int n16 = tile_list ? *tile_count : ntiles16;
    const int ngroups = dz2 ? 2 * n16 : n16;

    // register-staged pipeline: the global loads of the next tile are in flight during the MFMA phase of the current one
    f32x4 ra[2], rbv;
    auto issue = [&](int tv) {
        const bool second = dz2 && tv >= n16;
        const int t = second ? tv - n16 : tv;
        const int64_t tl = tile_list ? (int64_t)tile_list[t] : (int64_t)t;
        const float* __restrict__ asrc = (second ? dz2 : dz_l) + (size_t)tl * ACT_TILE_FLOATS;
        const float* __restrict__ bsrc = (second ? hsrc2 : hsrc) + (size_t)tl * ACT_TILE_FLOATS;
        ra[0] = *(const f32x4*)(asrc + 4 * (size_t)tid);
        ra[1] = *(const f32x4*)(asrc + 4 * (size_t)(tid + 512));
        rbv = *(const f32x4*)(bsrc + 4 * (size_t)(cchunk * 512 + tid));
    };
    int t = blockIdx.x;
    if (t >= ngroups) return;             // (block-uniform) nothing to add: skip the zero-valued atomic flush
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    issue(t);
#ifdef D3H_DWX_PROBE_NOWORK
    t = ngroups;
#endif
#if D3H_DWX_PIPE && !defined(D3H_EMULATED)
    {
        int cur = 0;
        dwx_put(TA3s[0], 256, 16 * wave + 4 * (lane >> 4), lane & 15, ra[0]);
        dwx_put(TA3s[0], 256, 16 * (wave + 8) + 4 * (lane >> 4), lane & 15, ra[1]);
        dwx_put(TB3s[0], 128, 16 * wave + 4 * (lane >> 4), lane & 15, rbv);
        __syncthreads();
        if (t + (int)gridDim.x < ngroups) issue(t + gridDim.x);
        for (; t < ngroups; t += gridDim.x) {
            const bool bias_now = want_db && db && (!dz2 || t >= n16);
            const bool more = t + (int)gridDim.x < ngroups;
            const unsigned* TAc = TA3s[cur];
            const unsigned* TBc = TB3s[cur];
            u32x4 A[2][3], B[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    A[a][pl] = *(const u32x4*)(TAc + (pl * 256 + (rg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
                    B[a][pl] = *(const u32x4*)(TBc + (pl * 128 + (cg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
                }
            // The next tile's transposition (its global loads were issued a whole tile ago) goes into the other image, interleaved with this
            // tile's MFMAs: ONE basic block per variant (after the last tile the put writes an image nobody reads -- unconditional on purpose: a
            // branch would split the block and the scheduler could not interleave), per MFMA five VALU instructions and one LDS write of the put
            auto body = [&](auto with_bias) {
                dwx_put(TA3s[cur ^ 1], 256, 16 * wave + 4 * (lane >> 4), lane & 15, ra[0]);
                dwx_put(TA3s[cur ^ 1], 256, 16 * (wave + 8) + 4 * (lane >> 4), lane & 15, ra[1]);
                dwx_put(TB3s[cur ^ 1], 128, 16 * wave + 4 * (lane >> 4), lane & 15, rbv);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x16 c = acc[a][b];
                        c = D3H_MFMA32_BF16X8(A[a][2], B[b][0], c);
                        c = D3H_MFMA32_BF16X8(A[a][0], B[b][2], c);
                        c = D3H_MFMA32_BF16X8(A[a][1], B[b][1], c);
                        c = D3H_MFMA32_BF16X8(A[a][1], B[b][0], c);
                        c = D3H_MFMA32_BF16X8(A[a][0], B[b][1], c);
                        c = D3H_MFMA32_BF16X8(A[a][0], B[b][0], c);
                        acc[a][b] = c;
                    }
                    if (decltype(with_bias)::value) {
                        accdb[a] = D3H_MFMA32_BF16X8(A[a][2], ones, accdb[a]);
                        accdb[a] = D3H_MFMA32_BF16X8(A[a][1], ones, accdb[a]);
                        accdb[a] = D3H_MFMA32_BF16X8(A[a][0], ones, accdb[a]);
                    }
                }
#pragma unroll
                for (int k = 0; k < (decltype(with_bias)::value ? 30 : 24); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            };
            (void)more;
            if (bias_now) body(std::true_type{}); else body(std::false_type{});
            __syncthreads();
            if (t + 2 * (int)gridDim.x < ngroups) issue(t + 2 * gridDim.x);
            cur ^= 1;
        }
    }
#else
    for (; t < ngroups; t += gridDim.x) {
        // tile-packed element u = rb * 64 + lane: features 16 rb + 4 (lane >> 4) + 0..3 of point lane & 15
#ifndef D3H_DWX_PROBE_NOPUT       // (diagnostic builds, results wrong: which part of the loop disturbs co-resident waves; see D3H_X3_CLAIM_SIMD)
        dwx_put(TA3, 256, 16 * wave + 4 * (lane >> 4), lane & 15, ra[0]);
        dwx_put(TA3, 256, 16 * (wave + 8) + 4 * (lane >> 4), lane & 15, ra[1]);
        dwx_put(TB3, 128, 16 * wave + 4 * (lane >> 4), lane & 15, rbv);
#endif
        const bool bias_now = want_db && db && (!dz2 || t >= n16);
        __syncthreads();
        if (t + (int)gridDim.x < ngroups) issue(t + gridDim.x);
        u32x4 A[2][NP], B[2][NP];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                A[a][pl] = *(const u32x4*)(TA3 + (pl * 256 + (rg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
                B[a][pl] = *(const u32x4*)(TB3 + (pl * 128 + (cg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
            }
#ifdef D3H_DWX_PROBE_NOMFMA
        acc[0][0][0] += __uint_as_float(A[0][0][0] ^ B[0][0][0] ^ A[1][NP - 1][3] ^ B[1][NP - 1][3]) * 0.f;
#else
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x16 c = acc[a][b];
                c = D3H_MFMA32_BF16X8(A[a][2], B[b][0], c);
                c = D3H_MFMA32_BF16X8(A[a][0], B[b][2], c);
                c = D3H_MFMA32_BF16X8(A[a][1], B[b][1], c);
                c = D3H_MFMA32_BF16X8(A[a][1], B[b][0], c);
                c = D3H_MFMA32_BF16X8(A[a][0], B[b][1], c);
                c = D3H_MFMA32_BF16X8(A[a][0], B[b][0], c);
                acc[a][b] = c;
            }
            if (bias_now) {
                accdb[a] = D3H_MFMA32_BF16X8(A[a][2], ones, accdb[a]);
                accdb[a] = D3H_MFMA32_BF16X8(A[a][1], ones, accdb[a]);
                accdb[a] = D3H_MFMA32_BF16X8(A[a][0], ones, accdb[a]);
            }
        }
#endif
        __syncthreads();
    }
#endif
#ifdef D3H_DWX_PROBE_NOFLUSH
    if (acc[0][0][0] != 12345.678f) return;
#endif
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = cchunk * 128 + (cg * 2 + b) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (rg * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(&dW[(size_t)row * ld + col], acc[a][b][r]);
            }
        }
        if (want_db && db && i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(&db[(rg * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h], accdb[a][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2c. the same six weight gradients on the fp16 matrix pipe, TWO 16-point tiles per barrier period (round 6)
// ------------------------------------------------------------------------------------------------
// Why another kernel: beside the render the eikonal chain gets a CU budget (geometry/hmsdf.py:_eikonal_async), i.e. ~13 x 12 workgroups for the
// dual-source launch -- and a workgroup of sdf_mlp_bwd_dw_layers_x3_kernel keeps ONE tile (24 KB) of loads in flight: 24 KB per ~2 us of memory
// latency = 12 GB/s per workgroup, which is what it ran at (profiles/r6_bench_config3_detail.json: 844 us for 1.8 GB on ~156 CUs).  Here a
// workgroup has two tiles in flight and one barrier pair per two tiles.  Arithmetic: every operand is multiplied by a power of two as it is
// transposed -- a gradient-valued source (sc_mask, as in the x3 kernel) by s = sc_dev ? sc_dev[0] : sc_imm, the others by 2^6 -- and split into
// fp16(v) + fp16(v - fp16(v)): with operands of magnitude 1 .. 10^3 the UNSCALED residual is a normal fp16 number, so the three products
// h.h + h.m + m.h go into ONE accumulator (no second set for the cross terms as in the forward-type sweeps, which must stay exact for operands of
// 10^-2 and keep the residual plane scaled); the sums are multiplied by 1 / (64 s) at the flush.  Products a_m b_m are dropped: 2^-22 relative.
constexpr float DWH_US = 64.0f;
__device__ __forceinline__ void dwh_split_pair(float a, float b, unsigned& h, unsigned& m) {
    h = h2_pk(a, b);
    m = h2_pk(a - h2_lo(h), b - h2_hi(h));
}
// four features f0 .. f0 + 3 of point j (one tile-packed f32x4), times sc -> T[plane 2][feature][pair word j / 2]
__device__ __forceinline__ void dwh_put(unsigned* T, int nrows, int f0, int j, f32x4 v, float sc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= sc;
    const bool odd = j & 1;
    const float ma = odd ? v[2] : v[0], mb = odd ? v[3] : v[1];
    const float xa = __shfl_xor(odd ? v[0] : v[2], 1), xb = __shfl_xor(odd ? v[1] : v[3], 1);
    const int f = f0 + (odd ? 2 : 0), w = j >> 1;
    unsigned h, m;
    dwh_split_pair(odd ? xa : ma, odd ? ma : xa, h, m);                // (earlier point, later point)
    T[(0 * nrows + f) * DWX_PITCH + w] = h;
    T[(1 * nrows + f) * DWX_PITCH + w] = m;
    dwh_split_pair(odd ? xb : mb, odd ? mb : xb, h, m);
    T[(0 * nrows + f + 1) * DWX_PITCH + w] = h;
    T[(1 * nrows + f + 1) * DWX_PITCH + w] = m;
}

__global__ __launch_bounds__(512) void sdf_mlp_bwd_dw_layers_h2_kernel(const float* __restrict__ a_base, const float* __restrict__ b_base, int ntiles16,
                                                                       float* __restrict__ dwh, float* __restrict__ dbh, float* __restrict__ dw4,
                                                                       float* __restrict__ db4, const int* __restrict__ tile_list,
                                                                       const int* __restrict__ tile_count, const float* __restrict__ a2_base,
                                                                       const float* __restrict__ b2_base, const float* __restrict__ sc_dev, float sc_imm,
                                                                       int sc_mask) {
    __shared__ __attribute__((aligned(16))) unsigned TAs[2][2 * 256 * DWX_PITCH];          // [tile slot][plane][feature row][pair word]
    __shared__ __attribute__((aligned(16))) unsigned TBs[2][2 * 128 * DWX_PITCH];
    D3H_X3_CLAIM_SIMD();          // THE CO-RESIDENCY RULE (sdf_mlp_x3.h): all 256 VGPRs, and the barrier that closes every period is after the last MFMA
    const float gs = sc_dev ? sc_dev[0] : sc_imm;
    const float gsi = sc_dev ? sc_dev[1] : 1.0f / sc_imm;
    const int l = blockIdx.z + 1;
    const int hi = (l < 4) ? (l - 1) : (l - 2);
    float* dW = (l == 4) ? dw4 : dwh + (size_t)hi * 65536;
    float* db = (l == 4) ? db4 : dbh + hi * 256;
    const int ld = (l == 4) ? 256 + EMB_DIM : 256;
    const float* dz_l = a_base + (size_t)l * ACT_LAYER_FLOATS;
    const float* hsrc = b_base + (size_t)(l - 1) * ACT_LAYER_FLOATS;
    const float* dz2 = a2_base ? a2_base + (size_t)l * ACT_LAYER_FLOATS : nullptr;
    const float* hsrc2 = b2_base ? b2_base + (size_t)(l - 1) * ACT_LAYER_FLOATS : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int rg = wave & 3, cg = wave >> 2;
    const int cchunk = blockIdx.y;
    const bool want_db = cchunk == 0 && cg == 0;

    f32x16 acc[2][2], accdb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        accdb[a] = (f32x16){0};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16){0};
    }
    const int n16 = tile_list ? *tile_count : ntiles16;
    const int ngroups = dz2 ? 2 * n16 : n16;
    const int G2 = 2 * (int)gridDim.x;
    int t = 2 * (int)blockIdx.x;
    if (t >= ngroups) return;             // (block-uniform) nothing to add: skip the zero-valued atomic flush

    f32x4 ra[2][2], rb[2];
    float sca[2], scb[2];
    auto issue = [&](int slot, int tv) {          // (tv < ngroups)
        const bool second = dz2 && tv >= n16;
        const int tt = second ? tv - n16 : tv;
        const int64_t tl = tile_list ? (int64_t)tile_list[tt] : (int64_t)tt;
        const float* __restrict__ asrc = (second ? dz2 : dz_l) + (size_t)tl * ACT_TILE_FLOATS;
        const float* __restrict__ bsrc = (second ? hsrc2 : hsrc) + (size_t)tl * ACT_TILE_FLOATS;
        ra[slot][0] = *(const f32x4*)(asrc + 4 * (size_t)tid);
        ra[slot][1] = *(const f32x4*)(asrc + 4 * (size_t)(tid + 512));
        rb[slot] = *(const f32x4*)(bsrc + 4 * (size_t)(cchunk * 512 + tid));
        sca[slot] = ((sc_mask >> (second ? 2 : 0)) & 1) ? gs : DWH_US;
        scb[slot] = ((sc_mask >> (second ? 3 : 1)) & 1) ? gs : DWH_US;
    };
    const u32x4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};          // (1, 1) as two fp16
    issue(0, t);
    if (t + 1 < ngroups) issue(1, t + 1);
    for (; t < ngroups; t += G2) {
        const bool two = t + 1 < ngroups;                 // (block-uniform)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl == 1 && !two) break;
            dwh_put(TAs[sl], 256, 16 * wave + 4 * (lane >> 4), lane & 15, ra[sl][0], sca[sl]);
            dwh_put(TAs[sl], 256, 16 * (wave + 8) + 4 * (lane >> 4), lane & 15, ra[sl][1], sca[sl]);
            dwh_put(TBs[sl], 128, 16 * wave + 4 * (lane >> 4), lane & 15, rb[sl], scb[sl]);
        }
        // db_l = sum of the SECOND source's dZ^ in dual mode, of the only source otherwise; its operand scale is sca of that tile (the flagged one)
        const bool bias0 = want_db && db && (!dz2 || t >= n16);
        const bool bias1 = want_db && db && two && (!dz2 || t + 1 >= n16);
        __syncthreads();
        if (t + G2 < ngroups) {
            issue(0, t + G2);
            if (t + G2 + 1 < ngroups) issue(1, t + G2 + 1);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl == 1 && !two) break;
            u32x4 A[2][2], B[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    A[a][pl] = *(const u32x4*)(TAs[sl] + (pl * 256 + (rg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
                    B[a][pl] = *(const u32x4*)(TBs[sl] + (pl * 128 + (cg * 2 + a) * 32 + i) * DWX_PITCH + 4 * h);
                }
            const bool bias_now = sl == 0 ? bias0 : bias1;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    c = D3H_MFMA32_F16X8(A[a][1], B[b][0], c);
                    c = D3H_MFMA32_F16X8(A[a][0], B[b][1], c);
                    c = D3H_MFMA32_F16X8(A[a][0], B[b][0], c);
                    acc[a][b] = c;
                }
                if (bias_now) {
                    accdb[a] = D3H_MFMA32_F16X8(A[a][1], ones, accdb[a]);
                    accdb[a] = D3H_MFMA32_F16X8(A[a][0], ones, accdb[a]);
                }
            }
        }
        __syncthreads();
    }
    // every product carries (flagged source: s) x (other source: 2^6); the bias sums carry the scale of their (A) source alone: s when it is
    // flagged (dual mode's second pair, the sparse sweep's only pair), 2^6 otherwise
    const float wsc = gsi * (1.0f / DWH_US);
    const bool a_flag = dz2 ? ((sc_mask >> 2) & 1) : (sc_mask & 1);
    const float bsc = a_flag ? gsi : (1.0f / DWH_US);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = cchunk * 128 + (cg * 2 + b) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (rg * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(&dW[(size_t)row * ld + col], acc[a][b][r] * wsc);
            }
        }
        if (want_db && db && i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(&db[(rg * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h], accdb[a][r] * bsc);
        }
    }
}
#endif  // D3H_MLP_NOUT == 1

// ------------------------------------------------------------------------------------------------
// 3. head: dW7[f] = sum_p g[p] h6[p][f], db7 = sum_p g[p]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sdf_mlp_bwd_last_kernel(const float* __restrict__ gout, const float* __restrict__ act6, int64_t n,
                                                               int ntiles16, float* __restrict__ dW7, float* __restrict__ db7,
                                                               const int* __restrict__ tile_list, const int* __restrict__ tile_count) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tile_list) ntiles16 = *tile_count;
    const int q = lane >> 4;
    const int o = blockIdx.y;                 // head output (NOUT = 1 for the SDF network)
    dW7 += 256 * o;
    if (db7) db7 += o;
    f32x4 part[4];                       // row blocks rb = wave + 4 a
#pragma unroll
    for (int a = 0; a < 4; ++a) part[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;
    // LAST_UNROLL tiles per trip: their loads are independent, so a workgroup keeps several KiB in flight (the grid is kept small
    // because every workgroup ends with 256 atomics on the same 1 KiB of dW7: 1024 workgroups spent 100 us in that flush alone)
    for (int ti0 = blockIdx.x * LAST_UNROLL; ti0 < ntiles16; ti0 += gridDim.x * LAST_UNROLL) {
        float g[LAST_UNROLL];
        f32x4 hh[LAST_UNROLL][4];
#pragma unroll
        for (int u = 0; u < LAST_UNROLL; ++u) {
            const int ti = ti0 + u;
            const bool live = ti < ntiles16;
            const int64_t t = live ? (tile_list ? (int64_t)tile_list[ti] : (int64_t)ti) : 0;
            int64_t p = t * 16 + (lane & 15);
            g[u] = (live && p < n) ? (gout ? gout[p * NOUT + o] : 1.f) : 0.f;      // gout == nullptr: plain column sums (eikonal pass)
            const float* base = act6 + (size_t)t * ACT_TILE_FLOATS;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                hh[u][a] = live ? *(const f32x4*)(base + (size_t)((wave + 4 * a) * 64 + lane) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < LAST_UNROLL; ++u) {
            if (wave == 0 && q == 0) gsum += g[u];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) part[a][k] = fmaf(g[u], hh[u][a][k], part[a][k]);
        }
    }
    // reduce over the 16 points (lanes with equal q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = part[a][k];
            for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if ((lane & 15) == 0) atomicAdd(&dW7[16 * (wave + 4 * a) + 4 * q + k], v);
        }
    for (int m = 8; m >= 1; m >>= 1) gsum += __shfl_xor(gsum, m);
    if (tid == 0 && db7) atomicAdd(db7, gsum);
}

#if D3H_MLP_NOUT == 1
// ------------------------------------------------------------------------------------------------
// 4. eikonal loss on the gradient field: sum (|g| - 1)^2 and u = scale * d/dg (|g| - 1)^2 = scale * 2 (|g| - 1) g / |g|
//    (|g| = sqrt(sum g^2) as hmsdf.py:875 writes it; g = 0 gives NaN exactly as torch's sqrt backward does)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eikonal_loss_kernel(const float* __restrict__ g, int64_t n, float scale, float* __restrict__ loss_sum,
                                                           float* __restrict__ u) {
    __shared__ float s4[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a = g[3 * i], b = g[3 * i + 1], c = g[3 * i + 2];
        float nrm = sqrtf(a * a + b * b + c * c);
        float d = nrm - 1.0f;
        acc += d * d;
        if (u) {
            float k = scale * 2.0f * d / nrm;
            u[3 * i] = k * a; u[3 * i + 1] = k * b; u[3 * i + 2] = k * c;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s4[0] + s4[1] + s4[2] + s4[3]);
}

#endif

}  // namespace

#if D3H_MLP_NOUT == 1
int d3h_sdf_mlp_fwd_x3_list_launch(const float* x, const float* deform, float disp, const unsigned* wpack3, int planes, float* act, int64_t n,
                                   const int* tile_list, const int* tile_count, hipStream_t s);
extern "C" int d3h_sdf_mlp_fwd_x3(const float* x, const float* deform, float disp, const unsigned* wpack3, float* sdf, float* xdef, float* act,
                                  int64_t n, int max_cus, void* stream);
extern "C" int d3h_sdf_mlp_fwd_h2(const float* x, const float* deform, float disp, const unsigned* wpackh2, float* sdf, float* xdef, float* act,
                                  int64_t n, int max_cus, void* stream);
#endif
static inline int64_t bwd_r4(int64_t v) { return (v + 3) & ~(int64_t)3; }           // every sub-array of the scratch starts 16-byte aligned
static inline int64_t bwd_p16(int64_t n) { return (n + 15) & ~(int64_t)15; }        // the gathered arrays are padded to whole 16-point tiles

// ints of the tile_list scratch of d3h_sdf_mlp_bwd for n points (covers both the position-tile list and the compact form)
extern "C" int64_t d3h_sdf_mlp_bwd_scratch_ints(int64_t n) { return 8 + bwd_r4(n) + bwd_r4(n / 16 + 2) + 7 * bwd_p16(n) + 64; }

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int64_t d3h_sdf_mlp_wpackt_floats(void) { return WPACKT_FLOATS; }

extern "C" int d3h_sdf_mlp_pack_t(const float* w0, const float* wh, const float* w4, float* wpackT, void* stream) {
    if (!w0 || !wh || !w4 || !wpackT) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack_t_kernel, dim3(d3h_cdiv(WPACKT_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, w0, wh, w4, wpackT);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

#if D3H_MLP_NOUT == 1
// dwords of a wpackT3 buffer (d3h_sdf_mlp_pack_t3)
extern "C" int64_t d3h_sdf_mlp_wpackt3_dwords(void) { return X3_WPACKT_DWORDS; }

// wpackT3 [d3h_sdf_mlp_wpackt3_dwords()] (overwritten) = the transposed weights of d3h_sdf_mlp_pack_t, each as three bf16 planes in the
// fragment order of the bf16-pipe data-backward sweeps (sdf_mlp_x3.h)
extern "C" int d3h_sdf_mlp_pack_t3(const float* w0, const float* wh, const float* w4, unsigned* wpackT3, void* stream) {
    if (!w0 || !wh || !w4 || !wpackT3) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack_t3_kernel<3>, dim3(d3h_cdiv(X3_WPACKT_DWORDS, 256)), dim3(256), 0, (hipStream_t)stream, w0, wh, w4, wpackT3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// dwords of a wpackTh2 buffer (d3h_sdf_mlp_pack_t_h2)
extern "C" int64_t d3h_sdf_mlp_wpackth2_dwords(void) { return XPT<2>::WPACKT_DWORDS; }

// wpackTh2 [d3h_sdf_mlp_wpackth2_dwords()] (overwritten) = the transposed weights of d3h_sdf_mlp_pack_t, each as TWO fp16 planes (sdf_mlp_x3.h
// "h2") in the fragment order of the data-backward sweeps; pass it as `wpackT3` with t_planes = 2
extern "C" int d3h_sdf_mlp_pack_t_h2(const float* w0, const float* wh, const float* w4, unsigned* wpackTh2, void* stream) {
    if (!w0 || !wh || !w4 || !wpackTh2) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack_t3_kernel<2>, dim3(d3h_cdiv(XPT<2>::WPACKT_DWORDS, 256)), dim3(256), 0, (hipStream_t)stream, w0, wh, w4, wpackTh2);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
#endif

#if D3H_MLP_NOUT == 1
// First half of the COMPACT backward (see d3h_sdf_mlp_bwd), callable as soon as the set of points that CAN receive a gradient is known --
// before the gradient itself exists: marks[n], non-zero = the point may receive one (for a training sweep: the grid vertices on sign-changing
// edges, which the SDF regulariser's forward visits anyway -- every vertex marching tets interpolates between and every edge the regulariser
// penalises is among them).  Builds the point list, gathers the points and recomputes their activations into `act`; d3h_sdf_mlp_bwd called
// with prepared = 1, the SAME tile_list scratch and the SAME `act` then skips those steps.  A marked point whose gradient turns out zero
// contributes zeros.  The three launches (~100 us at 9 k points) leave the serial tail of a training step for a stream of the caller's choice.
extern "C" int d3h_sdf_mlp_bwd_prepare(const float* x, const float* deform, float disp, const float* marks, int64_t n, int* tile_list,
                                       const unsigned* wpack3_recompute, int recompute_planes, float* act, void* stream) {
    if (n < 0 || n >= (int64_t)1 << 31 || !x || !marks || !tile_list || !wpack3_recompute || !act || (recompute_planes != 2 && recompute_planes != 3)) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    hipStream_t s = (hipStream_t)stream;
    int* counts = tile_list;
    int* pl = counts + 8;
    int* ident = pl + bwd_r4(n);
    float* xg = (float*)(ident + bwd_r4(n / 16 + 2));
    float* gg = xg + 3 * bwd_p16(n);
    (void)hipMemsetAsync(counts, 0, 8 * sizeof(int), s);
    hipLaunchKernelGGL(sdf_mlp_active_points_kernel, dim3((unsigned)d3h_cdiv(n, 256)), dim3(256), 0, s, marks, n, pl, counts);
    hipLaunchKernelGGL(sdf_mlp_gather_points_kernel, dim3(256), dim3(256), 0, s, x, deform, disp, marks, (const int*)pl, counts, ident, xg, gg);
    int e = d3h_sdf_mlp_fwd_x3_list_launch(xg, nullptr, 0.f, wpack3_recompute, recompute_planes, act, n, ident, counts + 1, s);
    if (e != 0) return e;
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
#endif

// gout: [n][NOUT] (NOUT = 1 for the SDF network).  Gradients are ACCUMULATED into dw0[256][EMB], db0[256], dwh[5][256][256], dbh[5][256],
// dw4[256][256 + EMB], db4[256], dw7[NOUT][256], db7[NOUT]  (EMB = 39 for the SDF network, 51 for the offset network)
// (caller zero-fills or passes .grad buffers); dx[n][3] is overwritten (may be NULL).  dz: scratch, d3h_sdf_mlp_act_floats(n).
// tile_list: int scratch of (n + 15) / 16 + 1 entries, or NULL.  When given, the backward runs only over the 16-point tiles that
// contain a non-zero gout (exact: the others contribute zero to every output) -- the normal case of a training sweep, where the loss
// reads the sdf only next to the extracted surface.
// wpack3_recompute: NULL = `act` holds the activations of the forward (d3h_sdf_mlp_fwd* with the save).  Otherwise (d3h_sdf_mlp_pack3 of the same
// weights) the forward ran WITHOUT the save and `act` is scratch of d3h_sdf_mlp_act_floats(n) floats: the activations the backward needs are
// recomputed here first.  With tile_list (then an int scratch of d3h_sdf_mlp_bwd_scratch_ints(n) entries) the backward runs in the COMPACT form:
// the points with a non-zero gout are gathered into dense 16-point tiles (see sdf_mlp_active_points_kernel) and only those are recomputed and
// back-propagated: ~3 % of a grid sweep instead of a 1.88 GB store in the forward of which ~15 % was read back.
// recompute_planes: which forward the sweep ran, i.e. what wpack3_recompute is: 3 = d3h_sdf_mlp_pack3 (d3h_sdf_mlp_fwd_x3), 2 = d3h_sdf_mlp_pack_h2
// (d3h_sdf_mlp_fwd_h2) -- the recompute repeats that sweep's arithmetic bit for bit; ignored when wpack3_recompute is NULL.
extern "C" int d3h_sdf_mlp_bwd(const float* x, const float* deform, float disp, const float* gout, const float* w7,
                               const float* wpackT, const unsigned* wpackT3, const float* act, float* dz, int64_t n, float* dx, float* dw0, float* db0,
                               float* dwh, float* dbh, float* dw4, float* db4, float* dw7, float* db7, int* tile_list,
                               const unsigned* wpack3_recompute, int recompute_planes, int t_planes, int prepared, void* stream) {
    if (n < 0) return D3H_ERR_ARG;
    if (prepared && !(tile_list && wpack3_recompute)) return D3H_ERR_ARG;
    if (wpack3_recompute && recompute_planes != 2 && recompute_planes != 3) return D3H_ERR_ARG;
    if (wpackT3 && t_planes != 2 && t_planes != 3) return D3H_ERR_ARG;
    const float* sc_dev = nullptr;          // {s, 1 / s} of the h2 sweeps (t_planes == 2), on the device
    float* sc_tmp = nullptr;
    if (n == 0) return D3H_OK;
    if (!x || !gout || !w7 || (!wpackT && !wpackT3) || !act || !dz || !dw0 || !db0 || !dwh || !dbh || !dw4 || !db4 || !dw7 || !db7) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int nt32 = ntiles * 4;
    int nt16 = ntiles * 8;
    int grid = sdf_chain_grid(ntiles, 0);
    const int* list = nullptr;
    const int* cnt = nullptr;
    // the arrays the sweeps below run on: the caller's, or the gathered ones of the compact form
    const float *xs = x, *dfs = deform, *gs = gout;
    float disps = disp, *dxs = dx;
    const int* plist = nullptr;
    const int* pcounts = nullptr;
#if D3H_MLP_NOUT == 1
    if (wpack3_recompute && !wpackT3) return D3H_ERR_ARG;          // the recompute pass is a bf16-pipe kernel: both packs or neither
    if (tile_list && wpack3_recompute) {
        if (n >= (int64_t)1 << 31) return D3H_ERR_ARG;
        int* counts = tile_list;
        int* pl = counts + 8;
        int* ident = pl + bwd_r4(n);
        float* xg = (float*)(ident + bwd_r4(n / 16 + 2));
        float* gg = xg + 3 * bwd_p16(n);
        float* dxg = gg + bwd_p16(n);
        if (dx) (void)hipMemsetAsync(dx, 0, (size_t)n * 3 * sizeof(float), s);
        if (prepared) {
            // d3h_sdf_mlp_bwd_prepare ran on this scratch and this `act` (list, gathered points, recomputed activations): what is left to
            // gather is the upstream gradient of the listed points (zero where a marked point received none) and its magnitude
            hipLaunchKernelGGL(sdf_mlp_gather_gout_kernel, dim3(64), dim3(256), 0, s, gout, (const int*)pl, counts, gg);
            if (t_planes == 2) {
                hipLaunchKernelGGL(h2_scale_from_absmax_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, (int64_t)0, (const unsigned*)(counts + 6), (float*)(counts + 4));
                sc_dev = (const float*)(counts + 4);
            }
        } else {
            (void)hipMemsetAsync(counts, 0, 8 * sizeof(int), s);
            hipLaunchKernelGGL(sdf_mlp_active_points_kernel, dim3((unsigned)d3h_cdiv(n, 256)), dim3(256), 0, s, gout, n, pl, counts);
            if (t_planes == 2) {
                hipLaunchKernelGGL(h2_scale_from_absmax_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, (int64_t)0, (const unsigned*)(counts + 2), (float*)(counts + 4));
                sc_dev = (const float*)(counts + 4);
            }
            hipLaunchKernelGGL(sdf_mlp_gather_points_kernel, dim3(256), dim3(256), 0, s, x, deform, disp, gout, (const int*)pl, counts, ident, xg, gg);
        }
        list = ident;
        cnt = counts + 1;
        xs = xg; dfs = nullptr; disps = 0.f; gs = gg; dxs = dx ? dxg : nullptr;
        plist = pl; pcounts = counts;
        if (!prepared) {
            int e = d3h_sdf_mlp_fwd_x3_list_launch(xs, nullptr, 0.f, wpack3_recompute, recompute_planes, (float*)act, n, list, cnt, s);
            if (e != 0) return e;
        }
    } else
#endif
    if (tile_list) {
        int nt16r = (int)((n + 15) / 16);
        int* count = tile_list + nt16r;
        (void)hipMemsetAsync(count, 0, sizeof(int), s);
        if (dx) (void)hipMemsetAsync(dx, 0, (size_t)n * 3 * sizeof(float), s);
        hipLaunchKernelGGL(sdf_mlp_active_tiles_kernel, dim3(d3h_cdiv(nt16r, 256)), dim3(256), 0, s, gout, n, nt16r, tile_list, count);
        list = tile_list;
        cnt = count;
    }
#if D3H_MLP_NOUT == 1
    if (wpack3_recompute && !tile_list) {          // dense backward without saved activations: every tile is visited, recompute them all
        int e = (recompute_planes == 2 ? d3h_sdf_mlp_fwd_h2 : d3h_sdf_mlp_fwd_x3)(x, deform, disp, wpack3_recompute, dz /* n floats of the dz scratch, overwritten below */, nullptr, (float*)act, n, 0, s);
        if (e != 0) return e;
    }
#else
    if (wpack3_recompute) return D3H_ERR_ARG;
#endif
#if D3H_MLP_NOUT == 1
    if (wpackT3 && t_planes == 2 && !sc_dev) {          // the paths without the active-point pre-pass: the scale from a reduction over gout
        if (hipMallocAsync((void**)&sc_tmp, 2 * sizeof(float), s) != hipSuccess) return D3H_ERR_ARG;
        hipLaunchKernelGGL(h2_scale_from_absmax_kernel, dim3(1), dim3(256), 0, s, gs, n * NOUT, (const unsigned*)nullptr, sc_tmp);
        sc_dev = sc_tmp;
    }
#endif
    const int ktb = d3h_ktime_begin(tile_list ? D3H_KT_SDF_BWD_DATA_SPARSE : D3H_KT_SDF_BWD_DATA, n, s);
#if D3H_MLP_NOUT == 1
    if (wpackT3)
        bwd_data_x_launch<false>(t_planes, grid, s, xs, dfs, disps, gs, w7, wpackT3, act, dz, dxs, n, ntiles, list, cnt, sc_dev, 1.0f);
    else
#endif
        hipLaunchKernelGGL((sdf_mlp_bwd_data_kernel<false>), dim3(grid), dim3(NTHREADS), 0, s, xs, dfs, disps, gs, w7, wpackT, act, dz, dxs, n,
                           ntiles, list, cnt);
    d3h_ktime_end(ktb, s);
#if D3H_MLP_NOUT == 1
    if (plist && dx) hipLaunchKernelGGL(sdf_mlp_scatter_dx_kernel, dim3(64), dim3(256), 0, s, (const float*)dxs, plist, pcounts, dx);
#endif
    // weight gradients: split the points over S workgroups per column chunk
    // split-K width of the weight-gradient GEMMs: every workgroup ends with a 256 x 128 atomic flush, so S x 2 x 32768 atomics per
    // launch.  Measured in the training step (tools/gpu_probe_dw.py, bench.py): S = 128 (one workgroup per CU) 10.7 ms/step, 256 (two
    // per CU, load/MFMA phases overlapped) 11.0, 64: 11.4 -- at 5 10^4..10^5 points the flush outweighs the overlap
    int S = dw_split(nt32, DW_SPLIT);
    const int SE = dw_split(nt32, D3H_DW_SPLIT_EMB);
    // sparse sweep (~2000 active tiles = ~8 groups per workgroup at S = 128): the flush IS the kernel; S = 32: 270 -> 180 us (21: same, 64: 212)
    const int SL = tile_list ? (nt32 < D3H_DW_SPLIT_SPARSE ? nt32 : D3H_DW_SPLIT_SPARSE) : S;
    const int SEL = (plist && SE > D3H_DW_SPLIT_SPARSE) ? D3H_DW_SPLIT_SPARSE : SE;          // compact form: a few hundred tiles, the flush is the kernel here too
    const float* nof = nullptr;
    const int ktw = d3h_ktime_begin(D3H_KT_SDF_DW_LAYERS_SPARSE, n, s);
#if D3H_MLP_NOUT == 1
    if (wpackT3 && dw_x3_enabled(2))      // the bf16-pipe arithmetic was asked for: the weight-gradient GEMMs follow (sdf_mlp_bwd_dw_layers_x3_kernel)
    {
        if (t_planes == 2 && dw_h2_enabled())
            hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_h2_kernel, dim3(SL, 2, 6), dim3(512), 0, s, dz, act, nt16, dwh, dbh, dw4, db4, list, cnt, nof, nof, sc_dev, 1.0f, 1);
        else
            hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_x3_kernel, dim3(SL, 2, 6), dim3(512), 0, s, dz, act, nt16, dwh, dbh, dw4, db4, list, cnt, nof, nof);
    }
    else
#endif
        hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_kernel, dim3(SL, 2, 6), dim3(512), 0, s, dz, act, xs, n, nt32, dwh, dbh, dw4, db4, list, cnt, nof, nof);
    d3h_ktime_end(ktw, s);
    {
        const DwEmbArgs e4 = {dz + (size_t)4 * ACT_LAYER_FLOATS, act + (size_t)3 * ACT_LAYER_FLOATS, dw4, 256 + EMB_DIM, 256, EMB_DIM, (float*)nullptr, nof, nof, nof};
        const DwEmbArgs e0 = {dz, act, dw0, EMB_DIM, 0, EMB_DIM, db0, nof, nof, nof};
        hipLaunchKernelGGL(sdf_mlp_bwd_dw_emb_pair_kernel, dim3(SEL, 1, 2), dim3(512), 0, s, e4, e0, xs, dfs, disps, n, nt32, list, cnt);
    }
    int g7 = nt16 < LAST_GRID ? nt16 : LAST_GRID;
    hipLaunchKernelGGL(sdf_mlp_bwd_last_kernel, dim3(g7, NOUT), dim3(256), 0, s, gs, act + (size_t)6 * ACT_LAYER_FLOATS, n, nt16, dw7, db7, list, cnt);
    if (sc_tmp) (void)hipFreeAsync(sc_tmp, s);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

#if D3H_MLP_NOUT == 1
// ------------------------------------------------------------------------------------------------
// eikonal term (geometry/hmsdf.py:856-876 of the reference: autograd.grad(sdf.sum(), x, create_graph=True) -> ((|g|-1)^2).mean() ->
// backward through the gradient graph).  Hand-derived second-order pass on the same kernels:
//   A. d3h_sdf_mlp_fwd with the activation save                                   h_l
//   B. d3h_sdf_mlp_grad_x: first-order backward with d(sdf) = 1                  g = grad_x f,  dZ_l = s_l * dH_l
//      (host: L = c * mean((|g|-1)^2), u = dL/dg)
//   C. tangent pass (sdf_mlp_fwd_kernel<true>): t_e = J_emb(x) u, q_l = W_l t_{l-1}, t_l = s_l q_l;  e_l = 100 (1 - s_l) dZ_l q_l
//   D. reverse sweep with injection (sdf_mlp_bwd_data_kernel<true>): dZ^_l = s_l (W_{l+1}^T dZ^_{l+1}) + e_l
//   E. dW_l += dZ_l (x) t_{l-1} + dZ^_l (x) h_{l-1},  db_l += sum dZ^_l,  dW_7 += sum t_6     (<u, g> is linear in every W_l given s)
// ------------------------------------------------------------------------------------------------
int d3h_sdf_mlp_jvp_launch(const float* x, const float* udir, const float* wpack, const float* act, const float* dz, float* tb, float* eb,
                           int64_t n, int max_cus, hipStream_t s);
int d3h_sdf_mlp_jvp_x3_launch(const float* x, const float* udir, const unsigned* wpack3, int planes, float uscale, const float* act, const float* dz, float* tb,
                              float* eb, int64_t n, int max_cus, hipStream_t s);

// g[n][3] = d(sdf)/d(x) from the saved activations of a forward with save; fills dz (tile-packed dZ_l, kept for d3h_sdf_mlp_eik_bwd)
// (max_cus: as d3h_sdf_mlp_fwd)
// wpackT3: optional (d3h_sdf_mlp_pack_t3 of the same weights): the sweep then runs on the bf16 matrix pipe (sdf_mlp_x3.h) and wpackT may be NULL
// t_planes: what wpackT3 is -- 3 = d3h_sdf_mlp_pack_t3 (bf16 x 3), 2 = d3h_sdf_mlp_pack_t_h2 (fp16 x 2; d(sdf) = 1 here, so the operand scale is a constant)
extern "C" int d3h_sdf_mlp_grad_x(const float* x, const float* w7, const float* wpackT, const unsigned* wpackT3, int t_planes, const float* act, float* dz,
                                  int64_t n, float* g, int max_cus, void* stream) {
    if (n < 0) return D3H_ERR_ARG;
    if (wpackT3 && t_planes != 2 && t_planes != 3) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    if (!x || !w7 || (!wpackT && !wpackT3) || !act || !dz || !g) return D3H_ERR_ARG;
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, max_cus);
    const int kt = d3h_ktime_begin(D3H_KT_SDF_BWD_DATA, n, (hipStream_t)stream);
    if (wpackT3)
        bwd_data_x_launch<false>(t_planes, grid, (hipStream_t)stream, x, nullptr, 0.f, nullptr, w7, wpackT3, act, dz, g, n, ntiles, nullptr, nullptr, nullptr,
                                 h2_grad_scale(1.0f));
    else
        hipLaunchKernelGGL((sdf_mlp_bwd_data_kernel<false>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, x, (const float*)nullptr, 0.f,
                           (const float*)nullptr, w7, wpackT, act, dz, g, n, ntiles, (const int*)nullptr, (const int*)nullptr);
    d3h_ktime_end(kt, (hipStream_t)stream);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// loss_sum[0] (zeroed here) = sum_p (|g_p| - 1)^2;  u[n][3] (optional) = scale * d(sum)/d(g): the seed of d3h_sdf_mlp_eik_bwd
extern "C" int d3h_eikonal_loss(const float* g, int64_t n, float scale, float* loss_sum, float* u, void* stream) {
    if (n < 0 || !loss_sum || (n > 0 && !g)) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(loss_sum, 0, sizeof(float), s);
    if (n > 0) hipLaunchKernelGGL(eikonal_loss_kernel, dim3(d3h_grid(n, 256)), dim3(256), 0, s, g, n, scale, loss_sum, u);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// Weight gradients of sum_p <u_p, grad_x f(x_p)> ACCUMULATED into dw0 .. dw7 (layouts as d3h_sdf_mlp_bwd; there is no db7 term).
// act / dz: from d3h_sdf_mlp_fwd(save) / d3h_sdf_mlp_grad_x on the same x; tb, eb: scratch of d3h_sdf_mlp_act_floats(n) floats each.
// max_cus: as d3h_sdf_mlp_fwd (the two sweeps; the weight-gradient GEMMs keep their split-K grids).
// wpack3 / wpackT3: optional (d3h_sdf_mlp_pack3 / d3h_sdf_mlp_pack_t3 of the same weights): the tangent / reverse sweep then runs on the
// bf16 matrix pipe (sdf_mlp_x3.h) and the f32 pack it replaces may be NULL.
// f_planes / t_planes: what wpack3 / wpackT3 are (3: d3h_sdf_mlp_pack3 / d3h_sdf_mlp_pack_t3, 2: d3h_sdf_mlp_pack_h2 / d3h_sdf_mlp_pack_t_h2).
// u_hint (needed when either is 2): the magnitude of the entries of udir, to within a factor of ~30 either way -- the h2 sweeps scale their
// operands by a power of two derived from it (tangents ~10 |u|, injected curvature terms ~10^2 |u|); for the eikonal loss: 2 * coeff / n.
extern "C" int d3h_sdf_mlp_eik_bwd(const float* x, const float* udir, const float* wpack, const float* wpackT, const unsigned* wpack3, int f_planes,
                                   const unsigned* wpackT3, int t_planes, float u_hint, const float* act,
                                   const float* dz, float* tb, float* eb, int64_t n, float* dw0, float* db0, float* dwh, float* dbh,
                                   float* dw4, float* db4, float* dw7, int max_cus, void* stream) {
    if (n < 0) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    if (!x || !udir || (!wpack && !wpack3) || (!wpackT && !wpackT3) || !act || !dz || !tb || !eb || !dw0 || !db0 || !dwh || !dbh || !dw4 || !db4 || !dw7)
        return D3H_ERR_ARG;
    if (wpackT3 && (t_planes != 2 && t_planes != 3)) return D3H_ERR_ARG;
    if (wpackT3 && t_planes == 2 && !(u_hint > 0.f)) return D3H_ERR_ARG;
    if (wpack3 && (f_planes != 2 && f_planes != 3)) return D3H_ERR_ARG;
    if (wpack3 && f_planes == 2 && !(u_hint > 0.f)) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int nt32 = ntiles * 4;
    int grid = sdf_chain_grid(ntiles, max_cus);
    {
        int e = wpack3 ? d3h_sdf_mlp_jvp_x3_launch(x, udir, wpack3, f_planes, f_planes == 2 ? h2_grad_scale(u_hint * 256.0f) : 1.0f, act, dz, tb, eb, n, max_cus, s)
                       : d3h_sdf_mlp_jvp_launch(x, udir, wpack, act, dz, tb, eb, n, max_cus, s);
        if (e != 0) return e;
    }
    // w7 is unused when INJECT (dH^_6 = 0): pass wpackT as a valid 256-float placeholder
    const int kti = d3h_ktime_begin(D3H_KT_SDF_BWD_INJECT, n, s);
    if (wpackT3)
        bwd_data_x_launch<true>(t_planes, grid, s, x, nullptr, 0.f, nullptr, (const float*)wpackT3, wpackT3, act, eb, nullptr, n, ntiles, nullptr, nullptr, nullptr,
                                h2_grad_scale(u_hint * 256.0f));         // (|u| s = 1/4: the injected terms, ~10^2 .. 3 10^3 |u|, land at 25 .. 750; their growth through W^T stays far below 65 504)
    else
        hipLaunchKernelGGL((sdf_mlp_bwd_data_kernel<true>), dim3(grid), dim3(NTHREADS), 0, s, x, (const float*)nullptr, 0.f, (const float*)nullptr,
                           wpackT, wpackT, act, eb, (float*)nullptr, n, ntiles, (const int*)nullptr, (const int*)nullptr);
    d3h_ktime_end(kti, s);
    int S = dw_split(nt32, DW_SPLIT);
    const int SE = dw_split(nt32, D3H_DW_SPLIT_EMB);
    const float* nof = nullptr;
    float* nob = nullptr;
    const int* noi = nullptr;
    // one dual launch for the six hidden layers: dz_l (x) t_{l-1} and dZ^_l (x) h_{l-1} share the accumulators and the atomic flush
    const int ktd = d3h_ktime_begin(D3H_KT_SDF_DW_LAYERS, n, s);
    if (wpackT3 && dw_x3_enabled(3)) {
        // the x3 kernel shares its SIMDs with nobody (D3H_X3_CLAIM_SIMD): next to the render it gets the chain's CU budget as a workgroup
        // count -- 12 workgroups per split step (2 column chunks x 6 layers), each on a CU of its own
        int Sx = S;
        static int env_s = -1;
        if (env_s < 0) { const char* e = getenv("D3H_DWX_DUAL_SPLIT"); env_s = e ? atoi(e) : 0; }
        if (env_s > 0) Sx = env_s;
        // (only when the chain was given the SMALL budget -- the step renders >= 2 Mpixel beside it, geometry/hmsdf.py:_eikonal_async; with the
        // 196-CU budget of a light render the launch is on the critical path and keeps the full split: config 2 3.2 vs 3.7 ms)
        else if (max_cus > 0 && max_cus <= 160 && max_cus / 12 >= 1 && max_cus / 12 < Sx) Sx = max_cus / 12;
        if (t_planes == 2 && dw_h2_enabled())          // first pair: dz (O(1)) x t (gradient-valued: scaled); second pair: dZ^ (scaled) x h
            hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_h2_kernel, dim3(Sx, 2, 6), dim3(512), 0, s, dz, tb, ntiles * 8, dwh, dbh, dw4, db4, noi, noi,
                               (const float*)eb, act, nof, h2_grad_scale(u_hint * 256.0f), 2 | 4);
        else
            hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_x3_kernel, dim3(Sx, 2, 6), dim3(512), 0, s, dz, tb, ntiles * 8, dwh, dbh, dw4, db4, noi, noi,
                               (const float*)eb, act);
    }
    else
        hipLaunchKernelGGL(sdf_mlp_bwd_dw_layers_kernel, dim3(S, 2, 6), dim3(512), 0, s, dz, tb, x, n, nt32, dwh, dbh, dw4, db4, noi, noi, (const float*)eb,
                           act);
    d3h_ktime_end(ktd, s);
    {
        const DwEmbArgs e4 = {dz + (size_t)4 * ACT_LAYER_FLOATS, tb + (size_t)3 * ACT_LAYER_FLOATS, dw4, 256 + EMB_DIM, 256, EMB_DIM, nob, udir,
                              (const float*)(eb + (size_t)4 * ACT_LAYER_FLOATS), act + (size_t)3 * ACT_LAYER_FLOATS};
        const DwEmbArgs e0 = {dz, act, dw0, EMB_DIM, 0, EMB_DIM, db0, udir, (const float*)eb, act};
        hipLaunchKernelGGL(sdf_mlp_bwd_dw_emb_pair_kernel, dim3(SE, 1, 2), dim3(512), 0, s, e4, e0, x, nof, 0.f, n, nt32, noi, noi);
    }
    // dW_7 += sum_p t_6[p]: the head kernel with g = 1 and no db7 output
    int nt16 = ntiles * 8;
    int g7 = nt16 < LAST_GRID ? nt16 : LAST_GRID;
    hipLaunchKernelGGL(sdf_mlp_bwd_last_kernel, dim3(g7), dim3(256), 0, s, nof, tb + (size_t)6 * ACT_LAYER_FLOATS, n, nt16, dw7, nob, noi, noi);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
#endif
