// sdf_mlp_x3.hip -- the SDF query (and the tangent sweep of the eikonal term) with the layer GEMMs on the bf16 matrix pipe at fp32 accuracy.
//
// Same network, same register-resident chain, same saved-activation layout and the same results to fp32 rounding as sdf_mlp.hip
// (geometry/embedding.py:21-38 + geometry/mlp.py:34-45 + geometry/hmsdf.py:433-444); what differs is the arithmetic of the seven
// 256-wide GEMMs: every fp32 operand travels as three bf16 numbers and every 16x16x32 block of a product is six
// v_mfma_f32_16x16x32_bf16 (sdf_mlp_x3.h has the error argument and the operand layout).  Per 16-point wave tile and hidden layer that
// is 768 MFMAs of 16 cycles against 1024 of 32: the matrix-pipe time of a sweep drops to 3/8, and the softplus epilogue of a SIMD's other
// wave now issues UNDER the MFMAs (the f32 MFMA shares the VALU's datapath, the bf16 MFMA does not).
//  * B operands: the epilogue leaves a layer's output in the D registers (fp32, 64 per lane); at the layer boundary they are split into the
//    three bf16 planes of the next layer's B operand (96 registers), in place -- no cross-lane traffic, see x3_feature().
//  * A operands: pre-split at pack time (d3h_sdf_mlp_pack3: 2.56 MB instead of 1.65 MB), streamed L2 -> LDS in 48 / 60 KiB chunks by
//    global_load_lds_dwordx4 (double buffered, one barrier per chunk), read conflict-free by ds_read_b128 (1 KiB per wave-instruction).
//  * The exact-f32 kernels stay in the library (D3H_SDF_X3=0 selects them in the Python layer; the parity tests run both).
#include "sdf_mlp_dev.h"
#include "sdf_mlp_x3.h"

#if D3H_MLP_NOUT == 1

using namespace D3H_MLP_NS;

// ------------------------------------------------------------------------------------------------
// pack: nn.Linear weights -> bf16 x 3 fragment order + the fp32 tail (biases, head)
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ void sdf_mlp_pack3_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ wh,
                                     const float* __restrict__ bh, const float* __restrict__ w4, const float* __restrict__ b4,
                                     const float* __restrict__ w7, const float* __restrict__ b7, unsigned* __restrict__ wpack3) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= XP<NP>::WPACK_DWORDS) return;
    if (idx >= XP<NP>::OFF_TAIL) {
        int j = idx - XP<NP>::OFF_TAIL;
        float v = 0.f;
        if (j < 256) v = b0[j];
        else if (j < 256 * 7) {
            int l = j >> 8, f = j & 255;
            if (l == 4) v = b4[f];
            else v = bh[((l < 4) ? (l - 1) : (l - 2)) * 256 + f];
        } else if (j < HEAD_B) v = w7[j - HEAD_W];
        else if (j < HEAD_B + NOUT) v = b7[j - HEAD_B];
        wpack3[idx] = __float_as_uint(v);
        return;
    }
    const int l = XP<NP>::layer_of_offset(idx);
    const int local = idx - XP<NP>::layer_offset(l);
    const int d = local & 3, lane = (local >> 2) & 63;
    int rest = local >> 8;                                       // flat (rbg, kb, part): chunks are contiguous in rbg
    const int part = rest % NP;
    rest /= NP;
    const int nkb = (l == 0) ? X3_EMB_KB : ((l == 4) ? X3_SKIP_KB : 8);
    const int kb = rest % nkb, rbg = rest / nkb;
    const int i = lane & 15, q = lane >> 4;
    const int out = 16 * rbg + i;
    unsigned bits[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int s = 2 * d + e;
        float v = 0.f;
        if (l == 0) {
            const int in = x3_feature(kb, q, s);
            if (in < EMB_DIM) v = w0[out * EMB_DIM + in];
        } else if (l == 4) {
            if (kb < 8) v = w4[out * (256 + EMB_DIM) + x3_feature(kb, q, s)];
            else {
                const int in = x3_feature(kb - 8, q, s);
                if (in < EMB_DIM) v = w4[out * (256 + EMB_DIM) + 256 + in];
            }
        } else {
            const int hi = (l < 4) ? (l - 1) : (l - 2);
            v = wh[(size_t)hi * 65536 + out * 256 + x3_feature(kb, q, s)];
        }
        unsigned h, m, lo = 0;
        if constexpr (NP == 3) x3_split_pair(v, 0.f, h, m, lo);
        else {
            h2_split_pair(v, 0.f, h, m);
            if (!(fabsf(v) <= 16384.0f)) h = m = 0x7e00u;       // outside the fp16 working range (or NaN): poison, never a silently wrong sweep
        }
        bits[e] = (part == 0 ? h : (part == 1 ? m : lo)) & 0xffffu;
    }
    wpack3[idx] = bits[0] | (bits[1] << 16);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
namespace {

// bias + softplus in place on one 16-feature block; optional tile-packed save for the backward pass (sdf_mlp.hip: epilogue)
__device__ __forceinline__ void x3_epilogue(f32x4& v, const float* bias_l, int rb, int lane, float* act_tile_layer) {
#ifdef D3H_X3_PROBE_NOEPI
    return;
#endif
    f32x4 b = *(const f32x4*)(bias_l + 16 * rb + 4 * (lane >> 4));
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = softplus100(v[r] + b[r]);
        v[r] = o[r];
    }
    if (act_tile_layer) __builtin_nontemporal_store(o, (f32x4*)(act_tile_layer + (rb * 64 + lane) * 4));
}

// JVP epilogue (sdf_mlp.hip: epilogue_jvp_pre): t = s q, e = 100 (1 - s) dz q from the saved h and the gradient pass's dz.  `usi`: the h2 tangent
// sweep carries its tangents multiplied by a power of two (the direction u is a loss gradient of magnitude ~1e-6); what it stores is un-scaled
__device__ __forceinline__ void x3_epilogue_jvp(f32x4& v, const f32x4 hh, const f32x4 dd, float* t_l, float* e_l, int rb, int lane, float usi) {
    const size_t off = (size_t)(rb * 64 + lane) * 4;
    f32x4 to, eo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float sg = dsoftplus_from_h(hh[r]);
        const float qv = v[r];
        v[r] = sg * qv;
        to[r] = v[r] * usi;
        eo[r] = 100.0f * (1.0f - sg) * dd[r] * (qv * usi);
    }
    *(f32x4*)(t_l + off) = to;
    *(f32x4*)(e_l + off) = eo;
}

}  // namespace

// (Tried and dropped: EVERY wave finishing a chunk's two blocks inside the next chunk's first MFMA chain, one value per k-block, interleaved
// with the MFMAs by sched_group_barrier {1 MFMA, 2 VALU}: the ISA interleaves as asked, the sweep gets 2-12 % SLOWER than the early / late
// bursts below -- 1.22 vs 1.20 ms with the save, 1.13 vs 1.01 without; 61-78 spilled registers instead of 25.)
// JVP / SMALL / the balanced tile assignment: as sdf_mlp_fwd_kernel (sdf_mlp.hip).  SAVE: `act` is written (compile-time).
// NP: operand planes of the GEMMs (sdf_mlp_x3.h): 3 = bf16 x 3 / six products, 2 = fp16 x 2 / three products ("h2"; `wpack3` is then a pack of
// d3h_sdf_mlp_pack_h2).  The h2 tangent sweep (JVP) multiplies its direction by `uscale` (a power of two, h2_grad_scale) as it loads it and
// what it stores by 1 / uscale: the direction is a loss gradient of magnitude ~1e-6, far below the fp16 normal range.
template <bool JVP, int SMALL, bool SAVE, int NP>
__global__ __launch_bounds__(NTHREADS, 2) void sdf_mlp_fwd_x3_kernel(const float* __restrict__ x, const float* __restrict__ deform, float disp,
                                                                    const unsigned* __restrict__ wpack3, float* __restrict__ sdf,
                                                                    float* __restrict__ xdef, float* __restrict__ act, int64_t n, int ntiles,
                                                                    const float* __restrict__ udir, const float* __restrict__ dzb,
                                                                    float* __restrict__ tb, float* __restrict__ eb,
                                                                    const int* __restrict__ tile_list, const int* __restrict__ tile_count, float uscale) {
    using P = XP<NP>;
    const float us = (JVP && NP == 2) ? uscale : 1.0f, usi = (JVP && NP == 2) ? 1.0f / uscale : 1.0f;
    __shared__ __attribute__((aligned(16))) unsigned wbuf[2][P::CHUNK_MAX];
    __shared__ __attribute__((aligned(16))) float bias[JVP ? 4 : BIAS_FLOATS];
    __shared__ __attribute__((aligned(16))) float jpf[JVP ? NWAVES * 4 * 256 : 4];

    D3H_X3_CLAIM_SIMD();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile pointers derived from it live in SGPRs
    const int q = lane >> 4;

    if (!JVP)
        for (int i = tid; i < BIAS_FLOATS; i += NTHREADS) bias[i] = __uint_as_float(wpack3[P::OFF_TAIL + i]);

    int pb = 0;
    x3_issue(wpack3, wbuf[0], P::L0_CHUNK / 4, tid);
    glds_commit();                                               // also publishes bias[]

    u32x4 Xs[8][NP];
    f32x4 Y[16];

    // tile_list (SMALL != 0 launches only): the sweep visits the 16-point tiles list[0 .. *tile_count) instead of all of them -- the recompute
    // pass of the sparse backward (d3h_sdf_mlp_bwd): activations are written at the tiles' own positions in `act`, `sdf` may be NULL
    constexpr bool BAL = JVP || SMALL != 0 || NWAVES != 8;
    const int64_t n16 = (BAL && tile_list) ? (int64_t)*tile_count : (int64_t)ntiles * 8;
    const int G = (int)gridDim.x;
    const int nrounds = BAL ? (int)((n16 + NWAVES * (int64_t)G - 1) / (NWAVES * (int64_t)G)) : (ntiles - (int)blockIdx.x + G - 1) / G;
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int tile = (int)blockIdx.x + rnd * G;
        int64_t t16 = BAL ? ((int64_t)rnd * NWAVES * G + (int64_t)wave * G + blockIdx.x) : ((int64_t)tile * 8 + wave);
        const bool on = !BAL || t16 < n16;                       // wave-uniform
        if (BAL && tile_list) t16 = on ? (int64_t)tile_list[t16] : 0;
        const int64_t p = t16 * 16 + (lane & 15);
        const bool valid = p < n;
        float* act_tile = (SAVE || JVP) ? act + t16 * ACT_TILE_FLOATS : nullptr;
        const float* dz_tile = JVP ? dzb + t16 * ACT_TILE_FLOATS : nullptr;
        float* t_tile = JVP ? tb + t16 * ACT_TILE_FLOATS : nullptr;
        float* e_tile = JVP ? eb + t16 * ACT_TILE_FLOATS : nullptr;

        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (valid) {
            x0 = x[3 * p + 0]; x1 = x[3 * p + 1]; x2 = x[3 * p + 2];
            if (deform) {   // hmsdf.py:433  verts + max_displacement * deform  (two roundings, no fma)
                x0 = __fadd_rn(x0, __fmul_rn(disp, deform[3 * p + 0]));
                x1 = __fadd_rn(x1, __fmul_rn(disp, deform[3 * p + 1]));
                x2 = __fadd_rn(x2, __fmul_rn(disp, deform[3 * p + 2]));
            }
            if (xdef && q == 0) { xdef[3 * p + 0] = x0; xdef[3 * p + 1] = x1; xdef[3 * p + 2] = x2; }
        }
        // the positional encoding (or its tangent), split once: the B operand of layer 0 and of the skip k-blocks of layer 4
        u32x4 E3[X3_EMB_KB][NP];
        {
            float u0 = 0.f, u1 = 0.f, u2 = 0.f;
            if (JVP && valid) { u0 = udir[3 * p + 0] * us; u1 = udir[3 * p + 1] * us; u2 = udir[3 * p + 2] * us; }
#pragma unroll
            for (int kb = 0; kb < X3_EMB_KB; ++kb) {
                f32x4 v0, v1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e0 = 32 * kb + 4 * q + r, e1 = e0 + 16;
                    v0[r] = JVP ? emb_tangent(e0, x0, x1, x2, u0, u1, u2) : emb_feature(e0, x0, x1, x2);
                    v1[r] = JVP ? emb_tangent(e1, x0, x1, x2, u0, u1, u2) : emb_feature(e1, x0, x1, x2);
                }
                xp_split_blocks<NP>(v0, v1, E3[kb]);
            }
        }

        // ---- layer 0: emb -> Y, two chunks of 8 row blocks ------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned* nsrc = (c == 0) ? wpack3 + P::L0_CHUNK : wpack3 + P::OFF_L1;
            const int nn4 = ((c == 0) ? P::L0_CHUNK : P::HID_CHUNK) / 4;
            x3_issue(nsrc, wbuf[pb ^ 1], nn4, tid);
            const unsigned* wl = wbuf[pb];
            if (on) {
#pragma unroll
                for (int rbl = 0; rbl < 8; ++rbl) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
                    xp_mac_blocks<NP, X3_EMB_KB, -1>(acc, lo, E3, wl + rbl * (X3_EMB_KB * NP * X3_FRAG), lane, X3None());
                    Y[8 * c + rbl] = xp_fold<NP>(acc, lo);
                }
            }
            glds_commit();
            pb ^= 1;
            if (on) {
#pragma unroll
                for (int rbl = 0; rbl < 8; ++rbl) {
                    const int rb = 8 * c + rbl;
                    if (JVP) {
                        const size_t off = (size_t)(rb * 64 + lane) * 4;
                        x3_epilogue_jvp(Y[rb], *(const f32x4*)(act_tile + off), *(const f32x4*)(dz_tile + off), t_tile, e_tile, rb, lane, usi);
                    } else x3_epilogue(Y[rb], bias, rb, lane, act_tile);
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xp_split_blocks<NP>(Y[2 * kb], Y[2 * kb + 1], Xs[kb]);

        // ---- layers 1..6: Xs -> Y, then Xs = split(Y) ------------------------------------------------------------------------------
        // Stagger (sdf_mlp.hip): waves 4..7 ("late") run a chunk's epilogue a quarter of a chunk into the NEXT chunk's MFMAs, so that one of
        // the two waves of a SIMD is always issuing MFMAs.  Their last pair of a layer (blocks 14, 15) is finished -- and split into
        // k-block 7 of the next layer's input -- in the first chunk of that layer, before k-block 7 is consumed.
        const bool late = !JVP && NWAVES == 8 && wave >= 4;
        auto epi = [&](f32x4& v, int l, int rb) { x3_epilogue(v, bias + 256 * l, rb, lane, act_tile ? act_tile + l * ACT_LAYER_FLOATS : nullptr); };
#pragma unroll 1
        for (int l = 1; l <= 6; ++l) {
            const bool skip = (l == 4);
            const int this_chunk = skip ? P::SKIP_CHUNK : P::HID_CHUNK;
            const int rstride = (skip ? X3_SKIP_KB : 8) * NP * X3_FRAG;
            const unsigned* lbase = wpack3 + P::layer_offset(l);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned* nsrc = (c < 7) ? lbase + (c + 1) * this_chunk : ((l == 6) ? wpack3 : wpack3 + P::layer_offset(l + 1));
                const int nn4 = ((c < 7) ? this_chunk : ((l == 6) ? P::L0_CHUNK : ((l == 3) ? P::SKIP_CHUNK : P::HID_CHUNK))) / 4;
                x3_issue(nsrc, wbuf[pb ^ 1], nn4, tid);
                if (JVP && on) {
                    float* pf = jpf + wave * (4 * 256);
                    const size_t o0 = (size_t)l * ACT_LAYER_FLOATS + (size_t)((2 * c) * 64 + lane) * 4;
                    D3H_GLDS16(act_tile + o0, pf);
                    D3H_GLDS16(dz_tile + o0, pf + 256);
                    D3H_GLDS16(act_tile + o0 + 256, pf + 512);
                    D3H_GLDS16(dz_tile + o0 + 256, pf + 768);
                }
                const unsigned* wl = wbuf[pb];
                if (on) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, lo0 = {0.f, 0.f, 0.f, 0.f}, lo1 = {0.f, 0.f, 0.f, 0.f};
                    xp_mac_blocks<NP, 8, 4>(acc0, lo0, Xs, wl, lane, [&] {
                        if (late) {
                            if (c > 0) { epi(Y[2 * c - 2], l, 2 * c - 2); epi(Y[2 * c - 1], l, 2 * c - 1); }
                            else if (l > 1) {
                                epi(Y[14], l - 1, 14);
                                epi(Y[15], l - 1, 15);
                                xp_split_blocks<NP>(Y[14], Y[15], Xs[7]);
                            }
                        }
                    });
#ifndef D3H_X3_PROBE_NOSKIP      // (timing probe: what the 24 registers of E3 cost inside the layer loop; results are wrong)
                    if (skip) xp_mac_blocks<NP, X3_EMB_KB, -1>(acc0, lo0, E3, wl + 8 * NP * X3_FRAG, lane, X3None());      // mlp.py:40-41 cat([x, emb])
#endif
                    xp_mac_blocks<NP, 8, -1>(acc1, lo1, Xs, wl + rstride, lane, X3None());
#ifndef D3H_X3_PROBE_NOSKIP
                    if (skip) xp_mac_blocks<NP, X3_EMB_KB, -1>(acc1, lo1, E3, wl + rstride + 8 * NP * X3_FRAG, lane, X3None());
#endif
                    Y[2 * c] = xp_fold<NP>(acc0, lo0);
                    Y[2 * c + 1] = xp_fold<NP>(acc1, lo1);
                }
                glds_commit();
                pb ^= 1;
                if (JVP) {
                    if (on) {
                        const float* pf = jpf + wave * (4 * 256) + lane * 4;
                        float* tl = t_tile + l * ACT_LAYER_FLOATS;
                        float* el = e_tile + l * ACT_LAYER_FLOATS;
                        x3_epilogue_jvp(Y[2 * c], *(const f32x4*)pf, *(const f32x4*)(pf + 256), tl, el, 2 * c, lane, usi);
                        x3_epilogue_jvp(Y[2 * c + 1], *(const f32x4*)(pf + 512), *(const f32x4*)(pf + 768), tl, el, 2 * c + 1, lane, usi);
                    }
                } else if (on && !late) { epi(Y[2 * c], l, 2 * c); epi(Y[2 * c + 1], l, 2 * c + 1); }
            }
            if (l < 6) {
#pragma unroll
                for (int kb = 0; kb < 7; ++kb) xp_split_blocks<NP>(Y[2 * kb], Y[2 * kb + 1], Xs[kb]);
                if (!late) xp_split_blocks<NP>(Y[14], Y[15], Xs[7]);
            }
        }
        if (on && late) { epi(Y[14], 6, 14); epi(Y[15], 6, 15); }          // flush the deferred pair of layer 6

        if (JVP || !on) continue;     // the tangent of the head is not needed: the eikonal loss does not depend on f itself
        // ---- layer 7: 256 -> 1 (net.14), fp32 VALU dots + cross-lane-group add --------------------------------------------------------
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float part = 0.f;
#pragma unroll
            for (int rb = 0; rb < 16; ++rb) {
                f32x4 w = *(const f32x4*)(bias + HEAD_W + 256 * o + 16 * rb + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(w[r], Y[rb][r], part);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            float tot = part + bias[HEAD_B + o];
            if (sdf && valid && q == 0) sdf[p * NOUT + o] = tot;
        }
    }
#if !defined(D3H_EMULATED)
    __builtin_amdgcn_s_waitcnt(0x0f70);      // the prefetch issued for a tile that never came is an LDS write: let it land before the LDS is released
#endif
}

// ------------------------------------------------------------------------------------------------
// C ABI (include/d3h.h)
// ------------------------------------------------------------------------------------------------
// dwords of a wpack3 buffer (d3h_sdf_mlp_pack3)
extern "C" int64_t d3h_sdf_mlp_wpack3_dwords(void) { return X3_WPACK_DWORDS; }

// wpack3 [d3h_sdf_mlp_wpack3_dwords()] (overwritten) = the weights of d3h_sdf_mlp_pack, each as three bf16 planes in the fragment order of
// d3h_sdf_mlp_fwd_x3, followed by the fp32 biases and head
extern "C" int d3h_sdf_mlp_pack3(const float* w0, const float* b0, const float* wh, const float* bh, const float* w4, const float* b4,
                                 const float* w7, const float* b7, unsigned* wpack3, void* stream) {
    if (!w0 || !b0 || !wh || !bh || !w4 || !b4 || !w7 || !b7 || !wpack3) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack3_kernel<3>, dim3(d3h_cdiv(X3_WPACK_DWORDS, 256)), dim3(256), 0, (hipStream_t)stream, w0, b0, wh, bh, w4, b4, w7,
                       b7, wpack3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// dwords of a wpackh2 buffer (d3h_sdf_mlp_pack_h2)
extern "C" int64_t d3h_sdf_mlp_wpackh2_dwords(void) { return XP<2>::WPACK_DWORDS; }

// wpackh2 [d3h_sdf_mlp_wpackh2_dwords()] (overwritten) = the weights of d3h_sdf_mlp_pack, each as TWO fp16 planes (value + residual scaled by
// 2^11; sdf_mlp_x3.h "h2") in the fragment order of d3h_sdf_mlp_fwd_h2, followed by the fp32 biases and head.  Weights beyond +-16384 (or NaN)
// are packed as NaN: the sweep then returns NaN instead of an overflowed value.
extern "C" int d3h_sdf_mlp_pack_h2(const float* w0, const float* b0, const float* wh, const float* bh, const float* w4, const float* b4,
                                   const float* w7, const float* b7, unsigned* wpackh2, void* stream) {
    if (!w0 || !b0 || !wh || !bh || !w4 || !b4 || !w7 || !b7 || !wpackh2) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack3_kernel<2>, dim3(d3h_cdiv(XP<2>::WPACK_DWORDS, 256)), dim3(256), 0, (hipStream_t)stream, w0, b0, wh, bh, w4, b4,
                       w7, b7, wpackh2);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

template <int NP>
static int fwd_xp_launch(const float* x, const float* deform, float disp, const unsigned* wpack, float* sdf, float* xdef, float* act, int64_t n, int max_cus,
                         void* stream) {
    if (n < 0 || (n > 0 && (!x || !wpack || !sdf))) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, max_cus);
    const int kt = d3h_ktime_begin(D3H_KT_SDF_FWD, n, (hipStream_t)stream);
#define X3_FWD(SMALL_, SAVE_)                                                                                                                          \
    hipLaunchKernelGGL((sdf_mlp_fwd_x3_kernel<false, SMALL_, SAVE_, NP>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, x, deform, disp, wpack, sdf, \
                       xdef, act, n, ntiles, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const int*)nullptr,          \
                       (const int*)nullptr, 1.0f)
    if (ntiles >= 1024) {
        if (act) X3_FWD(0, true); else X3_FWD(0, false);
    } else {
        if (act) X3_FWD(1, true); else X3_FWD(1, false);
    }
#undef X3_FWD
    d3h_ktime_end(kt, (hipStream_t)stream);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// d3h_sdf_mlp_fwd with the layer GEMMs on the bf16 matrix pipe (fp32 accuracy, sdf_mlp_x3.h); same outputs, same `act` layout
extern "C" int d3h_sdf_mlp_fwd_x3(const float* x, const float* deform, float disp, const unsigned* wpack3, float* sdf, float* xdef, float* act,
                                  int64_t n, int max_cus, void* stream) {
    return fwd_xp_launch<3>(x, deform, disp, wpack3, sdf, xdef, act, n, max_cus, stream);
}

// the same on the fp16 matrix pipe with the two-plane split: three v_mfma_f32_16x16x32_f16 per 16x16x32 block instead of six bf16 ones, fp32
// accuracy on operands of O(1) scale (sdf_mlp_x3.h "h2"); wpackh2 from d3h_sdf_mlp_pack_h2; same outputs, same `act` layout
extern "C" int d3h_sdf_mlp_fwd_h2(const float* x, const float* deform, float disp, const unsigned* wpackh2, float* sdf, float* xdef, float* act,
                                  int64_t n, int max_cus, void* stream) {
    return fwd_xp_launch<2>(x, deform, disp, wpackh2, sdf, xdef, act, n, max_cus, stream);
}

// tangent pass of the eikonal term on the bf16 pipe (internal to d3h_sdf_mlp_eik_bwd in sdf_mlp_bwd.hip)
// planes: what wpack3 is (3: d3h_sdf_mlp_pack3, 2: d3h_sdf_mlp_pack_h2 -- then `uscale` scales the direction, see the kernel)
int d3h_sdf_mlp_jvp_x3_launch(const float* x, const float* udir, const unsigned* wpack3, int planes, float uscale, const float* act, const float* dz, float* tb,
                              float* eb, int64_t n, int max_cus, hipStream_t s) {
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, max_cus);
    const int kt = d3h_ktime_begin(D3H_KT_SDF_TANGENT, n, s);
    if (planes == 2)
        hipLaunchKernelGGL((sdf_mlp_fwd_x3_kernel<true, 1, false, 2>), dim3(grid), dim3(NTHREADS), 0, s, x, (const float*)nullptr, 0.f, wpack3, (float*)nullptr,
                           (float*)nullptr, (float*)act, n, ntiles, udir, dz, tb, eb, (const int*)nullptr, (const int*)nullptr, uscale);
    else
    hipLaunchKernelGGL((sdf_mlp_fwd_x3_kernel<true, 1, false, 3>), dim3(grid), dim3(NTHREADS), 0, s, x, (const float*)nullptr, 0.f, wpack3, (float*)nullptr,
                       (float*)nullptr, (float*)act, n, ntiles, udir, dz, tb, eb, (const int*)nullptr, (const int*)nullptr, 1.0f);
    d3h_ktime_end(kt, s);
    return (int)hipGetLastError();
}

// forward with the activation save over a LIST of 16-point tiles (count on the device): the recompute pass of d3h_sdf_mlp_bwd.  The training
// sweep itself runs without the save (1.88 GB of activations per 262 144 points, of which the sparse backward read ~15 %).
int d3h_sdf_mlp_fwd_x3_list_launch(const float* x, const float* deform, float disp, const unsigned* wpack3, int planes, float* act, int64_t n,
                                   const int* tile_list, const int* tile_count, hipStream_t s) {
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, 0);
    const int kt = d3h_ktime_begin(D3H_KT_SDF_FWD_RECOMPUTE, n, s);
    if (planes == 2)          // `wpack3` is a pack of d3h_sdf_mlp_pack_h2: the sweep this pass repeats ran d3h_sdf_mlp_fwd_h2
        hipLaunchKernelGGL((sdf_mlp_fwd_x3_kernel<false, 1, true, 2>), dim3(grid), dim3(NTHREADS), 0, s, x, deform, disp, wpack3, (float*)nullptr,
                           (float*)nullptr, act, n, ntiles, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, tile_list, tile_count, 1.0f);
    else
    hipLaunchKernelGGL((sdf_mlp_fwd_x3_kernel<false, 1, true, 3>), dim3(grid), dim3(NTHREADS), 0, s, x, deform, disp, wpack3, (float*)nullptr, (float*)nullptr,
                       act, n, ntiles, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, tile_list, tile_count, 1.0f);
    d3h_ktime_end(kt, s);
    return (int)hipGetLastError();
}

#endif  // D3H_MLP_NOUT == 1
