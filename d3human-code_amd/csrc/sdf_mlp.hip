// sdf_mlp.hip -- fused positional-encoding + SDF MLP query for gfx950 (MI355X), forward.
//
// Replaces, for the reference's fixed network shape (train.py:1618-1621: n_freq 6, n_hidden 6,
// d_hidden 256, skip_in [3]):
//   geometry/embedding.py:21-38   Embedding.forward      x -> [x, sin(2^k x), cos(2^k x)]_{k<6}   (39)
//   geometry/mlp.py:34-45         MLP.forward            8 Linear, Softplus(beta=100) between
//   geometry/hmsdf.py:433-444     v_deformed = verts + max_displacement*deform; chunked sweep
//
// Design (MI355X-first, see DESIGN.md §SDF query):
//  * Everything is computed TRANSPOSED: H_l^T[256 x pts] = W_l[256 x K] * H_{l-1}^T[K x pts] on the exact-f32
//    matrix pipe (v_mfma_f32_16x16x4_f32).  One wave owns 16 points and all 256 features, so the
//    output accumulators of layer l (lane = point, registers = features) ARE the B operands of
//    layer l+1 -- activations never leave the register file between layers (no LDS round trip).
//    2 x 64 accumulator registers per wave => 8 waves per workgroup, TWO waves per SIMD: one wave's softplus
//    epilogue (VALU) and LDS waits overlap the other wave's MFMA stream.
//  * Weights are pre-packed once per optimiser step (d3h_sdf_mlp_pack) into the exact order the
//    A-operand fragments are consumed; the kernel streams them HBM/L2 -> LDS in 24-38 KB chunks
//    (double buffered, one barrier per chunk), each lane reads one 16-B fragment (4 k-steps) per
//    ds_read_b128, conflict-free (a wave reads 1 KiB contiguous).
//  * Points are read once (12 B) and the result written once (4 B): 16 algorithmic bytes/point.
//    With `act` != NULL the seven post-activation tensors are stored for the backward pass in the
//    register-tile order ("tile-packed": 1 KiB per wave-instruction, fully coalesced).
//
// k-order inside a dot product: feature f = 16*blk + 4*q + r  <->  (acc block blk, register r, lane group q = lane>>4),
// i.e. the D layout of the 16x16 MFMA (row = 4*(lane>>4) + r); see sdf_mlp_layout.h.
#include "sdf_mlp_dev.h"

using namespace D3H_MLP_NS;

// ------------------------------------------------------------------------------------------------
// pack: PyTorch nn.Linear weights ([out][in] row-major, geometry/mlp.py:13-31) -> fragment order
// ------------------------------------------------------------------------------------------------
__global__ void sdf_mlp_pack_kernel(const float* __restrict__ w0, const float* __restrict__ b0,   // [256][39]
                                    const float* __restrict__ wh, const float* __restrict__ bh,   // 5 x [256][256] (net.2,4,6,10,12)
                                    const float* __restrict__ w4, const float* __restrict__ b4,   // [256][295] (net.8)
                                    const float* __restrict__ w7, const float* __restrict__ b7,   // [1][256], [1]
                                    float* __restrict__ wpack) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= WPACK_FLOATS) return;
    float v = 0.f;
    if (idx < OFF_BIAS) {
        int l = layer_of_offset(idx);
        int local = idx - layer_offset(l);
        int r = local & 3, lane = (local >> 2) & 63, rest = local >> 8;          // rest = flat (chunk, rbl, blk)
        int i = lane & 15, q = lane >> 4;
        int nblk = (l == 0) ? EMB_BLKS : ((l == 4) ? SKIP_BLKS : 16);
        int blk = rest % nblk, rbg = rest / nblk;                                 // rbg = global 16-row block (chunks are contiguous)
        int out = 16 * rbg + i;
        int in = 16 * blk + 4 * q + r;
        if (l == 0) {
            if (in < EMB_DIM) v = w0[out * EMB_DIM + in];
        } else if (l == 4) {
            if (blk < 16) v = w4[out * (256 + EMB_DIM) + in];
            else {
                int e = in - 256;
                if (e < EMB_DIM) v = w4[out * (256 + EMB_DIM) + 256 + e];
            }
        } else {
            int hi = (l < 4) ? (l - 1) : (l - 2);   // index into the 5 plain hidden layers
            v = wh[(size_t)hi * 65536 + out * 256 + in];
        }
    } else {
        int j = idx - OFF_BIAS;
        if (j < 256) v = b0[j];
        else if (j < 256 * 7) {
            int l = j >> 8, f = j & 255;
            if (l == 4) v = b4[f];
            else v = bh[((l < 4) ? (l - 1) : (l - 2)) * 256 + f];
        } else if (j < HEAD_B) v = w7[j - HEAD_W];                       // [NOUT][256]
        else if (j < HEAD_B + NOUT) v = b7[j - HEAD_B];
    }
    wpack[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// acc += W_rows[:, emb part] * emb   (emb = 48 padded positional-encoding features, 12 per lane group; wl -> [blk 3][lane 64][4])
__device__ __forceinline__ void mac_emb(f32x4& acc, const float (&emb)[4 * EMB_BLKS], const float* wl, int lane) {
#pragma unroll
    for (int b = 0; b < EMB_BLKS; ++b) {
        f32x4 a = *(const f32x4*)(wl + (b * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], emb[4 * b + r], acc, 0, 0, 0);
    }
}

// JVP epilogue (eikonal second-order pass): q = W t_prev (in v); s = softplus'(z) from the saved h; t = s q;
// e = softplus''(z) d q = 100 (1 - s) (s d) q with (s d) = dz from the gradient pass.  Stores t and e tile-packed; v <- t.
__device__ __forceinline__ void epilogue_jvp(f32x4& v, const float* act_l, const float* dz_l, float* t_l, float* e_l, int rb, int lane) {
    size_t off = (size_t)(rb * 64 + lane) * 4;
    f32x4 hh = *(const f32x4*)(act_l + off);
    f32x4 dd = *(const f32x4*)(dz_l + off);
    f32x4 to, eo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sg = dsoftplus_from_h(hh[r]);
        float qv = v[r];
        to[r] = sg * qv;
        eo[r] = 100.0f * (1.0f - sg) * dd[r] * qv;
        v[r] = to[r];
    }
    *(f32x4*)(t_l + off) = to;
    *(f32x4*)(e_l + off) = eo;
}

// the same with h and dz already fetched (the tangent sweep prefetches them into LDS during the chunk's MFMA loop)
__device__ __forceinline__ void epilogue_jvp_pre(f32x4& v, const f32x4 hh, const f32x4 dd, float* t_l, float* e_l, int rb, int lane) {
    size_t off = (size_t)(rb * 64 + lane) * 4;
    f32x4 to, eo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sg = dsoftplus_from_h(hh[r]);
        float qv = v[r];
        to[r] = sg * qv;
        eo[r] = 100.0f * (1.0f - sg) * dd[r] * qv;
        v[r] = to[r];
    }
    *(f32x4*)(t_l + off) = to;
    *(f32x4*)(e_l + off) = eo;
}

// bias + softplus in place on one 16-feature block; optional tile-packed save for the backward pass
__device__ __forceinline__ void epilogue(f32x4& v, const float* bias_l, int rb, int lane, float* act_tile_layer) {
#ifdef D3H_PROBE_NO_EPI
    return;
#endif
    f32x4 b = *(const f32x4*)(bias_l + 16 * rb + 4 * (lane >> 4));
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r] = softplus100(v[r] + b[r]);
        v[r] = o[r];
    }
    // streamed once, read back only by the backward: a non-temporal store keeps the 1.88 GB of saved activations from evicting the
    // weight chunks every workgroup re-reads from L2
    if (act_tile_layer) __builtin_nontemporal_store(o, (f32x4*)(act_tile_layer + (rb * 64 + lane) * 4));
}

// JVP = false: the SDF query.  JVP = true: the tangent pass of the eikonal term (see sdf_mlp_bwd.hip, d3h_sdf_mlp_eik_bwd): the same
// weight stream and register-resident chain, input = J_emb(x) u, no bias, epilogue_jvp; `act` / `dzb` are read, `tb` / `eb` written.
// SMALL = the instantiation used for launches of fewer than 1024 point tiles (the 50 000 eikonal samples); profiler summaries, which
// aggregate by kernel name, thereby keep the full-grid sweeps (the roofline figure of bench.py) apart from them.  It also selects the
// BALANCED assignment of 16-point wave tiles: in round r, wave w of workgroup b owns wave tile r * 8G + w * G + b (G workgroups), so the
// ragged last round is spread over ALL workgroups with the idle waves skipping their MFMAs (wave-uniform `on`; they still take part in
// the weight staging and its barriers).  With whole 128-point tiles per workgroup 50 000 points = 391 tiles took two full rounds on 256
// CUs; balanced, the second round has 4.2 of 8 waves busy per workgroup, one per SIMD (tools/probe/simd_map.hip: waves w and w + 4 of a
// workgroup share a SIMD).  Measured: the reverse sweep of the eikonal term 572 -> 527 us, the forward and tangent sweeps unchanged --
// a round lasts as long with one wave per SIMD as with two, i.e. its duration is a wave's own dependency chain (LDS fragment reads,
// epilogue, barrier per chunk), not the matrix pipe, which two waves per SIMD keep ~70 % busy between them.
template <bool JVP, int SMALL>
__global__ __launch_bounds__(NTHREADS, 2) void sdf_mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ deform,
                                                                 float disp, const float* __restrict__ wpack,
                                                                 float* __restrict__ sdf, float* __restrict__ xdef,
                                                                 float* __restrict__ act, int64_t n, int ntiles,
                                                                 const float* __restrict__ udir, const float* __restrict__ dzb,
                                                                 float* __restrict__ tb, float* __restrict__ eb) {
    __shared__ __attribute__((aligned(16))) float wbuf[2][CHUNK_MAX_FLOATS];
    __shared__ __attribute__((aligned(16))) float bias[BIAS_FLOATS];
    // tangent sweep: h and dz of the chunk's two row blocks, fetched global -> LDS while the chunk's MFMAs run (per wave: 4 x 1 KiB).
    // Loaded inside the epilogue they cost a global round trip per block with nothing to hide it: 112 of them per tile.
    __shared__ __attribute__((aligned(16))) float jpf[JVP ? NWAVES * 4 * 256 : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;

    for (int i = tid; i < BIAS_FLOATS; i += NTHREADS) bias[i] = wpack[OFF_BIAS + i];

    // (weights are staged global -> registers -> LDS: the direct global_load_lds path of sdf_mlp_dev.h measured 3 % SLOWER in this
    // structure -- 1.995 vs 1.934 ms per 262 144-point sweep -- although it frees 20 VGPRs and removes the spills)
    Stage st;
    int pb = 0;
    SDF_STAGE_ISSUE(st, wpack, wbuf[0], L0_CHUNK_FLOATS / 4, tid);
    SDF_STAGE_COMMIT(st, wbuf[0], L0_CHUNK_FLOATS / 4, tid);   // also publishes bias[]

    f32x4 X[16], Y[16];
#if D3H_SDF_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    constexpr bool BAL = JVP || SMALL != 0 || NWAVES != 8;
    const int64_t n16 = (int64_t)ntiles * 8;        // incl. the padding wave tiles of the last 128-point tile (the backward reads them)
    const int G = (int)gridDim.x;
    const int nrounds = BAL ? (int)((n16 + NWAVES * (int64_t)G - 1) / (NWAVES * (int64_t)G)) : (ntiles - (int)blockIdx.x + G - 1) / G;
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int tile = (int)blockIdx.x + rnd * G;
        const int64_t t16 = BAL ? ((int64_t)rnd * NWAVES * G + (int64_t)wave * G + blockIdx.x) : ((int64_t)tile * 8 + wave);   // 16-point tile index
        const bool on = !BAL || t16 < n16;                       // wave-uniform
        const int64_t p = t16 * 16 + (lane & 15);
        const bool valid = p < n;
        float* act_tile = act ? act + t16 * ACT_TILE_FLOATS : nullptr;
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
        float emb[4 * EMB_BLKS];
#pragma unroll
        for (int b = 0; b < EMB_BLKS; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) emb[4 * b + r] = emb_feature(16 * b + 4 * q + r, x0, x1, x2);
        if (JVP) {
            float u0 = 0.f, u1 = 0.f, u2 = 0.f;
            if (valid) { u0 = udir[3 * p + 0]; u1 = udir[3 * p + 1]; u2 = udir[3 * p + 2]; }
#pragma unroll
            for (int b = 0; b < EMB_BLKS; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) emb[4 * b + r] = emb_tangent(16 * b + 4 * q + r, x0, x1, x2, u0, u1, u2);
        }

        // ---- layer 0: emb(39) -> X, two chunks of 8 row blocks ----------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* nsrc = (c == 0) ? wpack + L0_CHUNK_FLOATS : wpack + OFF_L1;
            const int nn4 = ((c == 0) ? L0_CHUNK_FLOATS : HID_CHUNK_FLOATS) / 4;
            SDF_STAGE_ISSUE(st, nsrc, wbuf[pb ^ 1], nn4, tid);
            const float* wl = wbuf[pb];
            if (on) {
#pragma unroll
                for (int rbl = 0; rbl < 8; ++rbl) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    mac_emb(acc, emb, wl + rbl * (EMB_BLKS * 256), lane);
                    X[8 * c + rbl] = acc;
                }
            }
            SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], nn4, tid);
            pb ^= 1;
            if (on) {
#pragma unroll
                for (int rbl = 0; rbl < 8; ++rbl) {
                    if (JVP) epilogue_jvp(X[8 * c + rbl], act_tile, dz_tile, t_tile, e_tile, 8 * c + rbl, lane);
                    else epilogue(X[8 * c + rbl], bias, 8 * c + rbl, lane, act_tile);
                }
            }
        }

        // ---- layers 1..6, two per iteration: X -> Y (l = 1,3,5), Y -> X (l = 2,4,6) --------------
        // Stagger: waves w and w + 4 share a SIMD and run the same chunk schedule, so without care both reach their LDS bursts, the
        // barrier and the softplus epilogue together and the matrix pipe idles meanwhile.  Waves 4..7 ("late") therefore run the
        // epilogue of a chunk half a chunk later, in the middle of the NEXT chunk's MFMA loop, while waves 0..3 run it right after
        // the barrier: one partner is always issuing MFMAs while the other does VALU / store work.  The values wait in X / Y (they are
        // not consumed before the next layer; the last chunk's pair is needed at k-blocks 14, 15 of the next layer's first chunk,
        // after the hook at k-block 8).  Results are bit-identical.
        const bool late = NWAVES == 8 && wave >= 4;
        auto epi = [&](f32x4& v, int l, int rb) {
            if (JVP) epilogue_jvp(v, act_tile + l * ACT_LAYER_FLOATS, dz_tile + l * ACT_LAYER_FLOATS, t_tile + l * ACT_LAYER_FLOATS,
                                  e_tile + l * ACT_LAYER_FLOATS, rb, lane);
            else epilogue(v, bias + 256 * l, rb, lane, act_tile ? act_tile + l * ACT_LAYER_FLOATS : nullptr);
        };
        // One layer body, executed six times (X -> Y, then X = Y: 64 register moves against 1024 MFMAs): the fully alternating
        // X -> Y / Y -> X form was 98 KB of straight-line code, more than the 64 KB instruction cache.
#pragma unroll 1
        for (int l = 1; l <= 6; ++l) {
            const bool skip = (l == 4);
            const int this_chunk = skip ? SKIP_CHUNK_FLOATS : HID_CHUNK_FLOATS;
            const int nblk = skip ? SKIP_BLKS : 16;
            const float* lbase = wpack + layer_offset(l);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // after the last chunk of layer 6 comes layer 0 of the next tile
                const float* nsrc = (c < 7) ? lbase + (c + 1) * this_chunk : ((l == 6) ? wpack : wpack + layer_offset(l + 1));
                const int nn4 = ((c < 7) ? this_chunk : ((l == 6) ? L0_CHUNK_FLOATS : ((l == 3) ? SKIP_CHUNK_FLOATS : HID_CHUNK_FLOATS))) / 4;
                SDF_STAGE_ISSUE(st, nsrc, wbuf[pb ^ 1], nn4, tid);
                if (JVP && on) {
                    float* pf = jpf + wave * (4 * 256);
                    const size_t o0 = (size_t)l * ACT_LAYER_FLOATS + (size_t)((2 * c) * 64 + lane) * 4;
                    D3H_GLDS16(act_tile + o0, pf);
                    D3H_GLDS16(dz_tile + o0, pf + 256);
                    D3H_GLDS16(act_tile + o0 + 256, pf + 512);
                    D3H_GLDS16(dz_tile + o0 + 256, pf + 768);
                }
                const float* wl = wbuf[pb];
                if (on) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mac_hidden2_mid(acc0, acc1, X, wl, nblk * 256, lane, [&] {
                        if (late && !JVP) {
                            if (c > 0) { epi(Y[2 * c - 2], l, 2 * c - 2); epi(Y[2 * c - 1], l, 2 * c - 1); }
                            else if (l > 1) { epi(X[14], l - 1, 14); epi(X[15], l - 1, 15); }    // layer 0 is not staggered
                        }
                    });
                    if (skip) {                                                          // mlp.py:40-41 cat([x, emb])
                        mac_emb(acc0, emb, wl + 16 * 256, lane);
                        mac_emb(acc1, emb, wl + (nblk + 16) * 256, lane);
                    }
                    Y[2 * c] = acc0;
                    Y[2 * c + 1] = acc1;
                }
                if (JVP && !D3H_SDF_GLDS) __builtin_amdgcn_s_waitcnt(0x0f70);      // the prefetch above must have landed (vmcnt(0))
                SDF_STAGE_COMMIT(st, wbuf[pb ^ 1], nn4, tid);
                pb ^= 1;
                if (JVP) {          // no stagger here: the operands of this chunk's epilogue sit in the single prefetch buffer
                    if (on) {
                        const float* pf = jpf + wave * (4 * 256) + lane * 4;
                        float* tl = t_tile + l * ACT_LAYER_FLOATS;
                        float* el = e_tile + l * ACT_LAYER_FLOATS;
                        epilogue_jvp_pre(Y[2 * c], *(const f32x4*)pf, *(const f32x4*)(pf + 256), tl, el, 2 * c, lane);
                        epilogue_jvp_pre(Y[2 * c + 1], *(const f32x4*)(pf + 512), *(const f32x4*)(pf + 768), tl, el, 2 * c + 1, lane);
                    }
                } else if (on && !late) { epi(Y[2 * c], l, 2 * c); epi(Y[2 * c + 1], l, 2 * c + 1); }
            }
#pragma unroll
            for (int rb = 0; rb < 16; ++rb) X[rb] = Y[rb];
        }
        if (on && late && !JVP) { epi(X[14], 6, 14); epi(X[15], 6, 15); }          // flush the deferred pair of layer 6

        if (JVP || !on) continue;     // the tangent of the head (W7 . t_6) is not needed: the eikonal loss does not depend on f itself
        // ---- layer 7: 256 -> NOUT (net.14), VALU dots + cross-lane-group add ------------------------------
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float part = 0.f;
#pragma unroll
            for (int rb = 0; rb < 16; ++rb) {
                f32x4 w = *(const f32x4*)(bias + HEAD_W + 256 * o + 16 * rb + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(w[r], X[rb][r], part);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            float tot = part + bias[HEAD_B + o];
            if (valid && q == 0) sdf[p * NOUT + o] = tot;
        }
    }
#if D3H_SDF_GLDS && !defined(D3H_PROBE_NO_STAGE)
    __builtin_amdgcn_s_waitcnt(0x0f70);      // the prefetch issued for a tile that never came is an LDS write: let it land before the LDS is released
#endif
}

// ------------------------------------------------------------------------------------------------
// C ABI (include/d3h.h)
// ------------------------------------------------------------------------------------------------
extern "C" int64_t d3h_sdf_mlp_wpack_floats(void) { return WPACK_FLOATS; }
extern "C" int64_t d3h_sdf_mlp_act_floats(int64_t n) { return ((n + TILE_PTS - 1) / TILE_PTS) * 8 * (int64_t)ACT_TILE_FLOATS; }

extern "C" int d3h_sdf_mlp_pack(const float* w0, const float* b0, const float* wh, const float* bh, const float* w4,
                                const float* b4, const float* w7, const float* b7, float* wpack, void* stream) {
    if (!w0 || !b0 || !wh || !bh || !w4 || !b4 || !w7 || !b7 || !wpack) return D3H_ERR_ARG;
    hipLaunchKernelGGL(sdf_mlp_pack_kernel, dim3(d3h_cdiv(WPACK_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, w0, b0, wh,
                       bh, w4, b4, w7, b7, wpack);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// max_cus: launches of fewer than 1024 tiles use at most this many workgroups (= CUs); 0 = the whole chip (sdf_mlp_layout.h)
extern "C" int d3h_sdf_mlp_fwd(const float* x, const float* deform, float disp, const float* wpack, float* sdf,
                               float* xdef, float* act, int64_t n, int max_cus, void* stream) {
    if (n < 0 || (n > 0 && (!x || !wpack || !sdf))) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, max_cus);
#if D3H_MLP_NOUT == 1
    const int kt = d3h_ktime_begin(D3H_KT_SDF_FWD, n, (hipStream_t)stream);
#endif
    if (ntiles >= 1024)
        hipLaunchKernelGGL((sdf_mlp_fwd_kernel<false, 0>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, x, deform, disp, wpack, sdf, xdef,
                           act, n, ntiles, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    else
        hipLaunchKernelGGL((sdf_mlp_fwd_kernel<false, 1>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, x, deform, disp, wpack, sdf, xdef,
                           act, n, ntiles, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
#if D3H_MLP_NOUT == 1
    d3h_ktime_end(kt, (hipStream_t)stream);
#endif
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

#if D3H_MLP_NOUT == 1
// tangent pass of the eikonal term (internal to d3h_sdf_mlp_eik_bwd in sdf_mlp_bwd.hip)
int d3h_sdf_mlp_jvp_launch(const float* x, const float* udir, const float* wpack, const float* act, const float* dz, float* tb, float* eb,
                           int64_t n, int max_cus, hipStream_t s) {
    int ntiles = (int)((n + TILE_PTS - 1) / TILE_PTS);
    int grid = sdf_chain_grid(ntiles, max_cus);
    const int kt = d3h_ktime_begin(D3H_KT_SDF_TANGENT, n, s);
    hipLaunchKernelGGL((sdf_mlp_fwd_kernel<true, 1>), dim3(grid), dim3(NTHREADS), 0, s, x, (const float*)nullptr, 0.f, wpack, (float*)nullptr,
                       (float*)nullptr, (float*)act, n, ntiles, udir, dz, tb, eb);
    d3h_ktime_end(kt, s);
    return (int)hipGetLastError();
}
#endif
