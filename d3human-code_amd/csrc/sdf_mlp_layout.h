// sdf_mlp_layout.h -- packed-weight and saved-activation layouts of the fused SDF MLP kernels (v2: 16x16x4 MFMA, 16-point wave tiles).
//
// Feature <-> register mapping of a 16-feature block `rb` of one wave tile (16 points): lane = j + 16 q (j = point, q = 0..3),
// register r (0..3)  <->  feature 16 rb + 4 q + r.  This is the D layout of v_mfma_f32_16x16x4_f32 (col = lane & 15,
// row = 4 (lane >> 4) + r), and -- used as the B operand of the next layer -- k-step (blk, r) pairs lane group q with input
// feature 16 blk + 4 q + r, so the packed A fragment is  element (rb, blk, lane = i + 16 q, r) = W[16 rb + i][16 blk + 4 q + r].
//
// wpack (floats), consumed in stream order by sdf_mlp_fwd_kernel:
//   layer 0 (net.0, 39->256)      2 chunks of [rbl 8][blk 3][lane 64][r 4]   (embedding padded to 48)     2 x 6144
//   layers 1,2,3,5,6 (256->256)   8 chunks of [rbl 2][blk 16][lane 64][r 4]                               8 x 8192 each
//   layer 4 (net.8, 295->256)     8 chunks of [rbl 2][blk 19][lane 64][r 4]  (blk >= 16: embedding)      8 x 9728
//   tail: bias0..bias6 (7 x 256), W7 (NOUT x 256), b7 (NOUT), padded to a multiple of 4
// wpackT (backward data, dH_{l-1}^T = W_l^T dZ_l^T), consumed in the order L6, L5, L4 (8 hidden + 2 embedding in-chunks), L3, L2,
//   L1, L0 (2 embedding in-chunks); every chunk [rbl 2][blk 16][lane 64][r 4] with
//   element = W_l[out = 16 blk + 4 q + r][in = 32 c + 16 rbl + i]   (embedding in-features: 32 c' + 16 rbl + i, zero beyond 38)
// act / dz (floats), per 16-point tile: [layer 7][rb 16][lane 64][r 4]  (1 KiB per wave-instruction)
//
// The same sources build two networks (compile-time configuration, see deform_mlp.hip):
//   SDF network   MLP(n_freq 6, d_out 1)           geometry/mlp.py:10-45      EMB_DIM 39, 3 embedding blocks   (default)
//   offset network MLP_deform(n_freq 8, d_out 3)   geometry/mlp.py:77-118     EMB_DIM 51, 4 embedding blocks; its 136-float pose code is
//                                                  constant over the points and is folded into the first bias by the host wrapper
#pragma once
#include <cstdlib>

#ifndef D3H_MLP_NFREQ
#define D3H_MLP_NFREQ 6
#endif
#ifndef D3H_MLP_NOUT
#define D3H_MLP_NOUT 1
#endif
#ifndef D3H_MLP_NS
#define D3H_MLP_NS d3h_mlp
#endif

namespace D3H_MLP_NS {

constexpr int NFREQ = D3H_MLP_NFREQ;
constexpr int NOUT = D3H_MLP_NOUT;                   // outputs of the head (net.14)
constexpr int EMB_DIM = 3 + 6 * NFREQ;               // 39 (51)
constexpr int EMB_BLKS = (EMB_DIM + 15) / 16;        // 3 (4): padded embedding features / 16 per block
constexpr int L0_CHUNK_FLOATS = 8 * EMB_BLKS * 256;  // 6144
constexpr int HID_CHUNK_FLOATS = 2 * 16 * 256;       // 8192
constexpr int SKIP_BLKS = 16 + EMB_BLKS;             // 19
constexpr int SKIP_CHUNK_FLOATS = 2 * SKIP_BLKS * 256;   // 9728
constexpr int CHUNK_MAX_FLOATS = SKIP_CHUNK_FLOATS;
// waves per workgroup of the register-resident chain kernels (forward / tangent / data-backward): 8 = two per SIMD (default);
// 4 = one per SIMD, which leaves half of every SIMD's register file to co-resident kernels of other streams (experiment)
#ifndef D3H_SDF_NWAVES
#define D3H_SDF_NWAVES 8
#endif
constexpr int NWAVES = D3H_SDF_NWAVES;
constexpr int NTHREADS = 64 * NWAVES;
constexpr int STAGE_F4 = (CHUNK_MAX_FLOATS / 4 + NTHREADS - 1) / NTHREADS;   // 5 float4 per thread per chunk
constexpr int OFF_L1 = 2 * L0_CHUNK_FLOATS;
constexpr int OFF_L2 = OFF_L1 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L3 = OFF_L2 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L4 = OFF_L3 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L5 = OFF_L4 + 8 * SKIP_CHUNK_FLOATS;
constexpr int OFF_L6 = OFF_L5 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_BIAS = OFF_L6 + 8 * HID_CHUNK_FLOATS;
constexpr int HEAD_W = 7 * 256;                      // within the tail: W7 [NOUT][256], then b7 [NOUT]
constexpr int HEAD_B = HEAD_W + NOUT * 256;
constexpr int BIAS_FLOATS = (HEAD_B + NOUT + 3) / 4 * 4;   // 2052 (2564)
constexpr int WPACK_FLOATS = OFF_BIAS + BIAS_FLOATS;
constexpr int ACT_LAYER_FLOATS = 16 * 64 * 4;        // 4096 = 16 points x 256 features
constexpr int ACT_TILE_FLOATS = 7 * ACT_LAYER_FLOATS;   // per 16-point tile
constexpr int TILE_PTS = 128;                        // points per workgroup tile (8 waves x 16)
// Workgroups of a chain launch: one persistent workgroup per CU.  Launches of fewer than 1024 tiles (the eikonal samples, a rank's shard
// of the grid sweep) take `max_cus` from their caller (0 = the whole chip; D3H_SDF_EIK_GRID overrides for experiments): the chain kernels
// take a CU's whole register file, so kernels of another stream only run on the CUs a chain launch leaves out.  A per-call argument, not
// process state: two scenes or two streams in one process cannot disturb each other.
// Those launches run the BALANCED kernels, which hand out 16-point WAVE tiles (tile r * 8G + w * G + b to wave w of workgroup b in round r)
// and whose time is the busiest SIMD's wave count: below two tiles per CU the launch is spread over twice as many workgroups (4 waves
// each, one per SIMD) -- 6 250 samples (49 tiles) finish in one wave-tile time on 98 CUs instead of two on 49.
inline int sdf_chain_grid(int ntiles, int max_cus) {
    static int env_cap = -1;
    if (env_cap < 0) {
        const char* e = getenv("D3H_SDF_EIK_GRID");
        int v = e ? atoi(e) : 0;
        env_cap = v > 0 ? (v > 256 ? 256 : v) : 0;
    }
    int cap = 256;
    int want = ntiles;
    if (ntiles < 1024) {
        if (env_cap > 0) cap = env_cap;
        else if (max_cus > 0 && max_cus < 256) cap = max_cus;
        want = 2 * ntiles;
    }
    return want < cap ? want : cap;
}

constexpr int T_CHUNK_FLOATS = HID_CHUNK_FLOATS;
constexpr int T_OFF_L6 = 0;
constexpr int T_OFF_L5 = T_OFF_L6 + 8 * T_CHUNK_FLOATS;
constexpr int T_OFF_L4 = T_OFF_L5 + 8 * T_CHUNK_FLOATS;
constexpr int T_OFF_L3 = T_OFF_L4 + 10 * T_CHUNK_FLOATS;
constexpr int T_OFF_L2 = T_OFF_L3 + 8 * T_CHUNK_FLOATS;
constexpr int T_OFF_L1 = T_OFF_L2 + 8 * T_CHUNK_FLOATS;
constexpr int T_OFF_L0 = T_OFF_L1 + 8 * T_CHUNK_FLOATS;
constexpr int WPACKT_FLOATS = T_OFF_L0 + 2 * T_CHUNK_FLOATS;   // 52 chunks

__host__ __device__ inline int t_layer_offset(int l) {
    switch (l) {
        case 6: return T_OFF_L6;
        case 5: return T_OFF_L5;
        case 4: return T_OFF_L4;
        case 3: return T_OFF_L3;
        case 2: return T_OFF_L2;
        case 1: return T_OFF_L1;
        default: return T_OFF_L0;
    }
}
__host__ __device__ inline int t_layer_of_offset(int idx) {
    if (idx < T_OFF_L5) return 6;
    if (idx < T_OFF_L4) return 5;
    if (idx < T_OFF_L3) return 4;
    if (idx < T_OFF_L2) return 3;
    if (idx < T_OFF_L1) return 2;
    if (idx < T_OFF_L0) return 1;
    return 0;
}
__host__ __device__ inline int layer_offset(int l) {
    switch (l) {
        case 0: return 0;
        case 1: return OFF_L1;
        case 2: return OFF_L2;
        case 3: return OFF_L3;
        case 4: return OFF_L4;
        case 5: return OFF_L5;
        case 6: return OFF_L6;
        default: return OFF_BIAS;
    }
}
__host__ __device__ inline int layer_of_offset(int idx) {
    if (idx < OFF_L1) return 0;
    if (idx < OFF_L2) return 1;
    if (idx < OFF_L3) return 2;
    if (idx < OFF_L4) return 3;
    if (idx < OFF_L5) return 4;
    if (idx < OFF_L6) return 5;
    return 6;
}

}  // namespace D3H_MLP_NS
