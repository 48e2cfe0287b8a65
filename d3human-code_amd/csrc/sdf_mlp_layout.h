// sdf_mlp_layout.h -- packed-weight and saved-activation layouts of the fused SDF MLP kernels.
//
// wpack (floats), consumed in stream order by sdf_mlp_fwd_kernel:
//   layer 0 (net.0, 39->256)      [rb 8][g 5][lane 64][k 4]                      10240
//   layers 1,2,3,5,6 (256->256)   per layer 8 chunks (rb) of [g 32][lane 64][k 4]  8 x 8192
//   layer 4 (net.8, 295->256)     per rb [g 37][lane 64][k 4]  (g >= 32: embedding) 8 x 9472
//   tail: bias0..bias6 (7 x 256), W7 (256), b7 (1), pad to 2052
// element (rb, g, lane = i + 32 h, k) = W[32 rb + i][8 g + 4 h + k]: the A fragment of
// v_mfma_f32_32x32x2_f32 for k-step (g, k) -- lane half h supplies input feature 8g + 4h + k.
//
// act (floats), per 32-point tile: [layer 7][rb 8][q 4][lane 64][k 4]; element = post-activation
// feature 32 rb + 8 q + 4 h + k of point (tile*32 + (lane & 31)).
#pragma once

namespace d3h_mlp {

constexpr int EMB_DIM = 39;
constexpr int EMB_GROUPS = 5;                       // 40 padded features / 8 per group
constexpr int L0_FLOATS = 8 * EMB_GROUPS * 256;     // 10240
constexpr int HID_CHUNK_FLOATS = 32 * 256;          // 8192
constexpr int SKIP_CHUNK_FLOATS = (32 + EMB_GROUPS) * 256;   // 9472
constexpr int CHUNK_MAX_FLOATS = L0_FLOATS;
constexpr int STAGE_F4 = CHUNK_MAX_FLOATS / 4 / 256;   // float4 per thread per chunk (256 threads)
constexpr int OFF_L1 = L0_FLOATS;
constexpr int OFF_L2 = OFF_L1 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L3 = OFF_L2 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L4 = OFF_L3 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_L5 = OFF_L4 + 8 * SKIP_CHUNK_FLOATS;
constexpr int OFF_L6 = OFF_L5 + 8 * HID_CHUNK_FLOATS;
constexpr int OFF_BIAS = OFF_L6 + 8 * HID_CHUNK_FLOATS;
constexpr int BIAS_FLOATS = 2052;
constexpr int WPACK_FLOATS = OFF_BIAS + BIAS_FLOATS;
constexpr int ACT_LAYER_FLOATS = 8 * 4 * 64 * 4;    // 8192 = 32 points x 256 features
constexpr int ACT_TILE_FLOATS = 7 * ACT_LAYER_FLOATS;

// ---- transposed pack for the backward-data kernel (dH_{l-1}^T = W_l^T dZ_l^T) ------------------------
// consumed in the order L6, L5, L4 (8 hidden + 2 embedding in-blocks), L3, L2, L1, L0 (2 embedding in-blocks);
// every chunk is [g 32][lane 64][k 4]; element (rb, g, lane = i + 32 h, k) = W_l[out = 8 g + 4 h + k][in = 32 rb + i]
constexpr int T_CHUNK_FLOATS = 32 * 256;            // 8192
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

}  // namespace d3h_mlp
