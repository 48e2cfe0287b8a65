// texmlp.hip -- fused multiresolution-grid encoding + tiny texture MLP (kd/ks lookup) for gfx950.
//
// Replaces render/mlptexture.py:51-107 (MLPTexture3D.sample): bbox normalisation + clamp (:94-96), the tiny-cuda-nn
// HashGrid encoding (:62-79: 5 levels x 2 features, base 16, per-level scale 1.4472692, 2^21 table -- every level is DENSE
// because 71^3 < 2^21), the bias-free 10 -> 32 -> 32 -> 6 ReLU MLP (_MLP :18-41, incl. the x128 input-gradient hook :31) and
// the sigmoid range map (:103).  tiny-cuda-nn is an un-vendored dependency (README.md:30): "parity unpinned", the grid
// semantics below restate its published algorithm and are pinned against oracle/texmlp.py.
//   level l: scale = 16 * s^l - 1, res = ceil(scale) + 1; p = x * scale + 0.5; trilinear over the 8 corners of floor(p);
//   dense index x + y res + z res^2; level tables padded to a multiple of 8 entries.
//
// MI355X design: one thread per shaded pixel, everything in registers -- 40 gathers of 8-B feature pairs (tables total
// 4.3 MB: resident in the XCD L2s), 1536 wave-uniform weights staged once per workgroup in LDS (broadcast reads), sigmoid
// epilogue fused.  Background pixels (mask == 0) are skipped.  The backward recomputes the forward, scatters feature
// gradients with fp32 atomics, and reduces the three weight-gradient outer products over the workgroup's 256 pixels
// through LDS (pitch 33), persistent workgroups keeping the partial dW in registers until the end.
#include <cstdlib>
#include "d3h_common.h"
#include "d3h_h2.h"

namespace {

constexpr int NL = 5, NF = 2, ENC = NL * NF;   // 10
constexpr int HID = 32, OUTC = 6;
constexpr int W1N = HID * ENC, W2N = HID * HID, W3N = OUTC * HID;   // 320, 1024, 192

struct GridCfg {
    float scale[NL];
    int res[NL];
    int offset[NL];      // in entries (pairs of floats)
    int size[NL];        // entries of the level (padded to a multiple of 8); indices wrap modulo this, as tcnn's grid_index does
};

struct TexParams {
    float b0[3], b1[3];  // bbox used by the reference: x_n = (x - b0) / (b1 - b0), clamped to [0, 1]
    float omin[OUTC], omax[OUTC];
    float in_grad_scale; // 128: render/mlptexture.py:31,78,88
    int merge_faces;     // table scatter: fold runs in face-adjacent cells before the atomics (D3H_TEX_MERGE_FACES=0: off, A/B)
};

__device__ __forceinline__ void encode(const GridCfg& g, const float* __restrict__ table, const float (&xn)[3], float (&enc)[ENC]) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        float p[3], fr[3];
        int pg[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            p[d] = fmaf(xn[d], g.scale[l], 0.5f);
            float fl = floorf(p[d]);
            fr[d] = p[d] - fl;
            pg[d] = (int)fl;
        }
        float f0 = 0.f, f1 = 0.f;
        const int res = g.res[l];
        const float2* tab = (const float2*)table + g.offset[l];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float w = 1.f;
            int idx = 0, stride = 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int bit = (c >> d) & 1;
                w *= bit ? fr[d] : (1.f - fr[d]);
                idx += (pg[d] + bit) * stride;
                stride *= res;
            }
            if (idx >= g.size[l]) idx -= g.size[l];     // only x == 1 on level 0 (scale 15 -> corner 16) can wrap
            float2 v = tab[idx];
            f0 = fmaf(w, v.x, f0);
            f1 = fmaf(w, v.y, f1);
        }
        enc[2 * l] = f0;
        enc[2 * l + 1] = f1;
    }
}

__device__ __forceinline__ bool normalise(const TexParams& tp, const float* __restrict__ x, float (&xn)[3], bool (&inside)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = (x[d] - tp.b0[d]) / (tp.b1[d] - tp.b0[d]);
        inside[d] = (v >= 0.f && v <= 1.f);
        xn[d] = fminf(fmaxf(v, 0.f), 1.f);
    }
    return true;
}

// weights: w1 [32][10], w2 [32][32], w3 [6][32] (nn.Linear [out][in]); wave-uniform addresses (scalar loads)
__device__ __forceinline__ void mlp_fwd(const float* sw, const float (&enc)[ENC], float (&z1)[HID], float (&z2)[HID], float (&o)[OUTC]) {
    const float* w1 = sw;
    const float* w2 = sw + W1N;
    const float* w3 = sw + W1N + W2N;
#pragma unroll
    for (int i = 0; i < HID; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < ENC; ++j) a = fmaf(w1[i * ENC + j], enc[j], a);
        z1[i] = a;
    }
#pragma unroll
    for (int i = 0; i < HID; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < HID; ++j) a = fmaf(w2[i * HID + j], fmaxf(z1[j], 0.f), a);
        z2[i] = a;
    }
#pragma unroll
    for (int i = 0; i < OUTC; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < HID; ++j) a = fmaf(w3[i * HID + j], fmaxf(z2[j], 0.f), a);
        o[i] = a;
    }
}

__global__ __launch_bounds__(256) void texmlp_fwd_kernel(GridCfg g, TexParams tp, const float* __restrict__ x, const float* __restrict__ mask,
                                                         const float* __restrict__ table, const float* __restrict__ w, int64_t n,
                                                         float* __restrict__ out, float* __restrict__ enc_out) {
    const float* sw = w;      // wave-uniform addresses: the compiler emits scalar loads and feeds the weights as SGPR operands
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mask && !(mask[i] > 0.f)) {
            if (out) {
#pragma unroll
                for (int c = 0; c < OUTC; ++c) out[i * OUTC + c] = 0.f;
            }
            if (enc_out) {
#pragma unroll
                for (int c = 0; c < ENC; ++c) enc_out[i * ENC + c] = 0.f;
            }
            continue;
        }
        float xn[3], enc[ENC];
        bool inside[3];
        normalise(tp, x + 3 * i, xn, inside);
        encode(g, table, xn, enc);
        if (enc_out) {
#pragma unroll
            for (int c = 0; c < ENC; ++c) enc_out[i * ENC + c] = enc[c];
        }
        if (out && w) {
            float z1[HID], z2[HID], o[OUTC];
            mlp_fwd(sw, enc, z1, z2, o);
#pragma unroll
            for (int c = 0; c < OUTC; ++c) {
                float s = 1.f / (1.f + expf(-o[c]));
                out[i * OUTC + c] = s * (tp.omax[c] - tp.omin[c]) + tp.omin[c];
            }
        }
    }
}

// ---- forward, second version: the MLP on the matrix pipe with the weights resident in registers ----------------------------------------
// The kernel above feeds every weight to the VALU as an SGPR operand: 96 s_load_dwordx16 per wave, each waited for with lgkmcnt(0)
// (scalar loads return out of order, so nothing can be in flight across a wait) in front of the 16 FMAs it feeds -- measured: 97 of the
// kernel's 154 us at 4 x 1024^2 are the MLP, ten times its arithmetic.  Here a wave loads W1 / W2 / W3 ONCE, as A operands of
// v_mfma_f32_32x32x2_f32 (A[m][k]: lane = m + 32 k, so a 32 x 32 matrix is 16 registers, all three 37), and keeps them for every tile.
// One MFMA chain serves 32 pixels (the two lane halves hold the even / odd k of the SAME 32 columns), so the wave's 64 pixels are two
// chains.  The D layout of a layer (lane half h holds output rows rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h in register r) is the B layout of
// the next one when k-step s of the next layer pairs input rows rho(s, 0), rho(s, 1) -- the weight columns are loaded in that order,
// activations never leave their registers.  Only the 10 encoding features must cross the lane halves once (5 swaps).
// fp32 throughout; the sum order inside a dot product differs from the VALU kernel (pairs of products per MFMA), like any GEMM.
__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(256) void texmlp_fwd_mfma_kernel(GridCfg g, TexParams tp, const float* __restrict__ x, const float* __restrict__ mask,
                                                              const float* __restrict__ table, const float* __restrict__ w, int64_t n,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const float* w1 = w;
    const float* w2 = w + W1N;
    const float* w3 = w + W1N + W2N;
    float a1[ENC / 2], a2[16], a3[16];
    bool loaded = false;                     // the weights are fetched by the first tile of this wave that has a covered pixel
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const int64_t p0 = i - lane;
        if (p0 >= n) continue;                                               // (wave-uniform)
        const bool active = i < n && !(mask && !(mask[i] > 0.f));
        const unsigned long long am = __ballot(active);
        float enc[ENC];
#pragma unroll
        for (int c = 0; c < ENC; ++c) enc[c] = 0.f;
        if (active) {
            float xn[3];
            bool inside[3];
            normalise(tp, x + 3 * i, xn, inside);
            encode(g, table, xn, enc);
        }
        // this lane finishes pixel p0 + m of chain 0 and pixel p0 + 32 + m of chain 1: rows 0..3 (h = 0) or 4, 5 (h = 1) of the output
        float o[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (am != 0ull) {
            if (!loaded) {
                loaded = true;
#pragma unroll
                for (int s = 0; s < ENC / 2; ++s) a1[s] = w1[m * ENC + 2 * s + h];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    a2[s] = w2[m * HID + rho(s, h)];
                    a3[s] = (m < OUTC) ? w3[m * HID + rho(s, h)] : 0.f;
                }
            }
            // B operands of layer 1: k-step s holds features 2s (h = 0) and 2s + 1 (h = 1) of the chain's 32 pixels
            float b1[2][ENC / 2];
#pragma unroll
            for (int s = 0; s < ENC / 2; ++s) {
                const float e0 = enc[2 * s], e1 = enc[2 * s + 1];
                const float r = __shfl_xor(h == 0 ? e1 : e0, 32);
                b1[0][s] = h == 0 ? e0 : r;                                  // chain 0: pixels of lanes 0..31
                b1[1][s] = h == 0 ? r : e1;                                  // chain 1: pixels of lanes 32..63
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                if (((am >> (32 * ch)) & 0xffffffffull) == 0ull) continue;   // (wave-uniform) nothing covered in this half
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < ENC / 2; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[ch][s], d, 0, 0, 0);
                f32x16 e;
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) e = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], fmaxf(d[s], 0.f), e, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[s], fmaxf(e[s], 0.f), d, 0, 0, 0);
                const bool act = (am >> (32 * ch + m)) & 1ull;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = r + 4 * h;                                 // output row rho(r, h) for r < 4
                    if (c < OUTC && act) {
                        const float sg = 1.f / (1.f + expf(-d[r]));
                        o[ch][r] = sg * (tp.omax[c] - tp.omin[c]) + tp.omin[c];
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int64_t p = p0 + 32 * ch + m;
            if (p < n) {
                float* q = out + p * OUTC + 4 * h;
                q[0] = o[ch][0];
                q[1] = o[ch][1];
                if (h == 0) { q[2] = o[ch][2]; q[3] = o[ch][3]; }
            }
        }
    }
}

constexpr int PITCH = 33;

// MODE 0: everything in one kernel.  MODE 1 (ENC_ONLY): the gradient arrives at the encoding output (tcnn.Encoding used stand-alone, or
// the second half of the split backward).  MODE 2 (MLP_ONLY): the MLP half of the split backward -- weight gradients and d(encoding)
// written to `genc` [n][10]; no table scatter.  The split exists because the fused kernel needs 238 VGPRs (2 waves per SIMD) while its
// gather / segmented-scan / atomic half is latency-bound: as its own kernel that half runs at 60 VGPRs (7 waves per SIMD).
template <int MODE>
__global__ __launch_bounds__(256) void texmlp_bwd_kernel(GridCfg g, TexParams tp, const float* __restrict__ x, const float* __restrict__ mask,
                                                         const float* __restrict__ table, const float* __restrict__ w, int64_t n,
                                                         const float* __restrict__ g_out, float* __restrict__ d_table, float* __restrict__ d_w,
                                                         float* __restrict__ d_x, float* __restrict__ genc) {
    constexpr bool ENC_ONLY = MODE == 1;
    constexpr bool MLP_ONLY = MODE == 2;
    __shared__ float sA[256 * PITCH];
    __shared__ float sB[256 * PITCH];
    __shared__ int s_any[2];
    constexpr int SPITCH = 17;
    __shared__ float s_stage[4][64 * SPITCH];   // per wave: the sums of up to 64 runs (16 floats + the cell index each)
    int par = 0;
    const int tid = threadIdx.x;
    const float* sw = w;      // wave-uniform addresses -> scalar loads
    float acc2[4] = {0.f, 0.f, 0.f, 0.f};     // dW2[i = tid&31][j = (tid>>5)*4 + q]
    float acc3 = 0.f;                          // dW3[tid/32][tid%32], tid < 192
    float acc1[2] = {0.f, 0.f};                // dW1 flat index tid, tid + 256 (< 320)
    const int64_t ntile = (n + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t i = tile * 256 + tid;
        const bool active = (i < n) && !(mask && !(mask[i] > 0.f));
        // tiles with no covered pixel (most of the image) skip the whole body, including the workgroup-wide dW reductions
        par ^= 1;                       // two flags alternate so that resetting one never races with late readers of the other
        if (tid == 0) s_any[par] = 0;
        __syncthreads();
        if (active) s_any[par] = 1;
        __syncthreads();
        if (!s_any[par]) {
            if (!MLP_ONLY && d_x && i < n) { d_x[3 * i] = 0.f; d_x[3 * i + 1] = 0.f; d_x[3 * i + 2] = 0.f; }
            continue;
        }
        float xn[3] = {0.f, 0.f, 0.f}, enc[ENC], g_enc[ENC];
        bool inside[3] = {false, false, false};
#pragma unroll
        for (int c = 0; c < ENC; ++c) { enc[c] = 0.f; g_enc[c] = 0.f; }
        if (active) {
            normalise(tp, x + 3 * i, xn, inside);
            encode(g, table, xn, enc);
        }
        if (ENC_ONLY) {
            if (active) {
#pragma unroll
                for (int c = 0; c < ENC; ++c) g_enc[c] = g_out[i * ENC + c];
            }
        } else {
            // The MLP backward is interleaved with the three weight-gradient reductions so that at most two 32-vectors are live
            // per thread at any time (z1 + one of z2 / gz2 / gz1): no scratch spills.
            const float* w1 = sw;
            const float* w2 = sw + W1N;
            const float* w3 = sw + W1N + W2N;
            float z1[HID], z2[HID], go[OUTC];
#pragma unroll
            for (int ii = 0; ii < HID; ++ii) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < ENC; ++j) a = fmaf(w1[ii * ENC + j], enc[j], a);
                z1[ii] = a;
            }
#pragma unroll
            for (int ii = 0; ii < HID; ++ii) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < HID; ++j) a = fmaf(w2[ii * HID + j], fmaxf(z1[j], 0.f), a);
                z2[ii] = a;
            }
#pragma unroll
            for (int c = 0; c < OUTC; ++c) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < HID; ++j) a = fmaf(w3[c * HID + j], fmaxf(z2[j], 0.f), a);
                float sg = 1.f / (1.f + expf(-a));
                go[c] = active ? g_out[i * OUTC + c] * (tp.omax[c] - tp.omin[c]) * sg * (1.f - sg) : 0.f;
            }
            // ---- dW3 = go^T h2 ----
            if (d_w) {
#pragma unroll
                for (int c = 0; c < HID; ++c) sB[tid * PITCH + c] = active ? fmaxf(z2[c], 0.f) : 0.f;
#pragma unroll
                for (int c = 0; c < OUTC; ++c) sA[tid * PITCH + c] = go[c];
                __syncthreads();
                if (tid < W3N) {
                    const int oo = tid >> 5, jj = tid & 31;
                    for (int p = 0; p < 256; ++p) acc3 = fmaf(sA[p * PITCH + oo], sB[p * PITCH + jj], acc3);
                }
                __syncthreads();
            }
            // gz2 (in place in z2)
#pragma unroll
            for (int j = 0; j < HID; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < OUTC; ++c) a = fmaf(w3[c * HID + j], go[c], a);
                z2[j] = z2[j] > 0.f ? a : 0.f;
            }
            // ---- dW2 = gz2^T h1 ----
            if (d_w) {
#pragma unroll
                for (int c = 0; c < HID; ++c) { sA[tid * PITCH + c] = z2[c]; sB[tid * PITCH + c] = active ? fmaxf(z1[c], 0.f) : 0.f; }
                __syncthreads();
                {
                    const int ii = tid & 31, j0 = (tid >> 5) * 4;
                    for (int p = 0; p < 256; ++p) {
                        float a = sA[p * PITCH + ii];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc2[q] = fmaf(a, sB[p * PITCH + j0 + q], acc2[q]);
                    }
                }
                __syncthreads();
            }
            // gz1 (in place in z1)
#pragma unroll
            for (int j = 0; j < HID; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < HID; ++c) a = fmaf(w2[c * HID + j], z2[c], a);
                z1[j] = z1[j] > 0.f ? a : 0.f;
            }
            // ---- dW1 = gz1^T enc ----
            if (d_w) {
#pragma unroll
                for (int c = 0; c < HID; ++c) sA[tid * PITCH + c] = z1[c];
#pragma unroll
                for (int c = 0; c < ENC; ++c) sB[tid * PITCH + c] = enc[c];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    int k = tid + 256 * r;
                    if (k < W1N) {
                        const int ii = k / ENC, jj = k % ENC;
                        float a = acc1[r];
                        for (int p = 0; p < 256; ++p) a = fmaf(sA[p * PITCH + ii], sB[p * PITCH + jj], a);
                        acc1[r] = a;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < ENC; ++j) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < HID; ++c) a = fmaf(w1[c * ENC + j], z1[c], a);
                g_enc[j] = a * tp.in_grad_scale;      // register_full_backward_hook: grad_input * 128
            }
        }
        if (MLP_ONLY) {
            if (active) {
#pragma unroll
                for (int c = 0; c < ENC; ++c) genc[i * ENC + c] = g_enc[c];
            }
            continue;
        }
        // ---- scatter to the feature tables (wave-aggregated) and chain to the position ----------------------------------------
        // Neighbouring pixels fall into the same grid cell (a level-4 cell is ~10 px wide, a level-0 cell ~45 px), so the 64 lanes of
        // a wave would hammer a handful of addresses with fp32 atomics (measured: 12 ms per backward at 1024^2 x 4).  Runs of lanes
        // in the same cell are reduced with a segmented wave scan and added once per run.
        {
            const int lane_id = tid & 63;
            float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                float fr[3];
                int pg[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float pp = fmaf(xn[d], g.scale[l], 0.5f);
                    float fl = floorf(pp);
                    fr[d] = pp - fl;
                    pg[d] = (int)fl;
                }
                const int res = g.res[l];
                const float2* tab = (const float2*)table + g.offset[l];
                float* dtab = d_table ? d_table + 2 * (size_t)g.offset[l] : nullptr;
                const float ge0 = active ? g_enc[2 * l] : 0.f, ge1 = active ? g_enc[2 * l + 1] : 0.f;
                const int cell = active ? (pg[0] + pg[1] * res + pg[2] * res * res) : -1;
                float v[16];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float wgt = 1.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) wgt *= ((c >> d) & 1) ? fr[d] : (1.f - fr[d]);
                    v[2 * c] = wgt * ge0;
                    v[2 * c + 1] = wgt * ge1;
                    if (d_x && active) {
                        int idx = pg[0] + ((c >> 0) & 1) + (pg[1] + ((c >> 1) & 1)) * res + (pg[2] + ((c >> 2) & 1)) * res * res;
                        if (idx >= g.size[l]) idx -= g.size[l];
                        float2 tv = tab[idx];
                        float fv = tv.x * ge0 + tv.y * ge1;
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            float dw = 1.f;
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
                                int bit = (c >> e) & 1;
                                if (e == d) dw *= bit ? 1.f : -1.f;
                                else dw *= bit ? fr[e] : (1.f - fr[e]);
                            }
                            gx[d] = fmaf(dw * g.scale[l], fv, gx[d]);
                        }
                    }
                }
                if (dtab) {
                    // segmented inclusive scan over runs of equal cell (pixels of a row are consecutive, so equal cells form runs);
                    // the last lane of every run holds the run's 16 sums.  A cell that re-appears in a later run just gets two adds.
                    const int prev = __shfl_up(cell, 1);
                    const bool head = (lane_id == 0) || (cell != prev);
                    const unsigned long long heads = __ballot(head);
                    const int start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane_id))));
                    const bool tail = ((lane_id == 63) || ((heads >> (lane_id + 1)) & 1ull)) && cell >= 0;
                    // Float atomics execute at the memory side as 64-B requests, whatever the number of useful bytes in them (one lane per
                    // address: 4 useful bytes per request).  The 16 sums of a run go to 4 x 16 contiguous bytes (feature pair x corner pair
                    // in x), so the runs' sums are staged in a wave-private LDS tile and re-read with 16 lanes per run, 4 consecutive lanes
                    // on consecutive floats: 4 requests per run instead of 16.
                    const unsigned long long tails = __ballot(tail);
                    const int ntail = __popcll(tails);
                    const int rank = __popcll(tails & ((1ull << lane_id) - 1ull));
                    float* st = s_stage[tid >> 6];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float sv = d3h_seg_sum(v[k], lane_id, start);
                        if (tail) st[rank * SPITCH + k] = sv;
                    }
                    if (tail) st[rank * SPITCH + 16] = __int_as_float(cell);
                    D3H_WAVE_SYNC();
                    // Consecutive runs of a pixel row mostly sit in FACE-ADJACENT cells (the surface point moves by less than a cell per pixel):
                    // run r + 1 in cell c + d, d = +-1 / +-res / +-res^2, shares the 8 entries of that face with run r (same table addresses).
                    // They are folded into run r + 1 before the atomics (zeroed entries are skipped below): ~40 % fewer fabric atomics, which
                    // is what this launch is made of.  Even pairs, then odd pairs: a run is source and destination in different phases only.
                    if (tp.merge_faces) {
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            for (int r = 2 * (lane_id >> 3) + ph; r + 1 < ntail; r += 16) {
                                const int d = __float_as_int(st[(r + 1) * SPITCH + 16]) - __float_as_int(st[r * SPITCH + 16]);
                                const int ad = d < 0 ? -d : d;
                                const int bit = ad == 1 ? 2 : (ad == res ? 4 : (ad == res * res ? 8 : 0));      // the axis bit of q (q & 1 = feature)
                                if (bit) {
                                    const int j = lane_id & 7;
                                    // the 8 values of q with the axis bit clear, enumerated by j
                                    const int lo = j & (bit - 1), qz = lo | ((j & ~(bit - 1)) << 1);
                                    const int qa = qz | (d > 0 ? bit : 0), qb = qz | (d > 0 ? 0 : bit);
                                    st[(r + 1) * SPITCH + qb] += st[r * SPITCH + qa];
                                    st[r * SPITCH + qa] = 0.f;
                                }
                            }
                            D3H_WAVE_SYNC();
                        }
                    }
                    const int q = lane_id & 15;
                    const int corner = ((q >> 1) & 1) + ((q >> 2) & 1) * res + ((q >> 3) & 1) * res * res;
                    for (int r = lane_id >> 4; r < ntail; r += 4) {
                        const float sv = st[r * SPITCH + q];
                        int idx = __float_as_int(st[r * SPITCH + 16]) + corner;
                        if (idx >= g.size[l]) idx -= g.size[l];
                        if (sv != 0.f) atomicAdd(dtab + 2 * (size_t)idx + (q & 1), sv);
                    }
                    D3H_WAVE_SYNC();
                }
            }
            if (d_x && i < n) {
#pragma unroll
                for (int d = 0; d < 3; ++d) d_x[3 * i + d] = (active && inside[d]) ? gx[d] / (tp.b1[d] - tp.b0[d]) : 0.f;
            }
        }
    }
    if (!ENC_ONLY && d_w) {        // (MODE 0 and 2)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int k = tid + 256 * r;
            if (k < W1N && acc1[r] != 0.f) atomicAdd(&d_w[k], acc1[r]);
        }
        const int ii = tid & 31, j0 = (tid >> 5) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (acc2[q] != 0.f) atomicAdd(&d_w[W1N + ii * HID + j0 + q], acc2[q]);
        if (tid < W3N && acc3 != 0.f) atomicAdd(&d_w[W1N + W2N + tid], acc3);
    }
}

// ---- MLP half of the split backward, second version -----------------------------------------------------------------------------------
// weight gradients + d(encoding) -> genc [n][10].  One WAVE owns 64 consecutive pixels and never talks to the other waves of its workgroup:
//  * the three weight-gradient outer products dW3 = go^T h2, dW2 = gz2^T h1, dW1 = gz1^T enc are contractions over the 64 pixels = over the
//    lanes.  They run on the matrix pipe (v_mfma_f32_32x32x2_f32, A[m][k] = lane (m = l & 31, k = l >> 5)): both operands are transposed
//    through a wave-private LDS tile (pitch 33: conflict-free both ways), two pixels per MFMA, the 32 x 32 partial sums stay in 3 x 16
//    accumulator registers over all tiles of the wave.  The first version reduced them with scalar loops over a 256-pixel LDS tile -- five
//    ds_read per 4 FMA, ~20 000 LDS cycles and 12 workgroup barriers per tile;
//  * the backward matrix-vector products walk W2 / W3 / W1 ROW by row (the row is the wave-uniform operand, fetched with wide scalar
//    loads) and accumulate into the output vector, instead of column by column (one scalar load per weight).
// gz2 / gz1 overwrite z2 / z1 in place; the ReLU masks travel as two 32-bit words.
constexpr int WPITCH = 33;

__device__ __forceinline__ f32x16 outer_mfma(const float* __restrict__ TA, const float* __restrict__ TB, int lane, int a_rows, f32x16 acc) {
    const int m = lane & 31, h = lane >> 5;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        const float a = (m < a_rows) ? TA[(2 * s + h) * WPITCH + m] : 0.f;
        const float b = TB[(2 * s + h) * WPITCH + m];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(256) D3H_WAVES_PER_EU(2) void texmlp_bwd_mlp_kernel(GridCfg g, TexParams tp, const float* __restrict__ x, const float* __restrict__ mask,
                                                             const float* __restrict__ table, const float* __restrict__ w, int64_t n,
                                                             const float* __restrict__ g_out, float* __restrict__ d_w, float* __restrict__ genc) {
    __shared__ float sT[4][2][64 * WPITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* TA = sT[wave][0];
    float* TB = sT[wave][1];
    const float* w1 = w;
    const float* w2 = w + W1N;
    const float* w3 = w + W1N + W2N;
    f32x16 acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const int64_t nwt = (n + 63) / 64;
    for (int64_t wt = (int64_t)blockIdx.x * 4 + wave; wt < nwt; wt += (int64_t)gridDim.x * 4) {
        const int64_t i = wt * 64 + lane;
        const bool active = (i < n) && !(mask && !(mask[i] > 0.f));
        if (__ballot(active) == 0ull) continue;                  // wave-uniform: nothing covered in these 64 pixels
        float xn[3] = {0.f, 0.f, 0.f}, enc[ENC];
        bool inside[3] = {false, false, false};
#pragma unroll
        for (int c = 0; c < ENC; ++c) enc[c] = 0.f;
        if (active) {
            normalise(tp, x + 3 * i, xn, inside);
            encode(g, table, xn, enc);
        }
        float z1[HID], z2[HID], go[OUTC];
#pragma unroll
        for (int ii = 0; ii < HID; ++ii) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < ENC; ++j) a = fmaf(w1[ii * ENC + j], enc[j], a);
            z1[ii] = a;
        }
#pragma unroll
        for (int ii = 0; ii < HID; ++ii) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < HID; ++j) a = fmaf(w2[ii * HID + j], fmaxf(z1[j], 0.f), a);
            z2[ii] = a;
        }
#pragma unroll
        for (int c = 0; c < OUTC; ++c) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < HID; ++j) a = fmaf(w3[c * HID + j], fmaxf(z2[j], 0.f), a);
            const float sg = 1.f / (1.f + expf(-a));
            go[c] = active ? g_out[i * OUTC + c] * (tp.omax[c] - tp.omin[c]) * sg * (1.f - sg) : 0.f;
        }
        // ---- dW3 += go^T h2 ----
        if (d_w) {
#pragma unroll
            for (int c = 0; c < OUTC; ++c) TA[lane * WPITCH + c] = go[c];
#pragma unroll
            for (int c = 0; c < HID; ++c) TB[lane * WPITCH + c] = active ? fmaxf(z2[c], 0.f) : 0.f;
            D3H_WAVE_SYNC();
            acc3 = outer_mfma(TA, TB, lane, OUTC, acc3);
            D3H_WAVE_SYNC();
        }
        // gz2 = relu'(z2) * (W3^T go), in place in z2: row c of W3 is the wave-uniform operand
        {
            unsigned m2 = 0;
#pragma unroll
            for (int j = 0; j < HID; ++j) { m2 |= (z2[j] > 0.f ? 1u : 0u) << j; z2[j] = 0.f; }
#pragma unroll
            for (int c = 0; c < OUTC; ++c)
#pragma unroll
                for (int j = 0; j < HID; ++j) z2[j] = fmaf(w3[c * HID + j], go[c], z2[j]);
#pragma unroll
            for (int j = 0; j < HID; ++j) z2[j] = ((m2 >> j) & 1u) ? z2[j] : 0.f;
        }
        // ---- dW2 += gz2^T h1 ----
        unsigned m1 = 0;
#pragma unroll
        for (int j = 0; j < HID; ++j) m1 |= (z1[j] > 0.f ? 1u : 0u) << j;
        if (d_w) {
#pragma unroll
            for (int c = 0; c < HID; ++c) { TA[lane * WPITCH + c] = z2[c]; TB[lane * WPITCH + c] = active ? fmaxf(z1[c], 0.f) : 0.f; }
            D3H_WAVE_SYNC();
            acc2 = outer_mfma(TA, TB, lane, HID, acc2);
            D3H_WAVE_SYNC();
        }
        // gz1 = relu'(z1) * (W2^T gz2), in place in z1
        {
#pragma unroll
            for (int j = 0; j < HID; ++j) z1[j] = 0.f;
#pragma unroll
            for (int c = 0; c < HID; ++c)
#pragma unroll
                for (int j = 0; j < HID; ++j) z1[j] = fmaf(w2[c * HID + j], z2[c], z1[j]);
#pragma unroll
            for (int j = 0; j < HID; ++j) z1[j] = ((m1 >> j) & 1u) ? z1[j] : 0.f;
        }
        // ---- dW1 += gz1^T enc ----
        if (d_w) {
#pragma unroll
            for (int c = 0; c < HID; ++c) TA[lane * WPITCH + c] = z1[c];
#pragma unroll
            for (int c = 0; c < HID; ++c) TB[lane * WPITCH + c] = (c < ENC) ? enc[c < ENC ? c : 0] : 0.f;
            D3H_WAVE_SYNC();
            acc1 = outer_mfma(TA, TB, lane, HID, acc1);
            D3H_WAVE_SYNC();
        }
        // d(encoding) = W1^T gz1 * in_grad_scale (register_full_backward_hook: grad_input * 128)
        if (active) {
            float ge[ENC];
#pragma unroll
            for (int j = 0; j < ENC; ++j) ge[j] = 0.f;
#pragma unroll
            for (int c = 0; c < HID; ++c)
#pragma unroll
                for (int j = 0; j < ENC; ++j) ge[j] = fmaf(w1[c * ENC + j], z1[c], ge[j]);
#pragma unroll
            for (int c = 0; c < ENC; ++c) genc[i * ENC + c] = ge[c] * tp.in_grad_scale;
        }
    }
    if (d_w) {
        // the four waves' partial sums meet in LDS, then one atomic per weight and workgroup.  D layout: col = lane & 31, row = (r & 3) +
        // 8 (r >> 2) + 4 (lane >> 5);  dW2[i][j] = D2[row i][col j], dW3[o][j] = D3[row o < 6][col j], dW1[i][e] = D1[row i][col e < 10]
        __syncthreads();
        float* red = &sT[0][0][0];                      // 4 x 2 x 64 x 33 floats >= 4 waves x 3 x 1024
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            red[(wave * 3 + 0) * 1024 + row * 32 + col] = acc1[r];
            red[(wave * 3 + 1) * 1024 + row * 32 + col] = acc2[r];
            red[(wave * 3 + 2) * 1024 + row * 32 + col] = acc3[r];
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * 1024; k += 256) {
            const int which = k >> 10, rc = k & 1023, row = rc >> 5, col = rc & 31;
            const float v = red[k] + red[3 * 1024 + k] + red[6 * 1024 + k] + red[9 * 1024 + k];
            if (v == 0.f) continue;
            if (which == 0) { if (col < ENC) atomicAdd(&d_w[row * ENC + col], v); }
            else if (which == 1) atomicAdd(&d_w[W1N + row * HID + col], v);
            else { if (row < OUTC) atomicAdd(&d_w[W1N + W2N + row * HID + col], v); }
        }
    }
}

// ---- MLP half of the split backward, third version (round 6): every product on the fp16 matrix pipe ------------------------------------------
// texmlp_bwd_mlp_kernel above is VALU-bound: per pixel 3 072 per-lane FMAs (forward recompute + two backward mat-vecs) next to 96
// v_mfma_f32_32x32x2_f32 (64 cycles each) for the outer products.  Here a wave handles its 64 pixels as two CHAINS of 32 (the columns of a
// 32 x 32 MFMA tile) and every mat-vec is a register-chained v_mfma_f32_32x32x16_f16 with the two-plane fp16 split (d3h_h2.h: three products,
// fp32 accuracy):
//   forward      Z1 = W1 enc,  Z2 = W2 relu(Z1),  Z3 = W3 relu(Z2)                    1 + 2 + 2 k-steps of 16
//   backward     GZ2 = relu'(Z2) W3^T go,  GZ1 = relu'(Z1) W2^T GZ2,  GE = W1^T GZ1    1 + 2 + 2 k-steps
// The D layout of a layer -- lane (n, h) holds rows rho(r, h) of pixel n -- IS the B layout of the next layer when k-step s pairs its eight
// k-values of lane half h with rows rho(8 s + j, h): the weights are loaded in that order once per wave (A planes: 9 k-steps x 8 registers) and
// activations never leave their registers; only the 10 encoding features and the 6 output gradients cross the lane halves (shuffles).
//   outer products   dW3 += go^T relu(Z2),  dW2 += GZ2^T relu(Z1),  dW1 += GZ1^T enc   contraction over the chain's 32 pixels = 2 k-steps each:
// both operands go through a wave-private LDS image [32 rows][32 pixels] (pitch 36) to get eight consecutive PIXELS of a row into a lane.
// Gradient scale: go = g_out * range * sigma' is ~1e-7 in training (a mean over 4 10^6 pixels) -- below the fp16 normal range; every chain
// multiplies its go by a power of two S (from the chain's largest |go|, wave reduction) and its results by 1 / S: exact, and a pixel 10^-4
// below the chain's largest keeps its full 22 bits.  A chain whose go is all zero is skipped.
constexpr int HP = 36;      // pitch of the transposition images (floats): 16-byte aligned rows, 8-float reads of 32 rows hit distinct bank quads pairwise

__device__ __forceinline__ void tex_h2_image_put(float* T, const f32x16& v, int n, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[rho(r, h) * HP + n] = v[r];
}
// eight consecutive pixels 16 s + 8 kh .. + 7 of row `row` -> operand planes
__device__ __forceinline__ D3hH2Frag tex_h2_image_frag(const float* T, int row, int s, int kh) {
    const float* p = T + row * HP + 16 * s + 8 * kh;
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return d3h_h2_frag(v);
}

// THE CO-RESIDENCY RULE (csrc/sdf_mlp_x3.h): a kernel that issues 16-bit MFMAs back to back must not share a SIMD with a wave of another kernel
// (packed-f32 VALU results of the foreign wave go wrong in lanes 48..63): 512-thread workgroups with all 256 VGPRs claimed -- two waves per
// SIMD own the register file -- and a workgroup barrier after the last MFMA (the flush below).  tests/test_mfma_claim.py checks both in the ISA.
__global__ __launch_bounds__(512) void texmlp_bwd_mlp_h2_kernel(GridCfg g, TexParams tp, const float* __restrict__ x, const float* __restrict__ mask,
                                                                const float* __restrict__ table, const float* __restrict__ w, int64_t n,
                                                                const float* __restrict__ g_out, float* __restrict__ d_w, float* __restrict__ genc) {
    __shared__ __attribute__((aligned(16))) float smem[8 * 2 * 32 * HP];          // the transposition images (8 waves x 2 x 32 x HP) and, at the end, the flush
    static_assert(4 * 3 * 1024 <= 8 * 2 * 32 * HP, "the flush buffer lives in the images");
#ifndef D3H_EMULATED
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    float* TA = smem + (wave * 2 + 0) * 32 * HP;
    float* TB = smem + (wave * 2 + 1) * 32 * HP;
    const float* w1 = w;
    const float* w2 = w + W1N;
    const float* w3 = w + W1N + W2N;
    // ---- the weights as A operands, once per workgroup, in LDS (9 k-steps x 2 planes x 64 lanes x 16 B = 18 KB; in registers they cost 72 VGPRs and
    // the kernel spilled 63): lane (m, h) holds eight k-values of row m per k-step.  Slots: 0 W1 | 1 W3^T | 2, 3 W2 | 4, 5 W3 | 6, 7 W2^T | 8, 9 W1^T
    __shared__ __attribute__((aligned(16))) d3h_u32x4 sA[10][2][64];
    if (wave == 0) {
        auto put = [&](int slot, const float (&v)[8]) {
            const D3hH2Frag f = d3h_h2_frag(v);
            sA[slot][0][lane] = f.h;
            sA[slot][1][lane] = f.m;
        };
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (8 * h + j < ENC) ? w1[m * ENC + 8 * h + j] : 0.f;                      // Z1 = W1 enc: k = encoding feature 8 h + j
        put(0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (8 * h + j < OUTC) ? w3[(8 * h + j) * HID + m] : 0.f;                   // GZ2 = W3^T go: k = output channel 8 h + j
        put(1, v);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w2[m * HID + rho(8 * s + j, h)];                                    // Z2 = W2 relu(Z1): k = hidden unit rho(8 s + j, h)
            put(2 + s, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (m < OUTC) ? w3[m * HID + rho(8 * s + j, h)] : 0.f;                 // Z3 = W3 relu(Z2)
            put(4 + s, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w2[rho(8 * s + j, h) * HID + m];                                    // GZ1 = W2^T GZ2
            put(6 + s, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (m < ENC) ? w1[rho(8 * s + j, h) * ENC + m] : 0.f;                  // GE = W1^T GZ1
            put(8 + s, v);
        }
    }
    __syncthreads();
    auto Aw = [&](int slot) { D3hH2Frag f; f.h = sA[slot][0][lane]; f.m = sA[slot][1][lane]; return f; };
    f32x16 acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const f32x16 zero16 = acc1;
    const int64_t nwt = (n + 63) / 64;
    for (int64_t wt = (int64_t)blockIdx.x * 8 + wave; wt < nwt; wt += (int64_t)gridDim.x * 8) {
        const int64_t p0 = wt * 64;
        const int64_t i = p0 + lane;
        const bool active = (i < n) && !(mask && !(mask[i] > 0.f));
        const unsigned long long am = __ballot(active);
        if (am == 0ull) continue;                                // wave-uniform: nothing covered in these 64 pixels
        float xn[3] = {0.f, 0.f, 0.f}, enc[ENC];
        bool inside[3] = {false, false, false};
#pragma unroll
        for (int c = 0; c < ENC; ++c) enc[c] = 0.f;
        if (active) {
            normalise(tp, x + 3 * i, xn, inside);
            encode(g, table, xn, enc);
        }
        float eo[ENC];                                           // the encoding of the pixel in the other lane half
#pragma unroll
        for (int c = 0; c < ENC; ++c) eo[c] = __shfl_xor(enc[c], 32);
#pragma unroll 1
        for (int ch = 0; ch < 2; ++ch) {
            if (((am >> (32 * ch)) & 0xffffffffull) == 0ull) continue;      // (wave-uniform) nothing covered in this chain
            const bool own = (h == ch);                          // this lane's own pixel belongs to the chain (lane half ch holds the chain's pixels)
            // B of layer 1: lane (n, kh = h) holds encoding features 8 h + j of the chain's pixel n
            float b8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e_own = (j < ENC) ? enc[j < ENC ? j : 0] : 0.f, e_oth = (j < ENC) ? eo[j < ENC ? j : 0] : 0.f;         // features 0 .. 7
                const float f_own = (8 + j < ENC) ? enc[8 + j < ENC ? 8 + j : 0] : 0.f, f_oth = (8 + j < ENC) ? eo[8 + j < ENC ? 8 + j : 0] : 0.f;   // features 8, 9
                b8[j] = (h == 0) ? (own ? e_own : e_oth) : (own ? f_own : f_oth);
            }
            const D3hH2Frag Benc = d3h_h2_frag(b8);
            f32x16 hi = zero16, lo = zero16;
            d3h_h2_mac32(hi, lo, Aw(0), Benc);
            const f32x16 Z1 = d3h_h2_fold(hi, lo);
            hi = zero16; lo = zero16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = fmaxf(Z1[8 * s + j], 0.f);
                d3h_h2_mac32(hi, lo, Aw(2 + s), d3h_h2_frag(b8));
            }
            const f32x16 Z2 = d3h_h2_fold(hi, lo);
            hi = zero16; lo = zero16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = fmaxf(Z2[8 * s + j], 0.f);
                d3h_h2_mac32(hi, lo, Aw(4 + s), d3h_h2_frag(b8));
            }
            const f32x16 Z3 = d3h_h2_fold(hi, lo);
            // go for the rows this lane holds: channel c = r + 4 h (r < 4) of pixel p0 + 32 ch + m
            const int64_t pp = p0 + 32 * ch + m;
            const bool pact = (am >> (32 * ch + m)) & 1ull;
            float go[4] = {0.f, 0.f, 0.f, 0.f};
            float gmax = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = r + 4 * h;
                if (c < OUTC && pact) {
                    const float sg = 1.f / (1.f + expf(-Z3[r]));
                    go[r] = g_out[pp * OUTC + c] * (tp.omax[c] - tp.omin[c]) * sg * (1.f - sg);
                    const float a = fabsf(go[r]);
                    gmax = (a < 3.0e38f) ? fmaxf(gmax, a) : gmax;
                }
            }
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, k));
            if (!(gmax > 0.f)) {                                  // (wave-uniform) no gradient reaches this chain: d(encoding) = 0 for its covered pixels
                if (pact) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rho(r, h);
                        if (row < ENC) genc[pp * ENC + row] = 0.f;
                    }
                }
                continue;
            }
            const float S = d3h_h2_pow2_scale(gmax), Si = 1.0f / S;
#pragma unroll
            for (int r = 0; r < 4; ++r) go[r] *= S;
            // B of the W3^T step: lane (n, 0) holds channels 0 .. 7 (0 .. 3 its own rows, 4, 5 from lane (n, 1)); lane (n, 1): channels 8 .. 15 = 0
            const float g4 = __shfl_xor(go[0], 32), g5 = __shfl_xor(go[1], 32);
            {
                const float v[8] = {h == 0 ? go[0] : 0.f, h == 0 ? go[1] : 0.f, h == 0 ? go[2] : 0.f, h == 0 ? go[3] : 0.f, h == 0 ? g4 : 0.f, h == 0 ? g5 : 0.f, 0.f, 0.f};
                hi = zero16; lo = zero16;
                d3h_h2_mac32(hi, lo, Aw(1), d3h_h2_frag(v));
            }
            f32x16 GZ2 = d3h_h2_fold(hi, lo);
#pragma unroll
            for (int r = 0; r < 16; ++r) GZ2[r] = (Z2[r] > 0.f) ? GZ2[r] : 0.f;
            hi = zero16; lo = zero16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = GZ2[8 * s + j];
                d3h_h2_mac32(hi, lo, Aw(6 + s), d3h_h2_frag(b8));
            }
            f32x16 GZ1 = d3h_h2_fold(hi, lo);
#pragma unroll
            for (int r = 0; r < 16; ++r) GZ1[r] = (Z1[r] > 0.f) ? GZ1[r] : 0.f;
            hi = zero16; lo = zero16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = GZ1[8 * s + j];
                d3h_h2_mac32(hi, lo, Aw(8 + s), d3h_h2_frag(b8));
            }
            const f32x16 GE = d3h_h2_fold(hi, lo);
            // d(encoding) = W1^T gz1 * in_grad_scale (register_full_backward_hook: grad_input * 128); rows rho(r, h) < 10 of pixel pp
            if (pact) {
                const float sc = tp.in_grad_scale * Si;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rho(r, h);
                    if (row < ENC) genc[pp * ENC + row] = GE[r] * sc;
                }
            }
            if (d_w) {
                // ---- the three outer products over the chain's 32 pixels: A side = (scaled) gradients, B side = activations ----
                auto outer = [&](f32x16& acc) {
                    f32x16 th = zero16, tl = zero16;
#pragma unroll
                    for (int s = 0; s < 2; ++s) d3h_h2_mac32(th, tl, tex_h2_image_frag(TA, m, s, h), tex_h2_image_frag(TB, m, s, h));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = fmaf(fmaf(tl[r], D3H_H2_INV_SCALE, th[r]), Si, acc[r]);
                };
                f32x16 gov = zero16, t16;
                // go as a 32-row image: rows 0 .. 3 from lane half 0, rows 4, 5 from lane half 1 (its registers 0, 1), everything else zero
#pragma unroll
                for (int r = 0; r < 4; ++r) gov[r] = (r + 4 * h < OUTC) ? go[r] : 0.f;
                tex_h2_image_put(TA, gov, m, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = pact ? fmaxf(Z2[r], 0.f) : 0.f;
                tex_h2_image_put(TB, t16, m, h);
                D3H_WAVE_SYNC();
                outer(acc3);                                      // dW3[o][j] += sum_p go[p][o] relu(Z2)[p][j]
                D3H_WAVE_SYNC();
                tex_h2_image_put(TA, GZ2, m, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = pact ? fmaxf(Z1[r], 0.f) : 0.f;
                tex_h2_image_put(TB, t16, m, h);
                D3H_WAVE_SYNC();
                outer(acc2);                                      // dW2[i][j] += sum_p gz2[p][i] relu(Z1)[p][j]
                D3H_WAVE_SYNC();
                tex_h2_image_put(TA, GZ1, m, h);
                // the encoding as a 32-row image: rows 0 .. 9; lane (n, h) writes rows rho(r, h) of its chain pixel n
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rho(r, h);
                    float e = 0.f;
#pragma unroll
                    for (int c = 0; c < ENC; ++c) e = (row == c) ? (own ? enc[c] : eo[c]) : e;
                    t16[r] = e;
                }
                tex_h2_image_put(TB, t16, m, h);
                D3H_WAVE_SYNC();
                outer(acc1);                                      // dW1[i][e] += sum_p gz1[p][i] enc[p][e]
                D3H_WAVE_SYNC();
            }
        }
    }
    __syncthreads();                 // (also THE barrier after the last MFMA of every wave)
    if (d_w) {
        // the eight waves' partial sums meet in LDS -- waves 4 .. 7 hand theirs to waves 0 .. 3 first (same register layout), then the four sums
        // are added and flushed with one atomic per weight and workgroup (as texmlp_bwd_mlp_kernel)
        float* red = smem;
        const int wslot = wave & 3;
        if (wave >= 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = rho(r, h) * 32 + m;
                red[(wslot * 3 + 0) * 1024 + o] = acc1[r];
                red[(wslot * 3 + 1) * 1024 + o] = acc2[r];
                red[(wslot * 3 + 2) * 1024 + o] = acc3[r];
            }
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = rho(r, h) * 32 + m;
                acc1[r] += red[(wslot * 3 + 0) * 1024 + o];
                acc2[r] += red[(wslot * 3 + 1) * 1024 + o];
                acc3[r] += red[(wslot * 3 + 2) * 1024 + o];
            }
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = rho(r, h) * 32 + m;
                red[(wslot * 3 + 0) * 1024 + o] = acc1[r];
                red[(wslot * 3 + 1) * 1024 + o] = acc2[r];
                red[(wslot * 3 + 2) * 1024 + o] = acc3[r];
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * 1024; k += 512) {
            const int which = k >> 10, rc = k & 1023, row = rc >> 5, col = rc & 31;
            const float v = red[k] + red[3 * 1024 + k] + red[6 * 1024 + k] + red[9 * 1024 + k];
            if (v == 0.f) continue;
            if (which == 0) { if (col < ENC) atomicAdd(&d_w[row * ENC + col], v); }
            else if (which == 1) atomicAdd(&d_w[W1N + row * HID + col], v);
            else { if (row < OUTC) atomicAdd(&d_w[W1N + W2N + row * HID + col], v); }
        }
    }
}

GridCfg make_cfg(double per_level_scale, int base_res) {
    GridCfg g;
    int off = 0;
    for (int l = 0; l < NL; ++l) {
        float scale = exp2f((float)l * log2f((float)per_level_scale)) * (float)base_res - 1.0f;
        int res = (int)ceilf(scale) + 1;
        g.scale[l] = scale;
        g.res[l] = res;
        g.offset[l] = off;
        long long cnt = (long long)res * res * res;
        cnt = (cnt + 7) / 8 * 8;
        g.size[l] = (int)cnt;
        off += (int)cnt;
    }
    return g;
}

TexParams make_tp(const float* bbox, const float* omin, const float* omax, float in_grad_scale) {
    TexParams tp;
    for (int d = 0; d < 3; ++d) { tp.b0[d] = bbox[d]; tp.b1[d] = bbox[3 + d]; }
    for (int c = 0; c < OUTC; ++c) { tp.omin[c] = omin ? omin[c] : 0.f; tp.omax[c] = omax ? omax[c] : 1.f; }
    tp.in_grad_scale = in_grad_scale;
    static const int merge = [] { const char* e = getenv("D3H_TEX_MERGE_FACES"); return (e && e[0] == '0') ? 0 : 1; }();
    tp.merge_faces = merge;
    return tp;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI   (bbox/omin/omax are HOST arrays: 6, 6, 6 floats)
// ------------------------------------------------------------------------------------------------
// number of floats of the grid parameter vector (tcnn.Encoding.params)
extern "C" int64_t d3h_hashgrid_param_floats(double per_level_scale, int base_res) {
    GridCfg g = make_cfg(per_level_scale, base_res);
    long long last = (long long)g.res[NL - 1] * g.res[NL - 1] * g.res[NL - 1];
    last = (last + 7) / 8 * 8;
    return ((int64_t)g.offset[NL - 1] + last) * NF;
}

// w = concat(W1[32][10], W2[32][32], W3[6][32]); out [n][6] and/or enc_out [n][10]; mask [n] optional (<= 0: skipped)
extern "C" int d3h_texmlp_fwd(const float* x, const float* mask, const float* table, const float* w, int64_t n, double per_level_scale,
                              int base_res, const float* bbox, const float* omin, const float* omax, float* out, float* enc_out, void* stream) {
    if (n < 0 || !bbox) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    GridCfg g = make_cfg(per_level_scale, base_res);
    TexParams tp = make_tp(bbox, omin, omax, 1.f);
    const int kt = d3h_ktime_begin(D3H_KT_TEX_FWD, n, (hipStream_t)stream);
    // one 256-pixel tile per workgroup: the covered pixels sit in the middle of every frame, so a grid-stride loop over 2048 workgroups gave
    // a quarter of them all eight of their tiles covered and the rest none (162 us per 4 x 1024^2 call); with one tile each the dispatcher
    // balances, background tiles retire at once
    const int64_t ntile = (n + 255) / 256;
    const unsigned grid = (unsigned)(ntile < (1 << 20) ? ntile : (1 << 20));
    if (out && w && !enc_out)      // persistent waves keep the weights: 2039 (prime) workgroups, so that the covered tiles spread over all of them
        hipLaunchKernelGGL(texmlp_fwd_mfma_kernel, dim3(grid < 2039u ? grid : 2039u), dim3(256), 0, (hipStream_t)stream, g, tp, x, mask, table, w, n, out);
    else
        hipLaunchKernelGGL(texmlp_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, tp, x, mask, table, w, n, out, enc_out);
    d3h_ktime_end(kt, (hipStream_t)stream);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// g_out [n][6] (or [n][10] when enc_only); d_table / d_w accumulated (caller zero-fills, may be NULL); d_x [n][3] overwritten (may be NULL)
extern "C" int d3h_texmlp_bwd(const float* x, const float* mask, const float* table, const float* w, int64_t n, double per_level_scale,
                              int base_res, const float* bbox, const float* omin, const float* omax, float in_grad_scale, int enc_only,
                              const float* g_out, float* d_table, float* d_w, float* d_x, float* genc_scratch, void* stream) {
    if (n < 0 || !bbox) return D3H_ERR_ARG;
    if (n == 0) return D3H_OK;
    GridCfg g = make_cfg(per_level_scale, base_res);
    TexParams tp = make_tp(bbox, omin, omax, in_grad_scale);
    int64_t ntile = (n + 255) / 256;
    // 1021 (prime) workgroups: an image row is W/256 tiles, so with a power-of-two grid every workgroup would keep visiting the same
    // image columns and the ones over the (centred) body would get all the covered tiles
    int grid = (int)(ntile < 1021 ? ntile : 1021);
    hipStream_t s = (hipStream_t)stream;
    float* nof = nullptr;
    if (enc_only) {
        const int kt = d3h_ktime_begin(D3H_KT_TEX_BWD_ENC, n, s);
        hipLaunchKernelGGL((texmlp_bwd_kernel<1>), dim3(grid), dim3(256), 0, s, g, tp, x, mask, table, w, n, g_out, d_table, d_w, d_x, nof);
        d3h_ktime_end(kt, s);
    } else if (genc_scratch) {
        // split backward: genc_scratch [n][10] carries d(encoding) (already scaled by in_grad_scale) between the two halves
        int grid2 = (int)(ntile < 4093 ? ntile : 4093);
        // MLP half: wave-granular (64-pixel) tiles, 509 (prime) workgroups of 4 waves -- two per CU, one flush of the weight gradients each
        int gridm = (int)(ntile < 509 ? ntile : 509);
        const int ktm = d3h_ktime_begin(D3H_KT_TEX_BWD_MLP, n, s);
        static int tex_h2 = -1;          // D3H_TEX_H2=0: the round-5 kernel (VALU mat-vecs + exact-f32 outer products), A/B
        if (tex_h2 < 0) { const char* e = getenv("D3H_TEX_H2"); tex_h2 = (e && e[0] == '0') ? 0 : 1; }
        // (512-thread workgroups, one per CU -- 251: prime, see above -- each wave a 64-pixel tile at a time)
        const int64_t nwt8 = (n + 511) / 512;
        if (tex_h2) hipLaunchKernelGGL(texmlp_bwd_mlp_h2_kernel, dim3((unsigned)(nwt8 < 251 ? nwt8 : 251)), dim3(512), 0, s, g, tp, x, mask, table, w, n, g_out, d_w, genc_scratch);
        else hipLaunchKernelGGL(texmlp_bwd_mlp_kernel, dim3(gridm), dim3(256), 0, s, g, tp, x, mask, table, w, n, g_out, d_w, genc_scratch);
        d3h_ktime_end(ktm, s);
        if (d_table || d_x) {
            const int kte = d3h_ktime_begin(D3H_KT_TEX_BWD_ENC, n, s);
            hipLaunchKernelGGL((texmlp_bwd_kernel<1>), dim3(grid2), dim3(256), 0, s, g, tp, x, mask, table, w, n, (const float*)genc_scratch, d_table, nof, d_x,
                               nof);
            d3h_ktime_end(kte, s);
        }
    } else {
        hipLaunchKernelGGL((texmlp_bwd_kernel<0>), dim3(grid), dim3(256), 0, s, g, tp, x, mask, table, w, n, g_out, d_table, d_w, d_x, nof);
    }
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
