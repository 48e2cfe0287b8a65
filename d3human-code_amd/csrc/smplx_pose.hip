// smplx_pose.hip -- SMPL-X pose -> per-joint skinning transforms A[B][55][4][4], forward and backward, one launch each (gfx950).
//
// Replaces the joint part of deform/smplx_exavatar/lbs.py: batch_rodrigues (:311-347, incl. the +1e-8 inside the norm),
// batch_rigid_transform (:361-413: relative joints, the 54 sequential 4x4 products down the kinematic tree, the rest-pose removal
// A = G - pad(G [J; 0])) as called by lbs() (:216-247) from SMPLX.forward (body_models.py:1225-1257).  The reference (and round 1 of this
// build, level-batched) issues dozens of tiny library kernels per call; the whole computation is 55 3x4 products per frame.
// One 64-lane workgroup per frame, lane = joint; the transforms live in LDS.  Latency-bound by construction (a few KB of data).
//   forward : T_j = [R(fp_j) | J_j - J_parent];  G_j = T_root ... T_j (each lane walks up its own ancestor chain);
//             A_j = [G_j.R | G_j.t - G_j.R J_j]
//   backward: dG_j = [dA_j.R - dA_j.t J_j^T | dA_j.t], dJ_j -= G_j.R^T dA_j.t; then level by level from the leaves: the children add
//             dG_j.R R_j^T + dG_j.t r_j^T and dG_j.t into their parent's dG (LDS atomics), and dT_j = G_parent^T dG_j yields d(rel joint)
//             and, through the Rodrigues derivative, d(fp_j).
#include "d3h_common.h"

namespace {

constexpr int NJ_MAX = 64;

struct M34 { float r[9]; float t[3]; };

__device__ __forceinline__ M34 mul34(const M34& a, const M34& b) {      // a * b for [R|t; 0 0 0 1] matrices
    M34 o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o.r[3 * i + j] = a.r[3 * i] * b.r[j] + a.r[3 * i + 1] * b.r[3 + j] + a.r[3 * i + 2] * b.r[6 + j];
        o.t[i] = a.r[3 * i] * b.t[0] + a.r[3 * i + 1] * b.t[1] + a.r[3 * i + 2] * b.t[2] + a.t[i];
    }
    return o;
}

// R = I + sin(th) K + (1 - cos(th)) K^2, K = skew(rv / th), th = |rv + 1e-8|  (lbs.py:311-347)
__device__ __forceinline__ void rodrigues(const float* rv, float* R, float* th_out) {
    const float a0 = rv[0] + 1e-8f, a1 = rv[1] + 1e-8f, a2 = rv[2] + 1e-8f;
    const float th = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float x = rv[0] / th, y = rv[1] / th, z = rv[2] / th;
    const float s = sinf(th), c1 = 1.f - cosf(th);
    // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  K^2 = d d^T - |d|^2 I
    const float dd = x * x + y * y + z * z;
    R[0] = 1.f + c1 * (x * x - dd); R[1] = -s * z + c1 * x * y;     R[2] = s * y + c1 * x * z;
    R[3] = s * z + c1 * x * y;      R[4] = 1.f + c1 * (y * y - dd); R[5] = -s * x + c1 * y * z;
    R[6] = -s * y + c1 * x * z;     R[7] = s * x + c1 * y * z;      R[8] = 1.f + c1 * (z * z - dd);
    *th_out = th;
}

__global__ __launch_bounds__(64) void smplx_pose_fwd_kernel(const float* __restrict__ fp, const float* __restrict__ J, int j_bstride,
                                                            const int* __restrict__ parents, int nj, float* __restrict__ A, float* __restrict__ G) {
    __shared__ float sT[NJ_MAX][12];
    __shared__ int sP[NJ_MAX];
    const int b = blockIdx.x, j = threadIdx.x;
    const float* Jb = J + (size_t)b * j_bstride;
    if (j < nj) {
        const int p = parents[j];
        sP[j] = (j == 0) ? -1 : p;
        float th;
        rodrigues(fp + ((size_t)b * nj + j) * 3, sT[j], &th);
#pragma unroll
        for (int k = 0; k < 3; ++k) sT[j][9 + k] = Jb[3 * j + k] - ((j > 0) ? Jb[3 * p + k] : 0.f);
    }
    __syncthreads();
    if (j >= nj) return;
    M34 g;
#pragma unroll
    for (int k = 0; k < 9; ++k) g.r[k] = sT[j][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) g.t[k] = sT[j][9 + k];
    for (int a = sP[j]; a >= 0; a = sP[a]) {
        M34 ta;
#pragma unroll
        for (int k = 0; k < 9; ++k) ta.r[k] = sT[a][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) ta.t[k] = sT[a][9 + k];
        g = mul34(ta, g);
    }
    float* Aj = A + ((size_t)b * nj + j) * 16;
    float* Gj = G + ((size_t)b * nj + j) * 12;
    const float jx = Jb[3 * j], jy = Jb[3 * j + 1], jz = Jb[3 * j + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Aj[4 * i] = g.r[3 * i]; Aj[4 * i + 1] = g.r[3 * i + 1]; Aj[4 * i + 2] = g.r[3 * i + 2];
        Aj[4 * i + 3] = g.t[i] - (g.r[3 * i] * jx + g.r[3 * i + 1] * jy + g.r[3 * i + 2] * jz);
    }
    Aj[12] = 0.f; Aj[13] = 0.f; Aj[14] = 0.f; Aj[15] = 1.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) Gj[k] = g.r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Gj[9 + k] = g.t[k];
}

__global__ __launch_bounds__(64) void smplx_pose_bwd_kernel(const float* __restrict__ fp, const float* __restrict__ J, int j_bstride,
                                                            const int* __restrict__ parents, int nj, const float* __restrict__ G,
                                                            const float* __restrict__ dA, float* __restrict__ d_fp, float* __restrict__ d_J) {
    __shared__ float sdG[NJ_MAX][12];
    __shared__ float sdJ[NJ_MAX][3];
    __shared__ int sP[NJ_MAX];
    __shared__ int sDepth[NJ_MAX];
    __shared__ int sMax;
    const int b = blockIdx.x, j = threadIdx.x;
    const float* Jb = J + (size_t)b * j_bstride;
    if (j == 0) sMax = 0;
    if (j < nj) sP[j] = (j == 0) ? -1 : parents[j];
    __syncthreads();
    float R[9], r[3], th = 1.f;
    int depth = 0;
    if (j < nj) {
        for (int a = sP[j]; a >= 0; a = sP[a]) ++depth;
        sDepth[j] = depth;
        atomicMax(&sMax, depth);
        rodrigues(fp + ((size_t)b * nj + j) * 3, R, &th);
        const int p = sP[j];
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = Jb[3 * j + k] - ((j > 0) ? Jb[3 * p + k] : 0.f);
        // own part: A_j = [G.R | G.t - G.R J_j]
        const float* dAj = dA + ((size_t)b * nj + j) * 16;
        const float* Gj = G + ((size_t)b * nj + j) * 12;
        const float jx = Jb[3 * j], jy = Jb[3 * j + 1], jz = Jb[3 * j + 2];
        float dj[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float dt = dAj[4 * i + 3];
            sdG[j][3 * i] = dAj[4 * i] - dt * jx;
            sdG[j][3 * i + 1] = dAj[4 * i + 1] - dt * jy;
            sdG[j][3 * i + 2] = dAj[4 * i + 2] - dt * jz;
            sdG[j][9 + i] = dt;
            dj[0] -= Gj[3 * i] * dt; dj[1] -= Gj[3 * i + 1] * dt; dj[2] -= Gj[3 * i + 2] * dt;
        }
        sdJ[j][0] = dj[0]; sdJ[j][1] = dj[1]; sdJ[j][2] = dj[2];
    }
    __syncthreads();
    const int maxd = sMax;
    float dRT[9], dr[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) dRT[k] = 0.f;
    dr[0] = dr[1] = dr[2] = 0.f;
    for (int level = maxd; level >= 0; --level) {
        if (j < nj && depth == level) {
            float g[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) g[k] = sdG[j][k];               // complete: every deeper level has been folded in
            if (level == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) dRT[k] = g[k];
                dr[0] = g[9]; dr[1] = g[10]; dr[2] = g[11];
            } else {
                const int p = sP[j];
                const float* Gp = G + ((size_t)b * nj + p) * 12;
                // dT_j = Gp.R^T dG_j
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) dRT[3 * i + k] = Gp[i] * g[k] + Gp[3 + i] * g[3 + k] + Gp[6 + i] * g[6 + k];
                    dr[i] = Gp[i] * g[9] + Gp[3 + i] * g[10] + Gp[6 + i] * g[11];
                }
                // parent: dGp.R += dG_j.R R_j^T + dG_j.t r_j^T;  dGp.t += dG_j.t
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        atomicAdd(&sdG[p][3 * i + k], g[3 * i] * R[3 * k] + g[3 * i + 1] * R[3 * k + 1] + g[3 * i + 2] * R[3 * k + 2] + g[9 + i] * r[k]);
                    atomicAdd(&sdG[p][9 + i], g[9 + i]);
                }
            }
        }
        __syncthreads();
    }
    if (j < nj) {
        // relative joint r_j = J_j - J_parent
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&sdJ[j][k], dr[k]);
        if (j > 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&sdJ[sP[j]][k], -dr[k]);
        }
        // Rodrigues derivative: R = I + s K + (1 - c) K^2, K = skew(d), d = rv / th
        const float* rv = fp + ((size_t)b * nj + j) * 3;
        const float x = rv[0] / th, y = rv[1] / th, z = rv[2] / th;
        const float s = sinf(th), c = cosf(th), c1 = 1.f - c;
        const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
        float K2[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) K2[3 * i + k] = K[3 * i] * K[k] + K[3 * i + 1] * K[3 + k] + K[3 * i + 2] * K[6 + k];
        float gs = 0.f, gk2 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { gs += dRT[k] * K[k]; gk2 += dRT[k] * K2[k]; }
        float dK[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // d<dR, K^2>/dK = dR K^T + K^T dR
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 3; ++m) a += dRT[3 * i + m] * K[3 * k + m] + K[3 * m + i] * dRT[3 * m + k];
                dK[3 * i + k] = s * dRT[3 * i + k] + c1 * a;
            }
        const float ddx = dK[7] - dK[5], ddy = dK[2] - dK[6], ddz = dK[3] - dK[1];
        float dth = gs * c + gk2 * s;                               // d/dth of (s K + (1 - c) K^2)
        dth -= (ddx * rv[0] + ddy * rv[1] + ddz * rv[2]) / (th * th);      // d = rv / th
        float* o = d_fp + ((size_t)b * nj + j) * 3;
        o[0] = ddx / th + dth * (rv[0] + 1e-8f) / th;
        o[1] = ddy / th + dth * (rv[1] + 1e-8f) / th;
        o[2] = ddz / th + dth * (rv[2] + 1e-8f) / th;
    }
    __syncthreads();
    if (j < nj && d_J) {
        float* o = d_J + ((size_t)b * nj + j) * 3;
        o[0] = sdJ[j][0]; o[1] = sdJ[j][1]; o[2] = sdJ[j][2];
    }
}

}  // namespace

// fp [nb][nj][3] axis-angle per joint, J [nb or 1][nj][3] rest joints (j_bstride floats between frames, 0 = shared), parents [nj] int32
// (device; parents[0] ignored), nj <= 64.  A [nb][nj][16] and G [nb][nj][12] (global [R | t] per joint, kept for the backward) overwritten.
extern "C" int d3h_smplx_pose_fwd(const float* fp, const float* J, int j_bstride, const int* parents, int nj, int nb, float* A, float* G,
                                  void* stream) {
    if (nj < 1 || nj > NJ_MAX || nb < 0 || (nb > 0 && (!fp || !J || !parents || !A || !G))) return D3H_ERR_ARG;
    if (nb == 0) return D3H_OK;
    hipLaunchKernelGGL(smplx_pose_fwd_kernel, dim3(nb), dim3(64), 0, (hipStream_t)stream, fp, J, j_bstride, parents, nj, A, G);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// dA [nb][nj][16] -> d_fp [nb][nj][3] and d_J [nb][nj][3] (per frame; the caller sums a shared J over the frames), both overwritten.
extern "C" int d3h_smplx_pose_bwd(const float* fp, const float* J, int j_bstride, const int* parents, int nj, int nb, const float* G,
                                  const float* dA, float* d_fp, float* d_J, void* stream) {
    if (nj < 1 || nj > NJ_MAX || nb < 0 || (nb > 0 && (!fp || !J || !parents || !G || !dA || !d_fp))) return D3H_ERR_ARG;
    if (nb == 0) return D3H_OK;
    hipLaunchKernelGGL(smplx_pose_bwd_kernel, dim3(nb), dim3(64), 0, (hipStream_t)stream, fp, J, j_bstride, parents, nj, G, dA, d_fp, d_J);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
