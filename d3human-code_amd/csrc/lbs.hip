// lbs.hip -- SMPL-X driven linear blend skinning of arbitrary surface points on gfx950.
//
// Replaces deform/smplx_exavatar_deformer.py:
//   interpolate_weights :363-383   K=1 nearest template vertex (pytorch3d knn_points -> third_parties/pytorch3d/cuda/knn.cu:205,
//                                  knn_cpu.cpp:13-69: squared L2, first minimum wins) + gather of its 55 skin weights
//   apply_lbs_inverse   :385-421   M_p = sum_j w_pj A_j  (the reference materialises [1,P,55,4,4] = 352 MB at P=1e5), inverse, apply
//   lbs_forward         :434-486   canonical = M0_p^-1 [p;1] with the init-pose transforms, posed = M_p [canonical;1] + trans
//
// MI355X design: one thread per point.  The nearest-neighbour search streams the 10 475-vertex template through LDS
// in 16-B broadcast reads (every lane reads the same vertex: conflict-free, no HBM re-reads; the template is 126 KB).
// The blend never materialises per-point joint stacks: the 55 weights of the nearest vertex are read once (220 B row,
// L2-resident table) and folded into a 3x4 matrix in registers.  The nearest-vertex id depends only on the canonical
// mesh, so it is computed once per iteration and shared by all frames of the batch.  The backward accumulates
// d(A) (3x4 per joint per frame) in LDS with ds_add_f32 -- skin weights are sparse (<= ~4 non-zeros per vertex) --
// and issues one global atomic per touched entry per workgroup.
#include "d3h_common.h"

namespace {

constexpr int KNN_TILE = 2048;   // template vertices per LDS tile (32 KB as float4)

// 16 queries per workgroup, each scanned by 16 lanes over interleaved slices of the template tile (a posed mesh has only ~10^4
// vertices: one query per thread leaves most of the chip idle and serialises 10 475 distance evaluations per thread).  The final
// (distance, index) reduction keeps the sequential scan's answer: smallest distance, lowest index among equals.
constexpr int KNN_Q = 16, KNN_PARTS = 16;

__global__ __launch_bounds__(256) void knn1_kernel(const float* __restrict__ pts, int np, const float* __restrict__ tmpl, int nv,
                                                   int* __restrict__ idx_out, float* __restrict__ dist_out) {
    __shared__ float4 tile[KNN_TILE];
    __shared__ float sd[256];
    __shared__ int si[256];
    const int q = threadIdx.x & (KNN_Q - 1), part = threadIdx.x / KNN_Q;
    const int p = blockIdx.x * KNN_Q + q;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (p < np) { px = pts[3 * (size_t)p]; py = pts[3 * (size_t)p + 1]; pz = pts[3 * (size_t)p + 2]; }
    float best = INFINITY;
    int besti = 0x7fffffff;
    for (int base = 0; base < nv; base += KNN_TILE) {
        int cnt = min(KNN_TILE, nv - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const float* v = tmpl + 3 * (size_t)(base + i);
            tile[i] = make_float4(v[0], v[1], v[2], 0.f);
        }
        __syncthreads();
        for (int i = part; i < cnt; i += KNN_PARTS) {
            float4 v = tile[i];
            float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
            float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));   // knn_cpu.cpp:36-40 order
            if (d < best) { best = d; besti = base + i; }     // strict <: first minimum of this slice
        }
    }
    sd[threadIdx.x] = best;
    si[threadIdx.x] = besti;
    __syncthreads();
    if (part == 0) {
        for (int k = 1; k < KNN_PARTS; ++k) {
            float d = sd[k * KNN_Q + q];
            int i = si[k * KNN_Q + q];
            if (d < best || (d == best && i < besti)) { best = d; besti = i; }
        }
        if (p < np) {
            idx_out[p] = besti == 0x7fffffff ? 0 : besti;     // all-NaN query: index 0, like the sequential scan
            if (dist_out) dist_out[p] = best;
        }
    }
}

// ---- grid-accelerated nearest vertex (same answer as knn1_kernel, ~100x fewer distance evaluations) -----------------------------
// The SMPL-X template is fixed after SMPLX_Deformer.initialize(), so its vertices are binned once (host, d3h/lbs.py:KnnGrid) into a
// uniform grid, stored cell-sorted as float4 (x, y, z, original index bits) with cell id = (z gy + y) gx + x: the cells of one
// x-row are contiguous, so a (2r+1)^3 cube of cells is (2r+1)^2 contiguous segments.  16 lanes share a query and stride over the
// segments.  Phase 1 scans the 3x3x3 cells around the query and accepts the best candidate when its squared distance is provably
// smaller than the distance to the nearest unscanned cell face (with a safety margin that covers the rounding of the binning);
// otherwise phase 2 adds the cell's precomputed seed vertex (nearest to the cell centre) as a candidate and scans the cells of the
// box of radius sqrt(best) around the query, which must contain the answer (queries away from the body).  Candidates are compared on (distance, original index), the distance is evaluated with
// the same expression as the exhaustive kernel, so the result is bit-identical to it: smallest distance, lowest index among equals.
struct KnnGrid {
    float lox, loy, loz, h, inv_h;
    int gx, gy, gz;
};

__device__ __forceinline__ void knn_take(float d, int i, float& best, int& besti) {
    if (d < best || (d == best && i < besti)) { best = d; besti = i; }
}

// scan the cells [x0..x1] x [y0..y1] x [z0..z1] with the 16 lanes of the query's group, then make every lane hold the group's best
__device__ __forceinline__ void knn_scan_box(const float4* __restrict__ cpts, const int* __restrict__ cstart, const KnnGrid& g, int x0, int x1,
                                             int y0, int y1, int z0, int z1, float px, float py, float pz, int l16, float& best, int& besti) {
    const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
    for (int s = 0; s < nrow; ++s) {
        const int y = y0 + s % ny, z = z0 + s / ny;
        const int row = (z * g.gy + y) * g.gx;
        const int a = cstart[row + x0], b = cstart[row + x1 + 1];
        for (int i = a + l16; i < b; i += 16) {
            float4 v = cpts[i];
            float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
            float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));   // knn_cpu.cpp:36-40 order
            knn_take(d, __float_as_int(v.w), best, besti);
        }
    }
#pragma unroll
    for (int off = 8; off; off >>= 1) {
        float od = __shfl_xor(best, off, 16);
        int oi = __shfl_xor(besti, off, 16);
        knn_take(od, oi, best, besti);
    }
}

// the same scan with one lane per x-row (many short rows: the per-row latency chain is spread over the 16 lanes)
__device__ __forceinline__ void knn_scan_rows(const float4* __restrict__ cpts, const int* __restrict__ cstart, const KnnGrid& g, int x0, int x1,
                                              int y0, int y1, int z0, int z1, float px, float py, float pz, int l16, float& best, int& besti) {
    const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
    for (int s = l16; s < nrow; s += 16) {
        const int y = y0 + s % ny, z = z0 + s / ny;
        const int row = (z * g.gy + y) * g.gx;
        const int a = cstart[row + x0], b = cstart[row + x1 + 1];
        for (int i = a; i < b; ++i) {
            float4 v = cpts[i];
            float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
            float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            knn_take(d, __float_as_int(v.w), best, besti);
        }
    }
#pragma unroll
    for (int off = 8; off; off >>= 1) {
        float od = __shfl_xor(best, off, 16);
        int oi = __shfl_xor(besti, off, 16);
        knn_take(od, oi, best, besti);
    }
}

__device__ __forceinline__ int knn_cell(float v, float lo, float inv_h, int n) {
    return min(max((int)floorf(fminf(fmaxf((v - lo) * inv_h, -1.f), (float)n)), 0), n - 1);
}

// Launches queued before the host knows how many rows the marching-tets extraction produced (d3h/mtets.py: speculative extraction) cover the
// CAPACITY of the vertex buffer and read the real count from the extraction's device-side counters: rows = P_wt + 3 n1 + 4 n2 (counts[0..2]).
__device__ __forceinline__ int counted_rows(int np, const int* __restrict__ counted) {
    if (!counted) return np;
    if (counted[10]) return 0;                             // the extraction outgrew its capacity and wrote nothing: the caller repeats it
    const long long r = (long long)counted[0] + 3LL * counted[1] + 4LL * counted[2];
    return r < np ? (int)r : np;
}

__global__ __launch_bounds__(256) void knn1_grid_kernel(const float* __restrict__ pts, int np, const float4* __restrict__ cpts,
                                                        const int* __restrict__ cstart, const int* __restrict__ cseed, int nv, KnnGrid g,
                                                        int* __restrict__ idx_out, float* __restrict__ dist_out, const int* __restrict__ counted) {
    const int l16 = threadIdx.x & 15;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    np = counted_rows(np, counted);
    if (p >= np) return;                                   // whole 16-lane groups leave together
    const float px = pts[3 * (size_t)p], py = pts[3 * (size_t)p + 1], pz = pts[3 * (size_t)p + 2];
    float best = INFINITY;
    int besti = 0x7fffffff;
    if (px == px && py == py && pz == pz) {                // a NaN query compares false against everything: index 0, distance inf
        const int cx = knn_cell(px, g.lox, g.inv_h, g.gx), cy = knn_cell(py, g.loy, g.inv_h, g.gy), cz = knn_cell(pz, g.loz, g.inv_h, g.gz);
        const float margin = 1e-4f * g.h;
        bool done = false;
        // phase 1: the 3x3x3 cells around the query; accepted when the best candidate provably beats every unscanned cell
        {
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
            const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.gy - 1);
            const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.gz - 1);
            knn_scan_box(cpts, cstart, g, x0, x1, y0, y1, z0, z1, px, py, pz, l16, best, besti);
            float bound = INFINITY;                        // distance to the nearest face of the scanned cube with cells behind it
            if (cx - 1 > 0) bound = fminf(bound, px - (g.lox + (float)x0 * g.h));
            if (cx + 1 < g.gx - 1) bound = fminf(bound, (g.lox + (float)(x1 + 1) * g.h) - px);
            if (cy - 1 > 0) bound = fminf(bound, py - (g.loy + (float)y0 * g.h));
            if (cy + 1 < g.gy - 1) bound = fminf(bound, (g.loy + (float)(y1 + 1) * g.h) - py);
            if (cz - 1 > 0) bound = fminf(bound, pz - (g.loz + (float)z0 * g.h));
            if (cz + 1 < g.gz - 1) bound = fminf(bound, (g.loz + (float)(z1 + 1) * g.h) - pz);
            bound -= margin;
            done = bound > 0.f && best < bound * bound;
        }
        // phase 2: the cell's seed (the vertex nearest to the cell centre) bounds the answer's distance even when the neighbourhood is
        // empty; the nearest vertex lies in the box of radius sqrt(best) around the query -- scan its cells
        if (!done) {
            float4 v = cpts[cseed[(cz * g.gy + cy) * g.gx + cx]];
            float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
            knn_take(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), __float_as_int(v.w), best, besti);
            int x0 = 0, x1 = g.gx - 1, y0 = 0, y1 = g.gy - 1, z0 = 0, z1 = g.gz - 1;
            if (best < INFINITY) {
                const float rad = sqrtf(best) * 1.00001f + margin;
                x0 = knn_cell(px - rad, g.lox, g.inv_h, g.gx); x1 = knn_cell(px + rad, g.lox, g.inv_h, g.gx);
                y0 = knn_cell(py - rad, g.loy, g.inv_h, g.gy); y1 = knn_cell(py + rad, g.loy, g.inv_h, g.gy);
                z0 = knn_cell(pz - rad, g.loz, g.inv_h, g.gz); z1 = knn_cell(pz + rad, g.loz, g.inv_h, g.gz);
            }
            if ((y1 - y0 + 1) * (z1 - z0 + 1) >= 16) knn_scan_rows(cpts, cstart, g, x0, x1, y0, y1, z0, z1, px, py, pz, l16, best, besti);
            else knn_scan_box(cpts, cstart, g, x0, x1, y0, y1, z0, z1, px, py, pz, l16, best, besti);
        }
    }
    if (l16 == 0) {
        idx_out[p] = besti == 0x7fffffff ? 0 : besti;
        if (dist_out) dist_out[p] = best;
    }
}

// M[3][4] = sum_j w[j] * A[j][0:3][0:4]; s = sum_j w[j] * A[j][3][3]
__device__ __forceinline__ void blend(const float* __restrict__ w, const float* __restrict__ A, int nj, float (&M)[12], float& s) {
#pragma unroll
    for (int e = 0; e < 12; ++e) M[e] = 0.f;
    s = 0.f;
    for (int j = 0; j < nj; ++j) {
        float wj = w[j];
        if (wj != 0.f) {
            const float* a = A + 16 * j;
#pragma unroll
            for (int e = 0; e < 12; ++e) M[e] = fmaf(wj, a[e], M[e]);
            s = fmaf(wj, a[15], s);
        }
    }
}

__device__ __forceinline__ void inv3(const float (&M)[12], float (&R)[9]) {
    float a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    float c0 = e * i - f * h, c1 = f * g - d * i, c2 = d * h - e * g;
    float det = a * c0 + b * c1 + c * c2;
    float id = 1.0f / det;
    R[0] = c0 * id; R[1] = (c * h - b * i) * id; R[2] = (b * f - c * e) * id;
    R[3] = c1 * id; R[4] = (a * i - c * g) * id; R[5] = (c * d - a * f) * id;
    R[6] = c2 * id; R[7] = (b * g - a * h) * id; R[8] = (a * e - b * d) * id;
}

// canonical point: xyz of M0^-1 [p;1], M0 = [[R t],[0 0 0 s]]  ->  R^-1 (p - t/s)
__device__ __forceinline__ void to_canonical(const float (&M0)[12], float s0, float px, float py, float pz, float (&Rinv)[9], float (&pc)[3]) {
    inv3(M0, Rinv);
    float is = 1.0f / s0;
    float qx = px - M0[3] * is, qy = py - M0[7] * is, qz = pz - M0[11] * is;
    pc[0] = Rinv[0] * qx + Rinv[1] * qy + Rinv[2] * qz;
    pc[1] = Rinv[3] * qx + Rinv[4] * qy + Rinv[5] * qz;
    pc[2] = Rinv[6] * qx + Rinv[7] * qy + Rinv[8] * qz;
}

// one thread per (vertex, frame): blockIdx.y = frame.  (The first version looped over the frames inside the thread: 27 000 vertices are
// 420 waves, fewer than one per SIMD, each walking 5 x 55 gathered weights in sequence -- 59 us for 4 frames.)
__global__ __launch_bounds__(256) void lbs_fwd_kernel(const float* __restrict__ pts, int np, const int* __restrict__ idx,
                                                      const float* __restrict__ lbs_w, int nj, const float* __restrict__ A0,
                                                      const float* __restrict__ A /*[nb][nj][16]*/, const float* __restrict__ trans /*[nb][3]*/,
                                                      int nb, float* __restrict__ out /*[nb][np][3]*/, float* __restrict__ pts_can,
                                                      const int* __restrict__ counted) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    np = counted_rows(np, counted);                        // (also the frame pitch of `out`: the frames stay dense at the real row count)
    if (p >= np) return;
    const float* w = lbs_w + (size_t)idx[p] * nj;
    float M0[12], s0, Rinv[9], pc[3];
    blend(w, A0, nj, M0, s0);
    to_canonical(M0, s0, pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2], Rinv, pc);
    if (pts_can && b == 0) { pts_can[3 * (size_t)p] = pc[0]; pts_can[3 * (size_t)p + 1] = pc[1]; pts_can[3 * (size_t)p + 2] = pc[2]; }
    float M[12], s;
    blend(w, A + (size_t)b * nj * 16, nj, M, s);
    float* o = out + ((size_t)b * np + p) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r)
        o[r] = (M[4 * r] * pc[0] + M[4 * r + 1] * pc[1] + M[4 * r + 2] * pc[2] + M[4 * r + 3]) + trans[3 * b + r];
}

constexpr int MAXJ = 64;

__global__ __launch_bounds__(256) void lbs_bwd_kernel(const float* __restrict__ pts, int np, const int* __restrict__ idx,
                                                      const float* __restrict__ lbs_w, int nj, const float* __restrict__ A0,
                                                      const float* __restrict__ A, int nb, const float* __restrict__ gout /*[nb][np][3]*/,
                                                      float* __restrict__ d_pts /*[nb][np][3]: one slice per frame*/,
                                                      float* __restrict__ dA /*[nb][nj][16] or null*/,
                                                      float* __restrict__ d_trans /*[nb][3] or null*/) {
    __shared__ float sA[MAXJ * 12];
    __shared__ float sT[3];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;                       // one frame per workgroup row; every frame writes its own d_pts slice (summed in fixed order below)
    const bool valid = p < np;
    const float* w = lbs_w + (size_t)(valid ? idx[p] : 0) * nj;
    float M0[12], s0, Rinv[9], pc[3];
    if (valid) {
        blend(w, A0, nj, M0, s0);
        to_canonical(M0, s0, pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2], Rinv, pc);
    }
    float gpc[3] = {0.f, 0.f, 0.f};
    if (dA) {
        for (int i = threadIdx.x; i < nj * 12; i += 256) sA[i] = 0.f;
    }
    if (threadIdx.x < 3) sT[threadIdx.x] = 0.f;
    __syncthreads();
    if (valid) {
        float M[12], s;
        blend(w, A + (size_t)b * nj * 16, nj, M, s);
        const float* g = gout + ((size_t)b * np + p) * 3;
        float g0 = g[0], g1 = g[1], g2 = g[2];
        gpc[0] = M[0] * g0 + M[4] * g1 + M[8] * g2;
        gpc[1] = M[1] * g0 + M[5] * g1 + M[9] * g2;
        gpc[2] = M[2] * g0 + M[6] * g1 + M[10] * g2;
        if (d_trans) { atomicAdd(&sT[0], g0); atomicAdd(&sT[1], g1); atomicAdd(&sT[2], g2); }
        if (dA) {
            float dM[12] = {g0 * pc[0], g0 * pc[1], g0 * pc[2], g0, g1 * pc[0], g1 * pc[1], g1 * pc[2], g1,
                            g2 * pc[0], g2 * pc[1], g2 * pc[2], g2};
            for (int j = 0; j < nj; ++j) {
                float wj = w[j];
                if (wj != 0.f) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) atomicAdd(&sA[j * 12 + e], wj * dM[e]);
                }
            }
        }
    }
    __syncthreads();
    if (dA) {
        for (int i = threadIdx.x; i < nj * 12; i += 256) {
            float v = sA[i];
            if (v != 0.f) atomicAdd(&dA[((size_t)b * nj + i / 12) * 16 + (i % 12)], v);
        }
    }
    if (d_trans && threadIdx.x < 3) atomicAdd(&d_trans[3 * b + threadIdx.x], sT[threadIdx.x]);
    if (valid && d_pts) {
        // pc = Rinv (p - t/s)  ->  d p = Rinv^T d pc
        float* o = d_pts + ((size_t)b * np + p) * 3;
        o[0] = Rinv[0] * gpc[0] + Rinv[3] * gpc[1] + Rinv[6] * gpc[2];
        o[1] = Rinv[1] * gpc[0] + Rinv[4] * gpc[1] + Rinv[7] * gpc[2];
        o[2] = Rinv[2] * gpc[0] + Rinv[5] * gpc[1] + Rinv[8] * gpc[2];
    }
}

// d_pts[i] = frames[0][i] + frames[1][i] + ... in frame order: the mesh-vertex gradient (and with it every SDF / deform update) is
// bit-reproducible from run to run, which float atomics across the frames' workgroups were not for nb > 2
__global__ __launch_bounds__(256) void lbs_bwd_sum_frames_kernel(const float* __restrict__ frames, long long n, int nb, float* __restrict__ d_pts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = frames[i];
    for (int b = 1; b < nb; ++b) acc += frames[(size_t)b * n + i];
    d_pts[i] = acc;
}

}  // namespace

extern "C" int d3h_knn1(const float* pts, int np, const float* tmpl, int nv, int* idx, float* dist, void* stream) {
    if (np < 0 || nv <= 0 || (np > 0 && (!pts || !tmpl || !idx))) return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    hipLaunchKernelGGL(knn1_kernel, dim3(d3h_cdiv(np, KNN_Q)), dim3(256), 0, (hipStream_t)stream, pts, np, tmpl, nv, idx, dist);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// Grid-accelerated d3h_knn1 (identical results).  cell_pts [nv] float4 = template vertices sorted by cell id (z gy + y) gx + x with the
// original index in the bits of .w (ascending inside a cell), cell_start [gx gy gz + 1] = first sorted position of every cell;
// cell_seed [gx gy gz] = sorted position of the vertex nearest to the cell centre; lo HOST[3] = grid origin, h = cell edge:
// cell(v) = clamp(floor((v - lo) / h), 0, g - 1) and every template vertex lies inside the grid box.  Built once per template by
// d3h/lbs.py:KnnGrid.
extern "C" int d3h_knn1_grid(const float* pts, int np, const float* cell_pts, const int* cell_start, const int* cell_seed, int nv,
                             const float* lo, float h, int gx, int gy, int gz, int* idx, float* dist, void* stream) {
    if (np < 0 || nv <= 0 || gx <= 0 || gy <= 0 || gz <= 0 || !(h > 0.f) || !lo ||
        (np > 0 && (!pts || !cell_pts || !cell_start || !cell_seed || !idx)))
        return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    KnnGrid g{lo[0], lo[1], lo[2], h, 1.0f / h, gx, gy, gz};
    hipLaunchKernelGGL(knn1_grid_kernel, dim3(d3h_cdiv(np, 16)), dim3(256), 0, (hipStream_t)stream, pts, np, (const float4*)cell_pts,
                       cell_start, cell_seed, nv, g, idx, dist, (const int*)nullptr);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// the same over the first min(np, counts3[0] + 3 counts3[1] + 4 counts3[2]) rows of pts, the counters read ON THE DEVICE (np = capacity)
extern "C" int d3h_knn1_grid_counted(const float* pts, int np, const int* counts3, const float* cell_pts, const int* cell_start, const int* cell_seed,
                                     int nv, const float* lo, float h, int gx, int gy, int gz, int* idx, void* stream) {
    if (np < 0 || nv <= 0 || gx <= 0 || gy <= 0 || gz <= 0 || !(h > 0.f) || !lo || !counts3 ||
        (np > 0 && (!pts || !cell_pts || !cell_start || !cell_seed || !idx)))
        return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    KnnGrid g{lo[0], lo[1], lo[2], h, 1.0f / h, gx, gy, gz};
    hipLaunchKernelGGL(knn1_grid_kernel, dim3(d3h_cdiv(np, 16)), dim3(256), 0, (hipStream_t)stream, pts, np, (const float4*)cell_pts,
                       cell_start, cell_seed, nv, g, idx, (float*)nullptr, counts3);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

extern "C" int d3h_lbs_fwd(const float* pts, int np, const int* idx, const float* lbs_w, int nj, const float* A0, const float* A,
                           const float* trans, int nb, float* out, float* pts_can, void* stream) {
    if (np < 0 || nj <= 0 || nj > MAXJ || nb <= 0) return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    const int kt_ = d3h_ktime_begin(D3H_KT_LBS_FWD, (long long)((long long)np * nb), (hipStream_t)(stream));
    hipLaunchKernelGGL(lbs_fwd_kernel, dim3(d3h_cdiv(np, 256), nb), dim3(256), 0, (hipStream_t)stream, pts, np, idx, lbs_w, nj, A0, A, trans, nb,
                       out, pts_can, (const int*)nullptr);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// the same over the first r = min(np, counts3[0] + 3 counts3[1] + 4 counts3[2]) rows, r read ON THE DEVICE (np = capacity); `out` (nb * np * 3
// floats) then holds the dense [nb][r][3] result in its leading floats
extern "C" int d3h_lbs_fwd_counted(const float* pts, int np, const int* counts3, const int* idx, const float* lbs_w, int nj, const float* A0,
                                   const float* A, const float* trans, int nb, float* out, void* stream) {
    if (np < 0 || nj <= 0 || nj > MAXJ || nb <= 0 || !counts3) return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    const int kt_ = d3h_ktime_begin(D3H_KT_LBS_FWD, (long long)((long long)np * nb), (hipStream_t)(stream));
    hipLaunchKernelGGL(lbs_fwd_kernel, dim3(d3h_cdiv(np, 256), nb), dim3(256), 0, (hipStream_t)stream, pts, np, idx, lbs_w, nj, A0, A, trans, nb,
                       out, (float*)nullptr, counts3);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// d_pts [np][3] is overwritten; dA / d_trans are accumulated into (caller zero-fills).  d_pts_frames: scratch [nb][np][3] holding every
// frame's contribution, summed into d_pts in frame order (deterministic); may be NULL when nb == 1 (the one frame writes d_pts itself).
extern "C" int d3h_lbs_bwd(const float* pts, int np, const int* idx, const float* lbs_w, int nj, const float* A0, const float* A, int nb,
                           const float* gout, float* d_pts, float* d_pts_frames, float* dA, float* d_trans, void* stream) {
    if (np < 0 || nj <= 0 || nj > MAXJ || nb <= 0) return D3H_ERR_ARG;
    if (np == 0) return D3H_OK;
    if (d_pts && nb > 1 && !d_pts_frames) return D3H_ERR_ARG;
    float* per_frame = !d_pts ? nullptr : (nb > 1 ? d_pts_frames : d_pts);
    const int kt_ = d3h_ktime_begin(D3H_KT_LBS_BWD, (long long)((long long)np * nb), (hipStream_t)(stream));
    hipLaunchKernelGGL(lbs_bwd_kernel, dim3(d3h_cdiv(np, 256), nb), dim3(256), 0, (hipStream_t)stream, pts, np, idx, lbs_w, nj, A0, A, nb, gout,
                       per_frame, dA, d_trans);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    if (d_pts && nb > 1) {
        const long long n = (long long)np * 3;
        hipLaunchKernelGGL(lbs_bwd_sum_frames_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_pts_frames, n, nb, d_pts);
        D3H_LAUNCH_CHECK();
    }
    return D3H_OK;
}
