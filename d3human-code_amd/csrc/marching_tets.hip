// marching_tets.hip -- sort-free G-Shell marching tetrahedra for gfx950 (forward + VJP).
//
// Replaces geometry/gshell_tets.py:253-447 (GShell_Tets.__call__) and geometry/hmsdf_tets_split.py:254-454
// (hmSDF_Tets.__call__; same algorithm, mSDF negated for type == "body").  The reference runs ~100 small
// kernels, two torch.unique (one a row sort of <= 6*N_valid int64 pairs) and ~15 host syncs per call.
//
// MI355X design (DESIGN.md §marching tets): the tet grid is static, so the global sorted-unique edge list
// E and the per-tet edge ids are built ONCE (geometry/hmsdf.py:382-388 already builds E).  A crossing edge's
// rank in torch.unique's output (gshell_tets.py:279-287) equals its rank among crossing edges of E, so vertex
// ids come from one ordered compaction (wave ballot + prefix sums) -- no sort, no hash.  Face groups (1-tri
// tets, then 2-tri tets, then the six mSDF-cut groups, :322-325,:413-420) are ordered compactions over tet ids.
// Everything is HBM-bound integer/byte work: int32 indices, one uint8 case code per tet, coalesced sweeps.
//
// Arithmetic follows the reference operation by operation (separate mul/add roundings, IEEE division) so that
// every sign test (sdf > 0, msdf_vert > 0) -- hence every index -- is bit-identical given the same inputs.
#include <chrono>
#include <cstring>
#include "d3h_common.h"

namespace {

__constant__ int8_t c_num_tri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};
__constant__ int8_t c_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
// polygon boundary loop (first 3 entries for a triangle, 4 for a quad)
__constant__ int8_t c_loop_table[16][4] = {
    {-1, -1, -1, -1}, {1, 0, 2, -1}, {4, 0, 3, -1}, {1, 3, 4, 2}, {3, 1, 5, -1}, {2, 5, 3, 0}, {1, 5, 4, 0}, {4, 2, 5, -1},
    {4, 5, 2, -1},    {4, 5, 1, 0},  {3, 5, 2, 0},  {1, 3, 5, -1}, {4, 3, 1, 2}, {3, 0, 4, -1}, {2, 0, 1, -1}, {-1, -1, -1, -1}};
__constant__ int8_t c_num_tri3[8] = {0, 1, 1, 2, 1, 2, 2, 1};
__constant__ int8_t c_num_tri4[16] = {0, 1, 1, 2, 1, 4, 2, 3, 1, 2, 4, 3, 2, 3, 3, 2};
__constant__ int8_t c_tri3_table[8][6] = {{-1, -1, -1, -1, -1, -1}, {4, 2, 5, -1, -1, -1}, {3, 1, 4, -1, -1, -1},
                                          {3, 1, 2, 3, 2, 5},       {0, 3, 5, -1, -1, -1}, {0, 3, 4, 0, 4, 2},
                                          {0, 1, 4, 0, 4, 5},       {0, 1, 2, -1, -1, -1}};
__constant__ int8_t c_tri4_table[16][12] = {
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, {6, 3, 7, -1, -1, -1, -1, -1, -1, -1, -1, -1},
    {5, 2, 6, -1, -1, -1, -1, -1, -1, -1, -1, -1},    {5, 2, 7, 3, 7, 2, -1, -1, -1, -1, -1, -1},
    {4, 1, 5, -1, -1, -1, -1, -1, -1, -1, -1, -1},    {4, 1, 5, 4, 5, 7, 5, 6, 7, 7, 6, 3},
    {4, 1, 2, 6, 4, 2, -1, -1, -1, -1, -1, -1},       {4, 1, 2, 7, 4, 2, 7, 2, 3, -1, -1, -1},
    {0, 4, 7, -1, -1, -1, -1, -1, -1, -1, -1, -1},    {0, 4, 6, 3, 0, 6, -1, -1, -1, -1, -1, -1},
    {0, 4, 5, 0, 5, 2, 0, 2, 6, 0, 6, 7},             {0, 4, 5, 0, 5, 2, 0, 2, 3, -1, -1, -1},
    {0, 1, 5, 7, 0, 5, -1, -1, -1, -1, -1, -1},       {0, 1, 5, 0, 5, 6, 0, 6, 3, -1, -1, -1},
    {0, 1, 2, 0, 2, 6, 0, 6, 7, -1, -1, -1},          {0, 1, 2, 0, 2, 3, -1, -1, -1, -1, -1, -1}};

constexpr int BLK = 256;

// Speculative extraction (d3h_mtets_emit_spec): the emit kernels are queued BEFORE the host knows the output sizes, into buffers allocated at
// the capacities `Caps` (the previous extraction's sizes plus a margin).  counts[0..2] are on the device by then (pass A + its scans): a launch
// whose sizes exceed a capacity returns at once, uniformly, having written nothing -- the host sees the same counts a moment later and repeats
// the extraction at the exact sizes.  The exact-size entry points pass INT_MAX.
struct Caps { int pwt, n1, n2; };
__device__ __forceinline__ bool over_caps(const int* __restrict__ counts, Caps c) {
    return counts[0] > c.pwt || counts[1] > c.n1 || counts[2] > c.n2;
}
constexpr Caps NO_CAPS = {0x7fffffff, 0x7fffffff, 0x7fffffff};

// ordered rank of `flag` inside a 256-thread block (4 waves); also returns the block total
__device__ __forceinline__ int block_rank(bool flag, int* s_wave /*[4]*/, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m = __ballot(flag);
    int r = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int c = s_wave[w];
        if (w < wave) base += c;
        total += c;
    }
    __syncthreads();
    return base + r;
}

// ---- pass A: per-block counts ------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void mt_count_edges(const float* __restrict__ sdf, const int* __restrict__ edges, int ne,
                                                      int* __restrict__ blk_e) {
    __shared__ int s_wave[4];
    int e = blockIdx.x * BLK + threadIdx.x;
    bool cross = false;
    if (e < ne) {
        int2 ab = *(const int2*)(edges + 2 * (size_t)e);
        cross = (sdf[ab.x] > 0.f) != (sdf[ab.y] > 0.f);
    }
    int total;
    block_rank(cross, s_wave, total);
    if (threadIdx.x == 0) blk_e[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLK) void mt_count_tets(const float* __restrict__ sdf, const int* __restrict__ tets, int nt,
                                                     uint8_t* __restrict__ tet_code, int* __restrict__ blk_t /*[nb][8]*/) {
    __shared__ int s_wave[4];
    int t = blockIdx.x * BLK + threadIdx.x;
    int code = 0;
    if (t < nt) {
        int4 v = *(const int4*)(tets + 4 * (size_t)t);
        code = (sdf[v.x] > 0.f ? 1 : 0) | (sdf[v.y] > 0.f ? 2 : 0) | (sdf[v.z] > 0.f ? 4 : 0) | (sdf[v.w] > 0.f ? 8 : 0);
        tet_code[t] = (uint8_t)code;
    }
    int n = c_num_tri[code];
    int t1, t2;
    block_rank(n == 1, s_wave, t1);
    block_rank(n == 2, s_wave, t2);
    if (threadIdx.x == 0) {
        blk_t[blockIdx.x * 8 + 0] = t1;
        blk_t[blockIdx.x * 8 + 1] = t2;
    }
}

// ---- exclusive scan of per-block counters (single workgroup; any nb: segments of 8192 entries) ------
// cnt: [nb][stride] -> in-place exclusive prefix for columns c0..c0+nc-1; totals -> out_counts[oc0 + c]
constexpr int MT_SCAN_MAXC = 6, MT_SCAN_PER = 8;
// Exclusive scan of nc (<= 6) interleaved count columns over nb workgroup entries, in place, totals to out_counts.  One
// workgroup; it sits on the launch-bound stretch between the SDF sweep and the mesh consumers, so what matters is latency: all
// columns advance together, every thread's (<= 8) entries are loaded up front in one burst and written back from registers, and the
// 1024 partials are scanned inside the waves plus one LDS hop for the 16 wave totals (the first version: per-column passes, dependent
// re-reads, a 10-step block-wide Hillis-Steele scan -- 3 launches took 110 us).
__global__ __launch_bounds__(1024) void mt_scan(int* __restrict__ cnt, int nb, int stride, int c0, int nc,
                                                int* __restrict__ out_counts, int oc0) {
    __shared__ int s_wave[MT_SCAN_MAXC][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Segments of 1024 x MT_SCAN_PER entries with a running carry: ONE segment up to 2 097 152 tets / edges (every grid of rounds 1-3, the
    // latency-optimised case the comment above describes); larger grids loop (round 4: the 2 M limit was a hard D3H_ERR_ARG).
    int carry[MT_SCAN_MAXC];
#pragma unroll
    for (int k = 0; k < MT_SCAN_MAXC; ++k) carry[k] = 0;
    for (int base = 0; base < nb; base += 1024 * MT_SCAN_PER) {
        const int nbs = nb - base < 1024 * MT_SCAN_PER ? nb - base : 1024 * MT_SCAN_PER;
        const int per = (nbs + 1023) / 1024;
        const int lo = base + tid * per, hi = base + nbs;
        int val[MT_SCAN_PER][MT_SCAN_MAXC];
#pragma unroll
        for (int j = 0; j < MT_SCAN_PER; ++j)
#pragma unroll
            for (int k = 0; k < MT_SCAN_MAXC; ++k)
                val[j][k] = (j < per && lo + j < hi && k < nc) ? cnt[(size_t)(lo + j) * stride + c0 + k] : 0;
        int sum[MT_SCAN_MAXC], incl[MT_SCAN_MAXC];
#pragma unroll
        for (int k = 0; k < MT_SCAN_MAXC; ++k) {
            sum[k] = 0;
#pragma unroll
            for (int j = 0; j < MT_SCAN_PER; ++j) sum[k] += val[j][k];
            int x = sum[k];                             // inclusive scan inside the wave, then one LDS hop for the 16 wave totals
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            incl[k] = x;
            if (lane == 63) s_wave[k][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MT_SCAN_MAXC; ++k) {
            if (k >= nc) continue;
            int off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                int t = s_wave[k][w];
                if (w < wave) off += t;
                tot += t;
            }
            int run = carry[k] + off + incl[k] - sum[k];           // exclusive
            carry[k] += tot;
#pragma unroll
            for (int j = 0; j < MT_SCAN_PER; ++j) {
                if (j < per && lo + j < hi) cnt[(size_t)(lo + j) * stride + c0 + k] = run;
                run += val[j][k];
            }
        }
        __syncthreads();                                // s_wave is rewritten by the next segment
    }
#pragma unroll
    for (int k = 0; k < MT_SCAN_MAXC; ++k)
        if (k < nc && tid == 1023) out_counts[oc0 + k] = carry[k];
}

// ---- pass B: watertight vertices (one per crossing edge, in global edge order) ----------------------
__global__ __launch_bounds__(BLK) void mt_emit_verts(const float* __restrict__ pos, const float* __restrict__ sdf,
                                                     const float* __restrict__ msdf, float msdf_sign,
                                                     const int* __restrict__ edges, int ne, const int* __restrict__ blk_e,
                                                     int* __restrict__ edge_vid, float* __restrict__ verts_wt,
                                                     float* __restrict__ msdf_vert, int* __restrict__ vert_edge,
                                                     int* __restrict__ counts, Caps caps) {
    __shared__ int s_wave[4];
    // counts[10] = "this speculative extraction wrote nothing": read by the counted launches queued behind it (csrc/lbs.hip: counted_rows)
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[10] = over_caps(counts, caps) ? 1 : 0;
    if (over_caps(counts, caps)) return;
    int e = blockIdx.x * BLK + threadIdx.x;
    bool cross = false;
    int2 ab = make_int2(0, 0);
    float s0 = 0.f, s1 = 0.f;
    if (e < ne) {
        ab = *(const int2*)(edges + 2 * (size_t)e);
        s0 = sdf[ab.x];
        s1 = sdf[ab.y];
        cross = (s0 > 0.f) != (s1 > 0.f);
    }
    int total;
    int r = block_rank(cross, s_wave, total);
    if (e >= ne) return;
    if (!cross) { edge_vid[e] = -1; return; }
    int vid = blk_e[blockIdx.x] + r;
    edge_vid[e] = vid;
    // gshell_tets.py:291-300
    float a = s0, b = -s1;
    float den = __fadd_rn(a, b);
    float sg = (den > 0.f) ? 1.f : ((den < 0.f) ? -1.f : 0.f);
    den = __fmul_rn(sg, __fadd_rn(fabsf(den), 1e-12f));
    if (den == 0.f) den = 1e-12f;
    float w0 = b / den, w1 = a / den;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        verts_wt[3 * (size_t)vid + c] = __fadd_rn(__fmul_rn(pos[3 * (size_t)ab.x + c], w0), __fmul_rn(pos[3 * (size_t)ab.y + c], w1));
    float m0 = msdf_sign * msdf[ab.x], m1 = msdf_sign * msdf[ab.y];
    msdf_vert[vid] = __fadd_rn(__fmul_rn(m0, w0), __fmul_rn(m1, w1));
    vert_edge[2 * (size_t)vid + 0] = ab.x;
    vert_edge[2 * (size_t)vid + 1] = ab.y;
}

// ---- pass C: watertight faces + polygon (mSDF) case per tet ------------------------------------------
__global__ __launch_bounds__(BLK) void mt_emit_faces_wt(const int* __restrict__ tet_edge /*[nt][6]*/, int nt,
                                                        uint8_t* __restrict__ tet_code, const int* __restrict__ blk_t,
                                                        const int* __restrict__ counts, const int* __restrict__ edge_vid,
                                                        const float* __restrict__ msdf_vert, int* __restrict__ faces_wt,
                                                        int64_t* __restrict__ faces_wt64, int* __restrict__ blk_t2 /*[nb][8]*/, Caps caps) {
    __shared__ int s_wave[4];
    if (over_caps(counts, caps)) return;
    const int t = blockIdx.x * BLK + threadIdx.x;
    const int n1 = counts[1];
    int code = (t < nt) ? (tet_code[t] & 15) : 0;
    int n = c_num_tri[code];
    int tot;
    int r1 = block_rank(n == 1, s_wave, tot);
    int r2 = block_rank(n == 2, s_wave, tot);
    int pcase = 0, ntaug = 0;
    if (n > 0) {
        int ev[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) ev[j] = edge_vid[tet_edge[6 * (size_t)t + j]];
        int nloop = (n == 1) ? 3 : 4;
        int f0 = (n == 1) ? (blk_t[blockIdx.x * 8 + 0] + r1) : (n1 + 2 * (blk_t[blockIdx.x * 8 + 1] + r2));
        for (int k = 0; k < 3 * n; ++k) {
            int v = ev[c_tri_table[code][k]];
            faces_wt[3 * (size_t)f0 + k] = v;
            faces_wt64[3 * (size_t)f0 + k] = v;
        }
        for (int k = 0; k < nloop; ++k) {
            int v = ev[c_loop_table[code][k]];
            pcase = (pcase << 1) | (msdf_vert[v] > 0.f ? 1 : 0);     // bits MSB-first, :401-404
        }
        ntaug = (n == 1) ? c_num_tri3[pcase] : c_num_tri4[pcase];
        tet_code[t] = (uint8_t)(code | (pcase << 4));
    }
    // six cut groups: (tri,1) (tri,2) (quad,1..4)
    int g = -1;
    if (n == 1 && ntaug > 0) g = ntaug - 1;
    if (n == 2 && ntaug > 0) g = 1 + ntaug;
    for (int c = 0; c < 6; ++c) {
        block_rank(g == c, s_wave, tot);
        if (threadIdx.x == 0) blk_t2[blockIdx.x * 8 + c] = tot;
    }
}

// ---- pass D: boundary vertices + cut faces --------------------------------------------------------------
__global__ __launch_bounds__(BLK) void mt_emit_aug(const int* __restrict__ tet_edge, int nt, const uint8_t* __restrict__ tet_code,
                                                   const int* __restrict__ blk_t, const int* __restrict__ blk_t2,
                                                   const int* __restrict__ counts, const int* __restrict__ edge_vid,
                                                   const float* __restrict__ verts_wt, const float* __restrict__ msdf_vert,
                                                   float* __restrict__ verts_aug, float* __restrict__ msdf_aug,
                                                   int* __restrict__ bnd_edge, int* __restrict__ faces_aug,
                                                   int64_t* __restrict__ faces_aug64, uint8_t* __restrict__ used, Caps caps) {
    __shared__ int s_wave[4];
    if (over_caps(counts, caps)) return;
    const int t = blockIdx.x * BLK + threadIdx.x;
    const int pwt = counts[0], n1 = counts[1];
    int goff[6];
    {
        int mult[6] = {1, 2, 1, 2, 3, 4};
        int acc = 0;
        for (int c = 0; c < 6; ++c) { goff[c] = acc; acc += counts[3 + c] * mult[c]; }
    }
    int full = (t < nt) ? tet_code[t] : 0;
    int code = full & 15, pcase = full >> 4;
    int n = c_num_tri[code];
    int tot;
    int r1 = block_rank(n == 1, s_wave, tot);
    int r2 = block_rank(n == 2, s_wave, tot);
    int ntaug = (n == 1) ? c_num_tri3[pcase] : ((n == 2) ? c_num_tri4[pcase] : 0);
    int g = -1;
    if (n == 1 && ntaug > 0) g = ntaug - 1;
    if (n == 2 && ntaug > 0) g = 1 + ntaug;
    int grank = 0;
    for (int c = 0; c < 6; ++c) {
        int r = block_rank(g == c, s_wave, tot);
        if (g == c) grank = blk_t2[blockIdx.x * 8 + c] + r;
    }
    if (n == 0) return;
    int ev[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) ev[j] = edge_vid[tet_edge[6 * (size_t)t + j]];
    const int nloop = (n == 1) ? 3 : 4;
    const int bbase = (n == 1) ? (pwt + 3 * (blk_t[blockIdx.x * 8 + 0] + r1)) : (pwt + 3 * n1 + 4 * (blk_t[blockIdx.x * 8 + 1] + r2));
    int map[8];
    for (int k = 0; k < nloop; ++k) {
        map[k] = ev[c_loop_table[code][k]];
        map[nloop + k] = bbase + k;
    }
    // boundary vertex k lies on polygon edge (map[k], map[(k+1)%nloop])  (:342-378)
    for (int k = 0; k < nloop; ++k) {
        int a = map[k], b = map[(k + 1 == nloop) ? 0 : k + 1];
        float ma = msdf_vert[a], mb = msdf_vert[b];
        float sa = (ma > 0.f) ? 1.f : ((ma < 0.f) ? -1.f : 0.f);
        float sb = (mb > 0.f) ? 1.f : ((mb < 0.f) ? -1.f : 0.f);
        bool ok = fabsf(sa + sb) != 2.f;
        float A = ma, B = -mb;
        float den = __fadd_rn(A, B);
        ok = ok && (fabsf(den) > 1e-12f);
        float w0 = ok ? (B / den) : 0.f, w1 = ok ? (A / den) : 0.f;
        size_t bi = (size_t)(bbase + k);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            verts_aug[3 * bi + c] = __fadd_rn(__fmul_rn(verts_wt[3 * (size_t)a + c], w0), __fmul_rn(verts_wt[3 * (size_t)b + c], w1));
        msdf_aug[bi] = __fadd_rn(__fmul_rn(ma, w0), __fmul_rn(mb, w1));
        bnd_edge[2 * (bi - pwt) + 0] = a;
        bnd_edge[2 * (bi - pwt) + 1] = b;
    }
    if (g < 0) return;
    int f0 = goff[g] + grank * ntaug;
    for (int k = 0; k < 3 * ntaug; ++k) {
        int slot = (n == 1) ? c_tri3_table[pcase][k] : c_tri4_table[pcase][k];
        int v = map[slot];
        faces_aug[3 * (size_t)f0 + k] = v;
        faces_aug64[3 * (size_t)f0 + k] = v;
        used[v] = 1;
    }
}

// ---- pass E: assemble verts_aug[0:P_wt] / msdf_aug[0:P_wt], zero unreferenced vertices (:423-427) ------------
__global__ void mt_finalize(const float* __restrict__ verts_wt, const float* __restrict__ msdf_vert, const uint8_t* __restrict__ used,
                            int pwt, int p, float* __restrict__ verts_aug, float* __restrict__ msdf_aug) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    bool u = used[i] != 0;
    if (i < pwt) {
        msdf_aug[i] = msdf_vert[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) verts_aug[3 * (size_t)i + c] = u ? verts_wt[3 * (size_t)i + c] : 0.f;
    } else if (!u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) verts_aug[3 * (size_t)i + c] = 0.f;
    }
}

// the same with the sizes read on the device (speculative extraction: the launch covers the capacity)
__global__ void mt_finalize_dev(const float* __restrict__ verts_wt, const float* __restrict__ msdf_vert, const uint8_t* __restrict__ used,
                                const int* __restrict__ counts, Caps caps, float* __restrict__ verts_aug, float* __restrict__ msdf_aug) {
    if (over_caps(counts, caps)) return;
    const int pwt = counts[0], p = pwt + 3 * counts[1] + 4 * counts[2];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    bool u = used[i] != 0;
    if (i < pwt) {
        msdf_aug[i] = msdf_vert[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) verts_aug[3 * (size_t)i + c] = u ? verts_wt[3 * (size_t)i + c] : 0.f;
    } else if (!u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) verts_aug[3 * (size_t)i + c] = 0.f;
    }
}

// ---- sizes to the host without a stream synchronisation ----------------------------------------------------------------------------------
// The host needs three integers (and later six more) of every extraction.  A read-back through the stream (hipMemcpy + synchronise) returns
// only when EVERYTHING queued on that stream before it is done, and costs ~20 us of wake-up latency; a copy on a second stream behind an event
// added two cross-queue hops (measured: the wake-up came 85 us later, profiles/r5_host_window_spec.txt).  Instead a one-thread kernel writes the
// values into coherent host memory (hipHostMallocCoherent) followed by a sequence number with system-scope release; the host spins on the
// sequence number (d3h_host_flag_wait, GIL released) -- kernels queued behind the publisher keep the GPU busy meanwhile.
__global__ void mt_publish(const int* __restrict__ src, int n, int* __restrict__ host, int seq_slot, int seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < n; ++k) host[k] = src[k];
        __threadfence_system();
#ifdef D3H_EMULATED
        host[seq_slot] = seq;
#else
        __hip_atomic_store(host + seq_slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    }
}

// ---- VJP ------------------------------------------------------------------------------------------------
// stage 1: boundary vertices -> per-watertight-vertex accumulators G_v[P_wt][3], G_m[P_wt] (differentiable msdf_vert),
//          G_sg[P_wt] (stop-gradient copy, :301-303,:388-389)
__global__ void mt_bwd_boundary(const float* __restrict__ g_verts_aug, const float* __restrict__ g_msdf_aug,
                                const uint8_t* __restrict__ used, const int* __restrict__ bnd_edge, const float* __restrict__ verts_wt,
                                const float* __restrict__ msdf_vert, int pwt, int nb, float* __restrict__ G_v,
                                float* __restrict__ G_m, float* __restrict__ G_sg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    int a = bnd_edge[2 * (size_t)i], b = bnd_edge[2 * (size_t)i + 1];
    size_t bi = (size_t)pwt + i;
    float ma = msdf_vert[a], mb = msdf_vert[b];
    float sa = (ma > 0.f) ? 1.f : ((ma < 0.f) ? -1.f : 0.f);
    float sb = (mb > 0.f) ? 1.f : ((mb < 0.f) ? -1.f : 0.f);
    float A = ma, B = -mb, den = A + B;
    bool ok = (fabsf(sa + sb) != 2.f) && (fabsf(den) > 1e-12f);
    if (!ok) return;   // weights are exactly zero: no gradient reaches anything
    float w0 = B / den, w1 = A / den;
    float gm = g_msdf_aug ? g_msdf_aug[bi] : 0.f;
    if (gm != 0.f) {
        atomicAdd(&G_sg[a], gm * w0);
        atomicAdd(&G_sg[b], gm * w1);
    }
    if (!g_verts_aug || !used[bi]) return;
    float gw0 = 0.f, gw1 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float g = g_verts_aug[3 * bi + c];
        atomicAdd(&G_v[3 * (size_t)a + c], g * w0);
        atomicAdd(&G_v[3 * (size_t)b + c], g * w1);
        gw0 = fmaf(g, verts_wt[3 * (size_t)a + c], gw0);
        gw1 = fmaf(g, verts_wt[3 * (size_t)b + c], gw1);
    }
    float inv2 = 1.f / (den * den);
    float gA = (gw1 - gw0) * B * inv2;     // d w0/dA = -B/den^2, d w1/dA = 1/den - A/den^2 = B/den^2
    float gB = (gw0 - gw1) * A * inv2;     // d w0/dB = A/den^2,  d w1/dB = -A/den^2
    atomicAdd(&G_m[a], gA);
    atomicAdd(&G_m[b], -gB);
}

// stage 2: watertight vertices -> grid vertices (pos, sdf, msdf)
__global__ void mt_bwd_verts(const float* __restrict__ g_verts_aug, const float* __restrict__ g_msdf_aug,
                             const float* __restrict__ g_verts_wt, const uint8_t* __restrict__ used, const int* __restrict__ vert_edge,
                             const float* __restrict__ pos, const float* __restrict__ sdf, const float* __restrict__ msdf,
                             float msdf_sign, const float* __restrict__ G_v, const float* __restrict__ G_m, const float* __restrict__ G_sg,
                             int pwt, float* __restrict__ d_pos, float* __restrict__ d_sdf, float* __restrict__ d_msdf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pwt) return;
    int n0 = vert_edge[2 * (size_t)i], n1 = vert_edge[2 * (size_t)i + 1];
    float s0 = sdf[n0], s1 = sdf[n1];
    float a = s0, b = -s1;
    float den = a + b;
    float sg = (den > 0.f) ? 1.f : ((den < 0.f) ? -1.f : 0.f);
    den = sg * (fabsf(den) + 1e-12f);
    if (den == 0.f) den = 1e-12f;
    float w0 = b / den, w1 = a / den;
    float gv[3];
    bool u = used[i] != 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float g = G_v[3 * (size_t)i + c];
        if (g_verts_aug && u) g += g_verts_aug[3 * (size_t)i + c];
        if (g_verts_wt) g += g_verts_wt[3 * (size_t)i + c];
        gv[c] = g;
    }
    float gm = G_m[i];
    float gsg = G_sg[i] + (g_msdf_aug ? g_msdf_aug[i] : 0.f);
    float m0 = msdf_sign * msdf[n0], m1 = msdf_sign * msdf[n1];
    float gw0 = gm * m0, gw1 = gm * m1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float p0 = pos[3 * (size_t)n0 + c], p1 = pos[3 * (size_t)n1 + c];
        if (gv[c] != 0.f) {
            atomicAdd(&d_pos[3 * (size_t)n0 + c], gv[c] * w0);
            atomicAdd(&d_pos[3 * (size_t)n1 + c], gv[c] * w1);
        }
        gw0 = fmaf(gv[c], p0, gw0);
        gw1 = fmaf(gv[c], p1, gw1);
    }
    if (d_msdf) {   // hmsdf_tets_split.py:261-264: the "body" pass negates msdf under no_grad -> no gradient
        float g = gm + gsg;
        if (g != 0.f) {
            atomicAdd(&d_msdf[n0], g * w0);
            atomicAdd(&d_msdf[n1], g * w1);
        }
    }
    float inv = 1.f / den, inv2 = inv * inv;
    float ga = gw0 * (-b * inv2) + gw1 * (inv - a * inv2);
    float gb = gw0 * (inv - b * inv2) + gw1 * (-a * inv2);
    if (ga != 0.f) atomicAdd(&d_sdf[n0], ga);
    if (gb != 0.f) atomicAdd(&d_sdf[n1], -gb);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline int nblk(int n) { return n > 0 ? (n + 255) / 256 : 1; }

// counts[0] = P_wt, counts[1] = #1-triangle tets, counts[2] = #2-triangle tets  (after this call)
extern "C" int d3h_mtets_count(const float* sdf, const int* tets, int nt, const int* edges, int ne, uint8_t* tet_code,
                               int* blk_e, int* blk_t, int* counts, void* stream) {
    if (!sdf || !tets || !edges || !tet_code || !blk_e || !blk_t || !counts || nt < 0 || ne < 0) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int kt_ = d3h_ktime_begin(D3H_KT_MTETS_COUNT, (long long)(nt), (hipStream_t)(stream));
    hipLaunchKernelGGL(mt_count_edges, dim3(nblk(ne)), dim3(256), 0, s, sdf, edges, ne, blk_e);
    hipLaunchKernelGGL(mt_count_tets, dim3(nblk(nt)), dim3(256), 0, s, sdf, tets, nt, tet_code, blk_t);
    hipLaunchKernelGGL(mt_scan, dim3(1), dim3(1024), 0, s, blk_e, nblk(ne), 1, 0, 1, counts, 0);
    hipLaunchKernelGGL(mt_scan, dim3(1), dim3(1024), 0, s, blk_t, nblk(nt), 8, 0, 2, counts, 1);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// counts[3..8] = number of polygons in each of the six cut groups (after this call)
extern "C" int d3h_mtets_emit_wt(const float* pos, const float* sdf, const float* msdf, float msdf_sign, const int* edges, int ne,
                                 const int* tet_edge, int nt, uint8_t* tet_code, const int* blk_e, const int* blk_t, int* blk_t2,
                                 int* counts, int* edge_vid, float* verts_wt, float* msdf_vert, int* vert_edge, int* faces_wt,
                                 int64_t* faces_wt64, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int kt_ = d3h_ktime_begin(D3H_KT_MTETS_EMIT, (long long)(nt), (hipStream_t)(stream));
    hipLaunchKernelGGL(mt_emit_verts, dim3(nblk(ne)), dim3(256), 0, s, pos, sdf, msdf, msdf_sign, edges, ne, blk_e, edge_vid, verts_wt,
                       msdf_vert, vert_edge, counts, NO_CAPS);
    hipLaunchKernelGGL(mt_emit_faces_wt, dim3(nblk(nt)), dim3(256), 0, s, tet_edge, nt, tet_code, blk_t, counts, edge_vid, msdf_vert,
                       faces_wt, faces_wt64, blk_t2, NO_CAPS);
    hipLaunchKernelGGL(mt_scan, dim3(1), dim3(1024), 0, s, blk_t2, nblk(nt), 8, 0, 6, counts, 3);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

extern "C" int d3h_mtets_emit_aug(const int* tet_edge, int nt, const uint8_t* tet_code, const int* blk_t, const int* blk_t2,
                                  const int* counts, const int* edge_vid, const float* verts_wt, const float* msdf_vert, int pwt, int p,
                                  float* verts_aug, float* msdf_aug, int* bnd_edge, int* faces_aug, int64_t* faces_aug64,
                                  uint8_t* used, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (p > 0) (void)hipMemsetAsync(used, 0, (size_t)p, s);
    hipLaunchKernelGGL(mt_emit_aug, dim3(nblk(nt)), dim3(256), 0, s, tet_edge, nt, tet_code, blk_t, blk_t2, counts, edge_vid, verts_wt,
                       msdf_vert, verts_aug, msdf_aug, bnd_edge, faces_aug, faces_aug64, used, NO_CAPS);
    if (p > 0)
        hipLaunchKernelGGL(mt_finalize, dim3(nblk(p)), dim3(256), 0, s, verts_wt, msdf_vert, used, pwt, p, verts_aug, msdf_aug);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// d3h_mtets_emit_wt + d3h_mtets_emit_aug queued BEFORE the host has read counts[0..2] (the sizes of this extraction): every output is
// allocated by the caller at the capacities cap_pwt / cap_n1 / cap_n2 -- verts_wt, msdf_vert, vert_edge: cap_pwt rows; faces_wt(64):
// cap_n1 + 2 cap_n2; verts_aug, msdf_aug, used: cap_pwt + 3 cap_n1 + 4 cap_n2; bnd_edge: 3 cap_n1 + 4 cap_n2; faces_aug(64): 2 cap_n1 +
// 4 cap_n2, faces_aug64 zero-filled -- and holds the exact-size results in its leading rows.  When a count exceeds its capacity NOTHING is
// written (counts[3..8] are then meaningless): the caller, who reads counts[0..2] anyway, repeats the extraction through the exact-size calls.
// host_flags (d3h_host_flags_alloc, or NULL): counts[3..8] are published to host_flags[8..13] with the sequence number `seq` in host_flags[15].
extern "C" int d3h_mtets_emit_spec(const float* pos, const float* sdf, const float* msdf, float msdf_sign, const int* edges, int ne,
                                   const int* tet_edge, int nt, uint8_t* tet_code, const int* blk_e, const int* blk_t, int* blk_t2,
                                   int* counts, int* edge_vid, int cap_pwt, int cap_n1, int cap_n2, float* verts_wt, float* msdf_vert,
                                   int* vert_edge, int* faces_wt, int64_t* faces_wt64, float* verts_aug, float* msdf_aug, int* bnd_edge,
                                   int* faces_aug, int64_t* faces_aug64, uint8_t* used, int* host_flags, int seq, void* stream) {
    if (cap_pwt < 0 || cap_n1 < 0 || cap_n2 < 0) return D3H_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const Caps caps = {cap_pwt, cap_n1, cap_n2};
    const long long cap_p = (long long)cap_pwt + 3LL * cap_n1 + 4LL * cap_n2;
    if (cap_p > 0x7fffffffLL) return D3H_ERR_ARG;
    const int kt_ = d3h_ktime_begin(D3H_KT_MTETS_EMIT, (long long)(nt), (hipStream_t)(stream));
    hipLaunchKernelGGL(mt_emit_verts, dim3(nblk(ne)), dim3(256), 0, s, pos, sdf, msdf, msdf_sign, edges, ne, blk_e, edge_vid, verts_wt,
                       msdf_vert, vert_edge, counts, caps);
    hipLaunchKernelGGL(mt_emit_faces_wt, dim3(nblk(nt)), dim3(256), 0, s, tet_edge, nt, tet_code, blk_t, counts, edge_vid, msdf_vert,
                       faces_wt, faces_wt64, blk_t2, caps);
    hipLaunchKernelGGL(mt_scan, dim3(1), dim3(1024), 0, s, blk_t2, nblk(nt), 8, 0, 6, counts, 3);
    if (host_flags) hipLaunchKernelGGL(mt_publish, dim3(1), dim3(64), 0, s, (const int*)counts + 3, 6, host_flags + 8, 7, seq);
    d3h_ktime_end(kt_, (hipStream_t)(stream));
    if (cap_p > 0) (void)hipMemsetAsync(used, 0, (size_t)cap_p, s);
    hipLaunchKernelGGL(mt_emit_aug, dim3(nblk(nt)), dim3(256), 0, s, tet_edge, nt, tet_code, blk_t, blk_t2, counts, edge_vid, verts_wt,
                       msdf_vert, verts_aug, msdf_aug, bnd_edge, faces_aug, faces_aug64, used, caps);
    if (cap_p > 0)
        hipLaunchKernelGGL(mt_finalize_dev, dim3(nblk((int)cap_p)), dim3(256), 0, s, verts_wt, msdf_vert, used, (const int*)counts, caps, verts_aug,
                           msdf_aug);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// d_pos/d_sdf/d_msdf must be zero-initialised by the caller (they are accumulated into); scratch: (5*pwt) floats, zeroed here
extern "C" int d3h_mtets_bwd(const float* g_verts_aug, const float* g_msdf_aug, const float* g_verts_wt, const uint8_t* used,
                             const int* bnd_edge, const int* vert_edge, const float* verts_wt, const float* msdf_vert, const float* pos,
                             const float* sdf, const float* msdf, float msdf_sign, int pwt, int p, float* scratch, float* d_pos,
                             float* d_sdf, float* d_msdf, void* stream) {
    if (pwt <= 0) return D3H_OK;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(scratch, 0, sizeof(float) * 5 * (size_t)pwt, s);
    float* G_v = scratch;
    float* G_m = scratch + 3 * (size_t)pwt;
    float* G_sg = scratch + 4 * (size_t)pwt;
    int nb = p - pwt;
    if (nb > 0)
        hipLaunchKernelGGL(mt_bwd_boundary, dim3(nblk(nb)), dim3(256), 0, s, g_verts_aug, g_msdf_aug, used, bnd_edge, verts_wt,
                           msdf_vert, pwt, nb, G_v, G_m, G_sg);
    hipLaunchKernelGGL(mt_bwd_verts, dim3(nblk(pwt)), dim3(256), 0, s, g_verts_aug, g_msdf_aug, g_verts_wt, used, vert_edge, pos, sdf,
                       msdf, msdf_sign, G_v, G_m, G_sg, pwt, d_pos, d_sdf, d_msdf);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}

// 64 ints of coherent, device-visible host memory (never freed: one per tet grid) for d3h_mtets_publish_sizes / d3h_host_flag_wait
extern "C" int d3h_host_flags_alloc(int** out) {
    if (!out) return D3H_ERR_ARG;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, 64 * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess || !p) return e != hipSuccess ? (int)e : D3H_ERR_ARG;
    memset(p, 0, 64 * sizeof(int));
    *out = (int*)p;
    return D3H_OK;
}
// queued behind d3h_mtets_count on the same stream: counts[0..2] -> host_flags[0..2], then `seq` -> host_flags[7]
extern "C" int d3h_mtets_publish_sizes(const int* counts, int* host_flags, int seq, void* stream) {
    if (!counts || !host_flags) return D3H_ERR_ARG;
    hipLaunchKernelGGL(mt_publish, dim3(1), dim3(64), 0, (hipStream_t)stream, counts, 3, host_flags, 7, seq);
    D3H_LAUNCH_CHECK();
    return D3H_OK;
}
// spin until host_flags[slot] == seq (the values published with it are then visible); D3H_ERR_ARG after timeout_ms
extern "C" int d3h_host_flag_wait(const int* host_flags, int slot, int seq, int timeout_ms) {
    if (!host_flags || slot < 0 || slot >= 64) return D3H_ERR_ARG;
    const volatile int* f = host_flags + slot;
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spin = 0;; ++spin) {
        if (__atomic_load_n((const int*)f, __ATOMIC_ACQUIRE) == seq) return D3H_OK;
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return D3H_ERR_ARG;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
