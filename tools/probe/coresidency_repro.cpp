// Torch-free reproducer for the "x3 co-residency hazard" (DESIGN.md section 3): does a kernel of another stream return
// different bits while sdf_mlp_bwd_dw_layers_x3_kernel (the -DD3H_DWX_SHARE_SIMDS build: 180 VGPRs, so foreign waves share its SIMDs) runs?
//
//   stream A: the eikonal second-order chain of the library (d3h_sdf_mlp_eik_bwd: tangent sweep, injected reverse sweep, the dual-source
//             bf16 weight-gradient kernel, the f32 embedding kernels), in a loop, on fixed inputs
//   stream B: (1) d3h_lbs_bwd -- the kernel the hazard was seen in -- on fixed inputs, into a ring of output buffers;
//             (2) two canaries compiled here: a register-heavy pure VALU function and an LDS round trip, likewise
//   after every round: bitwise comparison of every stream-B output with the outputs of the same launches made on an idle GPU.
//
// No torch, no caching allocator, no autograd: every buffer is hipMalloc'ed once and lives to the end.  If this reproduces, the cause is
// in the kernels / the hardware; if it does not (and the Python two-tick script does), it is in lifetimes / ordering on the host side.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe/coresidency_repro.cpp -o tools/probe/coresidency_repro -ldl
//   tools/probe/coresidency_repro d3human-code_amd/d3h/libd3h_share.so [rounds=40] [n_samples=50000] [np=700] [nb=2] [max_cus=0]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define D3(x) do { int e_ = (x); if (e_ != 0) { printf("d3h error %d at %s:%d\n", e_, __FILE__, __LINE__); exit(3); } } while (0)

typedef int64_t (*i64fn)(void);
typedef int64_t (*actfn)(int64_t);
typedef int (*packfn)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, void*);
typedef int (*pack3fn)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, unsigned*, void*);
typedef int (*packtfn)(const float*, const float*, const float*, float*, void*);
typedef int (*packt3fn)(const float*, const float*, const float*, unsigned*, void*);
typedef int (*fwdx3fn)(const float*, const float*, float, const unsigned*, float*, float*, float*, int64_t, int, void*);
typedef int (*gradxfn)(const float*, const float*, const float*, const unsigned*, int, const float*, float*, int64_t, float*, int, void*);
typedef int (*eiklossfn)(const float*, int64_t, float, float*, float*, void*);
typedef int (*eikbwdfn)(const float*, const float*, const float*, const float*, const unsigned*, int, const unsigned*, int, float, const float*, const float*, float*, float*,
                        int64_t, float*, float*, float*, float*, float*, float*, float*, int, void*);
typedef int (*lbsbwdfn)(const float*, int, const int*, const float*, int, const float*, const float*, int, const float*, float*, float*, float*, float*, void*);

// ---- canaries ---------------------------------------------------------------------------------------------------------------------------
// (1) a pure function of the thread's input with ~100 live VGPRs: any wave whose registers are disturbed returns different bits
__global__ __launch_bounds__(256) void canary_valu(const float* __restrict__ in, float* __restrict__ out, int n, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r[96];
#pragma unroll
    for (int k = 0; k < 96; ++k) r[k] = in[(i + 977 * k) % n];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 96; ++k) r[k] = fmaf(r[k], 0.999f, r[(k + 37) % 96] * 1.0e-3f);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 96; ++k) s += r[k] * (float)(k + 1);
    out[i] = s;
}
// (2) an LDS round trip: every thread writes 16 words, barriers, reads its neighbours' words
__global__ __launch_bounds__(256) void canary_lds(const float* __restrict__ in, float* __restrict__ out, int n, int iters) {
    __shared__ float T[256 * 17];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < 16; ++k) T[threadIdx.x * 17 + k] = in[(i + 131 * k + it) % n];
        __syncthreads();
        for (int k = 0; k < 16; ++k) acc += T[((threadIdx.x + 1 + k) & 255) * 17 + k] * (float)(k + 1);
        __syncthreads();
    }
    if (i < n) out[i] = acc;
}


// ---- diagnostic victims (restatements of pieces of csrc/lbs.hip:lbs_bwd_kernel): which VALUE goes wrong, in which lanes -----------------
__device__ __forceinline__ void v_blend(const float* __restrict__ w, const float* __restrict__ A, int nj, float (&M)[12], float& s) {
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
// (a) the blended frame matrix M (12) + s + the three upstream-gradient loads: 16 floats per (vertex, frame)
__global__ __launch_bounds__(256) void victim_blend_dump(int np, const int* __restrict__ idx, const float* __restrict__ lbs_w, int nj,
                                                         const float* __restrict__ A, const float* __restrict__ gout, float* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= np) return;
    const float* w = lbs_w + (size_t)idx[p] * nj;
    float M[12], s;
    v_blend(w, A + (size_t)b * nj * 16, nj, M, s);
    const float* g = gout + ((size_t)b * np + p) * 3;
    float* o = out + ((size_t)b * np + p) * 16;
#pragma unroll
    for (int e = 0; e < 12; ++e) o[e] = M[e];
    o[12] = s; o[13] = g[0]; o[14] = g[1]; o[15] = g[2];
}
// (b) the same WITHOUT the divergent skip (every joint multiplied, zero weights included): uniform control flow
__global__ __launch_bounds__(256) void victim_blend_uniform(int np, const int* __restrict__ idx, const float* __restrict__ lbs_w, int nj,
                                                            const float* __restrict__ A, float* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= np) return;
    const float* w = lbs_w + (size_t)idx[p] * nj;
    const float* Ab = A + (size_t)b * nj * 16;
    float M[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) M[e] = 0.f;
    for (int j = 0; j < nj; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int e = 0; e < 12; ++e) M[e] = fmaf(wj, Ab[16 * j + e], M[e]);
    }
    float* o = out + ((size_t)b * np + p) * 12;
#pragma unroll
    for (int e = 0; e < 12; ++e) o[e] = M[e];
}
// (c) plain gather-copy of three floats per thread (the load path alone)
__global__ __launch_bounds__(256) void victim_copy3(int n, const float* __restrict__ in, float* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    out[3 * (size_t)p] = in[3 * (size_t)p]; out[3 * (size_t)p + 1] = in[3 * (size_t)p + 1]; out[3 * (size_t)p + 2] = in[3 * (size_t)p + 2];
}
// (d) IEEE division / reciprocal chain (inv3 of lbs.hip: v_div_scale / v_rcp / v_div_fmas / v_div_fixup)
__global__ __launch_bounds__(256) void canary_div(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = in[i] + 1.5f, acc = 0.f;
    for (int k = 0; k < 64; ++k) { a = 1.0f / (a + 0.25f) + 1.0f; acc += a / (float)(k + 3); }
    out[i] = acc;
}
// (e) LDS float atomics + barrier (the d_trans / dA accumulation pattern of lbs_bwd)
__global__ __launch_bounds__(256) void canary_lds_atomic(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float S[64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 64) S[threadIdx.x] = 0.f;
    __syncthreads();
    atomicAdd(&S[threadIdx.x & 63], (float)(threadIdx.x >> 6) + 1.0f);       // exact small integers: order-independent
    atomicAdd(&S[(threadIdx.x * 7) & 63], 2.0f);
    __syncthreads();
    if (i < n) out[i] = S[threadIdx.x & 63] + in[i];
}

// (f) packed-f32 VALU chains (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: what -O3 SLP-packs the 3x3 algebra of lbs_bwd into) and
//     the same arithmetic pinned to the scalar forms with inline asm
typedef float f2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void canary_pk(const float* __restrict__ in, float* __restrict__ out, int n, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f2v a = {in[i], in[(i + 1) % n]}, b = {in[(i + 7) % n] + 2.f, in[(i + 9) % n] + 2.f}, c = {0.25f, -0.5f};
    for (int k = 0; k < iters; ++k) {
        if (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(c));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
        if (OP == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(a) : "v"(a), "v"(b));
        if (OP == 4) { float x = a.x, y = a.y; asm volatile("v_mul_f32 %0, %1, %2\n\tv_mul_f32 %3, %4, %5" : "=&v"(x), "+v"(x), "+v"(b.x), "=&v"(y), "+v"(y), "+v"(b.y)); a.x = x; a.y = y; }
        // keep the magnitudes bounded without leaving the packed domain's inputs constant
        if ((k & 7) == 7) { a.x = a.x * 0.001f + in[(i + k) % n]; a.y = a.y * 0.001f + in[(i + 2 * k) % n]; }
    }
    out[i] = a.x + 3.f * a.y;
}

// (g) lbs_bwd_kernel of csrc/lbs.hip restated with compile-time knobs: which ingredient makes it vulnerable?
//     F & 1: the LDS / global atomics of dA and d_trans are compiled in (never executed: dA = d_trans = NULL)
//     F & 2: the LDS zero fill + the two workgroup barriers
//     F & 4: M0 blend + to_canonical (3x3 inverse with IEEE divisions); else Rinv = identity
//     F & 8: the frame blend M; else M = identity
__device__ __forceinline__ void v_inv3(const float (&M)[12], float (&R)[9]) {
    float a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    float c0 = e * i - f * h, c1 = f * g - d * i, c2 = d * h - e * g;
    float det = a * c0 + b * c1 + c * c2;
    float id = 1.0f / det;
    R[0] = c0 * id; R[1] = (c * h - b * i) * id; R[2] = (b * f - c * e) * id;
    R[3] = c1 * id; R[4] = (a * i - c * g) * id; R[5] = (c * d - a * f) * id;
    R[6] = c2 * id; R[7] = (b * g - a * h) * id; R[8] = (a * e - b * d) * id;
}
template <int F>
__global__ __launch_bounds__(256) void victim_lbs(const float* __restrict__ pts, int np, const int* __restrict__ idx, const float* __restrict__ lbs_w, int nj,
                                                  const float* __restrict__ A0, const float* __restrict__ A, int nb, const float* __restrict__ gout,
                                                  float* __restrict__ d_pts, float* __restrict__ dA, float* __restrict__ d_trans) {
    __shared__ float sA[64 * 12];
    __shared__ float sT[3];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool valid = p < np;
    const float* w = lbs_w + (size_t)(valid ? idx[p] : 0) * nj;
    float M0[12], s0, Rinv[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, pc[3] = {0.f, 0.f, 0.f};
    if (valid && (F & 4)) {
        v_blend(w, A0, nj, M0, s0);
        v_inv3(M0, Rinv);
        float is = 1.0f / s0;
        float qx = pts[3 * (size_t)p] - M0[3] * is, qy = pts[3 * (size_t)p + 1] - M0[7] * is, qz = pts[3 * (size_t)p + 2] - M0[11] * is;
        pc[0] = Rinv[0] * qx + Rinv[1] * qy + Rinv[2] * qz;
        pc[1] = Rinv[3] * qx + Rinv[4] * qy + Rinv[5] * qz;
        pc[2] = Rinv[6] * qx + Rinv[7] * qy + Rinv[8] * qz;
    }
    float gpc[3] = {0.f, 0.f, 0.f};
    if (F & 2) {
        if ((F & 1) && dA) for (int i = threadIdx.x; i < nj * 12; i += 256) sA[i] = 0.f;
        if (threadIdx.x < 3) sT[threadIdx.x] = 0.f;
        __syncthreads();
    }
    if (valid) {
        float M[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f}, s = 1.f;
        if (F & 8) v_blend(w, A + (size_t)b * nj * 16, nj, M, s);
        const float* g = gout + ((size_t)b * np + p) * 3;
        float g0 = g[0], g1 = g[1], g2 = g[2];
        gpc[0] = M[0] * g0 + M[4] * g1 + M[8] * g2;
        gpc[1] = M[1] * g0 + M[5] * g1 + M[9] * g2;
        gpc[2] = M[2] * g0 + M[6] * g1 + M[10] * g2;
        if (F & 1) {
            if (d_trans) { atomicAdd(&sT[0], g0); atomicAdd(&sT[1], g1); atomicAdd(&sT[2], g2); }
            if (dA) {
                float dM[12] = {g0 * pc[0], g0 * pc[1], g0 * pc[2], g0, g1 * pc[0], g1 * pc[1], g1 * pc[2], g1, g2 * pc[0], g2 * pc[1], g2 * pc[2], g2};
                for (int j = 0; j < nj; ++j) {
                    float wj = w[j];
                    if (wj != 0.f) {
#pragma unroll
                        for (int e = 0; e < 12; ++e) atomicAdd(&sA[j * 12 + e], wj * dM[e]);
                    }
                }
            }
        }
    }
    if (F & 2) __syncthreads();
    if (F & 1) {
        if (dA) for (int i = threadIdx.x; i < nj * 12; i += 256) { float v = sA[i]; if (v != 0.f) atomicAdd(&dA[((size_t)b * nj + i / 12) * 16 + (i % 12)], v); }
        if (d_trans && threadIdx.x < 3) atomicAdd(&d_trans[3 * b + threadIdx.x], sT[threadIdx.x]);
    }
    if (valid && d_pts) {
        float* o = d_pts + ((size_t)b * np + p) * 3;
        o[0] = Rinv[0] * gpc[0] + Rinv[3] * gpc[1] + Rinv[6] * gpc[2] + ((F & 4) ? 0.f : pc[0]);
        o[1] = Rinv[1] * gpc[0] + Rinv[4] * gpc[1] + Rinv[7] * gpc[2];
        o[2] = Rinv[2] * gpc[0] + Rinv[5] * gpc[1] + Rinv[8] * gpc[2];
    }
}

static float frand() { return rand() / (float)RAND_MAX; }
template <class T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
template <class T> static T* dupload(const std::vector<T>& h) { T* p = dalloc<T>(h.size()); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s <libd3h_*.so> [rounds] [n_samples] [np] [nb] [max_cus] [victims_per_round]\n", argv[0]); return 1; }
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    const int64_t n = argc > 3 ? atoll(argv[3]) : 50000;
    const int np = argc > 4 ? atoi(argv[4]) : 700;
    const int nb = argc > 5 ? atoi(argv[5]) : 2;
    const int max_cus = argc > 6 ? atoi(argv[6]) : 0;
    const int RING = argc > 7 ? atoi(argv[7]) : 48;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
#define SYM(T, name) T name = (T)dlsym(h, "d3h_" #name); if (!name) { printf("missing d3h_" #name "\n"); return 1; }
    SYM(i64fn, sdf_mlp_wpack_floats) SYM(i64fn, sdf_mlp_wpack3_dwords) SYM(i64fn, sdf_mlp_wpackt_floats) SYM(i64fn, sdf_mlp_wpackt3_dwords)
    SYM(actfn, sdf_mlp_act_floats) SYM(packfn, sdf_mlp_pack) SYM(pack3fn, sdf_mlp_pack3) SYM(packtfn, sdf_mlp_pack_t) SYM(packt3fn, sdf_mlp_pack_t3)
    SYM(fwdx3fn, sdf_mlp_fwd_x3) SYM(gradxfn, sdf_mlp_grad_x) SYM(eiklossfn, eikonal_loss) SYM(eikbwdfn, sdf_mlp_eik_bwd) SYM(lbsbwdfn, lbs_bwd)
    srand(7);
    // ---- the network: geometric init-like magnitudes (what the chain sees in training: O(1) activations, non-trivial bf16 planes) ----
    auto rnd = [](size_t k, float s) { std::vector<float> v(k); for (auto& x : v) x = (frand() - 0.5f) * 2.f * s; return v; };
    float *w0 = dupload(rnd(256 * 39, 0.3f)), *b0 = dupload(rnd(256, 0.1f)), *wh = dupload(rnd(5 * 65536, 0.09f)), *bh = dupload(rnd(5 * 256, 0.1f));
    float *w4 = dupload(rnd(256 * 295, 0.09f)), *b4 = dupload(rnd(256, 0.1f)), *w7 = dupload(rnd(256, 0.1f)), *b7 = dupload(rnd(1, 0.1f));
    float* wp = dalloc<float>(sdf_mlp_wpack_floats());
    float* wpt = dalloc<float>(sdf_mlp_wpackt_floats());
    unsigned* wp3 = dalloc<unsigned>(sdf_mlp_wpack3_dwords());
    unsigned* wpt3 = dalloc<unsigned>(sdf_mlp_wpackt3_dwords());
    D3(sdf_mlp_pack(w0, b0, wh, bh, w4, b4, w7, b7, wp, nullptr));
    D3(sdf_mlp_pack3(w0, b0, wh, bh, w4, b4, w7, b7, wp3, nullptr));
    D3(sdf_mlp_pack_t(w0, wh, w4, wpt, nullptr));
    D3(sdf_mlp_pack_t3(w0, wh, w4, wpt3, nullptr));
    const int64_t na = sdf_mlp_act_floats(n);
    float *x = dupload(rnd(3 * n, 0.5f)), *sdf = dalloc<float>(n), *act = dalloc<float>(na), *dz = dalloc<float>(na), *tb = dalloc<float>(na), *eb = dalloc<float>(na);
    float *g = dalloc<float>(3 * n), *u = dalloc<float>(3 * n), *lsum = dalloc<float>(1);
    D3(sdf_mlp_fwd_x3(x, nullptr, 0.f, wp3, sdf, nullptr, act, n, 0, nullptr));
    D3(sdf_mlp_grad_x(x, w7, wpt, wpt3, 3, act, dz, n, g, 0, nullptr));          // (3: the bf16 x 3 transposed pack -- the aggressor of the hazard)
    D3(eikonal_loss(g, n, 0.3f / n, lsum, u, nullptr));
    const size_t ARENA = 256 * 39 + 256 + 5 * 65536 + 5 * 256 + 256 * 295 + 256 + 256 + 64;
    float* arena = dalloc<float>(ARENA);
    float *dw0 = arena, *db0 = dw0 + 256 * 39, *dwh = db0 + 256, *dbh = dwh + 5 * 65536, *dw4 = dbh + 5 * 256, *db4 = dw4 + 256 * 295, *dw7 = db4 + 256;
    CK(hipDeviceSynchronize());
    // ---- the victim: lbs_bwd on a mesh the size of the scene the hazard was seen on ----
    const int nj = 55, nv = 2048;
    std::vector<float> hw((size_t)nv * nj, 0.f);
    for (int v = 0; v < nv; ++v) { float s = 0.f; int j0 = rand() % nj; for (int k = 0; k < 4; ++k) { float a = frand() + 0.05f; hw[(size_t)v * nj + (j0 + 7 * k) % nj] += a; s += a; } for (int j = 0; j < nj; ++j) hw[(size_t)v * nj + j] /= s; }
    auto affine = [&](int count) { std::vector<float> A((size_t)count * 16, 0.f); for (int j = 0; j < count; ++j) { float* a = &A[(size_t)j * 16]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) a[4 * r + c] = (r == c ? 1.f : 0.f) + (frand() - 0.5f) * 0.2f; a[15] = 1.f; } return A; };
    float *lbs_w = dupload(hw), *A0 = dupload(affine(nj)), *A = dupload(affine(nb * nj));
    std::vector<int> hidx(np); for (auto& v : hidx) v = rand() % nv;
    int* idx = dupload(hidx);
    float *pts = dupload(rnd(3 * (size_t)np, 0.6f)), *gout = dupload(rnd(3 * (size_t)np * nb, 1e-3f));
    const int NC = 64 * 256;          // canary size: 64 workgroups
    float* cin = dupload(rnd(NC, 1.f));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    if (getenv("REPRO_SAME_STREAM")) { sb = sa; printf("SAME STREAM: victims are ordered after the chain (no overlap)\n"); }
    float *dA = dalloc<float>((size_t)nb * nj * 16), *dT = dalloc<float>(3 * nb), *lbs_sum = dalloc<float>(3 * (size_t)np * RING);
    struct Victim {
        const char* name; size_t elems; int per_thread; std::function<void(float*, int)> launch;
        float* out = nullptr; std::vector<float> ref, got; long bad = 0; long lane_hist[64] = {0}; long comp_hist[16] = {0}; int shown = 0;
    };
    const dim3 gl((np + 255) / 256, nb);
    std::vector<Victim> V;
    V.push_back({"lbs_bwd (library) per-frame d_pts", 3 * (size_t)np * nb, 3, [&](float* o, int slot) {
        D3(lbs_bwd(pts, np, idx, lbs_w, nj, A0, A, nb, gout, lbs_sum + 3 * (size_t)np * slot, o, dA, dT, sb)); }});
    float *scr1 = dalloc<float>(3 * (size_t)np * RING), *scr2 = dalloc<float>(3 * (size_t)np * RING), *scr3 = dalloc<float>(3 * (size_t)np * RING);
    V.push_back({"lbs_bwd, dA = d_trans = NULL (no atomics run)", 3 * (size_t)np * nb, 3, [&](float* o, int slot) {
        D3(lbs_bwd(pts, np, idx, lbs_w, nj, A0, A, nb, gout, scr1 + 3 * (size_t)np * slot, o, nullptr, nullptr, sb)); }});
    V.push_back({"lbs_bwd, d_trans only (3 uniform LDS atomics)", 3 * (size_t)np * nb, 3, [&](float* o, int slot) {
        D3(lbs_bwd(pts, np, idx, lbs_w, nj, A0, A, nb, gout, scr2 + 3 * (size_t)np * slot, o, nullptr, dT, sb)); }});
    V.push_back({"lbs_bwd, dA only (per-joint LDS atomics)", 3 * (size_t)np * nb, 3, [&](float* o, int slot) {
        D3(lbs_bwd(pts, np, idx, lbs_w, nj, A0, A, nb, gout, scr3 + 3 * (size_t)np * slot, o, dA, nullptr, sb)); }});
    V.push_back({"blend dump: M[12], s, g[3]", 16 * (size_t)np * nb, 16, [&](float* o, int) {
        hipLaunchKernelGGL(victim_blend_dump, gl, dim3(256), 0, sb, np, idx, lbs_w, nj, A, gout, o); }});
    V.push_back({"blend, uniform control flow", 12 * (size_t)np * nb, 12, [&](float* o, int) {
        hipLaunchKernelGGL(victim_blend_uniform, gl, dim3(256), 0, sb, np, idx, lbs_w, nj, A, o); }});
    V.push_back({"copy of 3 floats per thread", 3 * (size_t)np * nb, 3, [&](float* o, int) {
        hipLaunchKernelGGL(victim_copy3, dim3((np * nb + 255) / 256), dim3(256), 0, sb, np * nb, gout, o); }});
    V.push_back({"valu canary (96 live registers)", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_valu, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 40); }});
    V.push_back({"lds round-trip canary", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_lds, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 8); }});
    V.push_back({"division chain canary", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_div, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC); }});
    V.push_back({"lds atomic canary", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_lds_atomic, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC); }});
    V.push_back({"v_pk_mul_f32 chain", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_pk<0>, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 256); }});
    V.push_back({"v_pk_add_f32 chain", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_pk<1>, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 256); }});
    V.push_back({"v_pk_fma_f32 chain", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_pk<2>, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 256); }});
    V.push_back({"v_pk_mul_f32 op_sel_hi chain", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_pk<3>, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 256); }});
    V.push_back({"v_mul_f32 x2 chain (scalar control)", (size_t)NC, 1, [&](float* o, int) { hipLaunchKernelGGL(canary_pk<4>, dim3(NC / 256), dim3(256), 0, sb, cin, o, NC, 256); }});
#define VL(F, label) V.push_back({label, 3 * (size_t)np * nb, 3, [&](float* o, int) { \
        hipLaunchKernelGGL(victim_lbs<F>, gl, dim3(256), 0, sb, pts, np, idx, lbs_w, nj, A0, A, nb, gout, o, (float*)nullptr, (float*)nullptr); }})
    VL(15, "lbs restated: everything (F=15)");
    VL(14, "lbs restated: no atomics code (F=14)");
    VL(12, "lbs restated: no atomics, no barriers/LDS (F=12)");
    VL(8, "lbs restated: frame blend only (F=8)");
    VL(4, "lbs restated: M0 blend + inverse only (F=4)");
    VL(10, "lbs restated: frame blend + barriers (F=10)");
    VL(6, "lbs restated: M0 + inverse + barriers (F=6)");
    VL(2, "lbs restated: barriers only (F=2)");
    const char* only = getenv("REPRO_ONLY");          // comma-free substring filter: run only the victims whose name contains it
    if (only) { std::vector<Victim> W; for (auto& v : V) if (strstr(v.name, only)) W.push_back(v); V.swap(W); }
    for (auto& v : V) { v.out = dalloc<float>(v.elems * RING); v.ref.resize(v.elems); v.got.resize(v.elems * RING); }
    auto victims = [&](int slot) { for (auto& v : V) v.launch(v.out + v.elems * slot, slot); };
    // reference: the victims alone on an idle GPU
    victims(0);
    CK(hipDeviceSynchronize());
    for (auto& v : V) CK(hipMemcpy(v.ref.data(), v.out, v.elems * 4, hipMemcpyDeviceToHost));
    // the victims alone, repeated: are they deterministic at all?
    long bad_alone = 0;
    for (int s = 0; s < RING; ++s) victims(s);
    CK(hipDeviceSynchronize());
    for (auto& v : V) {
        CK(hipMemcpy(v.got.data(), v.out, v.elems * RING * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < RING; ++s) bad_alone += memcmp(&v.got[v.elems * s], v.ref.data(), v.elems * 4) != 0;
    }
    printf("victims alone: %ld of %zu launches differ from the reference\n", bad_alone, (size_t)RING * V.size());
    long total = 0;
    float chain_ms = 0.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool no_chain = getenv("REPRO_NO_CHAIN") != nullptr, sweep_only = getenv("REPRO_SWEEP") != nullptr;
    for (int r = 0; r < rounds; ++r) {
        CK(hipMemsetAsync(arena, 0, ARENA * 4, sa));
        CK(hipEventRecord(e0, sa));
        if (sweep_only) {          // REPRO_SWEEP=1: the aggressor is the library's forward sweep (16x16x32 bf16 MFMAs, its highest matrix-pipe utilisation), x3
            for (int k = 0; k < 3; ++k) D3(sdf_mlp_fwd_x3(x, nullptr, 0.f, wp3, sdf, nullptr, getenv("REPRO_SWEEP_SAVE") ? act : nullptr, n, max_cus, sa));
        } else if (!no_chain) D3(sdf_mlp_eik_bwd(x, u, wp, wpt, wp3, 3, wpt3, 3, 0.f, act, dz, tb, eb, n, dw0, db0, dwh, dbh, dw4, db4, dw7, max_cus, sa));
        CK(hipEventRecord(e1, sa));
        for (int s = 0; s < RING; ++s) victims(s);          // queued while the chain runs
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); chain_ms += ms;
        total += RING;
        for (auto& v : V) {
            CK(hipMemcpy(v.got.data(), v.out, v.elems * RING * 4, hipMemcpyDeviceToHost));
            for (int s = 0; s < RING; ++s) {
                const float* gp = &v.got[v.elems * s];
                if (!memcmp(gp, v.ref.data(), v.elems * 4)) continue;
                ++v.bad;
                size_t first = v.elems, last = 0, cnt = 0;
                for (size_t k = 0; k < v.elems; ++k) if (memcmp(&gp[k], &v.ref[k], 4)) {
                    if (first == v.elems) first = k;
                    last = k; ++cnt;
                    const size_t thread = k / v.per_thread;                 // global thread id of the launch (256-thread workgroups, np per frame row)
                    const size_t in_row = (v.per_thread == 1) ? thread : thread % np;
                    ++v.lane_hist[in_row & 63];
                    ++v.comp_hist[k % v.per_thread];
                }
                if (v.shown < 3) {
                    ++v.shown;
                    printf("  %s: round %d slot %d: %zu floats differ, threads %zu..%zu\n", v.name, r, s, cnt, first / v.per_thread, last / v.per_thread);
                    for (size_t k = first, shown = 0; k <= last && shown < 24; ++k) if (memcmp(&gp[k], &v.ref[k], 4)) {
                        printf("      thread %zu (lane %zu) elem %zu: %.9g  vs  %.9g   [bits %08x vs %08x]\n", k / v.per_thread, ((v.per_thread == 1 ? k : (k / v.per_thread) % np)) & 63,
                               k % v.per_thread, gp[k], v.ref[k], *(const unsigned*)&gp[k], *(const unsigned*)&v.ref[k]);
                        ++shown;
                    }
                }
            }
        }
    }
    printf("%s n=%lld np=%d nb=%d max_cus=%d: chain %.3f ms/round; launches differing from the idle-GPU reference (of %ld each):\n", argv[1], (long long)n, np, nb,
           max_cus, chain_ms / rounds, total);
    long any = 0;
    for (auto& v : V) {
        any += v.bad;
        printf("   %-40s %ld", v.name, v.bad);
        if (v.bad) {
            printf("   lanes:");
            for (int l = 0; l < 64; ++l) if (v.lane_hist[l]) printf(" %d:%ld", l, v.lane_hist[l]);
            printf("   elems:");
            for (int c = 0; c < v.per_thread && c < 16; ++c) if (v.comp_hist[c]) printf(" %d:%ld", c, v.comp_hist[c]);
        }
        printf("\n");
    }
    return any ? 10 : 0;
}
