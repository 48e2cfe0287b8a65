// Self-contained reproducer (no library, no torch) of the hazard behind DESIGN.md section 3 "co-residency": on MI355X (gfx950, ROCm 7.2) a wave
// that executes PACKED-F32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- what hipcc -O3 SLP-packs adjacent scalar f32
// arithmetic into) returns wrong values in lanes 48..63 when a wave of ANOTHER kernel issuing matrix instructions shares its SIMD.
//
//   stream A (aggressor): a register-only MFMA loop, templated on the instruction (32x32x16 bf16, 16x16x32 bf16, 32x32x2 f32, or plain VALU
//                         FMAs as the control), launched with few enough VGPRs that foreign waves fit beside its two waves per SIMD --
//                         or, with CLAIM, with v255 touched so that its two waves own the whole register file (the library's mitigation);
//   stream B (victim):    a skinning-style kernel (per-lane weighted sum of 3x4 matrices read through SGPRs, then a 3x3 mat-vec), built
//                         twice from one source: as hipcc -O3 compiles it (packed f32) and with the arithmetic pinned to scalar v_fma/v_mul;
//   every victim launch is compared bitwise with the same launch on an idle GPU.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe/mfma_pk_hazard.cpp -o tools/probe/mfma_pk_hazard && tools/probe/mfma_pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- aggressors --------------------------------------------------------------------------------------------------------------------------
// KIND 0: v_mfma_f32_32x32x16_bf16   1: v_mfma_f32_16x16x32_bf16   2: v_mfma_f32_32x32x2_f32   3: v_fma_f32 only (control)
// CLAIM: touch v255, so that the kernel is allocated 256 VGPRs and two waves per SIMD leave no register for a foreign wave
template <int KIND, bool CLAIM, int WG>
__global__ __launch_bounds__(WG) void aggressor(const unsigned* __restrict__ in, float* __restrict__ out, int iters) {
    if (CLAIM) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const unsigned long long t_start = wall_clock64();
    const int tid = threadIdx.x + blockIdx.x * WG;
    u32x4 a = *(const u32x4*)(in + 4 * (size_t)(tid & 4095)), b = *(const u32x4*)(in + 4 * (size_t)((tid + 977) & 4095));
    float res = 0.f;
    if (KIND == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
        }
        for (int k = 0; k < 16; ++k) res += c0[k] + c1[k] + c2[k] + c3[k];
    } else if (KIND == 1) {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < 2 * iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) res += c0[k] + c1[k] + c2[k] + c3[k];
    } else if (KIND == 20 || KIND == 21) {          // 16x16x32 bf16, then a workgroup barrier AFTER the last MFMA (20), + a 2 us MFMA-free epilogue (21): no wave of
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};   // the workgroup leaves its SIMD while a sibling still issues matrix instructions
        for (int i = 0; i < 2 * iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) res += c0[k] + c1[k] + c2[k] + c3[k];
        __syncthreads();
        if (KIND == 21) { for (int k = 0; k < 75; ++k) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"); }
    } else if (KIND >= 4) {          // 16x16x32 bf16 at a reduced duty cycle: KIND - 3 idle groups of 64 cycles per 4 MFMAs (64 cycles)
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < 2 * iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KIND - 3; ++k) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
        }
        for (int k = 0; k < 4; ++k) res += c0[k] + c1[k] + c2[k] + c3[k];
    } else if (KIND == 2) {
        f32x16 c0 = {0}, c1 = {0};
        const float fa = __uint_as_float(a[0] & 0x3fffffffu), fb = __uint_as_float(b[1] & 0x3fffffffu);
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c1, 0, 0, 0);
        }
        for (int k = 0; k < 16; ++k) res += c0[k] + c1[k];
    } else {
        float r[16];
        for (int k = 0; k < 16; ++k) r[k] = __uint_as_float((a[k & 3] & 0x007fffffu) | 0x3f000000u);
        for (int i = 0; i < 8 * iters; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(0.999f), "v"(1e-3f));
        for (int k = 0; k < 16; ++k) res += r[k];
    }
    out[tid] = res;
    if (threadIdx.x == 0 && blockIdx.x < 256)   // where the aggressor's first 256 workgroups ran
    {
        ((unsigned*)out)[(size_t)gridDim.x * WG + blockIdx.x] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 28) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0x0fffffffu);
        unsigned long long* ts = (unsigned long long*)((unsigned*)out + (size_t)gridDim.x * WG + 256);
        ts[2 * blockIdx.x] = t_start; ts[2 * blockIdx.x + 1] = wall_clock64();
    }
}

// ---- victim -------------------------------------------------------------------------------------------------------------------------------
// SCALAR = false: plain C, compiled by hipcc -O3 into v_pk_fma_f32 (weights x SGPR matrix rows) + v_pk_mul_f32 / v_pk_add_f32 (mat-vec);
// SCALAR = true: the same arithmetic, same order, through single-instruction asm helpers (no packing possible)
__device__ __forceinline__ float fma1(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float mul1(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float add1(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
template <bool SCALAR>
__global__ __launch_bounds__(256) void victim(int np, const int* __restrict__ idx, const float* __restrict__ w_all, int nj, const float* __restrict__ A,
                                              const float* __restrict__ gin, float* __restrict__ out, unsigned* __restrict__ hw, unsigned long long* __restrict__ ts) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const int wv = (b * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (hw && (threadIdx.x & 63) == 0) {        // where this wave runs: XCC id (bits 31..28), HW_ID (SE / SH / CU / SIMD / wave slot); when it starts
        hw[wv] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 28) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0x0fffffffu);
        if (ts) ts[2 * wv] = wall_clock64();
    }
    if (p >= np) return;
    const float* w = w_all + (size_t)idx[p] * nj;
    const float* Ab = A + (size_t)b * nj * 16;
    float M[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) M[e] = 0.f;
    for (int j = 0; j < nj; ++j) {
        const float wj = w[j];
        if (wj != 0.f) {
            const float* a = Ab + 16 * j;
#pragma unroll
            for (int e = 0; e < 12; ++e) M[e] = SCALAR ? fma1(wj, a[e], M[e]) : fmaf(wj, a[e], M[e]);
        }
    }
    const float* g = gin + ((size_t)b * np + p) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    float* o = out + ((size_t)b * np + p) * 3;
    if (SCALAR) {
        o[0] = add1(add1(mul1(M[0], g0), mul1(M[4], g1)), mul1(M[8], g2));
        o[1] = add1(add1(mul1(M[1], g0), mul1(M[5], g1)), mul1(M[9], g2));
        o[2] = add1(add1(mul1(M[2], g0), mul1(M[6], g1)), mul1(M[10], g2));
    } else {
        o[0] = M[0] * g0 + M[4] * g1 + M[8] * g2;
        o[1] = M[1] * g0 + M[5] * g1 + M[9] * g2;
        o[2] = M[2] * g0 + M[6] * g1 + M[10] * g2;
    }
    if (hw && ts && (threadIdx.x & 63) == 0) ts[2 * wv + 1] = wall_clock64();
}

static float frand() { return rand() / (float)RAND_MAX; }
template <class T> static T* dupload(const std::vector<T>& h) { T* p; CK(hipMalloc(&p, h.size() * sizeof(T))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

struct Agg { const char* name; void (*launch)(const unsigned*, float*, int, int, hipStream_t); int wgs = 2048, iters = 600; bool zeros = false; };
template <int KIND, bool CLAIM, int WG> static void launch_agg(const unsigned* in, float* out, int iters, int wgs, hipStream_t s) {
    hipLaunchKernelGGL((aggressor<KIND, CLAIM, WG>), dim3(wgs), dim3(WG), 60000, s, in, out, iters);      // 60 KB of (unused) LDS: at most 2 workgroups per CU, so victims find room
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const int np = argc > 2 ? atoi(argv[2]) : 8770, nb = 4, nj = 55, nv = 2048, RING = 48;
    const bool same_stream = argc > 3 && atoi(argv[3]) == 1;          // victims queued BEHIND the aggressor on its stream: no overlap in time
    srand(11);
    std::vector<float> hw((size_t)nv * nj, 0.f);
    for (int v = 0; v < nv; ++v) { float s = 0.f; int j0 = rand() % nj; for (int k = 0; k < 4; ++k) { float a = frand() + 0.05f; hw[(size_t)v * nj + (j0 + 7 * k) % nj] += a; s += a; } for (int j = 0; j < nj; ++j) hw[(size_t)v * nj + j] /= s; }
    std::vector<float> hA((size_t)nb * nj * 16, 0.f);
    for (int j = 0; j < nb * nj; ++j) { float* a = &hA[(size_t)j * 16]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) a[4 * r + c] = (r == c ? 1.f : 0.f) + (frand() - 0.5f) * 0.2f; a[15] = 1.f; }
    std::vector<int> hidx(np); for (auto& v : hidx) v = rand() % nv;
    std::vector<float> hg(3 * (size_t)np * nb); for (auto& v : hg) v = (frand() - 0.5f) * 2e-3f;
    std::vector<unsigned> hin(4 * 4096); for (auto& v : hin) { unsigned lo = 0x3f00u | (rand() & 0xff), hi = 0x3f00u | (rand() & 0xff); v = lo | (hi << 16); }     // bf16 pairs in [0.5, 1)
    float *w = dupload(hw), *A = dupload(hA), *g = dupload(hg);
    int* idx = dupload(hidx);
    unsigned* ain = dupload(hin);
    const int AWG = 16384;                       // (largest aggressor grid below)
    float* aout; CK(hipMalloc(&aout, ((size_t)AWG * 512 + 256 + 4 * 256) * 4));
    unsigned* azero; CK(hipMalloc(&azero, 4 * 4096 * 4)); CK(hipMemset(azero, 0, 4 * 4096 * 4));
    const size_t OUT = 3 * (size_t)np * nb;
    float *vo[2]; for (auto& p : vo) CK(hipMalloc(&p, OUT * RING * 4));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    if (same_stream) { sb = sa; printf("SAME STREAM: every victim launch is ordered after the aggressor kernel (no temporal overlap)\n"); }
    const dim3 gl((np + 255) / 256, nb);
    const int NW = gl.x * gl.y * 4;               // waves per victim launch
    unsigned* vhw; CK(hipMalloc(&vhw, (size_t)NW * RING * 4));
    std::vector<unsigned> hhw((size_t)NW * RING), ahw(256);
    unsigned long long* vts; CK(hipMalloc(&vts, 2 * (size_t)NW * RING * 8));
    std::vector<unsigned long long> hts(2 * (size_t)NW * RING), ats(512);
    auto victims = [&](int slot) {
        hipLaunchKernelGGL(victim<false>, gl, dim3(256), 0, sb, np, idx, w, nj, A, g, vo[0] + OUT * slot, vhw + (size_t)NW * slot, vts + 2 * (size_t)NW * slot);
        hipLaunchKernelGGL(victim<true>, gl, dim3(256), 0, sb, np, idx, w, nj, A, g, vo[1] + OUT * slot, (unsigned*)nullptr, (unsigned long long*)nullptr);
    };
    victims(0);
    CK(hipDeviceSynchronize());
    std::vector<float> ref[2] = {std::vector<float>(OUT), std::vector<float>(OUT)}, got(OUT * RING);
    for (int k = 0; k < 2; ++k) CK(hipMemcpy(ref[k].data(), vo[k], OUT * 4, hipMemcpyDeviceToHost));
    const Agg aggs[] = {
        {"none (victims alone)", nullptr},
        {"v_mfma_f32_32x32x16_bf16, 512-thread WG, shares SIMDs", launch_agg<0, false, 512>},
        {"v_mfma_f32_32x32x16_bf16, 256-thread WG, shares SIMDs", launch_agg<0, false, 256>},
        {"v_mfma_f32_16x16x32_bf16, 512-thread WG, shares SIMDs", launch_agg<1, false, 512>},
        {"v_mfma_f32_32x32x2_f32,   512-thread WG, shares SIMDs", launch_agg<2, false, 512>},
        {"v_fma_f32 only (control), 512-thread WG, shares SIMDs", launch_agg<3, false, 512>},
        {"v_mfma_f32_32x32x16_bf16, 512-thread WG, CLAIMS v0..v255", launch_agg<0, true, 512>},
        {"v_mfma_f32_16x16x32_bf16, 512-thread WG, CLAIMS v0..v255", launch_agg<1, true, 512>},
        // where does it come from?  (a) few long-lived claiming workgroups: victims can only run on OTHER CUs, and there are few transitions
        {"16x16x32 bf16 CLAIM, 128 persistent WGs (x16 iterations)", launch_agg<1, true, 512>, 128, 9600},
        {"16x16x32 bf16 CLAIM, 255 persistent WGs (x8 iterations)", launch_agg<1, true, 512>, 255, 4800},
        // (b) the churning launch with all-zero operands (same instruction stream, minimal switching activity)
        {"16x16x32 bf16 CLAIM, 2048 WGs, ZERO operands", launch_agg<1, true, 512>, 2048, 600, true},
        {"16x16x32 bf16 shares, 2048 WGs, ZERO operands", launch_agg<1, false, 512>, 2048, 600, true},
        // (c) many very short workgroups: the number of workgroup starts / ends goes up 8x at the same total work
        {"16x16x32 bf16 CLAIM, 16384 WGs of 75 iterations", launch_agg<1, true, 512>, 16384, 75},
        {"32x32x16 bf16 CLAIM, 16384 WGs of 75 iterations", launch_agg<0, true, 512>, 16384, 75},
        {"32x32x2 f32 CLAIM, 16384 WGs of 75 iterations", launch_agg<2, true, 512>, 16384, 75},
        {"v_fma_f32 only CLAIM, 16384 WGs of 75 iterations", launch_agg<3, true, 512>, 16384, 75},
        // matrix-pipe duty cycle (two waves per SIMD alternate: the pipe is busy 2 x 64 of every 64 + 64 k cycles per wave)
        {"16x16x32 bf16 CLAIM, 128 persistent WGs, 1 idle group per 4 MFMAs", launch_agg<4, true, 512>, 128, 4800},
        {"16x16x32 bf16 CLAIM, 128 persistent WGs, 3 idle groups per 4 MFMAs", launch_agg<6, true, 512>, 128, 2400},
        {"16x16x32 bf16 CLAIM, 128 persistent WGs, 7 idle groups per 4 MFMAs", launch_agg<10, true, 512>, 128, 1200},
        // the mitigation pattern of the library: claim the register file AND finish the workgroup's matrix phase with a barrier
        {"16x16x32 bf16 CLAIM + barrier after the last MFMA, 2048 WGs", launch_agg<20, true, 512>},
        {"16x16x32 bf16 CLAIM + barrier + 2 us MFMA-free tail, 2048 WGs", launch_agg<21, true, 512>},
        {"16x16x32 bf16 shares + barrier after the last MFMA, 2048 WGs", launch_agg<20, false, 512>},
        {"16x16x32 bf16 CLAIM + barrier, 128 persistent WGs", launch_agg<20, true, 512>, 128, 9600},
        {"16x16x32 bf16 CLAIM, 32 persistent WGs", launch_agg<1, true, 512>, 32, 9600},
        {"16x16x32 bf16 CLAIM, 8 persistent WGs", launch_agg<1, true, 512>, 8, 9600},
        {"16x16x32 bf16 CLAIM, 1 persistent WG", launch_agg<1, true, 512>, 1, 9600},
    };
    printf("%-62s %10s %22s %22s   lanes of the wrong values\n", "aggressor (stream A)", "ms/launch", "packed-f32 victim bad", "scalar-f32 victim bad");
    int rc = 0;
    for (const Agg& ag : aggs) {
        long bad[2] = {0, 0}, total = 0, lanes[64] = {0}, bad_waves = 0, bad_same_xcc = 0, bad_same_cu = 0, overlap_in_time = 0, inside = 0, after_end = 0, before_start = 0; int shown = 0;
        const int n_agg_wg = ag.launch ? (ag.wgs < 256 ? ag.wgs : 256) : 0;
        float ms_sum = 0.f;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, sa));
            if (ag.launch) ag.launch(ag.zeros ? azero : ain, aout, ag.iters, ag.wgs, sa);
            CK(hipEventRecord(e1, sa));
            for (int s = 0; s < RING; ++s) victims(s);
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms;
            total += RING;
            CK(hipMemcpy(hhw.data(), vhw, (size_t)NW * RING * 4, hipMemcpyDeviceToHost));
            if (ag.launch) CK(hipMemcpy(ahw.data(), (unsigned*)aout + (size_t)ag.wgs * 512, n_agg_wg * 4, hipMemcpyDeviceToHost));
            if (ag.launch) CK(hipMemcpy(ats.data(), (unsigned*)aout + (size_t)ag.wgs * 512 + 256, n_agg_wg * 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hts.data(), vts, 2 * (size_t)NW * RING * 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < 2; ++k) {
                CK(hipMemcpy(got.data(), vo[k], OUT * RING * 4, hipMemcpyDeviceToHost));
                for (int s = 0; s < RING; ++s) {
                    if (!memcmp(&got[OUT * s], ref[k].data(), OUT * 4)) continue;
                    ++bad[k];
                    for (size_t e = 0; e < OUT; ++e) if (memcmp(&got[OUT * s + e], &ref[k][e], 4)) {
                        ++lanes[((e / 3) % np) & 63];
                        if (k == 0 && (e % 3) == 0 && (((e / 3) % np) & 63) == 48) {          // one record per wrong wave
                            const size_t pt = (e / 3) % np, fr = (e / 3) / np;
                            const unsigned h = hhw[(size_t)NW * s + (fr * gl.x + pt / 256) * 4 + (pt % 256) / 64];
                            ++bad_waves;
                            bool same_cu = false, same_xcc = false;
                            const size_t wi = (size_t)NW * s + (fr * gl.x + pt / 256) * 4 + (pt % 256) / 64;
                            const unsigned long long v0 = hts[2 * wi], v1 = hts[2 * wi + 1];
                            for (int q = 0; q < n_agg_wg; ++q) {
                                if ((ahw[q] >> 28) == (h >> 28)) same_xcc = true;
                                if ((ahw[q] >> 28) == (h >> 28) && ((ahw[q] >> 8) & 0xff) == ((h >> 8) & 0xff)) {       // SE / SH / CU bits 15..8
                                    same_cu = true;
                                    // the victim wave's lifetime against this aggressor workgroup's (100 MHz wall clock: 10 ns ticks)
                                    if (v1 > ats[2 * q] && v0 < ats[2 * q + 1]) { ++overlap_in_time; if (v0 > ats[2 * q] && v1 < ats[2 * q + 1]) ++inside; }
                                    else if (v0 >= ats[2 * q + 1]) ++after_end; else ++before_start;
                                }
                            }
                            bad_same_xcc += same_xcc; bad_same_cu += same_cu;
                            if (shown < 6 && ag.wgs <= 32) { ++shown; printf("      wrong wave: xcc %u se/sh/cu %02x simd %u   (aggressor WG 0: xcc %u se/sh/cu %02x)\n", h >> 28, (h >> 8) & 0xff, (h >> 4) & 3, ahw[0] >> 28, (ahw[0] >> 8) & 0xff); }
                        }
                    }
                }
            }
        }
        char l[256] = ""; int n = 0;
        for (int q = 0; q < 4; ++q) { long c = 0; for (int i = 16 * q; i < 16 * q + 16; ++i) c += lanes[i]; n += snprintf(l + n, sizeof(l) - n, " %d-%d:%ld", 16 * q, 16 * q + 15, c); }
        printf("%-62s %10.3f %15ld of %ld %15ld of %ld  %s", ag.name, ms_sum / rounds, bad[0], total, bad[1], total, l);
        if (bad_waves) printf("   wrong waves %ld: on an XCD with an aggressor WG %ld, on a CU (se/sh/cu) with one %ld; vs that WG's lifetime: overlapping %ld (entirely inside %ld), after its end %ld, before its start %ld",
                              bad_waves, bad_same_xcc, bad_same_cu, overlap_in_time, inside, after_end, before_start);
        printf("\n");
        if (bad[0] || bad[1]) rc = 10;
    }
    return rc;
}
