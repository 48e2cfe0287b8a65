// Do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?  Workgroups of 8 waves (waves w and w + 4 share a
// SIMD, tools/probe/simd_map.hip): waves 0..3 run a v_mfma_f32_16x16x4_f32 loop (two chains), waves 4..7 a VALU loop (fma, or exp/log).
// Times: MFMA waves alone, VALU waves alone, both.  Overlap => both ~ max(alone); none => both ~ sum.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl tools/probe/mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int TRANS>   // MODE bit 0: MFMA waves work, bit 1: VALU waves work
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            }
        }
        if (c0[0] + c1[3] == 123.456f) out[threadIdx.x] = c0[0];
    } else {
        if (!(MODE & 2)) return;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a + j + threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < (TRANS ? 2 : 16); ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (TRANS) v[j] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(v[j] * b));
                    else v[j] = __builtin_fmaf(v[j], b, a);
                }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000, G = 256;
#define RUN(M, T) time_ms([&] { hipLaunchKernelGGL((k<M, T>), dim3(G), dim3(512), 0, 0, out, iters, 1.0001f, 0.9999f); })
    for (int rep = 0; rep < 2; ++rep) {
        printf("fma   : mfma alone %.3f ms, valu alone %.3f ms, both %.3f ms\n", RUN(1, 0), RUN(2, 0), RUN(3, 0));
        printf("exp/log: mfma alone %.3f ms, valu alone %.3f ms, both %.3f ms\n", RUN(1, 1), RUN(2, 1), RUN(3, 1));
    }
    printf("(16 MFMAs = 512 pipe cycles per iteration; VALU: 128 fma = 512 issue cycles, or 16 x (mul, exp, add, log))\n");
    return 0;
}
