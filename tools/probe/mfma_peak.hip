// mfma_peak.hip -- what the fp32 matrix pipe of THIS MI355X sustains: register-only v_mfma_f32_16x16x4_f32 / 32x32x2_f32 loops, no memory
// traffic.  Used to put the SDF-query kernel's 105-112 TFLOP/s against a measured ceiling next to the data-sheet 157.3 TFLOP/s
// (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz).   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probe/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x16){0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        int grid = 256 * waves_per_simd / 2 * 1;          // 512-thread workgroups = 8 waves = 2 per SIMD; half as many WGs for 1 per SIMD
        int block = waves_per_simd == 2 ? 512 : 256;
        grid = 256;
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k16<4>), dim3(grid), dim3(block), 0, 0, out, iters, 1.0f, 0.5f); });
            double fl = (double)grid * (block / 64) * iters * 4 * 2048.0;
            printf("16x16x4 f32, 4 chains, %d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s\n", waves_per_simd, ms, fl / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k16<2>), dim3(grid), dim3(block), 0, 0, out, iters, 1.0f, 0.5f); });
            double fl = (double)grid * (block / 64) * iters * 2 * 2048.0;
            printf("16x16x4 f32, 2 chains, %d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s\n", waves_per_simd, ms, fl / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k32<2>), dim3(grid), dim3(block), 0, 0, out, iters / 2, 1.0f, 0.5f); });
            double fl = (double)grid * (block / 64) * (iters / 2) * 2 * 4096.0;
            printf("32x32x2 f32, 2 chains, %d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s\n", waves_per_simd, ms, fl / ms / 1e9);
        }
    }
    // long run: does the clock hold for the ~2 ms a full-grid SDF sweep lasts, and for 50 ms?
    for (int it : {10000, 250000}) {
        double ms = time_ms([&] { hipLaunchKernelGGL((k16<4>), dim3(256), dim3(512), 0, 0, out, it, 1.0f, 0.5f); });
        double fl = 256.0 * 8 * it * 4 * 2048.0;
        printf("16x16x4 f32, 4 chains, 2 waves/SIMD, %d iters: %.2f ms  %.1f TFLOP/s\n", it, ms, fl / ms / 1e9);
    }
    hipFree(out);
    return 0;
}
