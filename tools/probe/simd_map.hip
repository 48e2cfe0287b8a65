// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// hipcc --offload-arch=gfx950 -O2 -o simd_map tools/probe/simd_map.hip && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void probe(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 8 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(G), dim3(512), 0, 0, d);
    static unsigned h[G * 8 * 2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < G; b += (b < 4 ? 1 : 127)) {
        printf("block %3d xcc %u cu %2u se %u: simd of waves 0..7 =", b, h[b * 16 + 1] & 15, (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
        printf("   wave slots =");
        for (int w = 0; w < 8; ++w) printf(" %u", h[(b * 8 + w) * 2] & 15);
        printf("\n");
    }
    int hist[8][4] = {};
    for (int b = 0; b < G; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[(b * 8 + w) * 2] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int same = 0;
    for (int b = 0; b < G; ++b) for (int w = 0; w < 4; ++w) same += ((h[(b * 8 + w) * 2] >> 4) & 3) == ((h[(b * 8 + w + 4) * 2] >> 4) & 3);
    printf("waves w and w+4 on the same SIMD: %d of %d\n", same, G * 4);
    return 0;
}
