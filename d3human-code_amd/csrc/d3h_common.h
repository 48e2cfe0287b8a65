// d3h_common.h -- shared helpers for the gfx950 kernels behind the C ABI in include/d3h.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define D3H_OK 0
#define D3H_ERR_ARG (-1)

// Kernels are enqueued on the caller's stream and never synchronise (reference plugin convention:
// render/renderutils/c_src/torch_bindings.cpp:25-41 -- launch, then check the launch error only).
#define D3H_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

static inline int d3h_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs several times over,
// capped so the tail is short (cdna_hip_programming.md Guideline 11).
static inline int d3h_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}
