// d3h_common.h -- shared helpers for the gfx950 kernels behind the C ABI in include/d3h.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ABI version of include/d3h.h: bumped whenever an existing entry point changes its signature or is removed (a caller built against an older
// header would otherwise pass arguments into the wrong slots with no diagnostic).  d3h/_lib.py refuses a library of another version.
//   4 (round 4): d3h_sdf_mlp_fwd / _grad_x / _eik_bwd gained `int max_cus` before `stream`; d3h_sdf_mlp_bwd gained `wpackT3`;
//                d3h_sdf_mlp_overlap_cus (process-wide state) was removed
//   5 (round 5): d3h_abi_version itself; see INTEGRATION.md "ABI history" for what else this round changed
//   6, 7 (round 6): the fp16 x 2 packs / plane counts; the occupancy cells of the SSIM passes
#define D3H_ABI_VERSION 7
#define D3H_OK 0
#define D3H_ERR_ARG (-1)

// Kernels are enqueued on the caller's stream and never synchronise (reference plugin convention:
// render/renderutils/c_src/torch_bindings.cpp:25-41 -- launch, then check the launch error only).
#define D3H_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// dynamic LDS of a kernel (the host emulation used by the tests substitutes its own definition)
#ifndef D3H_DYN_SHARED
#define D3H_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

// 16-byte direct global -> LDS load (global_load_lds_dwordx4, gfx950): every lane supplies its own global address; the data lands at
// `lds_wave_base` (wave-uniform) + lane * 16.  No VGPR destination: completion is tracked by vmcnt.
#ifndef D3H_GLDS16
#define D3H_GLDS16(gsrc, lds_wave_base)                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),                                 \
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#endif

// Register budget hint: at most 512 / n VGPRs + AGPRs so that n waves fit per SIMD (the host emulation defines it away).
#ifndef D3H_WAVES_PER_EU
#define D3H_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

// Wave-private LDS exchange (lanes of ONE wave write, then read each other's data): the LDS executes a wave's instructions in issue order,
// so no s_barrier is needed -- only the compiler must keep the order (the host emulation substitutes a wave-level rendezvous).
// nothing may be scheduled across this point (used to pin software-pipelined LDS reads ahead of the MFMAs that hide them)
#ifndef D3H_SCHED_FENCE
#define D3H_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef D3H_WAVE_SYNC
#define D3H_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

// ---- per-launch kernel timing (csrc/timing.hip); ids of the instrumented kernels --------------------------------------------------
enum { D3H_KT_SDF_FWD = 0, D3H_KT_SDF_BWD_DATA = 1, D3H_KT_SDF_TANGENT = 2, D3H_KT_SDF_BWD_INJECT = 3, D3H_KT_SDF_DW_LAYERS = 4,
       D3H_KT_SDF_BWD_DATA_SPARSE = 5, D3H_KT_SDF_DW_LAYERS_SPARSE = 6, D3H_KT_TEX_BWD_MLP = 7, D3H_KT_TEX_BWD_ENC = 8, D3H_KT_GBUFFER_BWD = 9,
       D3H_KT_TEX_FWD = 10, D3H_KT_AA_FWD = 11, D3H_KT_RASTER_FWD = 12, D3H_KT_AA_PREP = 13, D3H_KT_COMPOSITE_FWD = 14,
       D3H_KT_COMPOSITE_BWD = 15, D3H_KT_PIXLOSS_FWD = 16, D3H_KT_PIXLOSS_BWD = 17, D3H_KT_SSIM_FWD = 18, D3H_KT_SSIM_BWD = 19,
       D3H_KT_MTETS_COUNT = 20, D3H_KT_MTETS_EMIT = 21, D3H_KT_LBS_FWD = 22, D3H_KT_LBS_BWD = 23, D3H_KT_AA_BWD = 24, D3H_KT_GBUFFER_FWD = 25,
       D3H_KT_RASTER_BWD = 26, D3H_KT_SDF_FWD_RECOMPUTE = 27 };
#ifndef D3H_EMULATED
int d3h_ktime_begin(int id, long long units, hipStream_t s);
void d3h_ktime_end(int handle, hipStream_t s);
#else
static inline int d3h_ktime_begin(int, long long, hipStream_t) { return -1; }
static inline void d3h_ktime_end(int, hipStream_t) {}
#endif

static inline int d3h_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs several times over,
// capped so the tail is short (cdna_hip_programming.md Guideline 11).
static inline int d3h_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// ---- wave-level segmented sums -------------------------------------------------------------------------------------------------
// Pixels of an image row are consecutive lanes, and neighbouring pixels mostly hit the same triangle / grid cell: per-lane fp32
// atomics would hammer a handful of addresses.  Lanes with equal `key` that are adjacent form a run; d3h_seg_sum returns, in the
// LAST lane of each run, the sum over the run (an inclusive segmented scan), so one atomic per run suffices.  A key that re-appears
// in a later run simply gets a second add.  All 64 lanes must call these (no early returns before them).
struct D3hSeg { int start; bool tail; };
__device__ __forceinline__ D3hSeg d3h_seg_runs(int key, int lane) {
    const int prev = __shfl_up(key, 1);
    const bool head = (lane == 0) || (key != prev);
    const unsigned long long heads = __ballot(head);
    D3hSeg s;
    s.start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
    s.tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
    return s;
}
// Lane shifts inside a row of 16 lanes as DPP modifiers (row_shr:N; a lane without a source reads 0) and a scalar read of one lane:
// no LDS-crossbar traffic.  The first version of d3h_seg_sum used six __shfl_up (ds_bpermute_b32) per value; the texture-grid backward
// runs 80 segmented sums per 64 pixels and spent 78 % of its wave cycles parked on them (rocprofv3 SQ_WAIT_ANY).
#ifndef D3H_ROW_SHR
#define D3H_ROW_SHR(v, N) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + (N), 0xf, 0xf, false))
#define D3H_READLANE(v, L) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (L)))
#endif
__device__ __forceinline__ float d3h_seg_sum(float v, int lane, int start) {
    const int row0 = lane & ~15;
    const int lo = start > row0 ? start : row0;          // first lane of this lane's run inside its own row of 16
    float u;
    u = D3H_ROW_SHR(v, 1); if (lane - 1 >= lo) v += u;   // in-row inclusive segmented scan
    u = D3H_ROW_SHR(v, 2); if (lane - 2 >= lo) v += u;
    u = D3H_ROW_SHR(v, 4); if (lane - 4 >= lo) v += u;
    u = D3H_ROW_SHR(v, 8); if (lane - 8 >= lo) v += u;
    // a run that began in an earlier row continues through the last lane of the previous row, whose value is complete by then
    u = D3H_READLANE(v, 15); if (row0 == 16 && start <= 15) v += u;
    u = D3H_READLANE(v, 31); if (row0 == 32 && start <= 31) v += u;
    u = D3H_READLANE(v, 47); if (row0 == 48 && start <= 47) v += u;
    return v;
}
