/* d3h.h -- C ABI of libd3h_hip.so: the MI355X (gfx950) hot path of D3-Human's render-and-fit loop.
 *
 * Conventions (mirroring the reference's native plugin, render/renderutils/c_src/torch_bindings.cpp:25-41):
 *  - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch-ROCm allocations);
 *    tensors are contiguous row-major; float = fp32, indices int32 unless stated;
 *  - `stream` is a hipStream_t; kernels are enqueued on it and the call returns without synchronising;
 *  - return 0 on success, <0 for an argument error, >0 = hipError_t of a failed launch
 *    (the Python wrappers raise RuntimeError, as TORCH_CHECK does in the reference);
 *  - data-dependent output sizes are returned through small device counters the caller reads back.
 * Each entry point names the reference interface (file:line under the reference root) it replaces.
 */
#ifndef D3H_H
#define D3H_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- SDF query: geometry/embedding.py:21-38 + geometry/mlp.py:34-45 + geometry/hmsdf.py:433-444 ---- */
/* number of floats of the packed weight buffer / of the saved-activation buffer for n points */
int64_t d3h_sdf_mlp_wpack_floats(void);
int64_t d3h_sdf_mlp_act_floats(int64_t n);
/* pack nn.Linear weights of MLP(n_freq=6,d_hidden=256,n_hidden=6,skip_in=[3]) (geometry/mlp.py:10-32):
 * w0[256][39] b0[256] (net.0); wh[5][256][256] bh[5][256] (net.2,4,6,10,12); w4[256][295] b4[256] (net.8);
 * w7[1][256] b7[1] (net.14) */
int d3h_sdf_mlp_pack(const float* w0, const float* b0, const float* wh, const float* bh, const float* w4,
                     const float* b4, const float* w7, const float* b7, float* wpack, void* stream);
/* sdf[n] = MLP(x + disp*deform); deform may be NULL; xdef[n][3] (optional) receives the deformed points;
 * act (optional, d3h_sdf_mlp_act_floats(n) floats) receives the activations the backward pass needs */
int d3h_sdf_mlp_fwd(const float* x, const float* deform, float disp, const float* wpack, float* sdf, float* xdef,
                    float* act, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
