// deform_mlp.hip -- the offset network MLP_deform (geometry/mlp.py:77-118; seq stage, hmsdf.py:658-665) on the fused SDF-query kernels.
//
// MLP_deform(n_freq 8, d_hidden 256, n_hidden 6, skip_in [3], d_out 3) is the SDF network's shape with a 51-wide positional encoding,
// a 3-output head and a 136-float pose code prepended to the first layer's input.  The code is the same for every point, so
//   W0 [code; emb] + b0 = W0[:, 136:] emb + (b0 + W0[:, :136] code)
// and the host wrapper (d3h/deform_mlp.py) folds it into the first bias; what is left is exactly the kernel family of sdf_mlp.hip /
// sdf_mlp_bwd.hip, compiled a second time with EMB_DIM = 51 (4 embedding blocks) and NOUT = 3.  Entry points: d3h_deform_mlp_*
// (same signatures as d3h_sdf_mlp_*; x-gradients, the `deform` displacement input and the eikonal passes are not part of this variant).
#define D3H_MLP_NFREQ 8
#define D3H_MLP_NOUT 3
#define D3H_MLP_NS d3h_dmlp
#define sdf_mlp_pack_kernel deform_mlp_pack_kernel
#define sdf_mlp_fwd_kernel deform_mlp_fwd_kernel
#define d3h_sdf_mlp_wpack_floats d3h_deform_mlp_wpack_floats
#define d3h_sdf_mlp_act_floats d3h_deform_mlp_act_floats
#define d3h_sdf_mlp_pack d3h_deform_mlp_pack
#define d3h_sdf_mlp_fwd d3h_deform_mlp_fwd
#include "sdf_mlp.hip"
