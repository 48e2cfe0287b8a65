// deform_mlp_bwd.hip -- backward of the offset network MLP_deform: sdf_mlp_bwd.hip compiled with EMB_DIM = 51, NOUT = 3 (see
// deform_mlp.hip).  Entry points d3h_deform_mlp_{wpackt_floats, pack_t, bwd}.
#define D3H_MLP_NFREQ 8
#define D3H_MLP_NOUT 3
#define D3H_MLP_NS d3h_dmlp
#define d3h_sdf_mlp_wpackt_floats d3h_deform_mlp_wpackt_floats
#define d3h_sdf_mlp_pack_t d3h_deform_mlp_pack_t
#define d3h_sdf_mlp_bwd d3h_deform_mlp_bwd
#define d3h_sdf_mlp_bwd_scratch_ints d3h_deform_mlp_bwd_scratch_ints
#include "sdf_mlp_bwd.hip"
