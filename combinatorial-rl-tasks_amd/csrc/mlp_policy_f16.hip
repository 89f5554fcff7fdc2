// mlp_policy_f16.hip -- the float16 build of mlp_policy.hip's two kernels (ZENV_MLP_F16): the same source with
// elem_t = _Float16, v_mfma_f32_32x32x16_f16 / v_smfmac_f32_32x32x32_f16 and v_cvt_pk_f16_f32.  Exports
// launch_mlp_forward_f16 only (the kernels have internal linkage; the packer and the experience kernels exist once, in
// the bf16 build).
#define MLP_ELEM_F16 1
#include "mlp_policy.hip"
