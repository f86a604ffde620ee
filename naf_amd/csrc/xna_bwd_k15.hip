// Instantiations of the attention backward cell kernel for kernel_size = 15 (channel chunks of at most 64: xna_bwd.hip).  The single-sweep
// form of the kernel crashed hipcc 7.2 in 'AMDGPU Rewrite AGPR-Copy-MFMA' here (and spilled 233 registers with MFMA results in AGPRs);
// the swept form (kTwoSweep in xna_bwd_kernel.h) compiles in the library's VGPR form without scratch (-Rpass-analysis=kernel-resource-usage).
#include "xna_bwd_kernel.h"

int naf_xna_bwd_launch_k15(const XnaBwdParams& p, int Dv, hipStream_t s) {
    switch (Dv) {
        case 32: return xna_bwd_launch_one<15, 32>(p, s);
        case 64: return xna_bwd_launch_one<15, 64>(p, s);
    }
    naf_set_error("naf_xna_bwd: 15 x 15 windows run in channel chunks of 32 or 64, got %d", Dv);
    return NAF_ERR_UNSUPPORTED;
}
