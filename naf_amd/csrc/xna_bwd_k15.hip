// Instantiations of the attention backward cell kernel for kernel_size = 15 (channel chunks of at most 64: xna_bwd.hip).  The single-sweep
// form of the kernel crashed hipcc 7.2 in 'AMDGPU Rewrite AGPR-Copy-MFMA' here (and spilled 233 registers with MFMA results in AGPRs);
// the swept form (kTwoSweep in xna_bwd_kernel.h) compiles in the library's VGPR form without scratch (-Rpass-analysis=kernel-resource-usage).
#include "xna_bwd2_kernel.h"

int naf_xna_bwd_launch_k15(const XnaBwdParams& p, int Dv, hipStream_t s) {
    static const bool big8 = [] { const char* e = naf_knob("NAF_BWD_BIG8"); return !(e != nullptr && atoi(e) == 0); }();
    switch (Dv) {
        case 32: return big8 ? xna_bwd2_launch_one<15, 32>(p, s) : xna_bwd_launch_one<15, 32>(p, s);      // the eight-wave kernel (NAF_BWD_BIG8=0: the four-wave one, chunks of 64)
        case 64: return big8 ? xna_bwd2_launch_one<15, 64>(p, s) : xna_bwd_launch_one<15, 64>(p, s);
    }
    naf_set_error("naf_xna_bwd: 15 x 15 windows run in channel chunks of 32 or 64, got %d", Dv);
    return NAF_ERR_UNSUPPORTED;
}
