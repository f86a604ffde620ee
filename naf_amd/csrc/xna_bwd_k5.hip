// Instantiations of the attention backward cell kernel for kernel_size = 5.
#include "xna_bwd_kernel.h"

int naf_xna_bwd_launch_k5(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd_launch_ks<5>(p, Dv, s); }
