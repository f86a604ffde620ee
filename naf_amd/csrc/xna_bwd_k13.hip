// Instantiations of the attention backward cell kernel for kernel_size = 13.
#include "xna_bwd_kernel.h"

int naf_xna_bwd_launch_k13(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd_launch_ks<13>(p, Dv, s); }
