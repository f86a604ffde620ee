// Instantiations of the attention backward cell kernel for kernel_size = 11.
#include "xna_bwd_kernel.h"

int naf_xna_bwd_launch_k11(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd_launch_ks<11>(p, Dv, s); }
