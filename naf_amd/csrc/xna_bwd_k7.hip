// Instantiations of the attention backward cell kernels for kernel_size = 7 (wave-specialised kernel where it serves the shape).
#include "xna_bwd2_kernel.h"

int naf_xna_bwd_launch_k7(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd2_launch_ks<7>(p, Dv, s); }
