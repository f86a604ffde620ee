// Instantiations of the attention backward cell kernels for kernel_size = 9 (wave-specialised kernel where it serves the shape).
#include "xna_bwd2_kernel.h"

int naf_xna_bwd_launch_k9(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd2_launch_ks<9>(p, Dv, s); }
