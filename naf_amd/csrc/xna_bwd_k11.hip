// Instantiations of the attention backward cell kernels for kernel_size = 11 (wave-specialised kernel where it serves the shape: Dv <= 128).
#include "xna_bwd2_kernel.h"

int naf_xna_bwd_launch_k11(const XnaBwdParams& p, int Dv, hipStream_t s) { return xna_bwd2_launch_ks<11>(p, Dv, s); }
