// Instantiations of the sliding-window attention kernel for kernel_size = 9.
#include "xna_slide_kernel.h"

int naf_xna_slide_launch_k9(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s) { return xna_slide_launch_ks<9>(sp, dvt, out_dtype, s); }
