// Instantiations of the MFMA cell kernel for kernel_size = 9 (one translation unit per size so the
// build can compile them in parallel).
#include "xna_mfma_kernel.h"

int naf_xna_mfma_launch_k9(const XnaMfmaParams& p, int dvt, int out_dtype, hipStream_t s) {
    return xna_mfma_launch_ks<9>(p, dvt, out_dtype, s);
}
