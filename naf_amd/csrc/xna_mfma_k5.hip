// Instantiations of the MFMA cell kernel for kernel_size = 5 (one translation unit per size so the
// build can compile them in parallel).
#include "xna_mfma_kernel.h"

int naf_xna_mfma_launch_k5(const XnaMfmaParams& p, const XnaMfmaPlan& pl, int out_dtype, hipStream_t s) {
    return xna_mfma_launch_ks<5>(p, pl, out_dtype, s);
}
