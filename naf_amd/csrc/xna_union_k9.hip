// Instantiations of the table-driven MFMA attention kernel for kernel_size = 9.
#include "xna_union_kernel.h"

int naf_xna_union_launch_k9(const XnaUnionParams& p, int wt, int out_dtype, size_t lds, hipStream_t s) { return xna_union_launch_ks<9>(p, wt, out_dtype, lds, s); }
