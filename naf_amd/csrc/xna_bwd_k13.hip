// Instantiations of the attention backward cell kernels for kernel_size = 13: the eight-wave kernel (chunks of <= 64 value channels: bwd_chunk_limit in
// xna_bwd.hip) and, behind NAF_BWD_BIG8=0, the four-wave kernel (chunks of <= 128).
#include "xna_bwd2_kernel.h"

int naf_xna_bwd_launch_k13(const XnaBwdParams& p, int Dv, hipStream_t s) {
    static const bool big8 = [] { const char* e = naf_knob("NAF_BWD_BIG8"); return !(e != nullptr && atoi(e) == 0); }();
    return big8 ? xna_bwd2_launch_ks<13>(p, Dv, s) : xna_bwd_launch_ks<13>(p, Dv, s);
}
