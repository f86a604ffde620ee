// Device-side twin of naf_axis_index_table: the same rule (naf_common.h), one thread per table entry, so that a
// host that never touches the CPU tables (naf_forward, graph capture) gets bit-identical ones.
#include "naf_common.h"

__global__ __launch_bounds__(256) void axis_table_kernel(int32_t* __restrict__ out, int L_out, int L_in, int k) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L_out * k) return;
    const int i = e / k, t = e - i * k;
    const int dil = L_out / L_in;
    out[e] = naf_nearest_exact_src(naf_window_start(i, L_out, k, dil) + t * dil, L_in, L_out);
}

int naf_launch_axis_table(int32_t* out_dev, int L_out, int L_in, int k, hipStream_t s) {
    const int64_t n = (int64_t)L_out * k;
    if (n > 0x7fffffffLL) {
        naf_set_error("naf_axis_index_table_device: table of %lld entries out of range", (long long)n);
        return NAF_ERR_INVALID;
    }
    hipLaunchKernelGGL(axis_table_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, out_dev, L_out, L_in, k);
    return naf_check_launch("axis_table_kernel");
}
