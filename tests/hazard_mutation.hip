// TEST INFRASTRUCTURE (tests/test_host_logic.py::test_hazard_lint_catches_the_pattern_compiled_from_source): OUR OWN kernel of
// round 3 with the scalar-load hazard, and its fixed form, compiled to ISA by the test and handed to csrc/check_scalar_hazards.py --
// the lint must flag the first and pass the second with TODAY's compiler, not only on the ISA kept from round 3.
#include <hip/hip_runtime.h>
// round 3's k_colscale_after_truncate, verbatim in spirit: one thread carries cs[m] over to cs[p] and clears the rest
__global__ void k_colscale_r03(double *cs, int m, int p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double sm = cs[m];
        for (int c = 0; c <= m; ++c) cs[c] = 0.0;
        cs[p] = sm;
    }
}
// the fixed form: vector loads in every lane, a barrier, one lane per column
__global__ void k_colscale_fixed(double *cs, int m, int p) {
    __shared__ double s_sm;
    const int t = threadIdx.x;
    if (blockIdx.x != 0) return;
    const double mine = t <= m ? cs[t] : 0.0;
    if (t == m) s_sm = mine;
    __syncthreads();
    if (t <= m) cs[t] = (t == p) ? s_sm : 0.0;
}
