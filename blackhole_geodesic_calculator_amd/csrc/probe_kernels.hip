// probe_kernels.hip -- calibration of the fp64 VALU roofline on THE device the trace kernels run on (bhg_peak_probe).
//
// The trace kernels are priced against the vendor's fp64 vector peak (256 CU x 128 flop/clk x 2.4 GHz = 78.6 TFLOP/s,
// SURVEY.md section 8d).  The boxes of a pool differ by several per cent in what they sustain (clock, power cap), and a
// reader of one bench line cannot tell a slower kernel from a slower box.  Two probes, launched with the trace kernels'
// own geometry (one wave64 per workgroup, 12 resident waves per CU, persistent):
//
//   kind 0  nothing but v_fma_f64: eight independent dependency chains per lane, 512 FMAs per loop iteration
//           -> what the fp64 pipe of this box delivers at full issue ("fp64_fma_tflops_measured");
//   kind 1  the step loop's instruction MIX: per 503 VALU instructions 8 v_rcp_f64 + 8 v_rsq_f64 (quarter rate: the
//           reciprocals and reciprocal square roots of a DP5(4) step's seven right-hand sides) among 487 v_fma_f64
//           -> the issue-bound ceiling of a kernel with that mix ("issue_bound").
//
// Neither touches memory inside its loop; each lane writes one double at the end so that nothing is optimised away.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geodesic_kernels.h"

namespace bhg {

namespace {

constexpr int CHAINS = 8;

// 64 rounds over the eight chains = 512 FMAs per call; a and b are lane values the compiler cannot fold
#define BHG_FMA_ROUND(x, a, b)                 \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], a, b)

__global__ __launch_bounds__(64) void probe_fma_kernel(uint32_t iters, double a, double b, double *out)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = 1.0 + 1e-3 * (double)(threadIdx.x + 64 * c);
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            BHG_FMA_ROUND(x, a, b);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}

// 503 VALU instructions per iteration: 16 groups of {1 quarter-rate op, 30 FMAs} = 496, + 7 FMAs.  The transcendental's
// argument comes from one chain and its result goes into another, as in the right-hand side (r^2 -> 1/r -> ...).
__global__ __launch_bounds__(64) void probe_mix_kernel(uint32_t iters, double a, double b, double *out)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = 1.0 + 1e-3 * (double)(threadIdx.x + 64 * c);
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const double t = (g & 1) ? __builtin_amdgcn_rsq(x[g % CHAINS]) : __builtin_amdgcn_rcp(x[g % CHAINS]);
            // 30 FMAs: the first takes the transcendental into the next chain, then three rounds of eight and five more
            x[(g + 1) % CHAINS] = __builtin_fma(t, x[(g + 1) % CHAINS], b);
            BHG_FMA_ROUND(x, a, b);
            BHG_FMA_ROUND(x, a, b);
            BHG_FMA_ROUND(x, a, b);
#pragma unroll
            for (int c = 0; c < 5; c++) x[c] = __builtin_fma(x[c], a, b);
        }
#pragma unroll
        for (int c = 0; c < 7; c++) x[c] = __builtin_fma(x[c], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}

}  // namespace

// instructions per loop iteration of each probe: {VALU total, of which quarter-rate}
void probe_shape(int kind, uint32_t *valu_per_iter, uint32_t *quarter_per_iter)
{
    if (kind == 0) {
        *valu_per_iter = 512;
        *quarter_per_iter = 0;
    } else {
        *valu_per_iter = 16 * 31 + 7;   // 16 groups of {1 quarter-rate op, 30 FMAs} + 7 FMAs = 503, of which 16 quarter-rate
        *quarter_per_iter = 16;
    }
}

hipError_t launch_probe(int kind, int grid, uint32_t iters, double *out, hipStream_t s)
{
    // a, b: a contraction towards 1 (x <- 0.999 x + 0.001), so that every chain stays finite for any iteration count
    const double a = 0.999, b = 0.001;
    if (kind == 0) BHG_LAUNCH(probe_fma_kernel, dim3(grid), dim3(64), 0, s, iters, a, b, out);
    else BHG_LAUNCH(probe_mix_kernel, dim3(grid), dim3(64), 0, s, iters, a, b, out);
    return hipGetLastError();
}

}  // namespace bhg
