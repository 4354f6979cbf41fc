// geodesic_kernels_kerr.hip -- the Kerr (Boyer-Lindquist) instantiations of the kernels in geodesic_kernels.hip,
// as their own translation unit so that they can be built with different code-generation flags (see Makefile:
// machine LICM stays ON here -- the Kerr kernels run 2 waves/SIMD with registers to spare and are 3-4 % faster
// with hoisted constants -- and OFF for the Schwarzschild unit, where hoisting cost 38 SGPR spills).
#define BHG_TU_KERR 1
#include "geodesic_kernels.hip"
