// geodesic_kernels_kerr.hip -- the Kerr (Boyer-Lindquist) instantiations of the kernels in geodesic_kernels.hip,
// as their own translation unit: they compile for minutes and want their own occupancy (2 waves per SIMD), and a unit
// of their own keeps the door open for different code-generation flags.  Today the flags are the SAME as the
// Schwarzschild unit's (Makefile: machine LICM OFF for every kernel unit -- with the event drain inside the trace kernels
// the hoisted literals cost this unit 28 spilled doubles per step; up to round 3 it was built with LICM on).
#define BHG_TU_KERR 1
#include "geodesic_kernels.hip"
