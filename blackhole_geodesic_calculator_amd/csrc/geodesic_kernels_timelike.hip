// geodesic_kernels_timelike.hip -- the time-like (massive-particle, g(k, k) = -1) instantiations of the Cartesian Christoffel
// kernels in geodesic_kernels.hip, as their own translation unit: the constructor argument of the reference's solver object,
// GeodesicIntegratorSchwarzschild(mass=, time_like=, verbose=) (raytracer/RelativisticRenderEngine.py:134), has a second
// value the engine never passes.  Same source, RHS id BHG_RHS_CHRISTOFFEL_TL_, one event variant (see launch_trace_timelike).
#define BHG_TU_TIMELIKE 1
#include "geodesic_kernels.hip"
