// geodesic_kernels.h -- internal interface between the C-ABI layer (bhgeo_capi.hip) and the
// gfx950 kernels (geodesic_kernels.hip).  Not installed; the public surface is include/bhgeo.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Every launcher returns hipGetLastError() as THIS launch's status.  That call reports -- and clears -- the last error any
// earlier HIP call of the thread left behind, ours (a refused allocation) or another library's (a probing
// hipPointerGetAttributes): it is read away before the launch, so that a launch that went through is not reported
// with somebody else's error.
#define BHG_LAUNCH(...)                    \
    do {                                   \
        (void)hipGetLastError();           \
        hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

namespace bhg {

// mirrors of the public constants (include/bhgeo.h); static_asserts in bhgeo_capi.hip tie them
constexpr uint32_t BHG_FLAG_HIT_HORIZON_ = 1u;
constexpr uint32_t BHG_FLAG_START_INSIDE_ = 2u;
constexpr uint32_t BHG_FLAG_REACHED_END_ = 4u;
constexpr uint32_t BHG_FLAG_EXITED_SPHERE_ = 8u;
constexpr uint32_t BHG_FLAG_MAX_STEPS_ = 16u;
constexpr uint32_t BHG_FLAG_STEP_TOO_SMALL_ = 32u;
constexpr uint32_t BHG_FLAG_NAN_ = 64u;
constexpr uint32_t BHG_FLAG_HIT_DISK_ = 128u;
constexpr uint32_t BHG_FLAG_HIT_OBJECT_ = 0x88u;
constexpr int BHG_MAX_SPHERES_ = 8;
constexpr int BHG_METHOD_DP54_ = 0;
constexpr int BHG_METHOD_RK4_ = 1;
constexpr int BHG_RHS_CHRISTOFFEL_ = 0;
constexpr int BHG_RHS_REDUCED_ = 1;
constexpr int BHG_RHS_CHRISTOFFEL_TL_ = 3;   // internal: the Christoffel form with g(k, k) = -1 (bhg_params.time_like), own translation unit
constexpr int BHG_RHS_KERR_BL_ = 2;
// rays per trace launch: the kernels form a ray's result offsets (idx * 48 at most) in 32 bits
constexpr uint64_t BHG_MAX_RAYS_PER_LAUNCH = 1ull << 26;

// Kernel arguments (passed by value -> SGPRs).  All pointers are device addresses.
// Fields the step loop reads come first, in order of use, the ones only the queue fill and the event drain read
// behind them (the kernarg segment is fetched in 16-dword chunks and, short of SGPRs, the compiler spills and reloads
// scalars through VGPR lanes; the order was measured neutral on every configuration -- it is kept for the reader).
struct TraceArgs {
    // ---- chunk 0 (16 dwords): every iteration of the step loop
    double *end;                 // [n][6]
    uint8_t *flags;              // [n] (never null inside the kernels: the C-ABI layer substitutes a workspace); only ever holds final values
    uint32_t *n_steps;           // [n] (never null inside the kernels, like flags)
    uint32_t *n_accepted;        // [n] (ditto)
    double rtol, atol, lambda_end, max_step;
    // ---- chunk 1: the step loop's event tests and controller limits
    double r_hor;                // horizon event radius: r_s, or r_plus (1 + margin) for Kerr
    double r_s;
    double min_step_cap;         // >= 10 ulp(t) for all t in [0, lambda_end]
    double r_exit, disk_r_in, disk_r_out;
    double spin;                 // Kerr a (BHG_RHS_KERR_BL_)
    double mu2;                  // -g(k, k): 0 null rays, 1 time-like (read by the Kerr start conversion only; the Cartesian
                                 // time-like form is a right-hand side of its own, BHG_RHS_CHRISTOFFEL_TL_)
    double h_fixed;
    // ---- chunk 2: parking (in the step loop's event branch), then the rare paths
    double *end_dir;             // nullptr, or [n][3]: FINAL states are written as their direction half only, here (end then
                                 // only holds the parked / resume records of rays that need them: a workspace)
    double *ws;                  // [n][ws_stride] per-ray records: prepare {a0, h0, r0, 0, E, L}, park {a1, t, h, h_next, E, L}, resume {a, h, r, t, E, L}
                                 // (park / resume records are written and read back by ONE wavefront of the trace kernel)
    uint32_t max_steps;
    int32_t ws_stride;           // doubles per ray record in ws: 6, or 8 for Kerr ({E, L} appended)
    int32_t n_spheres;           // object spheres inside the curved region (Schwarzschild forms only)
    int32_t from_records;        // rays start from the records the prepare pass wrote (Kerr)
    const double *k0;            // [n][3]
    const double *x0;            // [n][3] or nullptr -> x0s
    unsigned long long *counter; // 8 slice counters (256 B apart), zero at launch
    unsigned long long *counter_next; // the set the NEXT launch of this context will use: this launch zeroes it
    uint64_t n;                  // rays in the call
    double k0s[3];               // the direction of a ONE-ray trajectory call (k0 == nullptr)
    double x0s[3];
    int32_t inline_prepare;      // set by the launcher: no prepare launch, the trace waves work the start records out (Schwarzschild forms)
    int32_t order_blocks;        // work-order hint: n = order_blocks * order_block_len, batches are
    uint64_t order_block_len;    // handed out chunk-major over the blocks; 0/1 = plain order
    int8_t *object_id;           // [n] or nullptr: sphere index of rays that end with BHG_FLAG_HIT_OBJECT, else -1
    unsigned long long *diag;    // diagnostic builds only (BHG_DIAG): [grid][8] per-wave stamps; + BHG_DIAG_HIST: event-coherence histogram
    uint32_t dbg_idx;            // diagnostic builds: ray whose controller trace is logged
    double spheres[BHG_MAX_SPHERES_][4];  // {cx, cy, cz, radius}, BH-centred
};

#define BHG_DIAG_HIST 400000   // u64 index into `diag`: H[0..64] iterations by lanes parking ONE short event, [65..129] the lanes
                               // stepping in those iterations, [130] all parked steps (zeroed after every dump)

// camera-ray generation (frame_kernels.hip)
struct RaygenArgs {
    const double *jitter;   // [S*H*W*2] MT19937 doubles, sample-major then row-major pixels, (u1, u2); nullptr = pixel
                            // centres (u1 = u2 = 1/2)
    int32_t compact;        // 1: the stream holds draws for the listed pixels only, [S][n_pixels][2] in list order (a
                            // mark window: the engine draws inside the window only, RelativisticRenderEngine.py:219)
    const int64_t *pixels;  // [n_pixels] flat pixel ids y*W+x, or nullptr = all pixels in order
    double *k0;             // [S*n_pixels][3], ray i = s*n_pixels + p
    uint64_t n_pixels;
    int32_t width, height, samples, rotate;
    double fov_x, fov_y;
    double rot[9];          // row-major camera rotation
};

// shading + per-pixel multisample mean (frame_kernels.hip)
struct ShadeArgs {
    const double *end;     // [S*n_pixels][6], or nullptr when dir is given
    const double *dir;     // [S*n_pixels][3] exit directions alone (sky-only scenes), or nullptr
    const uint8_t *flags;  // [S*n_pixels]
    const float *sky;      // [sky_h][sky_w][4] RGBA float32, equirectangular
    double *rgba;          // [n_pixels][4] fp64, or nullptr
    float *rgba_f32;       // [n_pixels or frame pixels][4] fp32, or nullptr
    const int64_t *scatter;  // nullptr, or where pixel p goes in rgba_f32
    uint64_t n_pixels;
    int32_t samples, sky_w, sky_h;
    // scene shading (bhg_shade_scene_device); all zero / null for the sky-only call
    const int8_t *object_id;  // [S*n_pixels] or nullptr
    const float *disk_tex;    // [disk_h][disk_w][4] RGBA float32 or nullptr (white)
    int32_t disk_w, disk_h;
    double disk_r_in, disk_r_out, disk_phase, disk_mean, disk_stddev, disk_intensity;
    int32_t n_spheres, n_lamps;
    double spheres[BHG_MAX_SPHERES_][4];
    double sphere_rgb[BHG_MAX_SPHERES_][3];
    double lamps[4][4];  // {x, y, z, intensity}
};

hipError_t launch_raygen(const RaygenArgs &a, hipStream_t s);
hipError_t launch_shade(const ShadeArgs &a, hipStream_t s);
hipError_t launch_split_end(const double *end, uint64_t n, double *loc, double *dir, hipStream_t s);
hipError_t launch_gather_rows4(const float *src, const int64_t *index, uint64_t n, float *dst, hipStream_t s);

// ev: nullptr, or 3 events recorded around prepare | trace on stream s
// evt: bit 0 = sphere-exit event compiled in, bit 1 = disk-plane event, bit 2 = object spheres (then all three)
hipError_t launch_trace(const TraceArgs &a, int method, int rhs, int evt, int grid, hipStream_t s, hipEvent_t *ev);
hipError_t trace_occupancy(int method, int rhs, int evt, int *blocks_per_cu);
// does a trace launch of this right-hand side read per-ray prepare records from TraceArgs::ws (Kerr always; every form
// when the kernels were built with -DBHG_INLINE_PREPARE=0)?  The C-ABI layer sizes the workspace by it.
bool needs_prepare_ws(int rhs);
// Kerr: after the last pass of a call, Boyer-Lindquist end states -> Cartesian
hipError_t launch_kerr_finalize(const TraceArgs &a, double *dir_out, hipStream_t s);
// the Kerr instantiations live in their own translation unit (geodesic_kernels_kerr.hip: same source, same flags --
// machine LICM off, see the Makefile -- compiled on its own)
hipError_t launch_trace_kerr(const TraceArgs &a, int method, int evt, int grid, hipStream_t s, hipEvent_t *ev);
hipError_t trace_occupancy_kerr(int method, int evt, int *blocks_per_cu);
hipError_t launch_trajectory_kerr(const TraceArgs &a, int method, double *traj, uint32_t *n_valid, uint32_t T, hipStream_t s);
// ... and so does the time-like Christoffel form (geodesic_kernels_timelike.hip; one event variant)
hipError_t launch_trace_timelike(const TraceArgs &a, int method, int grid, hipStream_t s, hipEvent_t *ev);
hipError_t trace_occupancy_timelike(int method, int *blocks_per_cu);
hipError_t launch_trajectory_timelike(const TraceArgs &a, int method, double *traj, uint32_t *n_valid, uint32_t T, hipStream_t s);
hipError_t launch_accel_timelike(const double *x, const double *k, double r_s, uint64_t n, double *acc, hipStream_t s);
// prepare + one-lane-per-ray sampled trajectories (+ Kerr finalize); traj [n][6][T], n_valid [n]
hipError_t launch_trajectory(const TraceArgs &a, int rhs, int method, double *traj, uint32_t *n_valid, uint32_t T, hipStream_t s);
// does a trajectory call of n rays run one wave per ray (the kernel then prepares the ray itself and NaN-fills the tail)?
bool trajectory_wave_per_ray(uint64_t n);
// rhs = Kerr: x, k and acc are Boyer-Lindquist (r, theta, phi) triples, E and L fixed by the null condition at each point
hipError_t launch_accel(const double *x, const double *k, double r_s, double spin, double mu2, uint64_t n, double *acc, int rhs,
                        hipStream_t s);
hipError_t launch_accel_kerr(const double *x, const double *k, double r_s, double spin, double mu2, uint64_t n, double *acc, hipStream_t s);

// roofline calibration probes (probe_kernels.hip): kind 0 = pure v_fma_f64, 1 = the DP5(4) step loop's instruction mix
hipError_t launch_probe(int kind, int grid, uint32_t iters, double *out, hipStream_t s);
void probe_shape(int kind, uint32_t *valu_per_iter, uint32_t *quarter_per_iter);

}  // namespace bhg
