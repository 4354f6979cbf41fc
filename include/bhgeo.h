/*
 * bhgeo.h -- C ABI of libbhgeo.so, the MI355X (gfx950) null-geodesic ray integrator.
 *
 * This is the drop-in boundary for ONE path of bldevries/blackhole_geodesic_calculator: the
 * per-ray geodesic solve the Blender render engine hands to the third-party `curvedpy` package.
 * The reference has no FFI today -- the boundary is a Python method call per ray:
 *
 *     self.GeoInt = curvedpy.GeodesicIntegratorSchwarzschild(mass=, time_like=False, verbose=False)
 *                                                   raytracer/RelativisticRenderEngine.py:134
 *     k_xyz, x_xyz, result = self.GeoInt.calc_trajectory(k0_xyz, x0_xyz, max_step=, curve_end=,
 *                                                   nr_points_curve=10000, verbose=False)
 *                                                   raytracer/RelativisticRenderEngine.py:293-294
 *
 * and, batched per frame, the arrays of a pre-traced camera:
 *
 *     cam.ray_blackhole_hit[iy, ix], cam.ray_end[iy, ix, 3:6]
 *                                                   raytracer/RelativisticRenderEngineCamEdition.py:225-228
 *
 * The entry points below are what a ctypes binding for that path binds (INTEGRATION.md shows the
 * stub).  Plain pointers and sizes only; no torch / numpy types.  All floating point is IEEE fp64.
 *
 * Array layouts (row-major, C-contiguous):
 *     k0    [n][3]   initial spatial direction k^i   (k0_xyz, RelativisticRenderEngine.py:287)
 *     x0    [3]      shared origin, BH-centred       (x0_xyz, :278, :288)   -- or [n][3] per ray
 *     end   [n][6]   {x, y, z, k_x, k_y, k_z} at the end of the curve
 *                    (= x_xyz[:, -1], k_xyz[:, -1] at :307-308; = ray_end[..., 0:6] of the Cam edition)
 *     flags [n]      BHG_FLAG_* bits (result['hit_blackhole'], result['start_inside_hole'], :296-297)
 *     n_steps [n]    attempted RK steps (accepted + rejected)
 *     n_accepted [n] accepted RK steps
 *
 * Threading: one bhg_context per device; calls on one context must not overlap (contexts of different
 * threads may run at the same time, on one device or several).  Host-buffer
 * calls block until the results are in the caller's buffers.  Device-buffer calls enqueue on the
 * given HIP stream and return; the library keeps no pointer after the call's work completes.
 * Device buffers need only the alignment of their element type (8 bytes for the doubles, 4 for the counts).
 *
 * Errors: every int-returning function returns BHG_OK (0) or a negative BHG_E_* code;
 * bhg_last_error() gives a thread-local message for the last failure.  A refused device allocation is
 * BHG_E_NOMEM and leaves the context usable.  The status of a call is its own: an error another caller of the
 * HIP runtime left behind on the thread (hipGetLastError() is sticky) is not reported, and the library leaves none
 * of its own behind.  There is no CPU
 * fallback: without a usable gfx950 device the calls fail with BHG_E_NO_DEVICE.
 */
#ifndef BHGEO_H
#define BHGEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHG_ABI_VERSION 8   /* 8: bhg_trajectory_objects (sampled curves that end on object spheres); nothing of ABI 7 changed.
                               7: the binder's handshake -- bhg_abi_check, bhg_*_size, bhg_default_params_sized; bhg_peak_probe.
                               6: bhg_frame_* (library-owned frame, N devices), bhg_deal_tiles, bhg_params.time_like (104 bytes) */

#define BHG_ABI_COMPAT_MIN 7 /* bhg_abi_check serves bindings written for this ABI or later: every symbol, struct layout and
                                meaning they know is unchanged (ABI 8 only ADDED an entry point) */

/* return codes */
#define BHG_OK 0
#define BHG_E_INVALID (-1)   /* bad argument (NULL pointer, non-finite / negative parameter) */
#define BHG_E_NO_DEVICE (-2) /* no HIP device / device index out of range */
#define BHG_E_HIP (-3)       /* a HIP runtime call failed; see bhg_last_error() */
#define BHG_E_NOMEM (-4)     /* device allocation failed */

/* per-ray flag bits */
#define BHG_FLAG_HIT_HORIZON 1u     /* result['hit_blackhole']  (RelativisticRenderEngine.py:297) */
#define BHG_FLAG_START_INSIDE 2u    /* result['start_inside_hole'] (:296, :311-313) */
#define BHG_FLAG_REACHED_END 4u     /* lambda reached curve_end */
#define BHG_FLAG_EXITED_SPHERE 8u   /* crossed r = r_exit outward (LimitedRelativisticRenderEngine.py:273-278) */
#define BHG_FLAG_MAX_STEPS 16u      /* attempted-step cap hit */
#define BHG_FLAG_STEP_TOO_SMALL 32u /* scipy's failure mode (rk.py:132-133) */
#define BHG_FLAG_NAN 64u            /* non-finite end state */
#define BHG_FLAG_HIT_DISK 128u      /* crossed z = 0 inside the annulus (LimitedRelativisticRenderEngine.py:413-438) */
#define BHG_FLAG_HIT_OBJECT 0x88u   /* ended on an object sphere (bhg_trace_objects*): the reference's collision stub
                                     * "NOW YOU DO COLLISION DETECTION", hit = False (RelativisticRenderEngine.py:304-305).
                                     * A composite value (EXITED_SPHERE | HIT_DISK never occur together otherwise):
                                     * test with (flags & 0x88) == 0x88 */
#define BHG_MAX_SPHERES 8

/* integrators */
#define BHG_METHOD_DP54 0 /* Dormand-Prince 5(4) with scipy RK45's controller (README.md:196) */
#define BHG_METHOD_RK4 1  /* classic fixed-step RK4 with step h_fixed */

/* right-hand-side formulations (algebraically identical for null rays) */
#define BHG_RHS_CHRISTOFFEL 0 /* -Gamma^i_{mu nu} k^mu k^nu, k^t from the null condition (README.md:198-209) */
#define BHG_RHS_REDUCED 1     /* -(3/2) r_s |x cross k|^2 x / r^5 (regular at the horizon) */
#define BHG_RHS_KERR_BL 2     /* Kerr, Boyer-Lindquist Christoffels (README.md:218 goal; `a = 0.9`,
                                 RelativisticRenderEngineCamEdition.py:210).  Same boundary: Cartesian x0, k0 in,
                                 Cartesian end state out (x = sqrt(r^2+a^2) sin th cos ph, z = r cos th); integrated
                                 in (r, theta, phi) with k^t from the Killing constants; the horizon event sits at
                                 r_plus (1 + BHG_KERR_HORIZON_MARGIN) because the coordinates are singular at r_plus */
#define BHG_KERR_HORIZON_MARGIN 1e-3

typedef struct bhg_params {
    double r_s;         /* horizon radius = 2*mass                 (RelativisticRenderEngine.py:95) */
    double lambda_end;  /* curve_end                               (:62, :294) */
    double max_step;    /* max_step; +inf for "unset" (-1)         (:57-60) */
    double rtol;        /* DP54 relative tolerance, scipy default 1e-3; below 100 eps it is raised to 100 eps, as solve_ivp does (_ivp/common.py:44-51) */
    double atol;        /* DP54 absolute tolerance, scipy default 1e-6; must be > 0 (solve_ivp takes 0 too and then fails on the first state
                           component that is exactly zero -- scale = 0 -- with "step size too small": refused here up front) */
    double h_fixed;     /* RK4 step */
    double r_exit;      /* 0 = off; else terminate when r crosses r_exit outward */
    int32_t method;     /* BHG_METHOD_* */
    int32_t rhs_form;   /* BHG_RHS_* */
    uint32_t max_steps; /* cap on attempted steps per ray; 0 = library default (1<<20) */
    uint32_t order_blocks; /* work-order hint, 0 = none: the n rays are this many equal blocks (e.g. the samples of a
                              frame laid out [sample][pixel]) and the library may start the corresponding parts of all
                              blocks together (better balance when the pixels are sorted longest-first); ignored unless
                              n / order_blocks is a multiple of 64.  Never changes a result. */
    double disk_r_in;   /* thin disk in the plane z = 0 (BH-centred frame): the ray ends at its first */
    double disk_r_out;  /* crossing with R_in <= sqrt(x^2+y^2) <= R_out; off when disk_r_out == 0
                           (disk_on / R_in / R_out of LimitedRelativisticRenderEngine.py:283-286).  With
                           BHG_RHS_KERR_BL: the equatorial plane theta = pi/2 (z = r cos theta = 0), the
                           annulus in the cylindrical radius sqrt(r^2 + a^2) */
    double spin;        /* Kerr a in length units, |a| < M = r_s/2 (BHG_RHS_KERR_BL only) */
    int32_t time_like;  /* 0: null geodesics, g(k, k) = 0 -- what the engine asks for (time_like=False, RelativisticRenderEngine.py:134); 1: the
                           constructor argument's other value, massive particles: g(k, k) = -1, lambda is the proper time.
                           With BHG_RHS_CHRISTOFFEL (the norm enters through (k^t)^2 = (|k|^2 + h (n.k)^2 + 1) / f) and
                           BHG_RHS_KERR_BL (through E and L at the start); BHG_RHS_REDUCED is the null closed form and
                           refuses it.  The plotting path (bhg_trajectory, bhg_trace): no frame pipeline asks for it. */
    int32_t reserved0;  /* 0 */
} bhg_params;

typedef struct bhg_context bhg_context;

/* --- library ---------------------------------------------------------------------------- */
int bhg_version(void);                /* BHG_ABI_VERSION */
int bhg_device_count(void);           /* number of HIP devices, 0 if none, never fails */
const char *bhg_last_error(void);     /* thread-local, never NULL */
void bhg_default_params(bhg_params *p); /* engine defaults: r_s=1 (mass 0.5), lambda_end=50,
                                           max_step=inf, rtol=1e-3, atol=1e-6, DP54, Christoffel
                                           (RelativisticRenderEngine.py:506-508; scipy rk.py:85-87).
                                           Writes sizeof(bhg_params) bytes AS THE LIBRARY LAYS THE STRUCT OUT:
                                           a binding checks its own layout first (below) */

/* --- the binder's handshake ---------------------------------------------------------------
 * The reference-side caller is a Python file (raytracer/RelativisticRenderEngine.py:134, :293-294): its binding declares
 * these structs by hand (ctypes), and a struct that grew between library versions (bhg_params: 96 bytes in ABI 5, 104
 * since ABI 6) is then written and read past its end without any error.  A binding therefore calls, once after loading,
 *
 *     bhg_abi_check(<the BHG_ABI_VERSION it was written for>, sizeof its bhg_params, bhg_camera, bhg_scene, bhg_frame_scene)
 *
 * -> BHG_OK, or BHG_E_INVALID with a bhg_last_error() message that names both figures.  A size of 0 = "this binding does
 * not declare that struct".  bhg_*_size() give the library's own sizes; bhg_default_params_sized() is bhg_default_params
 * for a caller that passes its struct's size along and gets BHG_E_INVALID -- and not one byte written -- on a mismatch. */
size_t bhg_params_size(void);
size_t bhg_camera_size(void);
size_t bhg_scene_size(void);
size_t bhg_frame_scene_size(void);
int bhg_abi_check(int abi_version, size_t params_size, size_t camera_size, size_t scene_size, size_t frame_scene_size);
int bhg_default_params_sized(bhg_params *p, size_t params_size);

/* --- context ----------------------------------------------------------------------------
 * Every entry point makes the context's device current for its own HIP calls and restores the calling
 * thread's current device before it returns. */
int bhg_create(int device, bhg_context **out); /* replaces the per-frame solver construction (:134) */
void bhg_destroy(bhg_context *ctx);
int bhg_device_name(bhg_context *ctx, char *buf, size_t buflen);
int bhg_num_cus(bhg_context *ctx);

/* --- the hot path ----------------------------------------------------------------------- */
/* Host buffers.  Batched replacement of N calc_trajectory calls (:293-294): copies k0 (and x0)
 * to the device, integrates all rays, copies end and whichever of flags / n_steps / n_accepted is not NULL
 * back.  x0_is_shared != 0: x0 is [3]; else [n][3].  Blocking.  Internally a pipeline over chunks of 2^20 rays
 * (upload of the next chunk, trace, download of the previous one overlap on three streams); arrays in pageable
 * memory go through a pinned staging ring with multi-threaded host copies, arrays in page-locked memory
 * (bhg_host_alloc, hipHostMalloc, hipHostRegister) are read / written by the copy engines directly. */
int bhg_trace(bhg_context *ctx, const bhg_params *p, const double *x0, int x0_is_shared,
              const double *k0, size_t n, double *end, uint8_t *flags, uint32_t *n_steps,
              uint32_t *n_accepted);

/* Rays resident on the device: the engine's camera rays (RelativisticRenderEngine.py:185-230 -- pinhole + MT19937
 * jitter, rotate, normalise; loop order sample -> row -> column) are generated ON the device from the jitter stream
 * and stay there, so the directions (24 B/ray) never cross PCIe, and a frame loop whose camera does not move (the
 * engine re-seeds identically on every render(), :189) traces the same ray set again and again.  This is the path
 * the frame driver and the pre-traced camera of the Python adaptor use (frame.py, camera.py).
 *   jitter: HOST array of random.random() draws, (u1, u2) per ray, or NULL = pixel centres (u = 1/2: the Cam
 *           edition's camera, CamEdition.py:225-228).  jitter_is_compact = 0: the full-frame stream
 *           [samples][height][width][2]; 1: draws for the listed pixels only, [samples][n_pixels][2] in list order
 *           (a mark window: the engine only draws inside it, :219).
 *   pixels: HOST array of flat pixel ids y * width + x, or NULL = every pixel in row-major order.
 * Ray s * n_pixels + p is sample s of pixel p.  The rays belong to ctx and must be destroyed before it.  An EMPTY pixel list
 * (pixels != NULL, n_pixels = 0: a shard that was dealt no tile) gives a ray set of 0 rays. */
typedef struct bhg_camera {
    int32_t width, height, samples, reserved;
    double fov_x, fov_y;  /* property values fov_x / fov_y of the engine (:504-505) */
    double rot[9];        /* row-major rotation matrix of the camera's Euler angles (:183); identity = unrotated */
    double origin[3];     /* camera position minus the hole's (:278) */
} bhg_camera;
typedef struct bhg_rays bhg_rays;
int bhg_rays_create(bhg_context *ctx, const bhg_camera *cam, const double *jitter, int jitter_is_compact,
                    const int64_t *pixels, size_t n_pixels, bhg_rays **out);
size_t bhg_rays_count(const bhg_rays *rays);
void bhg_rays_destroy(bhg_rays *rays);
/* Trace rays [first, first + n) of the set and bring back only what is asked for (HOST arrays, any may be NULL):
 * end [n][6], or its halves end_loc [n][3] / end_dir [n][3] -- spacetime_ray_cast's return values (:307-308) --
 * flags, n_steps, n_accepted, object_id.  Same pipeline, same kernels and bit-for-bit the same results as bhg_trace on
 * the same directions.  Blocking. */
int bhg_rays_trace(bhg_rays *rays, const bhg_params *p, const double *spheres, int32_t n_spheres, size_t first, size_t n,
                   double *end, double *end_loc, double *end_dir, uint8_t *flags, uint32_t *n_steps,
                   uint32_t *n_accepted, int8_t *object_id);

/* Page-locked host memory for the arrays handed to bhg_trace / bhg_trace_objects: results then arrive by DMA
 * with no host-side copy.  (numpy's own allocations are pageable; the Python adaptor allocates its result arrays
 * here and keeps a pool of them, page-locking being slow.)  ctx may be NULL in bhg_host_free. */
int bhg_host_alloc(bhg_context *ctx, size_t bytes, void **out);
int bhg_host_free(bhg_context *ctx, void *p);

/* Sampled curves, host buffers: what calc_trajectory returns for nr_points_curve samples
 * (RelativisticRenderEngine.py:293-294, :299-302; the trajectory plots of README.md Fig. 5/6).
 * t_eval = linspace(0, lambda_end, n_points); after every accepted step the samples t_eval <= lambda
 * are produced from the step's dense output, as solve_ivp does with t_eval; a ray that ends early (horizon,
 * exit sphere) yields n_valid[i] < n_points samples, the rest of its row is NaN.  traj [n][6][n_points]
 * (rows x, y, z, k_x, k_y, k_z).  end [n][6] / flags [n] (may be NULL): the same end state and flags
 * bhg_trace gives -- with the exit sphere and, since ABI 7, the thin disk (a ray that ends on it: BHG_FLAG_HIT_DISK, the
 * curve sampled up to the crossing, end = the crossing point: what checkHitDisk looks for on the sampled path,
 * LimitedRelativisticRenderEngine.py:284, :413-438).  BHG_METHOD_RK4 (ABI 7): fixed steps h_fixed, the samples on each step's
 * cubic Hermite interpolant (the one the fixed-step kernels locate events on).  At most 2^26 rays per call.  Small-n path: one
 * WAVE per ray up to 2048 rays (a step's samples are shared out over the 64 lanes: the engine's literal call, one ray
 * with 10,000 samples, takes about 0.1 ms), one lane per ray above; the same bits either way.  A `traj` of at most 4 MB in
 * PAGE-LOCKED memory (bhg_host_alloc) is written by the wave-per-ray kernel directly, over PCIe: no copy of the sample
 * block, no host-side split (the Python adaptor allocates it so). */
int bhg_trajectory(bhg_context *ctx, const bhg_params *p, const double *x0, int x0_is_shared, const double *k0,
                   size_t n, uint32_t n_points, double *traj, uint32_t *n_valid, double *end, uint8_t *flags);
/* ... with object spheres in the curved region (ABI 8): the engine's literal per-ray call is exactly where the reference put
 * its collision stub ("NOW YOU DO COLLISION DETECTION", RelativisticRenderEngine.py:293-305).  spheres [n_spheres][4] =
 * {cx, cy, cz, radius}, BH-centred, as in bhg_trace_objects: a ray that enters one ends there with BHG_FLAG_HIT_OBJECT --
 * the same flag, sphere index (object_id [n], -1 = none; may be NULL), entry point and step counts bhg_trace_objects gives
 * for that ray -- and its curve is sampled up to the entry point, NaN behind it.  n_spheres = 0 is bhg_trajectory. */
int bhg_trajectory_objects(bhg_context *ctx, const bhg_params *p, const double *spheres, int32_t n_spheres, const double *x0,
                           int x0_is_shared, const double *k0, size_t n, uint32_t n_points, double *traj, uint32_t *n_valid,
                           double *end, uint8_t *flags, int8_t *object_id);

/* Device buffers (all d_* are device addresses on ctx's device; x0_shared is a HOST [3] array or
 * NULL when d_x0 [n][3] is given).  Enqueues on `stream` (a hipStream_t; NULL = HIP's null
 * stream, as everywhere in HIP; bhg_context_stream() gives the context's own stream) and
 * returns without synchronising -- with or without a disk or objects: ONE persistent launch finishes
 * every ray (events are located and rays that carry on are resumed inside the trace kernel).  Two calls
 * on one context never overlap: they share the context's work counters and workspace, so a call issued on another
 * stream than the previous one is ordered behind it by the library: a call on a CALLER's stream records an event behind
 * itself before it returns, the next call on another stream waits on that event -- the library never touches a caller's
 * stream after the call that was given it has returned, so the caller may destroy it right away; calls that are to run
 * concurrently need a context each).  The launch is not graph-replayable (it consumes and re-arms those counters). */
int bhg_trace_device(bhg_context *ctx, const bhg_params *p, const double *x0_shared,
                     const double *d_x0, const double *d_k0, size_t n, double *d_end,
                     uint8_t *d_flags, uint32_t *d_n_steps, uint32_t *d_n_accepted, void *stream);

/* The same call for a caller that consumes only the DIRECTION half of the end states -- what a sky frame reads of
 * spacetime_ray_cast's return values (end_dir, RelativisticRenderEngine.py:308, :366-378; the flags say which rays
 * hit the hole): d_end_dir [n][3].  The trace kernel writes 24 instead of 48 bytes per ray and bhg_shade_dir_device
 * reads as many; bit-for-bit the directions bhg_trace_device gives.  (Kerr: traced into an internal record array and
 * split off -- same result, no saving.)  bhg_rays_trace takes this path by itself when only end_dir (and flags,
 * counts) are asked for. */
int bhg_trace_dir_device(bhg_context *ctx, const bhg_params *p, const double *x0_shared, const double *d_x0,
                         const double *d_k0, size_t n, double *d_end_dir, uint8_t *d_flags, uint32_t *d_n_steps,
                         uint32_t *d_n_accepted, void *stream);

/* Objects inside the curved region (the reference holds only the stub at
 * RelativisticRenderEngine.py:304-305, "hit = False", and README.md:225 lists it as a goal): up to
 * BHG_MAX_SPHERES spheres, HOST array spheres [n_spheres][4] = {cx, cy, cz, radius} in BH-centred
 * coordinates.  A ray that is outside sphere j at the start of an accepted step and either ends the
 * step inside it, or whose chord between the step ends passes through it while the step's dense output
 * at the chord's closest point lies inside, enters the sphere in that step; the entry point is the root
 * of |x(lambda) - c_j| - radius_j on the dense output (Brent, like every other event).  Of all terminal
 * events of a step the earliest wins.  Such rays end with BHG_FLAG_HIT_OBJECT, end = entry point and
 * direction there, object_id = j; all other rays get object_id -1.  object_id may be NULL.  With
 * BHG_RHS_KERR_BL the spheres are met in this same Cartesian frame (x = sqrt(r^2 + a^2) sin th cos ph, ...): chord rule on
 * the images of the step's ends, root on the image of the dense output.  The device-buffer form only enqueues (one launch), like bhg_trace_device.
 * With n_spheres = 0 they are bhg_trace / bhg_trace_device. */
int bhg_trace_objects(bhg_context *ctx, const bhg_params *p, const double *spheres, int32_t n_spheres,
                      const double *x0, int x0_is_shared, const double *k0, size_t n, double *end,
                      uint8_t *flags, uint32_t *n_steps, uint32_t *n_accepted, int8_t *object_id);
int bhg_trace_objects_device(bhg_context *ctx, const bhg_params *p, const double *spheres, int32_t n_spheres,
                             const double *x0_shared, const double *d_x0, const double *d_k0, size_t n,
                             double *d_end, uint8_t *d_flags, uint32_t *d_n_steps, uint32_t *d_n_accepted,
                             int8_t *d_object_id, void *stream);

/* --- the stages either side of the solve, on device ---------------------------------------- */
/* Camera rays with the reference's multisample jitter (RelativisticRenderEngine.py:185-188,
 * :224-230).  d_jitter [samples*height*width*2]: the random.random() stream after
 * random.seed(sampling_seed) (:189), sample-major, then rows, then columns, (u1, u2) per pixel.
 * d_pixels [n_pixels]: flat pixel ids y*width+x to generate (a GPU's tile shard), or NULL for all
 * pixels in order.  rot9: row-major camera rotation (HOST, may be NULL = identity).
 * Output d_k0 [samples*n_pixels][3], ray index = s*n_pixels + p. */
int bhg_raygen_device(bhg_context *ctx, int32_t width, int32_t height, int32_t samples, double fov_x,
                      double fov_y, const double *rot9, const double *d_jitter, const int64_t *d_pixels,
                      size_t n_pixels, double *d_k0, void *stream);

/* Shade escaping rays against an equirectangular RGBA float32 sky (background_hit, :366-378;
 * horizon rays are black, :242-244) and take the per-pixel mean over the samples (:250).
 * d_end/d_flags are bhg_trace_device outputs for rays laid out [samples][n_pixels];
 * d_rgba [n_pixels][4] fp64, alpha = 1 (:154-155). */
int bhg_shade_device(bhg_context *ctx, const double *d_end, const uint8_t *d_flags, size_t n_pixels,
                     int32_t samples, const float *d_sky, int32_t sky_w, int32_t sky_h, double *d_rgba,
                     void *stream);

/* The same with the scene the later engines add: rays that ended on the thin disk
 * (BHG_FLAG_HIT_DISK) get texture(texture_x, scale) * intensity with the Gaussian radial profile of
 * checkHitDisk (LimitedRelativisticRenderEngine.py:427-436, :300); rays that ended on an object sphere
 * (BHG_FLAG_HIT_OBJECT) get the Lambert point-lamp sum of spacetime_hit (RelativisticRenderEngine.py:
 * 341-363; light paths are straight, shadowed by the other spheres, n.l clamped at 0) times the sphere's
 * colour.  The struct lives on the HOST; d_* members are device addresses.  d_disk_tex may be NULL
 * (white); d_object_id may be NULL when n_spheres is 0. */
typedef struct bhg_scene {
    const float *d_sky;      /* [sky_h][sky_w][4] RGBA float32, equirectangular */
    int32_t sky_w, sky_h;
    const float *d_disk_tex; /* [disk_h][disk_w][4] RGBA float32 */
    int32_t disk_w, disk_h;
    double disk_r_in, disk_r_out;                                  /* 0, 0 = no disk */
    double disk_phase, disk_mean, disk_stddev, disk_intensity;     /* scene.disk_* (:55-58; defaults 0, 0.2, 0.3, 1) */
    int32_t n_spheres, n_lamps;                                    /* <= BHG_MAX_SPHERES, <= 4 */
    double spheres[BHG_MAX_SPHERES][4];                            /* as for bhg_trace_objects */
    double sphere_rgb[BHG_MAX_SPHERES][3];
    double lamps[4][4];                                            /* {x, y, z, intensity} (intensity = 10 at :317) */
} bhg_scene;
int bhg_shade_scene_device(bhg_context *ctx, const double *d_end, const uint8_t *d_flags, const int8_t *d_object_id,
                           size_t n_pixels, int32_t samples, const bhg_scene *scene, double *d_rgba, void *stream);
/* The same, written as float RGBA -- what Blender's layer.rect takes (RelativisticRenderEngine.py:163-164) --
 * and optionally scattered: d_scatter [n_pixels] (or NULL) gives, for each of this call's pixels, its index in
 * d_rgba_f32 (e.g. y*width + x for a GPU's tile shard, so the shard lands in frame order). */
int bhg_shade_scene_f32_device(bhg_context *ctx, const double *d_end, const uint8_t *d_flags,
                               const int8_t *d_object_id, size_t n_pixels, int32_t samples, const bhg_scene *scene,
                               float *d_rgba_f32, const int64_t *d_scatter, void *stream);

/* Sky-only shading + sample mean from exit directions alone (d_end_dir [samples*n_pixels][3] of bhg_trace_dir_device):
 * the same kernel, same filter and same bits as bhg_shade_device / bhg_shade_scene_f32_device on the records those
 * directions are halves of.  d_rgba (fp64 [n_pixels][4]) and / or d_rgba_f32 (float RGBA, optionally scattered by
 * d_scatter) -- either may be NULL, not both. */
int bhg_shade_dir_device(bhg_context *ctx, const double *d_end_dir, const uint8_t *d_flags, size_t n_pixels,
                         int32_t samples, const float *d_sky, int32_t sky_w, int32_t sky_h, double *d_rgba,
                         float *d_rgba_f32, const int64_t *d_scatter, void *stream);

/* Frame end on the root GPU of a sharded frame: the ranks' float RGBA slabs, gathered into one block
 * d_slabs [n_ranks * slab_pixels][4], are put into frame order, d_frame[p] = d_slabs[d_index[p]] for the n_pixels
 * pixels of the frame (d_index: the frame's permutation, computed once by the host from the tile dealing). */
int bhg_assemble_frame_f32_device(bhg_context *ctx, const float *d_slabs, const int64_t *d_index, size_t n_pixels,
                                  float *d_frame, void *stream);

/* --- the whole frame, owned by the library: one process, one or several GPUs, no PyTorch ------------------------
 * Replaces the body of the reference's frame loop as Blender calls it -- render() (RelativisticRenderEngine.py:50) ->
 * render_scene() (:152-168) -> ray_trace() (:172-267) on ONE render thread of ONE process; the author's commented-out
 * mp.Pool (:210-216) marks where the parallelism has to live.  A bhg_frame holds, per listed device, a context, that
 * device's tiles of the image (tile x tile pixels, all samples of a pixel on one device), its camera rays (generated on
 * the device from the MT19937 jitter stream, :185-230), result buffers and the scene's images; bhg_frame_render() runs
 * rays -> trace -> shade + sample mean on every device at once (one host thread enqueues, the devices work
 * concurrently), gathers the devices' float-RGBA slabs onto the FIRST listed device with ONE exchange per frame, puts
 * them into frame order there and copies one [height][width][4] float array back -- what layer.rect takes (:163-164).
 *
 *   devices   device indices, n_devices >= 1.  An index may be repeated (e.g. {0, 0}): the frame is then sharded over
 *             several contexts of ONE GPU and gathered by device-to-device copies -- the only way to run the N > 1 code
 *             path on a one-GPU machine, and bit-for-bit the image N distinct GPUs give (every ray is its own ODE).
 *   jitter    HOST array, the full-frame stream [samples][height][width][2] of random.random() draws after
 *             random.seed(sampling_seed) (:189), or NULL = pixel centres.  Copied; not referenced after the call.
 *   tile      tile edge in pixels (<= 0: 32).
 *   gather    BHG_FRAME_GATHER_AUTO: RCCL in single-process mode (ncclCommInitAll, grouped ncclSend / ncclRecv on the
 *             contexts' streams -- librccl.so is loaded at run time) when the listed devices are distinct and RCCL loads,
 *             else device-to-device copies (hipMemcpyPeerAsync over xGMI; same-device copies for a repeated device);
 *             _COPY / _RCCL force one (RCCL with a repeated device is BHG_E_INVALID; with ONE device it sends the
 *             frame's slab to itself -- the whole gather path on a single GPU, for tests).
 *             BHG_FRAME_GATHER_PEER (never chosen by AUTO): no exchange at all -- every device's shade kernel stores its
 *             pixels straight into the first device's image, in frame order, over xGMI peer access (16-byte stores in
 *             512-byte runs per tile row); no slabs, no gather, no assembly kernel, and the first device has no more to
 *             do than the others.  Needs peer access from every listed device to the first (else BHG_E_HIP).
 * Tiles are dealt cyclically ((tile_x + tile_y) mod n_devices) until bhg_frame_rebalance() re-deals them by the
 * MEASURED cost of the last render (attempted steps per tile, longest-processing-time-first across devices, each
 * device visiting its tiles longest first): the engine renders the same view sample after sample and frame after
 * frame (:242-250), so the last pass prices the next.  Results never depend on the dealing.
 * Threading: like a context -- one call at a time per frame. */
#define BHG_FRAME_GATHER_AUTO 0
#define BHG_FRAME_GATHER_COPY 1
#define BHG_FRAME_GATHER_RCCL 2
#define BHG_FRAME_GATHER_PEER 3
#define BHG_FRAME_GATHER_COPY_PEERCALL 4   /* _COPY with hipMemcpyPeerAsync on EVERY pair, contexts of one device included (bhg_frame_info
                                              reports _COPY): the N-device copy call, its arguments and stream order on a one-GPU box */
typedef struct bhg_frame bhg_frame;
/* The scene of a frame; everything lives on the HOST and is copied by bhg_frame_set_scene (images are uploaded to
 * every device on the next render).  Members as in bhg_scene.  sky = NULL keeps the current sky image (the first call
 * must bring one); disk_tex = NULL keeps the current disk texture (white if there never was one). */
typedef struct bhg_frame_scene {
    const float *sky;        /* [sky_h][sky_w][4] RGBA float32, equirectangular, rows bottom-up (v = -1 is row 0) */
    int32_t sky_w, sky_h;
    const float *disk_tex;   /* [disk_h][disk_w][4] */
    int32_t disk_w, disk_h;
    double disk_r_in, disk_r_out;                                /* 0, 0 = no disk; must equal the trace parameters' */
    double disk_phase, disk_mean, disk_stddev, disk_intensity;
    int32_t n_spheres, n_lamps;                                  /* <= BHG_MAX_SPHERES, <= 4 */
    double spheres[BHG_MAX_SPHERES][4];                          /* BH-centred {cx, cy, cz, radius} */
    double sphere_rgb[BHG_MAX_SPHERES][3];
    double lamps[4][4];                                          /* {x, y, z, intensity} */
} bhg_frame_scene;
int bhg_frame_create(const int32_t *devices, int32_t n_devices, const bhg_camera *cam, const double *jitter, int32_t tile,
                     int32_t gather, bhg_frame **out);
void bhg_frame_destroy(bhg_frame *frame);
int bhg_frame_set_scene(bhg_frame *frame, const bhg_frame_scene *scene);
/* Move the camera of an existing frame -- what the engine reads anew on every render, origin and rotation of
 * depsgraph.scene.camera.matrix_world (RelativisticRenderEngine.py:182-183), field of view (:72-73) -- while the frame object,
 * its jitter stream (re-seeded identically every render, :189), tile dealing and device buffers stay: origin, rotation and field of view may change, width / height / samples may not.  A new origin costs
 * nothing (rays are directions; the origin goes into every trace call); a new rotation or field of view regenerates the
 * rays on the devices at the next render. */
int bhg_frame_set_camera(bhg_frame *frame, const bhg_camera *cam);
/* One frame.  rgba_host [height][width][4] float (pageable or page-locked): blocking, the image is there on return.
 * rgba_host = NULL: the render is only enqueued and the image stays on the first device (bhg_frame_device_image;
 * bhg_frame_synchronize waits) -- an animation loop that consumes frames on the GPU, and what bench.py times.
 * A sky-only scene is traced direction-only (bhg_trace_dir_device); with a disk or objects, whole end records. */
int bhg_frame_render(bhg_frame *frame, const bhg_params *p, float *rgba_host);
int bhg_frame_synchronize(bhg_frame *frame);
const float *bhg_frame_device_image(bhg_frame *frame); /* device address (first listed device) of the last image */
/* Re-deal the tiles by the measured cost of the last render (see above); the next render regenerates the rays.
 * root_share in (0, 1] (0 = 1): the part of an equal share the FIRST device is dealt -- it also receives the gather and
 * assembles the frame, and with a smaller shard all devices finish together (e.g. 1 - (N-1)/N * t_root / t_trace from
 * bhg_frame_last_ms). */
int bhg_frame_rebalance(bhg_frame *frame, double root_share);
/* out = {rays, attempted steps, accepted steps, horizon rays} of the last render, summed over the devices (waits). */
int bhg_frame_stats(bhg_frame *frame, uint64_t out[4]);
/* out = {n_devices, gather mode in use (BHG_FRAME_GATHER_COPY / _RCCL), largest shard in pixels, smallest shard,
 * tile, 1 if dealt by measured cost, renders so far, 1 if the last render traced directions only}. */
int bhg_frame_info(const bhg_frame *frame, int64_t out[8]);
/* Per-render timing.  While profiling is on (a flag: it may be switched from render to render, e.g. on for every 4th
 * frame of a timed loop) every render records a HIP event pair around its trace call on each device's stream;
 * bhg_frame_last_ms waits for them and gives the MEAN trace-call milliseconds per listed device over the profiled
 * renders since the last call (trace_ms [n_devices]; for the Schwarzschild forms the call is the one trace kernel) and
 * the root's gather-wait + assembly time of the last profiled render (root_ms, may be NULL; 0 for a one-device frame). */
int bhg_frame_set_profiling(bhg_frame *frame, int enable);
int bhg_frame_last_ms(bhg_frame *frame, float *trace_ms, float *root_ms);
/* The frame's tile dealing as a function of its own (host only, no device needed): the flat pixel ids y * width + x
 * of device `rank` of `world`, tile after tile, row-major inside a tile.  tile_cost NULL: cyclic dealing; else one
 * figure per tile (row-major over the tile grid): dealt by cost ranking (root_share in (0, 1]: rank 0's part of an equal
 * share, see bhg_frame_rebalance) and, visit_by_cost != 0, visited longest first.  pixels NULL: only the count is
 * returned in *n_out. */
int bhg_deal_tiles(int32_t width, int32_t height, int32_t tile, int32_t world, const double *tile_cost, int32_t visit_by_cost,
                   double root_share, int32_t rank, int64_t *pixels, size_t capacity, size_t *n_out);

/* Acceleration probe: acc[n][3] = -Gamma^i_{mu nu} k^mu k^nu at (x[n][3], k[n][3]); host buffers.
 * Lets tests compare the device RHS with the oracle's term by term.  With rhs_form = BHG_RHS_KERR_BL the triples
 * are Boyer-Lindquist: x = (r, theta, phi), k = d(r, theta, phi)/dlambda, acc = d^2(r, theta, phi)/dlambda^2, and the
 * Killing constants E = -k_t, L = k_phi the right-hand side needs are fixed by the null condition at each point
 * (the trace fixes them the same way at the camera); p->spin is the Kerr a. */
int bhg_acceleration(bhg_context *ctx, const bhg_params *p, const double *x, const double *k,
                     size_t n, double *acc);

/* Wait for everything enqueued on the context's own stream. */
int bhg_synchronize(bhg_context *ctx);

/* The context's own non-blocking stream (a hipStream_t), used by the host-buffer calls. */
void *bhg_context_stream(bhg_context *ctx);

/* Per-pass timing of the trace calls.  A trace call runs up to two passes on the caller's stream:
 * PREPARE (per-ray setup: f0, initial step, Kerr's Cartesian -> Boyer-Lindquist conversion -- done inside
 * TRACE by the waves' queue fill for every form since ABI 6, so this slot reads 0; it was a launch of its own
 * for BHG_RHS_KERR_BL before) and TRACE (the integrate loop including the root search for
 * rays that end on an event; the dominant kernel).  With profiling enabled the library records HIP
 * events around each pass on that stream; bhg_last_pass_ms() waits for the last call's events and
 * returns {prepare, trace, post} in milliseconds: post is the pass after the trace kernel -- the
 * Boyer-Lindquist -> Cartesian finalize of BHG_RHS_KERR_BL, 0 for the Schwarzschild forms (up to the
 * ABI 1.x builds of round 1 the slot was a separate root-search pass). */
int bhg_set_profiling(bhg_context *ctx, int enable);
int bhg_last_pass_ms(bhg_context *ctx, float out_ms[3]);

/* Kernel launch geometry chosen for the last bhg_trace* call (for DESIGN/bench reporting):
 * out[0] = workgroups, out[1] = threads per workgroup, out[2] = resident waves per CU,
 * out[3] = number of trace launches the call took: 1 -- rays whose disk / object candidate step held no terminal
 *          event are resumed inside the same launch -- unless the call holds more than 2^26 rays (a launch takes at
 *          most that many; BASELINE's largest frame, 2048 x 2048 x 16, is exactly one). */
int bhg_last_launch(bhg_context *ctx, int32_t out[4]);

/* Roofline calibration on THIS device, in the trace kernels' own launch geometry (one wave64 per workgroup, 12 resident
 * waves per CU; no memory traffic inside the loop) -- bench.py's `roofline.calibration` block, so that two bench lines
 * from two boxes of a pool can be compared (SURVEY.md section 8d prices against the vendor's 78.6 TFLOP/s):
 *   BHG_PROBE_FMA       nothing but v_fma_f64, eight independent chains per lane: the fp64 vector rate this box sustains;
 *   BHG_PROBE_STEP_MIX  the DP5(4) step loop's mix -- per 503 VALU instructions 8 v_rcp_f64 + 8 v_rsq_f64 (quarter rate)
 *                       among 487 v_fma_f64: the issue-bound ceiling of a kernel with that mix.
 * target_ms: duration of one probe launch (0 = 1 ms); the loop is sized from a short launch, then the MEDIAN of five
 * launches is reported.  out = {TFLOP/s (FMA = 2 flop, rcp / rsq = 1), ms per launch (median), VALU wave-instructions
 * per launch, quarter-rate ones among them, fastest of the five launches in ms, the shader clock in MHz that a full-rate
 * fp64 pipe (128 flop per clock and CU) needs for that TFLOP/s figure}.  Blocking; runs on the context's own stream. */
#define BHG_PROBE_FMA 0
#define BHG_PROBE_STEP_MIX 1
int bhg_peak_probe(bhg_context *ctx, int32_t kind, double target_ms, double out[6]);

#ifdef __cplusplus
}
#endif
#endif /* BHGEO_H */
