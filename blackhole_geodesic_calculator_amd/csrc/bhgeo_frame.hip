// bhgeo_frame.hip -- the library-owned FRAME of libbhgeo.so (bhg_frame_* in include/bhgeo.h): jitter stream -> camera
// rays -> geodesics -> shaded, sample-averaged float RGBA pixels in frame order, on one or several GPUs of ONE process,
// with no PyTorch (or Python) anywhere near it.
//
// What it replaces: the whole body of the reference's frame loop, raytracer/RelativisticRenderEngine.py:172-267
// (ray_trace: pinhole + jitter :224-230, the per-ray solve :232 -> :293-294, background_hit :366-378, the sample mean
// :242-250) as Blender calls it -- render() :50 -> render_scene() :152-168 on ONE render thread of ONE process; the
// author's commented-out mp.Pool (:210-216) marks where the parallelism has to live.  Here: one host thread drives one
// bhg_context per listed device; 32x32-pixel tiles are dealt to the devices (all samples of a pixel on one device), each
// device generates, traces and shades its shard into a slab, the slabs are gathered onto the first device -- RCCL in
// single-process mode (ncclCommInitAll, grouped ncclSend / ncclRecv over xGMI) when the devices are distinct, plain
// device-to-device copies when the list repeats a device or RCCL cannot be loaded -- put into frame order by one kernel
// and handed back as one array.
//
// Built on the public C ABI (bhg_create, bhg_trace*_device, bhg_shade*_device, bhg_assemble_frame_f32_device): the frame
// is a client of the same boundary every other caller uses; only the ray generation goes to the internal launcher, for
// its compact-jitter form (each device gets the draws of ITS pixels, not the whole frame's stream).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/bhgeo.h"
#include "geodesic_kernels.h"
#include "tile_dealing.h"

namespace bhg {
int set_error(int code, const std::string &msg);   // bhgeo_capi.hip: the thread-local message of bhg_last_error()
void host_copy(bhg_context *c, void *dst, const void *src, size_t bytes, size_t piece);   // bhgeo_capi.hip: multi-threaded memcpy in jobs of `piece` bytes
}

namespace {

int fail(int code, const std::string &msg) { return bhg::set_error(code, msg); }

int fail_hip(hipError_t e, const char *what)
{
    (void)hipGetLastError();     // (reported here: not again by the next launch's status)
    return bhg::set_error(e == hipErrorOutOfMemory ? BHG_E_NOMEM : BHG_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// No C++ exception leaves the library through the C ABI: the entry points that use standard containers (the jitter stream and
// the images are copied into vectors, the tile dealing sorts) are function-try-blocks that end here.
int host_exception()
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(BHG_E_NOMEM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(BHG_E_HIP, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(BHG_E_HIP, "internal error: unknown exception");
    }
}

#define HIP_TRY(expr)                                     \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)
#define BHG_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != BHG_OK) return _rc; \
    } while (0)

struct DeviceScope {   // the calling thread's current device is put back on the way out of every entry point
    int prev = -1;
    DeviceScope()
    {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---- RCCL, loaded at run time (libbhgeo.so itself links only the HIP runtime) ---------------------------------------
typedef void *nccl_comm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib != nullptr; }
};
constexpr int NCCL_FLOAT32 = 7;   // ncclFloat32 (rccl.h)

Rccl load_rccl()
{
    Rccl r;
    // by SONAME first: a host process that already carries RCCL (PyTorch-ROCm bundles one) keeps ONE copy
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return r;
    Rccl t;
    t.lib = h;
    t.CommInitAll = (decltype(t.CommInitAll))dlsym(h, "ncclCommInitAll");
    t.CommDestroy = (decltype(t.CommDestroy))dlsym(h, "ncclCommDestroy");
    t.GroupStart = (decltype(t.GroupStart))dlsym(h, "ncclGroupStart");
    t.GroupEnd = (decltype(t.GroupEnd))dlsym(h, "ncclGroupEnd");
    t.Send = (decltype(t.Send))dlsym(h, "ncclSend");
    t.Recv = (decltype(t.Recv))dlsym(h, "ncclRecv");
    t.GetErrorString = (decltype(t.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (t.CommInitAll && t.CommDestroy && t.GroupStart && t.GroupEnd && t.Send && t.Recv && t.GetErrorString) r = t;
    return r;
}

Rccl &rccl()
{
    static Rccl r = load_rccl();   // (a function-local static: initialised once, also when two threads create frames at once)
    return r;
}

int fail_nccl(int e, const char *what)
{
    return fail(BHG_E_HIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"));
}
#define NCCL_TRY(expr)                         \
    do {                                       \
        int _e = (expr);                       \
        if (_e != 0) return fail_nccl(_e, #expr); \
    } while (0)

struct DevBuf {   // a device allocation that remembers its device
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    int ensure(int dev, size_t need)
    {
        if (p && bytes >= need && dev == device) return BHG_OK;
        release();
        if (need == 0) return BHG_OK;
        HIP_TRY(hipSetDevice(dev));
        HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        device = dev;
        return BHG_OK;
    }
    void release()
    {
        if (p) {
            (void)hipSetDevice(device);
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

struct Shard {
    int device = 0;
    bhg_context *ctx = nullptr;
    hipStream_t stream = nullptr;   // the context's own
    std::vector<int64_t> pixels;    // flat ids y * W + x, tile after tile
    size_t P = 0, n = 0;
    DevBuf pixels_d, jitter_d, k0, end, dir, flags, steps, acc, obj, slab, sky, disk_tex;
    hipEvent_t done = nullptr;
    bool rays_ready = false;
    bool jitter_ready = false;      // jitter_d holds the draws of THIS pixel list (kept: a rotating camera regenerates the rays from it)
    bool scene_ready = false;
    bool dir_traced = false;
    nccl_comm_t comm = nullptr;
    // profiling: HIP event pairs around the trace call of every profiled render since the last bhg_frame_last_ms()
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    size_t ev_used = 0;
};

}  // namespace

struct bhg_frame {
    bhg_camera cam;
    int32_t tile = 32;
    int32_t gather = BHG_FRAME_GATHER_COPY;   // the mode in use (never AUTO)
    std::vector<Shard> sh;
    std::vector<double> jitter;     // host copy of the caller's stream (empty: pixel centres)
    // scene (host copies; uploaded to every device on the next render)
    std::vector<float> sky, disk_tex;
    int32_t sky_w = 0, sky_h = 0, disk_w = 0, disk_h = 0;
    bhg_frame_scene scene;          // (its pointers are not used after bhg_frame_set_scene)
    // root (device of shard 0)
    DevBuf recv, perm, image;       // [n_dev * pmax][4] float, [H W] int64, [H W][4] float
    size_t pmax = 0;
    void *pin = nullptr;            // page-locked staging of the image for pageable callers
    size_t pin_bytes = 0;
    std::vector<double> tile_cost;  // measured attempted steps per tile of the last render (bhg_frame_rebalance)
    bool dealt_by_cost = false;
    double root_share = 1.0;        // part of an equal share the first device is dealt (bhg_frame_rebalance)
    bool rendered = false;
    bool profiling = false;
    bool peer_copy_always = false;  // BHG_FRAME_GATHER_COPY_PEERCALL: the gather's copies go through hipMemcpyPeerAsync also between
                                    // contexts of ONE device (the N-device call on a one-GPU box)
    std::vector<hipEvent_t> ev_root;   // around the root's gather + assembly (profiling)
    std::vector<hipEvent_t> ev_piece;  // the pieces of the image's way back to a pageable caller array
    hipEvent_t assembled = nullptr;    // the root has read the receive block of the last render (copies of the next wait for it)
    uint64_t renders = 0;
};

namespace {

// ---- tile dealing: tile_dealing.h (plain C++, also built with the host compiler under sanitizers) ------------------
using bhg::deal_tiles_into;

void deal_tiles(bhg_frame *f)
{
    const int W = f->cam.width, H = f->cam.height, T = f->tile, world = (int)f->sh.size();
    const int nt = ((W + T - 1) / T) * ((H + T - 1) / T);
    const bool by_cost = f->dealt_by_cost && (int)f->tile_cost.size() == nt;
    std::vector<std::vector<int64_t>> px;
    deal_tiles_into(W, H, T, world, by_cost ? f->tile_cost.data() : nullptr, world > 1, f->root_share, px);
    f->pmax = 0;
    for (size_t r = 0; r < f->sh.size(); r++) {
        Shard &s = f->sh[r];
        s.pixels.swap(px[r]);
        s.P = s.pixels.size();
        s.n = s.P * (size_t)f->cam.samples;
        s.rays_ready = false;
        s.jitter_ready = false;
        f->pmax = std::max(f->pmax, s.P);
    }
    // the devices' steps / flags arrays still hold the PREVIOUS pixel lists' rays (and may be smaller than the new shards):
    // nothing may read them by the new lists until the next render
    f->rendered = false;
}

int upload_shard_geometry(bhg_frame *f, Shard &s)
{
    // pixel list, compact jitter [S][P][2] in list order, ray buffers
    const size_t S = (size_t)f->cam.samples, frame_px = (size_t)f->cam.width * (size_t)f->cam.height;
    HIP_TRY(hipSetDevice(s.device));
    BHG_TRY(s.pixels_d.ensure(s.device, s.P * sizeof(int64_t)));
    BHG_TRY(s.k0.ensure(s.device, s.n * 3 * sizeof(double)));
    BHG_TRY(s.flags.ensure(s.device, s.n));
    BHG_TRY(s.steps.ensure(s.device, s.n * sizeof(uint32_t)));
    BHG_TRY(s.acc.ensure(s.device, s.n * sizeof(uint32_t)));
    if (s.P == 0) {
        s.rays_ready = true;
        return BHG_OK;
    }
    HIP_TRY(hipMemcpyAsync(s.pixels_d.p, s.pixels.data(), s.P * sizeof(int64_t), hipMemcpyHostToDevice, s.stream));
    std::vector<double> jc;
    if (!f->jitter.empty() && !s.jitter_ready) {
        jc.resize(2 * S * s.P);
        for (size_t sm = 0; sm < S; sm++)
            for (size_t p = 0; p < s.P; p++) {
                const size_t src = (sm * frame_px + (size_t)s.pixels[p]) * 2, dst = (sm * s.P + p) * 2;
                jc[dst] = f->jitter[src];
                jc[dst + 1] = f->jitter[src + 1];
            }
        BHG_TRY(s.jitter_d.ensure(s.device, jc.size() * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(s.jitter_d.p, jc.data(), jc.size() * sizeof(double), hipMemcpyHostToDevice, s.stream));
    }
    bhg::RaygenArgs a;
    std::memset(&a, 0, sizeof(a));
    a.jitter = f->jitter.empty() ? nullptr : s.jitter_d.as<double>();
    a.compact = 1;
    a.pixels = s.pixels_d.as<int64_t>();
    a.k0 = s.k0.as<double>();
    a.n_pixels = s.P;
    a.width = f->cam.width;
    a.height = f->cam.height;
    a.samples = f->cam.samples;
    a.fov_x = f->cam.fov_x;
    a.fov_y = f->cam.fov_y;
    static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    a.rotate = std::memcmp(f->cam.rot, eye, sizeof(eye)) != 0;
    std::memcpy(a.rot, f->cam.rot, sizeof(a.rot));
    HIP_TRY(bhg::launch_raygen(a, s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));   // (jc and the pixel list are host temporaries of this call)
    // the shard's draws stay on the device (16 B per ray): a camera that rotates from frame to frame regenerates its rays
    // with one raygen launch, no host-side gather of the stream and no upload
    s.jitter_ready = !f->jitter.empty();
    s.rays_ready = true;
    return BHG_OK;
}

int upload_shard_scene(bhg_frame *f, Shard &s)
{
    HIP_TRY(hipSetDevice(s.device));
    BHG_TRY(s.sky.ensure(s.device, f->sky.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(s.sky.p, f->sky.data(), f->sky.size() * sizeof(float), hipMemcpyHostToDevice, s.stream));
    if (!f->disk_tex.empty()) {
        BHG_TRY(s.disk_tex.ensure(s.device, f->disk_tex.size() * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(s.disk_tex.p, f->disk_tex.data(), f->disk_tex.size() * sizeof(float), hipMemcpyHostToDevice, s.stream));
    } else {
        s.disk_tex.release();
    }
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.scene_ready = true;
    return BHG_OK;
}

int build_root(bhg_frame *f)
{
    // receive block, frame permutation, image -- on the device of shard 0
    const int root = f->sh[0].device;
    const size_t HW = (size_t)f->cam.width * (size_t)f->cam.height, world = f->sh.size();
    BHG_TRY(f->image.ensure(root, HW * 4 * sizeof(float)));
    if (!f->assembled) {
        HIP_TRY(hipSetDevice(root));
        HIP_TRY(hipEventCreateWithFlags(&f->assembled, hipEventDisableTiming));
    }
    // (a ONE-device frame in RCCL mode sends its slab to itself: the whole gather path on a single GPU, for tests)
    if ((world > 1 && f->gather != BHG_FRAME_GATHER_PEER) || f->gather == BHG_FRAME_GATHER_RCCL) {
        BHG_TRY(f->recv.ensure(root, world * f->pmax * 4 * sizeof(float)));
        BHG_TRY(f->perm.ensure(root, HW * sizeof(int64_t)));
        std::vector<int64_t> perm(HW);
        for (size_t r = 0; r < world; r++)
            for (size_t p = 0; p < f->sh[r].P; p++) perm[(size_t)f->sh[r].pixels[p]] = (int64_t)(r * f->pmax + p);
        HIP_TRY(hipSetDevice(root));
        HIP_TRY(hipMemcpy(f->perm.p, perm.data(), HW * sizeof(int64_t), hipMemcpyHostToDevice));
        for (size_t r = (world > 1 ? 1 : 0); r < world; r++) BHG_TRY(f->sh[r].slab.ensure(f->sh[r].device, f->pmax * 4 * sizeof(float)));
    }
    return BHG_OK;
}

bool is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

void fill_scene(const bhg_frame *f, const Shard &s, bhg_scene *sc)
{
    std::memset(sc, 0, sizeof(*sc));
    sc->d_sky = s.sky.as<float>();
    sc->sky_w = f->sky_w;
    sc->sky_h = f->sky_h;
    sc->d_disk_tex = f->disk_tex.empty() ? nullptr : s.disk_tex.as<float>();
    sc->disk_w = f->disk_w;
    sc->disk_h = f->disk_h;
    sc->disk_r_in = f->scene.disk_r_in;
    sc->disk_r_out = f->scene.disk_r_out;
    sc->disk_phase = f->scene.disk_phase;
    sc->disk_mean = f->scene.disk_mean;
    sc->disk_stddev = f->scene.disk_stddev;
    sc->disk_intensity = f->scene.disk_intensity;
    sc->n_spheres = f->scene.n_spheres;
    sc->n_lamps = f->scene.n_lamps;
    std::memcpy(sc->spheres, f->scene.spheres, sizeof(sc->spheres));
    std::memcpy(sc->sphere_rgb, f->scene.sphere_rgb, sizeof(sc->sphere_rgb));
    std::memcpy(sc->lamps, f->scene.lamps, sizeof(sc->lamps));
}

void destroy_frame(bhg_frame *f)
{
    if (!f) return;
    for (auto &s : f->sh) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    for (auto &s : f->sh)
        if (s.comm && rccl().ok()) (void)rccl().CommDestroy(s.comm);
    for (auto &s : f->sh) {
        (void)hipSetDevice(s.device);
        for (DevBuf *b : {&s.pixels_d, &s.jitter_d, &s.k0, &s.end, &s.dir, &s.flags, &s.steps, &s.acc, &s.obj, &s.slab, &s.sky, &s.disk_tex})
            b->release();
        if (s.done) (void)hipEventDestroy(s.done);
        for (auto &e : s.evs) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (s.ctx) bhg_destroy(s.ctx);
    }
    f->recv.release();
    f->perm.release();
    f->image.release();
    if (!f->sh.empty()) (void)hipSetDevice(f->sh[0].device);
    for (auto e : f->ev_root)
        if (e) (void)hipEventDestroy(e);
    if (f->assembled) (void)hipEventDestroy(f->assembled);
    for (auto e : f->ev_piece)
        if (e) (void)hipEventDestroy(e);
    if (f->pin) (void)hipHostFree(f->pin);
    delete f;
}

}  // namespace

extern "C" {

int bhg_deal_tiles(int32_t width, int32_t height, int32_t tile, int32_t world, const double *tile_cost, int32_t visit_by_cost,
                   double root_share, int32_t rank, int64_t *pixels, size_t capacity, size_t *n_out)
try {
    if (width <= 0 || height <= 0 || tile <= 0 || world <= 0 || rank < 0 || rank >= world || !n_out || !(root_share > 0.0 && root_share <= 1.0))
        return fail(BHG_E_INVALID, "bad argument");
    if (!bhg::tile_grid_fits(width, height, tile)) return fail(BHG_E_INVALID, "more than 2^31 - 1 tiles");
    std::vector<std::vector<int64_t>> px;
    deal_tiles_into(width, height, tile, world, tile_cost, visit_by_cost != 0, root_share, px);
    const auto &mine = px[(size_t)rank];
    *n_out = mine.size();
    if (pixels) {
        if (capacity < mine.size()) return fail(BHG_E_INVALID, "pixel buffer too small");
        std::memcpy(pixels, mine.data(), mine.size() * sizeof(int64_t));
    }
    return BHG_OK;
} catch (...) {
    return host_exception();
}

int bhg_frame_create(const int32_t *devices, int32_t n_devices, const bhg_camera *cam, const double *jitter, int32_t tile,
                     int32_t gather, bhg_frame **out)
try {
    if (!out) return fail(BHG_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(BHG_E_INVALID, "devices: a list of 1 .. 64 device indices");
    if (!cam) return fail(BHG_E_INVALID, "camera is NULL");
    if (cam->width <= 0 || cam->height <= 0 || cam->samples <= 0) return fail(BHG_E_INVALID, "width, height, samples must be > 0");
    if (tile <= 0) tile = 32;
    if (gather != BHG_FRAME_GATHER_AUTO && gather != BHG_FRAME_GATHER_COPY && gather != BHG_FRAME_GATHER_RCCL &&
        gather != BHG_FRAME_GATHER_PEER && gather != BHG_FRAME_GATHER_COPY_PEERCALL)
        return fail(BHG_E_INVALID, "unknown gather mode");
    const size_t HW = (size_t)cam->width * (size_t)cam->height;
    if (HW * (size_t)cam->samples > 0xFFFFFFFFull) return fail(BHG_E_INVALID, "more than 2^32 rays in the frame");
    if (!bhg::tile_grid_fits(cam->width, cam->height, tile)) return fail(BHG_E_INVALID, "more than 2^31 - 1 tiles");
    const int n_vis = bhg_device_count();
    if (n_vis <= 0) return fail(BHG_E_NO_DEVICE, "no HIP device visible (libbhgeo has no CPU fallback)");
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= n_vis) return fail(BHG_E_NO_DEVICE, "device index out of range");
        for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    }
    if (gather == BHG_FRAME_GATHER_RCCL && !distinct)
        return fail(BHG_E_INVALID, "RCCL needs distinct devices (a repeated device gathers by device-to-device copies)");
    if (gather == BHG_FRAME_GATHER_RCCL && !rccl().ok()) return fail(BHG_E_HIP, "librccl.so could not be loaded");
    DeviceScope scope;
    struct Guard {        // (an exception on the way -- the jitter copy, the dealing -- must not leave the half-built frame behind)
        bhg_frame *f = nullptr;
        ~Guard() { if (f) destroy_frame(f); }
    } guard;
    bhg_frame *f = guard.f = new (std::nothrow) bhg_frame();
    if (!f) return fail(BHG_E_NOMEM, "host allocation failed");
    f->cam = *cam;
    f->tile = tile;
    if (gather == BHG_FRAME_GATHER_COPY_PEERCALL) {      // (an explicit mode of the call, not an environment variable: ADVICE r05)
        f->peer_copy_always = true;
        gather = BHG_FRAME_GATHER_COPY;
    }
    std::memset(&f->scene, 0, sizeof(f->scene));
    f->scene.disk_mean = 0.2;   // the Limited engine's defaults (LimitedRelativisticRenderEngine.py:495-498)
    f->scene.disk_stddev = 0.3;
    f->scene.disk_intensity = 1.0;
    if (jitter) f->jitter.assign(jitter, jitter + 2 * HW * (size_t)cam->samples);
    f->sh.resize((size_t)n_devices);
    int rc = BHG_OK;
    for (int i = 0; i < n_devices && rc == BHG_OK; i++) {
        Shard &s = f->sh[i];
        s.device = devices[i];
        rc = bhg_create(s.device, &s.ctx);
        if (rc != BHG_OK) break;
        s.stream = (hipStream_t)bhg_context_stream(s.ctx);
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail_hip(e, "hipEventCreate");
    }
    // gather mode: RCCL when asked for, or (AUTO) when there is something to gather between distinct devices
    f->gather = BHG_FRAME_GATHER_COPY;
    if (rc == BHG_OK && n_devices > 1 && distinct && gather != BHG_FRAME_GATHER_COPY && gather != BHG_FRAME_GATHER_PEER && rccl().ok())
        f->gather = BHG_FRAME_GATHER_RCCL;
    if (rc == BHG_OK && n_devices == 1 && gather == BHG_FRAME_GATHER_RCCL) f->gather = BHG_FRAME_GATHER_RCCL;   // (a 1-rank group: tests)
    if (rc == BHG_OK && f->gather == BHG_FRAME_GATHER_RCCL) {
        std::vector<nccl_comm_t> comms((size_t)n_devices, nullptr);
        std::vector<int> devs(devices, devices + n_devices);
        const int e = rccl().CommInitAll(comms.data(), n_devices, devs.data());
        if (e != 0) {
            if (gather == BHG_FRAME_GATHER_RCCL) rc = fail_nccl(e, "ncclCommInitAll");
            else f->gather = BHG_FRAME_GATHER_COPY;   // AUTO: fall back to copies
        } else {
            for (int i = 0; i < n_devices; i++) f->sh[i].comm = comms[i];
        }
    }
    if (rc == BHG_OK && gather == BHG_FRAME_GATHER_PEER) f->gather = BHG_FRAME_GATHER_PEER;
    if (rc == BHG_OK && (f->gather == BHG_FRAME_GATHER_COPY || f->gather == BHG_FRAME_GATHER_PEER) && n_devices > 1) {
        // peer access to the first device: for the copies best effort (without it the runtime stages through the host),
        // for the peer stores a must
        for (int i = 1; i < n_devices && rc == BHG_OK; i++) {
            if (devices[i] == devices[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
                (void)hipSetDevice(devices[i]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e != hipSuccess) (void)hipGetLastError();   // (already enabled is fine)
            } else if (f->gather == BHG_FRAME_GATHER_PEER) {
                rc = fail(BHG_E_HIP, "BHG_FRAME_GATHER_PEER: a listed device cannot access the first device's memory");
            }
        }
    }
    if (rc == BHG_OK) {
        deal_tiles(f);
        rc = build_root(f);
    }
    if (rc != BHG_OK) {
        const std::string msg = bhg_last_error();
        guard.f = nullptr;
        destroy_frame(f);
        return fail(rc, msg);
    }
    guard.f = nullptr;
    *out = f;
    return BHG_OK;
} catch (...) {
    return host_exception();
}

void bhg_frame_destroy(bhg_frame *f)
{
    DeviceScope scope;
    destroy_frame(f);
}

int bhg_frame_set_camera(bhg_frame *f, const bhg_camera *cam)
{
    if (!f || !cam) return fail(BHG_E_INVALID, "frame / camera is NULL");
    if (cam->width != f->cam.width || cam->height != f->cam.height || cam->samples != f->cam.samples)
        return fail(BHG_E_INVALID, "width, height and samples are fixed at bhg_frame_create (the jitter stream and the tile dealing belong to them)");
    // the rays are directions only (the origin is handed to every trace call): they are regenerated when what shapes them changes
    const bool dirs_change = cam->fov_x != f->cam.fov_x || cam->fov_y != f->cam.fov_y || std::memcmp(cam->rot, f->cam.rot, sizeof(cam->rot)) != 0;
    f->cam = *cam;
    if (dirs_change)
        for (auto &s : f->sh) s.rays_ready = false;
    return BHG_OK;
}

int bhg_frame_set_scene(bhg_frame *f, const bhg_frame_scene *sc)
try {
    if (!f || !sc) return fail(BHG_E_INVALID, "frame / scene is NULL");
    if (sc->n_spheres < 0 || sc->n_spheres > BHG_MAX_SPHERES || sc->n_lamps < 0 || sc->n_lamps > 4)
        return fail(BHG_E_INVALID, "n_spheres must be in [0, BHG_MAX_SPHERES], n_lamps in [0, 4]");
    if (sc->sky && (sc->sky_w <= 0 || sc->sky_h <= 0)) return fail(BHG_E_INVALID, "sky size must be > 0");
    if (!sc->sky && f->sky.empty()) return fail(BHG_E_INVALID, "the first scene must bring a sky image");
    if (sc->disk_r_out > 0.0 && (!(sc->disk_r_out > sc->disk_r_in) || !(sc->disk_r_in >= 0.0) || !(sc->disk_stddev > 0.0)))
        return fail(BHG_E_INVALID, "disk needs 0 <= r_in < r_out and stddev > 0");
    if (sc->disk_tex && (sc->disk_w <= 0 || sc->disk_h <= 0)) return fail(BHG_E_INVALID, "disk texture size must be > 0");
    for (int j = 0; j < sc->n_spheres; j++)
        if (!(sc->spheres[j][3] > 0.0)) return fail(BHG_E_INVALID, "sphere radii must be > 0");
    bool images_changed = false;
    if (sc->sky) {
        f->sky.assign(sc->sky, sc->sky + (size_t)sc->sky_w * (size_t)sc->sky_h * 4);
        f->sky_w = sc->sky_w;
        f->sky_h = sc->sky_h;
        images_changed = true;
    }
    if (sc->disk_tex) {
        f->disk_tex.assign(sc->disk_tex, sc->disk_tex + (size_t)sc->disk_w * (size_t)sc->disk_h * 4);
        f->disk_w = sc->disk_w;
        f->disk_h = sc->disk_h;
        images_changed = true;
    } else if (!(sc->disk_r_out > 0.0) && !f->disk_tex.empty()) {
        f->disk_tex.clear();
        f->disk_w = f->disk_h = 0;
        images_changed = true;
    }
    f->scene = *sc;
    f->scene.sky = f->scene.disk_tex = nullptr;   // (no pointer of the caller's is kept)
    if (images_changed)
        for (auto &s : f->sh) s.scene_ready = false;
    return BHG_OK;
} catch (...) {
    return host_exception();
}

int bhg_frame_render(bhg_frame *f, const bhg_params *p, float *rgba_host)
try {
    if (!f || !p) return fail(BHG_E_INVALID, "frame / params is NULL");
    if (f->sky.empty()) return fail(BHG_E_INVALID, "bhg_frame_set_scene() first (the frame has no sky)");
    if ((p->disk_r_out > 0.0) != (f->scene.disk_r_out > 0.0) ||
        (p->disk_r_out > 0.0 && (p->disk_r_in != f->scene.disk_r_in || p->disk_r_out != f->scene.disk_r_out)))
        return fail(BHG_E_INVALID, "the disk of the trace parameters and of the scene differ");
    DeviceScope scope;
    const size_t world = f->sh.size(), HW = (size_t)f->cam.width * (size_t)f->cam.height;
    const int S = f->cam.samples;
    const bool has_obj = f->scene.n_spheres > 0, has_disk = f->scene.disk_r_out > 0.0;
    const bool dir_only = !has_obj && !has_disk;
    const bool loopback = world == 1 && f->gather == BHG_FRAME_GATHER_RCCL;   // (see build_root)
    const bool peer = f->gather == BHG_FRAME_GATHER_PEER;     // every device's shade stores straight into the image
    const bool gathered = (world > 1 && !peer) || loopback;
    Shard &root = f->sh[0];
    bhg_params prm = *p;
    if (prm.order_blocks == 0 && S > 1) prm.order_blocks = (uint32_t)S;   // the rays are S blocks of P (sample-major)

    // -- every device: (rays, scene images once) trace, shade + sample mean into its slab ---------------------------
    for (size_t r = 0; r < world; r++) {
        Shard &s = f->sh[r];
        if (!s.rays_ready) BHG_TRY(upload_shard_geometry(f, s));
        if (!s.scene_ready) BHG_TRY(upload_shard_scene(f, s));
    }
    for (size_t r = 0; r < world; r++) {
        Shard &s = f->sh[r];
        if (s.P == 0) continue;
        HIP_TRY(hipSetDevice(s.device));
        // where this shard's pixels go: one device -- straight into the image, in frame order; several -- shard 0 into
        // its part of the receive block, the others into their own slab
        float *dst = !gathered ? f->image.as<float>() : ((r == 0 && !loopback) ? f->recv.as<float>() : s.slab.as<float>());
        const int64_t *scatter = !gathered ? s.pixels_d.as<int64_t>() : nullptr;
        if (f->profiling) {
            if (s.ev_used == s.evs.size()) {
                hipEvent_t e0, e1;
                HIP_TRY(hipEventCreate(&e0));
                HIP_TRY(hipEventCreate(&e1));
                s.evs.emplace_back(e0, e1);
            }
            HIP_TRY(hipEventRecord(s.evs[s.ev_used].first, s.stream));
        }
        if (dir_only) {
            BHG_TRY(s.dir.ensure(s.device, s.n * 3 * sizeof(double)));
            BHG_TRY(bhg_trace_dir_device(s.ctx, &prm, f->cam.origin, nullptr, s.k0.as<double>(), s.n, s.dir.as<double>(),
                                         s.flags.as<uint8_t>(), s.steps.as<uint32_t>(), s.acc.as<uint32_t>(), s.stream));
            if (f->profiling) HIP_TRY(hipEventRecord(s.evs[s.ev_used++].second, s.stream));
            BHG_TRY(bhg_shade_dir_device(s.ctx, s.dir.as<double>(), s.flags.as<uint8_t>(), s.P, S, s.sky.as<float>(), f->sky_w,
                                         f->sky_h, nullptr, dst, scatter, s.stream));
        } else {
            BHG_TRY(s.end.ensure(s.device, s.n * 6 * sizeof(double)));
            if (has_obj) BHG_TRY(s.obj.ensure(s.device, s.n));
            BHG_TRY(bhg_trace_objects_device(s.ctx, &prm, has_obj ? &f->scene.spheres[0][0] : nullptr, f->scene.n_spheres,
                                             f->cam.origin, nullptr, s.k0.as<double>(), s.n, s.end.as<double>(),
                                             s.flags.as<uint8_t>(), s.steps.as<uint32_t>(), s.acc.as<uint32_t>(),
                                             has_obj ? s.obj.as<int8_t>() : nullptr, s.stream));
            if (f->profiling) HIP_TRY(hipEventRecord(s.evs[s.ev_used++].second, s.stream));
            bhg_scene sc;
            fill_scene(f, s, &sc);
            BHG_TRY(bhg_shade_scene_f32_device(s.ctx, s.end.as<double>(), s.flags.as<uint8_t>(), has_obj ? s.obj.as<int8_t>() : nullptr,
                                               s.P, S, &sc, dst, scatter, s.stream));
        }
        s.dir_traced = dir_only;
    }
    // -- ONE gather onto the first device -----------------------------------------------------------------------------
    if (f->profiling) {
        if (f->ev_root.empty()) {
            HIP_TRY(hipSetDevice(root.device));
            f->ev_root.resize(2, nullptr);
            for (auto &e : f->ev_root) HIP_TRY(hipEventCreate(&e));
        }
    }
    if (peer && world > 1) {
        // no exchange at all: the image is complete when every device's shade kernel has retired
        for (size_t r = 1; r < world; r++) {
            if (f->sh[r].P == 0) continue;
            HIP_TRY(hipSetDevice(f->sh[r].device));
            HIP_TRY(hipEventRecord(f->sh[r].done, f->sh[r].stream));
        }
        HIP_TRY(hipSetDevice(root.device));
        if (f->profiling) HIP_TRY(hipEventRecord(f->ev_root[0], root.stream));
        for (size_t r = 1; r < world; r++)
            if (f->sh[r].P) HIP_TRY(hipStreamWaitEvent(root.stream, f->sh[r].done, 0));
        if (f->profiling) HIP_TRY(hipEventRecord(f->ev_root[1], root.stream));
    }
    if (gathered) {
        const size_t slab_floats = f->pmax * 4;
        if (f->gather == BHG_FRAME_GATHER_RCCL) {
            // grouped point-to-point: every device sends its slab, the root receives them side by side
            if (f->profiling) {
                HIP_TRY(hipSetDevice(root.device));
                HIP_TRY(hipEventRecord(f->ev_root[0], root.stream));
            }
            NCCL_TRY(rccl().GroupStart());
            int ne = 0;
            const char *what = "";
            for (size_t r = loopback ? 0 : 1; r < world && ne == 0; r++) {
                ne = rccl().Send(f->sh[r].slab.p, slab_floats, NCCL_FLOAT32, 0, f->sh[r].comm, f->sh[r].stream);
                what = "ncclSend";
                if (ne != 0) break;
                ne = rccl().Recv(f->recv.as<float>() + r * slab_floats, slab_floats, NCCL_FLOAT32, (int)r, root.comm, root.stream);
                what = "ncclRecv";
            }
            // (the group is closed on the error path too: a group left open would swallow the next frame's calls)
            const int ge = rccl().GroupEnd();
            if (ne != 0) return fail_nccl(ne, what);
            NCCL_TRY(ge);
        } else {
            for (size_t r = 1; r < world; r++) {
                Shard &s = f->sh[r];
                HIP_TRY(hipSetDevice(s.device));
                // (the root must have read the previous render's slabs out of the receive block)
                if (f->renders > 0) HIP_TRY(hipStreamWaitEvent(s.stream, f->assembled, 0));
                float *dst = f->recv.as<float>() + r * slab_floats;
                if (s.device == root.device && !f->peer_copy_always)
                    HIP_TRY(hipMemcpyAsync(dst, s.slab.p, slab_floats * sizeof(float), hipMemcpyDeviceToDevice, s.stream));
                else
                    HIP_TRY(hipMemcpyPeerAsync(dst, root.device, s.slab.p, s.device, slab_floats * sizeof(float), s.stream));
                HIP_TRY(hipEventRecord(s.done, s.stream));
            }
            HIP_TRY(hipSetDevice(root.device));
            if (f->profiling) HIP_TRY(hipEventRecord(f->ev_root[0], root.stream));
            for (size_t r = 1; r < world; r++) HIP_TRY(hipStreamWaitEvent(root.stream, f->sh[r].done, 0));
        }
        HIP_TRY(hipSetDevice(root.device));
        BHG_TRY(bhg_assemble_frame_f32_device(root.ctx, f->recv.as<float>(), f->perm.as<int64_t>(), HW, f->image.as<float>(), root.stream));
        HIP_TRY(hipEventRecord(f->assembled, root.stream));
        if (f->profiling) HIP_TRY(hipEventRecord(f->ev_root[1], root.stream));
    }
    // -- one array back -----------------------------------------------------------------------------------------------
    HIP_TRY(hipSetDevice(root.device));
    if (rgba_host) {
        const size_t bytes = HW * 4 * sizeof(float);
        if (is_pinned(rgba_host)) {
            HIP_TRY(hipMemcpyAsync(rgba_host, f->image.p, bytes, hipMemcpyDeviceToHost, root.stream));
            HIP_TRY(hipStreamSynchronize(root.stream));
        } else {
            if (f->pin_bytes < bytes) {
                if (f->pin) HIP_TRY(hipHostFree(f->pin));
                f->pin = nullptr;
                f->pin_bytes = 0;
                HIP_TRY(hipHostMalloc(&f->pin, bytes, hipHostMallocDefault));
                f->pin_bytes = bytes;
            }
            // in two halves: while the second half crosses PCIe the first goes from the staging block into the caller's array
            // (the context's copy threads, 512-kB jobs so that all of them have work on half a frame).  Measured on one box
            // (scripts/dev/dev_library_frame.py, 1024 x 1024 frame): 1.80 ms in one piece, 1.72 in two, 1.70-1.75 in four;
            // four pieces with the pool's usual 2-MB jobs: 2.15 (two jobs per piece leave six threads idle)
            constexpr int PIECES = 2;
            if (f->ev_piece.empty()) {
                // (all or nothing: a creation that fails half way must not leave null handles behind for the next render)
                hipEvent_t made[PIECES] = {};
                for (int q = 0; q < PIECES; q++) {
                    const hipError_t e = hipEventCreateWithFlags(&made[q], hipEventDisableTiming);
                    if (e != hipSuccess) {
                        for (int r = 0; r < q; r++) (void)hipEventDestroy(made[r]);
                        return fail_hip(e, "hipEventCreateWithFlags (image pieces)");
                    }
                }
                f->ev_piece.assign(made, made + PIECES);
            }
            const size_t piece = ((bytes / PIECES) + 4095) & ~size_t(4095);
            int n_pieces = 0;
            for (size_t off = 0; off < bytes; off += piece, n_pieces++) {
                const size_t len = std::min(piece, bytes - off);
                HIP_TRY(hipMemcpyAsync((char *)f->pin + off, (const char *)f->image.p + off, len, hipMemcpyDeviceToHost, root.stream));
                HIP_TRY(hipEventRecord(f->ev_piece[n_pieces], root.stream));
            }
            int k = 0;
            for (size_t off = 0; off < bytes; off += piece, k++) {
                HIP_TRY(hipEventSynchronize(f->ev_piece[k]));
                bhg::host_copy(root.ctx, (char *)rgba_host + off, (const char *)f->pin + off, std::min(piece, bytes - off), size_t(512) << 10);
            }
        }
        // (the other devices' streams have been waited for through the gather; a frame of ONE device has one stream)
    }
    f->rendered = true;
    f->renders++;
    return BHG_OK;
} catch (...) {
    return host_exception();
}

int bhg_frame_synchronize(bhg_frame *f)
{
    if (!f) return fail(BHG_E_INVALID, "frame is NULL");
    DeviceScope scope;
    for (auto &s : f->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return BHG_OK;
}

const float *bhg_frame_device_image(bhg_frame *f) { return f ? f->image.as<float>() : nullptr; }

int bhg_frame_stats(bhg_frame *f, uint64_t out[4])
try {
    if (!f || !out) return fail(BHG_E_INVALID, "bad argument");
    if (!f->rendered) return fail(BHG_E_INVALID, "no render yet");
    DeviceScope scope;
    const int W = f->cam.width, T = f->tile, tx = (W + T - 1) / T, ty = (f->cam.height + T - 1) / T;
    f->tile_cost.assign((size_t)tx * ty, 0.0);
    uint64_t rays = 0, att = 0, acc = 0, hor = 0;
    std::vector<uint32_t> hs, ha;
    std::vector<uint8_t> hf;
    for (auto &s : f->sh) {
        if (s.n == 0) continue;
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.stream));
        hs.resize(s.n);
        ha.resize(s.n);
        hf.resize(s.n);
        HIP_TRY(hipMemcpy(hs.data(), s.steps.p, s.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ha.data(), s.acc.p, s.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hf.data(), s.flags.p, s.n, hipMemcpyDeviceToHost));
        rays += s.n;
        for (size_t i = 0; i < s.n; i++) {
            att += hs[i];
            acc += ha[i];
            hor += (hf[i] & BHG_FLAG_HIT_HORIZON) ? 1 : 0;
            const int64_t pix = s.pixels[i % s.P];
            f->tile_cost[(size_t)((pix / W) / T) * tx + (size_t)((pix % W) / T)] += (double)hs[i];
        }
    }
    out[0] = rays;
    out[1] = att;
    out[2] = acc;
    out[3] = hor;
    return BHG_OK;
} catch (...) {
    return host_exception();
}

int bhg_frame_rebalance(bhg_frame *f, double root_share)
try {
    if (!f) return fail(BHG_E_INVALID, "frame is NULL");
    if (root_share == 0.0) root_share = 1.0;
    if (!(root_share > 0.0 && root_share <= 1.0)) return fail(BHG_E_INVALID, "root_share must be in (0, 1] (0 = 1)");
    uint64_t st[4];
    BHG_TRY(bhg_frame_stats(f, st));   // (fills tile_cost from the last render's attempted steps)
    DeviceScope scope;
    f->dealt_by_cost = true;
    f->root_share = root_share;
    deal_tiles(f);
    return build_root(f);
} catch (...) {
    return host_exception();
}

int bhg_frame_info(const bhg_frame *f, int64_t out[8])
{
    if (!f || !out) return fail(BHG_E_INVALID, "bad argument");
    size_t pmin = (size_t)-1;
    for (auto &s : f->sh) pmin = std::min(pmin, s.P);
    out[0] = (int64_t)f->sh.size();
    out[1] = f->gather;
    out[2] = (int64_t)f->pmax;
    out[3] = (int64_t)pmin;
    out[4] = f->tile;
    out[5] = f->dealt_by_cost ? 1 : 0;
    out[6] = (int64_t)f->renders;
    out[7] = (!f->sh.empty() && f->sh[0].dir_traced) ? 1 : 0;
    return BHG_OK;
}

int bhg_frame_set_profiling(bhg_frame *f, int enable)
{
    if (!f) return fail(BHG_E_INVALID, "frame is NULL");
    f->profiling = enable != 0;   // (a flag: may be switched per render, e.g. on for every 4th frame of a timed loop)
    return BHG_OK;
}

int bhg_frame_last_ms(bhg_frame *f, float *trace_ms, float *root_ms)
try {
    if (!f || !trace_ms) return fail(BHG_E_INVALID, "bad argument");
    DeviceScope scope;
    bool any = false;
    for (size_t r = 0; r < f->sh.size(); r++) {
        Shard &s = f->sh[r];
        trace_ms[r] = 0.0f;
        if (s.ev_used == 0) continue;
        any = true;
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipEventSynchronize(s.evs[s.ev_used - 1].second));
        double sum = 0.0;
        for (size_t i = 0; i < s.ev_used; i++) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, s.evs[i].first, s.evs[i].second));
            sum += ms;
        }
        trace_ms[r] = (float)(sum / (double)s.ev_used);
        s.ev_used = 0;
    }
    if (!any) return fail(BHG_E_INVALID, "no profiled render since the last call (bhg_frame_set_profiling)");
    if (root_ms) {
        *root_ms = 0.0f;
        if ((f->sh.size() > 1 || f->gather == BHG_FRAME_GATHER_RCCL) && f->ev_root.size() == 2) {
            HIP_TRY(hipSetDevice(f->sh[0].device));
            HIP_TRY(hipEventSynchronize(f->ev_root[1]));
            HIP_TRY(hipEventElapsedTime(root_ms, f->ev_root[0], f->ev_root[1]));
        }
    }
    return BHG_OK;
} catch (...) {
    return host_exception();
}

}  // extern "C"
