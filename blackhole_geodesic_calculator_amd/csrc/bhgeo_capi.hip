// bhgeo_capi.hip -- the C ABI of libbhgeo.so (declared in include/bhgeo.h).
//
// Each entry point replaces a piece of the reference's per-ray Python hand-off
// (raytracer/RelativisticRenderEngine.py:134, :293-313; batched contract of
// raytracer/RelativisticRenderEngineCamEdition.py:225-228).  There is no CPU fallback: every
// compute entry point needs a HIP device and fails with BHG_E_NO_DEVICE / BHG_E_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bhgeo.h"
#include "geodesic_kernels.h"

static_assert(BHG_FLAG_HIT_HORIZON == bhg::BHG_FLAG_HIT_HORIZON_, "flag mismatch");
static_assert(BHG_FLAG_START_INSIDE == bhg::BHG_FLAG_START_INSIDE_, "flag mismatch");
static_assert(BHG_FLAG_REACHED_END == bhg::BHG_FLAG_REACHED_END_, "flag mismatch");
static_assert(BHG_FLAG_EXITED_SPHERE == bhg::BHG_FLAG_EXITED_SPHERE_, "flag mismatch");
static_assert(BHG_FLAG_MAX_STEPS == bhg::BHG_FLAG_MAX_STEPS_, "flag mismatch");
static_assert(BHG_FLAG_STEP_TOO_SMALL == bhg::BHG_FLAG_STEP_TOO_SMALL_, "flag mismatch");
static_assert(BHG_FLAG_NAN == bhg::BHG_FLAG_NAN_, "flag mismatch");
static_assert(BHG_FLAG_HIT_OBJECT == bhg::BHG_FLAG_HIT_OBJECT_ && BHG_MAX_SPHERES == bhg::BHG_MAX_SPHERES_, "object constants mismatch");
static_assert(BHG_METHOD_DP54 == bhg::BHG_METHOD_DP54_ && BHG_METHOD_RK4 == bhg::BHG_METHOD_RK4_, "method mismatch");
static_assert(BHG_RHS_CHRISTOFFEL == bhg::BHG_RHS_CHRISTOFFEL_ && BHG_RHS_REDUCED == bhg::BHG_RHS_REDUCED_ &&
                  BHG_RHS_KERR_BL == bhg::BHG_RHS_KERR_BL_,
              "rhs mismatch");
static_assert(sizeof(bhg_params) == 104, "bhg_params layout is part of the ABI");
static_assert(sizeof(bhg_camera) == 128, "bhg_camera layout is part of the ABI");
static_assert(sizeof(bhg_scene) == 664 && sizeof(bhg_frame_scene) == 664, "scene layouts are part of the ABI");
static_assert(BHG_FLAG_HIT_DISK == bhg::BHG_FLAG_HIT_DISK_, "flag mismatch");

namespace {

thread_local std::string g_err = "";

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();     // (reported here: not again by the next launch's status)
    return (e == hipErrorOutOfMemory) ? BHG_E_NOMEM : BHG_E_HIP;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)

}  // namespace

namespace bhg {
// for the other translation units of the library (bhgeo_frame.hip): set the thread-local message of bhg_last_error()
int set_error(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace bhg

namespace {

// Worker threads for the host side of the host-buffer entry points: a 100-MB-class memcpy between a caller's
// pageable array and the pinned staging ring runs at one core's ~10 GB/s single-threaded, several times below what
// the PCIe link moves; split over a few threads it keeps up.  Created on first use, joined with the context.
class HostCopyPool {
public:
    ~HostCopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // dst <- src, blocking; the calling thread takes pieces too
    void copy(void *dst, const void *src, size_t bytes, size_t piece = size_t(2) << 20)
    {
        if (bytes <= 2 * piece) {
            std::memcpy(dst, src, bytes);
            return;
        }
        start();
        size_t n_jobs = 0;
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t off = 0; off < bytes; off += piece, n_jobs++)
                q_.push_back({(char *)dst + off, (const char *)src + off, std::min(piece, bytes - off)});
            pending_ += n_jobs;
        }
        cv_.notify_all();
        for (;;) {  // help until the queue is empty, then wait for the pieces still being copied
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                if (q_.empty()) {
                    done_.wait(g, [&] { return pending_ == 0; });
                    return;
                }
                j = q_.front();
                q_.pop_front();
            }
            std::memcpy(j.dst, j.src, j.bytes);
            finish_one();
        }
    }

private:
    struct Job {
        char *dst;
        const char *src;
        size_t bytes;
    };
    void start()
    {
        if (started_) return;
        started_ = true;
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = hw > 1 ? std::min(hw - 1, 7u) : 0u;  // + the calling thread
        // (a process at its thread limit: fewer workers, or none -- the calling thread copies what nobody else takes)
        for (unsigned i = 0; i < n; i++) {
            try {
                th_.emplace_back([this] { run(); });
            } catch (const std::exception &) {
                break;
            }
        }
    }
    void finish_one()
    {
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_all();
    }
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                j = q_.front();
                q_.pop_front();
            }
            std::memcpy(j.dst, j.src, j.bytes);
            finish_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<Job> q_;
    size_t pending_ = 0;
    bool stop_ = false, started_ = false;
};

// Entry points make the context's device current for their own HIP calls and put the caller's current device back on
// the way out: a host that created a context on device k while working on device j keeps working on j.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        if (prev != dev) err = hipSetDevice(dev);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ENTER_DEVICE(dev)         \
    DeviceGuard _device_guard(dev); \
    if (_device_guard.err != hipSuccess) return fail_hip(_device_guard.err, "hipSetDevice")

}  // namespace

struct bhg_context {
    int device = 0;
    hipStream_t stream = nullptr;
    // work counters (device): two sets of 8 slice counters, one 256-byte line each.  Consecutive launches alternate
    // between the sets; a launch zeroes the set the next one will use (trace kernels, block 0), so no memset launch
    // per call.  counters_clean = both sets are known to be in that state (false after a failed enqueue)
    unsigned long long *counter = nullptr;
    int counter_set = 0;
    bool counters_clean = false;
    hipStream_t last_stream = nullptr;   // the stream of the last trace launch (launches of a context must stay ordered)
    bool launched = false;
    hipEvent_t ev_order = nullptr;
    bool last_stream_foreign = false;    // last_stream is a caller's handle: ev_order was recorded behind that call
    int num_cus = 0;
    char name[256] = {0};
    // device buffers of the host-buffer entry points, grown on demand
    void *d_in = nullptr;
    size_t d_in_bytes = 0;
    void *d_out = nullptr;
    size_t d_out_bytes = 0;
    // their pipeline: copy streams either side of the compute stream, a two-slot pinned staging ring for callers'
    // pageable arrays, events per slot, worker threads for the host-side copies
    hipStream_t s_in = nullptr, s_out = nullptr;
    void *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    HostCopyPool pool;
    // per-ray workspace of the trace passes (prepare / event / resume records, internal flags)
    void *d_ws = nullptr;
    size_t d_ws_bytes = 0;
    void *d_endws = nullptr;   // end records as workspace of direction-only calls
    size_t d_endws_bytes = 0;
    int32_t last_launch[4] = {0, 0, 0, 0};
    int occupancy[96] = {0};  // resident waves per CU of each trace-kernel variant (0 = not asked yet)
    // optional per-pass timing (bhg_set_profiling)
    bool profiling = false;
    bool ev_valid = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // around prepare | trace | (Kerr) finalize
    bool ev_post = false;                                     // the last profiled call had a finalize pass
};

namespace bhg {
// for bhgeo_frame.hip: a context's worker threads copy a page-locked staging block into the caller's pageable array (a
// 16.8-MB frame through ONE thread's memcpy is 1.0-1.7 ms -- as long as the frame's trace)
void host_copy(bhg_context *c, void *dst, const void *src, size_t bytes, size_t piece) { c->pool.copy(dst, src, bytes, piece); }
}  // namespace bhg

namespace {

int ensure(void **p, size_t *have, size_t need)
{
    if (*have >= need) return BHG_OK;
    if (*p) {
        HIP_TRY(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    size_t want = need + need / 4 + 4096;
    HIP_TRY(hipMalloc(p, want));
    *have = want;
    return BHG_OK;
}

int ensure_pinned(void **p, size_t *have, size_t need)
{
    if (*have >= need) return BHG_OK;
    if (*p) {
        HIP_TRY(hipHostFree(*p));
        *p = nullptr;
        *have = 0;
    }
    HIP_TRY(hipHostMalloc(p, need, hipHostMallocDefault));
    *have = need;
    return BHG_OK;
}

// Is this host address page-locked memory HIP knows (hipHostMalloc / hipHostRegister, e.g. bhg_host_alloc)?  Then the
// copy engines reach it directly; a pageable array goes through the staging ring.
bool is_pinned(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// ... and a whole range [p, p + bytes): the allocation that holds p must hold its last byte too (hipMemGetAddressRange on
// the device alias of the block); where the runtime cannot tell the extent of a host allocation, both ends must at
// least be page-locked and map to one contiguous device range
bool is_pinned_range(const void *p, size_t bytes, void **dev_out)
{
    if (!p || bytes == 0 || !is_pinned(p)) return false;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void *>(p), 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        return false;
    }
    hipDeviceptr_t base = nullptr;
    size_t extent = 0;
    if (hipMemGetAddressRange(&base, &extent, (hipDeviceptr_t)dp) == hipSuccess && base && extent) {
        if ((const char *)dp + bytes > (const char *)base + extent) return false;
    } else {
        (void)hipGetLastError();
        const char *last = (const char *)p + bytes - 1;
        void *dl = nullptr;
        if (!is_pinned(last)) return false;
        if (hipHostGetDevicePointer(&dl, const_cast<char *>(last), 0) != hipSuccess || dl != (char *)dp + bytes - 1) {
            (void)hipGetLastError();
            return false;
        }
    }
    *dev_out = dp;
    return true;
}

// validate_tol (scipy _ivp/common.py:44-51): an rtol below 100 eps is raised to 100 eps -- scipy warns and carries on, and so
// does every solve the reference runs through solve_ivp (README.md:196)
inline double scipy_rtol(double rtol) { return rtol < 100.0 * DBL_EPSILON ? 100.0 * DBL_EPSILON : rtol; }

int validate(const bhg_params *p)
{
    if (!p) return fail(BHG_E_INVALID, "params is NULL");
    if (!(p->r_s >= 0.0) || !std::isfinite(p->r_s)) return fail(BHG_E_INVALID, "r_s must be finite and >= 0");
    if (!(p->lambda_end >= 0.0) || !std::isfinite(p->lambda_end))
        return fail(BHG_E_INVALID, "lambda_end must be finite and >= 0");
    if (!(p->r_exit >= 0.0) || !std::isfinite(p->r_exit)) return fail(BHG_E_INVALID, "r_exit must be finite and >= 0");
    if (p->method == BHG_METHOD_DP54) {
        if (!(p->max_step > 0.0)) return fail(BHG_E_INVALID, "max_step must be > 0 (use +inf for unset)");
        if (!(p->rtol > 0.0) || !(p->atol > 0.0) || !std::isfinite(p->rtol) || !std::isfinite(p->atol))
            return fail(BHG_E_INVALID, "rtol and atol must be finite and > 0");
    } else if (p->method == BHG_METHOD_RK4) {
        if (!(p->h_fixed > 0.0) || !std::isfinite(p->h_fixed)) return fail(BHG_E_INVALID, "h_fixed must be finite and > 0");
    } else {
        return fail(BHG_E_INVALID, "unknown method");
    }
    if (p->rhs_form != BHG_RHS_CHRISTOFFEL && p->rhs_form != BHG_RHS_REDUCED && p->rhs_form != BHG_RHS_KERR_BL)
        return fail(BHG_E_INVALID, "unknown rhs_form");
    if (p->rhs_form == BHG_RHS_KERR_BL) {
        if (!std::isfinite(p->spin) || !(std::fabs(p->spin) < 0.5 * p->r_s))
            return fail(BHG_E_INVALID, "Kerr needs |spin| < M = r_s/2");
    }
    if (p->time_like != 0 && p->time_like != 1) return fail(BHG_E_INVALID, "time_like must be 0 or 1");
    if (p->time_like && p->rhs_form == BHG_RHS_REDUCED)
        return fail(BHG_E_INVALID, "BHG_RHS_REDUCED is the closed form for null rays: time_like needs BHG_RHS_CHRISTOFFEL or BHG_RHS_KERR_BL");
    if (!(p->disk_r_in >= 0.0) || !(p->disk_r_out >= 0.0) || !std::isfinite(p->disk_r_in) || !std::isfinite(p->disk_r_out))
        return fail(BHG_E_INVALID, "disk radii must be finite and >= 0");
    if (p->disk_r_out > 0.0 && p->disk_r_in > p->disk_r_out) return fail(BHG_E_INVALID, "disk_r_in > disk_r_out");
    return BHG_OK;
}

}  // namespace

extern "C" {

int bhg_version(void) { return BHG_ABI_VERSION; }

int bhg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *bhg_last_error(void) { return g_err.c_str(); }

// the binder's handshake: struct sizes as THIS build of the library lays them out
size_t bhg_params_size(void) { return sizeof(bhg_params); }
size_t bhg_camera_size(void) { return sizeof(bhg_camera); }
size_t bhg_scene_size(void) { return sizeof(bhg_scene); }
size_t bhg_frame_scene_size(void) { return sizeof(bhg_frame_scene); }

int bhg_abi_check(int abi_version, size_t params_size, size_t camera_size, size_t scene_size, size_t frame_scene_size)
{
    // (a binding written for an older ABI whose every entry point and struct layout this library still has is served:
    // BHG_ABI_COMPAT_MIN .. BHG_ABI_VERSION)
    if (abi_version < BHG_ABI_COMPAT_MIN || abi_version > BHG_ABI_VERSION)
        return fail(BHG_E_INVALID, "ABI mismatch: the binding was written for ABI " + std::to_string(abi_version) + ", this libbhgeo.so is ABI " +
                                       std::to_string(BHG_ABI_VERSION) + " and serves bindings from ABI " + std::to_string(BHG_ABI_COMPAT_MIN) +
                                       " on (include/bhgeo.h)");
    const struct {
        const char *name;
        size_t theirs, ours;
    } t[] = {{"bhg_params", params_size, sizeof(bhg_params)},
             {"bhg_camera", camera_size, sizeof(bhg_camera)},
             {"bhg_scene", scene_size, sizeof(bhg_scene)},
             {"bhg_frame_scene", frame_scene_size, sizeof(bhg_frame_scene)}};
    for (const auto &e : t)
        if (e.theirs != 0 && e.theirs != e.ours)   // (0: the binding does not declare that struct)
            return fail(BHG_E_INVALID, std::string("ABI mismatch: the binding's ") + e.name + " is " + std::to_string(e.theirs) +
                                           " bytes, the library's is " + std::to_string(e.ours) + " (include/bhgeo.h, ABI " +
                                           std::to_string(BHG_ABI_VERSION) + ")");
    return BHG_OK;
}

int bhg_default_params_sized(bhg_params *p, size_t params_size)
{
    if (!p) return fail(BHG_E_INVALID, "params is NULL");
    if (params_size != sizeof(bhg_params))
        return fail(BHG_E_INVALID, "ABI mismatch: the caller's bhg_params is " + std::to_string(params_size) + " bytes, the library's is " +
                                       std::to_string(sizeof(bhg_params)) + " -- nothing was written");
    bhg_default_params(p);
    return BHG_OK;
}

void bhg_default_params(bhg_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->r_s = 1.0;  // mass 0.5 (RelativisticRenderEngine.py:506), r_s = 2 M (:95)
    p->lambda_end = 50.0;  // integration_depth default (:508)
    p->max_step = INFINITY;  // max_integration_step -1 -> inf (:59-60); property default 1e4 (:507)
    p->rtol = 1e-3;  // scipy rk.py:86
    p->atol = 1e-6;
    p->h_fixed = 0.1;
    p->r_exit = 0.0;
    p->method = BHG_METHOD_DP54;
    p->rhs_form = BHG_RHS_CHRISTOFFEL;
    p->max_steps = 0;
    p->order_blocks = 0;
    p->disk_r_in = 0.0;
    p->disk_r_out = 0.0;  // no disk
    p->spin = 0.0;
    p->time_like = 0;
}

int bhg_create(int device, bhg_context **out)
{
    if (!out) return fail(BHG_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = bhg_device_count();
    if (n <= 0) return fail(BHG_E_NO_DEVICE, "no HIP device visible (libbhgeo has no CPU fallback)");
    if (device < 0 || device >= n) return fail(BHG_E_NO_DEVICE, "device index out of range");
    ENTER_DEVICE(device);
    bhg_context *c = new (std::nothrow) bhg_context();
    if (!c) return fail(BHG_E_NOMEM, "host allocation failed");
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete c;
        return fail_hip(e, "hipGetDeviceProperties");
    }
    c->num_cus = prop.multiProcessorCount;
    std::snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail_hip(e, "hipStreamCreate");
    }
    e = hipMalloc((void **)&c->counter, 2 * 8 * 256);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail_hip(e, "hipMalloc(counter)");
    }
    *out = c;
    return BHG_OK;
}

void bhg_destroy(bhg_context *c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_endws) (void)hipFree(c->d_endws);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    for (int i = 0; i < 2; i++) {
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_k[i]) (void)hipEventDestroy(c->ev_k[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->counter) (void)hipFree(c->counter);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    for (int i = 0; i < 4; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bhg_device_name(bhg_context *c, char *buf, size_t buflen)
{
    if (!c || !buf || buflen == 0) return fail(BHG_E_INVALID, "bad argument");
    std::snprintf(buf, buflen, "%s", c->name);
    return BHG_OK;
}

int bhg_num_cus(bhg_context *c) { return c ? c->num_cus : fail(BHG_E_INVALID, "ctx is NULL"); }

int bhg_synchronize(bhg_context *c)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    ENTER_DEVICE(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BHG_OK;
}

int bhg_set_profiling(bhg_context *c, int enable)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    ENTER_DEVICE(c->device);
    if (enable && !c->ev[0])
        for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&c->ev[i]));
    c->profiling = enable != 0;
    c->ev_valid = false;
    return BHG_OK;
}

int bhg_last_pass_ms(bhg_context *c, float out_ms[3])
{
    if (!c || !out_ms) return fail(BHG_E_INVALID, "bad argument");
    if (!c->ev_valid) return fail(BHG_E_INVALID, "no profiled trace call yet (bhg_set_profiling)");
    ENTER_DEVICE(c->device);
    HIP_TRY(hipEventSynchronize(c->ev[c->ev_post ? 3 : 2]));
    for (int i = 0; i < 2; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], c->ev[i], c->ev[i + 1]));
    // third slot: the pass AFTER the trace kernel -- Kerr's Boyer-Lindquist -> Cartesian finalize (0 for the
    // Schwarzschild forms: events are resolved inside the trace kernel, there is no pass after it)
    out_ms[2] = 0.0f;
    if (c->ev_post) HIP_TRY(hipEventElapsedTime(&out_ms[2], c->ev[2], c->ev[3]));
    return BHG_OK;
}

void *bhg_context_stream(bhg_context *c) { return c ? (void *)c->stream : nullptr; }

int bhg_last_launch(bhg_context *c, int32_t out[4])
{
    if (!c || !out) return fail(BHG_E_INVALID, "bad argument");
    std::memcpy(out, c->last_launch, sizeof(c->last_launch));
    return BHG_OK;
}

}  // extern "C"

namespace {

int validate_spheres(const bhg_params *p, const double *spheres, int32_t n_spheres)
{
    if (n_spheres < 0 || n_spheres > BHG_MAX_SPHERES) return fail(BHG_E_INVALID, "n_spheres must be in [0, BHG_MAX_SPHERES]");
    if (n_spheres == 0) return BHG_OK;
    if (!spheres) return fail(BHG_E_INVALID, "spheres is NULL");
    for (int j = 0; j < n_spheres; j++) {
        const double *sp = spheres + 4 * j;
        if (!std::isfinite(sp[0]) || !std::isfinite(sp[1]) || !std::isfinite(sp[2]) || !std::isfinite(sp[3]) || !(sp[3] > 0.0))
            return fail(BHG_E_INVALID, "sphere centres must be finite and radii finite and > 0");
    }
    return BHG_OK;
}

// ONE launch: n <= BHG_MAX_RAYS_PER_LAUNCH rays (the kernels form a ray's byte offsets in 32 bits).
// d_end [n][6], or -- d_end == nullptr -- d_end_dir [n][3]: only the direction half of the final states is produced
int trace_device_one(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres,
                     const double *x0_shared, const double *d_x0, const double *d_k0, size_t n, double *d_end,
                     uint8_t *d_flags, uint32_t *d_n_steps, uint32_t *d_n_accepted, int8_t *d_object_id, void *stream,
                     double *d_end_dir)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    int rc = validate(p);
    if (rc != BHG_OK) return rc;
    rc = validate_spheres(p, spheres, n_spheres);
    if (rc != BHG_OK) return rc;
    if (n == 0) return BHG_OK;
    if (!d_k0 || (!d_end && !d_end_dir)) return fail(BHG_E_INVALID, "k0 / end is NULL");
    if (!x0_shared && !d_x0) return fail(BHG_E_INVALID, "neither x0_shared nor d_x0 given");
    if (n > bhg::BHG_MAX_RAYS_PER_LAUNCH) return fail(BHG_E_INVALID, "internal: more rays than one launch takes");
    ENTER_DEVICE(c->device);
    hipStream_t s = (hipStream_t)stream;
    // Launches of one context share its work counters (launch K zeroes the set launch K + 1 counts on) and its workspace:
    // they must execute in the order they were issued.  On ONE stream they do; a call that arrives on ANOTHER stream than
    // the previous one is ordered behind it here (an event on the old stream, a wait on the new one) -- it then cannot
    // overlap the previous call, but it cannot corrupt it either (two traces that are to overlap need two contexts).
    // The previous call's stream may be a handle the CALLER owns -- and may have destroyed since: it is never touched
    // again.  A call on a caller's stream leaves an event behind it at its own end (below, ev_order recorded on that stream
    // while the caller is still inside the call); only the context's own stream and the null stream, which cannot go away,
    // are recorded on after the fact.
    if (c->launched && s != c->last_stream) {
        if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
        if (!c->last_stream_foreign) HIP_TRY(hipEventRecord(c->ev_order, c->last_stream));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_order, 0));
    }
    c->last_stream = s;
    c->launched = true;

    // Workspace, grown on demand (the first call at a new size allocates; steady-state calls do not):
    //   ws      [n][8] doubles  only in a build without the inlined prepare: that pass's records {a0, h0, r0, 0, E, L} (the
    //                           trace kernels work their start records out themselves; parked steps live in the waves' LDS pools)
    //   flags   [n] bytes       when the caller does not want flags
    //   n_steps / n_accepted [n] u32 when the caller does not want them (the kernels never test these pointers)
    const bool has_exit = p->r_exit > 0.0;
    const bool kerr = p->rhs_form == BHG_RHS_KERR_BL;
    // the kernels' right-hand-side id: the time-like Christoffel form is one of its own (the Kerr kernels take the norm at the start)
    const int rhs_id = (p->time_like && p->rhs_form == BHG_RHS_CHRISTOFFEL) ? bhg::BHG_RHS_CHRISTOFFEL_TL_ : p->rhs_form;
    const bool needs_ws = bhg::needs_prepare_ws(p->rhs_form);   // only in a build without the inlined prepare
    const size_t sz_ws = needs_ws ? n * 8 * sizeof(double) : 0;
    const size_t sz_flags = d_flags ? 0 : ((n + 7) & ~size_t(7));
    const size_t sz_u32 = n * sizeof(uint32_t);
    const size_t sz_steps = !d_n_steps ? sz_u32 : 0;
    const size_t sz_acc = !d_n_accepted ? sz_u32 : 0;
    rc = ensure(&c->d_ws, &c->d_ws_bytes, sz_ws + sz_flags + sz_steps + sz_acc + 64);
    if (rc != BHG_OK) return rc;
    if (!d_end && kerr) {
        // direction-only Kerr call: the end records are workspace (the Boyer-Lindquist states the finalize pass converts),
        // final directions are split off into d_end_dir afterwards
        rc = ensure(&c->d_endws, &c->d_endws_bytes, n * 6 * sizeof(double) + 64);
        if (rc != BHG_OK) return rc;
        d_end = (double *)c->d_endws;
    } else if (d_end) {
        d_end_dir = nullptr;
    }
    char *wsb = (char *)c->d_ws;
    uint8_t *w_flags = (uint8_t *)(wsb + sz_ws);
    uint32_t *w_steps = (uint32_t *)(wsb + sz_ws + sz_flags);
    uint32_t *w_acc = (uint32_t *)(wsb + sz_ws + sz_flags + sz_steps);

    bhg::TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.k0 = d_k0;
    a.x0 = d_x0;
    a.end = d_end;
    // (Kerr end states are converted from Boyer-Lindquist by a pass over whole records: directions are split off after it)
    a.end_dir = kerr ? nullptr : d_end_dir;
    a.ws = needs_ws ? (double *)c->d_ws : nullptr;
    a.flags = d_flags ? d_flags : w_flags;
    a.n_steps = d_n_steps ? d_n_steps : w_steps;
    a.n_accepted = d_n_accepted ? d_n_accepted : w_acc;
    a.counter = c->counter + (size_t)c->counter_set * (8 * 256 / sizeof(unsigned long long));
    a.counter_next = c->counter + (size_t)(c->counter_set ^ 1) * (8 * 256 / sizeof(unsigned long long));
    a.n = n;
    if (!d_x0) {
        a.x0s[0] = x0_shared[0];
        a.x0s[1] = x0_shared[1];
        a.x0s[2] = x0_shared[2];
    }
    a.r_s = p->r_s;
    a.lambda_end = p->lambda_end;
    a.max_step = p->max_step;
    a.rtol = scipy_rtol(p->rtol);
    a.atol = p->atol;
    a.h_fixed = p->h_fixed;
    a.r_exit = p->r_exit;
    a.disk_r_in = p->disk_r_in;
    a.disk_r_out = p->disk_r_out;
    a.spin = p->spin;
    a.mu2 = p->time_like ? 1.0 : 0.0;
    a.r_hor = p->r_s;
    a.from_records = 0;
    a.ws_stride = 6;
    if (p->rhs_form == BHG_RHS_KERR_BL) {
        const double M = 0.5 * p->r_s;
        a.r_hor = (M + std::sqrt(M * M - p->spin * p->spin)) * (1.0 + BHG_KERR_HORIZON_MARGIN);
        a.from_records = needs_ws ? 1 : 0;     // (only a build without the inlined prepare starts Kerr rays from records)
        a.ws_stride = 8;
    }
    a.max_steps = p->max_steps ? p->max_steps : (1u << 20);
    a.min_step_cap = 40.0 * std::nextafter(std::fmax(p->lambda_end, 1.0), INFINITY) * 2.220446049250313e-16;
    // (lambda_end = 0: every ray is "already at t_bound" at its first step -- the rare-path prologue handles that, and an
    // infinite cap sends every lane there)
    if (p->lambda_end == 0.0) a.min_step_cap = INFINITY;
    // work-order hint: honoured when the call is that many equal blocks of whole 64-ray batches
    bool order_hint = p->order_blocks > 1 && n % p->order_blocks == 0 && (n / p->order_blocks) % 64 == 0;
#ifdef BHG_TUNING
    if (std::getenv("BHGEO_NO_ORDER_HINT")) order_hint = false;  // A/B aid, tuning builds only
#endif
    if (order_hint) {
        a.order_blocks = (int32_t)p->order_blocks;
        a.order_block_len = n / p->order_blocks;
    }
    a.object_id = d_object_id;
    a.n_spheres = n_spheres;
    for (int j = 0; j < n_spheres; j++)
        for (int q = 0; q < 4; q++) a.spheres[j][q] = spheres[4 * j + q];
    // kernel variant: bit 0 exit sphere, bit 1 disk, bit 2 objects.  With objects: 5 = exit sphere and no disk (the
    // orbiting-sphere frames), otherwise 7, which tests for the exit sphere and the disk at run time
    int evt = n_spheres > 0 ? ((has_exit && !(p->disk_r_out > 0.0)) ? 5 : 7)
                            : ((has_exit ? 1 : 0) | (p->disk_r_out > 0.0 ? 2 : 0));
    if (rhs_id == bhg::BHG_RHS_CHRISTOFFEL_TL_) evt = 7;    // (the time-like form exists in the all-events variant only)

    // resident waves per CU of the trace kernel variant: asked of the runtime once per variant and context
    const int vkey = ((p->method & 1) * 3 + (p->rhs_form % 3)) * 8 + evt + (rhs_id == bhg::BHG_RHS_CHRISTOFFEL_TL_ ? 48 : 0);
    int per_cu = c->occupancy[vkey];
    if (per_cu == 0) {
        HIP_TRY(bhg::trace_occupancy(p->method, rhs_id, evt, &per_cu));
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 32) per_cu = 32;
        c->occupancy[vkey] = per_cu;
    }
#ifdef BHG_TUNING
    if (const char *ov = std::getenv("BHGEO_WAVES_PER_CU")) {  // tuning / diagnostic override
        int v = std::atoi(ov);
        if (v >= 1 && v <= 64) per_cu = v;
    }
#endif
    // persistent waves: fill every resident wave slot once; never more waves than 64-ray batches
    size_t batches = (n + 63) / 64;
    size_t grid = (size_t)per_cu * (size_t)c->num_cus;
    if (grid > batches) grid = batches;
#ifdef BHG_DIAG
    {
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) HIP_TRY(hipMalloc((void **)&dbuf, 65536 * 8 * sizeof(unsigned long long)));
        a.diag = dbuf;
        a.dbg_idx = std::getenv("BHGEO_DBG_IDX") ? (uint32_t)std::atoi(std::getenv("BHGEO_DBG_IDX")) : 0xFFFFFFFFu;
        if (const char *path = std::getenv("BHGEO_DIAG_DUMP")) {
            static unsigned long long host[65536 * 8];
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(path, "wb")) {
                std::fwrite(host, 1, sizeof(host), f);
                std::fclose(f);
            }
            HIP_TRY(hipMemset(dbuf + BHG_DIAG_HIST, 0, 131 * sizeof(unsigned long long)));   // (the histogram accumulates: one launch per dump)
        }
    }
#endif
    // ONE persistent launch finishes every ray: events are resolved and rays resumed inside the trace kernel, so
    // the call only enqueues (Kerr: trace, finalize) and returns
    if (!c->counters_clean) HIP_TRY(hipMemsetAsync(c->counter, 0, 2 * 8 * 256, s));   // first call, or after a failed enqueue
    c->counters_clean = false;
    HIP_TRY(bhg::launch_trace(a, p->method, rhs_id, evt, (int)grid, s, c->profiling ? c->ev : nullptr));
    c->counter_set ^= 1;      // (the launch is in the stream: the next call of this context counts on the set it zeroes)
    c->counters_clean = true;
    c->ev_valid = c->profiling;
    c->last_launch[3] = 1;
    c->ev_post = false;
    if (p->rhs_form == BHG_RHS_KERR_BL) {
        // (direction-only calls: the finalize pass writes the Cartesian exit directions straight into d_end_dir)
        HIP_TRY(bhg::launch_kerr_finalize(a, d_end_dir, s));
        if (c->profiling) {
            HIP_TRY(hipEventRecord(c->ev[3], s));
            c->ev_post = true;
        }
    }
    c->last_launch[0] = (int32_t)grid;
    c->last_launch[1] = 64;
    c->last_launch[2] = per_cu;
    // a caller's stream: leave the ordering event behind this call now, while the handle is certainly alive (the next
    // call on another stream waits on it and never touches this stream again)
    c->last_stream_foreign = s != nullptr && s != c->stream;
    if (c->last_stream_foreign) {
        if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_order, s));
    }
    return BHG_OK;
}

// A trace call of any size: launches of at most BHG_MAX_RAYS_PER_LAUNCH rays, one after the other on the caller's stream
// (BASELINE's largest frame, 2048 x 2048 x 16 = 2^26 rays, is one launch).  Every ray is its own ODE: the split never
// changes a result.  The work-order hint describes ONE launch's rays and is dropped for a split call.
int trace_device_impl(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres,
                      const double *x0_shared, const double *d_x0, const double *d_k0, size_t n, double *d_end,
                      uint8_t *d_flags, uint32_t *d_n_steps, uint32_t *d_n_accepted, int8_t *d_object_id, void *stream,
                      double *d_end_dir = nullptr)
{
    if (n <= bhg::BHG_MAX_RAYS_PER_LAUNCH)
        return trace_device_one(c, p, spheres, n_spheres, x0_shared, d_x0, d_k0, n, d_end, d_flags, d_n_steps, d_n_accepted,
                                d_object_id, stream, d_end_dir);
    if (!p) return fail(BHG_E_INVALID, "params is NULL");
    if (n > 0xFFFFFFFFull) return fail(BHG_E_INVALID, "n must be < 2^32 per call");
    bhg_params q = *p;
    q.order_blocks = 0;
    for (size_t off = 0; off < n; off += bhg::BHG_MAX_RAYS_PER_LAUNCH) {
        const size_t m = std::min((size_t)bhg::BHG_MAX_RAYS_PER_LAUNCH, n - off);
        const int rc = trace_device_one(c, &q, spheres, n_spheres, x0_shared, d_x0 ? d_x0 + off * 3 : nullptr,
                                        d_k0 ? d_k0 + off * 3 : nullptr, m, d_end ? d_end + off * 6 : nullptr,
                                        d_flags ? d_flags + off : nullptr, d_n_steps ? d_n_steps + off : nullptr,
                                        d_n_accepted ? d_n_accepted + off : nullptr, d_object_id ? d_object_id + off : nullptr,
                                        stream, d_end_dir ? d_end_dir + off * 3 : nullptr);
        if (rc != BHG_OK) return rc;
    }
    c->last_launch[3] = (int32_t)((n + bhg::BHG_MAX_RAYS_PER_LAUNCH - 1) / bhg::BHG_MAX_RAYS_PER_LAUNCH);
    return BHG_OK;
}

}  // namespace

extern "C" {

int bhg_trace_device(bhg_context *c, const bhg_params *p, const double *x0_shared, const double *d_x0,
                     const double *d_k0, size_t n, double *d_end, uint8_t *d_flags, uint32_t *d_n_steps,
                     uint32_t *d_n_accepted, void *stream)
{
    return trace_device_impl(c, p, nullptr, 0, x0_shared, d_x0, d_k0, n, d_end, d_flags, d_n_steps, d_n_accepted, nullptr,
                             stream);
}

int bhg_trace_dir_device(bhg_context *c, const bhg_params *p, const double *x0_shared, const double *d_x0,
                         const double *d_k0, size_t n, double *d_end_dir, uint8_t *d_flags, uint32_t *d_n_steps,
                         uint32_t *d_n_accepted, void *stream)
{
    if (n && !d_end_dir) return fail(BHG_E_INVALID, "end_dir is NULL");
    return trace_device_impl(c, p, nullptr, 0, x0_shared, d_x0, d_k0, n, nullptr, d_flags, d_n_steps, d_n_accepted, nullptr,
                             stream, d_end_dir);
}

int bhg_trace_objects_device(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres,
                             const double *x0_shared, const double *d_x0, const double *d_k0, size_t n, double *d_end,
                             uint8_t *d_flags, uint32_t *d_n_steps, uint32_t *d_n_accepted, int8_t *d_object_id,
                             void *stream)
{
    return trace_device_impl(c, p, spheres, n_spheres, x0_shared, d_x0, d_k0, n, d_end, d_flags, d_n_steps, d_n_accepted,
                             d_object_id, stream);
}

int bhg_trace(bhg_context *c, const bhg_params *p, const double *x0, int x0_is_shared, const double *k0,
              size_t n, double *end, uint8_t *flags, uint32_t *n_steps, uint32_t *n_accepted)
{
    return bhg_trace_objects(c, p, nullptr, 0, x0, x0_is_shared, k0, n, end, flags, n_steps, n_accepted, nullptr);
}

}  // extern "C"

struct bhg_rays {
    bhg_context *ctx = nullptr;
    double *d_k0 = nullptr;  // [n][3], ray s * n_pixels + p = sample s of pixel p
    size_t n = 0;
    double origin[3] = {0, 0, 0};
};

namespace {

// what a pipelined call reads and writes on the host side
struct PipeIO {
    const double *h_k0 = nullptr, *h_x0 = nullptr;  // caller's rays (h_x0: per-ray origins) ...
    const double *d_k0 = nullptr;                   // ... or rays already resident on the device (no upload)
    const double *x0_shared = nullptr;              // host [3] when the origin is shared
    double *end = nullptr, *loc = nullptr, *dir = nullptr;  // end [n][6] and / or its halves [n][3]
    uint8_t *flags = nullptr;
    uint32_t *steps = nullptr, *acc = nullptr;
    int8_t *obj = nullptr;
};

// The host-buffer calls as a pipeline over chunks of rays:
//   host   : caller's k0 (x0) chunk -> pinned ring          (worker threads; skipped for pinned caller memory)
//   s_in   : H2D                                            (copy engine; nothing to do for resident rays)
//   stream : trace (one launch per chunk), end -> loc / dir (compute)
//   s_out  : D2H of the arrays the caller asked for         (the other copy engine)
//   host   : pinned ring -> caller's arrays                 (worker threads; skipped for pinned caller memory)
// Chunk c+1 is staged and uploaded while chunk c is traced and chunk c-1 comes back.  Results do not depend on the
// chunking: every ray is its own ODE.
int pipeline_impl(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres, const PipeIO &io, size_t n)
{
    ENTER_DEVICE(c->device);
    if (!c->s_in) {
        HIP_TRY(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    // Whatever way this call ends, nothing of it may still be running when it returns: copies and kernels of earlier
    // chunks DMA into the caller's page-locked arrays and into the pinned ring, and the next call's ensure() may free
    // buffers they use.  (The success path has waited already; the guard then costs three no-op synchronisations.)
    struct StreamsQuiet {
        bhg_context *c;
        ~StreamsQuiet()
        {
            if (c->s_in) (void)hipStreamSynchronize(c->s_in);
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            if (c->s_out) (void)hipStreamSynchronize(c->s_out);
        }
    } quiet{c};
    const bool upload = io.d_k0 == nullptr;
    const bool per_ray_x0 = io.h_x0 != nullptr;
    const bool dir_only = io.dir && !io.end && !io.loc;   // the trace writes the directions itself: no records, no split pass
    const bool split = (io.loc || io.dir) && !dir_only;
    // device arrays for the whole call (chunks are sub-ranges of each)
    const size_t in_bytes = upload ? n * 3 * sizeof(double) * (per_ray_x0 ? 2 : 1) : 0;
    const size_t off_flags = dir_only ? 0 : n * 6 * sizeof(double);   // (no record array in a direction-only call)
    const size_t off_steps = off_flags + ((n + 7) & ~size_t(7));
    const size_t off_acc = off_steps + n * sizeof(uint32_t);
    const size_t off_obj = off_acc + n * sizeof(uint32_t);
    const size_t off_loc = (off_obj + n + 7) & ~size_t(7);
    const size_t off_dir = off_loc + (split ? n * 3 * sizeof(double) : 0);
    const size_t out_bytes = off_dir + ((split || dir_only) ? n * 3 * sizeof(double) : 0);
    int rc = BHG_OK;
    if (upload) {
        rc = ensure(&c->d_in, &c->d_in_bytes, in_bytes);
        if (rc != BHG_OK) return rc;
    }
    rc = ensure(&c->d_out, &c->d_out_bytes, out_bytes);
    if (rc != BHG_OK) return rc;
    double *d_k0u = (double *)c->d_in;
    const double *d_k0 = upload ? d_k0u : io.d_k0;
    double *d_x0 = per_ray_x0 ? d_k0u + n * 3 : nullptr;
    char *o = (char *)c->d_out;
    double *d_end = (double *)o, *d_loc = (double *)(o + off_loc), *d_dir = (double *)(o + off_dir);
    uint8_t *d_flags = (uint8_t *)(o + off_flags);
    uint32_t *d_steps = (uint32_t *)(o + off_steps), *d_acc = (uint32_t *)(o + off_acc);
    int8_t *d_obj = io.obj ? (int8_t *)(o + off_obj) : nullptr;

    const size_t chunk = size_t(1) << 20;  // rays per chunk: 24 MB up, 57 MB back
    const size_t n_chunks = (n + chunk - 1) / chunk;
    const size_t cmax = std::min(chunk, n);
    const bool pin_k0 = !upload || is_pinned(io.h_k0), pin_x0 = !per_ray_x0 || is_pinned(io.h_x0);
    struct OutArr {
        void *host;
        const char *dev;
        size_t elem;
        bool pinned;
        size_t ring_off;
    } outs[7] = {{io.end, (const char *)d_end, 48, false, 0},   {io.loc, (const char *)d_loc, 24, false, 0},
                 {io.dir, (const char *)d_dir, 24, false, 0},   {io.flags, (const char *)d_flags, 1, false, 0},
                 {io.steps, (const char *)d_steps, 4, false, 0}, {io.acc, (const char *)d_acc, 4, false, 0},
                 {io.obj, (const char *)d_obj, 1, false, 0}};
    size_t so_slot = 0;
    bool stage_out = false;
    for (auto &a : outs) {
        if (!a.host) continue;
        a.pinned = is_pinned(a.host);
        a.ring_off = so_slot;
        so_slot += (cmax * a.elem + 63) & ~size_t(63);
        stage_out = stage_out || !a.pinned;
    }
    const size_t si_x0 = cmax * 24, si_slot = cmax * 48;
    if (!pin_k0 || !pin_x0) {
        rc = ensure_pinned(&c->pin_in, &c->pin_in_bytes, 2 * si_slot);
        if (rc != BHG_OK) return rc;
    }
    if (stage_out) {
        rc = ensure_pinned(&c->pin_out, &c->pin_out_bytes, 2 * so_slot);
        if (rc != BHG_OK) return rc;
    }

    auto copy_out = [&](size_t ch) -> int {  // host side of chunk ch's way back
        const int slot = (int)(ch & 1);
        const size_t off = ch * chunk, m = std::min(chunk, n - off);
        HIP_TRY(hipEventSynchronize(c->ev_out[slot]));
        const char *ps = (const char *)c->pin_out + (size_t)slot * so_slot;
        for (auto &a : outs)
            if (a.host && !a.pinned) c->pool.copy((char *)a.host + off * a.elem, ps + a.ring_off, m * a.elem);
        return BHG_OK;
    };

    for (size_t ch = 0; ch < n_chunks; ch++) {
        const int slot = (int)(ch & 1);
        const size_t off = ch * chunk, m = std::min(chunk, n - off);
        if (upload) {
            char *pi = (char *)c->pin_in + (size_t)slot * si_slot;
            // the slot's previous upload (chunk ch - 2) must have left the staging memory
            if (ch >= 2 && (!pin_k0 || !pin_x0)) HIP_TRY(hipEventSynchronize(c->ev_in[slot]));
            const void *src_k0 = io.h_k0 + off * 3, *src_x0 = per_ray_x0 ? io.h_x0 + off * 3 : nullptr;
            if (!pin_k0) {
                c->pool.copy(pi, src_k0, m * 24);
                src_k0 = pi;
            }
            if (per_ray_x0 && !pin_x0) {
                c->pool.copy(pi + si_x0, src_x0, m * 24);
                src_x0 = pi + si_x0;
            }
            HIP_TRY(hipMemcpyAsync(d_k0u + off * 3, src_k0, m * 24, hipMemcpyHostToDevice, c->s_in));
            if (per_ray_x0) HIP_TRY(hipMemcpyAsync(d_x0 + off * 3, src_x0, m * 24, hipMemcpyHostToDevice, c->s_in));
            HIP_TRY(hipEventRecord(c->ev_in[slot], c->s_in));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in[slot], 0));
        }
        rc = trace_device_impl(c, p, spheres, n_spheres, io.x0_shared, per_ray_x0 ? d_x0 + off * 3 : nullptr, d_k0 + off * 3, m,
                               dir_only ? nullptr : d_end + off * 6, d_flags + off, d_steps + off, d_acc + off,
                               d_obj ? d_obj + off : nullptr, c->stream, dir_only ? d_dir + off * 3 : nullptr);
        if (rc != BHG_OK) return rc;
        if (split)
            HIP_TRY(bhg::launch_split_end(d_end + off * 6, m, io.loc ? d_loc + off * 3 : nullptr, io.dir ? d_dir + off * 3 : nullptr,
                                          c->stream));
        HIP_TRY(hipEventRecord(c->ev_k[slot], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->s_out, c->ev_k[slot], 0));
        char *po = (char *)c->pin_out + (size_t)slot * so_slot;
        for (auto &a : outs)
            if (a.host)
                HIP_TRY(hipMemcpyAsync(a.pinned ? (void *)((char *)a.host + off * a.elem) : (void *)(po + a.ring_off),
                                       a.dev + off * a.elem, m * a.elem, hipMemcpyDeviceToHost, c->s_out));
        HIP_TRY(hipEventRecord(c->ev_out[slot], c->s_out));
        // while the GPU works on this chunk: the previous chunk's results go from the ring to the caller's arrays
        if (ch >= 1) {
            rc = copy_out(ch - 1);
            if (rc != BHG_OK) return rc;
        }
    }
    rc = copy_out(n_chunks - 1);
    if (rc != BHG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->s_out));
    return BHG_OK;
}

}  // namespace

extern "C" {

int bhg_trace_objects(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres, const double *x0,
                      int x0_is_shared, const double *k0, size_t n, double *end, uint8_t *flags, uint32_t *n_steps,
                      uint32_t *n_accepted, int8_t *object_id)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    int rc = validate(p);
    if (rc != BHG_OK) return rc;
    rc = validate_spheres(p, spheres, n_spheres);
    if (rc != BHG_OK) return rc;
    if (n == 0) return BHG_OK;
    if (!x0 || !k0 || !end) return fail(BHG_E_INVALID, "x0 / k0 / end is NULL");
    if (n > 0xFFFFFFFFull) return fail(BHG_E_INVALID, "n must be < 2^32 per call");
    PipeIO io;
    io.h_k0 = k0;
    io.h_x0 = x0_is_shared ? nullptr : x0;
    io.x0_shared = x0_is_shared ? x0 : nullptr;
    io.end = end;
    io.flags = flags;
    io.steps = n_steps;
    io.acc = n_accepted;
    io.obj = object_id;
    return pipeline_impl(c, p, spheres, n_spheres, io, n);
}

/* ---- rays resident on the device ------------------------------------------------------- */
int bhg_rays_create(bhg_context *c, const bhg_camera *cam, const double *jitter, int jitter_is_compact, const int64_t *pixels,
                    size_t n_pixels, bhg_rays **out)
{
    if (!out) return fail(BHG_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!c || !cam) return fail(BHG_E_INVALID, "ctx / camera is NULL");
    if (cam->width <= 0 || cam->height <= 0 || cam->samples <= 0) return fail(BHG_E_INVALID, "width, height, samples must be > 0");
    const size_t frame_px = (size_t)cam->width * (size_t)cam->height;
    if (!pixels) n_pixels = frame_px;
    if (n_pixels == 0) {
        // an empty pixel list -- a shard that was dealt no tile (more devices than tiles) -- is a ray set of 0 rays:
        // bhg_rays_count() = 0, bhg_rays_trace(..., 0, 0, ...) a no-op
        bhg_rays *r0 = new (std::nothrow) bhg_rays();
        if (!r0) return fail(BHG_E_NOMEM, "host allocation failed");
        r0->ctx = c;
        r0->n = 0;
        std::memcpy(r0->origin, cam->origin, sizeof(r0->origin));
        *out = r0;
        return BHG_OK;
    }
    if (jitter_is_compact && !jitter) return fail(BHG_E_INVALID, "compact jitter stream is NULL");
    const size_t n = n_pixels * (size_t)cam->samples;
    if (n > 0xFFFFFFFFull) return fail(BHG_E_INVALID, "more than 2^32 rays");
    ENTER_DEVICE(c->device);
    bhg_rays *r = new (std::nothrow) bhg_rays();
    if (!r) return fail(BHG_E_NOMEM, "host allocation failed");
    r->ctx = c;
    r->n = n;
    std::memcpy(r->origin, cam->origin, sizeof(r->origin));
    hipError_t e = hipMalloc((void **)&r->d_k0, n * 3 * sizeof(double));
    if (e != hipSuccess) {
        delete r;
        return fail_hip(e, "hipMalloc(rays)");
    }
    // the jitter stream and the pixel list are only needed to generate the rays: staged in the call's own buffers
    const size_t n_jit = jitter ? 2 * (size_t)cam->samples * (jitter_is_compact ? n_pixels : frame_px) : 0;
    const size_t tmp_bytes = n_jit * sizeof(double) + (pixels ? n_pixels * sizeof(int64_t) : 0);
    int rc = ensure(&c->d_in, &c->d_in_bytes, tmp_bytes + 64);
    if (rc != BHG_OK) {
        (void)hipFree(r->d_k0);
        delete r;
        return rc;
    }
    double *d_jit = (double *)c->d_in;
    int64_t *d_pix = (int64_t *)((char *)c->d_in + n_jit * sizeof(double));
    bhg::RaygenArgs a;
    std::memset(&a, 0, sizeof(a));
    hipError_t err = hipSuccess;
    if (jitter) err = hipMemcpyAsync(d_jit, jitter, n_jit * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (err == hipSuccess && pixels) err = hipMemcpyAsync(d_pix, pixels, n_pixels * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    a.jitter = jitter ? d_jit : nullptr;
    a.compact = jitter_is_compact ? 1 : 0;
    a.pixels = pixels ? d_pix : nullptr;
    a.k0 = r->d_k0;
    a.n_pixels = n_pixels;
    a.width = cam->width;
    a.height = cam->height;
    a.samples = cam->samples;
    a.fov_x = cam->fov_x;
    a.fov_y = cam->fov_y;
    static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    a.rotate = std::memcmp(cam->rot, eye, sizeof(eye)) != 0;
    std::memcpy(a.rot, cam->rot, sizeof(a.rot));
    if (err == hipSuccess) err = bhg::launch_raygen(a, c->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
    if (err != hipSuccess) {
        (void)hipFree(r->d_k0);
        delete r;
        return fail_hip(err, "ray generation");
    }
    *out = r;
    return BHG_OK;
}

size_t bhg_rays_count(const bhg_rays *r) { return r ? r->n : 0; }

void bhg_rays_destroy(bhg_rays *r)
{
    if (!r) return;
    if (r->d_k0) {
        DeviceGuard guard(r->ctx->device);
        (void)hipFree(r->d_k0);
    }
    delete r;
}

int bhg_rays_trace(bhg_rays *r, const bhg_params *p, const double *spheres, int32_t n_spheres, size_t first, size_t n,
                   double *end, double *end_loc, double *end_dir, uint8_t *flags, uint32_t *n_steps, uint32_t *n_accepted,
                   int8_t *object_id)
{
    if (!r) return fail(BHG_E_INVALID, "rays is NULL");
    int rc = validate(p);
    if (rc != BHG_OK) return rc;
    rc = validate_spheres(p, spheres, n_spheres);
    if (rc != BHG_OK) return rc;
    if (first > r->n || n > r->n - first) return fail(BHG_E_INVALID, "ray range out of bounds");
    if (n == 0) return BHG_OK;
    if (!end && !end_loc && !end_dir && !flags) return fail(BHG_E_INVALID, "no result array given");
    PipeIO io;
    io.d_k0 = r->d_k0 + first * 3;
    io.x0_shared = r->origin;
    io.end = end;
    io.loc = end_loc;
    io.dir = end_dir;
    io.flags = flags;
    io.steps = n_steps;
    io.acc = n_accepted;
    io.obj = object_id;
    return pipeline_impl(r->ctx, p, spheres, n_spheres, io, n);
}

int bhg_host_alloc(bhg_context *c, size_t bytes, void **out)
{
    if (!out) return fail(BHG_E_INVALID, "out is NULL");
    *out = nullptr;
    if (bytes == 0) return BHG_OK;
    if (c) {   // with the owning context's device current; without a context: whatever device the caller is on
        ENTER_DEVICE(c->device);
        HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
        return BHG_OK;
    }
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return BHG_OK;
}

int bhg_host_free(bhg_context *c, void *p)
{
    (void)c;
    if (!p) return BHG_OK;
    HIP_TRY(hipHostFree(p));
    return BHG_OK;
}

int bhg_raygen_device(bhg_context *c, int32_t width, int32_t height, int32_t samples, double fov_x, double fov_y,
                      const double *rot9, const double *d_jitter, const int64_t *d_pixels, size_t n_pixels,
                      double *d_k0, void *stream)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    if (width <= 0 || height <= 0 || samples <= 0) return fail(BHG_E_INVALID, "width, height, samples must be > 0");
    if (n_pixels == 0) return BHG_OK;
    if (!d_jitter || !d_k0) return fail(BHG_E_INVALID, "jitter / k0 is NULL");
    if (!d_pixels && n_pixels != (size_t)width * (size_t)height)
        return fail(BHG_E_INVALID, "n_pixels must be width*height when no pixel list is given");
    ENTER_DEVICE(c->device);
    bhg::RaygenArgs a;
    std::memset(&a, 0, sizeof(a));
    a.jitter = d_jitter;
    a.pixels = d_pixels;
    a.k0 = d_k0;
    a.n_pixels = n_pixels;
    a.width = width;
    a.height = height;
    a.samples = samples;
    a.fov_x = fov_x;
    a.fov_y = fov_y;
    a.rotate = 0;
    if (rot9) {
        static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        a.rotate = std::memcmp(rot9, eye, sizeof(eye)) != 0;
        std::memcpy(a.rot, rot9, sizeof(a.rot));
    }
    HIP_TRY(bhg::launch_raygen(a, (hipStream_t)stream));
    return BHG_OK;
}

int bhg_shade_device(bhg_context *c, const double *d_end, const uint8_t *d_flags, size_t n_pixels, int32_t samples,
                     const float *d_sky, int32_t sky_w, int32_t sky_h, double *d_rgba, void *stream)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    if (samples <= 0 || sky_w <= 0 || sky_h <= 0) return fail(BHG_E_INVALID, "samples, sky_w, sky_h must be > 0");
    if (n_pixels == 0) return BHG_OK;
    if (!d_end || !d_flags || !d_sky || !d_rgba) return fail(BHG_E_INVALID, "NULL device pointer");
    ENTER_DEVICE(c->device);
    bhg::ShadeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.end = d_end;
    a.flags = d_flags;
    a.sky = d_sky;
    a.rgba = d_rgba;
    a.n_pixels = n_pixels;
    a.samples = samples;
    a.sky_w = sky_w;
    a.sky_h = sky_h;
    HIP_TRY(bhg::launch_shade(a, (hipStream_t)stream));
    return BHG_OK;
}

int bhg_shade_dir_device(bhg_context *c, const double *d_end_dir, const uint8_t *d_flags, size_t n_pixels, int32_t samples,
                         const float *d_sky, int32_t sky_w, int32_t sky_h, double *d_rgba, float *d_rgba_f32,
                         const int64_t *d_scatter, void *stream)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    if (samples <= 0 || sky_w <= 0 || sky_h <= 0) return fail(BHG_E_INVALID, "samples, sky_w, sky_h must be > 0");
    if (n_pixels == 0) return BHG_OK;
    if (!d_end_dir || !d_flags || !d_sky || (!d_rgba && !d_rgba_f32)) return fail(BHG_E_INVALID, "NULL device pointer");
    ENTER_DEVICE(c->device);
    bhg::ShadeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.dir = d_end_dir;
    a.flags = d_flags;
    a.sky = d_sky;
    a.rgba = d_rgba;
    a.rgba_f32 = d_rgba_f32;
    a.scatter = d_scatter;
    a.n_pixels = n_pixels;
    a.samples = samples;
    a.sky_w = sky_w;
    a.sky_h = sky_h;
    HIP_TRY(bhg::launch_shade(a, (hipStream_t)stream));
    return BHG_OK;
}

}  // extern "C"

namespace {
int shade_scene_impl(bhg_context *c, const double *d_end, const uint8_t *d_flags, const int8_t *d_object_id,
                     size_t n_pixels, int32_t samples, const bhg_scene *sc, double *d_rgba, float *d_rgba_f32,
                     const int64_t *d_scatter, void *stream);
}

extern "C" {

int bhg_shade_scene_device(bhg_context *c, const double *d_end, const uint8_t *d_flags, const int8_t *d_object_id,
                           size_t n_pixels, int32_t samples, const bhg_scene *sc, double *d_rgba, void *stream)
{
    if (n_pixels && !d_rgba) return fail(BHG_E_INVALID, "NULL device pointer");
    return shade_scene_impl(c, d_end, d_flags, d_object_id, n_pixels, samples, sc, d_rgba, nullptr, nullptr, stream);
}

int bhg_shade_scene_f32_device(bhg_context *c, const double *d_end, const uint8_t *d_flags, const int8_t *d_object_id,
                               size_t n_pixels, int32_t samples, const bhg_scene *sc, float *d_rgba_f32,
                               const int64_t *d_scatter, void *stream)
{
    if (n_pixels && !d_rgba_f32) return fail(BHG_E_INVALID, "NULL device pointer");
    return shade_scene_impl(c, d_end, d_flags, d_object_id, n_pixels, samples, sc, nullptr, d_rgba_f32, d_scatter, stream);
}

}  // extern "C"

namespace {

int shade_scene_impl(bhg_context *c, const double *d_end, const uint8_t *d_flags, const int8_t *d_object_id,
                     size_t n_pixels, int32_t samples, const bhg_scene *sc, double *d_rgba, float *d_rgba_f32,
                     const int64_t *d_scatter, void *stream)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    if (!sc) return fail(BHG_E_INVALID, "scene is NULL");
    if (samples <= 0 || sc->sky_w <= 0 || sc->sky_h <= 0) return fail(BHG_E_INVALID, "samples, sky_w, sky_h must be > 0");
    if (sc->n_spheres < 0 || sc->n_spheres > BHG_MAX_SPHERES || sc->n_lamps < 0 || sc->n_lamps > 4)
        return fail(BHG_E_INVALID, "n_spheres must be in [0, BHG_MAX_SPHERES], n_lamps in [0, 4]");
    // (an EMPTY shard -- a rank without pixels: fewer tiles than ranks -- has no rays and no arrays: nothing to check them against)
    if (n_pixels > 0 && sc->n_spheres > 0 && !d_object_id) return fail(BHG_E_INVALID, "object_id is NULL but the scene has spheres");
    if (sc->disk_r_out > 0.0) {
        if (!(sc->disk_r_out > sc->disk_r_in) || !(sc->disk_stddev > 0.0))
            return fail(BHG_E_INVALID, "disk needs r_out > r_in and stddev > 0");
        if (sc->d_disk_tex && (sc->disk_w <= 0 || sc->disk_h <= 0)) return fail(BHG_E_INVALID, "disk texture size must be > 0");
    }
    for (int j = 0; j < sc->n_spheres; j++)
        if (!(sc->spheres[j][3] > 0.0)) return fail(BHG_E_INVALID, "sphere radii must be > 0");
    if (n_pixels == 0) return BHG_OK;
    if (!d_end || !d_flags || !sc->d_sky) return fail(BHG_E_INVALID, "NULL device pointer");
    ENTER_DEVICE(c->device);
    bhg::ShadeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.end = d_end;
    a.flags = d_flags;
    a.sky = sc->d_sky;
    a.rgba = d_rgba;
    a.rgba_f32 = d_rgba_f32;
    a.scatter = d_scatter;
    a.n_pixels = n_pixels;
    a.samples = samples;
    a.sky_w = sc->sky_w;
    a.sky_h = sc->sky_h;
    a.object_id = sc->n_spheres > 0 ? d_object_id : nullptr;
    a.disk_tex = sc->d_disk_tex;
    a.disk_w = sc->disk_w;
    a.disk_h = sc->disk_h;
    a.disk_r_in = sc->disk_r_in;
    a.disk_r_out = sc->disk_r_out;
    a.disk_phase = sc->disk_phase;
    a.disk_mean = sc->disk_mean;
    a.disk_stddev = sc->disk_stddev;
    a.disk_intensity = sc->disk_intensity;
    a.n_spheres = sc->n_spheres;
    a.n_lamps = sc->n_lamps;
    std::memcpy(a.spheres, sc->spheres, sizeof(a.spheres));
    std::memcpy(a.sphere_rgb, sc->sphere_rgb, sizeof(a.sphere_rgb));
    std::memcpy(a.lamps, sc->lamps, sizeof(a.lamps));
    HIP_TRY(bhg::launch_shade(a, (hipStream_t)stream));
    return BHG_OK;
}

}  // namespace

extern "C" {

int bhg_assemble_frame_f32_device(bhg_context *c, const float *d_slabs, const int64_t *d_index, size_t n_pixels,
                                  float *d_frame, void *stream)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    if (n_pixels == 0) return BHG_OK;
    if (!d_slabs || !d_index || !d_frame) return fail(BHG_E_INVALID, "NULL device pointer");
    ENTER_DEVICE(c->device);
    HIP_TRY(bhg::launch_gather_rows4(d_slabs, d_index, n_pixels, d_frame, (hipStream_t)stream));
    return BHG_OK;
}

int bhg_trajectory(bhg_context *c, const bhg_params *p, const double *x0, int x0_is_shared, const double *k0, size_t n,
                   uint32_t n_points, double *traj, uint32_t *n_valid, double *end, uint8_t *flags)
{
    return bhg_trajectory_objects(c, p, nullptr, 0, x0, x0_is_shared, k0, n, n_points, traj, n_valid, end, flags, nullptr);
}

int bhg_trajectory_objects(bhg_context *c, const bhg_params *p, const double *spheres, int32_t n_spheres, const double *x0,
                           int x0_is_shared, const double *k0, size_t n, uint32_t n_points, double *traj, uint32_t *n_valid,
                           double *end, uint8_t *flags, int8_t *object_id)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    int rc = validate(p);
    if (rc != BHG_OK) return rc;
    rc = validate_spheres(p, spheres, n_spheres);
    if (rc != BHG_OK) return rc;
    if (n_points < 2) return fail(BHG_E_INVALID, "n_points must be >= 2");
    if (n == 0) return BHG_OK;
    if (!x0 || !k0 || !traj || !n_valid) return fail(BHG_E_INVALID, "x0 / k0 / traj / n_valid is NULL");
    // (the kernels form a ray's byte offsets in 32 bits -- store_result: idx * 48 -- and this call is ONE launch)
    if (n > bhg::BHG_MAX_RAYS_PER_LAUNCH) return fail(BHG_E_INVALID, "bhg_trajectory takes at most 2^26 rays per call");
    ENTER_DEVICE(c->device);
    const size_t in_bytes = n * 3 * sizeof(double) * (x0_is_shared ? 1 : 2);
    const size_t sz_traj = n * 6 * (size_t)n_points * sizeof(double);
    const size_t off_end = sz_traj, off_nv = off_end + n * 6 * sizeof(double), off_steps = off_nv + n * sizeof(uint32_t);
    const size_t off_acc = off_steps + n * sizeof(uint32_t), off_flags = off_acc + n * sizeof(uint32_t);
    const size_t off_obj = off_flags + n;      // [n] int8: the sphere a ray ends on (-1: none); only with spheres
    const bool with_obj = n_spheres > 0;
    rc = ensure(&c->d_in, &c->d_in_bytes, in_bytes);
    if (rc != BHG_OK) return rc;
    rc = ensure(&c->d_out, &c->d_out_bytes, off_obj + n + 64);
    if (rc != BHG_OK) return rc;
    // up to 2048 rays the kernel runs one wave per ray, prepares the ray itself and fills what it never reaches with NaN:
    // no prepare records, no memset; the direction of a ONE-ray call (the engine's literal call) rides in the kernel arguments
    const bool wave = bhg::trajectory_wave_per_ray(n);
    const bool one = n == 1 && x0_is_shared;
    if (!wave) {
        rc = ensure(&c->d_ws, &c->d_ws_bytes, n * 8 * sizeof(double) + 64);
        if (rc != BHG_OK) return rc;
    }
    double *d_k0 = one ? nullptr : (double *)c->d_in, *d_x0 = x0_is_shared ? nullptr : (double *)c->d_in + n * 3;
    char *o = (char *)c->d_out;
    hipStream_t s = c->stream;
    // Zero-copy samples: a small call (the engine's literal one: one ray, 10,000 samples = 480 kB) whose `traj` is
    // page-locked memory (bhg_host_alloc) has the wave-per-ray kernel write its samples STRAIGHT into the caller's array
    // over PCIe -- no device-to-host copy of the block, no host-side split: 40 of the call's 98 us.  (The kernel's stores
    // are 512-byte runs per sample row; the array is complete when the stream has been waited for.)
    double *d_traj = (double *)o;
    bool direct = false;
    if (wave && sz_traj <= (size_t(4) << 20)) {
        void *dp = nullptr;
        if (is_pinned_range(traj, sz_traj, &dp)) {      // (the WHOLE block, not its first byte: the kernel writes all of it)
            d_traj = (double *)dp;
            direct = true;
        }
    }
    // ... and then the small arrays behind the sample block -- end state, n_valid, counts, flags: at most 2048 rays --
    // are written into the context's page-locked block the same way: the call is a launch and a stream wait, no copy
    char *os = o;     // where the kernel writes those (device address), laid out like the device block from off_end on
    if (direct) {
        rc = ensure_pinned(&c->pin_out, &c->pin_out_bytes, size_t(1) << 20);
        if (rc != BHG_OK) return rc;
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->pin_out, 0) == hipSuccess && dp) {
            os = (char *)dp - off_end;
        } else {
            (void)hipGetLastError();
            direct = false;
            d_traj = (double *)o;
        }
    }
    if (d_k0) HIP_TRY(hipMemcpyAsync(d_k0, k0, n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    if (d_x0) HIP_TRY(hipMemcpyAsync(d_x0, x0, n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    if (!wave) HIP_TRY(hipMemsetAsync(o, 0xFF, sz_traj, s));  // samples a ray never reaches read back as NaN

    bhg::TraceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.k0 = d_k0;
    a.x0 = d_x0;
    a.end = (double *)(os + off_end);
    a.ws = wave ? nullptr : (double *)c->d_ws;
    if (one) {
        a.k0s[0] = k0[0];
        a.k0s[1] = k0[1];
        a.k0s[2] = k0[2];
    }
    a.flags = (uint8_t *)(os + off_flags);
    a.n_steps = (uint32_t *)(os + off_steps);
    a.n_accepted = (uint32_t *)(os + off_acc);
    a.counter = c->counter;        // (the trajectory kernel hands out no batches)
    a.counter_next = nullptr;
    a.n = n;
    if (!d_x0) {
        a.x0s[0] = x0[0];
        a.x0s[1] = x0[1];
        a.x0s[2] = x0[2];
    }
    a.r_s = p->r_s;
    a.lambda_end = p->lambda_end;
    a.max_step = p->max_step;
    a.rtol = scipy_rtol(p->rtol);
    a.atol = p->atol;
    a.h_fixed = p->h_fixed;
    a.r_exit = p->r_exit;
    a.disk_r_in = p->disk_r_in;
    a.disk_r_out = p->disk_r_out;
    a.spin = p->spin;
    a.mu2 = p->time_like ? 1.0 : 0.0;
    a.r_hor = p->r_s;
    a.ws_stride = 6;
    if (p->rhs_form == BHG_RHS_KERR_BL) {
        const double M = 0.5 * p->r_s;
        a.r_hor = (M + std::sqrt(M * M - p->spin * p->spin)) * (1.0 + BHG_KERR_HORIZON_MARGIN);
        a.from_records = 1;                    // (the trajectory kernel reads the prepare pass's records)
        a.ws_stride = 8;
    }
    a.max_steps = p->max_steps ? p->max_steps : (1u << 20);
    a.min_step_cap = 0.0;
    a.n_spheres = n_spheres;
    for (int j = 0; j < n_spheres; j++)
        for (int q = 0; q < 4; q++) a.spheres[j][q] = spheres[4 * j + q];
    a.object_id = with_obj ? (int8_t *)(os + off_obj) : nullptr;
    HIP_TRY(bhg::launch_trajectory(a, (p->time_like && p->rhs_form == BHG_RHS_CHRISTOFFEL) ? bhg::BHG_RHS_CHRISTOFFEL_TL_ : p->rhs_form, p->method,
                                   d_traj, (uint32_t *)(os + off_nv), n_points, s));
    const size_t total = with_obj ? off_obj + n : off_flags + n;
    if (object_id && !with_obj) std::memset(object_id, 0xFF, n);      // (no spheres: no ray ends on one)
    if (direct) {
        const char *h = (const char *)c->pin_out - off_end;
        // (polling hipStreamQuery before the blocking wait was measured: no gain, the runtime already spins)
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(n_valid, h + off_nv, n * sizeof(uint32_t));
        if (end) std::memcpy(end, h + off_end, n * 6 * sizeof(double));
        if (flags) std::memcpy(flags, h + off_flags, n);
        if (object_id && with_obj) std::memcpy(object_id, h + off_obj, n);
        return BHG_OK;
    }
    if (total <= (size_t(4) << 20)) {
        // the engine's per-ray call (one ray, 10,000 samples: 480 kB): ONE copy of the whole output block into page-locked
        // memory and a host-side split, instead of four copies into the caller's pageable arrays (each of which the
        // runtime stages and waits for on its own)
        rc = ensure_pinned(&c->pin_out, &c->pin_out_bytes, total < (size_t(1) << 20) ? (size_t(1) << 20) : (size_t(4) << 20));
        if (rc != BHG_OK) return rc;
        const char *h = (const char *)c->pin_out;
        HIP_TRY(hipMemcpyAsync(c->pin_out, o, total, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(traj, h, sz_traj);
        std::memcpy(n_valid, h + off_nv, n * sizeof(uint32_t));
        if (end) std::memcpy(end, h + off_end, n * 6 * sizeof(double));
        if (flags) std::memcpy(flags, h + off_flags, n);
        if (object_id && with_obj) std::memcpy(object_id, h + off_obj, n);
        return BHG_OK;
    }
    HIP_TRY(hipMemcpyAsync(traj, o, sz_traj, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(n_valid, o + off_nv, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (end) HIP_TRY(hipMemcpyAsync(end, o + off_end, n * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (flags) HIP_TRY(hipMemcpyAsync(flags, o + off_flags, n, hipMemcpyDeviceToHost, s));
    if (object_id && with_obj) HIP_TRY(hipMemcpyAsync(object_id, o + off_obj, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BHG_OK;
}

int bhg_acceleration(bhg_context *c, const bhg_params *p, const double *x, const double *k, size_t n, double *acc)
{
    if (!c) return fail(BHG_E_INVALID, "ctx is NULL");
    int rc = validate(p);
    if (rc != BHG_OK) return rc;
    if (n == 0) return BHG_OK;
    if (!x || !k || !acc) return fail(BHG_E_INVALID, "x / k / acc is NULL");
    ENTER_DEVICE(c->device);
    rc = ensure(&c->d_in, &c->d_in_bytes, n * 6 * sizeof(double));
    if (rc != BHG_OK) return rc;
    rc = ensure(&c->d_out, &c->d_out_bytes, n * 3 * sizeof(double));
    if (rc != BHG_OK) return rc;
    double *dx = (double *)c->d_in, *dk = dx + 3 * n;
    HIP_TRY(hipMemcpyAsync(dx, x, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dk, k, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(bhg::launch_accel(dx, dk, p->r_s, p->spin, p->time_like ? 1.0 : 0.0, n, (double *)c->d_out,
                              (p->time_like && p->rhs_form == BHG_RHS_CHRISTOFFEL) ? bhg::BHG_RHS_CHRISTOFFEL_TL_ : p->rhs_form, c->stream));
    HIP_TRY(hipMemcpyAsync(acc, c->d_out, n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BHG_OK;
}

int bhg_peak_probe(bhg_context *c, int32_t kind, double target_ms, double out[6])
{
    if (!c || !out) return fail(BHG_E_INVALID, "bad argument");
    if (kind != BHG_PROBE_FMA && kind != BHG_PROBE_STEP_MIX) return fail(BHG_E_INVALID, "unknown probe kind");
    if (!(target_ms >= 0.0) || target_ms > 100.0) return fail(BHG_E_INVALID, "target_ms must be in [0, 100] (0 = 1 ms)");
    if (target_ms == 0.0) target_ms = 1.0;
    ENTER_DEVICE(c->device);
    const int per_cu = 12;                       // the Schwarzschild trace kernels' residency (3 waves per SIMD)
    const int grid = per_cu * c->num_cus;
    int rc = ensure(&c->d_out, &c->d_out_bytes, (size_t)grid * 64 * sizeof(double));
    if (rc != BHG_OK) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t he = hipEventCreate(&e1);
    if (he != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail_hip(he, "hipEventCreate");
    }
    auto timed = [&](uint32_t iters, float *ms) -> hipError_t {
        hipError_t e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) e = bhg::launch_probe(kind, grid, iters, (double *)c->d_out, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        return e;
    };
    // size the loop for the asked duration from a short launch, then take the median of five
    uint32_t iters = 64;
    float ms = 0.0f;
    he = timed(iters, &ms);
    if (he == hipSuccess) he = timed(iters, &ms);
    float runs[5] = {0, 0, 0, 0, 0};
    if (he == hipSuccess) {
        const double scale = target_ms / std::fmax((double)ms, 1e-3);
        iters = (uint32_t)std::fmin(std::fmax(64.0 * scale, 16.0), 4.0e6);
        he = timed(iters, &ms);   // (one untimed launch at the final size)
        for (int i = 0; i < 5 && he == hipSuccess; i++) he = timed(iters, &runs[i]);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail_hip(he, "bhg_peak_probe");
    std::sort(runs, runs + 5);
    uint32_t valu = 0, quarter = 0;
    bhg::probe_shape(kind, &valu, &quarter);
    const double waves = (double)grid;
    const double wave_insts = waves * (double)iters * (double)valu;
    // flops under SURVEY section 8d's counting rule: an FMA is 2, a reciprocal / reciprocal square root 1
    const double flops = waves * 64.0 * (double)iters * (2.0 * (double)(valu - quarter) + (double)quarter);
    const double sec = (double)runs[2] * 1e-3;
    out[0] = flops / sec * 1e-12;
    out[1] = (double)runs[2];
    out[2] = wave_insts;
    out[3] = waves * (double)iters * (double)quarter;
    out[4] = (double)runs[0];
    // the clock a full-rate fp64 pipe (128 flop per clock and CU) would need for that figure: for the pure-FMA probe the
    // sustained shader clock itself
    out[5] = out[0] * 1e12 / (128.0 * (double)c->num_cus) * 1e-6;
    return BHG_OK;
}

}  // extern "C"
