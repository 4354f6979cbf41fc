// frame_kernels.hip -- the data-parallel stages either side of the geodesic solve, on device:
//   * camera-ray generation with the reference's multisample jitter
//     (raytracer/RelativisticRenderEngine.py:185-188, :224-230), from a device-resident MT19937
//     jitter stream (produced once on the host from Python's own seeded state, bit-identical);
//   * escaping-ray shading against an equirectangular sky (background_hit, :366-378) and the
//     per-pixel multisample mean (sbuf += colour; buf = sbuf/(s+1), :242-250); rays that ended on the
//     thin disk get the Limited engine's disk colour (LimitedRelativisticRenderEngine.py:427-436, :300),
//     rays that ended on an object sphere the Lambert lamp sum of spacetime_hit (:356-363).
// Ray generation is an HBM-bound element-wise kernel.  The shade kernel runs one thread per RAY and stages the S samples of
// a pixel in LDS, where one thread sums them in sample order (the reference's order: the result is deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geodesic_kernels.h"
#include "device_math.h"

namespace bhg {

// d = (x_r + dx (u1 - 1/2), y_r + dy (u2 - 1/2), -1), rotated, normalised.  Operation order follows
// the host restatement (raygen.py) so that an unrotated camera gives bit-identical directions
// (the library is built with -ffp-contract=off).
__global__ void __launch_bounds__(256) raygen_kernel(const RaygenArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t P = A.n_pixels;
    if (i >= P * (uint64_t)A.samples) return;
    const uint64_t s = i / P, p = i - s * P;
    const int64_t pix = A.pixels ? A.pixels[p] : (int64_t)p;
    const int64_t py = pix / A.width, px = pix - py * A.width;
    const double W = (double)A.width, H = (double)A.height;
    const double aspect = H / W;
    const double dy = aspect / H, dx = 1.0 / W;
    const uint64_t j = A.compact ? i * 2 : (s * (uint64_t)A.width * (uint64_t)A.height + (uint64_t)pix) * 2;
    const double u1 = A.jitter ? A.jitter[j] : 0.5, u2 = A.jitter ? A.jitter[j + 1] : 0.5;
    const double x_render = A.fov_x * (double)(px - (int64_t)(A.width / 2)) / W;
    const double y_render = A.fov_y * (double)(py - (int64_t)(A.height / 2)) / H * aspect;
    double d0 = x_render + dx * (u1 - 0.5);
    double d1 = y_render + dy * (u2 - 0.5);
    double d2 = -1.0;
    if (A.rotate) {
        const double e0 = A.rot[0] * d0 + A.rot[1] * d1 + A.rot[2] * d2;
        const double e1 = A.rot[3] * d0 + A.rot[4] * d1 + A.rot[5] * d2;
        const double e2 = A.rot[6] * d0 + A.rot[7] * d1 + A.rot[8] * d2;
        d0 = e0;
        d1 = e1;
        d2 = e2;
    }
    const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    double *o = A.k0 + i * 3;
    o[0] = d0 / nrm;
    o[1] = d1 / nrm;
    o[2] = d2 / nrm;
}

// Bilinear lookup in an equirectangular RGBA float32 image.  Texture coordinates (u, v) in
// [-1, 1]^2 as background_hit passes them to Texture.evaluate (:375): u wraps, v clamps.
// (Blender's own texture filter cannot be reproduced outside Blender; this definition is the
// build's own and is restated in numpy in device_frame.py for the parity test.)
__device__ __forceinline__ void sky_lookup(const float *sky, int TW, int TH, double u, double v, double rgb[3])
{
    const double TWd = (double)TW;
    double fx = (u + 1.0) * 0.5 * TWd - 0.5;
    double fy = (v + 1.0) * 0.5 * (double)TH - 0.5;
    double x0f = floor(fx), y0f = floor(fy);
    double ax = fx - x0f, ay = fy - y0f;
    // column x0f mod TW (u wraps) in floating point -- the columns are small integers held exactly, the quotient's
    // rounding can only be off by one period, which the two selects put right; a 64-bit integer remainder is a
    // hundred-instruction sequence on this machine
    double xw = __builtin_fma(-TWd, floor(x0f * (1.0 / TWd)), x0f);
    xw = xw < 0.0 ? xw + TWd : (xw >= TWd ? xw - TWd : xw);
    const int x0 = (int)xw;
    const int x1 = (x0 + 1 == TW) ? 0 : x0 + 1;
    const double THm = (double)(TH - 1);
    const int y0 = (int)fmin(fmax(y0f, 0.0), THm), y1 = (int)fmin(fmax(y0f + 1.0, 0.0), THm);
    const float4 t00 = reinterpret_cast<const float4 *>(sky)[(long)y0 * TW + x0];
    const float4 t01 = reinterpret_cast<const float4 *>(sky)[(long)y0 * TW + x1];
    const float4 t10 = reinterpret_cast<const float4 *>(sky)[(long)y1 * TW + x0];
    const float4 t11 = reinterpret_cast<const float4 *>(sky)[(long)y1 * TW + x1];
    const double w00 = (1.0 - ax) * (1.0 - ay), w01 = ax * (1.0 - ay), w10 = (1.0 - ax) * ay, w11 = ax * ay;
    rgb[0] = __builtin_fma(w11, (double)t11.x, __builtin_fma(w10, (double)t10.x, __builtin_fma(w01, (double)t01.x, w00 * (double)t00.x)));
    rgb[1] = __builtin_fma(w11, (double)t11.y, __builtin_fma(w10, (double)t10.y, __builtin_fma(w01, (double)t01.y, w00 * (double)t00.y)));
    rgb[2] = __builtin_fma(w11, (double)t11.z, __builtin_fma(w10, (double)t10.z, __builtin_fma(w01, (double)t01.z, w00 * (double)t00.z)));
}

// Disk colour (LimitedRelativisticRenderEngine.py:427-436, :300): texture(texture_x, scale) * intensity with a
// Gaussian radial profile.  (y_disk / |y_disk| is NaN at y = 0 there; here the sign of +0 is +.)
__device__ __forceinline__ void disk_colour(const ShadeArgs &A, const double *e, double rgb[3])
{
    const double x = e[0], y = e[1];
    const double R = sqrt(x * x + y * y);
    const double scale = (R - A.disk_r_in) / (A.disk_r_out - A.disk_r_in);
    const double dm = scale - A.disk_mean;
    const double intensity = A.disk_intensity * exp(-(dm * dm) / (2.0 * A.disk_stddev * A.disk_stddev)) /
                             sqrt(2.0 * M_PI * A.disk_stddev);
    double cx = x / R;
    cx = cx > 1.0 ? 1.0 : (cx < -1.0 ? -1.0 : cx);
    const double texture_x = (A.disk_phase + acos(cx) * (y < 0.0 ? -1.0 : 1.0)) / M_PI;
    if (A.disk_tex)
        sky_lookup(A.disk_tex, A.disk_w, A.disk_h, texture_x, scale, rgb);
    else
        rgb[0] = rgb[1] = rgb[2] = 1.0;
    rgb[0] *= intensity;
    rgb[1] *= intensity;
    rgb[2] *= intensity;
}

// Object colour: pure Lambert sum over point lamps with 1/d^2 falloff, light paths straight (flat space) as in
// spacetime_hit (RelativisticRenderEngine.py:341-363: base_color = intensity, colour += base_color * intensity *
// n.l / d^2 unless a shadow ray from loc + eps * l hits something -- here: one of the other spheres).  n.l is
// clamped at 0 (the reference adds negative light on the far side; "This needs some serius work", :320).
__device__ __forceinline__ void object_colour(const ShadeArgs &A, const double *e, int j, double rgb[3])
{
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    if (j < 0 || j >= A.n_spheres) return;
    const double *sp = A.spheres[j];
    const double inv_rho = 1.0 / sp[3];
    const double n[3] = {(e[0] - sp[0]) * inv_rho, (e[1] - sp[1]) * inv_rho, (e[2] - sp[2]) * inv_rho};
    double sum = 0.0;
    for (int l = 0; l < A.n_lamps; l++) {
        const double lv[3] = {A.lamps[l][0] - e[0], A.lamps[l][1] - e[1], A.lamps[l][2] - e[2]};
        const double d2 = lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2];
        const double dist = sqrt(d2);
        const double ld[3] = {lv[0] / dist, lv[1] / dist, lv[2] / dist};
        const double ndl = n[0] * ld[0] + n[1] * ld[1] + n[2] * ld[2];
        if (!(ndl > 0.0)) continue;
        bool shadow = false;
        for (int q = 0; q < A.n_spheres; q++) {
            if (q == j) continue;
            const double *sq = A.spheres[q];
            const double oc[3] = {e[0] - sq[0], e[1] - sq[1], e[2] - sq[2]};
            const double b = oc[0] * ld[0] + oc[1] * ld[1] + oc[2] * ld[2];
            const double cq = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - sq[3] * sq[3];
            const double disc = b * b - cq;
            if (disc > 0.0) {
                const double sd = sqrt(disc);
                const double t0 = -b - sd, t1 = -b + sd;
                if ((t0 > 1e-5 && t0 < dist) || (t0 <= 1e-5 && t1 > 1e-5)) shadow = true;  // enters on the way, or starts inside
            }
        }
        if (!shadow) sum += A.lamps[l][3] * A.lamps[l][3] * ndl / d2;
    }
    rgb[0] = A.sphere_rgb[j][0] * sum;
    rgb[1] = A.sphere_rgb[j][1] * sum;
    rgb[2] = A.sphere_rgb[j][2] * sum;
}

// The colour of ONE ray (sample s of pixel p): black for a horizon ray (:242-244), the disk's / an object's colour, or the
// sky in its exit direction.
__device__ __forceinline__ void ray_colour(const ShadeArgs &A, uint64_t i, uint8_t fl, double c0, double c1, double c2, double rgb[3])
{
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    if (fl & BHG_FLAG_HIT_HORIZON_) return;
    const double *e = A.end + i * 6;
    if (fl == BHG_FLAG_HIT_DISK_ && A.disk_r_out > 0.0 && A.end) {
        disk_colour(A, e, rgb);
        return;
    }
    if (fl == BHG_FLAG_HIT_OBJECT_ && A.object_id && A.end) {
        object_colour(A, e, (int)A.object_id[i], rgb);
        return;
    }
    // theta = 1 - acos(d_z / |d|) / pi (:373), phi = atan2(d_y, d_x) / pi (:374); exit directions are not unit
    // vectors (the Cam edition normalises, CamEdition.py:433-437) -- both angles as scale-free atan2's:
    // acos(d_z / |d|) = atan2(sqrt(d_x^2 + d_y^2), d_z)
    const double rho = sqrt(c0 * c0 + c1 * c1);
    const double theta = 1.0 - atan2_fast(rho, c2) * 0.3183098861837907;
    const double phi = atan2_fast(c1, c0) * 0.3183098861837907;
    sky_lookup(A.sky, A.sky_w, A.sky_h, -phi, 2.0 * theta - 1.0, rgb);  // :375
}

__device__ __forceinline__ void write_pixel(const ShadeArgs &A, uint64_t p, const double acc[3])
{
    const double inv_s = 1.0 / (double)A.samples;  // buf = sbuf / (s+1) after the last sample (:250)
    if (A.rgba) {
        double *o = A.rgba + p * 4;
        reinterpret_cast<double2 *>(o)[0] = make_double2(acc[0] * inv_s, acc[1] * inv_s);
        reinterpret_cast<double2 *>(o)[1] = make_double2(acc[2] * inv_s, 1.0);
    }
    if (A.rgba_f32) {
        // what Blender's layer.rect holds (:163-164): float RGBA, alpha 1; optionally scattered straight to the
        // pixel's place in the frame (scatter[p] = y*W + x of this rank's p-th pixel)
        const uint64_t q = A.scatter ? (uint64_t)A.scatter[p] : p;
        reinterpret_cast<float4 *>(A.rgba_f32)[q] =
            make_float4((float)(acc[0] * inv_s), (float)(acc[1] * inv_s), (float)(acc[2] * inv_s), 1.0f);
    }
}

// One thread per RAY, the S samples of a pixel staged in LDS and summed by one thread in sample order (:242-250: sbuf +=
// colour, sample after sample -- the order is part of the result).  A workgroup of 256 threads takes PPB = 256 / S
// pixels; thread t = s * PPB + q is sample s of the block's q-th pixel, so that for a fixed s the block reads PPB
// consecutive rays of the [S][P] layout (coalesced).  Against one thread per pixel walking its samples one after the
// other (round 3; kept below for S > 256) this puts S times as many independent atan2 / texel-gather chains in flight:
// the kernel is a latency chain per ray, not a bandwidth problem (131 MB in, 16 MB out per config-2 frame).
__global__ void __launch_bounds__(256) shade_reduce_kernel(const ShadeArgs A, const uint32_t ppb)
{
    __shared__ double col[256 * 3];
    const uint32_t t = threadIdx.x, S = (uint32_t)A.samples;
    const uint32_t s = t / ppb, q = t - s * ppb;
    const uint64_t p = (uint64_t)blockIdx.x * ppb + q;
    const bool live = s < S && p < A.n_pixels;
    if (live) {
        const uint64_t i = (uint64_t)s * A.n_pixels + p;
        // exit directions: the second half of the end records, or (direction-only traces of sky frames) an array of their own
        const double *d = A.dir ? A.dir + i * 3 : A.end + i * 6 + 3;
        double rgb[3];
        ray_colour(A, i, A.flags[i], d[0], d[1], d[2], rgb);
        col[t * 3 + 0] = rgb[0];
        col[t * 3 + 1] = rgb[1];
        col[t * 3 + 2] = rgb[2];
    }
    __syncthreads();
    if (s == 0 && live) {
        // (a horizon sample contributes an exact 0: acc + 0.0 is acc, the sum is bit for bit the skipping loop's)
        double acc[3] = {0.0, 0.0, 0.0};
        for (uint32_t k = 0; k < S; k++) {
            const double *c = col + (size_t)(k * ppb + q) * 3;
            acc[0] += c[0];
            acc[1] += c[1];
            acc[2] += c[2];
        }
        write_pixel(A, p, acc);
    }
}

// More samples than a workgroup has threads: one thread per pixel, samples accumulated in registers in sample order.
__global__ void __launch_bounds__(256) shade_reduce_serial_kernel(const ShadeArgs A)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n_pixels) return;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int s = 0; s < A.samples; s++) {
        const uint64_t i = (uint64_t)s * A.n_pixels + p;
        const double *d = A.dir ? A.dir + i * 3 : A.end + i * 6 + 3;
        double rgb[3];
        ray_colour(A, i, A.flags[i], d[0], d[1], d[2], rgb);
        acc[0] += rgb[0];
        acc[1] += rgb[1];
        acc[2] += rgb[2];
    }
    write_pixel(A, p, acc);
}

// dst[i] = src[index[i]] for rows of four floats: puts the gathered per-rank slabs into frame order on the
// root GPU (index = the frame's permutation, computed once).  HBM-bound: 8 + 16 + 16 bytes per pixel.
__global__ void __launch_bounds__(256) gather_rows4_kernel(const float4 *src, const int64_t *index, uint64_t n, float4 *dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

// end[n][6] -> loc[n][3] and / or dir[n][3]: what spacetime_ray_cast hands back separately (end_loc, end_dir,
// RelativisticRenderEngine.py:307-308), so that only the arrays a caller reads cross PCIe
__global__ void __launch_bounds__(256) split_end_kernel(const double *end, uint64_t n, double *loc, double *dir)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const uint64_t r = i / 3, c = i - r * 3;
    if (loc) loc[i] = end[r * 6 + c];
    if (dir) dir[i] = end[r * 6 + 3 + c];
}

hipError_t launch_split_end(const double *end, uint64_t n, double *loc, double *dir, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    BHG_LAUNCH(split_end_kernel, dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0, s, end, n, loc, dir);
    return hipGetLastError();
}

hipError_t launch_gather_rows4(const float *src, const int64_t *index, uint64_t n, float *dst, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    BHG_LAUNCH(gather_rows4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(src), index, n, reinterpret_cast<float4 *>(dst));
    return hipGetLastError();
}

hipError_t launch_raygen(const RaygenArgs &a, hipStream_t s)
{
    const uint64_t n = a.n_pixels * (uint64_t)a.samples;
    if (n == 0) return hipSuccess;
    BHG_LAUNCH(raygen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_shade(const ShadeArgs &a, hipStream_t s)
{
    if (a.n_pixels == 0) return hipSuccess;
    if (a.samples > 256) {
        BHG_LAUNCH(shade_reduce_serial_kernel, dim3((unsigned)((a.n_pixels + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const uint32_t ppb = 256u / (uint32_t)a.samples;      // pixels per workgroup
    BHG_LAUNCH(shade_reduce_kernel, dim3((unsigned)((a.n_pixels + ppb - 1) / ppb)), dim3(256), 0, s, a, ppb);
    return hipGetLastError();
}

}  // namespace bhg
