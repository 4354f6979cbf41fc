/* examples/trace_frame.c -- the C ABI used from plain C, no Python: trace the rays of a small pinhole frame
 * around a Schwarzschild hole with one lit sphere in front of it, and print what the rays ended on.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/trace_frame.c -Lblackhole_geodesic_calculator_amd -lbhgeo \
 *       -Wl,-rpath,$PWD/blackhole_geodesic_calculator_amd -lm -o build/trace_frame && build/trace_frame
 *
 * This is the call sequence a compiled host (or another language's FFI) would make in place of the
 * reference's per-ray `calc_trajectory` loop (raytracer/RelativisticRenderEngine.py:293-308).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "bhgeo.h"

int main(void)
{
    const int W = 96, H = 64;
    /* the binder's handshake (include/bhgeo.h): header and library must agree on the ABI and on sizeof(bhg_params) */
    if (bhg_abi_check(BHG_ABI_VERSION, sizeof(bhg_params), 0, 0, 0) != BHG_OK) {
        fprintf(stderr, "%s\n", bhg_last_error());
        return 9;
    }
    const size_t n = (size_t)W * H;
    const double cam[3] = {1e-4, 0.0, 30.0}; /* BH-centred camera position, looking down -z */
    double *k0 = malloc(n * 3 * sizeof(double)), *end = malloc(n * 6 * sizeof(double));
    uint8_t *flags = malloc(n);
    uint32_t *steps = malloc(n * sizeof(uint32_t));
    int8_t *obj = malloc(n);
    if (!k0 || !end || !flags || !steps || !obj) return 2;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) { /* pinhole directions, field of view 0.6 (no jitter) */
            double d[3] = {0.6 * (x - W / 2) / W, 0.6 * (y - H / 2) / W, -1.0};
            double inv = 1.0 / sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int c = 0; c < 3; c++) k0[((size_t)y * W + x) * 3 + c] = d[c] * inv;
        }

    if (bhg_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "libbhgeo has no CPU fallback");
        return 3;
    }
    bhg_context *ctx = NULL;
    if (bhg_create(0, &ctx) != BHG_OK) {
        fprintf(stderr, "bhg_create: %s\n", bhg_last_error());
        return 3;
    }
    bhg_params p;
    bhg_default_params(&p);
    p.r_s = 1.0;         /* 2 * mass, mass = 0.5 */
    p.lambda_end = 60.0; /* curve_end */
    p.r_exit = 40.0;     /* leave the curved region there */
    const double sphere[4] = {2.5, 1.0, 10.0, 1.5};
    int rc = bhg_trace_objects(ctx, &p, sphere, 1, cam, 1, k0, n, end, flags, steps, NULL, obj);
    if (rc != BHG_OK) {
        fprintf(stderr, "bhg_trace_objects: %s\n", bhg_last_error());
        return 4;
    }
    size_t horizon = 0, escaped = 0, object = 0, other = 0;
    unsigned long long total_steps = 0;
    for (size_t i = 0; i < n; i++) {
        total_steps += steps[i];
        if ((flags[i] & BHG_FLAG_HIT_OBJECT) == BHG_FLAG_HIT_OBJECT) object++;
        else if (flags[i] & BHG_FLAG_HIT_HORIZON) horizon++;
        else if (flags[i] & BHG_FLAG_EXITED_SPHERE) escaped++;
        else other++;
    }
    char name[128];
    bhg_device_name(ctx, name, sizeof name);
    printf("%s: %zu rays, %zu on the horizon, %zu on the sphere, %zu left the region, %zu other; %.1f steps per ray\n",
           name, n, horizon, object, escaped, other, (double)total_steps / (double)n);
    for (int y = 0; y < H; y += 4) { /* a character picture: '#' hole, 'o' sphere, '.' sky */
        for (int x = 0; x < W; x += 2) {
            uint8_t f = flags[(size_t)y * W + x];
            putchar((f & BHG_FLAG_HIT_OBJECT) == BHG_FLAG_HIT_OBJECT ? 'o' : (f & BHG_FLAG_HIT_HORIZON) ? '#' : '.');
        }
        putchar('\n');
    }
    bhg_destroy(ctx);
    free(k0); free(end); free(flags); free(steps); free(obj);
    return 0;
}
