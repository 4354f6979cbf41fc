/* examples/render_frame.c -- a whole frame from plain C: jitter stream -> rays -> geodesics -> shaded, sample-averaged
 * float RGBA pixels, on one or several GPUs of this one process, with the library-owned frame (bhg_frame_*).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/render_frame.c -Lblackhole_geodesic_calculator_amd -lbhgeo \
 *       -Wl,-rpath,$PWD/blackhole_geodesic_calculator_amd -lm -o build/render_frame && build/render_frame [devices]
 *
 * `devices` is a comma-separated list of device indices (default "0"; "0,0" shards the frame over two contexts of one
 * GPU, "0,1,2,3,4,5,6,7" over the eight GPUs of a node).  This is the call sequence a compiled host would make in place
 * of the reference's frame loop (raytracer/RelativisticRenderEngine.py:152-267: render_scene -> ray_trace).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bhgeo.h"

int main(int argc, char **argv)
{
    enum { W = 160, H = 96, S = 2, SKY_W = 64, SKY_H = 32 };
    /* the binder's handshake: this program was compiled against include/bhgeo.h of ABI BHG_ABI_VERSION; a library of
     * another version (other struct sizes) says so here instead of writing past a struct */
    if (bhg_abi_check(BHG_ABI_VERSION, sizeof(bhg_params), sizeof(bhg_camera), sizeof(bhg_scene), sizeof(bhg_frame_scene)) != BHG_OK) {
        fprintf(stderr, "%s\n", bhg_last_error());
        return 9;
    }
    int32_t devices[16];
    int n_dev = 0;
    char list[256];
    snprintf(list, sizeof list, "%s", argc > 1 ? argv[1] : "0");
    for (char *tok = strtok(list, ","); tok && n_dev < 16; tok = strtok(NULL, ",")) devices[n_dev++] = atoi(tok);
    if (bhg_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "libbhgeo has no CPU fallback");
        return 3;
    }

    /* camera of the reference's pre-traced view (RelativisticRenderEngineCamEdition.py:216-221), pixel centres */
    bhg_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.width = W, cam.height = H, cam.samples = S;
    cam.fov_x = cam.fov_y = 0.6;
    cam.rot[0] = cam.rot[4] = cam.rot[8] = 1.0;
    cam.origin[0] = 1e-4, cam.origin[2] = 30.0;
    /* a jitter stream of our own (any numbers in [0, 1) do; the engine's are Python's random.random()) */
    double *jitter = malloc(sizeof(double) * 2 * S * W * H);
    unsigned long long lcg = 42;
    for (size_t i = 0; i < (size_t)2 * S * W * H; i++) {
        lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
        jitter[i] = (double)(lcg >> 11) / 9007199254740992.0;
    }
    /* sky: a gradient with a bright band, float RGBA, equirectangular */
    float *sky = malloc(sizeof(float) * SKY_W * SKY_H * 4);
    for (int y = 0; y < SKY_H; y++)
        for (int x = 0; x < SKY_W; x++) {
            float *t = sky + ((size_t)y * SKY_W + x) * 4;
            t[0] = (float)x / SKY_W, t[1] = (float)y / SKY_H, t[2] = (y > 12 && y < 20) ? 1.0f : 0.2f, t[3] = 1.0f;
        }

    bhg_frame *fr = NULL;
    if (bhg_frame_create(devices, n_dev, &cam, jitter, 32, BHG_FRAME_GATHER_AUTO, &fr) != BHG_OK) {
        fprintf(stderr, "bhg_frame_create: %s\n", bhg_last_error());
        return 4;
    }
    bhg_frame_scene sc;
    memset(&sc, 0, sizeof sc);
    sc.sky = sky, sc.sky_w = SKY_W, sc.sky_h = SKY_H;
    sc.disk_mean = 0.2, sc.disk_stddev = 0.3, sc.disk_intensity = 1.0;
    sc.n_spheres = 1, sc.n_lamps = 1;
    sc.spheres[0][0] = 2.5, sc.spheres[0][1] = 1.0, sc.spheres[0][2] = 10.0, sc.spheres[0][3] = 1.5;
    sc.sphere_rgb[0][0] = 1.0, sc.sphere_rgb[0][1] = 0.8, sc.sphere_rgb[0][2] = 0.6;
    sc.lamps[0][0] = 10.0, sc.lamps[0][1] = 10.0, sc.lamps[0][2] = 30.0, sc.lamps[0][3] = 30.0;
    if (bhg_frame_set_scene(fr, &sc) != BHG_OK) {
        fprintf(stderr, "bhg_frame_set_scene: %s\n", bhg_last_error());
        return 4;
    }
    bhg_params p;
    bhg_default_params(&p);
    p.lambda_end = 60.0;
    p.r_exit = 40.0;
    float *rgba = malloc(sizeof(float) * W * H * 4);
    if (bhg_frame_render(fr, &p, rgba) != BHG_OK) {
        fprintf(stderr, "bhg_frame_render: %s\n", bhg_last_error());
        return 5;
    }
    uint64_t st[4];
    int64_t info[8];
    bhg_frame_stats(fr, st);
    bhg_frame_info(fr, info);
    double sum = 0.0;
    size_t black = 0;
    for (size_t i = 0; i < (size_t)W * H; i++) {
        sum += rgba[4 * i] + rgba[4 * i + 1] + rgba[4 * i + 2];
        black += (rgba[4 * i] == 0.0f && rgba[4 * i + 1] == 0.0f && rgba[4 * i + 2] == 0.0f);
    }
    printf("%d device(s), gather by %s: %llu rays, %.2f steps per ray, %llu on the horizon; %zu black pixels, checksum %.6f\n",
           (int)info[0], info[1] == BHG_FRAME_GATHER_RCCL ? "rccl" : "copies", (unsigned long long)st[0],
           (double)st[1] / (double)st[0], (unsigned long long)st[3], black, sum);
    for (int y = H - 1; y >= 0; y -= 4) { /* a character picture, top row first (the image's rows run bottom-up) */
        for (int x = 0; x < W; x += 2) {
            const float *q = rgba + ((size_t)y * W + x) * 4;
            const float v = q[0] + q[1] + q[2];
            putchar(v == 0.0f ? '#' : v > 1.6f ? 'o' : v > 0.9f ? '+' : '.');
        }
        putchar('\n');
    }
    bhg_frame_destroy(fr);
    free(jitter), free(sky), free(rgba);
    return 0;
}
