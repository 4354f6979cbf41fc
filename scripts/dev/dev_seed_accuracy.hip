// dev: relative error of the v_rcp_f64 / v_rsq_f64 seeds and of one quadratic / one cubic Newton step
// build + run on the box: hipcc --offload-arch=gfx950 -O2 scripts/dev/dev_seed_accuracy.hip -o /tmp/seed && /tmp/seed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, y, 1.0);
    double yq = __builtin_fma(y, e, y);
    double t = __builtin_fma(e, e, e);
    double yc = __builtin_fma(y, t, y);
    double r = __builtin_amdgcn_rsq(v);
    double r2 = r * r;
    double er = __builtin_fma(-v, r2, 1.0);
    double rq = __builtin_fma(r * er, 0.5, r);
    double p = __builtin_fma(0.375, er, 0.5);
    double rc = __builtin_fma(r * er, p, r);
    out[6 * i + 0] = y; out[6 * i + 1] = yq; out[6 * i + 2] = yc;
    out[6 * i + 3] = r; out[6 * i + 4] = rq; out[6 * i + 5] = rc;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o(6 * n);
    for (int i = 0; i < n; i++) h[i] = std::exp(-14.0 + 28.0 * ((i * 2654435761u) % 1000003) / 1000003.0);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        long double xi = h[i], rc = 1.0L / xi, rs = 1.0L / sqrtl(xi);
        for (int j = 0; j < 3; j++) m[j] = fmax(m[j], (double)fabsl((o[6 * i + j] - rc) / rc));
        for (int j = 3; j < 6; j++) m[j] = fmax(m[j], (double)fabsl((o[6 * i + j] - rs) / rs));
    }
    printf("rcp: seed %.3g  quadratic %.3g  cubic %.3g\nrsq: seed %.3g  quadratic %.3g  cubic %.3g\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
