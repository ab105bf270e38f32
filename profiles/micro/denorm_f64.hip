// Does fp64 arithmetic on gfx950 keep denormal results (as Java / the oracle do) in the forms the kernels use?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *in, double *out) {
    const double a = in[0], b = in[1];
    out[0] = a * b;                       // v_mul_f64
    out[1] = __builtin_fma(a, b, 0.0);    // v_fma_f64
    out[2] = a * b + in[2];               // mul + add (contraction off)
    out[3] = (a * b) / in[3];             // division of a denormal
    out[4] = sqrt(a * b);                 // sqrt of a denormal
    out[5] = __builtin_fma(a, b, in[2]);
    out[6] = in[2] - in[4];               // difference of denormals
    out[7] = 1.0 / in[5];                 // reciprocal of a tiny normal
}
int main() {
    double h[6] = {1e-158, 1.5e-158, 2.5e-316, 1024.0, 1.0e-316, 4e-158}, *d, *o, r[8];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const double a = h[0], b = h[1];
    const double e[8] = {a * b, __builtin_fma(a, b, 0.0), a * b + h[2], (a * b) / h[3], __builtin_sqrt(a * b), __builtin_fma(a, b, h[2]), h[2] - h[4], 1.0 / h[5]};
    for (int i = 0; i < 8; i++) printf("%d gpu %.17g host %.17g %s\n", i, r[i], e[i], r[i] == e[i] ? "same" : "DIFFER");
    return 0;
}
