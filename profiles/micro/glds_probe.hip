// Probe: global_load_lds_dwordx4 from sources that are only 8-byte aligned (odd double index): does the data land right?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__global__ void probe(const double *src, double *out, int off) {
    __shared__ __attribute__((aligned(1024))) double lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -1.0;
    __syncthreads();
    // lane reads 16 B at src + off + 2*lane doubles -> LDS base + 16*lane
    const double *g = src + off + 2 * lane;
    __builtin_amdgcn_global_load_lds((gptr_t *)g, (lptr_t *)(lds + 4), 16, 0, 0);   // LDS base 32 B into the array
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = i;
    double *d, *o;
    hipMalloc(&d, 4096 * 8); hipMalloc(&o, 512 * 8);
    hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    for (int off : {0, 1, 2, 3, 17}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, off);
        std::vector<double> r(512);
        hipMemcpy(r.data(), o, 512 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 128; i++) if (r[4 + i] != off + i) bad++;
        printf("offset %d doubles (%s-aligned source): %d of 128 values wrong; lds[3]=%g lds[4]=%g lds[5]=%g lds[131]=%g lds[132]=%g\n", off, off % 2 ? "8-byte" : "16-byte", bad, r[3], r[4], r[5], r[131], r[132]);
    }
    return 0;
}
