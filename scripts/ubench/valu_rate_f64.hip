// Microbenchmark: v_fma_f64 issue rate on gfx950 (VGPR / SGPR multiplicand) at 1 / 2 / 4 waves per SIMD.
// Decides what bounds k_decim_poly_f64r (rr_decim.hip): 16 independent accumulators per lane, as that kernel's 8 x 2.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double c0) {
    double acc[16], x[4];
    for (int j = 0; j < 16; ++j) acc[j] = threadIdx.x * 1e-6 + j;
    for (int j = 0; j < 4; ++j) x[j] = 1.0 + threadIdx.x * 1e-9 * (j + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j & 3]), "v"(c0));
            else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j & 3]), "s"(c0));
        }
    }
    double s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int blocks_per_cu) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double fma = 16.0 * (double)iters * blocks * 256;
    // wave-instructions per SIMD: blocks_per_cu waves per SIMD, 16 iters each
    const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters * blocks_per_cu);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %6.1f TFLOP/s  %5.2f clocks of 2.4 GHz per wave instruction\n", name, blocks_per_cu, ms,
           2.0 * fma / ms / 1e9, cyc);
    hipFree(d);
}

int main() {
    for (int occ : {1, 2, 4}) {
        run<0>("v_fma_f64 vgpr,vgpr", occ);
        run<1>("v_fma_f64 vgpr,sgpr", occ);
    }
    return 0;
}
