// Microbenchmark: f32 FMA issue rates on gfx950 — v_fma_f32 vs v_pk_fma_f32,
// VGPR vs SGPR multiplicand, at 1/2/4 waves per SIMD.  Decides how the FIR
// inner loop of the fused chain kernel is written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float c0, float c1) {
    float2v acc[16];
    float2v x[4];
    for (int j = 0; j < 16; ++j) acc[j] = {threadIdx.x * 1e-6f + j, 1.0f - j};
    for (int j = 0; j < 4; ++j) x[j] = {1.0f + threadIdx.x * 1e-7f * (j + 1), 0.5f + j};
    float2v cc = {c0, c1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) {  // 2 x v_fma_f32, VGPR operands
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(x[j & 3].x), "v"(cc.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(x[j & 3].y), "v"(cc.x));
            } else if (MODE == 1) {  // 2 x v_fma_f32, SGPR multiplicand
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(x[j & 3].x), "s"(c0));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(x[j & 3].y), "s"(c0));
            } else if (MODE == 2) {  // 1 x v_pk_fma_f32, VGPR pair operands
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j & 3]), "v"(cc));
            } else if (MODE == 3) {  // v_pk_fma_f32, multiplicand low half broadcast (op_sel)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(x[j & 3]), "v"(cc));
            } else if (MODE == 4) {  // v_pk_fma_f32 with SGPR pair multiplicand
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j & 3]), "s"(cc));
            } else if (MODE == 5) {  // v_pk_fma_f32 with SGPR pair, low half broadcast
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(x[j & 3]), "s"(cc));
            }
        }
    }
    float2v s = {0, 0};
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE> void run(const char *name, int blocks_per_cu) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    float *d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.9999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double flop = 2.0 * 2 * 16 * (double)iters * blocks * 256;  // 32 FMA per iter per lane
    printf("%-40s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int occ : {1, 2, 4}) {
        run<0>("v_fma_f32 vgpr,vgpr", occ);
        run<1>("v_fma_f32 vgpr,sgpr", occ);
        run<2>("v_pk_fma_f32 vgpr,vgpr", occ);
        run<3>("v_pk_fma_f32 vgpr,vgpr lo-bcast", occ);
        run<4>("v_pk_fma_f32 vgpr,sgpr", occ);
        run<5>("v_pk_fma_f32 vgpr,sgpr lo-bcast", occ);
    }
    return 0;
}
