// Microbenchmark: what HBM rate does a streaming kernel with the fused FIR's traffic shape reach?
// R bytes read for every W bytes written (R:W = 4:1 for k_ols_wave, 1:1 for k_fft4096 / a copy,
// read-only as the upper end), 16-byte accesses, one 1 KiB row per wave per trip, sizes as in cfg2
// (512 MiB read).  Gives the denominator "what the memory system can do" for DESIGN.md section 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

// each wave takes chunks of 4 KiB (4 rows of 1 KiB = 64 lanes x 16 B): reads RD of them, writes WR rows
template <int RD, int WR, bool NT>
__global__ __launch_bounds__(256) void k(const f4 *__restrict__ in, f4 *__restrict__ out, size_t chunks) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nw = (size_t)gridDim.x * 4;
    const int l = threadIdx.x & 63;
    for (size_t c = wave; c < chunks; c += nw) {
        f4 v[RD];
#pragma unroll
        for (int r = 0; r < RD; ++r) v[r] = NT ? __builtin_nontemporal_load(in + (c * RD + r) * 64 + l) : in[(c * RD + r) * 64 + l];
        f4 s = v[0];
#pragma unroll
        for (int r = 1; r < RD; ++r) s += v[r];
        if (WR == 0) {
            if (s.x == 123.456f) out[0] = s;  // never true for the zero-filled input: read-only
        } else {
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                if (NT) __builtin_nontemporal_store(s, out + (c * WR + r) * 64 + l);
                else out[(c * WR + r) * 64 + l] = s;
            }
        }
    }
}

// RING input/output buffer pairs are used in turn, so that what one launch touches (up to 1 GiB) has left the
// 256 MB Infinity Cache long before it is touched again
constexpr int RING = 6;
template <int RD, int WR, bool NT> void run(const char *name, int blocks, size_t read_bytes) {
    const size_t chunks = read_bytes / (RD * 1024);
    const size_t wbytes = WR ? chunks * WR * 1024 : 1024;
    f4 *in, *out;
    hipMalloc(&in, read_bytes * RING);
    hipMalloc(&out, wbytes * RING);
    hipMemset(in, 0, read_bytes * RING);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto launch = [&](int i) {
        const int s = i % RING;
        hipLaunchKernelGGL((k<RD, WR, NT>), dim3(blocks), dim3(256), 0, 0, in + (read_bytes / 16) * s, out + (wbytes / 16) * s, chunks);
    };
    for (int i = 0; i < 30; ++i) launch(i);
    hipDeviceSynchronize();
    const int K = 96;
    hipEventRecord(a);
    for (int i = 0; i < K; ++i) launch(i);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= K;
    const double bytes = (double)chunks * (RD + WR) * 1024;
    printf("%-28s grid %6d: %.4f ms  %.0f GB/s (read %.0f MB, write %.0f MB)\n", name, blocks, ms, bytes / ms / 1e6,
           chunks * RD * 1024 / 1e6, chunks * WR * 1024 / 1e6);
    hipFree(in);
    hipFree(out);
}

int main() {
    const size_t rb = 512ull << 20;
    for (int blocks : {2048, 16384, 131072}) {
        run<4, 0, false>("read only", blocks, rb);
        run<4, 0, true>("read only, nontemporal", blocks, rb);
        run<4, 1, false>("4:1 read:write", blocks, rb);
        run<4, 1, true>("4:1 read:write, nontemporal", blocks, rb);
        run<4, 4, true>("1:1 (copy), nontemporal", blocks, rb);
        run<8, 2, true>("4:1, 8 rows in flight, nt", blocks, rb);
        run<1, 4, true>("1:4 read:write (Stft), nt", blocks, rb / 4);
        run<1, 4, false>("1:4 read:write (Stft)", blocks, rb / 4);
        run<1, 8, true>("1:8 read:write, nt", blocks, rb / 8);
    }
    return 0;
}
