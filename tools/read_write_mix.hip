// What does a small write stream cost a streaming read on this GPU?  1024 workgroups each read their own contiguous chunk with
// non-temporal 16-byte loads (the access pattern of the stream kernels); every `every`-th iteration a wave also stores 512 bytes
// (64 lanes x 8 B) to (a) one fixed line per wave, (b) its own contiguous output region, (c) a scattered region.
// Written bytes / read bytes = 512 / (every * 4 * 4096): every = 2 -> 1.6 % (the ratio of the fused symmetric product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void rw(const d2 *__restrict__ in, double *__restrict__ out, int64_t n, int every, int64_t out_elems, int win) {
    const int64_t per = n / gridDim.x;
    const d2 *p       = in + per * blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t it = 0;
    double *mine = out + ((int64_t)blockIdx.x * 4 + wave) * (out_elems / (gridDim.x * 4));
    for (int64_t i = threadIdx.x; i + 3 * 256 < per; i += 4 * 256, it++) {
        const d2 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 256), c = __builtin_nontemporal_load(p + i + 512), d = __builtin_nontemporal_load(p + i + 768);
        s0 += a.x + a.y;
        s1 += b.x + b.y;
        s2 += c.x + c.y;
        s3 += d.x + d.y;
        if (MODE && every && (it % every) == 0) {
            const int64_t k = it / every;
            if (MODE == 1)
                out[((int64_t)blockIdx.x * 4 + wave) * 64 + lane] = (s0 + s1) + (s2 + s3); // one fixed line per wave
            else if (MODE == 2)
                mine[k * 64 + lane] = (s0 + s1) + (s2 + s3); // contiguous per wave
            else if (MODE == 3)
                out[(((int64_t)blockIdx.x * 2654435761u + k * 40503u + wave) % (out_elems / 64)) * 64 + lane] = (s0 + s1) + (s2 + s3); // scattered 512-byte runs
            else if (MODE == 4)
                __builtin_nontemporal_store((s0 + s1) + (s2 + s3), mine + k * 64 + lane); // contiguous per wave, non-temporal
            else if (MODE == 7) { // contiguous per wave, but only every 8th workgroup writes (8 runs each time): the same volume from 1/8 of the CUs
                if ((blockIdx.x & 7) == 0)
                    for (int q = 0; q < 8; q++)
                        mine[(k * 8 + q) * 64 + lane] = (s0 + s1) + (s2 + s3);
            } else if (MODE == 8) { // contiguous per wave, write-through to memory (sc0 sc1)
                double v = (s0 + s1) + (s2 + s3);
                double *pp = mine + k * 64 + lane;
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(pp), "v"(v) : "memory");
            } else if (MODE == 9) { // contiguous per wave, nt + sc1 (streaming)
                double v = (s0 + s1) + (s2 + s3);
                double *pp = mine + k * 64 + lane;
                asm volatile("global_store_dwordx2 %0, %1, off nt sc1" ::"v"(pp), "v"(v) : "memory");
            } else if (MODE == 6) // one fixed 512-byte run per wave, the runs `win` x 512 bytes apart
                out[(((int64_t)blockIdx.x * 4 + wave) * win % (out_elems / 64)) * 64 + lane] = (s0 + s1) + (s2 + s3);
            else // contiguous per wave inside a window of `win` elements per wave (the whole grid writes 4096 * win * 8 bytes, again and again)
                mine[(k % (win / 64)) * 64 + lane] = (s0 + s1) + (s2 + s3);
        }
    }
    if (!MODE || s0 == 12345.678)
        out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = (s0 + s1) + (s2 + s3);
}
template <typename K>
static double run(K k, const d2 *a, double *o, int64_t n, int every, int64_t oe, int win = 64) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, a, o, n, every, oe, win);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, a, o, n, every, oe, win);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)n * 16 * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const int64_t n = (int64_t)(8ll << 30) / 16, oe = (int64_t)(1ll << 30) / 8;
    d2 *a;
    double *o;
    hipMalloc(&a, n * 16);
    hipMalloc(&o, oe * 8);
    hipMemset(a, 0, n * 16);
    printf("read only: %.0f GB/s\n", run(rw<0>, a, o, n, 0, oe));
    for (int every : {8, 4, 2, 1})
        printf("every %d (%.2f %% written): fixed line %.0f, contiguous %.0f, scattered %.0f GB/s (read bytes / time)\n", every, 100.0 * 512 / (every * 4 * 4096.0),
               run(rw<1>, a, o, n, every, oe), run(rw<2>, a, o, n, every, oe), run(rw<3>, a, o, n, every, oe));
    for (int win : {64, 512, 4096, 8192, 32768}) // 2 MB, 16 MB, 128 MB, 256 MB, 1 GB written region
        printf("every 2, window %.0f MB: contiguous-in-window %.0f GB/s\n", 4096.0 * win * 8 / 1e6, run(rw<5>, a, o, n, 2, oe, win));
    for (int stride : {1, 8, 128, 509, 512, 4096})
        printf("every 2, one fixed 512-byte run per wave, runs %d x 512 bytes apart: %.0f GB/s\n", stride, run(rw<6>, a, o, n, 2, oe, stride));
    printf("every 2: same volume written by every 8th workgroup only: %.0f GB/s\n", run(rw<7>, a, o, n, 2, oe));
    printf("every 2: contiguous, sc0 sc1 stores: %.0f GB/s\n", run(rw<8>, a, o, n, 2, oe));
    printf("every 2: contiguous, nt sc1 stores: %.0f GB/s\n", run(rw<9>, a, o, n, 2, oe));
    { // the written buffer in other kinds of device memory (the L2 treats them differently)
        for (unsigned flag : {0x3u /* hipDeviceMallocUncached */, 0x1u /* hipDeviceMallocFinegrained */}) {
            double *o2 = nullptr;
            if (hipExtMallocWithFlags((void **)&o2, oe * 8, flag) != hipSuccess) {
                printf("hipExtMallocWithFlags(0x%x) failed\n", flag);
                continue;
            }
            printf("every 2, written buffer hipExtMallocWithFlags(0x%x): contiguous %.0f, scattered %.0f GB/s\n", flag, run(rw<2>, a, o2, n, 2, oe), run(rw<3>, a, o2, n, 2, oe));
            hipFree(o2);
        }
    }
    printf("every 2: contiguous non-temporal store %.0f GB/s\n", run(rw<4>, a, o, n, 2, oe));
    return 0;
}
