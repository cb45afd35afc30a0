// What does a pure streaming read reach on this GPU?  Variants: grid size, loads in flight per lane, temporal vs non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const d2 *__restrict__ in, double *__restrict__ out, int64_t n) {
    int64_t i            = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        s[u] = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++)
            s[u] += v[u].x + v[u].y;
    }
    double t = 0;
#pragma unroll
    for (int u = 0; u < U; u++)
        t += s[u];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}
// contiguous chunk per workgroup (like the stream kernels: a wave walks its own region)
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd_chunk(const d2 *__restrict__ in, double *__restrict__ out, int64_t n) {
    const int64_t per = n / gridDim.x;
    const d2 *p       = in + per * blockIdx.x;
    double s[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        s[u] = 0;
    for (int64_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++)
            s[u] += v[u].x + v[u].y;
    }
    double t = 0;
#pragma unroll
    for (int u = 0; u < U; u++)
        t += s[u];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <typename K>
static double run(K k, int blocks, const d2 *a, double *o, int64_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, o, n);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, o, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)n * 16 * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const int64_t n = (int64_t)(16ll << 30) / 16;
    d2 *a;
    double *o;
    hipMalloc(&a, n * 16);
    hipMalloc(&o, 65536 * 256 * 8);
    hipMemset(a, 0, n * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        printf("grid %6d  strided: U4 nt %.0f  U8 nt %.0f  U16 nt %.0f  U8 temporal %.0f | chunked: U4 nt %.0f  U8 nt %.0f  U8 temporal %.0f GB/s\n", blocks,
               run(rd<4, true>, blocks, a, o, n), run(rd<8, true>, blocks, a, o, n), run(rd<16, true>, blocks, a, o, n), run(rd<8, false>, blocks, a, o, n),
               run(rd_chunk<4, true>, blocks, a, o, n), run(rd_chunk<8, true>, blocks, a, o, n), run(rd_chunk<8, false>, blocks, a, o, n));
    }
    return 0;
}
