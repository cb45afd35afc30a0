// Does a buffer that fits the 256 MB memory-side cache stream faster than one that does not?  Every workgroup reads its own contiguous chunk
// (16-byte loads, eight in flight per lane), the same buffer twenty launches in a row; temporal and non-temporal loads.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_bw.hip -o tools/_mall_bw && tools/_mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void rd_chunk(const d2 *__restrict__ in, double *__restrict__ out, int64_t n) {
    constexpr int U   = 8;
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
static double run(K k, const d2 *a, double *o, int64_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, a, o, n);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; r++)
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, a, o, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)n * 16 * 20 / (ms * 1e-3) / 1e9;
}
int main() {
    d2 *a;
    double *o;
    hipMalloc(&a, (size_t)8 << 30);
    hipMalloc(&o, 1024 * 256 * 8);
    hipMemset(a, 0, (size_t)8 << 30);
    for (int mb : {32, 64, 128, 192, 256, 384, 512, 2048, 8192}) {
        const int64_t n = ((int64_t)mb << 20) / 16;
        printf("%5d MB, 20 launches in a row: non-temporal %.0f GB/s, temporal %.0f GB/s\n", mb, run(rd_chunk<true>, a, o, n), run(rd_chunk<false>, a, o, n));
    }
    return 0;
}
