// Does the speed of a streaming read (+ a small write stream) depend on WHERE in a large allocation the read region and the written
// region lie?  (The engine's reduce_kernel takes 1.10 or 1.21 ms for the same 7.2 GB depending on where the driver put its arrays:
// tools/placement_builds.py.)  One 248 GiB allocation; the read region (4 GiB) and the written region (64 MiB) are placed at several offsets.
//   hipcc --offload-arch=gfx950 -O3 tools/placement_rw.hip -o /tmp/placement_rw && /tmp/placement_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int WRITE>
__global__ __launch_bounds__(256) void rw(const d2 *__restrict__ in, double *__restrict__ out, int64_t n, int every, int64_t out_elems) {
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
        if (WRITE && (it % every) == 0)
            mine[(it / every) * 64 + lane] = (s0 + s1) + (s2 + s3);
    }
    if (s0 == 12345.678)
        out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = (s0 + s1) + (s2 + s3);
}
template <typename K>
static double run(K k, const d2 *a, double *o, int64_t n, int every, int64_t oe, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, o, n, every, oe);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, o, n, every, oe);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return (double)n * 16 * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const int64_t GB = 1ll << 30, MB = 1ll << 20;
    const int64_t total = 248 * GB, rbytes = 4 * GB, zbytes = 64 * MB;
    char *B = nullptr;
    if (hipMalloc(&B, total) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    hipMemset(B, 0, total);
    const int64_t n = rbytes / 16, oe = zbytes / 8;
    const int grid = 4096;
    printf("base %p\n", (void *)B);
    // how the price of the write stream falls with its share, same class (written region next to the read region) against the best of the others
    {
        const int64_t ro = 2 * GB;
        for (int every : {1, 2, 8, 32, 64, 128}) {
            double same = run(rw<1>, (const d2 *)(B + ro), (double *)(B + ro + rbytes), n, every, oe, grid), best = 0;
            for (int64_t z = 32 * GB; z + zbytes <= total; z += 24 * GB)
                best = std::max(best, run(rw<1>, (const d2 *)(B + ro), (double *)(B + z), n, every, oe, grid));
            printf("written share %.3f %%: %.0f GB/s next to the read region, %.0f GB/s at the best of the other places\n", 100.0 * 512 / (every * 4 * 4096.0), same, best);
        }
    }
    // the map: written region every 2 GiB over the whole allocation, for three positions of the read region (1.6 % written)
    for (int64_t ro : {2 * GB, 70 * GB, 134 * GB, 198 * GB, 240 * GB}) {
        printf("read region at %lld GiB; GB/s with the written region at 0, 4, 8, ... GiB:\n", (long long)(ro / GB));
        for (int64_t z = 0; z + zbytes <= total; z += 4 * GB) {
            const bool inside = z + zbytes > ro && z < ro + rbytes;
            printf(" %s%.0f", inside ? "*" : "", inside ? 0.0 : run(rw<1>, (const d2 *)(B + ro), (double *)(B + z), n, 2, oe, grid));
        }
        printf("\n");
    }
    return 0;
}
