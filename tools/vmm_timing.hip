// What do large device allocations cost on this box, and does the virtual-memory API avoid it?
//   (1) hipMalloc sequences like a build makes them (pool, bigger pool, streams), live at the same time
//   (2) hipMemCreate of physical chunks + hipMemMap into a reserved range; unmap and map the same chunks again (recycling)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipFree(0);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    {
        const double seq[] = {40, 80, 50, 55, 20, 20, 20};
        std::vector<void *> live;
        for (double gb : seq) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
            double t1 = now();
            printf("hipMalloc %3.0f GB (live before: %zu buffers): %s %.3f s\n", gb, live.size(), hipGetErrorString(e), t1 - t0);
            if (e == hipSuccess)
                live.push_back(p);
            if (live.size() == 3) { // free the two oldest, like the pool growth + shrink do
                double t2 = now();
                hipFree(live[0]);
                hipFree(live[1]);
                printf("   2 x hipFree %.3f s\n", now() - t2);
                live.erase(live.begin(), live.begin() + 2);
            }
        }
        for (void *p : live)
            hipFree(p);
    }
    // ---- virtual memory management ----------------------------------------------------------------
    hipMemAllocationProp prop = {};
    prop.type          = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id   = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu bytes\n", gran);
    const size_t chunk = (size_t)2 << 30, nchunks = 48; // 96 GB
    void *va = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&va, chunk * nchunks, 0, nullptr, 0));
    printf("hipMemAddressReserve %zu GB: %.3f s\n", (chunk * nchunks) >> 30, now() - t0);
    std::vector<hipMemGenericAllocationHandle_t> h(nchunks);
    t0 = now();
    for (size_t i = 0; i < nchunks; i++)
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
    printf("hipMemCreate %zu x 2 GB: %.3f s\n", nchunks, now() - t0);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags    = hipMemAccessFlagsProtReadWrite;
    for (int rep = 0; rep < 2; rep++) {
        t0 = now();
        for (size_t i = 0; i < nchunks; i++)
            CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess(va, chunk * nchunks, &acc, 1));
        double t1 = now();
        CK(hipMemset(va, 1, chunk * nchunks));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipMemUnmap(va, chunk * nchunks));
        printf("map + set access %.3f s, memset of the range %.3f s, unmap %.3f s\n", t1 - t0, t2 - t1, now() - t2);
    }
    t0 = now();
    for (size_t i = 0; i < nchunks; i++)
        CK(hipMemRelease(h[i]));
    CK(hipMemAddressFree(va, chunk * nchunks));
    printf("release %.3f s\n", now() - t0);
    return 0;
}
