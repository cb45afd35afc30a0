// how long do large device allocations take on this box? (build-time budget of the compression pool / streams)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#include <cstdlib>
#include <cstdint>
#include <vector>
int main(int argc, char **argv) {
    hipFree(0);
    std::vector<double> sizes = {1.0, 8.0, 32.0, 115.0, 32.0};
    if (argc > 1) { // sizes in GB from the command line
        sizes.clear();
        for (int i = 1; i < argc; i++)
            sizes.push_back(atof(argv[i]));
    }
    for (double gb : sizes) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
        double t1 = now();
        hipMemset(p, 0, (size_t)(gb * 1e9));
        hipDeviceSynchronize();
        double t2 = now();
        hipFree(p);
        double t3 = now();
        printf("hipMalloc %.0f GB: %s malloc %.3fs  first memset %.3fs  free %.3fs\n", gb, hipGetErrorString(e), t1 - t0, t2 - t1, t3 - t2);
    }
    hipStream_t s;
    hipStreamCreate(&s);
    {
        hipMemPool_t pool;
        hipDeviceGetDefaultMemPool(&pool, 0);
        uint64_t keep = UINT64_MAX; // freed memory stays in the pool
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    for (double gb : sizes) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMallocAsync(&p, (size_t)(gb * 1e9), s);
        hipStreamSynchronize(s);
        double t1 = now();
        hipFreeAsync(p, s);
        hipStreamSynchronize(s);
        double t2 = now();
        printf("hipMallocAsync %.0f GB: %s malloc %.3fs free %.3fs\n", gb, hipGetErrorString(e), t1 - t0, t2 - t1);
    }
    return 0;
}
