// Does a process pay for the memory the PREVIOUS process released?  "fill <GB>": allocate, touch, exit.  "probe": time a sequence of
// allocations right after.  "probe_pretouch": first allocate (and free) 90 % of the free memory, then the same sequence.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void *A(double gb, bool touch = true) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
    double t1 = now();
    if (touch) {
        hipMemset(p, 0, (size_t)(gb * 1e9));
        hipDeviceSynchronize();
    }
    printf("  malloc %5.1f GB %s %.3f s (+ memset %.3f s)\n", gb, hipGetErrorString(e), t1 - t0, now() - t1);
    return p;
}
int main(int argc, char **argv) {
    const char *s = argc > 1 ? argv[1] : "probe";
    double t0 = now();
    hipFree(0);
    printf("%s (context %.3f s)\n", s, now() - t0);
    if (!strcmp(s, "fill")) {
        A(atof(argv[2]));
        return 0;
    }
    if (!strcmp(s, "probe_pretouch")) {
        size_t fr = 0, tot = 0;
        hipMemGetInfo(&fr, &tot);
        void *p = A(0.9 * fr / 1e9, false);
        double t1 = now();
        hipFree(p);
        printf("  free %.3f s\n", now() - t1);
    }
    void *a = A(20), *b = A(100);
    hipFree(a);
    hipFree(b);
    void *c = A(120);
    void *d = A(60);
    (void)c; (void)d;
    return 0;
}
