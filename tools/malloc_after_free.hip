// When does hipMalloc stall on this box?  Scenario from argv[1]; every buffer is touched (memset) so that it is really backed.
//   nofree   : 80, 50, 55 GB, nothing freed in between
//   free     : 80, 50, free(80), 55
//   freewait : 80, 50, free(80), sleep 6 s, 55
//   small    : 80, 50, free(50), 55... (20 GB freed) -> 20
//   same     : 80, free, 80 again
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void *A(double gb) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
    double t1 = now();
    hipMemset(p, 0, (size_t)(gb * 1e9));
    hipDeviceSynchronize();
    printf("  malloc %3.0f GB %s %.3f s (+ memset %.3f s)\n", gb, hipGetErrorString(e), t1 - t0, now() - t1);
    return p;
}
static void F(void *p) {
    double t0 = now();
    hipFree(p);
    printf("  free %.3f s\n", now() - t0);
}
int main(int argc, char **argv) {
    const char *s = argc > 1 ? argv[1] : "nofree";
    hipFree(0);
    printf("%s\n", s);
    if (!strcmp(s, "nofree")) { void *a = A(80), *b = A(50), *c = A(55); (void)a; (void)b; (void)c; }
    if (!strcmp(s, "free")) { void *a = A(80), *b = A(50); F(a); void *c = A(55); (void)b; (void)c; }
    if (!strcmp(s, "freewait")) { void *a = A(80), *b = A(50); F(a); std::this_thread::sleep_for(std::chrono::seconds(6)); void *c = A(55); (void)b; (void)c; }
    if (!strcmp(s, "small")) { void *a = A(80), *b = A(20); F(b); void *c = A(20); (void)a; (void)c; }
    if (!strcmp(s, "same")) { void *a = A(80); F(a); void *c = A(80); (void)c; }
    return 0; // the process exit frees the rest
}
