// cpu_scaling.cpp -- how many cores does this box really give a process?  Aggregate throughput of a compute-bound loop (sqrt + divide, the
// cost profile of a BEM kernel entry) for 1 ... 256 threads, plus the cgroup limits.  g++ -O2 -pthread tools/cpu_scaling.cpp -o tools/_cpu_scaling
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main() {
    for (const char *f : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"}) {
        FILE *fp = fopen(f, "r");
        if (fp) {
            char buf[256] = {0};
            if (fgets(buf, sizeof buf, fp))
                printf("%s: %s", f, buf);
            fclose(fp);
        }
    }
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (int nt : {1, 4, 8, 16, 32, 64, 128, 256}) {
        std::vector<double> out(nt);
        const long iters = 20000000;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                double s = 0, x = 1.0 + t;
                for (long i = 0; i < iters; i++) {
                    s += 1.0 / (1e-5 + std::sqrt(x));
                    x += 1e-3;
                }
                out[t] = s;
            });
        for (auto &x : th)
            x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %3d: %.3f s, %.2f G entries/s aggregate (%.2f per thread)\n", nt, dt, nt * iters / dt * 1e-9, iters / dt * 1e-9);
    }
    return 0;
}
