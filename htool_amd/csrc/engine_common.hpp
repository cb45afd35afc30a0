// engine_common.hpp -- what every translation unit of the device side shares: error macro, device arrays with RAII, the in-process
// buffer cache and the reserved slabs, memory queries, small host helpers.  The engine itself (engine_body.hpp + kernels_body.hpp) is
// compiled once per coefficient type (engine_inst.hip, -DHMX_INST=0..3) so that the four instantiations build in parallel; engine.hip
// holds the C ABI and the DistributedOperator layer and reaches the instantiations through the prototypes of engine_api.hpp.
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <numeric>
#include <string>
#include <vector>

#include "hmx_host.hpp"
#include "kernels_common.hpp"

namespace hmx {

#define HMX_HIP(call)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
            return HMX_ERR_HIP;                                                                               \
        }                                                                                                     \
    } while (0)

// Large device buffers are recycled inside the process: after big frees the runtime's next large hipMalloc can take seconds
// (tools/malloc_timing.hip, tools/build_timing.py rep 1), which hurts callers that rebuild operators (parameter sweeps, time
// stepping).  release() parks buffers >= 64 MiB in a per-device free list (at most HMX_CACHE_GB, default 48, in total), alloc()
// takes the smallest parked buffer that fits with at most 25 % waste.  hmx_device_trim_cache() returns everything to the driver.
struct DeviceCache {
    struct Entry {
        void *p;
        size_t bytes;
        int device;
    };
    std::mutex mu;
    std::vector<Entry> parked;
    size_t total = 0;
    static DeviceCache &get() {
        static DeviceCache c;
        return c;
    }
    static size_t limit() {
        static const size_t l = (size_t)((getenv("HMX_CACHE_GB") ? atof(getenv("HMX_CACHE_GB")) : 48.0) * 1073741824.0);
        return l;
    }
    void *take(size_t bytes, size_t *got) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        int best = -1;
        for (size_t i = 0; i < parked.size(); i++)
            if (parked[i].device == dev && parked[i].bytes >= bytes && parked[i].bytes <= bytes + bytes / 4 && (best < 0 || parked[i].bytes < parked[best].bytes))
                best = (int)i;
        if (best < 0)
            return nullptr;
        void *p = parked[best].p;
        *got    = parked[best].bytes;
        total -= parked[best].bytes;
        parked.erase(parked.begin() + best);
        return p;
    }
    // `dev` is the device the buffer was allocated on (recorded by DArr, not the caller's current device).  hipFree used to
    // synchronise implicitly; a parked buffer can be handed out again at once, so work still in flight on ANY stream of the owning
    // device (a caller's non-blocking stream, a graph launch) is waited for first.  Large releases are rare (operator teardown,
    // buffer growth), so the device-wide wait costs nothing that matters.
    bool park(void *p, size_t bytes, int dev) {
        if (bytes < (size_t(64) << 20) || bytes > limit())
            return false;
        int cur = dev;
        (void)hipGetDevice(&cur);
        if (cur != dev)
            (void)hipSetDevice(dev);
        const hipError_t se = hipDeviceSynchronize();
        if (cur != dev)
            (void)hipSetDevice(cur);
        if (se != hipSuccess) { // e.g. a stream capture in progress: do not recycle what cannot be proven idle
            (void)hipGetLastError();
            return false;
        }
        std::lock_guard<std::mutex> lock(mu);
        while (total + bytes > limit() && !parked.empty()) { // evict the oldest
            (void)hipFree(parked.front().p);
            total -= parked.front().bytes;
            parked.erase(parked.begin());
        }
        parked.push_back({p, bytes, dev});
        total += bytes;
        return true;
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto &e : parked)
            (void)hipFree(e.p);
        parked.clear();
        total = 0;
    }
};

// hmx_device_reserve(): one slab per call, taken from the driver once; every later device array of >= 1 MiB is carved out of a slab
// (first fit over an offset-ordered free list, 2 MiB granularity, neighbours coalesced on release) before hipMalloc is asked.
// On this platform hipMalloc stalls for seconds when memory was released shortly before -- by this process or by the one that ran
// before it (tools/malloc_after_exit.hip: ~25 ms per GB until the driver has scrubbed what came back) -- and an operator build
// allocates its two largest arrays (cross pool, streams) right where that hurts; a caller that builds operators repeatedly, or
// times a build, pays once and outside.  Released ranges become reusable after a device-wide synchronisation, like hipFree.
struct DeviceSlabs {
    struct Slab {
        char *base;
        size_t bytes;
        int device;
        std::map<size_t, size_t> free_at; // offset -> size
        size_t in_use = 0;
    };
    std::mutex mu;
    std::vector<Slab> slabs;
    std::atomic<long long> generation{0}; // changes when a slab comes or goes: what was measured about places of the old ones is void (PlacementCache)
    static constexpr size_t GRAIN = size_t(2) << 20;
    static DeviceSlabs &get() {
        static DeviceSlabs s;
        return s;
    }
    hipError_t reserve(int dev, size_t bytes) {
        bytes = (bytes + GRAIN - 1) / GRAIN * GRAIN;
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess)
            return e;
        std::lock_guard<std::mutex> lock(mu);
        Slab s{static_cast<char *>(p), bytes, dev, {}, 0};
        s.free_at[0] = bytes;
        slabs.push_back(std::move(s));
        generation++;
        return hipSuccess;
    }
    void *take(int dev, size_t bytes, size_t *got) {
        const size_t need = (bytes + GRAIN - 1) / GRAIN * GRAIN;
        std::lock_guard<std::mutex> lock(mu);
        for (Slab &s : slabs) {
            if (s.device != dev)
                continue;
            for (auto it = s.free_at.begin(); it != s.free_at.end(); ++it)
                if (it->second >= need) {
                    const size_t off = it->first, rest = it->second - need;
                    s.free_at.erase(it);
                    if (rest)
                        s.free_at[off + need] = rest;
                    s.in_use += need;
                    *got = need;
                    return s.base + off;
                }
        }
        return nullptr;
    }
    // ... a range as close as possible to the fraction `frac` of a slab's extent (frac = 1: from the top): for arrays that should NOT lie
    // next to what first fit hands out (place_written)
    void *take_at(int dev, size_t bytes, double frac, size_t *got) {
        const size_t need = (bytes + GRAIN - 1) / GRAIN * GRAIN;
        std::lock_guard<std::mutex> lock(mu);
        Slab *best_s = nullptr;
        size_t best_off = 0, best_dist = ~size_t(0), best_range = 0;
        for (Slab &s : slabs) {
            if (s.device != dev)
                continue;
            const size_t target = (size_t)(frac * (double)(s.bytes - std::min(s.bytes, need))) / GRAIN * GRAIN;
            for (const auto &r : s.free_at) {
                if (r.second < need)
                    continue;
                const size_t lo = r.first, hi = r.first + r.second - need; // possible starts inside this free range
                const size_t at = target < lo ? lo : (target > hi ? hi : target);
                const size_t dist = at > target ? at - target : target - at;
                if (dist < best_dist) {
                    best_dist = dist, best_off = at, best_range = r.first, best_s = &s;
                }
            }
        }
        if (!best_s)
            return nullptr;
        const size_t r_off = best_range, r_len = best_s->free_at[best_range];
        best_s->free_at.erase(best_range);
        if (best_off > r_off)
            best_s->free_at[r_off] = best_off - r_off;
        if (best_off + need < r_off + r_len)
            best_s->free_at[best_off + need] = r_off + r_len - (best_off + need);
        best_s->in_use += need;
        *got = need;
        return best_s->base + best_off;
    }
    // true when p belongs to a slab (and is free again afterwards)
    bool give_back(void *p, size_t bytes) {
        std::lock_guard<std::mutex> lock(mu);
        for (Slab &s : slabs) {
            char *c = static_cast<char *>(p);
            if (c < s.base || c >= s.base + s.bytes)
                continue;
            size_t off = (size_t)(c - s.base), len = bytes;
            s.in_use -= len;
            auto next = s.free_at.lower_bound(off);
            if (next != s.free_at.end() && off + len == next->first) { // merge with the range after
                len += next->second;
                next = s.free_at.erase(next);
            }
            if (next != s.free_at.begin()) { // ... and with the one before
                auto prev = std::prev(next);
                if (prev->first + prev->second == off) {
                    off = prev->first;
                    len += prev->second;
                    s.free_at.erase(prev);
                }
            }
            s.free_at[off] = len;
            return true;
        }
        return false;
    }
    bool owns(const void *p) {
        std::lock_guard<std::mutex> lock(mu);
        for (const Slab &s : slabs)
            if (static_cast<const char *>(p) >= s.base && static_cast<const char *>(p) < s.base + s.bytes)
                return true;
        return false;
    }
    size_t free_bytes(int dev) {
        std::lock_guard<std::mutex> lock(mu);
        size_t f = 0;
        for (const Slab &s : slabs)
            if (s.device == dev)
                f += s.bytes - s.in_use;
        return f;
    }
    // the largest single array a slab of this device can still hold (free ranges do not join across slabs or across used ranges)
    size_t largest_hole(int dev) {
        std::lock_guard<std::mutex> lock(mu);
        size_t f = 0;
        for (const Slab &s : slabs)
            if (s.device == dev)
                for (const auto &r : s.free_at)
                    f = std::max(f, r.second);
        return f;
    }
    // slabs nothing is carved out of go back to the driver
    size_t release_idle() {
        std::lock_guard<std::mutex> lock(mu);
        size_t kept = 0;
        for (size_t i = 0; i < slabs.size();) {
            if (slabs[i].in_use == 0) {
                (void)hipFree(slabs[i].base);
                slabs.erase(slabs.begin() + i);
                generation++;
            } else {
                kept += slabs[i].bytes;
                i++;
            }
        }
        return kept;
    }
};

// wall time this process spent inside hipMalloc (large allocations sporadically take seconds on this platform: tools/malloc_timing.hip);
// bench.py reports it next to the build time
inline std::atomic<long long> g_malloc_ns{0}; // one counter for every translation unit of the library
// device arrays handed out so far, wherever they came from (driver, reserved slab, in-process cache): hmx_device_alloc_count()
inline std::atomic<long long> g_alloc_count{0};
struct MallocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~MallocTimer() { g_malloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// std::vector whose resize() leaves the new elements uninitialised: the large host index arrays are written completely (by several
// threads) right after they are sized, and zero-filling 150 MB first costs as much as filling it
template <typename T>
struct DefaultInitAlloc : std::allocator<T> {
    template <typename U>
    struct rebind {
        typedef DefaultInitAlloc<U> other;
    };
    template <typename U, typename... Args>
    void construct(U *p, Args &&...args) {
        if constexpr (sizeof...(Args) == 0)
            ::new ((void *)p) U;
        else
            ::new ((void *)p) U(std::forward<Args>(args)...);
    }
};
typedef std::vector<int32_t, DefaultInitAlloc<int32_t>> hvec32;

template <typename T>
struct DArr { // device array with RAII
    T *d        = nullptr;
    size_t n    = 0;
    size_t cap_ = 0; // bytes actually owned (>= n * sizeof(T) when the buffer came from the cache)
    int dev_    = 0; // device the buffer lives on (current device at alloc time)
    bool plain_ = false; // never from a reserved slab (buffers handed to RCCL: their allocation is what peers map)
    DArr() {}
    DArr(const DArr &)            = delete;
    DArr &operator=(const DArr &) = delete;
    ~DArr() { release(); }
    void release() {
        if (d && DeviceSlabs::get().owns(d)) { // a range of a reserved slab: reusable once nothing on the device can still touch it
            int cur = dev_;
            (void)hipGetDevice(&cur);
            if (cur != dev_)
                (void)hipSetDevice(dev_);
            const hipError_t se = hipDeviceSynchronize();
            if (cur != dev_)
                (void)hipSetDevice(cur);
            if (se == hipSuccess)
                (void)DeviceSlabs::get().give_back(d, cap_);
            else // e.g. a stream capture in progress: what cannot be proven idle is not handed out again (the range stays taken)
                (void)hipGetLastError();
        } else if (d && !DeviceCache::get().park(d, cap_, dev_)) {
            (void)hipFree(d);
        }
        d    = nullptr;
        n    = 0;
        cap_ = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0)
            return hipSuccess;
        g_alloc_count++;
        const size_t bytes = count * sizeof(T);
        size_t got         = 0;
        (void)hipGetDevice(&dev_);
        if (void *p = DeviceCache::get().take(bytes, &got)) {
            d    = static_cast<T *>(p);
            cap_ = got;
            return hipSuccess;
        }
        if (bytes >= (size_t(1) << 20) && !plain_)
            if (void *p = DeviceSlabs::get().take(dev_, bytes, &got)) {
                d    = static_cast<T *>(p);
                cap_ = got;
                return hipSuccess;
            }
        cap_               = bytes;
        MallocTimer timer;
        const hipError_t e = hipMalloc((void **)&d, bytes);
        if (e != hipSuccess) { // out of memory: give the parked buffers and the slabs nothing lives in back and try once more
            DeviceCache::get().trim();
            (void)DeviceSlabs::get().release_idle();
            (void)hipGetLastError();
            return hipMalloc((void **)&d, bytes);
        }
        return e;
    }
    // from a reserved slab only, near the fraction `frac` of its extent; hipErrorOutOfMemory when no slab has room there
    hipError_t alloc_at(size_t count, double frac) {
        release();
        n = count;
        if (count == 0)
            return hipSuccess;
        (void)hipGetDevice(&dev_);
        size_t got = 0;
        if (void *p = DeviceSlabs::get().take_at(dev_, count * sizeof(T), frac, &got)) {
            g_alloc_count++;
            d    = static_cast<T *>(p);
            cap_ = got;
            return hipSuccess;
        }
        n = 0;
        return hipErrorOutOfMemory;
    }
    void swap(DArr &o) {
        std::swap(d, o.d);
        std::swap(n, o.n);
        std::swap(cap_, o.cap_);
        std::swap(dev_, o.dev_);
        std::swap(plain_, o.plain_);
    }
    template <typename A>
    hipError_t upload(const std::vector<T, A> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty())
            return e;
        return hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t zero() { return n ? hipMemset(d, 0, n * sizeof(T)) : hipSuccess; }
};

// free device memory as the library sees it: what the driver has left plus what is free inside the reserved slabs
static hipError_t hmx_mem_info(size_t *free_b, size_t *total_b) {
    const hipError_t e = hipMemGetInfo(free_b, total_b);
    if (e != hipSuccess)
        return e;
    int dev = 0;
    (void)hipGetDevice(&dev);
    *free_b += DeviceSlabs::get().free_bytes(dev);
    return hipSuccess;
}
// ... and the largest SINGLE array that can still be allocated: no allocation spans the driver's memory and a slab, or two holes of a
// slab, so budgets for one array (cross pool, scratch) are capped by this, not by the sum above
static hipError_t hmx_mem_largest(size_t *largest_b) {
    size_t free_b = 0, total_b = 0;
    const hipError_t e = hipMemGetInfo(&free_b, &total_b);
    if (e != hipSuccess)
        return e;
    int dev = 0;
    (void)hipGetDevice(&dev);
    *largest_b = std::max(free_b, DeviceSlabs::get().largest_hole(dev));
    return hipSuccess;
}

// ---- where the WRITTEN arrays of a product go --------------------------------------------------------------------------------------
// A streaming read loses 16-23 % of its rate to a write stream of 0.4-1.6 % of its bytes when both lie in the same third of the physical
// memory (HBM3E stacks of 12 dies: three groups per channel), and 7-10 % when they do not (tools/placement_rw.hip).  Which third a virtual
// address belongs to is the driver's business (a 64 GB slab is one third, or pieces of two or three), so it is MEASURED: the small arrays
// the sweeps write (the reduced coefficients a, partial sums, column sums) are tried at a few places of the reserved slab against a sample
// of the stream that is read while they are written, and stay where the pair runs fastest.  Without a reserved slab (hmx_device_reserve)
// nothing is tried: the driver decides.
// Round 6: probes run in builds only (hmx_hmatrix_compress / finalize / recompress), on a stream of their own that waits for what the build
// has queued -- never inside a product call, never on the null stream.  A build leaves the winning place of the arrays written next to its
// E- and R-streams in the operator (HMat::place_e / place_r); arrays a product allocates later (work areas of the multi-RHS and host-memory
// entry points) go there without measuring anything (place_like).  What was measured is remembered per slab and stream position
// (PlacementCache): an operator rebuilt at the same place -- parameter sweeps, bench.py's second build -- launches no probe at all.
double placement_probe(const void *stream, size_t stream_bytes, void *cand, size_t cand_bytes, hipStream_t st, hipEvent_t after); // engine.hip
struct PlacementReport {
    double read_only = 0, first = 0, chosen = 0; // GB/s of the probe: the stream alone, with the array where first fit put it, where it stays
    int tried = 0;                               // candidates probed by THIS call (0: nothing measured -- no slab, disabled, or a cache hit)
    double frac = -1;                            // where the array stays: fraction of the slab's extent handed to take_at, -1 = where first fit put it
    bool cached = false;
    bool known  = false;                         // measured now or before: `frac` means something
};
struct PlacementCache {
    struct Key {
        uintptr_t stream_gb;
        size_t bytes_gb;
        bool operator<(const Key &o) const { return stream_gb != o.stream_gb ? stream_gb < o.stream_gb : bytes_gb < o.bytes_gb; }
    };
    std::mutex mu;
    std::map<Key, PlacementReport> seen;
    long long generation = -1; // DeviceSlabs::generation the entries belong to
    void sync_generation() {   // (under mu) a slab came or went: the same virtual addresses may now be other memory
        const long long g = DeviceSlabs::get().generation.load();
        if (g != generation) {
            seen.clear();
            generation = g;
        }
    }
    static PlacementCache &get() {
        static PlacementCache c;
        return c;
    }
    static Key key(const void *stream, size_t bytes) { return Key{(uintptr_t)stream >> 28, bytes >> 28}; } // 256 MiB: the classes come in pieces of >= 8 GiB
    bool find(const void *stream, size_t bytes, PlacementReport *r) {
        std::lock_guard<std::mutex> lock(mu);
        sync_generation();
        auto it = seen.find(key(stream, bytes));
        if (it == seen.end())
            return false;
        *r = it->second;
        return true;
    }
    void store(const void *stream, size_t bytes, const PlacementReport &r) {
        std::lock_guard<std::mutex> lock(mu);
        sync_generation();
        seen[key(stream, bytes)] = r;
    }
};
// the array at the place `frac` of a slab (-1 or no room there: wherever first fit puts it), zero-filled on `st`; measures nothing
template <typename T>
static hipError_t place_like(DArr<T> &arr, size_t count, double frac, hipStream_t st) {
    hipError_t e = hipErrorOutOfMemory;
    if (frac >= 0 && count * sizeof(T) >= (size_t(1) << 20))
        e = arr.alloc_at(count, frac);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = arr.alloc(count);
    }
    if (e != hipSuccess || count == 0)
        return e;
    return hipMemsetAsync(arr.d, 0, count * sizeof(T), st);
}
// `after`: an event recorded behind the work the probe must not overlap with (a build's pack kernels on the null stream); the probes run on a
// stream of their own.  The array is zero-filled afterwards (the probe writes sums of stream bytes into its candidates).
template <typename T>
static hipError_t place_written(DArr<T> &arr, size_t count, const void *stream, size_t stream_bytes, bool enabled, PlacementReport *rep = nullptr, hipEvent_t after = nullptr) {
    PlacementReport r;
    const size_t bytes = count * sizeof(T);
    const bool eligible = enabled && count > 0 && stream && stream_bytes >= (size_t(256) << 20) && bytes >= (size_t(1) << 20) && DeviceSlabs::get().owns(stream);
    if (eligible && PlacementCache::get().find(stream, stream_bytes, &r)) { // measured before for a stream at this place: no probe
        r.tried  = 0;
        r.cached = true;
        r.known  = true;
        if (rep)
            *rep = r;
        return place_like(arr, count, r.frac, nullptr);
    }
    const hipError_t e0 = arr.alloc(count);
    if (e0 != hipSuccess || !eligible || !DeviceSlabs::get().owns(arr.d)) {
        if (e0 == hipSuccess && count > 0)
            (void)hipMemsetAsync(arr.d, 0, bytes, nullptr);
        return e0;
    }
    hipStream_t ps = nullptr;
    if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return hipMemsetAsync(arr.d, 0, bytes, nullptr);
    }
    r.read_only = placement_probe(stream, stream_bytes, nullptr, 0, ps, after);
    r.first = r.chosen = placement_probe(stream, stream_bytes, arr.d, bytes, ps, nullptr);
    r.tried = 1;
    // different thirds: 0.90-0.93 of the read alone at 1.6 % written (0.85-0.88 on operators' streams); the same third: 0.75-0.80.  The search
    // ends with the first place that is out of the stream's third: another one would not be better
    auto good = [&] { return r.chosen >= 0.85 * r.read_only || r.chosen >= 1.10 * r.first; };
    if (r.first > 0 && !good())
        for (double frac : {1.0, 0.5, 0.75, 0.25, 0.0, 0.875, 0.625, 0.375, 0.125}) {
            DArr<T> cand;
            if (cand.alloc_at(count, frac) != hipSuccess)
                continue;
            const double rate = placement_probe(stream, stream_bytes, cand.d, bytes, ps, nullptr);
            r.tried++;
            if (rate > 1.03 * r.chosen) {
                (void)hipStreamSynchronize(ps);
                arr.swap(cand); // (the loser goes back to the slab when `cand` leaves the scope)
                r.chosen = rate;
                r.frac   = frac;
            }
            if (good())
                break;
        }
    (void)hipMemsetAsync(arr.d, 0, bytes, ps);
    (void)hipStreamSynchronize(ps);
    (void)hipStreamDestroy(ps);
    r.known = r.first > 0;
    if (r.known)
        PlacementCache::get().store(stream, stream_bytes, r);
    if (rep)
        *rep = r;
    return hipSuccess;
}

struct DEvent { // hipEvent_t with RAII, so error returns between create and destroy do not leak it
    hipEvent_t e = nullptr;
    DEvent() { (void)hipEventCreate(&e); }
    DEvent(const DEvent &)            = delete;
    DEvent &operator=(const DEvent &) = delete;
    ~DEvent() {
        if (e)
            (void)hipEventDestroy(e);
    }
    operator hipEvent_t() const { return e; }
};

// ---- per-operator options (include/hmx.h: hmx_option) ------------------------------------------------------------------------------
// One table: id, environment variable that gives the INITIAL value when an operator is created, default, valid range, when it may change.
// Nothing outside Options::from_environment() reads these variables: builds and products look at the operator's own values.
enum OptWhen { OPT_LAYOUT, OPT_BUILD, OPT_PRODUCT };
struct OptionSpec {
    int id;
    const char *env;
    bool env_inverted; // HMX_NO_*: the variable switches the feature OFF
    double def, lo, hi;
    OptWhen when;
};
static const OptionSpec HMX_OPTION_SPECS[] = {
    {HMX_OPT_R_PIECE_ROWS, "HMX_SR_MAX", false, 512, 64, 1 << 20, OPT_LAYOUT},
    {HMX_OPT_R_TREE_PIECES, "HMX_R_TREE_PIECES", false, 1, 0, 1, OPT_LAYOUT},
    {HMX_OPT_LAYOUT_THREADS, "HMX_LAYOUT_THREADS", false, 0, 0, 1024, OPT_LAYOUT},
    {HMX_OPT_TASK_ORDER, "HMX_SORT_TASKS", false, 1, 0, 3, OPT_LAYOUT},
    {HMX_OPT_XCD_UNIT_ROWS, "HMX_XCD_UNIT_ROWS", false, 512, 64, 1 << 20, OPT_LAYOUT},
    {HMX_OPT_SYM_STORAGE, "HMX_SYM_EXPANDED", false, 0, 0, 1, OPT_LAYOUT},
    {HMX_OPT_SYM_GROUP, "HMX_SYM_GROUP", false, 4, 1, 16, OPT_LAYOUT},
    {HMX_OPT_SYM_GROUP_SLOTS, "HMX_SYM_GROUP_SLOTS", false, -1, -1, 1024, OPT_LAYOUT},
    {HMX_OPT_BUILD_TIMING, "HMX_BUILD_TIMING", false, 0, 0, 1, OPT_BUILD},
    {HMX_OPT_REDUCE_WAVES, "HMX_REDUCE_WAVES", false, 0, 0, 8, OPT_PRODUCT},
    {HMX_OPT_EXPAND_WAVES, "HMX_EXPAND_WAVES", false, 0, 0, 8, OPT_PRODUCT},
    {HMX_OPT_MULTI_RHS_FUSED, "HMX_NO_FUSED_MU", true, 1, 0, 1, OPT_PRODUCT},
    {HMX_OPT_MATRIX_CORES, "HMX_NO_MFMA", true, 1, 0, 1, OPT_PRODUCT},
    {HMX_OPT_MATRIX_CORES_F32, "HMX_MFMA_F32", false, 1, 0, 1, OPT_PRODUCT},
    {HMX_OPT_WIDE_SWEEPS, "HMX_MFMA_WIDE", false, 1, 0, 1, OPT_PRODUCT},
    {HMX_OPT_SCALAR_OPERANDS, "HMX_MU_SCALAR", false, -1, -1, 1, OPT_PRODUCT},
    {HMX_OPT_SYM_MULTI_RHS, "HMX_SYM_MU_FUSED", false, -1, -1, 1, OPT_PRODUCT},
    {HMX_OPT_SYM_NO_VIEW, "HMX_SYM_NO_VIEW", false, 0, 0, 1, OPT_PRODUCT},
    {HMX_OPT_TRANSPOSED_LAYOUT, "HMX_TRANS_STREAMS", false, -1, -1, 1, OPT_PRODUCT},
    {HMX_OPT_CALLBACK_THREADS, "HMX_CALLBACK_THREADS", false, 0, 0, 256, OPT_BUILD},
    {HMX_OPT_CALLBACK_DRIVERS, "HMX_CALLBACK_DRIVERS", false, 8, 1, 64, OPT_BUILD},
    {HMX_OPT_POOL_SAMPLE, "HMX_POOL_SAMPLE", false, 1, 0, 1, OPT_BUILD},
    {HMX_OPT_POOL_RANK_GUESS, "HMX_POOL_RANK_GUESS", false, 0, 0, 1e6, OPT_BUILD},
    {HMX_OPT_ACA_TEAMS, "HMX_ACA_TEAM", false, 1, 0, 1, OPT_BUILD},
    {HMX_OPT_ACA_TEAM_MIN, "HMX_ACA_TEAM_MIN", false, 4096, 2, 1 << 30, OPT_BUILD},
    {HMX_OPT_ACA_TEAM_AFTER, "HMX_ACA_TEAM_Q", false, 48, 1, 1 << 20, OPT_BUILD},
    {HMX_OPT_ACA_TEAM_SLICE, "HMX_ACA_TEAM_SLICE", false, 0, 0, 1 << 20, OPT_BUILD},
    {HMX_OPT_PLACE_WRITTEN, "HMX_PLACE_WRITTEN", false, 1, 0, 1, OPT_PRODUCT},
    {HMX_OPT_ACA_WAVE_MAX, "HMX_ACA_WAVE_MAX", false, 256, 0, 256, OPT_BUILD},
};
struct Options {
    static constexpr int MAX_ID = 40;
    double v[MAX_ID];
    static const OptionSpec *spec(int id) {
        for (const OptionSpec &s : HMX_OPTION_SPECS)
            if (s.id == id)
                return &s;
        return nullptr;
    }
    Options() {
        for (double &x : v)
            x = 0;
        for (const OptionSpec &s : HMX_OPTION_SPECS)
            v[s.id] = s.def;
    }
    // the process environment as the initial values of a NEW operator (A/B runs of unmodified programs); out-of-range values are clamped
    static Options from_environment() {
        Options o;
        for (const OptionSpec &s : HMX_OPTION_SPECS)
            if (const char *e = getenv(s.env)) {
                double x = atof(e);
                if (s.env_inverted)
                    x = x != 0 ? 0 : 1;
                o.v[s.id] = std::min(s.hi, std::max(s.lo, x));
            }
        return o;
    }
    int i(int id) const { return (int)v[id]; }
    double d(int id) const { return v[id]; }
};

// One family of streams (E = expand over target ranges, R = reduce over source ranges)
enum LeafKind { LK_PENDING = 0, LK_DENSE_GEN = 1, LK_DENSE_STAGED = 2, LK_LOWRANK = 3 };


static int ensure_device(int device) {
    int count     = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available: libhmx has no CPU path (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ")");
        return HMX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device id out of range");
        return HMX_ERR_INVALID;
    }
    HMX_HIP(hipSetDevice(device));
    return HMX_OK;
}

// split [lo,hi) at the sorted breakpoints, then cut pieces longer than maxlen evenly
static void make_ranges(std::vector<int> &bp, int maxlen, int origin, std::vector<int32_t> &off, std::vector<int32_t> &len) {
    std::sort(bp.begin(), bp.end());
    bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
    off.clear();
    len.clear();
    for (size_t k = 0; k + 1 < bp.size(); k++) {
        const int a = bp[k], L = bp[k + 1] - bp[k];
        const int pieces = (L + maxlen - 1) / maxlen;
        for (int p = 0; p < pieces; p++) {
            const int s = a + (int)((int64_t)L * p / pieces), e = a + (int)((int64_t)L * (p + 1) / pieces);
            off.push_back(s - origin);
            len.push_back(e - s);
        }
    }
}

// host loops over millions of (leaf, range) pairs with disjoint outputs: split over a few threads
template <typename F>
static void parallel_for(size_t n, F &&body) {
    const size_t nt = std::min<size_t>({(size_t)16, (size_t)host_cores(), n / 65536 + 1});
    if (nt <= 1) {
        body((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++)
        th.emplace_back([&, t] { body(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th)
        x.join();
}

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double s() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- binary dump of the compressed operator (no counterpart in the reference) ---------------------------
// Layout: HmxFileHeader, hmx_leaf[nleaves] (ranks filled in), then per leaf in htool's leaf order either
// U (M x r, column-major) followed by V (r x N, column-major) -- LowRankMatrix' own layout (lrmat.hpp:15-128) -- or the
// dense block (M x N, column-major).  Loading goes through set_block_* + finalize, i.e. the upload path.
struct HmxFileHeader {
    char magic[8];
    int32_t elem_size, reserved;
    int64_t nleaves;
    int32_t T0, nT, S0, nS;
    int32_t symmetry, uplo;
    double epsilon;
};
static const char HMX_FILE_MAGIC[8] = {'H', 'M', 'X', 'B', 'I', 'N', '1', '\0'};

typedef void (*after_chunk_fn)(void *user, int chunk, int row_lo, int row_hi);

} // namespace hmx
