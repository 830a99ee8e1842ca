// Host side of libenzymm_amd.so: packs templates and structures into their HBM layouts,
// plans the per-template search order, launches the kernels and returns the hit table.
// C ABI declared in include/enzymm_amd.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <tuple>
#include <vector>

#include "tm_kernels.hip"  // single translation unit: kernels + host API
#include "tm_pdb.hip"
#include "tm_ingest.hip"

#include <emmintrin.h>

using namespace tmx;

namespace {

thread_local std::string g_err;

// ENZYMM_AMD_TRACE=1: one stderr line per call with the host-side milliseconds between marks.
struct PhaseTrace {
    const char *label;
    bool on = std::getenv("ENZYMM_AMD_TRACE") != nullptr;
    explicit PhaseTrace(const char *l = "tm_query") : label(l) {}
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    std::string line;
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        line += std::string(what) + " " + std::to_string(std::chrono::duration<double, std::milli>(now - last).count()).substr(0, 6) + " | ";
        last = now;
    }
    ~PhaseTrace() {
        // (start: milliseconds since the first traced call of the process; thread: which host thread made the call -- enough
        // to lay the calls of a pipeline's threads side by side)
        static const std::chrono::steady_clock::time_point origin = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "%s host ms: %s total %.3f | start %.3f thread %lx\n", label, line.c_str(),
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                             std::chrono::duration<double, std::milli>(t0 - origin).count(), (unsigned long)pthread_self() & 0xffffff);
    }
};

// Host threads one native call may use: ENZYMM_AMD_HOST_THREADS, else the smallest of the hardware threads, the
// affinity mask, the container's CPU quota (cgroup v2 cpu.max; a box that shows 256 CPUs may grant 16) and 16, divided
// by LOCAL_WORLD_SIZE (the ranks of one node share its host).
// More threads than the quota grants are throttled in bursts, which shows as run-to-run variance.
// Atoms a slice of a batch-creation loop holds at least (ENZYMM_AMD_HOST_GRAIN): the loops run on a persistent team
// (HostPool), so a slice of a few thousand atoms -- 50 microseconds of work -- is worth handing out.
uint32_t host_grain() {
    static const uint32_t grain = [] {
        if (const char *e = std::getenv("ENZYMM_AMD_HOST_GRAIN")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v >= 256) return (uint32_t)v;
        }
        return 4096u;
    }();
    return grain;
}

uint32_t host_team() {
    static const uint32_t team = [] {
        if (const char *e = std::getenv("ENZYMM_AMD_HOST_THREADS")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v >= 1) return (uint32_t)std::min<long>(v, 256);
        }
        uint32_t n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<uint32_t>(n, std::max(1, CPU_COUNT(&set)));
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
                const long q = std::strtol(quota, nullptr, 10);
                if (q > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long>(1, q / period));
            }
            std::fclose(f);
        } else {  // cgroup v1
            long q = -1, period = 0;
            if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(fq, "%ld", &q) != 1) q = -1;
                std::fclose(fq);
            }
            if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(fp, "%ld", &period) != 1) period = 0;
                std::fclose(fp);
            }
            if (q > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long>(1, q / period));
        }
        // one process per GPU on one node (torch.distributed.run exports LOCAL_WORLD_SIZE): the ranks share the host, each
        // takes its share of the quota instead of sizing its team as if it were alone (8 ranks x 16 threads on 16 cores)
        if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) {
            const long ranks = std::strtol(e, nullptr, 10);
            if (ranks > 1) n = std::max<uint32_t>(1u, n / (uint32_t)std::min<long>(ranks, 1024));
        }
        return std::min(n, 16u);
    }();
    return team;
}

// A persistent team of host threads for the short parallel loops of batch creation (encode, flatten, file reads): a
// std::thread per slice costs about as much to start as a slice of a sixteen-molecule batch takes to run.  One loop at
// a time: a caller that finds the team busy (another Python thread creating a batch) runs its loop inline.  After a
// fork() the child starts its own team on first use (the parent's threads do not exist there).
class HostPool {
public:
    // kTeams teams: a loop takes the first team that is free (the ingest workers of Matcher.run_files and the leaders of
    // pyjess's launches create batches side by side); all busy: the caller runs its loop inline
    static constexpr int kTeams = 3;
    static HostPool &get(int team = 0) {
        static std::once_flag once;
        std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { for (int t = 0; t < kTeams; ++t) instance(t).store(nullptr); }); });
        HostPool *p = instance(team).load();
        if (!p) {
            static std::mutex create;
            std::lock_guard<std::mutex> lock(create);
            p = instance(team).load();
            if (!p) { p = new HostPool(host_team() > 1 ? host_team() - 1 : 0); instance(team).store(p); }
        }
        return *p;
    }
    // f(i) for i in [0, n): on a free team and the calling thread
    static void run_anywhere(uint32_t n, const std::function<void(uint32_t)> &f) {
        if (n > 1)
            for (int t = 0; t < kTeams; ++t)
                if (get(t).run(n, f, false)) return;
        for (uint32_t i = 0; i < n; ++i) f(i);
    }
    // the same on THIS team; false (nothing done) when it is busy and `inline_if_busy` is not set
    bool run(uint32_t n, const std::function<void(uint32_t)> &f, bool inline_if_busy = true) {
        std::unique_lock<std::mutex> one(run_mu_, std::try_to_lock);
        if (n <= 1 || workers_.empty() || !one.owns_lock()) {
            if (!inline_if_busy && n > 1 && !workers_.empty()) return false;
            for (uint32_t i = 0; i < n; ++i) f(i);
            return true;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            job_ = &f; n_ = n; next_.store(0); left_ = n; ++generation_;
        }
        work_.notify_all();
        pull();
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [&] { return left_ == 0 && active_ == 0; });  // (no worker is still looking at this job)
        job_ = nullptr;
        return true;
    }

private:
    explicit HostPool(uint32_t n_threads) {
        for (uint32_t t = 0; t < n_threads; ++t) workers_.emplace_back([this] { loop(); }).detach();
    }
    static std::atomic<HostPool *> &instance(int team) { static std::atomic<HostPool *> p[kTeams]; return p[team]; }
    void pull() {
        uint32_t mine = 0;
        for (;;) {
            const uint32_t i = next_.fetch_add(1);
            if (i >= n_) break;
            (*job_)(i);
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> lock(mu_);
            left_ -= mine;
            if (left_ == 0 && active_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                work_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
                if (!job_) continue;
                ++active_;
            }
            pull();
            {
                std::lock_guard<std::mutex> lock(mu_);
                --active_;
                if (left_ == 0 && active_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable work_, done_;
    const std::function<void(uint32_t)> *job_ = nullptr;
    uint32_t n_ = 0, left_ = 0, active_ = 0;
    std::atomic<uint32_t> next_{0};
    uint64_t generation_ = 0;
};

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return fail(TM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Waits for `stream` without spinning: an event with hipEventBlockingSync behind what is queued, one per host thread
// and device.  The ingest threads wait for uploads of a few milliseconds several times per batch; a spinning wait
// (hipStreamSynchronize) holds a core of the container's CPU quota each, which the threads reading the next files want.
// ENZYMM_AMD_SPIN_WAIT=1: the spinning wait (A/B).
hipError_t wait_blocking(hipStream_t stream) {
    static const bool spin = [] { const char *e = std::getenv("ENZYMM_AMD_SPIN_WAIT"); return e && e[0] == '1'; }();
    if (spin) return hipStreamSynchronize(stream);
    struct Events {
        std::map<int, hipEvent_t> by_device;
        ~Events() { for (auto &kv : by_device) if (hipSetDevice(kv.first) == hipSuccess) (void)hipEventDestroy(kv.second); }
    };
    thread_local Events events;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    auto it = events.by_device.find(device);
    if (it == events.by_device.end()) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess) return e;
        it = events.by_device.emplace(device, ev).first;
    }
    e = hipEventRecord(it->second, stream);
    return e != hipSuccess ? e : hipEventSynchronize(it->second);
}

// Device memory is recycled through a process-wide cache: a query allocates a dozen buffers, and
// hipMalloc / hipFree (which synchronises the device) would otherwise cost more than a millisecond
// per call.  Blocks are rounded up to size classes and handed back on release; a block is only
// released once the work that used it has completed (tm_result_destroy waits for the stream,
// tm_batch_destroy / tm_library_destroy for the device).
class DevicePool {
  public:
    static DevicePool &get() {
        static DevicePool *pool = new DevicePool;  // never destroyed: the HIP runtime may be gone at exit
        return *pool;
    }
    static size_t size_class(size_t bytes) {
        size_t c = 256;
        while (c < bytes && c < (size_t(1) << 20)) c <<= 1;
        if (c >= bytes) return c;
        // above 1 MiB: eight classes per octave (at most 12.5 % over), so that batches of about the same size --
        // whose pools differ by a few per cent -- recycle each other's blocks instead of going to hipMalloc
        size_t top = size_t(1) << 20;
        while ((top << 1) <= bytes) top <<= 1;
        const size_t step = top >> 3;
        return (bytes + step - 1) / step * step;
    }
    hipError_t acquire(size_t bytes, void **p, size_t *granted) {
        int device = 0;
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess) return e;
        const size_t c = size_class(bytes);
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = free_.lower_bound({device, c});  // the smallest cached block of this device that fits, within a quarter
            if (it != free_.end() && it->first.first == device && it->first.second <= c + c / 4) {
                *p = it->second;
                *granted = it->first.second;
                cached_ -= it->first.second;
                free_.erase(it);
                return hipSuccess;
            }
        }
        e = hipMalloc(p, c);
        if (e != hipSuccess) {  // give cached blocks back to the driver and retry once
            trim(device);
            e = hipMalloc(p, c);
        }
        *granted = c;
        return e;
    }
    void release(void *p, size_t c) {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return;
        std::lock_guard<std::mutex> lock(mu_);
        if (cached_ + c > kMaxCached) {
            (void)hipFree(p);
            return;
        }
        free_.insert({{device, c}, p});
        cached_ += c;
    }
    void trim(int device) {
        std::lock_guard<std::mutex> lock(mu_);
        for (auto it = free_.begin(); it != free_.end();) {
            if (it->first.first == device) {
                (void)hipFree(it->second);
                cached_ -= it->first.second;
                it = free_.erase(it);
            } else {
                ++it;
            }
        }
    }

  private:
    // Round 6: 48 GB of 288 GB HBM (ENZYMM_AMD_DEVICE_CACHE_GB; 16 until then).  A batch of a thousand structures holds
    // 2.8 GB while it lives -- 2.2 GB of it the generously sized pair-table pool -- and Matcher.run keeps six alive (three
    // handed over ahead, two searched, one collected): with 16 GB every second batch's release went to hipFree, which
    // waits for the device, and its successor's acquire to hipMalloc (tm_query's host time 2.2 ms instead of 0.55).
    const size_t kMaxCached = [] {
        const char *e = std::getenv("ENZYMM_AMD_DEVICE_CACHE_GB");
        const long gb = e ? std::atol(e) : 48;
        return (size_t)std::max<long>(1, std::min<long>(gb, 256)) << 30;
    }();
    std::mutex mu_;
    std::multimap<std::pair<int, size_t>, void *> free_;
    size_t cached_ = 0;
};

// Owning device buffer (memory from DevicePool; the current device must be the owning one on release).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t bytes = 0;  // granted size class
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) DevicePool::get().release(p, bytes);
        p = nullptr;
        n = 0;
        bytes = 0;
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(bytes, o.bytes); }
    hipError_t alloc(size_t count) {
        const size_t need = std::max<size_t>(count, 1) * sizeof(T);
        if (p && bytes >= need && bytes <= 2 * DevicePool::size_class(need)) {  // keep a block that still fits
            n = count;
            return hipSuccess;
        }
        release();
        n = count;
        return DevicePool::get().acquire(need, reinterpret_cast<void **>(&p), &bytes);
    }
    hipError_t upload_async(const T *src, size_t count, hipStream_t stream) {  // caller synchronises the stream
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, stream);
    }
    // With a stream the copy is ordered on (and waited for on) that stream only; a plain hipMemcpy goes through
    // the null stream and would wait for whatever other blocking streams are running.
    hipError_t upload(const std::vector<T> &v, hipStream_t stream = nullptr) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        if (!stream) return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        e = hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream);
        return e != hipSuccess ? e : wait_blocking(stream);
    }
};

// Page-locked host staging memory, recycled like device memory: tm_batch_create flattens >100 MB per thousand
// structures, and fresh pageable vectors cost more in page faults and staged DMA than the flattening itself.
class PinnedPool {
  public:
    static PinnedPool &get() {
        static PinnedPool *pool = new PinnedPool;  // never destroyed: the HIP runtime may be gone at exit
        return *pool;
    }
    hipError_t acquire(size_t bytes, void **p, size_t *granted) {
        const size_t c = DevicePool::size_class(bytes);
        {
            // the smallest cached block that holds the request and is not more than half as large again: batches of a
            // thousand files differ by a few per cent in size, and a miss is a hipHostMalloc -- 8 ms for 190 MB, measured
            // in every second tm_batch_create_from_files while blocks were matched by their exact size class
            std::lock_guard<std::mutex> lock(mu_);
            auto it = free_.lower_bound(c);
            if (it != free_.end() && it->first <= c + c / 2) {
                *p = it->second;
                *granted = it->first;
                cached_ -= it->first;
                free_.erase(it);
                return hipSuccess;
            }
        }
        *granted = c;
        return hipHostMalloc(p, c, hipHostMallocDefault);
    }
    void release(void *p, size_t c) {
        std::lock_guard<std::mutex> lock(mu_);
        if (cached_ + c > kMaxCached) {
            (void)hipHostFree(p);
            return;
        }
        free_.insert({c, p});
        cached_ += c;
    }

  private:
    // (ENZYMM_AMD_PINNED_CACHE_GB, default 6, 2 until round 6: Matcher.run_files keeps the text of the batches it has in
    // flight -- 190 MB each -- in page-locked memory, and a release beyond the cap is a hipHostFree, its successor a
    // hipHostMalloc of 8 ms)
    const size_t kMaxCached = [] {
        const char *e = std::getenv("ENZYMM_AMD_PINNED_CACHE_GB");
        const long gb = e ? std::atol(e) : 6;
        return (size_t)std::max<long>(1, std::min<long>(gb, 64)) << 30;
    }();
    std::mutex mu_;
    std::multimap<size_t, void *> free_;
    size_t cached_ = 0;
};

// Owning page-locked host array (contents uninitialised).
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0, bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) PinnedPool::get().release(p, bytes);
        p = nullptr; n = 0; bytes = 0;
    }
    hipError_t alloc(size_t count) {
        const size_t need = std::max<size_t>(count, 1) * sizeof(T);
        n = count;
        if (p && bytes >= need) return hipSuccess;  // still fits
        if (p) PinnedPool::get().release(p, bytes);
        p = nullptr;
        return PinnedPool::get().acquire(need, reinterpret_cast<void **>(&p), &bytes);
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// compute units of a device, looked up once per device
int device_cus(int device) {
    static std::mutex mu;
    static std::map<int, int> known;
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) n = 0;
    known[device] = n;
    return n;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a ceiling: it is raised when a launch needs more dynamic LDS than any
// launch of that kernel on that device before it, not set again with every query (the call costs as much as a launch).
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[{kernel, device}];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// A template selection with its thresholds, staged for the search kernel (k_stage_templates).
// Batches are normally queried with the same selection over and over: the library keeps the last
// one and tm_query reuses it when the arguments are identical.
struct StagedSelection {
    std::vector<uint32_t> sel;
    std::vector<double> rmsd, cut, maxdyn;
    double slack = kPrefilterSlack;  // fp32 prefilter slack the windows and table classes were staged with
    DevBuf<uint32_t> d_sel;
    DevBuf<double> d_rmsd, d_cut, d_maxdyn;
    DevBuf<float2> d_bounds;
    DevBuf<Staged> d_staged;
    DevBuf<BlockHdr> d_hdr;      // per staged template: the record of the table-driven stages (k_search reads 64 at a time)
    DevBuf<LookHdr> d_look;      // per staged template: the record of the look at search level 3
    DevBuf<ResCheck> d_rescheck; // per staged template: the records of stage D0 (kResChecks slots each)
    DevBuf<double2> d_exact;     // per staged template and pair: fp64 (template distance, threshold) for the re-check
    // Directory of the pages of triples waiting for k_deep: a structure of n atoms owns one entry per wave that works on
    // it (the partly filled last page of each) + dir_milli * n^2 / 1e7 entries for its full pages.  Starts at 0.1 page
    // per 1e4 atoms^2 (a 300-residue structure under the default thresholds fills about twenty pages); raised to what a
    // query that overflowed needed.
    std::atomic<unsigned long long> dir_milli{100};
    DevBuf<uint16_t> d_lead8;    // per staged template: lead-kind index of each of its residues (0xffff: none)
    // staged positions the template-at-a-time launch looks at: templates of fewer than three residues, and (for batches
    // with a structure large enough for a pair table of 2^25 entries) all of them
    std::vector<uint32_t> small;
    DevBuf<uint32_t> d_small, d_all;
    // The device works on the selection sorted by descending estimated cost (every workgroup's dynamic queue
    // then starts with the expensive templates and ends on the cheap ones); order[pos] is the caller's index.
    std::vector<uint32_t> order;
    DevBuf<uint32_t> d_order;
    // What queries with this selection have needed so far, per 1000 atoms of their batches: candidates and
    // (template, structure) pairs with candidates.  tm_query sizes a result's tables for 1.5 x that when the caller
    // leaves the capacities to the library -- loose thresholds (enzymm's -j) produce orders of magnitude more
    // candidates than the defaults allow for, and an overflow costs a whole second run.
    std::atomic<unsigned long long> cands_per_katom{0}, pairs_per_katom{0};
    // The logistic model tables of the hit epilogue as last uploaded for this selection (enzymm passes the same ones
    // with every query): a query whose tables are identical uses these instead of four blocking uploads of its own,
    // each of which would wait for the index and table build it has just enqueued.
    std::mutex models_mu;
    std::vector<int32_t> h_model_set;  // CALLER order (d_model_set is in staged order: h_model_set[order[i]])
    std::vector<uint32_t> h_model_offsets, h_model_votes;
    std::vector<double> h_models;
    DevBuf<int32_t> d_model_set;
    DevBuf<uint32_t> d_model_offsets, d_model_votes;
    DevBuf<double> d_models;
    bool models_cached = false;
    // pair tables the selection's templates read (see k_tab_* in tm_kernels.hip) and, per staged template, which ones
    std::vector<TableDesc> tables;
    DevBuf<TableDesc> d_tables;
    DevBuf<uint16_t> d_pair_table, d_triple_table;
    DevBuf<uint8_t> d_triple_key;
    uint64_t serial = 0;  // identity for the batch-side table cache
    // Order of the selection on the device: first by an a-priori estimate; the first query with the selection also
    // clocks every template's searches (Q.cost), and when that result has been waited for the selection is staged
    // again in descending MEASURED cost -- a workgroup's queue then ends on its cheapest searches, which is what
    // keeps the other fifteen waves from idling behind one long search (results do not depend on the order).
    DevBuf<unsigned long long> d_cost;
    std::vector<float> measured;           // per caller index, valid in state 2
    std::atomic<int> calibration{0};       // 0 not measured, 1 measuring query in flight, 2 measured, 3 staged by measured cost
    // mean atoms per structure of the batch the costs were measured on (the relative cost of templates moves with the
    // size of the structures: a selection calibrated on 300-residue chains is measured again when it meets batches
    // whose mean differs by more than a quarter)
    std::atomic<uint32_t> calibrated_on{0}, calibrating_on{0};
    hipEvent_t ready = nullptr;  // recorded after k_stage_templates
    ~StagedSelection() {
        if (ready) (void)hipEventDestroy(ready);
    }
};

}  // namespace

struct tm_library {
    mutable std::mutex sel_mu;
    // staged selections of this library, most recently used first.  enzymm queries a library with ONE parameter set
    // per size bucket, but a caller that alternates between a few (per-size queries through one Jess, a threshold
    // sweep) would otherwise stage 6780 templates and plan their tables again with every call.
    static constexpr size_t kSelectionsKept = 8;
    mutable std::vector<std::shared_ptr<StagedSelection>> selections;
    int device = 0;
    uint32_t n_templates = 0, n_atoms = 0, n_kinds = 0, n_lead = 0;
    std::vector<uint32_t> t_off;
    DevBuf<uint32_t> d_t_off, d_k_resmask, d_lead_kind, d_r_off;
    DevBuf<double> d_xyz, d_w, d_r_vec;
    DevBuf<uint16_t> d_kind;
    DevBuf<uint8_t> d_group, d_orig, d_chain, d_k_any, d_ref, d_r_first, d_r_second;
    bool has_orient = false;
    DevBuf<uint64_t> d_k_namemask;
    DevBuf<int32_t> d_k_lead;
    uint64_t serial = 0;  // identity for the batch-side index cache
    // per template: expected size of the cross product of its two top lead lists (residue types x atom names
    // admitted by each) -- the estimate of search cost the selection is ordered by
    std::vector<float> cost_weight;
    // host copies in SEARCH order, for planning the pair tables of a selection: positions 0 .. n_groups - 1 of a
    // template are its lead atoms
    std::vector<double> h_xyz, h_w;
    std::vector<uint16_t> h_lead_li;   // per atom: lead-kind index (lead positions only)
    std::vector<uint8_t> h_n_groups;   // per template
    LibView view() const {
        LibView v;
        v.n_templates = n_templates; v.n_kinds = n_kinds; v.n_lead = n_lead;
        v.t_off = d_t_off.p; v.a_xyz = d_xyz.p; v.a_w = d_w.p; v.a_kind = d_kind.p;
        v.a_group = d_group.p; v.a_ref = d_ref.p; v.a_orig = d_orig.p; v.a_chain = d_chain.p;
        v.k_resmask = d_k_resmask.p; v.k_namemask = d_k_namemask.p; v.k_any = d_k_any.p;
        v.k_lead = d_k_lead.p; v.lead_kind = d_lead_kind.p;
        v.r_off = has_orient ? d_r_off.p : nullptr; v.r_first = d_r_first.p; v.r_second = d_r_second.p;
        v.r_vec = d_r_vec.p;
        return v;
    }
};

// What a query result may have to tell its batch after the fact (it can outlive the batch, hence shared).
struct IndexShared {
    std::atomic<unsigned long long> pool_wanted{0};  // kind-list entries an overflowed build needed
    std::atomic<unsigned long long> rows_wanted{0}, ents_wanted{0};  // pair-table row slots / entries, likewise
    std::atomic<bool> stale{false};                  // the index in the batch is incomplete: rebuild before use
};

struct tm_batch {
    int device = 0;
    uint32_t n_structs = 0, n_atoms = 0, max_atoms = 0, max_res = 0;
    // WIDE structures (more than TM_NARROW_STRUCTURE_ATOMS atoms, round 5): the upper bits of their atom indices live in
    // byte arrays beside the 16-bit ones (allocated only when the batch holds one); the LDS layouts of the tuned kernels
    // are sized for the largest NARROW structure -- the wide ones are searched by a launch of their own (tm_query)
    bool has_wide = false;
    uint32_t max_atoms_narrow = 0, max_res_narrow = 0;
    DevBuf<uint8_t> d_res_start_hi, d_kl_pool_hi;
    // structures beyond TM_MAX_STRUCTURE_ATOMS: kept in the batch as EMPTY structures (no hits), listed here with the
    // atom counts they came with (tm_batch_refused) -- one oversized file does not fail the batch it arrives in
    std::vector<uint32_t> refused, refused_atoms;
    float extent = 0.f;  // largest |centred coordinate| of any atom (A): sizes the fp32 prefilter slack of its queries
    std::vector<uint32_t> s_off;
    DevBuf<uint32_t> d_s_off, d_s_roff, d_orig, d_kl_off;
    DevBuf<double> d_xyz;
    DevBuf<float4> d_rec;
    DevBuf<uint8_t> d_chain;
    DevBuf<uint16_t> d_res_start, d_kl_pool;
    DevBuf<unsigned long long> d_cursor;
    DevBuf<uint32_t> d_overflow;
    DevBuf<uint32_t> d_line_off;  // device-parsed batches: byte offset of every atom record in its file
    // tm_batch_create_from_structures (round 5): the hand-over arena on both sides and the event behind k_flatten.  The
    // constructor returns when its copies are queued; queries wait for `ready` on their stream; the staging memory goes
    // back to its pools once the event has been seen complete (settle).
    PinnedBuf<uint8_t> h_stage, h_side;
    PinnedBuf<uint32_t> h_stage_off;
    DevBuf<uint8_t> d_stage, d_side;
    hipEvent_t ready = nullptr;
    uint64_t upload_bytes = 0;  // host-to-device bytes the constructor queued (tm_batch_upload_bytes)
    std::mutex ready_mutex;  // `ready` and the staging blocks: two threads may issue the first query of a batch
    void settle() {  // the upload and k_flatten are known to have finished: release what only they needed
        std::lock_guard<std::mutex> lock(ready_mutex);
        settle_locked();
    }
    void settle_locked() {
        h_stage.release(); h_side.release(); h_stage_off.release(); d_stage.release(); d_side.release();
        if (ready) { (void)hipEventDestroy(ready); ready = nullptr; }
    }
    ~tm_batch() { if (ready) (void)hipEventDestroy(ready); }
    PinnedBuf<uint32_t> h_line_off;  // device-parsed batches: d_line_off on the host (copied behind the parse; tm_batch_pick_records)
    PinnedBuf<char> h_text;       // tm_batch_create_from_files: the files' text, kept for tm_batch_file_text
    std::vector<uint64_t> h_text_off;
    DevBuf<uint32_t> d_residue_count;  // device-parsed batches with counted residue names: distinct residue numbers
    bool has_residue_count = false;
    // index build of long structures (more than kKindSeg atoms): their segments, made at the first build (build_index)
    DevBuf<KindSeg> d_kind_segs;
    PinnedBuf<KindSeg> h_kind_segs;
    DevBuf<KindLong> d_kind_longs;
    PinnedBuf<KindLong> h_kind_longs;
    DevBuf<uint32_t> d_kind_seg_cnt;  // [segments][lead kinds]: counts, then places
    DevBuf<uint32_t> d_kind_unsearchable;  // [long structures]: a kind list beyond 16-bit positions (k_kind_seg_offsets)
    uint32_t n_kind_segs = 0, n_kind_longs = 0;
    bool kind_segs_made = false;
    // search work items (structure, round), cached for the (cx, n_sel) they were planned for
    DevBuf<WorkItem> d_work;
    PinnedBuf<WorkItem> h_work;      // page-locked source of d_work: uploaded on the query's stream without a host wait
    uint32_t work_items = 0, work_cx = 0xffffffffu, work_n_sel = 0;
    hipStream_t last_stream = nullptr;  // stream of the last query that read this batch
    bool queried = false;
    uint64_t index_for = 0;  // serial of the library the candidate index was built for
    unsigned long long pool_capacity = 0;
    // pair tables of one staged selection (rebuilt with the candidate index, or when the selection changes)
    uint64_t tables_for = 0;  // serial of the selection
    uint32_t n_tables = 0;
    DevBuf<uint32_t> d_tab_dir, d_tab_rows, d_tab_ok, d_tab_overflow;
    DevBuf<uint2> d_tab_seg, d_tab_ent;
    DevBuf<uint4> d_tab_ref;
    DevBuf<unsigned long long> d_tab_cursors;  // [0] row slots, [1] entries handed out
    DevBuf<TabWork> d_tab_work;
    PinnedBuf<TabWork> h_tab_work;   // likewise
    uint32_t tab_work_items = 0, tab_work_tables = 0xffffffffu;
    uint32_t tab_work_n_lead = 0xffffffffu;
    uint32_t tab_work_small = 0, tab_work_long_from = 0xffffffffu;  // items [0, small): the batch's LDS layout; the rest: long structures
    // long structures: position records and chunk spheres for their table build (k_tab_gather), regions laid out by the host
    DevBuf<TabGather> d_tab_gather;
    PinnedBuf<TabGather> h_tab_gather;
    uint32_t n_tab_gather = 0;
    DevBuf<float4> d_tab_lr, d_tab_sph;
    DevBuf<uint32_t> d_tab_lr_base, d_tab_sph_base;
    unsigned long long row_capacity = 0, ent_capacity = 0;
    // The index build is not waited for: a query whose build overflowed the kind-list pool finds out in
    // tm_result_wait and leaves the size it needed here (shared with the results, which may outlive a batch);
    // the next build starts from it.
    std::shared_ptr<IndexShared> index_shared = std::make_shared<IndexShared>();
    BatchView view() const {
        BatchView v;
        v.n_structs = n_structs; v.s_off = d_s_off.p; v.s_roff = d_s_roff.p; v.q_xyz = d_xyz.p;
        v.q_rec = d_rec.p; v.q_chain = d_chain.p; v.q_orig = d_orig.p; v.res_start = d_res_start.p;
        v.kl_off = d_kl_off.p; v.kl_pool = d_kl_pool.p;
        v.kl_pool_hi = has_wide ? d_kl_pool_hi.p : nullptr; v.res_start_hi = has_wide ? d_res_start_hi.p : nullptr;
        v.n_tables = n_tables; v.tab_dir = d_tab_dir.p; v.tab_rows = d_tab_rows.p; v.tab_ent = d_tab_ent.p;
        v.tab_seg = d_tab_seg.p; v.tab_ok = d_tab_ok.p; v.tab_ref = d_tab_ref.p;
        return v;
    }
};

struct tm_result {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n_structs = 0, n_sel = 0;
    unsigned long long hit_capacity = 0, cand_capacity = 0;
    DevBuf<tm_hit> d_hits;
    DevBuf<Cand> d_cands;
    DevBuf<uint8_t> d_cands_hi;  // bits 16-23 of the candidates' atoms (batches with a wide structure)
    DevBuf<WorkItem> d_wide_work;  // the work items of the wide structures' launch
    PinnedBuf<WorkItem> h_wide_work;
    DevBuf<PairEntry> d_pairs;
    DevBuf<unsigned long long> d_counters;  // [0] hits, [1] candidates, [2] pairs, [3] pages of triples
    DevBuf<uint32_t> d_queue;  // the queued block search: next unit per chunk class (8 counters)
    // shared state of every (structure, template) pair: candidate count | cut short << 31, last candidate + 1
    DevBuf<uint32_t> d_pair_count, d_pair_last, d_pair_tie;
    DevBuf<unsigned long long> d_pair_min;  // (both initialised on the device per listed pair, not per query)
    // triples between k_search and k_deep: pages, their per-structure directory, its cursors
    DevBuf<DeepEntry> d_pages;
    DevBuf<uint2> d_dir;
    DevBuf<uint32_t> d_dir_off, d_dir_n, d_dir_head, d_dir_triples;
    PinnedBuf<uint32_t> h_dir_off;
    std::vector<uint32_t> atoms_of, dir_base;  // per structure: atoms, fixed part of its directory share (read again after an overflow)
    unsigned long long page_capacity = 0;
    bool count_candidates = false;
    std::shared_ptr<StagedSelection> selection;
    unsigned long long n_atoms = 0;                                 // atoms of the batch (for the selection's per-atom needs)
    bool calibrates = false;                                        // this query clocks the selection's templates
    std::shared_ptr<IndexShared> index_shared;                      // the batch's (see tm_batch)
    DevBuf<unsigned long long> d_pool_used;                         // kind-list entries the index build needed
    DevBuf<unsigned long long> d_tab_used;                          // pair-table row slots, entries the build needed
    unsigned long long pool_capacity = 0, row_capacity = 0, ent_capacity = 0;
    DevBuf<uint32_t> d_overflow;
    DevBuf<double> d_cand_rmsd, d_cand_fit, d_models;
    DevBuf<int32_t> d_model_set;
    DevBuf<uint32_t> d_model_offsets, d_model_votes;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [4]: between the table stages and the depth-first part
    bool waited = false;
    uint32_t overflow = 0;
    unsigned long long n_hits = 0, n_cands = 0, n_pairs = 0;
    bool split_recorded = false;  // ev[4] (between the block search and the depth-first part) was recorded
    // The query's counters and flags travel to page-locked host memory behind its last kernel, and an event marks their
    // arrival: tm_result_wait waits for THAT, not for the whole stream -- searches of the next pass may already be queued
    // on it (bench.py's N > 1 loop finishes pass j while pass j + 1 runs).  [0..3] counters, [4] overflow word,
    // [5] kind-list entries the index needed, [6..7] pair-table row slots / entries.
    PinnedBuf<unsigned long long> h_outcome;
    // wide batches only: per long structure of the batch, whether the index build found it unsearchable (see tm_query)
    PinnedBuf<uint32_t> h_unsearchable;
    std::vector<uint32_t> long_structures, unsearched;
    hipEvent_t ev_done = nullptr, ev_appended = nullptr;
    bool outcome_queued = false, appended = false;
    ~tm_result() {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (ev_appended) (void)hipEventDestroy(ev_appended);
    }
};

namespace {

std::atomic<uint64_t> g_serial{0};

// Search order of one template.  Any order enumerates the same set of assignments; this one keeps
// the search tree narrow:
//   1. the LEAD atom (most selective atom) of every template residue first -- greedily, starting
//      with the rarest kind and then always the residue whose lead is rare AND close to the leads
//      already placed (a short template distance means a thin shell of admissible positions);
//      lead-to-lead distances prune whole residues before any residue is expanded;
//   2. then the remaining atoms, residue by residue in that order, each looked up inside the target
//      residue its lead was assigned to.
struct Plan {
    std::vector<int> order;  // search position -> template atom
    std::vector<int> group;  // search position -> group index (numbered in lead order)
    std::vector<int> ref;    // search position -> search position of the group's lead
};

Plan plan_template(int k, const uint8_t *group, const double *est, const double *xyz) {
    std::vector<int> ids;  // distinct group ids in first-appearance order
    for (int i = 0; i < k; ++i)
        if (std::find(ids.begin(), ids.end(), (int)group[i]) == ids.end()) ids.push_back(group[i]);
    const int G = (int)ids.size();
    std::vector<int> lead(G, -1);
    for (int g = 0; g < G; ++g)
        for (int i = 0; i < k; ++i)
            if (group[i] == ids[g] && (lead[g] < 0 || est[i] < est[lead[g]])) lead[g] = i;
    std::vector<int> chosen;  // group slots in lead order
    std::vector<bool> used(G, false);
    for (int step = 0; step < G; ++step) {
        int best = -1;
        double best_score = 0;
        for (int g = 0; g < G; ++g) {
            if (used[g]) continue;
            double score = est[lead[g]];
            if (!chosen.empty()) {
                double dmin2 = 1e300;
                for (int c : chosen) {
                    double d2 = 0;
                    for (int a = 0; a < 3; ++a) {
                        const double d = xyz[3 * lead[g] + a] - xyz[3 * lead[c] + a];
                        d2 += d * d;
                    }
                    dmin2 = std::min(dmin2, d2);
                }
                score *= dmin2 + 1.0;
            }
            if (best < 0 || score < best_score) { best = g; best_score = score; }
        }
        used[best] = true;
        chosen.push_back(best);
    }
    Plan p;
    for (int n = 0; n < G; ++n) {
        p.order.push_back(lead[chosen[n]]);
        p.group.push_back(n);
        p.ref.push_back(n);
    }
    for (int n = 0; n < G; ++n)
        for (int i = 0; i < k; ++i)
            if (group[i] == ids[chosen[n]] && i != lead[chosen[n]]) {
                p.order.push_back(i);
                p.group.push_back(n);
                p.ref.push_back(n);
            }
    return p;
}

}  // namespace

extern "C" {

int tm_abi_version(void) { return TM_ABI_VERSION; }

const char *tm_last_error(void) { return g_err.c_str(); }

int tm_stream_create(int device, void **stream) {
    if (!stream) return fail(TM_E_INVALID, "stream is NULL");
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = st;
    return TM_OK;
}

int tm_stream_destroy(int device, void *stream) {
    if (!stream) return TM_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return TM_OK;
}

int tm_host_threads(void) { return (int)host_team(); }

int tm_device_count(int *count) {
    if (!count) return fail(TM_E_INVALID, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return TM_OK;
}

int tm_library_create(int device, const tm_library_desc *d, tm_library **out) {
    if (!d || !out || !d->atom_offsets || (d->n_templates && (!d->xyz || !d->res_mask || !d->name_mask || !d->group)))
        return fail(TM_E_INVALID, "tm_library_create: NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    const uint32_t nt = d->n_templates;
    const uint32_t na = d->atom_offsets[nt];
    std::vector<double> xyz(3 * (size_t)na), w(na);
    std::vector<uint16_t> kind(na);
    std::vector<uint8_t> group(na), orig(na), chain(na), ref(na);
    std::map<std::tuple<uint8_t, uint32_t, uint64_t>, uint32_t> kinds;
    std::vector<float> cost_weight(nt, 1.f);
    std::vector<uint32_t> k_resmask;
    std::vector<uint64_t> k_namemask;
    std::vector<uint8_t> k_any;
    std::vector<int32_t> k_lead;
    std::vector<uint32_t> lead_kind;

    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t o = d->atom_offsets[t];
        const int k = (int)(d->atom_offsets[t + 1] - o);
        if (k < 1 || k > kMaxK)
            return fail(TM_E_UNSUPPORTED, "template " + std::to_string(t) + " has " + std::to_string(k) +
                                              " atoms; this build supports 1.." + std::to_string(kMaxK));
        // expected admissible atoms per template atom: (#residue types) x (#atom names)
        double est[kMaxK];
        for (int i = 0; i < k; ++i) {
            const bool any = d->any_residue && d->any_residue[o + i];
            const double nres = any ? 20.0 : (double)__builtin_popcount(d->res_mask[o + i]);
            est[i] = nres * (double)__builtin_popcountll(d->name_mask[o + i]);
        }
        Plan p = plan_template(k, d->group + o, est, d->xyz + 3 * (size_t)o);
        if (*std::max_element(p.group.begin(), p.group.end()) + 1 > kMaxG)
            return fail(TM_E_UNSUPPORTED, "template " + std::to_string(t) + " has more than " +
                                              std::to_string(kMaxG) + " residues");
        for (int pos = 0; pos < k; ++pos) {
            const int i = p.order[pos];
            const uint32_t dst = o + pos, src = o + i;
            for (int a = 0; a < 3; ++a) xyz[3 * (size_t)dst + a] = d->xyz[3 * (size_t)src + a];
            w[dst] = d->dist_weight ? d->dist_weight[src] : 0.0;
            group[dst] = (uint8_t)p.group[pos];
            ref[dst] = (uint8_t)p.ref[pos];
            orig[dst] = (uint8_t)i;
            chain[dst] = d->chain ? d->chain[src] : 0;
            const uint8_t any = (d->any_residue && d->any_residue[src]) ? 1 : 0;
            auto key = std::make_tuple(any, any ? 0u : d->res_mask[src], d->name_mask[src]);
            auto it = kinds.find(key);
            if (it == kinds.end()) {
                if (kinds.size() >= 65535) return fail(TM_E_UNSUPPORTED, "more than 65535 distinct atom kinds");
                it = kinds.emplace(key, (uint32_t)kinds.size()).first;
                k_any.push_back(any);
                k_resmask.push_back(std::get<1>(key));
                k_namemask.push_back(std::get<2>(key));
                k_lead.push_back(-1);
            }
            kind[dst] = (uint16_t)it->second;
            const bool lead = p.ref[pos] == pos;
            if (lead && k_lead[it->second] < 0) {
                k_lead[it->second] = (int32_t)lead_kind.size();
                lead_kind.push_back(it->second);
            }
        }
        {
            const int G = *std::max_element(p.group.begin(), p.group.end()) + 1;
            auto expected = [&](uint32_t pos) {
                const uint32_t kd = kind[o + pos];
                return (double)(k_any[kd] ? 20 : __builtin_popcount(k_resmask[kd])) * (double)__builtin_popcountll(k_namemask[kd]);
            };
            cost_weight[t] = (float)(expected(0) * (G >= 2 ? expected(1) : 1.0));
        }
    }

    tm_library *lib = new (std::nothrow) tm_library;
    if (!lib) return fail(TM_E_NOMEM, "out of host memory");
    lib->device = device;
    lib->n_templates = nt; lib->n_atoms = na;
    lib->n_kinds = (uint32_t)k_any.size(); lib->n_lead = (uint32_t)lead_kind.size();
    lib->t_off.assign(d->atom_offsets, d->atom_offsets + nt + 1);
    lib->cost_weight = std::move(cost_weight);
    lib->serial = ++g_serial;
    lib->h_lead_li.assign(na, 0xffffu);
    lib->h_n_groups.assign(nt, 0);
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t o = d->atom_offsets[t], k = d->atom_offsets[t + 1] - o;
        uint8_t g = 0;
        for (uint32_t i = 0; i < k; ++i) g = std::max<uint8_t>(g, (uint8_t)(group[o + i] + 1));
        lib->h_n_groups[t] = g;
        for (uint32_t i = 0; i < g; ++i) lib->h_lead_li[o + i] = (uint16_t)k_lead[kind[o + i]];
    }
    lib->h_xyz = xyz;
    lib->h_w = w;
    hipError_t e = hipSuccess;
    auto up = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    up(lib->d_t_off.upload(lib->t_off)); up(lib->d_xyz.upload(xyz)); up(lib->d_w.upload(w));
    up(lib->d_kind.upload(kind)); up(lib->d_group.upload(group)); up(lib->d_ref.upload(ref)); up(lib->d_orig.upload(orig));
    up(lib->d_chain.upload(chain)); up(lib->d_k_resmask.upload(k_resmask));
    up(lib->d_k_namemask.upload(k_namemask)); up(lib->d_k_any.upload(k_any));
    up(lib->d_k_lead.upload(k_lead)); up(lib->d_lead_kind.upload(lead_kind));
    if (d->residue_offsets && d->orient_first && d->orient_second && d->orient_vec) {
        const uint32_t nr = d->residue_offsets[nt];
        up(lib->d_r_off.upload(std::vector<uint32_t>(d->residue_offsets, d->residue_offsets + nt + 1)));
        up(lib->d_r_first.upload(std::vector<uint8_t>(d->orient_first, d->orient_first + nr)));
        up(lib->d_r_second.upload(std::vector<uint8_t>(d->orient_second, d->orient_second + nr)));
        up(lib->d_r_vec.upload(std::vector<double>(d->orient_vec, d->orient_vec + 3 * (size_t)nr)));
        lib->has_orient = true;
    }
    if (e != hipSuccess) {
        delete lib;
        return fail(TM_E_DEVICE, std::string("tm_library_create upload: ") + hipGetErrorString(e));
    }
    *out = lib;
    return TM_OK;
}

void tm_library_destroy(tm_library *lib) {
    if (!lib) return;
    (void)hipSetDevice(lib->device);
    (void)hipDeviceSynchronize();  // its buffers go back to the allocation cache: nothing may still read them
    delete lib;
}

static hipError_t ingest_stream(int device, hipStream_t *out);

}  // extern "C" (reopened below: the next function is shared by the three batch constructors)

namespace {
// Encoded structures -> residue-sorted records, uploaded.  The coordinates of structure s start at xyz_of[s]
// (the structures need not sit back to back in host memory); the encoded columns are contiguous.
int batch_from_encoded(int device, const tm_batch_desc *d, const double *const *xyz_of, const uint32_t *nres_known,
                       tm_batch **out) {
    if (!d || !out || !d->atom_offsets ||
        (d->n_structures && d->atom_offsets[d->n_structures] && (!xyz_of || !d->res_type || !d->name_id || !d->residue)))
        return fail(TM_E_INVALID, "tm_batch_create: NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    PhaseTrace trace("tm_batch_create");
    const uint32_t ns = d->n_structures;
    const uint32_t na = d->atom_offsets[ns];
    PinnedBuf<double> xyz;
    PinnedBuf<float4> rec;
    PinnedBuf<uint8_t> chain;
    PinnedBuf<uint32_t> orig;
    PinnedBuf<uint16_t> res_start;
    PinnedBuf<uint8_t> res_start_hi;  // (bits 16-23: a batch with a structure of more than 65535 atoms)
    HIP_TRY(xyz.alloc(3 * (size_t)na)); HIP_TRY(rec.alloc(na)); HIP_TRY(chain.alloc(na)); HIP_TRY(orig.alloc(na));
    std::vector<uint32_t> s_roff(ns + 1, 0), nres_of(ns, 0);
    uint32_t max_atoms = 0, max_res = 0, max_atoms_narrow = 0, max_res_narrow = 0;
    bool has_wide = false;

    // pass 1: limits and residue counts (they fix where every structure's residue table starts)
    for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t o = d->atom_offsets[s], n = d->atom_offsets[s + 1] - o;
        if (n > TM_MAX_STRUCTURE_ATOMS)
            return fail(TM_E_UNSUPPORTED, "structure " + std::to_string(s) + " has " + std::to_string(n) +
                                              " atoms; limit " + std::to_string(TM_MAX_STRUCTURE_ATOMS));
        uint32_t nres = nres_known ? nres_known[s] : 0u;
        if (!nres_known)
            for (uint32_t i = 0; i < n; ++i) nres = std::max(nres, d->residue[o + i] + 1);
        if (nres > TM_MAX_STRUCTURE_RESIDUES)
            return fail(TM_E_UNSUPPORTED, "structure " + std::to_string(s) + " has more than " +
                                              std::to_string(TM_MAX_STRUCTURE_RESIDUES) + " residues");
        nres_of[s] = nres;
        s_roff[s + 1] = s_roff[s] + nres + 1;
        max_atoms = std::max(max_atoms, n);
        max_res = std::max(max_res, nres);
        if (n > TM_NARROW_STRUCTURE_ATOMS) has_wide = true;
        else { max_atoms_narrow = std::max(max_atoms_narrow, n); max_res_narrow = std::max(max_res_narrow, nres); }
    }
    HIP_TRY(res_start.alloc(s_roff[ns]));
    if (has_wide) HIP_TRY(res_start_hi.alloc(s_roff[ns]));
    trace.mark("limits");

    // pass 2: the structures are independent; a small team of threads flattens them
    std::atomic<uint32_t> extent_bits{0};  // largest |centred coordinate| (non-negative floats order like their bits)
    auto flatten = [&](uint32_t s_begin, uint32_t s_end) {
        std::vector<uint32_t> count, perm, fill;
        float reach = 0.f;
        for (uint32_t s = s_begin; s < s_end; ++s) {
            const uint32_t o = d->atom_offsets[s], n = d->atom_offsets[s + 1] - o, nres = nres_of[s];
            // counting sort by residue ordinal: atoms of a residue become contiguous, file order kept inside
            count.assign(nres + 1, 0);
            for (uint32_t i = 0; i < n; ++i) count[d->residue[o + i] + 1]++;
            for (uint32_t r = 0; r < nres; ++r) count[r + 1] += count[r];
            for (uint32_t r = 0; r <= nres; ++r) {
                res_start[s_roff[s] + r] = (uint16_t)count[r];
                if (has_wide) res_start_hi[s_roff[s] + r] = (uint8_t)(count[r] >> 16);
            }
            perm.assign(n, 0);
            fill.assign(count.begin(), count.end() - 1);
            for (uint32_t i = 0; i < n; ++i) perm[fill[d->residue[o + i]]++] = i;
            const double *sx = xyz_of[s];
            double c[3] = {0, 0, 0};
            for (uint32_t i = 0; i < n; ++i)
                for (int a = 0; a < 3; ++a) c[a] += sx[3 * (size_t)i + a];
            if (n)
                for (int a = 0; a < 3; ++a) c[a] /= (double)n;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t src = o + perm[j], dst = o + j;
                const double *px = sx + 3 * (size_t)perm[j];
                for (int a = 0; a < 3; ++a) xyz[3 * (size_t)dst + a] = px[a];
                float4 r;
                r.x = (float)(px[0] - c[0]);
                r.y = (float)(px[1] - c[1]);
                r.z = (float)(px[2] - c[2]);
                reach = std::max(reach, std::max(std::fabs(r.x), std::max(std::fabs(r.y), std::fabs(r.z))));
                const uint32_t nm = std::min<uint32_t>(d->name_id[src], TM_NAME_OTHER);
                const uint32_t rt = std::min<uint32_t>(d->res_type[src], TM_RES_OTHER);
                const uint32_t m = meta_pack(d->residue[src], nm, rt);
                std::memcpy(&r.w, &m, 4);
                rec[dst] = r;
                chain[dst] = d->chain ? d->chain[src] : 0;
                orig[dst] = perm[j];
            }
        }
        if (!(reach <= 3.0e38f)) reach = 3.0e38f;  // a NaN or infinite coordinate: the widest slack there is
        uint32_t bits, seen = extent_bits.load();
        std::memcpy(&bits, &reach, 4);
        while (seen < bits && !extent_bits.compare_exchange_weak(seen, bits)) {}
    };
    {
        const uint32_t team = std::min<uint32_t>(host_team(), std::max(1u, na / host_grain()));
        if (team <= 1) {
            flatten(0, ns);
        } else {
            // contiguous slices of structures with about the same number of atoms each
            std::vector<std::pair<uint32_t, uint32_t>> slices;
            uint32_t s0 = 0;
            for (uint32_t t = 0; t < team; ++t) {
                const uint64_t target = (uint64_t)na * (t + 1) / team;
                uint32_t s1 = s0;
                while (s1 < ns && (t + 1 == team || d->atom_offsets[s1 + 1] <= target)) ++s1;
                if (s1 > s0) slices.emplace_back(s0, s1);
                s0 = s1;
            }
            HostPool::run_anywhere((uint32_t)slices.size(), [&](uint32_t t) { flatten(slices[t].first, slices[t].second); });
        }
    }

    trace.mark("flatten");
    tm_batch *b = new (std::nothrow) tm_batch;
    if (!b) return fail(TM_E_NOMEM, "out of host memory");
    b->device = device;
    b->n_structs = ns; b->n_atoms = na; b->max_atoms = max_atoms; b->max_res = max_res;
    b->has_wide = has_wide; b->max_atoms_narrow = max_atoms_narrow; b->max_res_narrow = max_res_narrow;
    {
        const uint32_t bits = extent_bits.load();
        std::memcpy(&b->extent, &bits, 4);
    }
    b->s_off.assign(d->atom_offsets, d->atom_offsets + ns + 1);
    hipError_t e = hipSuccess;
    auto up = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    hipStream_t stream = nullptr;
    up(ingest_stream(device, &stream));
    up(b->d_xyz.upload_async(xyz.p, xyz.n, stream)); up(b->d_rec.upload_async(rec.p, rec.n, stream));
    up(b->d_chain.upload_async(chain.p, chain.n, stream)); up(b->d_orig.upload_async(orig.p, orig.n, stream));
    up(b->d_res_start.upload_async(res_start.p, res_start.n, stream));
    if (has_wide) up(b->d_res_start_hi.upload_async(res_start_hi.p, res_start_hi.n, stream));
    up(b->d_s_off.upload(b->s_off, stream)); up(b->d_s_roff.upload(s_roff, stream));  // these also wait for the stream
    up(b->d_cursor.alloc(1)); up(b->d_overflow.alloc(1));
    if (e == hipSuccess && stream) up(hipStreamSynchronize(stream));
    trace.mark("upload");
    if (e != hipSuccess) {
        delete b;
        return fail(TM_E_DEVICE, std::string("tm_batch_create upload: ") + hipGetErrorString(e));
    }
    *out = b;
    return TM_OK;
}
}  // namespace

extern "C" {

int tm_batch_create(int device, const tm_batch_desc *d, tm_batch **out) {
    if (!d || !out || !d->atom_offsets) return fail(TM_E_INVALID, "tm_batch_create: NULL argument");
    const uint32_t ns = d->n_structures;
    std::vector<const double *> xyz_of(ns, nullptr);
    if (d->xyz)
        for (uint32_t s = 0; s < ns; ++s) xyz_of[s] = d->xyz + 3 * (size_t)d->atom_offsets[s];
    // a structure beyond TM_MAX_STRUCTURE_ATOMS stays in the batch as an empty one (tm_batch_refused): the per-atom
    // columns are copied without its atoms
    // (so does one of more than TM_MAX_STRUCTURE_RESIDUES residues: the kernels number residues with 16 bits)
    std::vector<uint32_t> refused, refused_atoms;
    for (uint32_t s = 0; s < ns; ++s) {
        if (d->atom_offsets[s + 1] < d->atom_offsets[s]) continue;
        const uint32_t o = d->atom_offsets[s], n = d->atom_offsets[s + 1] - o;
        bool drop = n > TM_MAX_STRUCTURE_ATOMS;
        if (!drop && n > TM_MAX_STRUCTURE_RESIDUES && d->residue) {
            uint32_t top = 0;
            for (uint32_t i = 0; i < n; ++i) top = std::max(top, d->residue[o + i]);
            drop = top >= TM_MAX_STRUCTURE_RESIDUES;
        }
        if (drop) {
            refused.push_back(s);
            refused_atoms.push_back(n);
        }
    }
    if (refused.empty()) return batch_from_encoded(device, d, d->xyz ? xyz_of.data() : nullptr, nullptr, out);
    if (d->atom_offsets[ns] && (!d->res_type || !d->name_id || !d->residue)) return fail(TM_E_INVALID, "tm_batch_create: NULL argument");
    std::vector<uint32_t> off(ns + 1, 0), residue;
    std::vector<uint8_t> res_type, name_id, chain;
    size_t next = 0;
    for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t o = d->atom_offsets[s], n = d->atom_offsets[s + 1] - o;
        const bool drop = next < refused.size() && refused[next] == s;
        if (drop) ++next;
        else {
            res_type.insert(res_type.end(), d->res_type + o, d->res_type + o + n);
            name_id.insert(name_id.end(), d->name_id + o, d->name_id + o + n);
            residue.insert(residue.end(), d->residue + o, d->residue + o + n);
            if (d->chain) chain.insert(chain.end(), d->chain + o, d->chain + o + n);
        }
        off[s + 1] = (uint32_t)res_type.size();
    }
    tm_batch_desc kept = *d;
    kept.atom_offsets = off.data();
    kept.xyz = nullptr;
    kept.res_type = res_type.data(); kept.name_id = name_id.data(); kept.residue = residue.data();
    kept.chain = d->chain ? chain.data() : nullptr;
    const int rc = batch_from_encoded(device, &kept, d->xyz ? xyz_of.data() : nullptr, nullptr, out);
    if (rc == TM_OK) { (*out)->refused = std::move(refused); (*out)->refused_atoms = std::move(refused_atoms); }
    return rc;
}

}  // extern "C"

namespace {

// Raw atom-table columns, one tm_structure_columns per structure -> the compact hand-over of tm_ingest.hip (round 5).
// On host threads, per structure: vocabulary ids, residue ordinals (by first appearance; only their equality matters)
// and chain ordinals into one packed word + one byte per atom, the coordinates as int32 thousandths of an Angstrom
// when every one of them is such a value EXACTLY -- (double)k / 1000.0 has the bits of the caller's double; a PDB
// file's 3-decimal fields always are -- else the structure's fp64 values go along in a side array.  17 bytes per atom
// cross PCIe instead of 46, the host no longer sorts, centres or copies records (k_flatten does), and the call returns
// when the copies are QUEUED: the batch carries an event (tm_batch::ready) that its queries wait for on their stream.
// A structure whose residues are not contiguous in the file (rare) is sorted here, stably by residue ordinal, and sends
// its file-order indices along.
struct EncodedStructure {
    uint32_t nres = 0;
    bool exact = true, scattered = false;
    double c[3] = {0, 0, 0};
    float reach = 0.f;
    std::vector<uint32_t> perm;  // scattered only: device position -> file index
};

// n3 doubles -> int32 thousandths; false when one of them is not (double)k / 1000.0 bit for bit (then out is garbage)
inline bool to_milli(const double *x, size_t n3, int32_t *out) {
    const __m128d k1000 = _mm_set1_pd(1000.0);
    __m128i differ = _mm_setzero_si128();
    size_t i = 0;
    for (; i + 2 <= n3; i += 2) {
        const __m128d v = _mm_loadu_pd(x + i);
        const __m128i k = _mm_cvtpd_epi32(_mm_mul_pd(v, k1000));  // round to nearest; out of range or NaN: 0x80000000
        const __m128d back = _mm_div_pd(_mm_cvtepi32_pd(k), k1000);
        differ = _mm_or_si128(differ, _mm_xor_si128(_mm_castpd_si128(back), _mm_castpd_si128(v)));  // bitwise: -0.0 is not 0.0
        _mm_storel_epi64(reinterpret_cast<__m128i *>(out + i), k);
    }
    bool exact = _mm_movemask_epi8(_mm_cmpeq_epi32(differ, _mm_setzero_si128())) == 0xffff;
    for (; i < n3; ++i) {
        const __m128d v = _mm_set_sd(x[i]);
        const int32_t k = _mm_cvtsd_si32(_mm_mul_sd(v, k1000));
        const double back = (double)k / 1000.0;
        exact = exact && std::memcmp(&back, x + i, 8) == 0;
        out[i] = k;
    }
    return exact;
}

int batch_from_structures(int device, uint32_t ns, const tm_structure_columns *src, const tm_vocab *vocab, tm_batch **out,
                          std::vector<uint32_t> *many_residues = nullptr);

// A structure of more than TM_MAX_STRUCTURE_RESIDUES residues is found out by the encoder: the batch is then laid out
// once more with that structure emptied -- it stays in the batch without atoms, listed by tm_batch_refused -- instead
// of failing the other structures with it (as a file beyond the atom limit is treated).
int batch_from_structures_refusing(int device, uint32_t ns, const tm_structure_columns *src, const tm_vocab *vocab,
                                   tm_batch **out) {
    std::vector<uint32_t> many;
    int rc = batch_from_structures(device, ns, src, vocab, out, &many);
    if (rc != TM_OK || many.empty()) return rc;
    std::vector<tm_structure_columns> kept(src, src + ns);
    std::vector<uint32_t> atoms;
    for (uint32_t s : many) {
        atoms.push_back((uint32_t)std::min<uint64_t>(kept[s].n_atoms, 0xffffffffull));
        kept[s].n_atoms = 0;
    }
    rc = batch_from_structures(device, ns, kept.data(), vocab, out, nullptr);
    if (rc != TM_OK) return rc;
    // merge with what the second pass refused for its atom count (ascending structure numbers)
    std::vector<std::pair<uint32_t, uint32_t>> all;
    for (size_t i = 0; i < (*out)->refused.size(); ++i) all.emplace_back((*out)->refused[i], (*out)->refused_atoms[i]);
    for (size_t i = 0; i < many.size(); ++i) all.emplace_back(many[i], atoms[i]);
    std::sort(all.begin(), all.end());
    (*out)->refused.clear(); (*out)->refused_atoms.clear();
    for (auto &e : all) { (*out)->refused.push_back(e.first); (*out)->refused_atoms.push_back(e.second); }
    return TM_OK;
}

int batch_from_structures(int device, uint32_t ns, const tm_structure_columns *src, const tm_vocab *vocab, tm_batch **out,
                          std::vector<uint32_t> *many_residues) {
    PhaseTrace trace("tm_batch_create_from_structures");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    std::vector<uint32_t> refused, refused_atoms;
    std::vector<tm_structure_columns> kept;  // (a copy of the descriptors only when one has to be emptied)
    for (uint32_t s = 0; s < ns; ++s)
        if (src[s].n_atoms > TM_MAX_STRUCTURE_ATOMS) {
            if (kept.empty()) kept.assign(src, src + ns);
            refused.push_back(s);
            refused_atoms.push_back((uint32_t)std::min<uint64_t>(src[s].n_atoms, 0xffffffffull));
            kept[s].n_atoms = 0;
        }
    if (!kept.empty()) src = kept.data();
    tm_batch *b = new (std::nothrow) tm_batch;
    if (!b) return fail(TM_E_NOMEM, "out of host memory");
    struct Guard { tm_batch *b; ~Guard() { if (b) { (void)hipDeviceSynchronize(); delete b; } } } guard{b};  // (copies may be queued)
    b->device = device; b->n_structs = ns;
    b->s_off.assign(ns + 1, 0);
    std::vector<uint32_t> &atom_offsets = b->s_off;
    for (uint32_t s = 0; s < ns; ++s) {
        const uint64_t n = src[s].n_atoms;
        if (n && (!src[s].xyz || !src[s].name || !src[s].residue_name || !src[s].chain_id || !src[s].residue_number ||
                  !src[s].insertion_code))
            return fail(TM_E_INVALID, "structure " + std::to_string(s) + ": NULL column");
        if ((uint64_t)atom_offsets[s] + n > 0xffffffffull) return fail(TM_E_UNSUPPORTED, "more than 2^32 atoms in one batch");
        atom_offsets[s + 1] = atom_offsets[s] + (uint32_t)n;
    }
    const uint32_t na = atom_offsets[ns];
    // the vocabulary as open-addressed tables keyed by the raw field (a linear scan per atom name was the cost of
    // this function: names change with every atom)
    struct Lookup {
        std::vector<uint32_t> key;
        std::vector<uint16_t> val;  // 0xffff: empty
        uint32_t mask = 0;
        void build(uint32_t n, const char *fields, uint32_t width, const uint8_t *ids) {
            uint32_t cap = 16;
            while (cap < 4 * n) cap <<= 1;
            key.assign(cap, 0); val.assign(cap, 0xffffu); mask = cap - 1;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t k = 0;
                std::memcpy(&k, fields + (size_t)width * i, width);
                uint32_t h = (k * 2654435761u) >> 7 & mask;
                while (val[h] != 0xffffu && key[h] != k) h = (h + 1) & mask;
                if (val[h] == 0xffffu) { key[h] = k; val[h] = ids[i]; }  // the first listing of a field wins, as in a scan
            }
        }
        bool find(uint32_t k, uint32_t *id) const {
            uint32_t h = (k * 2654435761u) >> 7 & mask;
            while (val[h] != 0xffffu) {
                if (key[h] == k) { *id = val[h]; return true; }
                h = (h + 1) & mask;
            }
            return false;
        }
    } names, elements, residues;
    names.build(vocab->n_atom_names, vocab->atom_names, 4, vocab->atom_name_ids);
    elements.build(vocab->n_elements, vocab->elements, 2, vocab->element_ids);
    residues.build(vocab->n_residues, vocab->residue_names, 3, vocab->residue_ids);

    // one page-locked arena: [descriptors | packed words | thousandths | chain ordinals]; the side arrays (fp64
    // coordinates of inexact structures, file-order indices of scattered ones) follow in a second, usually empty, one
    const size_t o_desc = 0, o_meta = o_desc + sizeof(FlatDesc) * (size_t)ns, o_milli = o_meta + 4 * (size_t)na,
                 o_chain = o_milli + 12 * (size_t)na, arena_bytes = (o_chain + (size_t)na + 15) & ~(size_t)15;
    HIP_TRY(b->h_stage.alloc(arena_bytes));
    FlatDesc *h_desc = reinterpret_cast<FlatDesc *>(b->h_stage.p + o_desc);
    uint32_t *h_meta = reinterpret_cast<uint32_t *>(b->h_stage.p + o_meta);
    int32_t *h_milli = reinterpret_cast<int32_t *>(b->h_stage.p + o_milli);
    uint8_t *h_chain = reinterpret_cast<uint8_t *>(b->h_stage.p + o_chain);
    std::vector<EncodedStructure> enc(ns);
    trace.mark("setup");

    auto encode = [&](uint32_t s_begin, uint32_t s_end) {
        std::vector<uint16_t> seen_chain;
        std::vector<std::pair<uint64_t, uint32_t>> table;  // residue key -> ordinal, open addressing, per structure
        std::vector<uint32_t> count, fill, t_meta;
        std::vector<int32_t> t_milli;
        std::vector<uint8_t> t_chain;
        for (uint32_t s = s_begin; s < s_end; ++s) {
            const tm_structure_columns &c = src[s];
            const uint32_t o = atom_offsets[s], n = atom_offsets[s + 1] - o;
            EncodedStructure &E = enc[s];
            if (n == 0) continue;
            seen_chain.clear();
            uint32_t cap = 64;
            while (cap < 2 * n) cap <<= 1;  // at most one residue per atom
            table.assign(cap, {~0ull, 0u});
            uint32_t n_keys = 0;
            uint64_t last_key = ~0ull;
            uint32_t last_ord = 0;
            uint32_t last_res = 0, last_res_id = TM_RES_OTHER;
            bool have_res = false, scattered = false;
            uint16_t last_chain = 0;
            uint8_t last_chain_ord = 0;
            bool have_chain = false;
            for (uint32_t i = 0; i < n; ++i) {
                const size_t a = (size_t)o + i;
                uint32_t nm = 0;
                std::memcpy(&nm, c.name + 4 * (size_t)i, 4);
                uint32_t id = TM_NAME_OTHER;
                if (!names.find(nm, &id) || id == TM_NAME_OTHER) (void)elements.find(nm & 0xffffu, &id);  // by element (cols 13-14)
                uint32_t rn = 0;
                std::memcpy(&rn, c.residue_name + 3 * (size_t)i, 3);
                if (!have_res || rn != last_res) {
                    uint32_t rid = TM_RES_OTHER;
                    (void)residues.find(rn, &rid);
                    last_res = rn; last_res_id = rid; have_res = true;
                }
                const uint16_t ch = (uint16_t)((uint8_t)c.chain_id[2 * (size_t)i] << 8 | (uint8_t)c.chain_id[2 * (size_t)i + 1]);
                const uint64_t key = ((uint64_t)ch << 40) | ((uint64_t)(uint32_t)(c.residue_number[i] + (1 << 24)) << 8) |
                                     (uint8_t)c.insertion_code[i];
                if (key != last_key) {
                    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (cap - 1);
                    while (table[h].first != ~0ull && table[h].first != key) h = (h + 1) & (cap - 1);
                    if (table[h].first == ~0ull) table[h] = {key, n_keys++};
                    else scattered = true;  // a residue seen before, and not the one just left: its atoms have to be gathered
                    last_key = key;
                    last_ord = table[h].second;
                }
                h_meta[a] = meta_pack(last_ord, std::min<uint32_t>(id, TM_NAME_OTHER), std::min<uint32_t>(last_res_id, TM_RES_OTHER));
                if (!have_chain || ch != last_chain) {
                    uint32_t k = 0;
                    while (k < seen_chain.size() && seen_chain[k] != ch) ++k;
                    if (k == seen_chain.size()) seen_chain.push_back(ch);
                    last_chain = ch; last_chain_ord = (uint8_t)(k & 0xffu); have_chain = true;
                }
                h_chain[a] = last_chain_ord;
            }
            E.nres = n_keys;
            E.scattered = scattered;
            if (n_keys > TM_MAX_STRUCTURE_RESIDUES) continue;  // (the caller fails the batch)
            int32_t *mk = h_milli + 3 * (size_t)o;
            E.exact = to_milli(c.xyz, 3 * (size_t)n, mk);
            if (E.exact) {
                long long sum[3] = {0, 0, 0};
                int32_t lo[3] = {mk[0], mk[1], mk[2]}, hi[3] = {mk[0], mk[1], mk[2]};
                for (uint32_t i = 0; i < n; ++i)
                    for (int ax = 0; ax < 3; ++ax) {
                        const int32_t v = mk[3 * (size_t)i + ax];
                        sum[ax] += v;
                        lo[ax] = std::min(lo[ax], v); hi[ax] = std::max(hi[ax], v);
                    }
                float reach = 0.f;
                for (int ax = 0; ax < 3; ++ax) {
                    E.c[ax] = ((double)sum[ax] / 1000.0) / (double)n;
                    // (what k_flatten computes for the outermost atoms; x -> (float)(x - c) is monotone)
                    reach = std::max(reach, std::fabs((float)((double)lo[ax] / 1000.0 - E.c[ax])));
                    reach = std::max(reach, std::fabs((float)((double)hi[ax] / 1000.0 - E.c[ax])));
                }
                E.reach = reach;
            } else {
                double cc[3] = {0, 0, 0};
                for (uint32_t i = 0; i < n; ++i)
                    for (int ax = 0; ax < 3; ++ax) cc[ax] += c.xyz[3 * (size_t)i + ax];
                float reach = 0.f;
                for (int ax = 0; ax < 3; ++ax) E.c[ax] = cc[ax] / (double)n;
                for (uint32_t i = 0; i < n; ++i)
                    for (int ax = 0; ax < 3; ++ax) reach = std::max(reach, std::fabs((float)(c.xyz[3 * (size_t)i + ax] - E.c[ax])));
                E.reach = reach <= 3.0e38f ? reach : 3.0e38f;  // a NaN or infinite coordinate: the widest slack there is
            }
            if (scattered) {
                // counting sort by residue ordinal: atoms of a residue become contiguous, file order kept inside
                count.assign(n_keys + 1, 0);
                for (uint32_t i = 0; i < n; ++i) count[meta_res(h_meta[o + i]) + 1]++;
                for (uint32_t r = 0; r < n_keys; ++r) count[r + 1] += count[r];
                fill.assign(count.begin(), count.end() - 1);
                E.perm.assign(n, 0);
                for (uint32_t i = 0; i < n; ++i) E.perm[fill[meta_res(h_meta[o + i])]++] = i;
                t_meta.assign(h_meta + o, h_meta + o + n);
                t_chain.assign(h_chain + o, h_chain + o + n);
                t_milli.assign(mk, mk + 3 * (size_t)n);
                for (uint32_t j = 0; j < n; ++j) {
                    const uint32_t i = E.perm[j];
                    h_meta[o + j] = t_meta[i]; h_chain[o + j] = t_chain[i];
                    for (int ax = 0; ax < 3; ++ax) mk[3 * (size_t)j + ax] = t_milli[3 * (size_t)i + ax];
                }
            }
        }
    };
    {
        const uint32_t team = std::min<uint32_t>(host_team(), std::max(1u, na / host_grain()));
        if (team <= 1) {
            encode(0, ns);
        } else {
            // slices of about a quarter of a thread's share: the team takes them from a counter, so a thread that is
            // descheduled for a while (the box grants a CPU quota, not cores) does not hold the others up
            const uint32_t n_slices = team * 4;
            std::vector<std::pair<uint32_t, uint32_t>> slices;
            uint32_t s0 = 0;
            for (uint32_t t = 0; t < n_slices; ++t) {
                const uint64_t target = (uint64_t)na * (t + 1) / n_slices;
                uint32_t s1 = s0;
                while (s1 < ns && (t + 1 == n_slices || atom_offsets[s1 + 1] <= target)) ++s1;
                if (s1 > s0) slices.emplace_back(s0, s1);
                s0 = s1;
            }
            HostPool::run_anywhere((uint32_t)slices.size(), [&](uint32_t t) { encode(slices[t].first, slices[t].second); });
        }
    }
    trace.mark("encode");

    // limits, residue-table offsets, the side arrays' layout, the batch's extent
    PinnedBuf<uint32_t> &h_off = b->h_stage_off;  // [s_off | s_roff]
    HIP_TRY(h_off.alloc(2 * ((size_t)ns + 1)));
    uint32_t *h_s_off = h_off.p, *h_s_roff = h_off.p + ns + 1;
    std::memcpy(h_s_off, atom_offsets.data(), ((size_t)ns + 1) * sizeof(uint32_t));
    h_s_roff[0] = 0;
    uint32_t max_atoms = 0, max_res = 0;
    size_t side_xyz = 0, side_orig = 0;
    float extent = 0.f;
    if (many_residues) {
        for (uint32_t s = 0; s < ns; ++s)
            if (enc[s].nres > TM_MAX_STRUCTURE_RESIDUES) many_residues->push_back(s);
        if (!many_residues->empty()) return TM_OK;  // (*out is still NULL: the caller lays the batch out again without them)
    }
    for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t n = atom_offsets[s + 1] - atom_offsets[s];
        const EncodedStructure &E = enc[s];
        if (E.nres > TM_MAX_STRUCTURE_RESIDUES)
            return fail(TM_E_UNSUPPORTED, "structure " + std::to_string(s) + " has more than " +
                                              std::to_string(TM_MAX_STRUCTURE_RESIDUES) + " residues");
        h_s_roff[s + 1] = h_s_roff[s] + E.nres + 1;
        max_atoms = std::max(max_atoms, n); max_res = std::max(max_res, E.nres);
        if (n > TM_NARROW_STRUCTURE_ATOMS) b->has_wide = true;
        else { b->max_atoms_narrow = std::max(b->max_atoms_narrow, n); b->max_res_narrow = std::max(b->max_res_narrow, E.nres); }
        extent = std::max(extent, E.reach);
        FlatDesc &D = h_desc[s];
        D.c[0] = E.c[0]; D.c[1] = E.c[1]; D.c[2] = E.c[2];
        D.xyz_off = kNoOffset; D.orig_off = kNoOffset;
        if (n && !E.exact) { D.xyz_off = (uint32_t)side_xyz; side_xyz += n; }
        if (n && E.scattered) { D.orig_off = (uint32_t)side_orig; side_orig += n; }
    }
    b->n_atoms = na; b->max_atoms = max_atoms; b->max_res = max_res; b->extent = extent;
    const size_t side_bytes = 24 * side_xyz + 4 * side_orig;
    double *h_side_xyz = nullptr;
    uint32_t *h_side_orig = nullptr;
    if (side_bytes) {
        HIP_TRY(b->h_side.alloc(side_bytes));
        h_side_xyz = reinterpret_cast<double *>(b->h_side.p);
        h_side_orig = reinterpret_cast<uint32_t *>(b->h_side.p + 24 * side_xyz);
        std::vector<uint32_t> todo;
        for (uint32_t s = 0; s < ns; ++s)
            if (h_desc[s].xyz_off != kNoOffset || h_desc[s].orig_off != kNoOffset) todo.push_back(s);
        auto fill_side = [&](uint32_t t) {
            const uint32_t s = todo[t], n = atom_offsets[s + 1] - atom_offsets[s];
            const FlatDesc &D = h_desc[s];
            const EncodedStructure &E = enc[s];
            if (D.orig_off != kNoOffset) std::memcpy(h_side_orig + D.orig_off, E.perm.data(), 4 * (size_t)n);
            if (D.xyz_off != kNoOffset) {
                double *dst = h_side_xyz + 3 * (size_t)D.xyz_off;
                if (!E.scattered) std::memcpy(dst, src[s].xyz, 24 * (size_t)n);
                else
                    for (uint32_t j = 0; j < n; ++j)
                        for (int ax = 0; ax < 3; ++ax) dst[3 * (size_t)j + ax] = src[s].xyz[3 * (size_t)E.perm[j] + ax];
            }
        };
        if (todo.size() >= 8 && side_bytes >= (1u << 20)) HostPool::run_anywhere((uint32_t)todo.size(), fill_side);
        else
            for (uint32_t t = 0; t < todo.size(); ++t) fill_side(t);
    }
    trace.mark("layout");

    hipStream_t stream = nullptr;
    HIP_TRY(ingest_stream(device, &stream));
    HIP_TRY(b->d_stage.alloc(arena_bytes));
    HIP_TRY(b->d_xyz.alloc(3 * (size_t)na)); HIP_TRY(b->d_rec.alloc(na)); HIP_TRY(b->d_chain.alloc(na)); HIP_TRY(b->d_orig.alloc(na));
    HIP_TRY(b->d_res_start.alloc(h_s_roff[ns]));
    if (b->has_wide) HIP_TRY(b->d_res_start_hi.alloc(h_s_roff[ns]));
    HIP_TRY(b->d_s_off.alloc((size_t)ns + 1)); HIP_TRY(b->d_s_roff.alloc((size_t)ns + 1));
    HIP_TRY(b->d_cursor.alloc(1)); HIP_TRY(b->d_overflow.alloc(1));
    HIP_TRY(hipMemcpyAsync(b->d_stage.p, b->h_stage.p, o_chain, hipMemcpyHostToDevice, stream));
    if (na) HIP_TRY(hipMemcpyAsync(b->d_chain.p, h_chain, na, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(b->d_s_off.p, h_s_off, ((size_t)ns + 1) * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(b->d_s_roff.p, h_s_roff, ((size_t)ns + 1) * 4, hipMemcpyHostToDevice, stream));
    if (side_bytes) {
        HIP_TRY(b->d_side.alloc(side_bytes));
        HIP_TRY(hipMemcpyAsync(b->d_side.p, b->h_side.p, side_bytes, hipMemcpyHostToDevice, stream));
    }
    b->upload_bytes = o_chain + na + 8 * ((size_t)ns + 1) + side_bytes;
    if (ns) {
        const uint32_t chunks = std::max(1u, (max_atoms + kFlattenChunk - 1) / kFlattenChunk);
        hipLaunchKernelGGL(k_flatten, dim3(ns, chunks), dim3(256), 0, stream, b->d_s_off.p, b->d_s_roff.p,
                           reinterpret_cast<const FlatDesc *>(b->d_stage.p + o_desc),
                           reinterpret_cast<const uint32_t *>(b->d_stage.p + o_meta),
                           reinterpret_cast<const int32_t *>(b->d_stage.p + o_milli),
                           reinterpret_cast<const double *>(b->d_side.p),
                           reinterpret_cast<const uint32_t *>(b->d_side.p ? b->d_side.p + 24 * side_xyz : nullptr),
                           b->d_xyz.p, b->d_rec.p, b->d_orig.p, b->d_res_start.p, b->has_wide ? b->d_res_start_hi.p : nullptr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->ready, stream));
    // ENZYMM_AMD_SYNC_UPLOAD=1 (A/B, tests): wait here, as every batch constructor did before round 5
    static const bool sync_upload = [] { const char *e = std::getenv("ENZYMM_AMD_SYNC_UPLOAD"); return e && e[0] == '1'; }();
    if (sync_upload) { HIP_TRY(hipEventSynchronize(b->ready)); b->settle(); }
    trace.mark("queue");
    b->refused = std::move(refused); b->refused_atoms = std::move(refused_atoms);
    guard.b = nullptr;
    *out = b;
    return TM_OK;
}

}  // namespace

extern "C" {

int tm_batch_create_from_structures(int device, uint32_t n_structures, const tm_structure_columns *structures,
                                    const tm_vocab *vocab, tm_batch **out) {
    if (!vocab || !out || (n_structures && !structures))
        return fail(TM_E_INVALID, "tm_batch_create_from_structures: NULL argument");
    return batch_from_structures_refusing(device, n_structures, structures, vocab, out);
}

// The same from columns that hold all structures back to back.
int tm_batch_create_from_columns(int device, const tm_columns_desc *d, const tm_vocab *vocab, tm_batch **out) {
    if (!d || !vocab || !out || !d->atom_offsets)
        return fail(TM_E_INVALID, "tm_batch_create_from_columns: NULL argument");
    const uint32_t ns = d->n_structures;
    const uint32_t na = d->atom_offsets[ns];
    if (na && (!d->xyz || !d->name || !d->residue_name || !d->chain_id || !d->residue_number || !d->insertion_code))
        return fail(TM_E_INVALID, "tm_batch_create_from_columns: NULL column");
    std::vector<tm_structure_columns> src(ns);
    for (uint32_t s = 0; s < ns; ++s) {
        const size_t o = d->atom_offsets[s];
        if (d->atom_offsets[s + 1] < o) return fail(TM_E_INVALID, "tm_batch_create_from_columns: offsets must not decrease");
        src[s].n_atoms = d->atom_offsets[s + 1] - o;
        src[s].xyz = d->xyz + 3 * o;
        src[s].name = d->name + 4 * o;
        src[s].residue_name = d->residue_name + 3 * o;
        src[s].chain_id = d->chain_id + 2 * o;
        src[s].residue_number = d->residue_number + o;
        src[s].insertion_code = d->insertion_code + o;
    }
    return batch_from_structures_refusing(device, ns, src.data(), vocab, out);
}

// exclusive prefix sum of n u32 values on the device; *total (device) receives the sum
static int device_scan(const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *total, DevBuf<uint32_t> &block_sums,
                       hipStream_t stream) {
    const uint64_t per_block = (uint64_t)kScanBlock * kScanItems;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, (n + per_block - 1) / per_block);
    HIP_TRY(block_sums.alloc(blocks));
    hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(kScanBlock), 0, stream, in, n, block_sums.p);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, block_sums.p, blocks, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(blocks), dim3(kScanBlock), 0, stream, in, n, block_sums.p, out);
    HIP_TRY(hipGetLastError());
    return TM_OK;
}

// Ingestion runs on a private non-blocking stream (one per host thread and device), so that a caller can parse
// the next batch on one thread while a query of the previous batch runs on another stream.
static hipError_t ingest_stream(int device, hipStream_t *out) {
    struct Streams {  // destroyed with the host thread that used them
        std::map<int, hipStream_t> by_device;
        ~Streams() {
            for (auto &kv : by_device) {
                if (hipSetDevice(kv.first) == hipSuccess) (void)hipStreamDestroy(kv.second);
            }
        }
    };
    thread_local Streams streams;
    auto it = streams.by_device.find(device);
    if (it == streams.by_device.end()) {
        hipStream_t st = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        it = streams.by_device.emplace(device, st).first;
    }
    *out = it->second;
    return hipSuccess;
}

}  // extern "C"

namespace {

// One PDB file laid out on the host exactly as k_pdb_parse / k_pdb_finalize lay out the others: used by
// tm_batch_create_from_pdb for the files its device path cannot handle.
struct HostFlattened {
    uint32_t file = 0, n_res = 0, residue_count = 0;
    std::vector<double> xyz;       // device (residue-contiguous) order
    std::vector<float4> rec;
    std::vector<uint8_t> chain;
    std::vector<uint32_t> orig;    // device position -> kept-record index in file order
    std::vector<uint16_t> res_start;
    std::vector<uint8_t> res_start_hi;  // bits 16-23 (a file of more than 65535 kept atoms)
};

}  // namespace

extern "C" int tm_pdb_count_host(const char *, uint64_t, uint32_t, uint64_t *, char[4], int *);
extern "C" int tm_pdb_parse_host(const char *, uint64_t, uint32_t, uint64_t, const tm_pdb_columns *, uint64_t *, uint64_t *);

static int flatten_pdb_on_host(const char *text, uint64_t length, uint32_t flags, double min_bfactor,
                               const std::vector<uint32_t> &v_names, const uint8_t *name_ids,
                               const std::vector<uint16_t> &v_el, const uint8_t *el_ids,
                               const std::vector<uint32_t> &v_res, const uint8_t *res_ids,
                               const std::vector<uint32_t> &v_counted, uint32_t expect_atoms, uint32_t file,
                               HostFlattened *out) {
    uint64_t total = 0;
    int rc = tm_pdb_count_host(text, length, flags, &total, nullptr, nullptr);
    if (rc != TM_OK) return rc;
    std::vector<int32_t> serial(total), resnum(total);
    std::vector<char> name(4 * total), altloc(total), resname(3 * total), chain_id(2 * total), icode(total),
        segment(4 * total), element(2 * total);
    std::vector<double> xyz(3 * total), occ(total), bfac(total);
    std::vector<int8_t> charge(total);
    tm_pdb_columns cols{serial.data(), name.data(), altloc.data(), resname.data(), chain_id.data(), resnum.data(),
                        icode.data(), xyz.data(), occ.data(), bfac.data(), segment.data(), element.data(), charge.data()};
    rc = tm_pdb_parse_host(text, length, flags, total, &cols, nullptr, nullptr);
    if (rc != TM_OK) return rc;
    // kept records in file order (Molecule.conserved: temperature factor >= cutoff)
    std::vector<uint32_t> kept;
    kept.reserve(total);
    for (uint64_t i = 0; i < total; ++i)
        if (!(min_bfactor > 0.0) || bfac[i] >= min_bfactor) kept.push_back((uint32_t)i);
    const uint32_t n = (uint32_t)kept.size();
    if (n != expect_atoms)
        return fail(TM_E_INVALID, "structure " + std::to_string(file) + ": host and device readers disagree (" +
                                      std::to_string(n) + " vs " + std::to_string(expect_atoms) + " atoms)");
    // residue and chain ordinals by first appearance, vocabulary ids, counted residue numbers
    std::map<uint64_t, uint32_t> by_key;
    std::vector<uint16_t> seen_chain;
    std::vector<uint32_t> residue(n);
    std::vector<uint8_t> chain(n), name_id(n), res_type(n);
    std::vector<bool> counted_seen(11000, false);
    uint32_t distinct_counted = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const size_t a = kept[j];
        uint32_t nm = 0, rn = 0;
        std::memcpy(&nm, &name[4 * a], 4);
        std::memcpy(&rn, &resname[3 * a], 3);
        uint32_t id = TM_NAME_OTHER;
        for (size_t q = 0; q < v_names.size(); ++q)
            if (v_names[q] == nm) { id = name_ids[q]; break; }
        if (id == TM_NAME_OTHER) {
            const uint16_t el = (uint16_t)(nm & 0xffffu);
            for (size_t q = 0; q < v_el.size(); ++q)
                if (v_el[q] == el) { id = el_ids[q]; break; }
        }
        name_id[j] = (uint8_t)id;
        uint32_t rt = TM_RES_OTHER;
        for (size_t q = 0; q < v_res.size(); ++q)
            if (v_res[q] == rn) { rt = res_ids[q]; break; }
        res_type[j] = (uint8_t)rt;
        for (size_t q = 0; q < v_counted.size(); ++q)
            if (v_counted[q] == rn) {
                const uint32_t v = (uint32_t)(resnum[a] + 999);
                if (v < 11000u && !counted_seen[v]) { counted_seen[v] = true; ++distinct_counted; }
                break;
            }
        const uint16_t ch = (uint16_t)((uint8_t)chain_id[2 * a] << 8 | (uint8_t)chain_id[2 * a + 1]);
        const uint64_t key = ((uint64_t)ch << 40) | ((uint64_t)(uint32_t)(resnum[a] + (1 << 24)) << 8) | (uint8_t)icode[a];
        auto it = by_key.find(key);
        if (it == by_key.end()) it = by_key.emplace(key, (uint32_t)by_key.size()).first;
        residue[j] = it->second;
        uint32_t c = 0;
        while (c < seen_chain.size() && seen_chain[c] != ch) ++c;
        if (c == seen_chain.size()) seen_chain.push_back(ch);
        chain[j] = (uint8_t)(c & 0xffu);
    }
    const uint32_t nres = (uint32_t)by_key.size();
    if (nres > TM_MAX_STRUCTURE_RESIDUES) {  // (the caller refuses this file alone, not its batch)
        out->n_res = nres;
        return TM_OK;
    }
    // stable counting sort by residue ordinal, centred fp32 records
    std::vector<uint32_t> count(nres + 1, 0), fill;
    for (uint32_t j = 0; j < n; ++j) count[residue[j] + 1]++;
    for (uint32_t r = 0; r < nres; ++r) count[r + 1] += count[r];
    out->n_res = nres;
    out->residue_count = distinct_counted;
    out->res_start.resize(nres + 1);
    out->res_start_hi.resize(nres + 1);
    for (uint32_t r = 0; r <= nres; ++r) { out->res_start[r] = (uint16_t)count[r]; out->res_start_hi[r] = (uint8_t)(count[r] >> 16); }
    fill.assign(count.begin(), count.end() - 1);
    out->orig.resize(n);
    for (uint32_t j = 0; j < n; ++j) out->orig[fill[residue[j]]++] = j;
    double c[3] = {0, 0, 0};
    for (uint32_t j = 0; j < n; ++j)
        for (int a = 0; a < 3; ++a) c[a] += xyz[3 * (size_t)kept[j] + a];
    if (n)
        for (int a = 0; a < 3; ++a) c[a] /= (double)n;
    out->xyz.resize(3 * (size_t)n); out->rec.resize(n); out->chain.resize(n);
    for (uint32_t d = 0; d < n; ++d) {
        const uint32_t j = out->orig[d];
        const size_t a = kept[j];
        for (int k = 0; k < 3; ++k) out->xyz[3 * (size_t)d + k] = xyz[3 * a + k];
        float4 r;
        r.x = (float)(xyz[3 * a] - c[0]); r.y = (float)(xyz[3 * a + 1] - c[1]); r.z = (float)(xyz[3 * a + 2] - c[2]);
        const uint32_t m = meta_pack(residue[j], name_id[j], res_type[j]);
        std::memcpy(&r.w, &m, 4);
        out->rec[d] = r;
        out->chain[d] = chain[j];
    }
    return TM_OK;
}

extern "C" {

// Waits for a stream when it goes out of scope while armed: declared AFTER the buffers whose copies are queued on the
// stream, it runs before they return to their pools on every early return.
struct StreamWaitGuard {
    hipStream_t s;
    bool armed;
    ~StreamWaitGuard() { if (armed) (void)hipStreamSynchronize(s); }
};

static int batch_from_pdb_text(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                               const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out,
                               std::vector<uint32_t> *oversized, std::vector<uint32_t> *oversized_atoms,
                               DevBuf<char> *uploaded = nullptr);
static int batch_from_pdb(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                          const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out, DevBuf<char> *uploaded);

// A file with more than TM_MAX_STRUCTURE_ATOMS kept atoms is found out only after the device has classified its lines:
// the batch is then laid out once more with that file's text left out -- it stays in the batch as an empty structure,
// listed by tm_batch_refused -- instead of failing the other files with it.
int tm_batch_create_from_pdb(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                             const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out) {
    return batch_from_pdb(device, n_files, text, length, vocab, flags, min_bfactor, out, nullptr);
}

// `uploaded`: the text as it already lies in device memory (tm_batch_create_from_files sends it up chunk by chunk while
// it reads the next files), copies queued on this thread's ingest stream
static int batch_from_pdb(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                          const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out, DevBuf<char> *uploaded) {
    if (!out || !vocab || (n_files && (!text || !length))) return fail(TM_E_INVALID, "tm_batch_create_from_pdb: NULL argument");
    std::vector<uint32_t> oversized, oversized_atoms;
    int rc = batch_from_pdb_text(device, n_files, text, length, vocab, flags, min_bfactor, out, &oversized, &oversized_atoms, uploaded);
    if (rc != TM_OK || oversized.empty()) return rc;
    std::vector<uint64_t> kept(length, length + n_files);
    for (uint32_t f : oversized) kept[f] = 0;
    rc = batch_from_pdb_text(device, n_files, text, kept.data(), vocab, flags, min_bfactor, out, nullptr, nullptr);
    if (rc == TM_OK) { (*out)->refused = std::move(oversized); (*out)->refused_atoms = std::move(oversized_atoms); }
    return rc;
}

static int batch_from_pdb_text(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                               const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out,
                               std::vector<uint32_t> *oversized, std::vector<uint32_t> *oversized_atoms,
                               DevBuf<char> *uploaded) {
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = nullptr;
    HIP_TRY(ingest_stream(device, &stream));
    PhaseTrace trace("tm_batch_create_from_pdb");

    // ---- one contiguous text buffer, every file newline-terminated, padded to 16-byte granules
    std::vector<uint64_t> file_off(n_files + 1, 0);
    for (uint32_t f = 0; f < n_files; ++f) {
        const bool need_nl = length[f] == 0 || text[f][length[f] - 1] != '\n';
        file_off[f + 1] = file_off[f] + length[f] + (need_nl ? 1 : 0);
    }
    const uint64_t n_bytes = (file_off[n_files] + 15) & ~15ull;
    if (n_bytes >= (1ull << 36)) return fail(TM_E_UNSUPPORTED, "more than 64 GiB of PDB text in one batch");
    // every file goes straight from the caller's buffer to its place in HBM (no host-side concatenation)
    DevBuf<char> d_text;
    DevBuf<uint64_t> d_file_off, d_nl;
    DevBuf<uint32_t> d_counts, d_rank, d_sums, d_total, d_line_file, d_first_end, d_keep, d_atom_rank, d_bad;
    DevBuf<uint8_t> d_kind;
    const bool have_text = uploaded && uploaded->p && uploaded->bytes >= n_bytes + 16;
    if (have_text) d_text.swap(*uploaded);
    else HIP_TRY(d_text.alloc(n_bytes + 16));
    // (the caller's chunk copies into d_text, or the copies queued below, may be in flight when a call further down fails:
    // the stream is waited for before d_text goes back to its pool; disarmed at the first wait of the normal path)
    StreamWaitGuard text_guard{stream, true};
    trace.mark("text alloc");
    HIP_TRY(hipMemsetAsync(d_text.p + file_off[n_files], ' ', n_bytes + 16 - file_off[n_files], stream));
    // files that already sit back to back in host memory, each ending in a newline (tm_batch_create_from_files
    // lays them out so), are the device layout: one copy.  Otherwise one copy per file.
    bool contiguous = n_files > 0;
    for (uint32_t f = 0; f < n_files && contiguous; ++f)
        contiguous = file_off[f + 1] - file_off[f] == length[f] && (f + 1 == n_files || text[f] + length[f] == text[f + 1]);
    if (have_text) {
        // (already on its way: tm_batch_create_from_files)
    } else if (contiguous) {
        if (file_off[n_files]) HIP_TRY(hipMemcpyAsync(d_text.p, text[0], file_off[n_files], hipMemcpyHostToDevice, stream));
    } else {
        for (uint32_t f = 0; f < n_files; ++f) {
            if (length[f]) HIP_TRY(hipMemcpyAsync(d_text.p + file_off[f], text[f], length[f], hipMemcpyHostToDevice, stream));
            if (file_off[f + 1] - file_off[f] != length[f])
                HIP_TRY(hipMemsetAsync(d_text.p + file_off[f + 1] - 1, '\n', 1, stream));
        }
    }
    HIP_TRY(d_file_off.upload(file_off, stream));
    const uint64_t n_gran = n_bytes / 16;
    HIP_TRY(d_counts.alloc(n_gran)); HIP_TRY(d_rank.alloc(n_gran)); HIP_TRY(d_total.alloc(2)); HIP_TRY(d_bad.alloc(1));
    HIP_TRY(hipMemsetAsync(d_bad.p, 0, sizeof(uint32_t), stream));
    uint32_t n_lines = 0, n_atoms = 0;
    if (n_gran) {
        hipLaunchKernelGGL(k_pdb_count_newlines, dim3(2048), dim3(256), 0, stream,
                           reinterpret_cast<const uint4 *>(d_text.p), n_gran, d_counts.p);
        int rc = device_scan(d_counts.p, n_gran, d_rank.p, d_total.p, d_sums, stream);
        if (rc != TM_OK) return rc;
        HIP_TRY(hipMemcpyAsync(&n_lines, d_total.p, sizeof n_lines, hipMemcpyDeviceToHost, stream));
        HIP_TRY(wait_blocking(stream));
        text_guard.armed = false;
        trace.mark("sync0");
    }
    HIP_TRY(d_nl.alloc(n_lines)); HIP_TRY(d_line_file.alloc(n_lines)); HIP_TRY(d_kind.alloc(n_lines));
    HIP_TRY(d_keep.alloc(n_lines)); HIP_TRY(d_atom_rank.alloc(n_lines));
    std::vector<uint32_t> first_end(std::max<uint32_t>(n_files, 1), 0xffffffffu);
    HIP_TRY(d_first_end.upload(first_end, stream));
    DevBuf<uint32_t> d_s_off_tmp;
    HIP_TRY(d_s_off_tmp.alloc(n_files + 1));
    if (n_lines) {
        const uint32_t lb = (n_lines + 255) / 256;
        hipLaunchKernelGGL(k_pdb_mark_lines, dim3(2048), dim3(256), 0, stream,
                           reinterpret_cast<const uint8_t *>(d_text.p), n_gran, d_rank.p, d_nl.p);
        hipLaunchKernelGGL(k_pdb_classify, dim3(lb), dim3(256), 0, stream, reinterpret_cast<const uint8_t *>(d_text.p),
                           d_nl.p, n_lines, d_file_off.p, n_files, (flags & TM_PDB_IGNORE_ENDMDL) ? 1u : 0u,
                           d_line_file.p, d_kind.p, d_first_end.p);
        hipLaunchKernelGGL(k_pdb_keep, dim3(lb), dim3(256), 0, stream, reinterpret_cast<const uint8_t *>(d_text.p),
                           d_nl.p, n_lines, d_line_file.p, d_kind.p, d_first_end.p, min_bfactor, d_keep.p, d_bad.p);
        HIP_TRY(hipGetLastError());
        int rc = device_scan(d_keep.p, n_lines, d_atom_rank.p, d_total.p + 1, d_sums, stream);
        if (rc != TM_OK) return rc;
        HIP_TRY(hipMemcpyAsync(&n_atoms, d_total.p + 1, sizeof n_atoms, hipMemcpyDeviceToHost, stream));
        HIP_TRY(wait_blocking(stream));
        trace.mark("sync1");
    }

    tm_batch *b = new (std::nothrow) tm_batch;
    if (!b) return fail(TM_E_NOMEM, "out of host memory");
    struct Guard { tm_batch *b; ~Guard() { delete b; } } guard{b};
    b->device = device; b->n_structs = n_files; b->n_atoms = n_atoms;
    DevBuf<uint32_t> d_meta, d_runs, d_needs_host;
    DevBuf<uint64_t> d_key;
    HIP_TRY(b->d_xyz.alloc(3 * (size_t)n_atoms)); HIP_TRY(d_meta.alloc(n_atoms)); HIP_TRY(d_key.alloc(n_atoms));
    HIP_TRY(b->d_line_off.alloc(n_atoms)); HIP_TRY(d_runs.alloc(n_files)); HIP_TRY(d_needs_host.alloc(n_files));
    HIP_TRY(hipMemsetAsync(d_needs_host.p, 0, std::max<uint32_t>(n_files, 1) * sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(d_runs.p, 0, std::max<uint32_t>(n_files, 1) * sizeof(uint32_t), stream));
    ParsedAtoms P{b->d_xyz.p, d_meta.p, d_key.p, b->d_line_off.p};

    // the vocabulary as integer tables
    std::vector<uint32_t> v_names(vocab->n_atom_names), v_res(vocab->n_residues);
    std::vector<uint16_t> v_el(vocab->n_elements);
    for (uint32_t i = 0; i < vocab->n_atom_names; ++i) std::memcpy(&v_names[i], vocab->atom_names + 4 * i, 4);
    for (uint32_t i = 0; i < vocab->n_elements; ++i) std::memcpy(&v_el[i], vocab->elements + 2 * i, 2);
    for (uint32_t i = 0; i < vocab->n_residues; ++i) { v_res[i] = 0; std::memcpy(&v_res[i], vocab->residue_names + 3 * i, 3); }
    DevBuf<uint32_t> dv_names, dv_res;
    DevBuf<uint16_t> dv_el;
    DevBuf<uint8_t> dv_name_ids, dv_el_ids, dv_res_ids;
    HIP_TRY(dv_names.upload(v_names, stream)); HIP_TRY(dv_res.upload(v_res, stream)); HIP_TRY(dv_el.upload(v_el, stream));
    HIP_TRY(dv_name_ids.upload(std::vector<uint8_t>(vocab->atom_name_ids, vocab->atom_name_ids + vocab->n_atom_names), stream));
    HIP_TRY(dv_el_ids.upload(std::vector<uint8_t>(vocab->element_ids, vocab->element_ids + vocab->n_elements), stream));
    HIP_TRY(dv_res_ids.upload(std::vector<uint8_t>(vocab->residue_ids, vocab->residue_ids + vocab->n_residues), stream));
    const uint32_t n_counted = vocab->counted_residue_names ? vocab->n_counted_residues : 0u;
    std::vector<uint32_t> v_counted(n_counted, 0u);
    for (uint32_t i = 0; i < n_counted; ++i) std::memcpy(&v_counted[i], vocab->counted_residue_names + 3 * i, 3);
    DevBuf<uint32_t> dv_counted;
    HIP_TRY(dv_counted.upload(v_counted, stream));
    VocabView V{vocab->n_atom_names, vocab->n_elements, vocab->n_residues, dv_names.p, dv_name_ids.p,
                dv_el.p, dv_el_ids.p, dv_res.p, dv_res_ids.p, n_counted, dv_counted.p};

    hipLaunchKernelGGL(k_pdb_file_offsets, dim3((n_files + 256) / 256), dim3(256), 0, stream, d_nl.p, n_lines,
                       d_file_off.p, n_files, d_atom_rank.p, d_total.p + 1, d_s_off_tmp.p);
    if (n_lines) {
        hipLaunchKernelGGL(k_pdb_parse, dim3((n_lines + 255) / 256), dim3(256), 0, stream,
                           reinterpret_cast<const uint8_t *>(d_text.p), d_nl.p, n_lines, d_keep.p, d_atom_rank.p,
                           d_line_file.p, d_file_off.p, V, P, d_bad.p);
    }
    if (n_files) {
        const size_t key_lds = (size_t)kMaxSortedRuns * sizeof(uint64_t);
        HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(k_pdb_residues), key_lds));
        hipLaunchKernelGGL(k_pdb_residues, dim3(n_files), dim3(256), key_lds, stream, d_s_off_tmp.p, d_key.p, d_runs.p,
                           d_needs_host.p);
    }
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> s_off(n_files + 1, 0), runs(n_files, 0), needs_host(n_files, 0);
    uint32_t bad = 0;
    if (n_lines == 0) {
        HIP_TRY(hipMemsetAsync(d_s_off_tmp.p, 0, (n_files + 1) * sizeof(uint32_t), stream));
    }
    HIP_TRY(hipMemcpyAsync(s_off.data(), d_s_off_tmp.p, (n_files + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (n_files) {
        HIP_TRY(hipMemcpyAsync(runs.data(), d_runs.p, n_files * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(needs_host.data(), d_needs_host.p, n_files * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipMemcpyAsync(&bad, d_bad.p, sizeof bad, hipMemcpyDeviceToHost, stream));
    HIP_TRY(wait_blocking(stream));
    trace.mark("sync2");
    if (bad) return fail(TM_E_INVALID, "malformed ATOM/HETATM record in the PDB text");
    std::vector<uint32_t> s_roff(n_files + 1, 0);
    uint32_t max_atoms = 0, max_res = 0;
    // Files the device path cannot lay out -- a residue whose records are not contiguous (its atoms have to be
    // gathered), or more residue runs than k_pdb_residues can check -- are flattened on the host, one by one, and
    // patched into the batch below; the rest of the batch stays on the device path.
    std::vector<HostFlattened> host_files;
    for (uint32_t f = 0; f < n_files; ++f) {
        const uint32_t n = s_off[f + 1] - s_off[f];
        if (n > TM_MAX_STRUCTURE_ATOMS) {
            if (!oversized)
                return fail(TM_E_UNSUPPORTED, "structure " + std::to_string(f) + " has " + std::to_string(n) + " atoms; limit " +
                                                  std::to_string(TM_MAX_STRUCTURE_ATOMS));
            oversized->push_back(f);
            oversized_atoms->push_back(n);
            continue;
        }
        if (oversized && !oversized->empty()) continue;  // (this layout is abandoned: the caller starts over)
        if (needs_host[f]) {
            host_files.emplace_back();
            HostFlattened &h = host_files.back();
            h.file = f;
            int rc = flatten_pdb_on_host(text[f], length[f], flags, min_bfactor, v_names, vocab->atom_name_ids, v_el,
                                         vocab->element_ids, v_res, vocab->residue_ids, v_counted, n, f, &h);
            if (rc != TM_OK) return rc;
            runs[f] = h.n_res;
            if (h.n_res > TM_MAX_STRUCTURE_RESIDUES) {
                // more residues than the kernels number (16 bits): refused like a file beyond the atom limit -- it stays
                // in the batch as an empty structure, the other files are searched
                if (!oversized)
                    return fail(TM_E_UNSUPPORTED, "structure " + std::to_string(f) + " has more than " +
                                                      std::to_string(TM_MAX_STRUCTURE_RESIDUES) + " residues");
                oversized->push_back(f);
                oversized_atoms->push_back(n);
                continue;
            }
        }
        s_roff[f + 1] = s_roff[f] + runs[f] + 1;
        max_atoms = std::max(max_atoms, n);
        max_res = std::max(max_res, runs[f]);
        // (a file of more than 65535 kept atoms came through the host path above: k_pdb_residues sends it there)
        if (n > TM_NARROW_STRUCTURE_ATOMS) b->has_wide = true;
        else { b->max_atoms_narrow = std::max(b->max_atoms_narrow, n); b->max_res_narrow = std::max(b->max_res_narrow, runs[f]); }
    }
    if (oversized && !oversized->empty()) return TM_OK;  // (*out is still NULL: tm_batch_create_from_pdb lays the batch out again)
    b->max_atoms = max_atoms; b->max_res = max_res; b->s_off = s_off;
    HIP_TRY(b->d_s_off.upload(s_off, stream)); HIP_TRY(b->d_s_roff.upload(s_roff, stream));
    HIP_TRY(b->d_rec.alloc(n_atoms)); HIP_TRY(b->d_chain.alloc(n_atoms)); HIP_TRY(b->d_orig.alloc(n_atoms));
    HIP_TRY(b->d_res_start.alloc(s_roff[n_files]));
    if (b->has_wide) {
        HIP_TRY(b->d_res_start_hi.alloc(s_roff[n_files]));
        HIP_TRY(hipMemsetAsync(b->d_res_start_hi.p, 0, s_roff[n_files], stream));  // (zero for every file the device finalises)
    }
    HIP_TRY(b->d_cursor.alloc(1)); HIP_TRY(b->d_overflow.alloc(1));
    HIP_TRY(b->d_residue_count.alloc(n_files));
    b->has_residue_count = n_counted != 0;
    if (n_files) {
        hipLaunchKernelGGL(k_pdb_finalize, dim3(n_files), dim3(256), 0, stream, b->d_s_off.p, b->d_s_roff.p, P,
                           b->d_rec.p, b->d_chain.p, b->d_orig.p, b->d_res_start.p, b->d_residue_count.p, d_needs_host.p);
        HIP_TRY(hipGetLastError());
    }
    for (const HostFlattened &h : host_files) {  // same stream: ordered after the kernels that wrote these ranges
        const size_t a0 = s_off[h.file], n = h.rec.size();
        if (n) {
            HIP_TRY(hipMemcpyAsync(b->d_xyz.p + 3 * a0, h.xyz.data(), 3 * n * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(b->d_rec.p + a0, h.rec.data(), n * sizeof(float4), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(b->d_chain.p + a0, h.chain.data(), n, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(b->d_orig.p + a0, h.orig.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        }
        HIP_TRY(hipMemcpyAsync(b->d_res_start.p + s_roff[h.file], h.res_start.data(), h.res_start.size() * sizeof(uint16_t),
                               hipMemcpyHostToDevice, stream));
        if (b->has_wide)
            HIP_TRY(hipMemcpyAsync(b->d_res_start_hi.p + s_roff[h.file], h.res_start_hi.data(), h.res_start_hi.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(b->d_residue_count.p + h.file, &h.residue_count, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    }
    // the records' byte offsets also go to the host (the other direction of the link, idle during ingestion): reporting
    // a match needs the text of its atoms' records, and the main thread then finds them without asking the device
    if (n_atoms) {
        HIP_TRY(b->h_line_off.alloc(n_atoms));
        HIP_TRY(hipMemcpyAsync(b->h_line_off.p, b->d_line_off.p, (size_t)n_atoms * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    // the batch's extent (tm_batch::extent), measured where the records are
    DevBuf<uint32_t> d_extent;
    PinnedBuf<uint32_t> h_extent;
    HIP_TRY(d_extent.alloc(1)); HIP_TRY(h_extent.alloc(1));
    h_extent[0] = 0;
    HIP_TRY(hipMemsetAsync(d_extent.p, 0, sizeof(uint32_t), stream));
    if (n_atoms) {
        hipLaunchKernelGGL(k_extent, dim3(std::min<uint32_t>(1024u, (uint32_t)((n_atoms + 255) / 256))), dim3(256), 0, stream,
                           b->d_rec.p, (uint32_t)n_atoms, d_extent.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_extent.p, d_extent.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(wait_blocking(stream));
    std::memcpy(&b->extent, h_extent.p, 4);
    trace.mark("sync3");
    guard.b = nullptr;
    *out = b;
    return TM_OK;
}

int tm_batch_refused(const tm_batch *b, uint32_t *structures, uint32_t *atoms, uint32_t capacity, uint32_t *count) {
    if (!b || !count) return fail(TM_E_INVALID, "NULL argument");
    *count = (uint32_t)b->refused.size();
    for (uint32_t i = 0; i < capacity && i < b->refused.size(); ++i) {
        if (structures) structures[i] = b->refused[i];
        if (atoms) atoms[i] = b->refused_atoms[i];
    }
    return TM_OK;
}

int tm_batch_atom_counts(tm_batch *b, uint32_t *atom_offsets) {
    if (!b || !atom_offsets) return fail(TM_E_INVALID, "NULL argument");
    std::memcpy(atom_offsets, b->s_off.data(), b->s_off.size() * sizeof(uint32_t));
    return TM_OK;
}

int tm_batch_upload_bytes(const tm_batch *b, uint64_t *bytes) {
    if (!b || !bytes) return fail(TM_E_INVALID, "NULL argument");
    *bytes = b->upload_bytes;
    return TM_OK;
}

int tm_batch_wait(tm_batch *b) {
    if (!b) return fail(TM_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(b->device));
    {
        std::lock_guard<std::mutex> lock(b->ready_mutex);
        if (b->ready) {
            HIP_TRY(hipEventSynchronize(b->ready));
            b->settle_locked();
        }
    }
    return TM_OK;
}

int tm_batch_copy_layout(tm_batch *b, double *xyz, float *records, uint8_t *chain, uint32_t *orig, uint32_t *residue_offsets,
                         uint16_t *residue_start) {
    if (!b) return fail(TM_E_INVALID, "NULL argument");
    int rc = tm_batch_wait(b);
    if (rc != TM_OK) return rc;
    if (b->queried) HIP_TRY(hipStreamSynchronize(b->last_stream));
    const size_t na = b->n_atoms, ns = b->n_structs;
    std::vector<uint32_t> roff(ns + 1, 0);
    HIP_TRY(hipMemcpy(roff.data(), b->d_s_roff.p, (ns + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (residue_offsets) std::memcpy(residue_offsets, roff.data(), (ns + 1) * sizeof(uint32_t));
    if (xyz && na) HIP_TRY(hipMemcpy(xyz, b->d_xyz.p, 24 * na, hipMemcpyDeviceToHost));
    if (records && na) HIP_TRY(hipMemcpy(records, b->d_rec.p, 16 * na, hipMemcpyDeviceToHost));
    if (chain && na) HIP_TRY(hipMemcpy(chain, b->d_chain.p, na, hipMemcpyDeviceToHost));
    if (orig && na) HIP_TRY(hipMemcpy(orig, b->d_orig.p, 4 * na, hipMemcpyDeviceToHost));
    if (residue_start && roff[ns]) HIP_TRY(hipMemcpy(residue_start, b->d_res_start.p, 2 * (size_t)roff[ns], hipMemcpyDeviceToHost));
    return TM_OK;
}

int tm_batch_line_offsets(tm_batch *b, uint32_t *out, uint64_t n) {
    if (!b || (!out && n)) return fail(TM_E_INVALID, "NULL argument");
    if (b->d_line_off.n != b->n_atoms) return fail(TM_E_INVALID, "batch was not created from PDB text");
    if (n < b->n_atoms) return fail(TM_E_INVALID, "output buffer too small");
    HIP_TRY(hipSetDevice(b->device));
    if (b->n_atoms) HIP_TRY(hipMemcpy(out, b->d_line_off.p, b->n_atoms * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TM_OK;
}

int tm_batch_line_offsets_at(tm_batch *b, const uint32_t *records, uint64_t n, uint32_t *out) {
    if (!b || (n && (!records || !out))) return fail(TM_E_INVALID, "NULL argument");
    if (b->d_line_off.n != b->n_atoms) return fail(TM_E_INVALID, "batch was not created from PDB text");
    if (n == 0) return TM_OK;
    if (n > 0xffffffffull) return fail(TM_E_UNSUPPORTED, "more than 2^32 records");
    for (uint64_t i = 0; i < n; ++i)
        if (records[i] >= b->n_atoms) return fail(TM_E_INVALID, "record index out of range");
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t stream = nullptr;
    HIP_TRY(ingest_stream(b->device, &stream));
    DevBuf<uint32_t> d_idx, d_out;
    PinnedBuf<uint32_t> h;
    HIP_TRY(d_idx.alloc(n)); HIP_TRY(d_out.alloc(n)); HIP_TRY(h.alloc(n));
    std::memcpy(h.p, records, n * sizeof(uint32_t));
    HIP_TRY(hipMemcpyAsync(d_idx.p, h.p, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_gather_u32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, b->d_line_off.p, d_idx.p, (uint32_t)n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h.p, d_out.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(out, h.p, n * sizeof(uint32_t));
    return TM_OK;
}

int tm_batch_residue_counts(tm_batch *b, uint32_t *out) {
    if (!b || (!out && b->n_structs)) return fail(TM_E_INVALID, "NULL argument");
    if (!b->has_residue_count)
        return fail(TM_E_UNSUPPORTED, "batch was not created from PDB text with tm_vocab.counted_residue_names");
    HIP_TRY(hipSetDevice(b->device));
    if (b->n_structs)
        HIP_TRY(hipMemcpy(out, b->d_residue_count.p, b->n_structs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TM_OK;
}

void tm_batch_destroy(tm_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    // its buffers go back to the allocation cache: nothing may still read them.  Only queries read a batch
    // after its creation returned, so waiting for the last query's stream is enough (a device-wide wait would
    // also wait for another thread's ingestion of the next batch).
    if (b->queried) (void)hipStreamSynchronize(b->last_stream);
    if (b->ready) (void)hipEventSynchronize(b->ready);  // (a batch destroyed before its upload has finished)
    delete b;
}

// (Re)builds the batch's candidate index for `lib` on `stream` without waiting for it.  The kind-list pool is
// sized generously (32 entries per atom; a protein needs about 3) or to what an earlier build reported; should it
// overflow all the same, the structures that did not fit get empty lists, bit 3 of the query's overflow word is
// set, and tm_result_wait reports TM_E_OVERFLOW so that the caller queries again (the pool then has the size the
// failed build asked for).
static int build_index(const tm_library *lib, tm_batch *b, tm_result *r, hipStream_t stream) {
    const size_t off_n = (size_t)b->n_structs * (lib->n_lead + 1);
    if (b->d_kl_off.n != off_n) HIP_TRY(b->d_kl_off.alloc(off_n));
    unsigned long long cap = b->pool_capacity ? b->pool_capacity : std::max<unsigned long long>(1024, 32ull * b->n_atoms);
    if (!b->pool_capacity)
        if (const char *env = std::getenv("ENZYMM_AMD_INDEX_POOL"))  // test hook: start from a pool that is too small
            cap = std::max<unsigned long long>(64, std::strtoull(env, nullptr, 10));
    const unsigned long long wanted = b->index_shared->pool_wanted.load();
    b->index_shared->stale = false;
    if (wanted > cap) cap = wanted + wanted / 8 + 1024;
    if (cap > 0xffffffffull)
        return fail(TM_E_UNSUPPORTED, "candidate index needs more than 2^32 entries: split the batch");
    if (b->d_kl_pool.n != cap + 8) {  // (+8: k_search copies a structure's lists in 16-byte words, the last one in full)
        if (b->queried) HIP_TRY(hipStreamSynchronize(b->last_stream));  // an earlier query may still read the old pool
        HIP_TRY(b->d_kl_pool.alloc(cap + 8));
    }
    if (b->has_wide && b->d_kl_pool_hi.n != cap + 8) {
        if (b->queried) HIP_TRY(hipStreamSynchronize(b->last_stream));
        HIP_TRY(b->d_kl_pool_hi.alloc(cap + 8));
    }
    b->pool_capacity = cap;
    {
        ZeroList z = {};
        z.p[0] = b->d_cursor.p; z.words[0] = 2;
        z.p[1] = b->d_overflow.p; z.words[1] = 1;
        hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, stream, z);
        HIP_TRY(hipGetLastError());
    }
    if (b->n_structs && lib->n_lead) {
        const uint32_t cap_atoms = (std::min<uint32_t>(b->max_atoms, kKindSeg) + 1u) & ~1u;
        const size_t lds = kind_lds_bytes(lib->n_lead, cap_atoms);
        uint8_t *pool_hi = b->has_wide ? b->d_kl_pool_hi.p : nullptr;
        HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(k_kind_lists<kKindWhole>), lds));
        hipLaunchKernelGGL(k_kind_lists<kKindWhole>, dim3(b->n_structs), dim3(512), lds, stream, lib->view(), b->view(),
                           (const KindSeg *)nullptr, (uint32_t *)nullptr, b->d_kl_off.p, b->d_kl_pool.p, pool_hi,
                           b->d_cursor.p, cap, b->d_overflow.p, cap_atoms);
        HIP_TRY(hipGetLastError());
        if (b->max_atoms > kKindSeg) {
            // long structures (large complexes): segments of kKindSeg atoms, counted, placed and written by three launches
            if (!b->kind_segs_made) {
                std::vector<KindSeg> segs;
                std::vector<KindLong> longs;
                for (uint32_t s = 0; s < b->n_structs; ++s) {
                    const uint32_t n = b->s_off[s + 1] - b->s_off[s];
                    if (n <= kKindSeg) continue;
                    longs.push_back({s, (uint32_t)segs.size(), (n + kKindSeg - 1) / kKindSeg});
                    for (uint32_t lo = 0; lo < n; lo += kKindSeg) segs.push_back({s, lo});
                }
                HIP_TRY(b->h_kind_segs.alloc(segs.size()));
                std::copy(segs.begin(), segs.end(), b->h_kind_segs.p);
                HIP_TRY(b->d_kind_segs.upload_async(b->h_kind_segs.p, segs.size(), stream));  // no host wait: see build_tables
                HIP_TRY(b->h_kind_longs.alloc(longs.size()));
                std::copy(longs.begin(), longs.end(), b->h_kind_longs.p);
                HIP_TRY(b->d_kind_longs.upload_async(b->h_kind_longs.p, longs.size(), stream));
                b->n_kind_segs = (uint32_t)segs.size(); b->n_kind_longs = (uint32_t)longs.size();
                b->kind_segs_made = true;
            }
            const size_t cnt_n = (size_t)b->n_kind_segs * lib->n_lead;
            if (b->d_kind_seg_cnt.n < cnt_n) {
                if (b->queried) HIP_TRY(hipStreamSynchronize(b->last_stream));
                HIP_TRY(b->d_kind_seg_cnt.alloc(cnt_n));
            }
            const size_t lds_seg = kind_lds_bytes(lib->n_lead, kKindSeg);
            HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(k_kind_lists<kKindCount>), lds_seg));
            HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(k_kind_lists<kKindWrite>), lds_seg));
            hipLaunchKernelGGL(k_kind_lists<kKindCount>, dim3(b->n_kind_segs), dim3(512), lds_seg, stream, lib->view(), b->view(),
                               (const KindSeg *)b->d_kind_segs.p, b->d_kind_seg_cnt.p, b->d_kl_off.p, b->d_kl_pool.p, pool_hi,
                               b->d_cursor.p, cap, b->d_overflow.p, kKindSeg);
            HIP_TRY(hipGetLastError());
            if (b->d_kind_unsearchable.n < b->n_kind_longs) HIP_TRY(b->d_kind_unsearchable.alloc(b->n_kind_longs));
            hipLaunchKernelGGL(k_kind_seg_offsets, dim3(b->n_kind_longs), dim3(256), ((size_t)lib->n_lead + 1) * sizeof(uint32_t), stream,
                               lib->n_lead, (const KindLong *)b->d_kind_longs.p, b->d_kind_seg_cnt.p, b->d_kl_off.p,
                               b->d_cursor.p, cap, b->d_overflow.p, b->d_kind_unsearchable.p);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_kind_lists<kKindWrite>, dim3(b->n_kind_segs), dim3(512), lds_seg, stream, lib->view(), b->view(),
                               (const KindSeg *)b->d_kind_segs.p, b->d_kind_seg_cnt.p, b->d_kl_off.p, b->d_kl_pool.p, pool_hi,
                               b->d_cursor.p, cap, b->d_overflow.p, kKindSeg);
            HIP_TRY(hipGetLastError());
        }
    }
    b->index_for = lib->serial;
    return TM_OK;
}

// The outcome of the index build the query searches -- pool entries needed, overflow bit -- travels with EVERY query
// that uses that index, not only with the one that launched the build: a second query issued before the first one
// was waited for reuses the index, and must report the same TM_E_OVERFLOW instead of returning with hits missing.
static int carry_build_outcome(tm_batch *b, tm_result *r, hipStream_t stream) {
    hipLaunchKernelGGL(k_carry_outcome, dim3(1), dim3(1), 0, stream, b->d_cursor.p, b->d_overflow.p,
                       b->d_tab_cursors.p, b->d_tab_overflow.p, r->d_pool_used.p, r->d_tab_used.p, r->d_overflow.p);
    HIP_TRY(hipGetLastError());
    r->pool_capacity = b->pool_capacity;
    r->row_capacity = b->row_capacity; r->ent_capacity = b->ent_capacity;
    return TM_OK;
}

// (Re)builds the batch's pair tables for selection `S` on `stream`, like the candidate index without waiting: the
// pools are sized generously (or to what an earlier build reported); should one overflow, the structures that did
// not fit are skipped by the search, bit 4 of the query's overflow word is set and tm_result_wait reports
// TM_E_OVERFLOW so that the caller queries again with room.
// ENZYMM_AMD_SYNC_EACH=1 (debugging): wait for every kernel of a query before the next is launched and say so on stderr --
// a GPU fault then aborts the process right after the line of the last kernel that ran to its end.
static void sync_each(hipStream_t stream, const char *what) {
    static const bool on = std::getenv("ENZYMM_AMD_SYNC_EACH") != nullptr;
    if (!on) return;
    std::fprintf(stderr, "[sync] %s launched\n", what);
    std::fflush(stderr);
    (void)hipStreamSynchronize(stream);
    std::fprintf(stderr, "[sync] %s done\n", what);
    std::fflush(stderr);
}

static int build_tables(const tm_library *lib, tm_batch *b, const StagedSelection *S, hipStream_t stream) {
    const uint32_t nt = (uint32_t)S->tables.size(), ns = b->n_structs;
    b->n_tables = nt;
    const size_t dir_n = (size_t)ns * nt;
    if (b->queried && (b->d_tab_dir.n != dir_n || b->d_tab_ok.n != ns)) HIP_TRY(hipStreamSynchronize(b->last_stream));
    if (b->d_tab_dir.n != dir_n) { HIP_TRY(b->d_tab_dir.alloc(dir_n)); HIP_TRY(b->d_tab_seg.alloc(dir_n)); }
    if (b->d_tab_ok.n != ns) HIP_TRY(b->d_tab_ok.alloc(ns));
    if (!b->d_tab_cursors.p) { HIP_TRY(b->d_tab_cursors.alloc(2)); HIP_TRY(b->d_tab_overflow.alloc(1)); }
    unsigned long long rows = b->row_capacity ? b->row_capacity
                                              : std::max<unsigned long long>(4096, 16ull * b->n_atoms + (unsigned long long)ns * (nt + 8));
    // a structure's tables are dealt to `parts` workgroups, parts growing with the square of its atom count (a
    // 300-residue structure is one item); a batch too small to fill the GPU with one workgroup per structure (a single
    // query structure, say) has every structure's tables dealt to more workgroups
    double spread = ns < 256 ? std::ceil(256.0 / std::max(1u, ns)) : 1.0;
    if (const char *env = std::getenv("ENZYMM_AMD_TAB_PARTS")) spread = std::max(spread, (double)std::atoi(env));  // experiments: workgroups per structure at least
    auto parts_of = [&](uint32_t s) {
        const double n = (double)(b->s_off[s + 1] - b->s_off[s]);
        return (uint32_t)std::min<double>(std::min<double>(64.0, std::max(1u, nt / 4u)),
                                          std::max(spread, std::floor(n * n / (3000.0 * 3000.0) + 0.5)));
    };
    // entries: about 35 per atom, plus what the build's per-wave chunks leave unused -- every one of the 16 waves of
    // every workgroup that builds a share of the structure's tables ends on a partly used chunk
    unsigned long long ents = b->ent_capacity;
    if (!ents) {
        ents = 65536 + 48ull * b->n_atoms;
        for (uint32_t s = 0; s < ns; ++s) ents += (20ull * parts_of(s)) * tab_chunk(b->s_off[s + 1] - b->s_off[s]);
    }
    if (!b->ent_capacity)
        if (const char *env = std::getenv("ENZYMM_AMD_TABLE_POOL"))  // test hook: start from a pool that is too small
            ents = std::max<unsigned long long>(64, std::strtoull(env, nullptr, 10));
    const unsigned long long rows_wanted = b->index_shared->rows_wanted.load(), ents_wanted = b->index_shared->ents_wanted.load();
    if (rows_wanted > rows) rows = rows_wanted + rows_wanted / 8 + 1024;
    if (ents_wanted > ents) ents = ents_wanted + ents_wanted / 8 + 1024;
    if (rows > 0xfffffff0ull || ents > 0xfffffff0ull)
        return fail(TM_E_UNSUPPORTED, "pair tables need more than 2^32 entries: split the batch");
    // 64 slots of slack: a search prefetches 64 row slots / entries from a table's start, whatever its size
    const size_t ref_n = (size_t)ns * S->sel.size();
    if (b->d_tab_rows.n != rows + 64 || b->d_tab_ent.n != ents + 64 || b->d_tab_ref.n != ref_n) {
        if (b->queried) HIP_TRY(hipStreamSynchronize(b->last_stream));  // an earlier query may still read the old pools
        if (b->d_tab_rows.n != rows + 64) HIP_TRY(b->d_tab_rows.alloc(rows + 64));
        if (b->d_tab_ent.n != ents + 64) HIP_TRY(b->d_tab_ent.alloc(ents + 64));
        if (b->d_tab_ref.n != ref_n) HIP_TRY(b->d_tab_ref.alloc(ref_n));
    }
    b->row_capacity = rows; b->ent_capacity = ents;
    {
        ZeroList z = {};
        z.p[0] = b->d_tab_cursors.p; z.words[0] = 4;
        z.p[1] = b->d_tab_overflow.p; z.words[1] = 1;
        z.p[2] = b->d_tab_ok.p; z.words[2] = std::max<uint32_t>(ns, 1);
        hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, stream, z);
        HIP_TRY(hipGetLastError());
    }
    b->tables_for = S->serial;
    if (!ns) return TM_OK;
    if (!nt) {  // no template of the selection has two residues: nothing to build, every structure is searchable
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->d_tab_ok.p, 1, ns, stream));
        HIP_TRY(hipMemsetAsync(b->d_tab_ref.p, 0, std::max<size_t>(ref_n, 1) * sizeof(uint4), stream));
        return TM_OK;
    }
    const BatchView B = b->view();
    // LDS of a build workgroup: the atom record of every kind-list position of its structure (about 2.5 per atom)
    // (a batch of mostly large structures would not use the copy: without it two workgroups fit a CU and hide the
    // latency of reading the records through L2 twice as well)
    const bool mostly_large = (size_t)b->n_atoms > (size_t)3800 * ns;
    // (such a batch still reserves 4096 records = 64 KB, two workgroups per CU: the staging area for list B of the
    // big tables its workgroups build together)
    // (sized for the largest NARROW structure: a wide one reads its records through L2 inside whichever layout runs)
    uint32_t lr_cap = mostly_large ? 4096u : (uint32_t)std::min<size_t>(9600, std::max<size_t>(1024, 3 * (size_t)b->max_atoms_narrow));
    size_t lds_bytes = (size_t)lr_cap * sizeof(float4) + ((size_t)lib->n_lead + 2) * sizeof(uint32_t);  // s_lr + s_kl
    // Compact layout -- atom records once (16 B per atom) and the kind lists as positions (2 B each) instead of a record
    // per position: when the batch's largest structure then leaves room for TWO workgroups per CU (80 KB each), the
    // build runs at eight waves per SIMD instead of four, which is worth the second LDS read per operand.
    uint32_t atom_cap = 0;
    {
        const char *compact_env = std::getenv("ENZYMM_AMD_TAB_COMPACT");  // experiments and tests: 0 = a record per list position
        const bool off = compact_env && std::atoi(compact_env) == 0;
        const size_t budget = 80 * 1024 - 512;  // static LDS of the kernel and allocation granularity
        const size_t fixed = (size_t)b->max_atoms_narrow * sizeof(float4) + ((size_t)lib->n_lead + 2) * sizeof(uint32_t);
        if (!off && !mostly_large && b->max_atoms_narrow && fixed + 2 * 2 * (size_t)b->max_atoms_narrow + 16 <= budget) {  // room for two positions per atom at least
            atom_cap = b->max_atoms_narrow;
            lr_cap = (uint32_t)std::min<size_t>(65528, ((budget - fixed) / sizeof(uint16_t)) & ~size_t(7));
            lds_bytes = (size_t)atom_cap * sizeof(float4) + (size_t)lr_cap * sizeof(uint16_t) + ((size_t)lib->n_lead + 2) * sizeof(uint32_t);
        }
    }
    unsigned long long shared_limit = kTabShared;
    if (const char *env = std::getenv("ENZYMM_AMD_TAB_SHARED"))  // test hook: smaller tables take the shared build too
        shared_limit = std::max<unsigned long long>(1, std::strtoull(env, nullptr, 10));
    unsigned long long derive_min = kDeriveMin;
    if (const char *env = std::getenv("ENZYMM_AMD_TAB_DERIVE_MIN"))  // test hook: 0 = every grouped table derived, huge = every table swept
        derive_min = std::strtoull(env, nullptr, 10);
    uint32_t lr_arg = lr_cap;
#ifdef TM_ABLATE
    if (const char *env = std::getenv("ENZYMM_AMD_TAB_DEBUG")) lr_arg |= (uint32_t)std::atoi(env) << 30;
#endif
    // LONG structures -- more atoms than the LDS layout holds -- are built by the large-structure instantiation (a launch of
    // their own beside the batch's layout, or the only one when the batch is mostly large), from position records and
    // chunk spheres gathered first (k_tab_gather; ENZYMM_AMD_TAB_CULL=0: records through the kind lists, plain sweeps)
    const uint32_t long_from = mostly_large ? 0u : atom_cap != 0u ? atom_cap : 4096u;
    const bool cull_on = [] { const char *e = std::getenv("ENZYMM_AMD_TAB_CULL"); return !(e && std::atoi(e) == 0); }();  // (read per build: tests switch it)
    // work items of the build: one per (structure, part); the long structures' items behind the others
    if (b->tab_work_tables != nt || b->tab_work_long_from != long_from || b->tab_work_n_lead != lib->n_lead) {  // (the sphere regions hold a slot per lead kind)
        std::vector<TabWork> items, items_long;
        std::vector<TabGather> gather;
        uint64_t lr_total = 0, sph_total = 0;
        for (uint32_t s = 0; s < ns; ++s) {
            const uint32_t parts = parts_of(s), n = b->s_off[s + 1] - b->s_off[s];
            const bool is_long = n > long_from;
            for (uint32_t r = 0; r < parts; ++r) (is_long ? items_long : items).push_back(TabWork{s, (uint16_t)r, (uint16_t)parts});
            if (is_long && n) {
                const uint64_t cap_pos = std::min<uint64_t>(6ull * n + 4096ull, 0xffffff00ull);
                gather.push_back(TabGather{s, (uint32_t)lr_total, (uint32_t)sph_total, (uint32_t)cap_pos});
                lr_total += cap_pos;
                sph_total += (cap_pos >> 4) + lib->n_lead + 2;
            }
        }
        if (lr_total > 0xfffffff0ull) { gather.clear(); lr_total = sph_total = 0; }  // (beyond 32-bit bases: built the old way)
        // heaviest structures first
        auto heavier = [&](const TabWork &x, const TabWork &y) {
            return b->s_off[x.s + 1] - b->s_off[x.s] > b->s_off[y.s + 1] - b->s_off[y.s];
        };
        std::stable_sort(items.begin(), items.end(), heavier);
        std::stable_sort(items_long.begin(), items_long.end(), heavier);
        b->tab_work_small = (uint32_t)items.size();
        items.insert(items.end(), items_long.begin(), items_long.end());
        // (a plain hipMemcpy from the vector would wait for everything enqueued so far -- the candidate index of this
        // very query: the list goes through a page-locked copy owned by the batch, asynchronously on the stream)
        if (b->d_tab_work.p) HIP_TRY(hipStreamSynchronize(stream));  // a previous query may still read the old list
        if (b->d_tab_work.p && b->queried && b->last_stream != stream) HIP_TRY(hipStreamSynchronize(b->last_stream));
        HIP_TRY(b->h_tab_work.alloc(items.size()));
        std::copy(items.begin(), items.end(), b->h_tab_work.p);
        HIP_TRY(b->d_tab_work.upload_async(b->h_tab_work.p, items.size(), stream));
        HIP_TRY(b->h_tab_gather.alloc(gather.size()));
        std::copy(gather.begin(), gather.end(), b->h_tab_gather.p);
        HIP_TRY(b->d_tab_gather.upload_async(b->h_tab_gather.p, gather.size(), stream));
        b->n_tab_gather = (uint32_t)gather.size();
        if (!gather.empty()) {
            HIP_TRY(b->d_tab_lr.alloc(lr_total)); HIP_TRY(b->d_tab_sph.alloc(sph_total));
            HIP_TRY(b->d_tab_lr_base.alloc(ns)); HIP_TRY(b->d_tab_sph_base.alloc(ns));
        }
        b->tab_work_items = (uint32_t)items.size(); b->tab_work_tables = nt; b->tab_work_long_from = long_from; b->tab_work_n_lead = lib->n_lead;
    }
    TabAux aux = {};
    hipLaunchKernelGGL(k_tab_layout, dim3(ns), dim3(256), 0, stream, lib->n_lead, B, S->d_tables.p, b->d_tab_dir.p,
                       b->d_tab_cursors.p, rows, b->d_tab_overflow.p, b->d_tab_ok.p);
    sync_each(stream, "k_tab_layout");
    if (cull_on && b->n_tab_gather) {
        hipLaunchKernelGGL(k_tab_gather, dim3(b->n_tab_gather, 32), dim3(256), 0, stream, lib->n_lead, B,
                           (const TabGather *)b->d_tab_gather.p, b->d_tab_lr.p, b->d_tab_sph.p, b->d_tab_lr_base.p, b->d_tab_sph_base.p);
        sync_each(stream, "k_tab_gather");
        aux.lr = b->d_tab_lr.p; aux.sph = b->d_tab_sph.p; aux.lr_base = b->d_tab_lr_base.p; aux.sph_base = b->d_tab_sph_base.p;
    }
    if (b->tab_work_small) {
        const auto tab_build = atom_cap != 0u ? k_tab_build<kTabCompact> : k_tab_build<kTabRecords>;
        HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(tab_build), lds_bytes));
        hipLaunchKernelGGL(tab_build, dim3(b->tab_work_small), dim3(kTabBlock), lds_bytes, stream, lib->n_lead, B,
                           S->d_tables.p, b->d_tab_work.p, b->d_tab_ok.p, b->d_tab_rows.p, b->d_tab_ent.p, b->d_tab_seg.p,
                           b->d_tab_cursors.p, ents, b->d_tab_overflow.p, lr_arg, shared_limit, atom_cap, derive_min, TabAux{});
        sync_each(stream, "k_tab_build");
    }
    if (b->tab_work_items > b->tab_work_small) {
        // (4096 records = 64 KB, two workgroups per CU: the staging area for list B of the big tables a structure that was
        // not gathered has its workgroups build together)
        const uint32_t stage = 4096u;
        const size_t lds_long = (size_t)stage * sizeof(float4) + ((size_t)lib->n_lead + 2) * sizeof(uint32_t);
        HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(k_tab_build<kTabLarge>), lds_long));
        hipLaunchKernelGGL(k_tab_build<kTabLarge>, dim3(b->tab_work_items - b->tab_work_small), dim3(kTabBlock), lds_long, stream,
                           lib->n_lead, B, S->d_tables.p, b->d_tab_work.p + b->tab_work_small, b->d_tab_ok.p, b->d_tab_rows.p,
                           b->d_tab_ent.p, b->d_tab_seg.p, b->d_tab_cursors.p, ents, b->d_tab_overflow.p,
                           stage | (lr_arg & 0xc0000000u), shared_limit, 0u, derive_min, aux);
        sync_each(stream, "k_tab_build (long structures)");
    }
    {
        const uint32_t n_sel = (uint32_t)S->sel.size();
        hipLaunchKernelGGL(k_tab_refs, dim3(ns, (n_sel + 255) / 256), dim3(256), ((size_t)lib->n_lead + 15) & ~size_t(15), stream, B, lib->n_lead, n_sel, S->d_pair_table.p,
                           S->d_triple_table.p, S->d_lead8.p, b->d_tab_ref.p);
        sync_each(stream, "k_tab_refs");
    }
    HIP_TRY(hipGetLastError());
    return TM_OK;
}

int tm_query(const tm_library *lib, tm_batch *b, const tm_query_params *p, tm_result **out) {
    if (!lib || !b || !p || !out) return fail(TM_E_INVALID, "tm_query: NULL argument");
    PhaseTrace trace;
    *out = nullptr;
    if (lib->device != b->device) return fail(TM_E_INVALID, "library and batch live on different devices");
    if (p->n_selected && (!p->template_index || !p->rmsd_threshold || !p->distance_cutoff || !p->max_dynamic_distance))
        return fail(TM_E_INVALID, "tm_query: NULL parameter array");
    for (uint32_t i = 0; i < p->n_selected; ++i)
        if (p->template_index[i] >= lib->n_templates) return fail(TM_E_INVALID, "template index out of range");
    HIP_TRY(hipSetDevice(lib->device));
    hipStream_t stream = (hipStream_t)p->stream;
    // a batch whose constructor returned with its copies queued (tm_batch_create_from_structures): this stream waits
    // for them on the device, the host does not
    {
        std::lock_guard<std::mutex> lock(b->ready_mutex);
        if (b->ready) {
            if (hipEventQuery(b->ready) == hipSuccess) b->settle_locked();
            else HIP_TRY(hipStreamWaitEvent(stream, b->ready, 0));
        }
    }

    tm_result *r = new (std::nothrow) tm_result;
    if (!r) return fail(TM_E_NOMEM, "out of host memory");
    struct Guard { tm_result *r; ~Guard() { delete r; } } guard{r};
    r->device = lib->device; r->stream = stream;
    r->n_structs = b->n_structs; r->n_sel = p->n_selected;
    const uint32_t n_sel = p->n_selected;
    for (auto &e : r->ev) HIP_TRY(hipEventCreate(&e));

    // fp32 prefilter slack: a property of the batch's extent (2e-3 A unless a structure reaches beyond 4096 A of its
    // centroid) and, with the thresholds, of the staged windows and table classes
    const double slack = prefilter_slack((double)b->extent);
    // the staged selection: reused when this library was queried with identical arguments before (a few are kept)
    std::shared_ptr<StagedSelection> S;
    bool stage_now = false;
    {
        std::lock_guard<std::mutex> lock(lib->sel_mu);
        std::shared_ptr<StagedSelection> last;
        size_t found = lib->selections.size();
        for (size_t i = 0; i < lib->selections.size(); ++i) {
            const auto &c = lib->selections[i];
            if (c->sel.size() == n_sel && c->slack == slack && std::equal(c->sel.begin(), c->sel.end(), p->template_index) &&
                std::equal(c->rmsd.begin(), c->rmsd.end(), p->rmsd_threshold) &&
                std::equal(c->cut.begin(), c->cut.end(), p->distance_cutoff) &&
                std::equal(c->maxdyn.begin(), c->maxdyn.end(), p->max_dynamic_distance)) {
                found = i;
                last = c;
                break;
            }
        }
        const bool same = last != nullptr;
        const bool restage = same && last->calibration.load() == 2;  // measured: stage once more, in measured order
        if (same && !restage) {
            S = last;
            if (found != 0) std::rotate(lib->selections.begin(), lib->selections.begin() + found, lib->selections.begin() + found + 1);
        } else {
            S = std::make_shared<StagedSelection>();
            S->slack = slack;
            if (const char *env = std::getenv("ENZYMM_AMD_DIR_MILLI"))  // test hook: start from a page directory that is too small
                S->dir_milli = std::strtoull(env, nullptr, 10);
            S->sel.assign(p->template_index, p->template_index + n_sel);
            S->rmsd.assign(p->rmsd_threshold, p->rmsd_threshold + n_sel);
            S->cut.assign(p->distance_cutoff, p->distance_cutoff + n_sel);
            S->maxdyn.assign(p->max_dynamic_distance, p->max_dynamic_distance + n_sel);
            S->order.resize(n_sel);
            std::iota(S->order.begin(), S->order.end(), 0u);
            if (restage) {
                const std::vector<float> &cost = last->measured;
                std::stable_sort(S->order.begin(), S->order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
                S->calibration = 3;
                S->calibrated_on = last->calibrating_on.load();
                // what the selection has learned stays with it: the sizes its queries needed ...
                S->cands_per_katom = last->cands_per_katom.load();
                S->pairs_per_katom = last->pairs_per_katom.load();
                S->dir_milli = last->dir_milli.load();
            } else {
                std::stable_sort(S->order.begin(), S->order.end(), [&](uint32_t x, uint32_t y) {
                    return lib->cost_weight[S->sel[x]] > lib->cost_weight[S->sel[y]];
                });
                // ENZYMM_AMD_ORDER (experiments, round 6): 1 = templates that read the same pair table -- the same lead kinds
                // 0 and 1 -- next to each other (a block's lanes then ask for the same table entries, and a table read by six
                // templates comes from HBM once), in the a-priori order within a group and the groups by their heaviest member;
                // 2 = the same with lead kind 2 (the triple table's second list) as a further key
                static const int order_env = [] { const char *e = std::getenv("ENZYMM_AMD_ORDER"); return e ? std::atoi(e) : 0; }();
                if (order_env == 1 || order_env == 2) {
                    auto key_of = [&](uint32_t x) {
                        const uint32_t t = S->sel[x], off = lib->t_off[t];
                        const int G = lib->h_n_groups[t];
                        const uint64_t a = G >= 1 ? lib->h_lead_li[off] : 0xffffu, b = G >= 2 ? lib->h_lead_li[off + 1] : 0xffffu,
                                       c = G >= 3 && order_env == 2 ? lib->h_lead_li[off + 2] : 0u;
                        return (a << 32) | (b << 16) | c;
                    };
                    std::map<uint64_t, uint32_t> first_seen;  // group -> rank of its heaviest member in the a-priori order
                    for (uint32_t i = 0; i < n_sel; ++i) first_seen.emplace(key_of(S->order[i]), i);
                    std::stable_sort(S->order.begin(), S->order.end(), [&](uint32_t x, uint32_t y) {
                        return first_seen[key_of(x)] < first_seen[key_of(y)];
                    });
                }
            }
            {
                std::vector<uint32_t> sel(n_sel);
                std::vector<double> rm(n_sel), cu(n_sel), md(n_sel);
                for (uint32_t i = 0; i < n_sel; ++i) {
                    const uint32_t o = S->order[i];
                    sel[i] = S->sel[o]; rm[i] = S->rmsd[o]; cu[i] = S->cut[o]; md[i] = S->maxdyn[o];
                }
                HIP_TRY(S->d_sel.upload(sel)); HIP_TRY(S->d_rmsd.upload(rm));
                HIP_TRY(S->d_cut.upload(cu)); HIP_TRY(S->d_maxdyn.upload(md));
                HIP_TRY(S->d_order.upload(S->order));
            }
            {
                // pair tables: one per (lead kind A, lead kind B, distance class) some template reads.  A template
                // scans the (lead 0, lead 1) table and looks up rows of the (key, lead 2) table, the key being the
                // lead (0 or 1) nearer to lead 2 in the template -- the thinner shell, the shorter rows.
                std::map<std::tuple<uint16_t, uint16_t, int>, uint16_t> ids;
                std::vector<uint16_t> tp(n_sel, 0xffffu), tt(n_sel, 0xffffu);
                std::vector<uint8_t> key(n_sel, 0);
                auto table_for = [&](uint16_t a, uint16_t b, double hi) -> int {
                    int c = 0;
                    while (c + 1 < kTableClasses && hi > kTableClassLimit[c]) ++c;
                    auto it = ids.find(std::make_tuple(a, b, c));
                    if (it == ids.end()) {
                        if (S->tables.size() >= 0xffffu) return -1;
                        const double lim = kTableClassLimit[c];
                        const float d2 = c + 1 == kTableClasses ? 3.0e38f : (float)(lim * lim) * (1.0f + 4e-6f);
                        it = ids.emplace(std::make_tuple(a, b, c), (uint16_t)S->tables.size()).first;
                        S->tables.push_back(TableDesc{a, b, d2, (uint16_t)S->tables.size(), (uint16_t)0xffffu});
                    }
                    return it->second;
                };
                for (uint32_t i = 0; i < n_sel; ++i) {
                    const uint32_t o = S->order[i], t = S->sel[o], off = lib->t_off[t];
                    const int G = lib->h_n_groups[t];
                    auto upper = [&](int x, int y) {  // upper edge of the fp32 window (x, y), as k_stage_templates builds it
                        double d2 = 0;
                        for (int a = 0; a < 3; ++a) {
                            const double d = lib->h_xyz[3 * (size_t)(off + x) + a] - lib->h_xyz[3 * (size_t)(off + y) + a];
                            d2 += d * d;
                        }
                        const double thr = std::min(S->cut[o] + lib->h_w[off + x] + lib->h_w[off + y], S->maxdyn[o]);
                        return std::sqrt(d2) + thr + S->slack;
                    };
                    int id = 0;
                    if (G >= 2) {
                        id = table_for(lib->h_lead_li[off], lib->h_lead_li[off + 1], upper(1, 0));
                        if (id < 0) return fail(TM_E_UNSUPPORTED, "selection needs more than 65534 pair tables");
                        tp[i] = (uint16_t)id;
                    }
                    if (G >= 3) {
                        const double h20 = upper(2, 0), h21 = upper(2, 1);
                        key[i] = h21 < h20 ? 1 : 0;
                        id = table_for(lib->h_lead_li[off + key[i]], lib->h_lead_li[off + 2], std::min(h20, h21));
                        if (id < 0) return fail(TM_E_UNSUPPORTED, "selection needs more than 65534 pair tables");
                        tt[i] = (uint16_t)id;
                    }
                }
                if (std::getenv("ENZYMM_AMD_TRACE")) {  // census of the table plan
                    std::map<std::pair<uint16_t, uint16_t>, int> ordered, unordered;
                    int per_class[kTableClasses] = {0};
                    for (const auto &kv : ids) {
                        const uint16_t a = std::get<0>(kv.first), bb = std::get<1>(kv.first);
                        ordered[{a, bb}]++;
                        unordered[{std::min(a, bb), std::max(a, bb)}]++;
                        per_class[std::get<2>(kv.first)]++;
                    }
                    std::fprintf(stderr, "table plan: %zu tables over %u lead kinds; %zu ordered kind pairs, %zu unordered; per class",
                                 S->tables.size(), lib->n_lead, ordered.size(), unordered.size());
                    for (int c = 0; c < kTableClasses; ++c) std::fprintf(stderr, " %d", per_class[c]);
                    std::fprintf(stderr, "\n");
                }
                // groups of tables over the same two kind lists: the one with the largest limit leads, the others are
                // derived from it (ids is ordered by (a, b, class): a group's members are consecutive, the leader last)
                for (auto it = ids.begin(); it != ids.end();) {
                    auto last = it;
                    std::vector<uint16_t> members;
                    for (; it != ids.end() && std::get<0>(it->first) == std::get<0>(last->first) &&
                           std::get<1>(it->first) == std::get<1>(last->first); ++it)
                        members.push_back(it->second);
                    const uint16_t leader = members.back();
                    uint16_t chain = 0xffffu;
                    for (size_t m = 0; m + 1 < members.size(); ++m) {
                        S->tables[members[m]].src = leader;
                        S->tables[members[m]].next = chain;
                        chain = members[m];
                    }
                    S->tables[leader].src = leader;
                    S->tables[leader].next = chain;
                }
                if (std::getenv("ENZYMM_AMD_TAB_DERIVE") && std::atoi(std::getenv("ENZYMM_AMD_TAB_DERIVE")) == 0)  // experiments: every table swept
                    for (size_t t = 0; t < S->tables.size(); ++t) { S->tables[t].src = (uint16_t)t; S->tables[t].next = 0xffffu; }
                for (uint32_t i = 0; i < n_sel; ++i)
                    if (tt[i] == 0xffffu) S->small.push_back(i);
                {
                    std::vector<uint32_t> all(n_sel);
                    std::iota(all.begin(), all.end(), 0u);
                    HIP_TRY(S->d_small.upload(S->small)); HIP_TRY(S->d_all.upload(all));
                }
                HIP_TRY(S->d_tables.upload(S->tables)); HIP_TRY(S->d_pair_table.upload(tp));
                HIP_TRY(S->d_triple_table.upload(tt)); HIP_TRY(S->d_triple_key.upload(key));
                S->serial = ++g_serial;
            }
            HIP_TRY(S->d_bounds.alloc((size_t)n_sel * kMaxPairs));
            HIP_TRY(S->d_staged.alloc(n_sel));
            HIP_TRY(S->d_hdr.alloc(n_sel));
            HIP_TRY(S->d_look.alloc(n_sel));
            HIP_TRY(S->d_rescheck.alloc((size_t)n_sel * kResChecks));
            HIP_TRY(S->d_exact.alloc((size_t)n_sel * kMaxPairs));
            HIP_TRY(S->d_lead8.alloc((size_t)n_sel * kMaxG));
            HIP_TRY(hipEventCreateWithFlags(&S->ready, hipEventDisableTiming));
            if (n_sel) {
                hipLaunchKernelGGL(k_stage_templates, dim3(n_sel), dim3(64), 0, stream, lib->view(), n_sel, S->d_sel.p,
                                   S->d_cut.p, S->d_maxdyn.p, S->d_pair_table.p, S->d_triple_table.p, S->d_triple_key.p,
                                   S->d_staged.p, S->d_bounds.p, S->d_hdr.p, S->d_look.p, S->d_lead8.p, S->d_rescheck.p, S->d_exact.p, S->slack);
                HIP_TRY(hipGetLastError());
            }
            if (std::getenv("ENZYMM_AMD_TRACE") && n_sel) {  // census: how many templates share the records of the table stages
                HIP_TRY(hipStreamSynchronize(stream));
                std::vector<BlockHdr> h(n_sel);
                HIP_TRY(hipMemcpy(h.data(), S->d_hdr.p, n_sel * sizeof(BlockHdr), hipMemcpyDeviceToHost));
                std::map<std::string, int> whole, pair_part;
                for (const BlockHdr &b : h) {
                    whole[std::string(reinterpret_cast<const char *>(&b), sizeof b)]++;
                    std::string k(reinterpret_cast<const char *>(&b.w10), sizeof b.w10);
                    k.append(reinterpret_cast<const char *>(b.li), 2 * sizeof(uint16_t));
                    pair_part[k]++;
                }
                std::fprintf(stderr, "block headers: %u templates, %zu distinct (leads 0-2 and their three windows), %zu distinct pair stages (leads 0-1, window (1,0))\n",
                             n_sel, whole.size(), pair_part.size());
            }
            HIP_TRY(hipEventRecord(S->ready, stream));
            if (restage) {
                // ... and the logistic model tables of the hit epilogue (model_set follows the new order)
                std::lock_guard<std::mutex> mlock(last->models_mu);
                if (last->models_cached) {
                    std::vector<int32_t> ms(n_sel);
                    for (uint32_t i = 0; i < n_sel; ++i) ms[i] = last->h_model_set[S->order[i]];
                    HIP_TRY(S->d_model_set.upload(ms)); HIP_TRY(S->d_model_offsets.upload(last->h_model_offsets));
                    HIP_TRY(S->d_model_votes.upload(last->h_model_votes)); HIP_TRY(S->d_models.upload(last->h_models));
                    S->h_model_set = last->h_model_set; S->h_model_offsets = last->h_model_offsets;
                    S->h_model_votes = last->h_model_votes; S->h_models = last->h_models;
                    S->models_cached = true;
                }
                lib->selections[found] = S;
                if (found != 0) std::rotate(lib->selections.begin(), lib->selections.begin() + found, lib->selections.begin() + found + 1);
            } else {
                lib->selections.insert(lib->selections.begin(), S);
                if (lib->selections.size() > tm_library::kSelectionsKept) lib->selections.pop_back();
            }
            stage_now = true;
        }
    }
    if (!stage_now) HIP_TRY(hipStreamWaitEvent(stream, S->ready, 0));  // staged on another stream, possibly
    trace.mark("selection");
    r->selection = S;
    r->n_atoms = b->n_atoms;
    unsigned long long hcap = p->hit_capacity, ccap = p->candidate_capacity;
    const unsigned long long katoms = (b->n_atoms + 999ull) / 1000ull;
    if (!hcap) hcap = std::max<unsigned long long>({4096, 64ull * b->n_structs, S->pairs_per_katom.load() * katoms * 3 / 2});
    if (!ccap) ccap = std::min<unsigned long long>(0xfffffffeull,
        std::max<unsigned long long>({65536, 256ull * b->n_structs, S->cands_per_katom.load() * katoms * 3 / 2}));
    if (ccap > 0xfffffffeull) return fail(TM_E_UNSUPPORTED, "candidate_capacity must be below 2^32 - 1");
    r->hit_capacity = hcap; r->cand_capacity = ccap;
    HIP_TRY(r->d_hits.alloc(hcap));
    HIP_TRY(r->d_pairs.alloc(hcap));
    HIP_TRY(r->d_cands.alloc(ccap));
    HIP_TRY(r->d_cand_rmsd.alloc(ccap));
    const unsigned long long fit_cap = std::min<unsigned long long>(ccap, 1ull << 20);  // (96 B each: at most 96 MB)
    HIP_TRY(r->d_cand_fit.alloc((size_t)fit_cap * 12));
    if (b->has_wide) {
        // (zeroed per query: the launches that search the narrow structures write no upper bits)
        HIP_TRY(r->d_cands_hi.alloc((size_t)ccap * kMaxK));
        HIP_TRY(hipMemsetAsync(r->d_cands_hi.p, 0, (size_t)ccap * kMaxK, stream));
    }
    HIP_TRY(r->d_counters.alloc(4));
    HIP_TRY(r->d_queue.alloc(8));
    HIP_TRY(r->d_overflow.alloc(1));
    HIP_TRY(r->d_pool_used.alloc(1));
    HIP_TRY(r->d_tab_used.alloc(2));
    r->index_shared = b->index_shared;
    r->count_candidates = (p->flags & TM_FLAG_COUNT_CANDIDATES) != 0;
    // the shared state of the (structure, template) pairs, the pages of triples between k_search and k_deep and their
    // per-structure directory (every structure's share from its atom count and what the selection has needed so far)
    const size_t n_pairs_all = (size_t)b->n_structs * n_sel;
    HIP_TRY(r->d_pair_count.alloc(n_pairs_all));
    HIP_TRY(r->d_pair_last.alloc(n_pairs_all));
    if (p->flags & TM_FLAG_BEST_MATCH) { HIP_TRY(r->d_pair_min.alloc(n_pairs_all)); HIP_TRY(r->d_pair_tie.alloc(n_pairs_all)); }
    HIP_TRY(r->d_dir_n.alloc(b->n_structs));
    HIP_TRY(r->d_dir_head.alloc(b->n_structs));
    HIP_TRY(r->d_dir_triples.alloc(b->n_structs));
    {
        ZeroList z = {};
        z.p[0] = r->d_pool_used.p; z.words[0] = 2;
        z.p[1] = r->d_counters.p; z.words[1] = 8;
        z.p[2] = r->d_overflow.p; z.words[2] = 1;
        z.p[3] = r->d_tab_used.p; z.words[3] = 4;
        z.p[4] = r->d_dir_n.p; z.words[4] = b->n_structs;
        z.p[5] = r->d_dir_head.p; z.words[5] = b->n_structs;
        z.p[6] = r->d_dir_triples.p; z.words[6] = b->n_structs;
        z.p[7] = r->d_queue.p; z.words[7] = 8;
        hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, stream, z);
        HIP_TRY(hipGetLastError());
    }
    if (n_pairs_all) {
        HIP_TRY(hipMemsetAsync(r->d_pair_count.p, 0, n_pairs_all * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(r->d_pair_last.p, 0, n_pairs_all * sizeof(uint32_t), stream));
    }

    if (b->queried && b->last_stream != stream) HIP_TRY(hipStreamSynchronize(b->last_stream));
    b->last_stream = stream; b->queried = true;
    HIP_TRY(hipEventRecord(r->ev[0], stream));
    trace.mark("alloc");
    const bool rebuild = b->index_for != lib->serial || (p->flags & TM_FLAG_REBUILD_INDEX) || b->index_shared->stale.load();
    if (rebuild) {
        int rc = build_index(lib, b, r, stream);
        if (rc != TM_OK) return rc;
    }
    trace.mark("index");
    if (rebuild || b->tables_for != S->serial) {
        int rc = build_tables(lib, b, S.get(), stream);
        if (rc != TM_OK) return rc;
    }
    {
        int rc = carry_build_outcome(b, r, stream);
        if (rc != TM_OK) return rc;
    }
    if (b->has_wide && b->n_kind_longs && b->d_kind_unsearchable.n >= b->n_kind_longs) {
        // which long structures the index build found unsearchable with this library (a kind list beyond 16-bit
        // positions; bit 0x100 of the overflow word): the flags travel with the result, tm_result_unsearched lists them
        HIP_TRY(r->h_unsearchable.alloc(b->n_kind_longs));
        r->long_structures.resize(b->n_kind_longs);
        for (uint32_t i = 0; i < b->n_kind_longs; ++i) { r->h_unsearchable[i] = 0u; r->long_structures[i] = b->h_kind_longs.p[i].s; }
        HIP_TRY(hipMemcpyAsync(r->h_unsearchable.p, b->d_kind_unsearchable.p, b->n_kind_longs * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipEventRecord(r->ev[1], stream));
    trace.mark("tables");

    QueryView Q;
    Q.n_sel = n_sel; Q.sel = S->d_sel.p; Q.rmsd_thr = S->d_rmsd.p; Q.cutoff = S->d_cut.p; Q.maxdyn = S->d_maxdyn.p;
    Q.staged = S->d_staged.p; Q.hdr = S->d_hdr.p; Q.look = S->d_look.p; Q.rescheck = S->d_rescheck.p; Q.exact = S->d_exact.p; Q.bounds = S->d_bounds.p; Q.order = S->d_order.p; Q.max_candidates = p->max_candidates; Q.flags = p->flags;
#if defined(TM_STATS) || defined(TM_ABLATE)
    if (const char *env = std::getenv("ENZYMM_AMD_DEBUG_FLAGS")) Q.flags |= (uint32_t)std::strtoul(env, nullptr, 0);  // phase ablation bits for every query of a process
#endif
    if (const char *env = std::getenv("ENZYMM_AMD_RUN_LIST"))  // test hook: k_deep claims by page position, as for a large complex
        if (std::atoi(env) == 0) Q.flags |= 0x4000u;
    if (const char *env = std::getenv("ENZYMM_AMD_LOCAL_DEEP"))  // test hook / A-B: every surviving triple is paged out to k_deep (round 3's flow)
        if (std::atoi(env) == 0) Q.flags |= 0x8000u;
    Q.n_tables = (uint32_t)S->tables.size(); Q.tables = S->d_tables.p;
    Q.cost = nullptr;
    {
        int expected = 0;
        const uint32_t mean_atoms = b->n_structs ? (uint32_t)(b->n_atoms / b->n_structs) : 0u;
        if (S->calibration.load() == 3) {  // measured before: once more when these structures are of another size
            const uint32_t ref = S->calibrated_on.load();
            const uint32_t off = mean_atoms > ref ? mean_atoms - ref : ref - mean_atoms;
            if (ref && 4u * off > ref) expected = 3;
        }
        // (Round 2 searched a template per wave and ordered the selection by measured cost.  The block search spreads a
        // block's items over the lanes and k_deep shares the triples: the order matters little, and sorting by cost puts
        // all the heavy templates into the same blocks.  Measuring stays available for experiments.)
        static const bool calibrate = std::getenv("ENZYMM_AMD_CALIBRATE") != nullptr;
        if (calibrate && n_sel && b->n_structs >= 64 && S->calibration.compare_exchange_strong(expected, 1)) {  // enough searches to average
            S->calibrating_on = mean_atoms;
            HIP_TRY(S->d_cost.alloc(n_sel));
            HIP_TRY(hipMemsetAsync(S->d_cost.p, 0, n_sel * sizeof(unsigned long long), stream));
            Q.cost = S->d_cost.p;
            r->calibrates = true;
        }
    }
    Q.model_set = nullptr; Q.model_offsets = nullptr; Q.model_votes = nullptr; Q.models = nullptr;
    if (p->model_set && p->n_model_sets && p->model_offsets && p->model_votes && p->models) {
        const uint32_t ns = p->n_model_sets, nm = p->model_offsets[ns];
        for (uint32_t i = 0; i < n_sel; ++i)
            if (p->model_set[i] >= (int32_t)ns) return fail(TM_E_INVALID, "model_set index out of range");
        std::vector<int32_t> ms(n_sel), ms_caller(p->model_set, p->model_set + n_sel);
        for (uint32_t i = 0; i < n_sel; ++i) ms[i] = p->model_set[S->order[i]];
        std::vector<uint32_t> mo(p->model_offsets, p->model_offsets + ns + 1), mv(p->model_votes, p->model_votes + ns);
        std::vector<double> mm(p->models, p->models + 4 * (size_t)nm);
        bool shared = false;
        {
            std::lock_guard<std::mutex> lock(S->models_mu);
            if (!S->models_cached) {  // first query with models on this selection: upload once, keep
                HIP_TRY(S->d_model_set.upload(ms)); HIP_TRY(S->d_model_offsets.upload(mo));
                HIP_TRY(S->d_model_votes.upload(mv)); HIP_TRY(S->d_models.upload(mm));
                S->h_model_set = ms_caller; S->h_model_offsets = mo; S->h_model_votes = mv; S->h_models = mm;
                S->models_cached = true;
            }
            shared = S->h_model_set == ms_caller && S->h_model_offsets == mo && S->h_model_votes == mv && S->h_models == mm;
        }
        if (shared) {  // r->selection keeps S (and these buffers) alive for as long as the result exists
            Q.model_set = S->d_model_set.p; Q.model_offsets = S->d_model_offsets.p;
            Q.model_votes = S->d_model_votes.p; Q.models = S->d_models.p;
        } else {  // other tables than the cached ones: this result carries its own
            HIP_TRY(r->d_model_set.upload(ms)); HIP_TRY(r->d_model_offsets.upload(mo));
            HIP_TRY(r->d_model_votes.upload(mv)); HIP_TRY(r->d_models.upload(mm));
            Q.model_set = r->d_model_set.p; Q.model_offsets = r->d_model_offsets.p;
            Q.model_votes = r->d_model_votes.p; Q.models = r->d_models.p;
        }
    }
    if (const char *env = std::getenv("ENZYMM_AMD_HANDOVER"))  // test hooks: 0 = a wave keeps every subtree it starts, 2 = it hands over whatever it can
        Q.flags |= std::atoi(env) == 0 ? 0x10000u : std::atoi(env) == 2 ? 0x20000u : 0u;
    OutView O;
    O.cands = r->d_cands.p; O.cands_hi = b->has_wide ? r->d_cands_hi.p : nullptr; O.cand_rmsd = r->d_cand_rmsd.p; O.cand_fit = r->d_cand_fit.p; O.cand_fit_capacity = fit_cap; O.n_cands = r->d_counters.p + 1; O.cand_capacity = ccap;
    O.pairs = r->d_pairs.p; O.n_pairs = r->d_counters.p + 2; O.pair_capacity = hcap;
    O.hits = r->d_hits.p; O.n_hits = r->d_counters.p; O.hit_capacity = hcap; O.overflow = r->d_overflow.p;
    O.pair_count = r->d_pair_count.p; O.pair_last = r->d_pair_last.p;
    O.pair_min = r->d_pair_min.p; O.pair_tie = r->d_pair_tie.p;
    O.n_pages = r->d_counters.p + 3; O.dir_n = r->d_dir_n.p; O.dir_head = r->d_dir_head.p; O.dir_triples = r->d_dir_triples.p;
    O.pages = nullptr; O.page_capacity = 0; O.dir = nullptr; O.dir_off = nullptr;  // sized below, with the work items
    if (n_sel && b->n_structs) {
        // enough workgroups to fill 256 CUs several times over
        const uint32_t min_chunk = 2 * kSearchWaves;
        // template chunks: 2^cx of them spread over the XCDs (each XCD then touches 1/2^cx of the staged
        // templates), times `rounds` more when the batch alone cannot fill the GPU
        // fewer, longer workgroups: every workgroup ends with waves idling behind its last searches, so for a batch
        // that fills the GPU anyway two chunks per structure beat four (5.6 vs 6.0 ms) and eight lose 30 %
        uint32_t cx = 1;
        if (b->n_structs * 2u < 1024u) cx = 2;
        if (b->n_structs * 4u < 1024u) cx = 3;
        if (const char *env = std::getenv("ENZYMM_AMD_CX")) cx = (uint32_t)std::atoi(env) & 3u;  // experiments
        while (cx > 0 && (n_sel >> cx) < min_chunk) --cx;
        // work items: every structure gets rounds ~ its share of the estimated cost (pair tests grow with the
        // square of the atom count), at least one; heaviest structures first
        if (b->work_cx != cx || b->work_n_sel != n_sel) {
            const uint32_t ns = b->n_structs;
            // (ENZYMM_AMD_UNITS: work items per chunk class to aim for -- experiments with finer units at the end of a queued launch)
            static const uint32_t units_env = [] { const char *e = std::getenv("ENZYMM_AMD_UNITS"); return e ? (uint32_t)std::max(0, std::atoi(e)) : 0u; }();
            const uint32_t per_class_target = units_env ? std::max<uint32_t>(ns, units_env) : std::max<uint32_t>(ns, 2048u >> cx);
            const uint32_t max_rounds = std::min<uint32_t>(std::max<uint32_t>(1u, n_sel / (min_chunk << cx)), 0xffffu);
            std::vector<double> cost(ns);
            double total = 0.0;
            for (uint32_t s = 0; s < ns; ++s) {
                const double n = std::max<double>(64.0, (double)(b->s_off[s + 1] - b->s_off[s]));
                total += cost[s] = n * n;
            }
            std::vector<uint32_t> order(ns);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
            std::vector<WorkItem> items;
            items.reserve(ns);
            // (ENZYMM_AMD_TAIL = f, experiments: the structures that hold the last fraction f of the estimated cost -- the
            // units a queued launch ends on -- are cut into twice the rounds: a finer tail for a shorter drain)
            static const double tail_env = [] { const char *e = std::getenv("ENZYMM_AMD_TAIL"); return e ? std::atof(e) : 0.0; }();
            double seen = 0.0;
            for (uint32_t s : order) {
                const double share = cost[s] * (double)per_class_target / total;
                uint32_t rounds = (uint32_t)std::min<double>(max_rounds, std::max<double>(1.0, std::floor(share + 0.5)));
                if (tail_env > 0.0 && seen >= (1.0 - tail_env) * total) rounds = std::min<uint32_t>(max_rounds, 2u * rounds);
                seen += cost[s];
                for (uint32_t r = 0; r < rounds; ++r) items.push_back(WorkItem{s, (uint16_t)r, (uint16_t)rounds});
            }
            if (b->d_work.p) HIP_TRY(hipStreamSynchronize(stream));  // a previous query may still read the old list
            HIP_TRY(b->h_work.alloc(items.size()));
            std::copy(items.begin(), items.end(), b->h_work.p);
            HIP_TRY(b->d_work.upload_async(b->h_work.p, items.size(), stream));  // no host wait: see build_tables
            b->work_items = (uint32_t)items.size(); b->work_cx = cx; b->work_n_sel = n_sel;
        }
        // The page directory of the depth-first stage.  A structure lists one partly filled page per wave that worked on
        // it (its workgroups are known now) plus its full pages: from its atom count and what the selection has needed.
        {
            const uint32_t ns = b->n_structs;
            HIP_TRY(r->h_dir_off.alloc((size_t)ns + 1));
            r->atoms_of.resize(ns);
            r->dir_base.assign(ns, 16u);
            for (uint32_t i = 0; i < b->work_items; ++i) r->dir_base[b->h_work[i].s] += (uint32_t)kSearchWaves << cx;
            if (std::getenv("ENZYMM_AMD_DIR_MILLI")) r->dir_base.assign(ns, 0u);  // test hook: nothing but the learned part
            const double milli = (double)S->dir_milli.load();
            unsigned long long total = 0;
            for (uint32_t s = 0; s < ns; ++s) {
                const double n = (double)(b->s_off[s + 1] - b->s_off[s]);
                r->atoms_of[s] = b->s_off[s + 1] - b->s_off[s];
                r->h_dir_off[s] = (uint32_t)total;
                total += (unsigned long long)r->dir_base[s] + (unsigned long long)std::ceil(milli * n * n / 1e7);
                if (total > 0xfffffff0ull) return fail(TM_E_UNSUPPORTED, "page directory beyond 2^32 entries: split the batch");
            }
            r->h_dir_off[ns] = (uint32_t)total;
            r->page_capacity = total;
            HIP_TRY(r->d_dir.alloc(r->page_capacity));
            HIP_TRY(r->d_pages.alloc((size_t)r->page_capacity * kPageTriples));
            HIP_TRY(r->d_dir_off.upload_async(r->h_dir_off.p, (size_t)ns + 1, stream));
            O.pages = r->d_pages.p; O.page_capacity = r->page_capacity;
            O.dir = r->d_dir.p; O.dir_off = r->d_dir_off.p;
        }
        const uint32_t per_group = 8u >> cx;  // work items per group of 8 consecutive workgroups
        uint32_t grid = (b->work_items + per_group - 1) / per_group * 8u;
        const bool ic = (p->flags & TM_FLAG_IGNORE_CHAIN) != 0;
        // (the LDS layouts are sized for the largest NARROW structure: the wide ones are passed over by these launches)
        SearchArgs args{lib->view(), b->view(), Q, O, b->d_work.p, b->work_items, cx, b->max_atoms_narrow, b->max_res_narrow, 0u, nullptr, 0u, nullptr, 0u, 0u};
        // One launch of k_search in `mode`: the structure of a workgroup goes to LDS when the mode's per-wave scratch
        // leaves room for the largest structure of the batch; whatever LDS is left after that holds the structure's
        // kind lists (one workgroup per CU either way).
        auto launch = [&](int mode, bool force_global = false) -> int {
            const size_t scratch = search_scratch_bytes(mode);
            const size_t base = scratch + (size_t)b->max_atoms_narrow * sizeof(float4) + search_res_bytes(b->max_res_narrow) +
                                search_kl_words(lib->view().n_lead) * sizeof(uint32_t);
            static const size_t lds_budget = std::getenv("ENZYMM_AMD_LDS_BUDGET") ? (size_t)std::atol(std::getenv("ENZYMM_AMD_LDS_BUDGET")) : 160 * 1024;  // experiments
            const size_t lds_limit = lds_budget - 64 - search_static_bytes(mode);
            // (a structure in LDS has fewer than 8192 atoms -- 128 KB of records alone would not fit --, which the block search's
            // packed pair queue relies on: list positions in 13 bits; said here rather than assumed)
            const bool lds_path = base <= lds_limit && !force_global && b->max_atoms_narrow < 8192u;
            args.pool_cap = lds_path ? (uint32_t)((lds_limit - base) / sizeof(uint16_t)) & ~7u : 0u;  // (copied in 16-byte words)
            const size_t lds_bytes = lds_path ? lds_limit : scratch + (mode == kModeDeep ? search_cache_bytes() : 0);
            void (*kernel)(SearchArgs) = nullptr;
#define TM_PICK(m) (lds_path ? (ic ? k_search<true, true, m> : k_search<true, false, m>) : (ic ? k_search<false, true, m> : k_search<false, false, m>))
            kernel = mode == kModeBlocks ? TM_PICK(kModeBlocks) : mode == kModeSingly ? TM_PICK(kModeSingly) : TM_PICK(kModeDeep);
#undef TM_PICK
            HIP_TRY(allow_dynamic_lds(reinterpret_cast<const void *>(kernel), lds_bytes));
            // the block search over structures in LDS holds a whole CU per workgroup: one workgroup per CU, each taking
            // units from the launch's queues (k_search); ENZYMM_AMD_QUEUED=0: one workgroup per unit (tests, A/B)
            const int n_cus = device_cus(r->device);  // (per device: a process may drive GPUs of different sizes)
            const char *queued_env = std::getenv("ENZYMM_AMD_QUEUED");
            const bool queued = mode == kModeBlocks && lds_path;
            const bool once = n_cus <= 0 || (queued_env && std::atoi(queued_env) == 0);
            args.queue = queued ? r->d_queue.p : nullptr;
            args.queue_once = once ? 1u : 0u;
            const uint32_t launch_grid = queued && !once ? std::min<uint32_t>(grid, ((uint32_t)n_cus + 7u) & ~7u) : grid;
            hipLaunchKernelGGL(kernel, dim3(launch_grid), dim3(kSearchBlock), lds_bytes, stream, args);
            sync_each(stream, mode == kModeBlocks ? "k_search blocks" : mode == kModeSingly ? "k_search singly" : "k_deep");
            HIP_TRY(hipGetLastError());
            return TM_OK;
        };
        int rc = launch(kModeBlocks);
        if (rc != TM_OK) return rc;
        // templates the block search leaves out (k_tab_refs marks them kRefSingly), one template per wave at a time
        const bool huge = (uint64_t)b->max_atoms_narrow * b->max_atoms_narrow >= (uint64_t)kRefLenMask;
        if (huge || !S->small.empty()) {
            args.singly = huge ? S->d_all.p : S->d_small.p;
            args.n_singly = huge ? n_sel : (uint32_t)S->small.size();
            rc = launch(kModeSingly);
            if (rc != TM_OK) return rc;
        }
        if (b->has_wide) {
            // The wide structures (more than 65535 atoms: 24-bit atom indices): every template, one per wave at a time, the
            // structure's records in global memory -- a launch of its own over work items of its own (k_tab_refs marked
            // every template of such a structure for it; the launches above passed those structures over).
            std::vector<WorkItem> items;
            const uint32_t max_rounds = std::min<uint32_t>(std::max<uint32_t>(1u, n_sel / ((2u * kSearchWaves) << cx)), 0xffffu);
            uint32_t n_wide = 0;
            for (uint32_t s = 0; s < b->n_structs; ++s) n_wide += b->s_off[s + 1] - b->s_off[s] > TM_NARROW_STRUCTURE_ATOMS ? 1u : 0u;
            const uint32_t rounds = std::min<uint32_t>(max_rounds, std::max<uint32_t>(1u, (512u >> cx) / std::max(1u, n_wide)));
            for (uint32_t s = 0; s < b->n_structs; ++s)
                if (b->s_off[s + 1] - b->s_off[s] > TM_NARROW_STRUCTURE_ATOMS)
                    for (uint32_t k = 0; k < rounds; ++k) items.push_back(WorkItem{s, (uint16_t)k, (uint16_t)rounds});
            HIP_TRY(r->h_wide_work.alloc(items.size()));
            std::copy(items.begin(), items.end(), r->h_wide_work.p);
            HIP_TRY(r->d_wide_work.upload_async(r->h_wide_work.p, items.size(), stream));
            const WorkItem *narrow_work = args.work;
            const uint32_t narrow_items = args.n_items, narrow_grid = grid;
            args.work = r->d_wide_work.p; args.n_items = (uint32_t)items.size(); args.wide_only = 1u;
            args.singly = S->d_all.p; args.n_singly = n_sel;
            grid = ((uint32_t)items.size() + per_group - 1) / per_group * 8u;
            rc = launch(kModeSingly, true);
            args.work = narrow_work; args.n_items = narrow_items; args.wide_only = 0u; grid = narrow_grid;
            if (rc != TM_OK) return rc;
        }
        HIP_TRY(hipEventRecord(r->ev[4], stream));
        r->split_recorded = true;
        // the depth-first search below the triples the block search paged out: the same work items (every workgroup
        // of a structure takes triples of that structure until none are left)
        rc = launch(kModeDeep);
        if (rc != TM_OK) return rc;
    }
    HIP_TRY(hipEventRecord(r->ev[2], stream));
    trace.mark("search launch");
    if (n_sel && b->n_structs) {
        // candidate and pair counts live on the device: fixed grids with grid-stride loops, no host sync
        hipLaunchKernelGGL(k_superpose, dim3(512), dim3(256), 0, stream, lib->view(), b->view(), Q, O);
        sync_each(stream, "k_superpose");
        HIP_TRY(hipGetLastError());
        if (p->flags & TM_FLAG_BEST_MATCH) {
            hipLaunchKernelGGL(k_ties, dim3(512), dim3(256), 0, stream, Q, O);
            sync_each(stream, "k_ties");
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_select, dim3(256), dim3(256), 0, stream, lib->view(), b->view(), Q, O);
        sync_each(stream, "k_select");
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(r->ev[3], stream));
    {
        HIP_TRY(r->h_outcome.alloc(8));
        for (int i = 0; i < 8; ++i) r->h_outcome[i] = 0ull;
        HIP_TRY(hipMemcpyAsync(r->h_outcome.p, r->d_counters.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(r->h_outcome.p + 4, r->d_overflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(r->h_outcome.p + 5, r->d_pool_used.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(r->h_outcome.p + 6, r->d_tab_used.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(r->ev_done, stream));
        r->outcome_queued = true;
    }
    trace.mark("finish launch");
    guard.r = nullptr;
    *out = r;
    return TM_OK;
}

int tm_result_wait(tm_result *r) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    if (!r->waited) {
        HIP_TRY(hipSetDevice(r->device));
        unsigned long long counters[4] = {0, 0, 0, 0};
        unsigned long long pool_used = 0, tab_used[2] = {0, 0};
        if (r->outcome_queued) {  // (copied behind the query's last kernel: wait for that copy, not for the stream)
            HIP_TRY(hipEventSynchronize(r->ev_done));
            for (int i = 0; i < 4; ++i) counters[i] = r->h_outcome[i];
            r->overflow = (uint32_t)r->h_outcome[4];
            pool_used = r->h_outcome[5];
            tab_used[0] = r->h_outcome[6]; tab_used[1] = r->h_outcome[7];
        } else {
            HIP_TRY(hipMemcpyAsync(counters, r->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipMemcpyAsync(&r->overflow, r->d_overflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipMemcpyAsync(&pool_used, r->d_pool_used.p, sizeof pool_used, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipMemcpyAsync(tab_used, r->d_tab_used.p, sizeof tab_used, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
        }
        if ((r->overflow & 16u) && r->index_shared) {  // a pair-table pool ran out: remember what the build needed
            auto raise = [](std::atomic<unsigned long long> &a, unsigned long long v) {
                unsigned long long seen = a.load();
                while (seen < v && !a.compare_exchange_weak(seen, v)) {}
            };
            raise(r->index_shared->rows_wanted, tab_used[0]);
            raise(r->index_shared->ents_wanted, tab_used[1]);
            r->index_shared->stale = true;
        }
        r->n_hits = counters[0]; r->n_cands = counters[1]; r->n_pairs = counters[2];
        r->waited = true;
        if ((r->overflow & 32u) && r->selection && !r->atoms_of.empty()) {
            // the page directory (or the page pool behind it) was too small: what did the structures list?
            std::vector<uint32_t> listed(r->atoms_of.size());
            HIP_TRY(hipMemcpyAsync(listed.data(), r->d_dir_n.p, listed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
            double need = 0.0;
            for (size_t s = 0; s < listed.size(); ++s) {
                const double n = std::max(64.0, (double)r->atoms_of[s]);
                const double fixed = (double)r->dir_base[s];
                if ((double)listed[s] > fixed) need = std::max(need, ((double)listed[s] - fixed) * 1e7 / (n * n));
            }
            const unsigned long long want = (unsigned long long)(need * 1.25) + 1ull;
            unsigned long long seen = r->selection->dir_milli.load();
            while (seen < want && !r->selection->dir_milli.compare_exchange_weak(seen, want)) {}
        }
        if (r->selection && r->n_atoms) {  // what this selection needs per 1000 atoms (sizes the next results)
            auto raise = [](std::atomic<unsigned long long> &a, unsigned long long v) {
                unsigned long long seen = a.load();
                while (seen < v && !a.compare_exchange_weak(seen, v)) {}
            };
            const unsigned long long katoms = (r->n_atoms + 999ull) / 1000ull;
            raise(r->selection->cands_per_katom, (r->n_cands + katoms - 1) / katoms);
            raise(r->selection->pairs_per_katom, (std::max(r->n_pairs, r->n_hits) + katoms - 1) / katoms);
        }
        if (r->calibrates && r->selection) {  // the measured cost of every template, back in the caller's numbering
            StagedSelection &S = *r->selection;
            std::vector<unsigned long long> ticks(S.sel.size());
            if (!ticks.empty()) {
                HIP_TRY(hipMemcpyAsync(ticks.data(), S.d_cost.p, ticks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, r->stream));
                HIP_TRY(hipStreamSynchronize(r->stream));
            }
            S.measured.assign(S.sel.size(), 0.f);
            for (size_t i = 0; i < ticks.size(); ++i) S.measured[S.order[i]] = (float)ticks[i];
            S.calibration = 2;
        }
        if ((r->overflow & 8u) && r->index_shared) {  // the index build ran out of pool: remember what it needed
            unsigned long long seen = r->index_shared->pool_wanted.load();
            while (seen < pool_used && !r->index_shared->pool_wanted.compare_exchange_weak(seen, pool_used)) {}
            r->index_shared->stale = true;
        }
    }
    if ((r->overflow & 0x100u) && r->unsearched.empty()) {
        // a structure with more than 65535 atoms admissible to one template atom kind (beyond the 16-bit list positions of
        // the pair tables) was given empty lists: it has no hits, tm_result_unsearched names it, the others were searched
        for (size_t i = 0; i < r->long_structures.size(); ++i)
            if (r->h_unsearchable.p && r->h_unsearchable[i]) r->unsearched.push_back(r->long_structures[i]);
    }
    if (r->overflow & 16u)
        return fail(TM_E_OVERFLOW, "pair-table pools too small (" + std::to_string(r->row_capacity) + " row slots, " +
                                       std::to_string(r->ent_capacity) +
                                       " entries): query again, the batch now reserves what this build needed");
    if (r->overflow & 8u)
        return fail(TM_E_OVERFLOW, "candidate index pool too small (" + std::to_string(r->pool_capacity) +
                                       " entries): query again, the batch now reserves what this build needed");
    if (r->overflow & 32u)
        return fail(TM_E_OVERFLOW, "page directory of the depth-first stage too small (" + std::to_string(r->page_capacity) +
                                       " pages): query again, the selection now reserves what this query needed");
    if (r->overflow & ~0x100u)
        return fail(TM_E_OVERFLOW, "result tables too small: " + std::to_string(r->n_cands) + " candidates (capacity " +
                                       std::to_string(r->cand_capacity) + "), " + std::to_string(r->n_pairs) +
                                       " pairs (capacity " + std::to_string(r->hit_capacity) + ")");
    return TM_OK;
}

int tm_result_unsearched(tm_result *r, uint32_t *structures, uint32_t capacity, uint32_t *count) {
    if (!r || !count) return fail(TM_E_INVALID, "tm_result_unsearched: NULL argument");
    int rc = tm_result_wait(r);
    if (rc != TM_OK && rc != TM_E_OVERFLOW) return rc;
    *count = (uint32_t)r->unsearched.size();
    for (uint32_t i = 0; i < capacity && i < r->unsearched.size(); ++i)
        if (structures) structures[i] = r->unsearched[i];
    return TM_OK;
}

int tm_result_needed(tm_result *r, uint64_t *candidates, uint64_t *pairs) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    int rc = tm_result_wait(r);
    if (rc != TM_OK && rc != TM_E_OVERFLOW) return rc;
    if (candidates) *candidates = r->n_cands;
    if (pairs) *pairs = std::max(r->n_pairs, r->n_hits);  // the hit table shares hit_capacity
    return TM_OK;
}

int tm_result_hit_count(tm_result *r, uint64_t *n) {
    if (!r || !n) return fail(TM_E_INVALID, "NULL argument");
    int rc = tm_result_wait(r);
    *n = r->n_hits;
    return rc;
}

int tm_result_copy_hits(tm_result *r, tm_hit *out, uint64_t capacity) {
    int rc = tm_result_wait(r);
    if (rc != TM_OK) return rc;
    if (r->n_hits > capacity) return fail(TM_E_INVALID, "output buffer too small");
    if (r->n_hits == 0) return TM_OK;
    if (!out) return fail(TM_E_INVALID, "out is NULL");
    // on the query's own stream: a copy through the null stream would also wait for queries other streams run
    HIP_TRY(hipMemcpyAsync(out, r->d_hits.p, r->n_hits * sizeof(tm_hit), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    // (structure, template) order; several hits of one pair (best_match off) by ascending atom tuple,
    // which is the enumeration order of the reference semantics
    std::sort(out, out + r->n_hits, [](const tm_hit &a, const tm_hit &b) {
        if (a.structure != b.structure) return a.structure < b.structure;
        if (a.template_sel != b.template_sel) return a.template_sel < b.template_sel;
        return std::lexicographical_compare(a.atoms, a.atoms + a.n_atoms, b.atoms, b.atoms + b.n_atoms);
    });
    return TM_OK;
}

int tm_result_copy_candidates(tm_result *r, uint32_t *out, uint64_t n) {
    int rc = tm_result_wait(r);
    if (rc != TM_OK && rc != TM_E_OVERFLOW) return rc;
    if (!r->count_candidates) return fail(TM_E_INVALID, "query was not run with TM_FLAG_COUNT_CANDIDATES");
    const uint64_t need = (uint64_t)r->n_structs * r->n_sel;
    if (n < need || !out) return fail(TM_E_INVALID, "output buffer too small");
    if (need) HIP_TRY(hipMemcpy(out, r->d_pair_count.p, need * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TM_OK;
}

int tm_result_timing(tm_result *r, float *index_ms, float *search_ms, float *finish_ms) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev[3]));
    float a = 0, s = 0, f = 0;
    HIP_TRY(hipEventElapsedTime(&a, r->ev[0], r->ev[1]));
    HIP_TRY(hipEventElapsedTime(&s, r->ev[1], r->ev[2]));
    HIP_TRY(hipEventElapsedTime(&f, r->ev[2], r->ev[3]));
    if (index_ms) *index_ms = a;
    if (search_ms) *search_ms = s;
    if (finish_ms) *finish_ms = f;
    return TM_OK;
}

int tm_result_search_split(tm_result *r, float *front_ms, float *deep_ms) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev[3]));
    float a = 0, d = 0;
    if (r->split_recorded) {  // (a query with templates and structures; hipEventQuery also succeeds on an event never recorded)
        HIP_TRY(hipEventElapsedTime(&a, r->ev[1], r->ev[4]));
        HIP_TRY(hipEventElapsedTime(&d, r->ev[4], r->ev[2]));
    }
    if (front_ms) *front_ms = a;
    if (deep_ms) *deep_ms = d;
    return TM_OK;
}

int tm_result_page_census(tm_result *r, uint32_t *max_pages, uint64_t *triples) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    int rc = tm_result_wait(r);
    if (rc != TM_OK && rc != TM_E_OVERFLOW) return rc;
    HIP_TRY(hipSetDevice(r->device));
    std::vector<uint32_t> listed(r->n_structs, 0), held(r->n_structs, 0);
    if (r->n_structs && r->d_dir_n.p && r->d_dir_triples.p) {
        HIP_TRY(hipMemcpyAsync(listed.data(), r->d_dir_n.p, listed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
        HIP_TRY(hipMemcpyAsync(held.data(), r->d_dir_triples.p, held.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));
    }
    uint32_t most = 0;
    uint64_t total = 0;
    for (uint32_t v : listed) most = std::max(most, v);
    for (uint32_t v : held) total += v;
    if (max_pages) *max_pages = most;
    if (triples) *triples = total;
    return TM_OK;
}

int tm_result_wait_appended(tm_result *r) {
    if (!r) return fail(TM_E_INVALID, "result is NULL");
    if (!r->appended) return TM_OK;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev_appended));
    return TM_OK;
}

const void *tm_result_device_hits(tm_result *r) { return r ? r->d_hits.p : nullptr; }

int tm_result_append_hits(tm_result *r, void *rows, uint64_t capacity, uint64_t *count, uint32_t structure_offset) {
    if (!r || !count || (!rows && capacity)) return fail(TM_E_INVALID, "tm_result_append_hits: NULL argument");
    HIP_TRY(hipSetDevice(r->device));
    hipLaunchKernelGGL(k_append_hits, dim3(128), dim3(256), 0, r->stream, r->d_hits.p, r->d_counters.p, r->hit_capacity,
                       reinterpret_cast<tm_hit *>(rows), (unsigned long long)capacity,
                       reinterpret_cast<unsigned long long *>(count), structure_offset);
    HIP_TRY(hipGetLastError());
    if (!r->ev_appended) HIP_TRY(hipEventCreateWithFlags(&r->ev_appended, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->ev_appended, r->stream));
    r->appended = true;
    return TM_OK;
}

#ifdef TM_STATS
// diagnostic build only (make stats): read and reset the search counters
int tm_debug_stats(unsigned long long *out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(tmx::g_stats), 320 * sizeof(unsigned long long)));
    unsigned long long zero[320] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tmx::g_stats), zero, sizeof zero));
    return TM_OK;
}
#endif

#ifdef TM_TIMES
// diagnostic build only (-DTM_TIMES): read and reset the block search's per-stage wave clocks (g_times in tm_kernels.hip)
int tm_debug_times(unsigned long long *out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(tmx::g_times), 16 * sizeof(unsigned long long)));
    unsigned long long zero[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tmx::g_times), zero, sizeof zero));
    return TM_OK;
}
#endif

#ifdef TM_ABLATE
// timing build only (make ablate): read and reset the block search's log of units -- n x {start, end, hw_id | xcc_id << 32, first wave's end, sum of the waves' ends}
// (n with bit 31 set: the log of k_tab_build's workgroups instead)
int tm_debug_wg_log(unsigned long long *out, unsigned int n) {
    HIP_TRY(hipDeviceSynchronize());
    const bool tab = (n & 0x80000000u) != 0u, deep = (n & 0x40000000u) != 0u;  // (bit 30: k_deep's workgroups)
    n &= 0x3fffffffu;
    std::vector<tmx::WgLog> log(tmx::kWgLog);
    const void *sym = tab ? HIP_SYMBOL(tmx::g_tab_log) : deep ? HIP_SYMBOL(tmx::g_deep_log) : HIP_SYMBOL(tmx::g_wg_log);
    HIP_TRY(hipMemcpyFromSymbol(log.data(), sym, log.size() * sizeof(tmx::WgLog)));
    for (unsigned int i = 0; i < n && i < tmx::kWgLog; ++i) {
        out[5 * i] = log[i].start; out[5 * i + 1] = log[i].end;
        out[5 * i + 2] = (unsigned long long)log[i].hw_id | ((unsigned long long)log[i].xcc_id << 32);
        out[5 * i + 3] = log[i].first_end; out[5 * i + 4] = log[i].sum_end;
    }
    std::fill(log.begin(), log.end(), tmx::WgLog{0, 0, ~0ull, 0, 0, 0});
    HIP_TRY(hipMemcpyToSymbol(sym, log.data(), log.size() * sizeof(tmx::WgLog)));
    return TM_OK;
}
#endif

void tm_result_destroy(tm_result *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (!r->waited) (void)hipStreamSynchronize(r->stream);
    delete r;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- host-side PDB reader

namespace {

struct PdbLine {
    const char *p;   // start of the line
    uint32_t len;    // without the line terminator(s)
};

// Calls f(line) for every line; stops when f returns false.
template <class F>
void for_each_line(const char *text, uint64_t length, F &&f) {
    // (round 6: the newlines of sixteen bytes at a time -- a memchr call per 81-byte record was a third of the host reader's
    // counting pass)
    uint64_t start = 0;  // first byte of the line being looked for
    auto emit = [&](uint64_t end) {
        uint64_t len = end - start;
        if (len > 0 && text[end - 1] == '\r') --len;
        const bool more = f(PdbLine{text + start, (uint32_t)std::min<uint64_t>(len, 0xffffffffu)});
        start = end + 1;
        return more;
    };
    uint64_t i = 0;
    const __m128i nl = _mm_set1_epi8('\n');
    for (; i + 16 <= length; i += 16) {
        unsigned mask = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(text + i)), nl));
        while (mask) {
            const uint64_t end = i + (uint64_t)__builtin_ctz(mask);
            mask &= mask - 1u;
            if (!emit(end)) return;
        }
    }
    for (; i < length; ++i)
        if (text[i] == '\n' && !emit(i)) return;
    if (start < length) (void)emit(length);  // the last line has no newline
}

inline bool tag_is(const PdbLine &l, const char *tag6) {
    if (l.len >= 6) return std::memcmp(l.p, tag6, 6) == 0;  // (two integer compares; the rule below for shorter lines)
    for (int k = 0; k < 6; ++k) {
        const char c = (uint32_t)k < l.len ? l.p[k] : ' ';
        if (c != tag6[k]) return false;
    }
    return true;
}
// A line's record name as one integer -- columns 1-6, blanks past the end of the line --: the readers compare it with four
// constants instead of walking up to four tags character by character per line.
inline uint64_t tag_of(const PdbLine &l) {
    uint64_t t = 0x202020202020ull;  // six blanks
    if (l.len >= 6) std::memcpy(&t, l.p, 6);
    else std::memcpy(&t, l.p, l.len);
    return t;
}
constexpr uint64_t tag_const(const char (&s)[7]) {
    return (uint64_t)(unsigned char)s[0] | (uint64_t)(unsigned char)s[1] << 8 | (uint64_t)(unsigned char)s[2] << 16 |
           (uint64_t)(unsigned char)s[3] << 24 | (uint64_t)(unsigned char)s[4] << 32 | (uint64_t)(unsigned char)s[5] << 40;
}
constexpr uint64_t kTagAtom = tag_const("ATOM  "), kTagHetatm = tag_const("HETATM"), kTagEndmdl = tag_const("ENDMDL"),
                   kTagHeader = tag_const("HEADER");

// One decimal field of `width` <= 8 columns starting at column `lo` (columns past the end of the line are
// blanks): [spaces][sign]digits[.digits][spaces].  value = mantissa / 10^frac, both exact, so the quotient
// is the correctly rounded double.
inline bool parse_field(const PdbLine &l, int lo, int width, bool blank_ok, bool integer, long long *mant_out,
                        int *frac_out) {
    long long mant = 0;
    int frac = 0, ndig = 0;
    bool seen_dot = false, started = false, ended = false, neg = false, bad = false;
    for (int k = 0; k < width; ++k) {
        const unsigned char c = (uint32_t)(lo + k) < l.len ? (unsigned char)l.p[lo + k] : ' ';
        const bool is_d = c >= '0' && c <= '9', is_dot = c == '.', is_sign = c == '-' || c == '+', is_sp = c == ' ';
        bad |= !(is_d || is_dot || is_sign || is_sp);
        bad |= ended && !is_sp;
        bad |= is_sign && started;
        bad |= is_dot && seen_dot;
        ended |= started && is_sp;
        if (is_d) {
            mant = (long long)((unsigned long long)mant * 10ull + (unsigned long long)(c - '0'));
            frac += seen_dot ? 1 : 0;
            ++ndig;
        }
        neg |= c == '-';
        seen_dot |= is_dot;
        started |= is_d || is_dot || is_sign;
    }
    *mant_out = neg ? -mant : mant;
    *frac_out = frac;
    return !bad && ndig <= 18 && (ndig > 0 || (!started && blank_ok)) && !(integer && seen_dot);
}

// The same field when it has the plain shape [spaces][sign]digits[.digits][spaces] and lies inside the line -- every field a
// PDB writer produces --, in one pass of dependent-free loops (round 6: the general parser's eight flags per character were
// 0.3 us per record of the host reader).  Anything else -- blank, a stray character, a second dot, a sign in the middle, an
// integer field with a dot -- returns false and is left to parse_field, which alone decides what is an error.
inline bool parse_field_plain(const char *p, int width, bool integer, long long *mant_out, int *frac_out) {
    int k = 0;
    while (k < width && p[k] == ' ') ++k;
    if (k == width) return false;
    bool neg = false;
    if (p[k] == '-') { neg = true; ++k; }
    else if (p[k] == '+') ++k;
    unsigned long long m = 0;
    int nd = 0, fr = 0;
    while (k < width && (unsigned)(p[k] - '0') < 10u) { m = m * 10ull + (unsigned long long)(p[k] - '0'); ++k; ++nd; }
    if (k < width && p[k] == '.') {
        if (integer) return false;
        ++k;
        while (k < width && (unsigned)(p[k] - '0') < 10u) { m = m * 10ull + (unsigned long long)(p[k] - '0'); ++k; ++nd; ++fr; }
    }
    while (k < width && p[k] == ' ') ++k;
    if (k != width || nd == 0 || nd > 18) return false;
    *mant_out = neg ? -(long long)m : (long long)m;
    *frac_out = fr;
    return true;
}

const double kPow10[19] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                           1e17, 1e18};

inline char col(const PdbLine &l, int k) { return (uint32_t)k < l.len ? l.p[k] : ' '; }

inline bool is_strip_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// columns [lo, hi) stripped of surrounding whitespace, left-aligned, NUL padded (numpy.char.strip into an S field)
inline void stripped(const PdbLine &l, int lo, int hi, char *out) {
    int a = lo, b = hi;
    while (a < b && is_strip_space(col(l, a))) ++a;
    while (b > a && is_strip_space(col(l, b - 1))) --b;
    int n = 0;
    for (int k = a; k < b; ++k) out[n++] = col(l, k);
    for (; n < hi - lo; ++n) out[n] = '\0';
}

}  // namespace

extern "C" {

int tm_pdb_count_host(const char *text, uint64_t length, uint32_t flags, uint64_t *n_atoms, char header_id[4],
                      int *has_header) {
    if ((!text && length) || !n_atoms) return fail(TM_E_INVALID, "tm_pdb_count_host: NULL argument");
    uint64_t n = 0;
    bool open = true, header = false;
    for_each_line(text, length, [&](const PdbLine &l) {
        const uint64_t tag = tag_of(l);
        if (open && (tag == kTagAtom || tag == kTagHetatm)) ++n;
        else if (open && !(flags & TM_PDB_IGNORE_ENDMDL) && tag == kTagEndmdl) open = false;
        else if (!header && tag == kTagHeader) {
            header = true;
            if (has_header) *has_header = l.len >= 66 ? 1 : 0;
            if (header_id && l.len >= 66) std::memcpy(header_id, l.p + 62, 4);
        }
        return open || !header;  // nothing left to learn once the model is closed and the header seen
    });
    if (!header && has_header) *has_header = 0;
    *n_atoms = n;
    return TM_OK;
}

int tm_pdb_parse_host(const char *text, uint64_t length, uint32_t flags, uint64_t capacity, const tm_pdb_columns *out,
                      uint64_t *bad_begin, uint64_t *bad_length) {
    if ((!text && length) || !out) return fail(TM_E_INVALID, "tm_pdb_parse_host: NULL argument");
    // error precedence of the vectorised reference implementation: the first record shorter than 54 columns;
    // otherwise, field by field (serial, resSeq, x, y, z, occupancy, tempFactor), the first record whose field
    // is malformed
    const PdbLine none{nullptr, 0};
    PdbLine short_line = none, bad_field[7] = {none, none, none, none, none, none, none};
    static const int lo[7] = {6, 22, 30, 38, 46, 54, 60}, width[7] = {5, 4, 8, 8, 8, 6, 6};
    uint64_t n = 0;
    bool overflow = false;
    for_each_line(text, length, [&](const PdbLine &l) {
        const uint64_t tag = tag_of(l);
        if (!(flags & TM_PDB_IGNORE_ENDMDL) && tag == kTagEndmdl) return false;
        if (!(tag == kTagAtom || tag == kTagHetatm)) return true;
        if (n >= capacity) { overflow = true; return false; }
        if (l.len < 54) {
            if (!short_line.p) short_line = l;
            ++n;
            return true;
        }
        long long mant[7];
        int frac[7];
        for (int f = 0; f < 7; ++f) {
            if ((uint32_t)(lo[f] + width[f]) <= l.len && parse_field_plain(l.p + lo[f], width[f], f < 2, &mant[f], &frac[f])) continue;
            if (!parse_field(l, lo[f], width[f], f >= 5, f < 2, &mant[f], &frac[f]) && !bad_field[f].p) bad_field[f] = l;
        }
        out->serial[n] = (int32_t)mant[0];
        out->residue_number[n] = (int32_t)mant[1];
        for (int a = 0; a < 3; ++a) out->xyz[3 * n + a] = (double)mant[2 + a] / kPow10[std::min(frac[2 + a], 18)];
        out->occupancy[n] = (double)mant[5] / kPow10[std::min(frac[5], 18)];
        out->temperature_factor[n] = (double)mant[6] / kPow10[std::min(frac[6], 18)];
        for (int k = 0; k < 4; ++k) out->name[4 * n + k] = col(l, 12 + k);
        out->altloc[n] = col(l, 16);
        for (int k = 0; k < 3; ++k) out->residue_name[3 * n + k] = col(l, 17 + k);
        for (int k = 0; k < 2; ++k) out->chain_id[2 * n + k] = col(l, 20 + k);
        out->insertion_code[n] = col(l, 26);
        stripped(l, 72, 76, out->segment + 4 * n);
        stripped(l, 76, 78, out->element + 2 * n);
        const char digit = col(l, 78), sign = col(l, 79);
        out->charge[n] = (digit >= '0' && digit <= '9') ? (int8_t)((digit - '0') * (sign == '-' ? -1 : 1)) : (int8_t)0;
        ++n;
        return true;
    });
    if (overflow) return fail(TM_E_INVALID, "tm_pdb_parse_host: more records than capacity");
    PdbLine bad = short_line;
    for (int f = 0; f < 7 && !bad.p; ++f) bad = bad_field[f];
    if (bad.p) {
        if (bad_begin) *bad_begin = (uint64_t)(bad.p - text);
        if (bad_length) *bad_length = bad.len;
        return fail(TM_E_INVALID, "malformed PDB record");
    }
    return TM_OK;
}

}  // extern "C"

namespace {

// Runs f(i) for i in [0, n) on a team of host threads (contiguous slices, at least `grain` items each).
template <class F>
void parallel_for(uint32_t n, uint32_t grain, F &&f) {
    const uint32_t team = std::min<uint32_t>(host_team(), std::max(1u, n / std::max(1u, grain)));
    if (team <= 1) {
        for (uint32_t i = 0; i < n; ++i) f(i);
        return;
    }
    HostPool::run_anywhere(team, [&](uint32_t t) {
        const uint32_t lo = (uint32_t)((uint64_t)n * t / team), hi = (uint32_t)((uint64_t)n * (t + 1) / team);
        for (uint32_t i = lo; i < hi; ++i) f(i);
    });
}

tm_pdb_columns columns_at(const tm_pdb_columns &c, uint64_t o) {
    tm_pdb_columns r;
    r.serial = c.serial + o; r.name = c.name + 4 * o; r.altloc = c.altloc + o; r.residue_name = c.residue_name + 3 * o;
    r.chain_id = c.chain_id + 2 * o; r.residue_number = c.residue_number + o; r.insertion_code = c.insertion_code + o;
    r.xyz = c.xyz + 3 * o; r.occupancy = c.occupancy + o; r.temperature_factor = c.temperature_factor + o;
    r.segment = c.segment + 4 * o; r.element = c.element + 2 * o; r.charge = c.charge + o;
    return r;
}

}  // namespace

extern "C" {

int tm_files_sizes(uint32_t n_files, const char *const *paths, uint64_t *sizes, int64_t *bad_file) {
    if (n_files && (!paths || !sizes)) return fail(TM_E_INVALID, "tm_files_sizes: NULL argument");
    std::vector<uint8_t> failed(n_files, 0);
    parallel_for(n_files, 32, [&](uint32_t f) {
        struct stat st;
        if (::stat(paths[f], &st) != 0 || !S_ISREG(st.st_mode)) { failed[f] = 1; sizes[f] = 0; }
        else sizes[f] = (uint64_t)st.st_size;
    });
    if (bad_file) *bad_file = -1;
    for (uint32_t f = 0; f < n_files; ++f)
        if (failed[f]) {
            if (bad_file) *bad_file = f;
            return fail(TM_E_INVALID, std::string("cannot stat ") + paths[f]);
        }
    return TM_OK;
}

int tm_files_read(uint32_t n_files, const char *const *paths, const uint64_t *offsets, char *buffer, int64_t *bad_file) {
    if (n_files && (!paths || !offsets || !buffer)) return fail(TM_E_INVALID, "tm_files_read: NULL argument");
    std::vector<uint8_t> failed(n_files, 0);
    parallel_for(n_files, 16, [&](uint32_t f) {
        const uint64_t want = offsets[f + 1] - offsets[f];
        const int fd = ::open(paths[f], O_RDONLY | O_CLOEXEC);
        if (fd < 0) { failed[f] = 1; return; }
        uint64_t got = 0;
        while (got < want) {
            const ssize_t r = ::read(fd, buffer + offsets[f] + got, want - got);
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        ::close(fd);
        if (got != want) failed[f] = 1;  // the file changed size since tm_files_sizes
    });
    if (bad_file) *bad_file = -1;
    for (uint32_t f = 0; f < n_files; ++f)
        if (failed[f]) {
            if (bad_file) *bad_file = f;
            return fail(TM_E_INVALID, std::string("cannot read ") + paths[f]);
        }
    return TM_OK;
}

int tm_pdb_count_host_many(uint32_t n_files, const char *const *text, const uint64_t *length, uint32_t flags,
                           uint64_t *n_atoms, char *header_ids, int *has_header) {
    if (n_files && (!text || !length || !n_atoms || !header_ids || !has_header))
        return fail(TM_E_INVALID, "tm_pdb_count_host_many: NULL argument");
    std::atomic<int> rc{TM_OK};
    parallel_for(n_files, 8, [&](uint32_t f) {
        const int r = tm_pdb_count_host(text[f], length[f], flags, &n_atoms[f], header_ids + 4 * (size_t)f, &has_header[f]);
        if (r != TM_OK) rc = r;
    });
    return rc == TM_OK ? TM_OK : fail(rc, "tm_pdb_count_host_many: invalid file");
}

int tm_pdb_parse_host_many(uint32_t n_files, const char *const *text, const uint64_t *length, uint32_t flags,
                           const uint64_t *atom_offsets, const tm_pdb_columns *out, int64_t *bad_file,
                           uint64_t *bad_begin, uint64_t *bad_length) {
    if (n_files && (!text || !length || !atom_offsets || !out))
        return fail(TM_E_INVALID, "tm_pdb_parse_host_many: NULL argument");
    std::vector<uint64_t> begin(n_files, 0), len(n_files, 0);
    std::vector<int> status(n_files, TM_OK);
    parallel_for(n_files, 8, [&](uint32_t f) {
        const tm_pdb_columns cols = columns_at(*out, atom_offsets[f]);
        status[f] = tm_pdb_parse_host(text[f], length[f], flags, atom_offsets[f + 1] - atom_offsets[f], &cols, &begin[f],
                                      &len[f]);
    });
    for (uint32_t f = 0; f < n_files; ++f)
        if (status[f] != TM_OK) {
            if (bad_file) *bad_file = f;
            if (bad_begin) *bad_begin = begin[f];
            if (bad_length) *bad_length = len[f];
            return fail(status[f], "malformed PDB record");
        }
    if (bad_file) *bad_file = -1;
    return TM_OK;
}

}  // extern "C"

extern "C" {

// dst <- src with non-temporal stores where dst is 16-byte aligned (the bytes before and after with plain stores)
static void stream_copy(char *dst, const char *src, size_t n) {
    size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    if (head > n) head = n;
    std::memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t blocks = n / 64;
    for (size_t i = 0; i < blocks; ++i) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 16)),
                      c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 32)), d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst), a); _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 32), c); _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 48), d);
        src += 64; dst += 64;
    }
    std::memcpy(dst, src, n - blocks * 64);
    _mm_sfence();
}

int tm_batch_create_from_files(int device, uint32_t n_files, const char *const *paths, const tm_vocab *vocab,
                               uint32_t flags, double min_bfactor, tm_batch **out, int64_t *bad_file) {
    if (!out || !vocab || (n_files && !paths)) return fail(TM_E_INVALID, "tm_batch_create_from_files: NULL argument");
    *out = nullptr;
    if (bad_file) *bad_file = -1;
    HIP_TRY(hipSetDevice(device));
    PhaseTrace trace("tm_batch_create_from_files");
    // sizes, then every file into its slot of ONE page-locked buffer, a newline after each (an empty extra line
    // when the file already ends in one): the buffer is the device layout and goes up in a single DMA
    std::vector<uint64_t> sizes(n_files, 0), off(n_files + 1, 0);
    int rc = tm_files_sizes(n_files, paths, sizes.data(), bad_file);
    if (rc != TM_OK) return rc;
    for (uint32_t f = 0; f < n_files; ++f) off[f + 1] = off[f] + sizes[f] + 1;
    trace.mark("stat");
    PinnedBuf<char> text;
    HIP_TRY(text.alloc(off[n_files] + 16));
    trace.mark("pinned");
    // The files are read in a few chunks, and each chunk's bytes go up while the next one is read (round 5: reading
    // 190 MB took 2.6 ms and uploading them 4 ms, one after the other; the device parse starts behind the last copy)
    DevBuf<char> d_text;
    hipStream_t stream = nullptr;
    HIP_TRY(ingest_stream(device, &stream));
    const uint64_t n_bytes = (off[n_files] + 15) & ~15ull;
    HIP_TRY(d_text.alloc(n_bytes + 16));
    // Declared after `text` and `d_text`, so destroyed BEFORE them: whatever way this function is left once a chunk's copy
    // is queued, the ingest stream is waited for before the page-locked and the device block go back to their pools (a
    // failing HIP call further down returned with DMA still reading `text` -- ADVICE r05); disarmed when the batch takes
    // the buffers over.
    StreamWaitGuard copy_guard{stream, false};
    {
        std::vector<uint8_t> failed(n_files, 0);
        static const uint32_t n_chunks = [] {
            if (const char *e = std::getenv("ENZYMM_AMD_READ_CHUNKS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1 && v <= 64) return (uint32_t)v; }
            return 6u;
        }();
        // ENZYMM_AMD_READ_TURNS=1 (experiment): one call reads its files at a time, process-wide -- the ingest threads of a
        // pipeline then stagger (one reads while another's upload finishes and a third's text is parsed) instead of all
        // reading, then all uploading, in step
        static const bool read_turns = [] { const char *e = std::getenv("ENZYMM_AMD_READ_TURNS"); return e && e[0] == '1'; }();
        // files are read through a buffer that stays in the reading core's cache and streamed on from there (see below);
        // ENZYMM_AMD_READ_BOUNCE: its size in KB (default 96; 0 = read() straight into the page-locked buffer, as until round 5)
        static const size_t read_bounce = [] {
            if (const char *e = std::getenv("ENZYMM_AMD_READ_BOUNCE")) { const long v = std::strtol(e, nullptr, 10); if (v >= 0 && v <= 4096) return (size_t)v * 1024; }
            return (size_t)96 * 1024;
        }();
        static std::mutex read_turn;
        std::unique_lock<std::mutex> turn(read_turn, std::defer_lock);
        if (read_turns) turn.lock();
        uint32_t f0 = 0;
        for (uint32_t c = 0; c < n_chunks && f0 < n_files; ++c) {
            const uint64_t target = off[n_files] * (c + 1) / n_chunks;
            uint32_t f1 = f0;
            while (f1 < n_files && (c + 1 == n_chunks || off[f1 + 1] <= target || f1 == f0)) ++f1;
            parallel_for(f1 - f0, 4, [&](uint32_t i) {
                const uint32_t f = f0 + i;
                const uint64_t want = sizes[f];
                const int fd = ::open(paths[f], O_RDONLY | O_CLOEXEC);
                if (fd < 0) { failed[f] = 1; return; }
                uint64_t got = 0;
                if (read_bounce) {
                    // through a buffer that stays in the core's cache, on to the page-locked buffer with streaming stores: the
                    // destination's lines are not read before they are overwritten -- a quarter of the DRAM traffic of a batch
                    // (run_files 157-165 k -> 172-189 k structures/s, A/B on one box: tests/gpu_ab_bounce.sh)
                    static thread_local std::vector<char> bounce;
                    if (bounce.size() < read_bounce + 64) bounce.resize(read_bounce + 64);
                    char *bb = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(bounce.data()) + 63) & ~(uintptr_t)63);
                    while (got < want) {
                        const ssize_t r = ::read(fd, bb, std::min<uint64_t>(want - got, read_bounce));
                        if (r <= 0) break;
                        stream_copy(text.p + off[f] + got, bb, (size_t)r);
                        got += (uint64_t)r;
                    }
                } else
                while (got < want) {
                    const ssize_t r = ::read(fd, text.p + off[f] + got, want - got);
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                ::close(fd);
                if (got != want) failed[f] = 1;
                text.p[off[f] + want] = '\n';
            });
            bool any = false;
            for (uint32_t f = f0; f < f1; ++f) any = any || failed[f];
            if (!any && off[f1] > off[f0]) {
                copy_guard.armed = true;
                HIP_TRY(hipMemcpyAsync(d_text.p + off[f0], text.p + off[f0], off[f1] - off[f0], hipMemcpyHostToDevice, stream));
            }
            f0 = f1;
        }
        for (uint32_t f = 0; f < n_files; ++f)
            if (failed[f]) {
                (void)hipStreamSynchronize(stream);  // (copies out of `text` may be in flight)
                if (bad_file) *bad_file = f;
                return fail(TM_E_INVALID, std::string("cannot read ") + paths[f]);
            }
    }
    trace.mark("read+upload");
    std::vector<const char *> ptr(n_files);
    std::vector<uint64_t> len(n_files);
    for (uint32_t f = 0; f < n_files; ++f) { ptr[f] = text.p + off[f]; len[f] = sizes[f] + 1; }
    rc = batch_from_pdb(device, n_files, ptr.data(), len.data(), vocab, flags, min_bfactor, out, &d_text);
    trace.mark("from_pdb");
    if (rc != TM_OK) return rc;  // (copy_guard waits)
    tm_batch *b = *out;
    copy_guard.armed = false;
    std::swap(b->h_text.p, text.p); std::swap(b->h_text.n, text.n); std::swap(b->h_text.bytes, text.bytes);
    b->h_text_off = std::move(off);
    return TM_OK;
}

// The records (text lines) of the atoms a hit table matched, for a batch made by tm_batch_create_from_files: per
// structure the distinct matched atoms in ascending order, each with the first `width` bytes of its record's line (read
// from the batch's text buffer, clamped at its end).  keys[i] = structure << 32 | atom (file-order index), lines[i * width ..].
// Host-side only (text and offsets are kept in page-locked memory by the batch); on a team of host threads.
int tm_batch_pick_records(tm_batch *b, const tm_hit *hits, uint64_t n_hits, uint32_t width, uint64_t capacity, uint64_t *keys,
                          uint8_t *lines, uint64_t *count) {
    if (!b || !count || (n_hits && !hits) || (capacity && (!keys || !lines)) || width == 0)
        return fail(TM_E_INVALID, "tm_batch_pick_records: bad argument");
    if (b->h_text_off.size() != (size_t)b->n_structs + 1 || b->h_line_off.n != b->n_atoms)
        return fail(TM_E_INVALID, "batch was not created from files");
    *count = 0;
    // runs of equal structure (the table is sorted by structure; an unsorted table just gives several runs per structure)
    std::vector<std::pair<uint64_t, uint64_t>> runs;
    for (uint64_t i = 0; i < n_hits;) {
        uint64_t j = i + 1;
        while (j < n_hits && hits[j].structure == hits[i].structure) ++j;
        if (hits[i].structure >= b->n_structs) return fail(TM_E_INVALID, "tm_batch_pick_records: structure index out of range");
        runs.emplace_back(i, j);
        i = j;
    }
    std::vector<std::vector<uint32_t>> atoms(runs.size());
    std::vector<uint64_t> first(runs.size() + 1, 0);
    std::atomic<int> bad{0};
    auto collect = [&](uint32_t r) {
        std::vector<uint32_t> &a = atoms[r];
        const uint32_t s = hits[runs[r].first].structure, n = b->s_off[s + 1] - b->s_off[s];
        for (uint64_t i = runs[r].first; i < runs[r].second; ++i)
            for (uint32_t k = 0; k < hits[i].n_atoms && k < TM_MAX_TEMPLATE_ATOMS; ++k) {
                if (hits[i].atoms[k] >= n) { bad.store(1); return; }
                a.push_back(hits[i].atoms[k]);
            }
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    };
    const uint32_t nr = (uint32_t)runs.size();
    if (nr >= 64) parallel_for(nr, 16, collect);
    else for (uint32_t r = 0; r < nr; ++r) collect(r);
    if (bad.load()) return fail(TM_E_INVALID, "tm_batch_pick_records: atom index out of range");
    for (uint32_t r = 0; r < nr; ++r) first[r + 1] = first[r] + atoms[r].size();
    *count = first[nr];
    if (first[nr] > capacity) return fail(TM_E_OVERFLOW, "tm_batch_pick_records: output too small");
    const uint64_t text_end = b->h_text_off[b->n_structs];
    // Two passes per slice of runs: where every record starts (a load from the offsets' table per atom), then the copies
    // with the lines of the record a few rows ahead asked for -- the text is a few hundred megabytes that no cache holds,
    // and a copy that waits for its own miss costs a hundred nanoseconds (the team of host threads is usually busy
    // reading the next batch's files, so this often runs on the calling thread alone)
    auto fill = [&](uint32_t r0, uint32_t r1) {
        std::vector<uint64_t> at;
        at.reserve(first[r1] - first[r0]);
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t s = hits[runs[r].first].structure;
            uint64_t o = first[r];
            for (uint32_t atom : atoms[r]) {
                keys[o++] = ((uint64_t)s << 32) | atom;
                at.push_back(b->h_text_off[s] + b->h_line_off[b->s_off[s] + atom]);
            }
        }
        constexpr size_t kAhead = 12;
        const char *text = b->h_text.p;
        for (size_t i = 0; i < at.size(); ++i) {
            if (i + kAhead < at.size() && at[i + kAhead] < text_end) {
                __builtin_prefetch(text + at[i + kAhead]);
                __builtin_prefetch(text + std::min<uint64_t>(at[i + kAhead] + 64, text_end - 1));
            }
            const uint64_t take = at[i] < text_end ? std::min<uint64_t>(width, text_end - at[i]) : 0;
            uint8_t *dst = lines + (first[r0] + i) * width;
            if (take) std::memcpy(dst, text + at[i], take);
            if (take < width) std::memset(dst + take, '\n', width - take);
        }
    };
    const uint32_t slices = std::min<uint32_t>(std::max(1u, nr / 64), host_team());
    if (slices > 1) {
        bool done = false;
        for (int t = 0; t < HostPool::kTeams && !done; ++t)
            done = HostPool::get(t).run(slices, [&](uint32_t k) { fill((uint32_t)((uint64_t)nr * k / slices), (uint32_t)((uint64_t)nr * (k + 1) / slices)); }, false);
        if (!done) fill(0, nr);
    } else {
        fill(0, nr);
    }
    return TM_OK;
}

int tm_batch_text_layout(tm_batch *b, const char **base, uint64_t *offsets) {
    if (!b || !base || !offsets) return fail(TM_E_INVALID, "NULL argument");
    if (b->h_text_off.size() != (size_t)b->n_structs + 1) return fail(TM_E_INVALID, "batch was not created from files");
    *base = b->h_text.p;
    std::memcpy(offsets, b->h_text_off.data(), b->h_text_off.size() * sizeof(uint64_t));
    return TM_OK;
}

int tm_batch_file_text(tm_batch *b, uint32_t file, const char **text, uint64_t *length) {
    if (!b || !text || !length) return fail(TM_E_INVALID, "NULL argument");
    if (b->h_text_off.size() != (size_t)b->n_structs + 1) return fail(TM_E_INVALID, "batch was not created from files");
    if (file >= b->n_structs) return fail(TM_E_INVALID, "file index out of range");
    *text = b->h_text.p + b->h_text_off[file];
    *length = b->h_text_off[file + 1] - b->h_text_off[file] - 1;  // without the newline added after the file
    return TM_OK;
}

}  // extern "C"
