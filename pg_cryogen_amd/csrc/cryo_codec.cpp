/*
 * cryo_codec.cpp -- host side of the C ABI declared in include/cryo_codec.h.
 *
 * Thin by design: argument checks, HIP plumbing (stream, events, buffers) and
 * kernel launches.  All codec arithmetic is in the .hip kernels.  No CPU codec
 * exists in this library: if HIP or the device is unavailable every entry
 * point fails with CRYO_E_NODEV / CRYO_E_HIP.
 */
#include "cryo_codec.h"
#include "kernels.h"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <pthread.h>
#include <signal.h>
#include <thread>
#include <unordered_map>
#include <vector>

/* A few host threads kept by a handle (staging copies of the K-block calls) or by the multi-GPU dispatcher (one per
 * further device).  The caller of the C ABI is a PostgreSQL backend: workers are created with every signal blocked
 * (the backend's SIGUSR1/SIGTERM/SIGINT handlers must only ever run on its own thread), they are created once and
 * joined when their owner is closed, and a worker that cannot be started just is not there: run() then executes
 * the shares on the calling thread.  Nothing here throws. */
namespace {
class WorkerPool {
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned n_ = 0, next_ = 0, running_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;

    void loop()
    {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || (gen_ != seen && next_ < n_); });
            if (stop_) return;
            seen = gen_;
            while (next_ < n_) {
                const unsigned i = next_++;
                running_++;
                lk.unlock();
                (*job_)(i);
                lk.lock();
                running_--;
            }
            if (running_ == 0) done_.notify_all();
        }
    }

public:
    /* cpus: where the workers may run (the GPU's NUMA node for the staging workers; nullptr: anywhere) */
    explicit WorkerPool(unsigned workers, const cpu_set_t *cpus = nullptr) noexcept
    {
        sigset_t all, old;
        sigfillset(&all);
        const bool masked = pthread_sigmask(SIG_SETMASK, &all, &old) == 0;
        for (unsigned i = 0; i < workers; i++) {
            try { th_.emplace_back([this] { loop(); }); } catch (...) { break; } /* EAGAIN, bad_alloc: fewer workers */
            if (cpus) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof *cpus, cpus);
        }
        if (masked) (void)pthread_sigmask(SIG_SETMASK, &old, nullptr);
    }
    ~WorkerPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
    }
    unsigned workers() const { return (unsigned)th_.size(); }
    /* f(0) .. f(n-1), spread over the workers and the calling thread; returns when all have run.  f must not throw. */
    void run(unsigned n, const std::function<void(unsigned)> &f) noexcept
    {
        if (n == 0) return;
        if (th_.empty() || n == 1) { for (unsigned i = 0; i < n; i++) f(i); return; }
        std::unique_lock<std::mutex> lk(mu_);
        job_ = &f; n_ = n; next_ = 0; gen_++;
        cv_.notify_all();
        while (next_ < n_) {
            const unsigned i = next_++;
            running_++;
            lk.unlock();
            f(i);
            lk.lock();
            running_--;
        }
        done_.wait(lk, [&] { return running_ == 0; });
        job_ = nullptr; n_ = 0;
    }
};

/* C++ exceptions (bad_alloc from the small host-side vectors) never cross the C ABI */
template <class F>
int guarded(F &&f) noexcept
{
    try { return f(); } catch (const std::bad_alloc &) { return CRYO_E_NOMEM; } catch (...) { return CRYO_E_HIP; }
}
} // namespace

struct cryo_codec {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char err[256] = {0};
    cryo_codec_counters ctr = {};
    /* scratch for the single-block host API */
    uint8_t *d_in = nullptr, *d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    uint64_t *d_off = nullptr;
    uint32_t *d_size = nullptr;
    int32_t *d_status = nullptr;
    /* zstd decode workspace */
    void *d_ws = nullptr;
    size_t ws_cap = 0;
    /* host-buffer batch API: grow-only device buffers and one pinned staging buffer */
    uint8_t *hb_src = nullptr, *hb_dst = nullptr, *hb_meta = nullptr;
    size_t hb_src_cap = 0, hb_dst_cap = 0, hb_meta_cap = 0;
    void *pin = nullptr;
    size_t pin_cap = 0;
    /* pipelined K-block calls: two pinned input and two pinned output staging buffers, a transfer stream for the
     * device-to-host direction, events */
    void *pipe_pin[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pipe_pin_cap[4] = {0, 0, 0, 0};
    hipStream_t xfer = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    /* side streams of the zstd batch pipeline (created on first use) */
    cryo::ZstdAux aux = {};
    bool have_aux = false;
    /* options (cryo_codec_set_option) */
    cryo::Lz4DecodeOpts lz4_opts = {};
    bool lz4_side_failed = false; /* the optional side stream could not be created: not tried again */
    int zstd_path = 0;
    size_t pipe_min_bytes = (size_t)64 << 20;
    /* NUMA: the cpus of the node this GPU hangs on (sysfs local_cpulist of its PCI function, cut to what the process may
     * use); the staging workers run there and the pinned buffers are allocated from there */
    cpu_set_t local_cpus;
    bool have_local_cpus = false;
    int numa_local = 1;     /* CRYO_OPT_NUMA_LOCAL */
    int64_t ws_keep = -1;   /* CRYO_OPT_WORKSPACE_KEEP_BYTES: -1 = keep everything between calls */
    size_t ws_max = 0;      /* CRYO_OPT_WORKSPACE_MAX_BYTES: 0 = automatic */
    /* staging-copy workers (created by the first K-block call that is large enough to want them) */
    WorkerPool *pool = nullptr;
    /* device-resident block pool (CRYO_OPT_POOL_BYTES): decoded blocks of keyed calls, first in first out */
    struct PoolSlot { uint64_t key = 0, fp = 0; uint32_t csize = 0; bool valid = false; };
    size_t pool_bytes = 0, pool_block = 0;
    uint8_t *d_pool = nullptr;
    std::vector<PoolSlot> pool_slots;
    std::unordered_map<uint64_t, uint32_t> pool_index; /* key -> slot */
    uint32_t pool_head = 0;                            /* next slot to fill */
    cryo_codec_transfer_counters xfer_ctr = {};
};

static void pool_drop(cryo_codec *c);

namespace {

int fail(cryo_codec *c, hipError_t e, const char *what)
{
    if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    return CRYO_E_HIP;
}

#define HIP_TRY(c, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail((c), e_, #call);                                         \
    } while (0)

/* One handle = one GPU, but the calling thread's current device is whatever the process last set (another
 * handle's cryo_codec_open, torch.cuda.set_device, ...).  Every entry point that allocates, launches or records
 * events makes the handle's device current for its duration and restores the caller's afterwards. */
struct DevGuard {
    int prev = -1;
    bool switched = false;
    explicit DevGuard(const cryo_codec *c)
    {
        if (!c) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DevGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

bool method_ok(int m) { return m == CRYO_METHOD_LZ4 || m == CRYO_METHOD_ZSTD; }

int ensure(cryo_codec *c, uint8_t **p, size_t *cap, size_t need)
{
    if (*cap >= need) return CRYO_OK;
    if (*p) { HIP_TRY(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIP_TRY(c, hipMalloc((void **)p, need));
    *cap = need;
    return CRYO_OK;
}

/* the calling thread on the GPU's node for the lifetime of the object (restored afterwards): a pinned buffer is placed
 * where the thread that allocates it runs, and a K-block call's share of the staging copies runs on the caller */
struct ScopedLocalCpus {
    cpu_set_t old;
    bool active = false;
    explicit ScopedLocalCpus(const cryo_codec *c)
    {
        if (!c || !c->have_local_cpus || !c->numa_local) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
        active = pthread_setaffinity_np(pthread_self(), sizeof c->local_cpus, &c->local_cpus) == 0;
    }
    ~ScopedLocalCpus() { if (active) (void)pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};

int ensure_pinned(cryo_codec *c, size_t need)
{
    if (c->pin_cap >= need) return CRYO_OK;
    if (c->pin) { HIP_TRY(c, hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
    need += need / 4; /* grow-only, with head room */
    ScopedLocalCpus numa_(c);
    hipError_t e = hipHostMalloc(&c->pin, need, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return CRYO_E_NOMEM; }
    if (e != hipSuccess) return fail(c, e, "hipHostMalloc");
    c->pin_cap = need;
    return CRYO_OK;
}

int ensure_pipe(cryo_codec *c, int which, size_t need)
{
    if (c->pipe_pin_cap[which] >= need) return CRYO_OK;
    if (c->pipe_pin[which]) { HIP_TRY(c, hipHostFree(c->pipe_pin[which])); c->pipe_pin[which] = nullptr; c->pipe_pin_cap[which] = 0; }
    need += need / 8;
    ScopedLocalCpus numa_(c);
    hipError_t e = hipHostMalloc(&c->pipe_pin[which], need, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return CRYO_E_NOMEM; }
    if (e != hipSuccess) return fail(c, e, "hipHostMalloc");
    c->pipe_pin_cap[which] = need;
    return CRYO_OK;
}

int ensure_pipe_streams(cryo_codec *c)
{
    if (c->xfer) return CRYO_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->xfer, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
    }
    return CRYO_OK;
}

/* host copies of a K-block call, spread over a few threads (one thread moves ~8 GB/s; the staging copies of a
 * 4096-block call were its longest part) */
struct CopyJob { void *dst; const void *src; size_t len; };
unsigned host_threads()
{
    static const unsigned v = [] {
        const char *e = getenv("CRYO_HOST_THREADS");
        unsigned t = e ? (unsigned)atoi(e) : 8u, hw = std::thread::hardware_concurrency();
        if (hw && t > hw) t = hw;
        return t < 1u ? 1u : t;
    }();
    return v;
}
void parallel_copy(cryo_codec *c, const std::vector<CopyJob> &jobs)
{
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.len;
    if (total >= (4u << 20) && !c->pool && host_threads() > 1u)
        c->pool = new (std::nothrow) WorkerPool(host_threads() - 1u, c->have_local_cpus && c->numa_local ? &c->local_cpus : nullptr);
    const unsigned T = (total < (4u << 20) || !c->pool) ? 1u : c->pool->workers() + 1u;
    if (T == 1u) { for (const CopyJob &j : jobs) memcpy(j.dst, j.src, j.len); return; }
    /* equal byte shares: share t takes the jobs (or parts of jobs) covering bytes [t, t+1) * total / T */
    const std::function<void(unsigned)> share = [&](unsigned t) {
        const size_t lo = total * t / T, hi = total * (t + 1) / T;
        size_t pos = 0;
        for (const CopyJob &j : jobs) {
            const size_t a = pos > lo ? pos : lo, b = pos + j.len < hi ? pos + j.len : hi;
            if (a < b) memcpy((uint8_t *)j.dst + (a - pos), (const uint8_t *)j.src + (a - pos), b - a);
            pos += j.len;
            if (pos >= hi) break;
        }
    };
    c->pool->run(T, share);
}

/* the most workspace a call may plan for: the option, or 70 % of the free device memory plus what the handle holds */
size_t ws_budget(cryo_codec *c)
{
    if (c->ws_max) return c->ws_max;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return ~(size_t)0; }
    return fr / 10u * 7u + c->ws_cap;
}
/* host-buffer calls end synchronised: give back what the handle holds on the device beyond CRYO_OPT_WORKSPACE_KEEP_BYTES --
 * the kernels' workspace first (LZ4 sequence index, zstd tiles), then the staging areas of the host-buffer calls themselves
 * (n x block_size each: round 4 left those out, and a backend kept GiBs of them after one large call).  Every path that ends
 * a host-buffer call comes through here: the public wrappers and each handle's share of a cryo_multi_* call (multi_run). */
void ws_trim_after_call(cryo_codec *c)
{
    if (c->ws_keep < 0) return;
    const size_t keep = (size_t)c->ws_keep;
    const size_t hb = c->hb_src_cap + c->hb_dst_cap + c->hb_meta_cap;
    const bool drop_ws = c->d_ws && c->ws_cap > keep;
    const bool drop_hb = hb != 0 && hb + (drop_ws ? 0 : c->ws_cap) > keep;
    if (!drop_ws && !drop_hb) return;
    DevGuard dev_(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->xfer) (void)hipStreamSynchronize(c->xfer);
    if (drop_ws) {
        (void)hipFree(c->d_ws);
        c->d_ws = nullptr;
        c->ws_cap = 0;
    }
    if (drop_hb) {
        auto drop = [](uint8_t *&p, size_t &cap) { if (p) (void)hipFree(p); p = nullptr; cap = 0; };
        drop(c->hb_src, c->hb_src_cap);
        drop(c->hb_dst, c->hb_dst_cap);
        drop(c->hb_meta, c->hb_meta_cap);
    }
}

int ensure_ws(cryo_codec *c, size_t need)
{
    if (c->ws_cap >= need) return CRYO_OK;
    if (c->d_ws) { HIP_TRY(c, hipFree(c->d_ws)); c->d_ws = nullptr; c->ws_cap = 0; }
    HIP_TRY(c, hipMalloc(&c->d_ws, need));
    c->ws_cap = need;
    return CRYO_OK;
}

} // namespace

extern "C" {

const char *cryo_codec_version(void)
{
    return "cryo-codec 0.2 gfx950 (lz4 block format as liblz4 1.9.3; zstd frames as libzstd 1.4.8)";
}

} /* extern "C" */
/* the cpus of the NUMA node the handle's GPU hangs on: /sys/bus/pci/devices/<bus id>/local_cpulist, cut to what the process
 * may use.  Measured on the 2-socket bench host (profiles/r04_host_api.txt): a 4 096 x 128 KiB decompress call moved between
 * 25 and 32 GB/s from call to call with its staging threads and pinned buffers wherever the scheduler put them. */
static void discover_local_cpus(cryo_codec *c)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *p = bus; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return;
    char line[1024] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t allowed, want;
    if (pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) != 0) return;
    CPU_ZERO(&want);
    int n = 0;
    for (char *p = line; *p;) { /* "64-127,192-255" */
        char *end;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long k = a; k <= b && k < CPU_SETSIZE; k++)
            if (k >= 0 && CPU_ISSET((int)k, &allowed)) { CPU_SET((int)k, &want); n++; }
        p = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    if (n > 0 && n < CPU_COUNT(&allowed)) { c->local_cpus = want; c->have_local_cpus = true; } /* one node only: nothing to choose */
}
extern "C" {

int cryo_codec_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return CRYO_E_NODEV;
    return n;
}

int cryo_codec_open(int device, cryo_codec **out)
{
    if (!out) return CRYO_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CRYO_E_NODEV;
    if (device < 0 || device >= n) return CRYO_E_ARG;
    cryo_codec *c = new (std::nothrow) cryo_codec;
    if (!c) return CRYO_E_NOMEM;
    c->device = device;
    if (const char *e = cryo_tuning_env("CRYO_PIPE_MIN_MB")) c->pipe_min_bytes = (size_t)atoll(e) << 20; /* 0 = always, huge = never */
    if (const char *e = cryo_tuning_env("CRYO_LZ4_DECODE_PATH")) c->lz4_opts.path = atoi(e);             /* tuning aids: the options' */
    if (const char *e = cryo_tuning_env("CRYO_LZ4_INDEX_WALKERS")) c->lz4_opts.walkers = atoi(e);        /* initial values          */
    if (const char *e = cryo_tuning_env("CRYO_ZSTD_DECODE_PATH")) c->zstd_path = atoi(e);
    DevGuard dev_(c); /* the caller's current device is restored on return */
    hipError_t e = hipSuccess;
    if (!dev_.switched && dev_.prev != device) e = hipSetDevice(device); /* no current device yet, or the switch failed: report it */
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_off, sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_size, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_status, sizeof(int32_t));
    if (e == hipSuccess) { /* what one round of the LZ4 decoder holds depends on the device's compute units (a partition has fewer) */
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->lz4_opts.cus = cus;
        else (void)hipGetLastError();
    }
    if (e != hipSuccess) {
        cryo_codec_close(c);
        return CRYO_E_HIP;
    }
    discover_local_cpus(c);
    *out = c;
    return CRYO_OK;
}

void cryo_codec_close(cryo_codec *c)
{
    if (!c) return;
    DevGuard dev_(c);
    delete c->pool;
    c->pool = nullptr;
    if (c->d_pool) { if (c->stream) (void)hipStreamSynchronize(c->stream); (void)hipFree(c->d_pool); c->d_pool = nullptr; }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_size) (void)hipFree(c->d_size);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->hb_src) (void)hipFree(c->hb_src);
    if (c->hb_dst) (void)hipFree(c->hb_dst);
    if (c->hb_meta) (void)hipFree(c->hb_meta);
    if (c->pin) (void)hipHostFree(c->pin);
    for (int i = 0; i < 4; i++) if (c->pipe_pin[i]) (void)hipHostFree(c->pipe_pin[i]);
    if (c->xfer) { (void)hipStreamSynchronize(c->xfer); (void)hipStreamDestroy(c->xfer); }
    for (int i = 0; i < 2; i++) {
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_k[i]) (void)hipEventDestroy(c->ev_k[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    for (int l = 0; l < cryo::kZstdLanes; l++) {
        if (c->aux.lane[l]) { (void)hipStreamSynchronize(c->aux.lane[l]); (void)hipStreamDestroy(c->aux.lane[l]); }
        if (c->aux.side[l]) { (void)hipStreamSynchronize(c->aux.side[l]); (void)hipStreamDestroy(c->aux.side[l]); }
        if (c->aux.join[l]) (void)hipEventDestroy(c->aux.join[l]);
        if (c->aux.planned[l]) (void)hipEventDestroy(c->aux.planned[l]);
        if (c->aux.seqs_done[l]) (void)hipEventDestroy(c->aux.seqs_done[l]);
    }
    if (c->aux.fork) (void)hipEventDestroy(c->aux.fork);
    if (c->lz4_opts.side) { (void)hipStreamSynchronize(c->lz4_opts.side); (void)hipStreamDestroy(c->lz4_opts.side); }
    if (c->lz4_opts.fork) (void)hipEventDestroy(c->lz4_opts.fork);
    if (c->lz4_opts.join) (void)hipEventDestroy(c->lz4_opts.join);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *cryo_codec_last_error(const cryo_codec *c) { return c ? c->err : ""; }

int cryo_codec_trim(cryo_codec *c)
{
    if (!c) return CRYO_E_ARG;
    DevGuard dev_(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->xfer) HIP_TRY(c, hipStreamSynchronize(c->xfer));
    if (c->have_aux)
        for (int l = 0; l < cryo::kZstdLanes; l++)
            if (c->aux.lane[l]) HIP_TRY(c, hipStreamSynchronize(c->aux.lane[l]));
    if (c->have_aux)
        for (int l = 0; l < cryo::kZstdLanes; l++)
            if (c->aux.side[l]) HIP_TRY(c, hipStreamSynchronize(c->aux.side[l]));
    auto drop = [](auto *&p, size_t &cap) { if (p) (void)hipFree(p); p = nullptr; cap = 0; };
    { void *w = c->d_ws; if (w) (void)hipFree(w); c->d_ws = nullptr; c->ws_cap = 0; }
    drop(c->d_in, c->in_cap);
    drop(c->d_out, c->out_cap);
    drop(c->hb_src, c->hb_src_cap);
    drop(c->hb_dst, c->hb_dst_cap);
    drop(c->hb_meta, c->hb_meta_cap);
    if (c->pin) { (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_cap = 0; }
    for (int i = 0; i < 4; i++)
        if (c->pipe_pin[i]) { (void)hipHostFree(c->pipe_pin[i]); c->pipe_pin[i] = nullptr; c->pipe_pin_cap[i] = 0; }
    return CRYO_OK;
}

int cryo_codec_set_option(cryo_codec *c, int option, int64_t value)
{
    if (!c) return CRYO_E_ARG;
    switch (option) {
    case CRYO_OPT_LZ4_DECODE_PATH:
        if (value < 0 || value > 3) return CRYO_E_ARG;
        c->lz4_opts.path = (int)value;
        return CRYO_OK;
    case CRYO_OPT_LZ4_INDEX_WALKERS:
        if (value < 0 || value > 64 || (value & (value - 1)) != 0) return CRYO_E_ARG;
        c->lz4_opts.walkers = (int)value;
        return CRYO_OK;
    case CRYO_OPT_LZ4_DECODE_WAVES:
        if (value < 0 || value > 2) return CRYO_E_ARG;
        c->lz4_opts.waves = (int)value;
        return CRYO_OK;
    case CRYO_OPT_PIPE_MIN_BYTES:
        if (value < 0) return CRYO_E_ARG;
        c->pipe_min_bytes = (size_t)value;
        return CRYO_OK;
    case CRYO_OPT_ZSTD_DECODE_PATH:
        if (value < 0 || value > 3) return CRYO_E_ARG;
        c->zstd_path = (int)value;
        return CRYO_OK;
    case CRYO_OPT_WORKSPACE_KEEP_BYTES:
        if (value < -1) return CRYO_E_ARG;
        c->ws_keep = value;
        return CRYO_OK;
    case CRYO_OPT_WORKSPACE_MAX_BYTES:
        if (value < 0) return CRYO_E_ARG;
        c->ws_max = (size_t)value;
        return CRYO_OK;
    case CRYO_OPT_NUMA_LOCAL:
        if (value != 0 && value != 1) return CRYO_E_ARG;
        c->numa_local = (int)value; /* workers already started keep their placement */
        return CRYO_OK;
    case CRYO_OPT_POOL_BYTES: {
        if (value < 0) return CRYO_E_ARG;
        DevGuard dev_(c);
        c->pool_bytes = (size_t)value;
        if (c->pool_bytes == 0 && c->d_pool) pool_drop(c); /* a new size takes effect at the next keyed call */
        return CRYO_OK;
    }
    default:
        return CRYO_E_ARG;
    }
}

int cryo_codec_get_option(const cryo_codec *c, int option, int64_t *value)
{
    if (!c || !value) return CRYO_E_ARG;
    switch (option) {
    case CRYO_OPT_LZ4_DECODE_PATH: *value = c->lz4_opts.path; return CRYO_OK;
    case CRYO_OPT_LZ4_INDEX_WALKERS: *value = c->lz4_opts.walkers; return CRYO_OK;
    case CRYO_OPT_LZ4_DECODE_WAVES: *value = c->lz4_opts.waves; return CRYO_OK;
    case CRYO_OPT_PIPE_MIN_BYTES: *value = (int64_t)c->pipe_min_bytes; return CRYO_OK;
    case CRYO_OPT_POOL_BYTES: *value = (int64_t)c->pool_bytes; return CRYO_OK;
    case CRYO_OPT_ZSTD_DECODE_PATH: *value = c->zstd_path; return CRYO_OK;
    case CRYO_OPT_WORKSPACE_KEEP_BYTES: *value = c->ws_keep; return CRYO_OK;
    case CRYO_OPT_WORKSPACE_MAX_BYTES: *value = (int64_t)c->ws_max; return CRYO_OK;
    case CRYO_OPT_NUMA_LOCAL: *value = c->numa_local && c->have_local_cpus ? 1 : 0; return CRYO_OK;
    default: return CRYO_E_ARG;
    }
}
void *cryo_codec_stream(cryo_codec *c) { return c ? (void *)c->stream : nullptr; }

int cryo_codec_sync(cryo_codec *c)
{
    DevGuard dev_(c);
    if (!c) return CRYO_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRYO_OK;
}

size_t cryo_codec_bound(int method, size_t n)
{
    if (method == CRYO_METHOD_LZ4) {
        /* LZ4_compressBound: n + n/255 + 16, 0 above LZ4_MAX_INPUT_SIZE */
        return n > 0x7E000000u ? 0 : n + n / 255 + 16;
    }
    if (method == CRYO_METHOD_ZSTD) {
        /* ZSTD_COMPRESSBOUND: n + n/256 + (n < 128 KiB ? (128 KiB - n) >> 11 : 0) */
        return n + (n >> 8) + (n < (128u << 10) ? ((128u << 10) - n) >> 11 : 0);
    }
    return 0;
}

/* ---- device memory plumbing ---- */
int cryo_dev_alloc(cryo_codec *c, size_t bytes, void **d_ptr)
{
    DevGuard dev_(c);
    if (!c || !d_ptr) return CRYO_E_ARG;
    *d_ptr = nullptr;
    /* +64: the kernels read compressed input in aligned 16-byte pieces (up to 15 bytes past a block's end) */
    hipError_t e = hipMalloc(d_ptr, bytes + 64);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return CRYO_E_NOMEM; }
    if (e != hipSuccess) return fail(c, e, "hipMalloc");
    return CRYO_OK;
}
int cryo_dev_free(cryo_codec *c, void *d_ptr)
{
    DevGuard dev_(c);
    if (!c) return CRYO_E_ARG;
    if (!d_ptr) return CRYO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(d_ptr));
    return CRYO_OK;
}
int cryo_dev_upload(cryo_codec *c, void *d_dst, const void *h_src, size_t bytes)
{
    DevGuard dev_(c);
    if (!c || (bytes && (!d_dst || !h_src))) return CRYO_E_ARG;
    if (!bytes) return CRYO_OK;
    HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRYO_OK;
}
int cryo_dev_download(cryo_codec *c, void *h_dst, const void *d_src, size_t bytes)
{
    DevGuard dev_(c);
    if (!c || (bytes && (!h_dst || !d_src))) return CRYO_E_ARG;
    if (!bytes) return CRYO_OK;
    HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRYO_OK;
}
int cryo_dev_memset(cryo_codec *c, void *d_dst, int value, size_t bytes)
{
    DevGuard dev_(c);
    if (!c || (bytes && !d_dst)) return CRYO_E_ARG;
    if (!bytes) return CRYO_OK;
    HIP_TRY(c, hipMemsetAsync(d_dst, value, bytes, c->stream));
    return CRYO_OK;
}

/* ---- batch codec ---- */
int cryo_codec_compress_batch(cryo_codec *c, int method, int param, const void *d_src,
                              uint64_t src_stride, uint32_t block_size, uint64_t n_blocks,
                              void *d_dst, uint64_t dst_stride, uint32_t *d_out_size,
                              int32_t *d_status)
{
    DevGuard dev_(c);
    if (!c || !method_ok(method) || block_size == 0) return CRYO_E_ARG;
    if (n_blocks == 0) return CRYO_OK;
    if (!d_src || !d_dst || !d_out_size || !d_status || src_stride < block_size) return CRYO_E_ARG;
    if (dst_stride < cryo_codec_bound(method, block_size)) return CRYO_E_DSTSIZE;
    if (method == CRYO_METHOD_LZ4) {
        HIP_TRY(c, cryo::launch_lz4_compress(c->stream, (const uint8_t *)d_src, src_stride, block_size,
                                             n_blocks, (uint8_t *)d_dst, dst_stride, param,
                                             d_out_size, d_status));
    } else {
        /* levels whose strategy has a kernel (fast, dfast, greedy, lazy, lazy2: -5..10); others: CRYO_E_UNSUPPORTED */
        if (!cryo::zstd_compress_supported(param, block_size)) return CRYO_E_UNSUPPORTED;
        const size_t need = cryo::zstd_compress_workspace(n_blocks, param, block_size);
        int rc = ensure_ws(c, need);
        if (rc != CRYO_OK) return rc;
        HIP_TRY(c, cryo::launch_zstd_compress(c->stream, (const uint8_t *)d_src, src_stride, block_size, n_blocks,
                                              (uint8_t *)d_dst, dst_stride, param, d_out_size, d_status, c->d_ws,
                                              c->ws_cap));
    }
    c->ctr.blocks_compressed += n_blocks;
    c->ctr.bytes_in += n_blocks * (uint64_t)block_size;
    c->ctr.launches++;
    return CRYO_OK;
}

int cryo_codec_decompress_batch(cryo_codec *c, int method, const void *d_src,
                                const uint64_t *d_src_off, const uint32_t *d_src_size, void *d_dst,
                                uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks,
                                int32_t *d_status)
{
    DevGuard dev_(c);
    if (!c || !method_ok(method) || block_size == 0) return CRYO_E_ARG;
    if (n_blocks == 0) return CRYO_OK;
    if (!d_src || !d_src_off || !d_src_size || !d_dst || !d_status || dst_stride < block_size)
        return CRYO_E_ARG;
    if (method == CRYO_METHOD_LZ4) {
        const size_t need = cryo::lz4_decompress_workspace(n_blocks, block_size, c->lz4_opts);
        if (need != 0) {
            int rc = ensure_ws(c, need);
            if (rc != CRYO_OK) return rc;
        }
        if (!c->lz4_opts.side && !c->lz4_side_failed) {
            /* the side stream of the decoder's last round (lz4_dec2.hip): lowest priority, made once -- stream and both events or
             * none of them: a handle that cannot have them decodes on one stream (an optimisation, never an error) */
            int least = 0, greatest = 0;
            hipStream_t side = nullptr;
            hipEvent_t fork = nullptr, join = nullptr;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least) == hipSuccess &&
                hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess) {
                c->lz4_opts.side = side; c->lz4_opts.fork = fork; c->lz4_opts.join = join;
            } else {
                (void)hipGetLastError();
                if (join) (void)hipEventDestroy(join);
                if (fork) (void)hipEventDestroy(fork);
                if (side) (void)hipStreamDestroy(side);
                c->lz4_side_failed = true;
            }
        }
        HIP_TRY(c, cryo::launch_lz4_decompress(c->stream, (const uint8_t *)d_src, d_src_off, d_src_size,
                                               (uint8_t *)d_dst, dst_stride, block_size, n_blocks,
                                               d_status, need ? c->d_ws : nullptr, need ? c->ws_cap : 0, c->lz4_opts));
    } else {
        const size_t need = cryo::zstd_decompress_workspace(n_blocks, block_size, c->zstd_path, ws_budget(c));
        int rc = ensure_ws(c, need);
        if (rc != CRYO_OK) return rc;
        if (!c->have_aux) {
            for (int l = 0; l < cryo::kZstdLanes; l++) {
                HIP_TRY(c, hipStreamCreateWithFlags(&c->aux.lane[l], hipStreamNonBlocking));
                HIP_TRY(c, hipEventCreateWithFlags(&c->aux.join[l], hipEventDisableTiming));
                HIP_TRY(c, hipStreamCreateWithFlags(&c->aux.side[l], hipStreamNonBlocking));
                HIP_TRY(c, hipEventCreateWithFlags(&c->aux.planned[l], hipEventDisableTiming));
                HIP_TRY(c, hipEventCreateWithFlags(&c->aux.seqs_done[l], hipEventDisableTiming));
            }
            HIP_TRY(c, hipEventCreateWithFlags(&c->aux.fork, hipEventDisableTiming));
            c->have_aux = true;
        }
        HIP_TRY(c, cryo::launch_zstd_decompress(c->stream, (const uint8_t *)d_src, d_src_off, d_src_size,
                                                (uint8_t *)d_dst, dst_stride, block_size, n_blocks,
                                                d_status, c->d_ws, c->ws_cap, &c->aux, c->zstd_path));
    }
    c->ctr.blocks_decompressed += n_blocks;
    c->ctr.bytes_out += n_blocks * (uint64_t)block_size;
    c->ctr.launches++;
    return CRYO_OK;
}

/* ---- single block, host buffers ---- */
int cryo_codec_compress_block(cryo_codec *c, int method, int param, const void *h_src,
                              size_t block_size, void *h_dst, size_t dst_cap, size_t *out_size)
{
    DevGuard dev_(c);
    if (!c || !h_src || !h_dst || !out_size || !method_ok(method)) return CRYO_E_ARG;
    if (block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    const size_t bound = cryo_codec_bound(method, block_size);
    if (dst_cap < bound) return CRYO_E_DSTSIZE;
    int rc;
    if ((rc = ensure(c, &c->d_in, &c->in_cap, block_size)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->d_out, &c->out_cap, bound)) != CRYO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in, h_src, block_size, hipMemcpyHostToDevice, c->stream));
    rc = cryo_codec_compress_batch(c, method, param, c->d_in, block_size, (uint32_t)block_size, 1,
                                   c->d_out, bound, c->d_size, c->d_status);
    if (rc != CRYO_OK) return rc;
    uint32_t csize = 0;
    int32_t st = 0;
    HIP_TRY(c, hipMemcpyAsync(&csize, c->d_size, sizeof csize, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (st != CRYO_OK) return st;
    if (csize == 0 || csize > bound) return CRYO_E_HIP;
    HIP_TRY(c, hipMemcpyAsync(h_dst, c->d_out, csize, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out_size = csize;
    return CRYO_OK;
}

int cryo_codec_decompress_block(cryo_codec *c, int method, const void *h_src, size_t src_size,
                                void *h_dst, size_t block_size)
{
    DevGuard dev_(c);
    if (!c || !h_src || !h_dst || !method_ok(method)) return CRYO_E_ARG;
    if (block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (src_size == 0 || src_size > 0xFFFFFFFFu) return CRYO_E_CORRUPT;
    int rc;
    if ((rc = ensure(c, &c->d_in, &c->in_cap, src_size)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->d_out, &c->out_cap, block_size)) != CRYO_OK) return rc;
    const uint64_t off = 0;
    const uint32_t sz = (uint32_t)src_size;
    HIP_TRY(c, hipMemcpyAsync(c->d_in, h_src, src_size, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_off, &off, sizeof off, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_size, &sz, sizeof sz, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); /* off/sz are stack variables */
    rc = cryo_codec_decompress_batch(c, method, c->d_in, c->d_off, c->d_size, c->d_out, block_size,
                                     (uint32_t)block_size, 1, c->d_status);
    if (rc != CRYO_OK) return rc;
    int32_t st = 0;
    HIP_TRY(c, hipMemcpyAsync(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (st != CRYO_OK) return st;
    HIP_TRY(c, hipMemcpyAsync(h_dst, c->d_out, block_size, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRYO_OK;
}

/* ---- K blocks, host buffers ---- */
} /* extern "C" */

/* ---- pipelined K-block calls -------------------------------------------------------------------------------
 * A call with tens of megabytes is cut into chunks; while chunk c runs on the GPU, chunk c+1 is gathered into a
 * pinned buffer by a few host threads and chunk c-1 travels back on a second stream and is scattered to the
 * caller's memory.  One chunk's life: host gather -> H2D (codec stream) -> kernel(s) (codec stream) -> D2H
 * (transfer stream) -> host scatter.  The one-shot path (one H2D of everything, the kernel, one D2H from pageable
 * memory, single-threaded copies) reached 13-19 GB/s on 4096 x 128 KiB, a quarter of the link. */
static size_t pipe_chunk_blocks(size_t n, size_t block_size, int method)
{
    (void)method;
    size_t k = (n + 7) / 8;
    const size_t min_blocks = (8u << 20) / block_size + 1; /* at least 8 MiB per chunk */
    if (k < min_blocks) k = min_blocks;
    return (k + 63) & ~(size_t)63;
}
/* one block is one wavefront's serial job: a launch needs thousands of blocks to fill the chip, so kernels run per
 * chunk only when a chunk still has that many (LZ4 decode of 128 KiB blocks); otherwise the transfers are chunked
 * around ONE launch over the whole call (the encoders took 8 x longer cut in eight) */
static bool pipe_kernel_per_chunk(size_t chunk_blocks, int method, bool encode)
{
    return !encode && method == CRYO_METHOD_LZ4 && chunk_blocks >= 512;
}

/* h_src / h_dst: contiguous K-block buffers; or (multi-GPU shares, where a handle's blocks are every G-th of the call)
 * one pointer per block in h_src_each / h_dst_each */
static int compress_blocks_piped(cryo_codec *c, int method, int param, const uint8_t *h_src, const void *const *h_src_each,
                                 size_t block_size, size_t n, uint8_t *h_dst, void *const *h_dst_each, size_t dst_stride,
                                 uint32_t *h_out_size)
{
    const size_t bound = cryo_codec_bound(method, block_size);
    const size_t dstride = (bound + 15) & ~(size_t)15;
    const size_t K = pipe_chunk_blocks(n, block_size, method), nch = (n + K - 1) / K;
    int rc;
    if ((rc = ensure_pipe_streams(c)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * dstride + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, n * 16 + 64)) != CRYO_OK) return rc;
    if ((rc = ensure_pinned(c, n * 8 + 64)) != CRYO_OK) return rc;
    for (int b = 0; b < 2; b++) {
        if ((rc = ensure_pipe(c, b, K * block_size)) != CRYO_OK) return rc;
        if ((rc = ensure_pipe(c, 2 + b, K * dstride)) != CRYO_OK) return rc;
    }
    uint32_t *d_sz = (uint32_t *)c->hb_meta;
    int32_t *d_st = (int32_t *)(c->hb_meta + ((n * 4 + 15) & ~(size_t)15));
    uint32_t *p_sz = (uint32_t *)c->pin;
    int32_t *p_st = (int32_t *)((uint8_t *)c->pin + n * 4);
    /* in: host gather of chunk c+1 overlaps the H2D of chunk c */
    for (size_t ch = 0; ch < nch; ch++) {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n, cnt = hi - lo;
        const int b = (int)(ch & 1);
        if (ch >= 2) HIP_TRY(c, hipEventSynchronize(c->ev_in[b])); /* staging buffer b has left for the device */
        if (h_src) parallel_copy(c, {{c->pipe_pin[b], h_src + lo * block_size, cnt * block_size}});
        else {
            std::vector<CopyJob> jobs;
            for (size_t i = lo; i < hi; i++) jobs.push_back({(uint8_t *)c->pipe_pin[b] + (i - lo) * block_size, h_src_each[i], block_size});
            parallel_copy(c, jobs);
        }
        HIP_TRY(c, hipMemcpyAsync(c->hb_src + lo * block_size, c->pipe_pin[b], cnt * block_size, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_in[b], c->stream));
    }
    rc = cryo_codec_compress_batch(c, method, param, c->hb_src, block_size, (uint32_t)block_size, n, c->hb_dst, dstride, d_sz, d_st);
    if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    HIP_TRY(c, hipMemcpyAsync(p_sz, d_sz, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(p_st, d_st, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        if (p_st[i] != CRYO_OK) return p_st[i];
        if (p_sz[i] == 0 || p_sz[i] > bound) return CRYO_E_HIP;
        h_out_size[i] = p_sz[i];
    }
    /* out: the D2H of chunk c+1 overlaps the host scatter of chunk c; only the bytes the blocks occupy travel */
    auto d2h = [&](size_t ch) -> int {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n;
        HIP_TRY(c, hipMemcpyAsync(c->pipe_pin[2 + (ch & 1)], c->hb_dst + lo * dstride, (hi - lo - 1) * dstride + p_sz[hi - 1],
                                  hipMemcpyDeviceToHost, c->xfer));
        HIP_TRY(c, hipEventRecord(c->ev_out[ch & 1], c->xfer));
        return CRYO_OK;
    };
    if ((rc = d2h(0)) != CRYO_OK) return rc;
    for (size_t ch = 0; ch < nch; ch++) {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n;
        if (ch + 1 < nch && (rc = d2h(ch + 1)) != CRYO_OK) return rc;
        HIP_TRY(c, hipEventSynchronize(c->ev_out[ch & 1]));
        const uint8_t *po = (const uint8_t *)c->pipe_pin[2 + (ch & 1)];
        std::vector<CopyJob> jobs;
        for (size_t i = lo; i < hi; i++) jobs.push_back({h_dst ? (void *)(h_dst + i * dst_stride) : h_dst_each[i], po + (i - lo) * dstride, p_sz[i]});
        parallel_copy(c, jobs);
    }
    return CRYO_OK;
}

/* h_dst: one contiguous K-block buffer; or one destination per block in h_dst_each (the cache's slots, a multi-GPU share):
 * then a block that failed leaves its destination untouched -- the statuses of a chunk travel with its blocks */
static int decompress_blocks_piped(cryo_codec *c, int method, const void *const *h_src, const uint32_t *h_src_size, size_t n,
                                   uint8_t *h_dst, void *const *h_dst_each, size_t block_size, int32_t *h_status)
{
    const size_t K = pipe_chunk_blocks(n, block_size, method), nch = (n + K - 1) / K;
    const bool per_chunk = pipe_kernel_per_chunk(K, method, false);
    /* device layout of the compressed side: [offsets u64 x n][sizes u32 x n][blocks, 16-byte aligned] */
    const size_t o_off = 0, o_sz = n * 8, o_data = (n * 12 + 63) & ~(size_t)63;
    std::vector<uint64_t> pos(n + 1);
    size_t total = 0, max_chunk_in = 0;
    for (size_t i = 0; i < n; i++) {
        if (h_src_size[i] != 0 && !h_src[i]) return CRYO_E_ARG;
        pos[i] = total;
        total += ((size_t)h_src_size[i] + 15) & ~(size_t)15;
    }
    pos[n] = total;
    for (size_t ch = 0; ch < nch; ch++) {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n;
        if (pos[hi] - pos[lo] > max_chunk_in) max_chunk_in = pos[hi] - pos[lo];
    }
    int rc;
    if ((rc = ensure_pipe_streams(c)) != CRYO_OK) return rc;
    if ((rc = ensure_pinned(c, o_data + n * 4 + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, o_data + total + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, n * 16 + 64)) != CRYO_OK) return rc;
    for (int b = 0; b < 2; b++) {
        if ((rc = ensure_pipe(c, b, max_chunk_in + 64)) != CRYO_OK) return rc;
        if ((rc = ensure_pipe(c, 2 + b, K * block_size)) != CRYO_OK) return rc;
    }
    uint8_t *pin = (uint8_t *)c->pin;
    uint64_t *p_off = (uint64_t *)(pin + o_off);
    uint32_t *p_sz = (uint32_t *)(pin + o_sz);
    int32_t *p_st = (int32_t *)(pin + o_data);
    for (size_t i = 0; i < n; i++) { p_off[i] = o_data + pos[i]; p_sz[i] = h_src_size[i]; }
    HIP_TRY(c, hipMemcpyAsync(c->hb_src, pin, o_data, hipMemcpyHostToDevice, c->stream));
    c->xfer_ctr.h2d_bytes += o_data + total;
    c->xfer_ctr.d2h_bytes += n * block_size + n * sizeof(int32_t);
    int32_t *d_st = (int32_t *)c->hb_meta;
    const uint64_t *d_off = (const uint64_t *)(c->hb_src + o_off);
    const uint32_t *d_sz = (const uint32_t *)(c->hb_src + o_sz);
    auto scatter = [&](size_t ch) -> int {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n;
        HIP_TRY(c, hipEventSynchronize(c->ev_out[ch & 1]));
        if (h_dst) { parallel_copy(c, {{h_dst + lo * block_size, c->pipe_pin[2 + (ch & 1)], (hi - lo) * block_size}}); return CRYO_OK; }
        std::vector<CopyJob> jobs;
        for (size_t i = lo; i < hi; i++) {
            if (p_st[i] != CRYO_OK) continue;
            if (!h_dst_each[i]) return CRYO_E_ARG;
            jobs.push_back({h_dst_each[i], (const uint8_t *)c->pipe_pin[2 + (ch & 1)] + (i - lo) * block_size, block_size});
        }
        parallel_copy(c, jobs);
        return CRYO_OK;
    };
    for (size_t ch = 0; ch < nch; ch++) {
        const size_t lo = ch * K, hi = lo + K < n ? lo + K : n, cnt = hi - lo;
        const int b = (int)(ch & 1);
        if (ch >= 2) HIP_TRY(c, hipEventSynchronize(c->ev_in[b]));
        {
            uint8_t *pi = (uint8_t *)c->pipe_pin[b];
            std::vector<CopyJob> jobs;
            for (size_t i = lo; i < hi; i++)
                if (h_src_size[i]) jobs.push_back({pi + (pos[i] - pos[lo]), h_src[i], h_src_size[i]});
            parallel_copy(c, jobs);
        }
        if (pos[hi] > pos[lo])
            HIP_TRY(c, hipMemcpyAsync(c->hb_src + o_data + pos[lo], c->pipe_pin[b], pos[hi] - pos[lo], hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_in[b], c->stream));
        if (!per_chunk) continue;
        rc = cryo_codec_decompress_batch(c, method, c->hb_src, d_off + lo, d_sz + lo, c->hb_dst + lo * block_size, block_size,
                                         (uint32_t)block_size, cnt, d_st + lo);
        if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->xfer); return rc; }
        HIP_TRY(c, hipEventRecord(c->ev_k[b], c->stream));
        if (ch >= 2 && (rc = scatter(ch - 2)) != CRYO_OK) return rc; /* frees output buffer b */
        HIP_TRY(c, hipStreamWaitEvent(c->xfer, c->ev_k[b], 0));
        HIP_TRY(c, hipMemcpyAsync(c->pipe_pin[2 + b], c->hb_dst + lo * block_size, cnt * block_size, hipMemcpyDeviceToHost, c->xfer));
        if (!h_dst) HIP_TRY(c, hipMemcpyAsync(p_st + lo, d_st + lo, cnt * 4, hipMemcpyDeviceToHost, c->xfer)); /* the scatter skips failed blocks */
        HIP_TRY(c, hipEventRecord(c->ev_out[b], c->xfer));
    }
    if (!per_chunk) {
        /* one launch over the whole call, then the D2H of chunk c+1 overlaps the host scatter of chunk c */
        rc = cryo_codec_decompress_batch(c, method, c->hb_src, d_off, d_sz, c->hb_dst, block_size, (uint32_t)block_size, n, d_st);
        if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
        HIP_TRY(c, hipEventRecord(c->ev_k[0], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->xfer, c->ev_k[0], 0));
        if (!h_dst) HIP_TRY(c, hipMemcpyAsync(p_st, d_st, n * 4, hipMemcpyDeviceToHost, c->xfer)); /* before the first chunk's event */
        auto d2h = [&](size_t ch) -> int {
            const size_t lo = ch * K, hi = lo + K < n ? lo + K : n;
            HIP_TRY(c, hipMemcpyAsync(c->pipe_pin[2 + (ch & 1)], c->hb_dst + lo * block_size, (hi - lo) * block_size, hipMemcpyDeviceToHost, c->xfer));
            HIP_TRY(c, hipEventRecord(c->ev_out[ch & 1], c->xfer));
            return CRYO_OK;
        };
        if ((rc = d2h(0)) != CRYO_OK) return rc;
        for (size_t ch = 0; ch < nch; ch++) {
            if (ch + 1 < nch) {
                if (ch >= 1) { /* buffer (ch+1)&1 was scattered in the previous iteration */ }
                if ((rc = d2h(ch + 1)) != CRYO_OK) return rc;
            }
            if ((rc = scatter(ch)) != CRYO_OK) return rc;
        }
        HIP_TRY(c, hipMemcpyAsync(p_st, d_st, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(h_status, p_st, n * 4);
        return CRYO_OK;
    }
    /* the trailing scatters read p_st on the host (which blocks failed): the whole-call status copy goes out behind them,
     * not under them */
    for (size_t ch = nch >= 2 ? nch - 2 : 0; ch < nch; ch++)
        if ((rc = scatter(ch)) != CRYO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(p_st, d_st, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(h_status, p_st, n * 4);
    return CRYO_OK;
}

static bool pipe_worth_it(const cryo_codec *c, size_t n, size_t block_size)
{
    return n * block_size >= c->pipe_min_bytes && n >= 128;
}
/* the four pinned staging buffers of the pipelined calls are K x block_size each: a 4096 x 1 MiB call leaves 2.3 GB of
 * pinned host memory behind.  What exceeds this cap is given back when the call ends. */
static void pipe_trim(cryo_codec *c)
{
    constexpr size_t kKeep = (size_t)256 << 20;
    for (int i = 0; i < 4; i++)
        if (c->pipe_pin_cap[i] > kKeep) { (void)hipHostFree(c->pipe_pin[i]); c->pipe_pin[i] = nullptr; c->pipe_pin_cap[i] = 0; }
}
extern "C" {

/* K blocks from / to host memory.  Device buffers and the pinned staging buffer live in the handle
 * (grow-only); transfers are bulk: one H2D of the K blocks, one D2H of the K output slots. */
static int compress_blocks_body(cryo_codec *c, int method, int param, const void *h_src, size_t block_size,
                                size_t n, void *h_dst, size_t dst_stride, uint32_t *h_out_size)
{
    DevGuard dev_(c);
    if (!c || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!h_src || !h_dst || !h_out_size) return CRYO_E_ARG;
    const size_t bound = cryo_codec_bound(method, block_size);
    if (dst_stride < bound) return CRYO_E_DSTSIZE;
    ScopedLocalCpus numa_(n * block_size >= ((size_t)8 << 20) ? c : nullptr); /* the caller's share of the copies next to the GPU */
    if (pipe_worth_it(c, n, block_size)) {
        const int rc = compress_blocks_piped(c, method, param, (const uint8_t *)h_src, nullptr, block_size, n, (uint8_t *)h_dst, nullptr, dst_stride, h_out_size);
        if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); if (c->xfer) (void)hipStreamSynchronize(c->xfer); }
        pipe_trim(c);
        return rc;
    }
    /* the device slots use the caller's stride, so the output goes back in one copy (a slot may be
     * written beyond out_size[i], up to bound) */
    const bool bulk = dst_stride <= bound + 4096;
    const size_t dstride = bulk ? dst_stride : ((bound + 15) & ~(size_t)15);
    int rc;
    if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * dstride + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, n * 16 + 64)) != CRYO_OK) return rc;
    uint32_t *d_sz = (uint32_t *)c->hb_meta;
    int32_t *d_st = (int32_t *)(c->hb_meta + ((n * 4 + 15) & ~(size_t)15));
    if ((rc = ensure_pinned(c, n * 4)) != CRYO_OK) return rc;
    int32_t *h_st = (int32_t *)c->pin;
    HIP_TRY(c, hipMemcpyAsync(c->hb_src, h_src, n * block_size, hipMemcpyHostToDevice, c->stream));
    rc = cryo_codec_compress_batch(c, method, param, c->hb_src, block_size, (uint32_t)block_size, n, c->hb_dst, dstride, d_sz, d_st);
    if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    HIP_TRY(c, hipMemcpyAsync(h_out_size, d_sz, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_st, d_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (bulk) HIP_TRY(c, hipMemcpyAsync(h_dst, c->hb_dst, (n - 1) * dstride + bound, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        if (h_st[i] != CRYO_OK) return h_st[i];
        if (h_out_size[i] == 0 || h_out_size[i] > bound) return CRYO_E_HIP;
    }
    if (!bulk) {
        for (size_t i = 0; i < n; i++)
            HIP_TRY(c, hipMemcpyAsync((uint8_t *)h_dst + i * dst_stride, c->hb_dst + i * dstride, h_out_size[i], hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CRYO_OK;
}

int cryo_codec_compress_blocks(cryo_codec *c, int method, int param, const void *h_src, size_t block_size,
                               size_t n, void *h_dst, size_t dst_stride, uint32_t *h_out_size)
{
    return guarded([&] {
        const int rc = compress_blocks_body(c, method, param, h_src, block_size, n, h_dst, dst_stride, h_out_size);
        if (c) ws_trim_after_call(c);
        return rc;
    });
}

static int decompress_blocks_impl(cryo_codec *c, int method, const void *const *h_src, const uint32_t *h_src_size,
                                  size_t n, void *h_dst, void *const *h_dst_each, size_t block_size, int32_t *h_status)
{
    DevGuard dev_(c);
    if (!c || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!h_src || !h_src_size || (!h_dst && !h_dst_each) || !h_status) return CRYO_E_ARG;
    ScopedLocalCpus numa_(n * block_size >= ((size_t)8 << 20) ? c : nullptr);
    if (pipe_worth_it(c, n, block_size)) {
        const int rc = decompress_blocks_piped(c, method, h_src, h_src_size, n, (uint8_t *)h_dst, h_dst_each, block_size, h_status);
        /* an error return must not leave copies in flight into the handle's pinned buffers or the caller's memory */
        if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); if (c->xfer) (void)hipStreamSynchronize(c->xfer); }
        pipe_trim(c);
        return rc;
    }
    /* pinned staging: [offsets u64 x n][sizes u32 x n][compressed blocks, 16-byte aligned], sent in one copy */
    const size_t o_off = 0, o_sz = n * 8, o_data = (n * 12 + 63) & ~(size_t)63;
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (h_src_size[i] != 0 && !h_src[i]) return CRYO_E_ARG;
        total += ((size_t)h_src_size[i] + 15) & ~(size_t)15;
    }
    int rc;
    if ((rc = ensure_pinned(c, o_data + total + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, o_data + total + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, n * 16 + 64)) != CRYO_OK) return rc;
    uint8_t *pin = (uint8_t *)c->pin;
    uint64_t *p_off = (uint64_t *)(pin + o_off);
    uint32_t *p_sz = (uint32_t *)(pin + o_sz);
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        p_off[i] = o_data + pos;
        p_sz[i] = h_src_size[i];
        if (h_src_size[i]) memcpy(pin + o_data + pos, h_src[i], h_src_size[i]);
        pos += ((size_t)h_src_size[i] + 15) & ~(size_t)15;
    }
    int32_t *d_st = (int32_t *)c->hb_meta;
    HIP_TRY(c, hipMemcpyAsync(c->hb_src, pin, o_data + total, hipMemcpyHostToDevice, c->stream));
    c->xfer_ctr.h2d_bytes += o_data + total;
    c->xfer_ctr.d2h_bytes += n * block_size + n * sizeof(int32_t);
    rc = cryo_codec_decompress_batch(c, method, c->hb_src, (const uint64_t *)(c->hb_src + o_off), (const uint32_t *)(c->hb_src + o_sz),
                                     c->hb_dst, block_size, (uint32_t)block_size, n, d_st);
    if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    HIP_TRY(c, hipMemcpyAsync(h_status, d_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (h_dst) {
        HIP_TRY(c, hipMemcpyAsync(h_dst, c->hb_dst, n * block_size, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return CRYO_OK;
    }
    /* one destination per block: the blocks come back in ONE copy into the pinned buffer (the compressed side of it
     * is no longer needed) and are handed to their destinations from there; a block that failed leaves its destination
     * untouched */
    if (ensure_pinned(c, n * block_size) == CRYO_OK) {
        HIP_TRY(c, hipMemcpyAsync(c->pin, c->hb_dst, n * block_size, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<CopyJob> jobs;
        for (size_t i = 0; i < n; i++) {
            if (h_status[i] != CRYO_OK) continue;
            if (!h_dst_each[i]) return CRYO_E_ARG;
            jobs.push_back({h_dst_each[i], (const uint8_t *)c->pin + i * block_size, block_size});
        }
        parallel_copy(c, jobs);
        return CRYO_OK;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        if (h_status[i] != CRYO_OK) continue;
        if (!h_dst_each[i]) return CRYO_E_ARG;
        HIP_TRY(c, hipMemcpyAsync(h_dst_each[i], c->hb_dst + i * block_size, block_size, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRYO_OK;
}

int cryo_codec_decompress_blocks(cryo_codec *c, int method, const void *const *h_src, const uint32_t *h_src_size,
                                 size_t n, void *h_dst, size_t block_size, int32_t *h_status)
{
    if (!h_dst) return CRYO_E_ARG;
    return guarded([&] {
        const int rc = decompress_blocks_impl(c, method, h_src, h_src_size, n, h_dst, nullptr, block_size, h_status);
        if (c) ws_trim_after_call(c);
        return rc;
    });
}

int cryo_codec_decompress_blocks_to(cryo_codec *c, int method, const void *const *h_src, const uint32_t *h_src_size,
                                    size_t n, void *const *h_dst, size_t block_size, int32_t *h_status)
{
    if (!h_dst) return CRYO_E_ARG;
    return guarded([&] {
        const int rc = decompress_blocks_impl(c, method, h_src, h_src_size, n, nullptr, h_dst, block_size, h_status);
        if (c) ws_trim_after_call(c);
        return rc;
    });
}

} /* extern "C" */

/* ---- device-resident block pool ---- */
/* Identity of a compressed stream in the pool: a 64-bit hash of ALL its bytes (four multiply-rotate lanes, 32 bytes per
 * step, about 10 GB/s on one host core -- next to nothing beside the transfer and the decode a hit saves).  Round 3
 * sampled 24 bytes (first, middle, last 8): a rewritten block of the same compressed size that happened to share them --
 * the first 8 bytes of a zstd frame are a near-constant header -- would have been served stale. */
static uint64_t stream_fingerprint(const void *p, uint32_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto round1 = [&](uint64_t acc, uint64_t v) { return rotl(acc + v * P2, 31) * P1; };
    uint64_t h;
    uint32_t i = 0;
    if (n >= 32u) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; i + 32u <= n; i += 32u) {
            uint64_t w[4];
            memcpy(w, b + i, 32);
            v1 = round1(v1, w[0]); v2 = round1(v2, w[1]); v3 = round1(v3, w[2]); v4 = round1(v4, w[3]);
        }
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = (h ^ round1(0, v1)) * P1 + P3;
        h = (h ^ round1(0, v2)) * P1 + P3;
        h = (h ^ round1(0, v3)) * P1 + P3;
        h = (h ^ round1(0, v4)) * P1 + P3;
    } else {
        h = P3;
    }
    h += n;
    for (; i + 8u <= n; i += 8u) { uint64_t w; memcpy(&w, b + i, 8); h = rotl(h ^ round1(0, w), 27) * P1 + P3; }
    for (; i < n; i++) h = rotl(h ^ (b[i] * P3), 11) * P1;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

static void pool_drop(cryo_codec *c)
{
    if (c->d_pool) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->d_pool); c->d_pool = nullptr; }
    c->pool_slots.clear();
    c->pool_index.clear();
    c->pool_head = 0;
    c->pool_block = 0;
}

/* (re)shape the pool for this block size; false: no pool (off, or the memory is not there) */
static bool pool_ready(cryo_codec *c, size_t block_size)
{
    if (c->pool_bytes < block_size) { if (c->d_pool) pool_drop(c); return false; }
    const size_t slots = c->pool_bytes / block_size;
    if (c->d_pool && c->pool_block == block_size && c->pool_slots.size() == slots) return true;
    pool_drop(c);
    if (hipMalloc((void **)&c->d_pool, slots * block_size + 64) != hipSuccess) { (void)hipGetLastError(); c->d_pool = nullptr; return false; }
    c->pool_slots.assign(slots, cryo_codec::PoolSlot());
    c->pool_block = block_size;
    return true;
}

static int decompress_blocks_keyed_impl(cryo_codec *c, int method, const uint64_t *keys, const void *const *h_src,
                                        const uint32_t *h_src_size, size_t n, void *const *h_dst, size_t block_size, int32_t *h_status)
{
    DevGuard dev_(c);
    if (!c || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!keys || !h_src || !h_src_size || !h_dst || !h_status) return CRYO_E_ARG;
    if (!pool_ready(c, block_size)) return decompress_blocks_impl(c, method, h_src, h_src_size, n, nullptr, h_dst, block_size, h_status);
    const uint32_t S = (uint32_t)c->pool_slots.size();
    /* who is in the pool already */
    std::vector<uint32_t> slot(n, 0xffffffffu), miss;
    std::vector<uint64_t> fp(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (h_src_size[i] != 0 && !h_src[i]) return CRYO_E_ARG;
        if (!h_dst[i]) return CRYO_E_ARG;
        if (keys[i] != 0 && h_src_size[i] != 0) {
            fp[i] = stream_fingerprint(h_src[i], h_src_size[i]);
            auto it = c->pool_index.find(keys[i]);
            if (it != c->pool_index.end()) {
                const cryo_codec::PoolSlot &ps = c->pool_slots[it->second];
                if (ps.valid && ps.csize == h_src_size[i] && ps.fp == fp[i]) { slot[i] = it->second; h_status[i] = CRYO_OK; continue; }
            }
        }
        miss.push_back((uint32_t)i);
    }
    c->xfer_ctr.pool_hits += n - miss.size();
    c->xfer_ctr.pool_misses += miss.size();
    int rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure_pinned(c, n * block_size + n * 16 + 64)) != CRYO_OK) return rc;
    /* 1. the blocks that are there leave their slots first (a miss below may take a slot over) */
    auto gather = [&](const std::vector<uint32_t> &who) -> int {
        for (size_t a = 0; a < who.size();) {
            /* consecutive blocks of the call, up to 64 per launch, land next to each other in the staging area */
            cryo::GatherSlots gs;
            uint32_t cnt = 0;
            const uint32_t first = who[a];
            while (a < who.size() && cnt < 64u && who[a] == first + cnt) { gs.slot[cnt++] = slot[who[a]]; a++; }
            HIP_TRY(c, cryo::launch_gather_blocks(c->stream, c->d_pool, gs, cnt, c->hb_dst, (uint32_t)block_size, first));
        }
        return CRYO_OK;
    };
    {
        std::vector<uint32_t> hits;
        for (size_t i = 0; i < n; i++) if (slot[i] != 0xffffffffu) hits.push_back((uint32_t)i);
        if ((rc = gather(hits)) != CRYO_OK) return rc;
    }
    /* 2. the others are decoded INTO the pool, at most one pool-full per round, then gathered like the rest */
    std::vector<int32_t> mst(miss.size(), 0);
    for (size_t lo = 0; lo < miss.size(); lo += S) {
        const size_t m = miss.size() - lo < S ? miss.size() - lo : S;
        const size_t o_off = 0, o_sz = m * 8, o_data = (m * 12 + 63) & ~(size_t)63;
        size_t total = 0;
        for (size_t k = 0; k < m; k++) total += ((size_t)h_src_size[miss[lo + k]] + 15) & ~(size_t)15;
        if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, o_data + total + 64)) != CRYO_OK) return rc;
        if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, m * 16 + 64)) != CRYO_OK) return rc;
        /* staging of the compressed side lives behind the output staging in the pinned buffer */
        if (c->pin_cap < n * block_size + o_data + total + m * 4 + 128 && (rc = ensure_pinned(c, n * block_size + o_data + total + m * 4 + 128)) != CRYO_OK) return rc;
        uint8_t *pin = (uint8_t *)c->pin + n * block_size;
        uint64_t *p_off = (uint64_t *)(pin + o_off);
        uint32_t *p_sz = (uint32_t *)(pin + o_sz);
        size_t pos = 0;
        for (size_t k = 0; k < m; k++) {
            const uint32_t i = miss[lo + k];
            p_off[k] = o_data + pos;
            p_sz[k] = h_src_size[i];
            if (h_src_size[i]) memcpy(pin + o_data + pos, h_src[i], h_src_size[i]);
            pos += ((size_t)h_src_size[i] + 15) & ~(size_t)15;
        }
        HIP_TRY(c, hipMemcpyAsync(c->hb_src, pin, o_data + total, hipMemcpyHostToDevice, c->stream));
        c->xfer_ctr.h2d_bytes += o_data + total;
        int32_t *d_st = (int32_t *)c->hb_meta;
        /* ring slots head .. head+m-1 (two launches when the ring wraps) */
        for (size_t done = 0; done < m;) {
            const uint32_t head = c->pool_head;
            const size_t run = (S - head) < (m - done) ? (S - head) : (m - done);
            for (size_t k = 0; k < run; k++) {
                cryo_codec::PoolSlot &ps = c->pool_slots[head + k];
                if (ps.valid) { auto it = c->pool_index.find(ps.key); if (it != c->pool_index.end() && it->second == head + k) c->pool_index.erase(it); }
                ps.valid = false;
                slot[miss[lo + done + k]] = head + (uint32_t)k;
            }
            rc = cryo_codec_decompress_batch(c, method, c->hb_src, (const uint64_t *)(c->hb_src + o_off) + done, (const uint32_t *)(c->hb_src + o_sz) + done,
                                             c->d_pool + (size_t)head * block_size, block_size, (uint32_t)block_size, run, d_st + done);
            if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
            c->pool_head = (uint32_t)((head + run) % S);
            done += run;
        }
        HIP_TRY(c, hipMemcpyAsync(pin + o_data + total, d_st, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        {
            std::vector<uint32_t> who(miss.begin() + lo, miss.begin() + lo + m);
            if ((rc = gather(who)) != CRYO_OK) return rc;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream)); /* the statuses of this round, and the compressed staging is free again */
        const int32_t *pst = (const int32_t *)(pin + o_data + total);
        for (size_t k = 0; k < m; k++) {
            const uint32_t i = miss[lo + k];
            h_status[i] = pst[k];
            if (pst[k] == CRYO_OK && keys[i] != 0) {
                cryo_codec::PoolSlot &ps = c->pool_slots[slot[i]];
                ps.key = keys[i]; ps.fp = fp[i]; ps.csize = h_src_size[i]; ps.valid = true;
                c->pool_index[keys[i]] = slot[i];
            }
        }
        c->xfer_ctr.d2h_bytes += m * sizeof(int32_t);
    }
    /* 3. one copy back, then to the destinations */
    HIP_TRY(c, hipMemcpyAsync(c->pin, c->hb_dst, n * block_size, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->xfer_ctr.d2h_bytes += n * block_size;
    std::vector<CopyJob> jobs;
    for (size_t i = 0; i < n; i++)
        if (h_status[i] == CRYO_OK) jobs.push_back({h_dst[i], (const uint8_t *)c->pin + i * block_size, block_size});
    parallel_copy(c, jobs);
    return CRYO_OK;
}

extern "C" {
int cryo_codec_decompress_blocks_keyed(cryo_codec *c, int method, const uint64_t *keys, const void *const *h_src,
                                       const uint32_t *h_src_size, size_t n, void *const *h_dst, size_t block_size, int32_t *h_status)
{
    return guarded([&] {
        const int rc = decompress_blocks_keyed_impl(c, method, keys, h_src, h_src_size, n, h_dst, block_size, h_status);
        if (c) ws_trim_after_call(c);
        return rc;
    });
}

int cryo_codec_pool_invalidate(cryo_codec *c, uint32_t key_hi, int all_entries)
{
    if (!c) return CRYO_E_ARG;
    return guarded([&] {
        for (size_t k = 0; k < c->pool_slots.size(); k++) {
            cryo_codec::PoolSlot &ps = c->pool_slots[k];
            if (ps.valid && (all_entries || (uint32_t)(ps.key >> 32) == key_hi)) { c->pool_index.erase(ps.key); ps.valid = false; }
        }
        return (int)CRYO_OK;
    });
}

int cryo_codec_get_transfer_counters(const cryo_codec *c, cryo_codec_transfer_counters *out)
{
    if (!c || !out) return CRYO_E_ARG;
    *out = c->xfer_ctr;
    out->pool_blocks = c->pool_index.size();
    out->pool_capacity = c->pool_slots.size();
    return CRYO_OK;
}
} /* extern "C" */

extern "C" {

/* ---- several GPUs behind one call ---- */
struct cryo_multi {
    std::vector<cryo_codec *> h;
    WorkerPool *pool = nullptr; /* one worker per further device */
    char err[320] = {0};
};

/* K blocks given by pointer, results to pointers: what one device's host thread runs for its share */
static int compress_blocks_ptrs(cryo_codec *c, int method, int param, const void *const *h_src, size_t block_size, size_t n,
                                void *const *h_dst, uint32_t *out_size)
{
    DevGuard dev_(c);
    if (n == 0) return CRYO_OK;
    const size_t bound = cryo_codec_bound(method, block_size);
    const size_t dstride = (bound + 15) & ~(size_t)15;
    ScopedLocalCpus numa_(n * block_size >= ((size_t)8 << 20) ? c : nullptr);
    if (pipe_worth_it(c, n, block_size)) { /* the single-handle path's staging: pinned double buffers, second stream, worker threads */
        const int rc = compress_blocks_piped(c, method, param, nullptr, h_src, block_size, n, nullptr, h_dst, dstride, out_size);
        if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); if (c->xfer) (void)hipStreamSynchronize(c->xfer); }
        pipe_trim(c);
        return rc;
    }
    int rc;
    if ((rc = ensure(c, &c->hb_src, &c->hb_src_cap, n * block_size + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_dst, &c->hb_dst_cap, n * dstride + 64)) != CRYO_OK) return rc;
    if ((rc = ensure(c, &c->hb_meta, &c->hb_meta_cap, n * 16 + 64)) != CRYO_OK) return rc;
    if ((rc = ensure_pinned(c, n * dstride + n * 8 + 64)) != CRYO_OK) return rc; /* dstride >= block_size: both directions fit */
    uint8_t *pin = (uint8_t *)c->pin;
    {
        std::vector<CopyJob> jobs;
        for (size_t i = 0; i < n; i++) jobs.push_back({pin + i * block_size, h_src[i], block_size});
        parallel_copy(c, jobs);
    }
    uint32_t *d_sz = (uint32_t *)c->hb_meta;
    int32_t *d_st = (int32_t *)(c->hb_meta + ((n * 4 + 15) & ~(size_t)15));
    HIP_TRY(c, hipMemcpyAsync(c->hb_src, pin, n * block_size, hipMemcpyHostToDevice, c->stream));
    rc = cryo_codec_compress_batch(c, method, param, c->hb_src, block_size, (uint32_t)block_size, n, c->hb_dst, dstride, d_sz, d_st);
    if (rc != CRYO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    int32_t *h_st = (int32_t *)(pin + n * dstride);
    uint32_t *h_sz = (uint32_t *)(pin + n * dstride + n * 4);
    HIP_TRY(c, hipMemcpyAsync(h_sz, d_sz, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_st, d_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        if (h_st[i] != CRYO_OK) return h_st[i];
        if (h_sz[i] == 0 || h_sz[i] > bound) return CRYO_E_HIP;
        out_size[i] = h_sz[i];
    }
    /* the compressed blocks come back in ONE copy into the pinned buffer (its input side is no longer needed) and go to
     * their destinations from there on the worker threads; round 3 copied block by block into pageable memory */
    HIP_TRY(c, hipMemcpyAsync(pin, c->hb_dst, (n - 1) * dstride + out_size[n - 1], hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<CopyJob> jobs;
    for (size_t i = 0; i < n; i++) jobs.push_back({h_dst[i], pin + i * dstride, out_size[i]});
    parallel_copy(c, jobs);
    return CRYO_OK;
}

int cryo_multi_open(const int *devices, int n_devices, cryo_multi **out)
{
    if (!out || !devices || n_devices <= 0) return CRYO_E_ARG;
    *out = nullptr;
    return guarded([&]() -> int {
        cryo_multi *m = new (std::nothrow) cryo_multi;
        if (!m) return CRYO_E_NOMEM;
        for (int i = 0; i < n_devices; i++) {
            cryo_codec *c = nullptr;
            const int rc = cryo_codec_open(devices[i], &c);
            if (rc != CRYO_OK) { cryo_multi_close(m); return rc; }
            m->h.push_back(c);
        }
        if (n_devices > 1) m->pool = new (std::nothrow) WorkerPool((unsigned)n_devices - 1u); /* none: the shares run one after the other */
        *out = m;
        return CRYO_OK;
    });
}

void cryo_multi_close(cryo_multi *m)
{
    if (!m) return;
    delete m->pool;
    for (cryo_codec *c : m->h) cryo_codec_close(c);
    delete m;
}

int cryo_multi_count(const cryo_multi *m) { return m ? (int)m->h.size() : 0; }
const char *cryo_multi_last_error(const cryo_multi *m) { return m ? m->err : ""; }

} /* extern "C" */
/* block i -> handle i mod G; `fn(g, idx)` runs the block indices of handle g, every handle's share on its own host thread */
static int multi_run(cryo_multi *m, size_t n, const std::function<int(size_t, const std::vector<size_t> &)> &fn)
{
    const size_t G = m->h.size();
    std::vector<std::vector<size_t>> share(G);
    for (size_t i = 0; i < n; i++) share[i % G].push_back(i);
    std::vector<int> rc(G, CRYO_OK);
    const std::function<void(unsigned)> one = [&](unsigned g) {
        if (!share[g].empty()) {
            rc[g] = guarded([&] { return fn(g, share[g]); });
            ws_trim_after_call(m->h[g]); /* the keep limit holds per handle, whoever dispatched the call */
        }
    };
    if (m->pool) m->pool->run((unsigned)G, one);
    else for (unsigned g = 0; g < G; g++) one(g);
    for (size_t g = 0; g < G; g++)
        if (rc[g] != CRYO_OK) {
            snprintf(m->err, sizeof m->err, "device handle %zu: %s", g, cryo_codec_last_error(m->h[g]));
            return rc[g];
        }
    return CRYO_OK;
}
extern "C" {

int cryo_multi_compress_blocks(cryo_multi *m, int method, int param, const void *h_src, size_t block_size, size_t n,
                               void *h_dst, size_t dst_stride, uint32_t *h_out_size)
{
    if (!m || m->h.empty() || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!h_src || !h_dst || !h_out_size) return CRYO_E_ARG;
    if (dst_stride < cryo_codec_bound(method, block_size)) return CRYO_E_DSTSIZE;
    if (m->h.size() == 1) return cryo_codec_compress_blocks(m->h[0], method, param, h_src, block_size, n, h_dst, dst_stride, h_out_size);
    return guarded([&] {
        return multi_run(m, n, [&](size_t g, const std::vector<size_t> &idx) {
            std::vector<const void *> src(idx.size());
            std::vector<void *> dst(idx.size());
            std::vector<uint32_t> sz(idx.size());
            for (size_t k = 0; k < idx.size(); k++) {
                src[k] = (const uint8_t *)h_src + idx[k] * block_size;
                dst[k] = (uint8_t *)h_dst + idx[k] * dst_stride;
            }
            const int rc = compress_blocks_ptrs(m->h[g], method, param, src.data(), block_size, idx.size(), dst.data(), sz.data());
            if (rc == CRYO_OK) for (size_t k = 0; k < idx.size(); k++) h_out_size[idx[k]] = sz[k];
            return rc;
        });
    });
}

/* one destination per block (h_dst_each) or one strided area (h_dst) */
static int multi_decompress(cryo_multi *m, int method, const void *const *h_src, const uint32_t *h_src_size, size_t n,
                            void *h_dst, void *const *h_dst_each, size_t block_size, int32_t *h_status)
{
    return guarded([&] {
        return multi_run(m, n, [&](size_t g, const std::vector<size_t> &idx) {
            std::vector<const void *> src(idx.size());
            std::vector<void *> dst(idx.size());
            std::vector<uint32_t> sz(idx.size());
            std::vector<int32_t> st(idx.size());
            for (size_t k = 0; k < idx.size(); k++) {
                src[k] = h_src[idx[k]];
                sz[k] = h_src_size[idx[k]];
                dst[k] = h_dst_each ? h_dst_each[idx[k]] : (void *)((uint8_t *)h_dst + idx[k] * block_size);
            }
            const int rc = decompress_blocks_impl(m->h[g], method, src.data(), sz.data(), idx.size(), nullptr, dst.data(), block_size, st.data());
            if (rc == CRYO_OK) for (size_t k = 0; k < idx.size(); k++) h_status[idx[k]] = st[k];
            return rc;
        });
    });
}

int cryo_multi_decompress_blocks(cryo_multi *m, int method, const void *const *h_src, const uint32_t *h_src_size, size_t n,
                                 void *h_dst, size_t block_size, int32_t *h_status)
{
    if (!m || m->h.empty() || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!h_src || !h_src_size || !h_dst || !h_status) return CRYO_E_ARG;
    if (m->h.size() == 1) return cryo_codec_decompress_blocks(m->h[0], method, h_src, h_src_size, n, h_dst, block_size, h_status);
    return multi_decompress(m, method, h_src, h_src_size, n, h_dst, nullptr, block_size, h_status);
}

int cryo_multi_decompress_blocks_to(cryo_multi *m, int method, const void *const *h_src, const uint32_t *h_src_size, size_t n,
                                    void *const *h_dst, size_t block_size, int32_t *h_status)
{
    if (!m || m->h.empty() || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!h_src || !h_src_size || !h_dst || !h_status) return CRYO_E_ARG;
    if (m->h.size() == 1) return cryo_codec_decompress_blocks_to(m->h[0], method, h_src, h_src_size, n, h_dst, block_size, h_status);
    return multi_decompress(m, method, h_src, h_src_size, n, nullptr, h_dst, block_size, h_status);
}

int cryo_multi_decompress_blocks_keyed(cryo_multi *m, int method, const uint64_t *keys, const void *const *h_src,
                                       const uint32_t *h_src_size, size_t n, void *const *h_dst, size_t block_size, int32_t *h_status)
{
    if (!m || m->h.empty() || !method_ok(method) || block_size == 0 || block_size > 0x7E000000u) return CRYO_E_ARG;
    if (n == 0) return CRYO_OK;
    if (!keys || !h_src || !h_src_size || !h_dst || !h_status) return CRYO_E_ARG;
    if (m->h.size() == 1) return cryo_codec_decompress_blocks_keyed(m->h[0], method, keys, h_src, h_src_size, n, h_dst, block_size, h_status);
    /* a keyed block always goes to the same handle (its pool entry lives there); unkeyed ones round-robin */
    return guarded([&] {
        const size_t G = m->h.size();
        std::vector<std::vector<size_t>> share(G);
        for (size_t i = 0; i < n; i++) share[keys[i] ? keys[i] % G : i % G].push_back(i);
        std::vector<int> rc(G, CRYO_OK);
        const std::function<void(unsigned)> one = [&](unsigned g) {
            const std::vector<size_t> &idx = share[g];
            if (idx.empty()) return;
            rc[g] = guarded([&] {
                std::vector<const void *> src(idx.size());
                std::vector<void *> dst(idx.size());
                std::vector<uint32_t> sz(idx.size());
                std::vector<uint64_t> ky(idx.size());
                std::vector<int32_t> st(idx.size());
                for (size_t k = 0; k < idx.size(); k++) { src[k] = h_src[idx[k]]; sz[k] = h_src_size[idx[k]]; dst[k] = h_dst[idx[k]]; ky[k] = keys[idx[k]]; }
                const int r = cryo_codec_decompress_blocks_keyed(m->h[g], method, ky.data(), src.data(), sz.data(), idx.size(), dst.data(), block_size, st.data());
                if (r == CRYO_OK) for (size_t k = 0; k < idx.size(); k++) h_status[idx[k]] = st[k];
                return r;
            });
        };
        if (m->pool) m->pool->run((unsigned)G, one);
        else for (unsigned g = 0; g < G; g++) one(g);
        for (size_t g = 0; g < G; g++)
            if (rc[g] != CRYO_OK) {
                snprintf(m->err, sizeof m->err, "device handle %zu: %s", g, cryo_codec_last_error(m->h[g]));
                return rc[g];
            }
        return (int)CRYO_OK;
    });
}

int cryo_multi_set_option(cryo_multi *m, int option, int64_t value)
{
    if (!m || m->h.empty()) return CRYO_E_ARG;
    /* a pool capacity is the total over the handles */
    const int64_t v = option == CRYO_OPT_POOL_BYTES ? value / (int64_t)m->h.size() : value;
    for (cryo_codec *c : m->h) {
        const int rc = cryo_codec_set_option(c, option, v);
        if (rc != CRYO_OK) return rc;
    }
    return CRYO_OK;
}

int cryo_multi_trim(cryo_multi *m)
{
    if (!m) return CRYO_E_ARG;
    for (cryo_codec *c : m->h) {
        const int rc = cryo_codec_trim(c);
        if (rc != CRYO_OK) return rc;
    }
    return CRYO_OK;
}

int cryo_multi_pool_invalidate(cryo_multi *m, uint32_t key_hi, int all_entries)
{
    if (!m) return CRYO_E_ARG;
    for (cryo_codec *c : m->h) {
        const int rc = cryo_codec_pool_invalidate(c, key_hi, all_entries);
        if (rc != CRYO_OK) return rc;
    }
    return CRYO_OK;
}

int cryo_multi_get_transfer_counters(const cryo_multi *m, cryo_codec_transfer_counters *out)
{
    if (!m || !out) return CRYO_E_ARG;
    memset(out, 0, sizeof *out);
    for (const cryo_codec *c : m->h) {
        cryo_codec_transfer_counters t;
        if (cryo_codec_get_transfer_counters(c, &t) != CRYO_OK) return CRYO_E_ARG;
        out->h2d_bytes += t.h2d_bytes; out->d2h_bytes += t.d2h_bytes; out->pool_hits += t.pool_hits; out->pool_misses += t.pool_misses;
        out->pool_blocks += t.pool_blocks; out->pool_capacity += t.pool_capacity;
    }
    return CRYO_OK;
}

/* ---- helpers ---- */
int cryo_codec_synth_batch(cryo_codec *c, uint64_t seed, uint64_t first_block, uint64_t block_step,
                           uint64_t n_blocks, uint32_t block_size, int dist, void *d_dst, uint64_t dst_stride)
{
    DevGuard dev_(c);
    if (!c || block_size < 64 || dist < 0 || dist > 4) return CRYO_E_ARG;
    if (n_blocks == 0) return CRYO_OK;
    if (!d_dst || dst_stride < block_size) return CRYO_E_ARG;
    HIP_TRY(c, cryo::launch_synth(c->stream, seed, first_block, block_step ? block_step : 1, n_blocks, block_size, dist,
                                  (uint8_t *)d_dst, dst_stride));
    c->ctr.launches++;
    return CRYO_OK;
}

int cryo_codec_checksum_batch(cryo_codec *c, const void *d_src, uint64_t src_stride,
                              const uint32_t *d_sizes, uint32_t fixed_size, uint64_t n_blocks,
                              uint64_t *d_sums)
{
    DevGuard dev_(c);
    if (!c) return CRYO_E_ARG;
    if (n_blocks == 0) return CRYO_OK;
    if (!d_src || !d_sums) return CRYO_E_ARG;
    HIP_TRY(c, cryo::launch_checksum(c->stream, (const uint8_t *)d_src, src_stride, d_sizes, fixed_size,
                                     n_blocks, d_sums));
    c->ctr.launches++;
    return CRYO_OK;
}

int cryo_codec_compare_batch(cryo_codec *c, const void *d_a, uint64_t a_stride, const void *d_b,
                             uint64_t b_stride, uint32_t block_size, uint64_t n_blocks,
                             uint64_t *d_mismatch)
{
    DevGuard dev_(c);
    if (!c) return CRYO_E_ARG;
    if (n_blocks == 0) return CRYO_OK;
    if (!d_a || !d_b || !d_mismatch) return CRYO_E_ARG;
    HIP_TRY(c, cryo::launch_compare(c->stream, (const uint8_t *)d_a, a_stride, (const uint8_t *)d_b,
                                    b_stride, block_size, n_blocks, d_mismatch));
    c->ctr.launches++;
    return CRYO_OK;
}

/* ---- timing ---- */
int cryo_codec_timer_start(cryo_codec *c)
{
    DevGuard dev_(c);
    if (!c) return CRYO_E_ARG;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return CRYO_OK;
}
int cryo_codec_timer_stop(cryo_codec *c, float *ms)
{
    DevGuard dev_(c);
    if (!c || !ms) return CRYO_E_ARG;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CRYO_OK;
}

int cryo_codec_get_counters(const cryo_codec *c, cryo_codec_counters *out)
{
    if (!c || !out) return CRYO_E_ARG;
    *out = c->ctr;
    return CRYO_OK;
}

} /* extern "C" */
