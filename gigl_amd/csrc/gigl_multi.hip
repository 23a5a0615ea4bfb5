// gigl_multi.hip — the multi-device entry points of the C ABI (BASELINE.json
// north_star: "the BVH is replicated and rays are split across the 8 GPUs of one MI355X node, with
// hit buffers gathered ... over xGMI").
//
// One process, a list of devices.  The scene's flattened tree and triangle records are replicated
// device to device (hipMemcpyPeer: over xGMI between two GPUs), rays are split into contiguous ranges
// (sizes differ by at most one ray, or one 64-ray word for any-hit batches), every device traces its
// range with the ordinary single-device launch on a stream of its own, and the results (16-byte hit
// records, or occlusion words) come back to the first device of the list (or to the host) by direct
// peer copies: each peer uses its own xGMI link, there is no ring and no reduction (rays are
// independent).  RCCL is not linked: a gather to one root is exactly one point-to-point copy per peer,
// which hipMemcpyPeerAsync already is, and the host process may be a PyTorch process that carries its
// own RCCL under the same soname.
//
// Device-pointer batches are stream-ordered end to end and need no host thread: the call records an
// event on the caller's stream, every replica's stream waits for it, scatters its range, traces,
// gathers and records an event the caller's stream then waits for.  The *_async forms return as soon
// as that is enqueued; the plain forms also synchronise the caller's stream.  A handle keeps two sets
// of replica streams and staging buffers and alternates between them, so two batches issued from two
// caller streams overlap (the drain of one traversal under the start of the next).
// Host-pointer batches block per range (pipelined copies inside gigl_trace_closest / gigl_trace_any),
// so each replica has one persistent worker thread that runs its ranges.
//
// The reference has no counterpart (it is a single-process CPU program); the entry points extend
// Scene::trace over a batch (Source/Common/Scene.cpp:234-243) to several devices.
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

#include "gigl_internal.h"

using namespace gd;

namespace {

// One persistent host thread per replica, for the blocking host-pointer ranges.
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(std::move(job));
        }
        cv_.notify_one();
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;   // stop requested and nothing left to run
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> jobs_;
    bool stop_ = false;
    std::thread th_;   // last member: the thread starts after the others exist
};

struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    int left;
    explicit Latch(int n) : left(n) {}
    void done() {
        std::lock_guard<std::mutex> lk(mu);
        if (--left == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return left == 0; });
    }
};

constexpr int kSlots = 2;

}  // namespace

struct gigl_multi {
    struct Slot {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        // staging on this replica's device for ranges whose rays / results live on devices[0]
        void* d_o = nullptr;
        void* d_d = nullptr;
        void* d_out = nullptr;
        size_t cap_rays = 0, cap_out = 0;   // rays d_o / d_d hold; bytes d_out holds
    };
    struct Replica {
        gigl_scene* scene = nullptr;   // replicas[0] borrows the caller's scene
        bool owned = false;
        Slot slot[kSlots];
        std::unique_ptr<Worker> worker;   // started by the first host-pointer batch
    };
    std::vector<Replica> reps;
    hipEvent_t start[kSlots] = {nullptr, nullptr};   // on devices[0]: what the caller's stream held at the call
    std::mutex mu;                                   // enqueueing one batch (slot choice, staging growth) at a time
    unsigned next_slot = 0;
};

namespace {

void range_of(uint64_t n, int parts, int k, uint64_t& lo, uint64_t& hi) {
    const uint64_t base = n / (uint64_t)parts, rem = n % (uint64_t)parts;
    lo = (uint64_t)k * base + std::min<uint64_t>((uint64_t)k, rem);
    hi = lo + base + ((uint64_t)k < rem ? 1 : 0);
}

// Ray range of replica k: closest-hit batches split by rays, any-hit batches by 64-ray words (every
// range then owns whole words of the output).
void ray_range(uint64_t n, int parts, int k, bool any, uint64_t& lo, uint64_t& hi) {
    if (!any) return range_of(n, parts, k, lo, hi);
    range_of((n + 63) / 64, parts, k, lo, hi);
    lo = std::min<uint64_t>(lo * 64, n);
    hi = std::min<uint64_t>(hi * 64, n);
}

size_t out_bytes(uint64_t m, bool any) { return any ? (size_t)((m + 63) / 64 * 8) : (size_t)m * sizeof(gigl_hit); }
size_t out_offset(uint64_t lo, bool any) { return any ? (size_t)(lo / 64 * 8) : (size_t)lo * sizeof(gigl_hit); }

int check_args(const gigl_multi* m, const void* o, const void* d, uint64_t n, const void* out) {
    if (!m) return fail(GIGL_ERR_INVALID, "multi handle is NULL");
    if (n == 0) return GIGL_OK;
    if (!o || !d || !out) return fail(GIGL_ERR_INVALID, "NULL ray or output pointer");
    if (n > (uint64_t)kMaxRays * m->reps.size()) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 rays per device");
    return GIGL_OK;
}

// Device pointers on devices[0]: everything is enqueued, nothing is waited for.  On return the caller's
// stream is ordered behind every replica's last operation.
int enqueue_device(gigl_multi* m, const void* o, const void* d, uint64_t n, void* out, bool any, hipStream_t caller) {
    int rc = check_args(m, o, d, n, out);
    if (rc != GIGL_OK || n == 0) return rc;
    const int k_n = (int)m->reps.size();
    const int root_dev = m->reps[0].scene->device;
    std::lock_guard<std::mutex> lk(m->mu);
    const int q = (int)(m->next_slot++ % kSlots);
    std::string err;
    auto hip_fail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return GIGL_ERR_HIP;
    };
    {
        DeviceGuard g(root_dev);
        const hipError_t e = hipEventRecord(m->start[q], caller);
        if (e != hipSuccess) return fail(GIGL_ERR_HIP, std::string("hipEventRecord(start): ") + hipGetErrorString(e));
    }
    int enqueued = 0;
    for (int k = 0; k < k_n && rc == GIGL_OK; ++k) {
        uint64_t lo, hi;
        ray_range(n, k_n, k, any, lo, hi);
        const uint64_t cnt = hi - lo;
        if (cnt == 0) continue;
        gigl_multi::Replica& R = m->reps[(size_t)k];
        gigl_multi::Slot& S = R.slot[q];
        gigl_scene* s = R.scene;
        const size_t ob = out_bytes(cnt, any);
        const char *src_o = (const char*)o + lo * 16, *src_d = (const char*)d + lo * 16;
        char* dst = (char*)out + out_offset(lo, any);
        // the root replica traces its range where it lies; every other replica stages its range (also one
        // that shares the root's device: a device list may repeat a device, which is how the scatter /
        // gather path is tested on a one-GPU box)
        const bool in_place = k == 0;
        {
            DeviceGuard g(s->device);
            hipError_t e = hipStreamWaitEvent(S.stream, m->start[q], 0);
            if (e != hipSuccess) { rc = hip_fail("hipStreamWaitEvent(start)", e); break; }
            const void *to = src_o, *td = src_d;
            void* th = dst;
            if (!in_place) {
                if (S.cap_rays < cnt || S.cap_out < ob) {
                    // growing: the slot's earlier batches still use the old buffers
                    (void)hipStreamSynchronize(S.stream);
                    if (S.d_o) (void)hipFree(S.d_o);
                    if (S.d_d) (void)hipFree(S.d_d);
                    if (S.d_out) (void)hipFree(S.d_out);
                    S.d_o = S.d_d = S.d_out = nullptr;
                    S.cap_rays = S.cap_out = 0;
                    const size_t want_out = std::max(ob, (size_t)cnt * sizeof(gigl_hit));   // serves either kind of batch
                    e = hipMalloc(&S.d_o, cnt * 16);
                    if (e == hipSuccess) e = hipMalloc(&S.d_d, cnt * 16);
                    if (e == hipSuccess) e = hipMalloc(&S.d_out, want_out);
                    if (e != hipSuccess) { rc = hip_fail("hipMalloc(staging)", e); break; }
                    S.cap_rays = cnt;
                    S.cap_out = want_out;
                }
                e = hipMemcpyPeerAsync(S.d_o, s->device, src_o, root_dev, cnt * 16, S.stream);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(S.d_d, s->device, src_d, root_dev, cnt * 16, S.stream);
                if (e != hipSuccess) { rc = hip_fail("ray scatter", e); break; }
                to = S.d_o; td = S.d_d; th = S.d_out;
            }
            rc = any ? gigl_trace_any_device(s, to, td, cnt, (uint64_t*)th, S.stream)
                     : gigl_trace_closest_device(s, to, td, cnt, (gigl_hit*)th, S.stream);
            if (rc != GIGL_OK) { err = gigl_last_error(); break; }
            if (!in_place) {
                e = hipMemcpyPeerAsync(dst, root_dev, S.d_out, s->device, ob, S.stream);
                if (e != hipSuccess) { rc = hip_fail("result gather", e); break; }
            }
            e = hipEventRecord(S.done, S.stream);
            if (e != hipSuccess) { rc = hip_fail("hipEventRecord(done)", e); break; }
        }
        DeviceGuard g(root_dev);
        const hipError_t e = hipStreamWaitEvent(caller, S.done, 0);
        if (e != hipSuccess) { rc = hip_fail("hipStreamWaitEvent(done)", e); break; }
        ++enqueued;
    }
    if (rc != GIGL_OK) {
        // whatever was enqueued still reads the caller's rays and writes the caller's output: let it finish
        for (auto& R : m->reps) {
            DeviceGuard g(R.scene->device);
            (void)hipStreamSynchronize(R.slot[q].stream);
        }
        return fail(rc, err);
    }
    (void)enqueued;
    return GIGL_OK;
}

int run_device(gigl_multi* m, const void* o, const void* d, uint64_t n, void* out, bool any, void* stream, bool wait) {
    const int rc = enqueue_device(m, o, d, n, out, any, (hipStream_t)stream);
    if (rc != GIGL_OK || !wait || n == 0) return rc;
    DeviceGuard g(m->reps[0].scene->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GIGL_OK;
}

// Host pointers: every device fetches and returns its own range through the pipelined host-pointer call
// (gigl_rt.hip, trace_host): uploads, traversal and downloads of a range overlap, and so do the devices.
int run_host(gigl_multi* m, const float* o, const float* d, uint64_t n, void* out, bool any) {
    int rc = check_args(m, o, d, n, out);
    if (rc != GIGL_OK || n == 0) return rc;
    const int k_n = (int)m->reps.size();
    {
        std::lock_guard<std::mutex> lk(m->mu);
        for (auto& R : m->reps)
            if (!R.worker) R.worker.reset(new Worker());
    }
    std::vector<int> rcs((size_t)k_n, GIGL_OK);
    std::vector<std::string> errs((size_t)k_n);
    Latch latch(k_n);
    for (int k = 0; k < k_n; ++k) {
        uint64_t lo, hi;
        ray_range(n, k_n, k, any, lo, hi);
        gigl_scene* s = m->reps[(size_t)k].scene;
        m->reps[(size_t)k].worker->post([&, k, lo, hi, s]() {
            if (hi > lo) {
                char* dst = (char*)out + out_offset(lo, any);
                rcs[(size_t)k] = any ? gigl_trace_any(s, o + lo * 4, d + lo * 4, hi - lo, (uint64_t*)dst)
                                     : gigl_trace_closest(s, o + lo * 4, d + lo * 4, hi - lo, (gigl_hit*)dst);
                if (rcs[(size_t)k] != GIGL_OK) errs[(size_t)k] = gigl_last_error();   // thread-local: read on the worker
            }
            latch.done();
        });
    }
    latch.wait();
    for (int k = 0; k < k_n; ++k)
        if (rcs[(size_t)k] != GIGL_OK)
            return fail(rcs[(size_t)k], "device " + std::to_string(m->reps[(size_t)k].scene->device) + ": " + errs[(size_t)k]);
    return GIGL_OK;
}

}  // namespace

extern "C" {

int gigl_multi_create(const gigl_scene* s, const int* devices, int n_devices, gigl_multi** out) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!s || !devices || n_devices < 1 || n_devices > 64) return fail(GIGL_ERR_INVALID, "bad device list");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (devices[0] != s->device) return fail(GIGL_ERR_INVALID, "devices[0] must be the device the scene was created on");
    gigl_multi* m = new gigl_multi();
    m->reps.resize((size_t)n_devices);
    int rc = GIGL_OK;
    {
        DeviceGuard g(s->device);
        for (int q = 0; q < kSlots && rc == GIGL_OK; ++q)
            if (hipEventCreateWithFlags(&m->start[q], hipEventDisableTiming) != hipSuccess)
                rc = fail(GIGL_ERR_HIP, "hipEventCreateWithFlags");
    }
    for (int k = 0; k < n_devices && rc == GIGL_OK; ++k) {
        gigl_multi::Replica& R = m->reps[(size_t)k];
        if (k == 0) R.scene = const_cast<gigl_scene*>(s);
        else {
            rc = scene_clone_to_device(s, devices[k], &R.scene);
            R.owned = rc == GIGL_OK;
        }
        if (rc != GIGL_OK) break;
        DeviceGuard g(R.scene->device);
        // direct peer access between the root and this device makes the copies one xGMI hop
        if (R.scene->device != s->device) {
            (void)hipDeviceEnablePeerAccess(s->device, 0);
            (void)hipGetLastError();   // "already enabled" is not an error here
        }
        for (int q = 0; q < kSlots && rc == GIGL_OK; ++q) {
            if (hipStreamCreateWithFlags(&R.slot[q].stream, hipStreamNonBlocking) != hipSuccess)
                rc = fail(GIGL_ERR_HIP, "hipStreamCreateWithFlags");
            else if (hipEventCreateWithFlags(&R.slot[q].done, hipEventDisableTiming) != hipSuccess)
                rc = fail(GIGL_ERR_HIP, "hipEventCreateWithFlags");
        }
    }
    if (rc != GIGL_OK) {
        std::string keep = gigl_last_error();
        gigl_multi_destroy(m);
        return fail(rc, keep);
    }
    *out = m;
    return GIGL_OK;
}

int gigl_multi_destroy(gigl_multi* m) {
    if (!m) return GIGL_OK;
    for (auto& R : m->reps) {
        R.worker.reset();   // joins after the ranges it still holds
        if (!R.scene) continue;
        DeviceGuard g(R.scene->device);
        for (auto& S : R.slot) {
            if (S.stream) { (void)hipStreamSynchronize(S.stream); (void)hipStreamDestroy(S.stream); }
            if (S.done) (void)hipEventDestroy(S.done);
            if (S.d_o) (void)hipFree(S.d_o);
            if (S.d_d) (void)hipFree(S.d_d);
            if (S.d_out) (void)hipFree(S.d_out);
        }
        if (R.owned) (void)gigl_scene_destroy(R.scene);
    }
    if (!m->reps.empty() && m->reps[0].scene) {
        DeviceGuard g(m->reps[0].scene->device);
        for (hipEvent_t e : m->start)
            if (e) (void)hipEventDestroy(e);
    }
    delete m;
    return GIGL_OK;
}

int gigl_multi_device_count(const gigl_multi* m, int* count) {
    if (!m || !count) return fail(GIGL_ERR_INVALID, "NULL argument");
    *count = (int)m->reps.size();
    return GIGL_OK;
}

int gigl_multi_trace_closest_device(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n, gigl_hit* hits,
                                    void* stream) {
    return run_device(m, o_tmin, d_tmax, n, hits, false, stream, true);
}

int gigl_multi_trace_closest_device_async(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                          gigl_hit* hits, void* stream) {
    return run_device(m, o_tmin, d_tmax, n, hits, false, stream, false);
}

int gigl_multi_trace_any_device(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                uint64_t* occluded_bits, void* stream) {
    return run_device(m, o_tmin, d_tmax, n, occluded_bits, true, stream, true);
}

int gigl_multi_trace_any_device_async(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                      uint64_t* occluded_bits, void* stream) {
    return run_device(m, o_tmin, d_tmax, n, occluded_bits, true, stream, false);
}

int gigl_multi_trace_closest(gigl_multi* m, const float* o_tmin, const float* d_tmax, uint64_t n, gigl_hit* hits) {
    return run_host(m, o_tmin, d_tmax, n, hits, false);
}

int gigl_multi_trace_any(gigl_multi* m, const float* o_tmin, const float* d_tmax, uint64_t n, uint64_t* occluded_bits) {
    return run_host(m, o_tmin, d_tmax, n, occluded_bits, true);
}

}  // extern "C"
