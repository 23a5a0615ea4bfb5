// gigl_multi.hip — the multi-device entry points of the C ABI (BASELINE.json
// north_star: "the BVH is replicated and rays are split across the 8 GPUs of one MI355X node, with
// hit buffers gathered ... over xGMI").
//
// One process, a list of devices.  The scene's flattened tree and triangle records are replicated
// device to device (hipMemcpyPeer: over xGMI between two GPUs), rays are split into contiguous ranges
// (sizes differ by at most one), every device traces its range with the ordinary single-device launch
// on a stream of its own, driven by a host thread of its own, and the 16-byte hit records come back to
// the first device of the list (or to the host) by direct peer copies: each peer uses its own xGMI
// link, there is no ring and no reduction (rays are independent).  RCCL is not linked: a gather to
// one root is exactly one point-to-point copy per peer, which hipMemcpyPeerAsync already is, and the
// host process may be a PyTorch process that carries its own RCCL under the same soname.
//
// The reference has no counterpart (it is a single-process CPU program); the entry points extend
// Scene::trace over a batch (Source/Common/Scene.cpp:234-243) to several devices.
#include <thread>

#include "gigl_internal.h"

using namespace gd;

struct gigl_multi {
    struct Replica {
        gigl_scene* scene = nullptr;   // replicas[0] borrows the caller's scene
        bool owned = false;
        hipStream_t stream = nullptr;
        // staging on this replica's device for ranges whose rays / hits live elsewhere
        float4* d_o = nullptr;
        float4* d_d = nullptr;
        gigl_hit* d_hits = nullptr;
        size_t cap = 0;                // rays the staging buffers hold
    };
    std::vector<Replica> reps;
    std::mutex mu;                     // one batch at a time per handle (the staging buffers are per handle)
};

namespace {

void range_of(uint64_t n, int parts, int k, uint64_t& lo, uint64_t& hi) {
    const uint64_t base = n / (uint64_t)parts, rem = n % (uint64_t)parts;
    lo = (uint64_t)k * base + std::min<uint64_t>((uint64_t)k, rem);
    hi = lo + base + ((uint64_t)k < rem ? 1 : 0);
}

// Runs on the replica's own host thread.  src_dev >= 0: rays and hits are device pointers on
// src_dev; src_dev < 0: host pointers (handed to the replica's own host-pointer call).
int run_range(gigl_multi::Replica& R, bool root, const char* o, const char* d, char* hits, uint64_t lo, uint64_t hi,
              int src_dev, std::string& err) {
    const uint64_t m = hi - lo;
    if (m == 0) return GIGL_OK;
    gigl_scene* s = R.scene;
    if (src_dev < 0) {
        // host pointers: every device fetches and returns its own range through the pipelined host-pointer
        // call (gigl_rt.hip, trace_host): uploads, traversal and downloads of a range overlap, and so do the devices
        const int rc = gigl_trace_closest(s, (const float*)(o + lo * 16), (const float*)(d + lo * 16), m, (gigl_hit*)(hits + lo * 16));
        if (rc != GIGL_OK) err = gigl_last_error();
        return rc;
    }
    DeviceGuard g(s->device);
    auto hip_fail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return GIGL_ERR_HIP;
    };
    // the root replica traces device-resident ranges where they lie; every other replica stages its range
    // (also one that shares the root's device: a device list may repeat a device, which is how the
    // scatter / gather path is tested on a one-GPU box)
    const bool in_place = root && src_dev == s->device;
    if (!in_place && R.cap < m) {
        if (R.d_o) (void)hipFree(R.d_o);
        if (R.d_d) (void)hipFree(R.d_d);
        if (R.d_hits) (void)hipFree(R.d_hits);
        R.d_o = R.d_d = nullptr; R.d_hits = nullptr; R.cap = 0;
        hipError_t e = hipMalloc((void**)&R.d_o, m * 16);
        if (e == hipSuccess) e = hipMalloc((void**)&R.d_d, m * 16);
        if (e == hipSuccess) e = hipMalloc((void**)&R.d_hits, m * 16);
        if (e != hipSuccess) return hip_fail("hipMalloc(staging)", e);
        R.cap = m;
    }
    const void *to = o + lo * 16, *td = d + lo * 16;
    void* th = hits + lo * 16;
    hipError_t e = hipSuccess;
    if (!in_place) {
        e = hipMemcpyPeerAsync(R.d_o, s->device, to, src_dev, m * 16, R.stream);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(R.d_d, s->device, td, src_dev, m * 16, R.stream);
        if (e != hipSuccess) return hip_fail("ray scatter", e);
        to = R.d_o; td = R.d_d; th = R.d_hits;
    }
    int rc = gigl_trace_closest_device(s, to, td, m, (gigl_hit*)th, R.stream);
    if (rc != GIGL_OK) { err = gigl_last_error(); return rc; }
    if (!in_place) {
        e = hipMemcpyPeerAsync(hits + lo * 16, src_dev, R.d_hits, s->device, m * 16, R.stream);
        if (e != hipSuccess) return hip_fail("hit gather", e);
    }
    e = hipStreamSynchronize(R.stream);
    if (e != hipSuccess) return hip_fail("hipStreamSynchronize", e);
    return GIGL_OK;
}

int run_all(gigl_multi* m, const void* o, const void* d, uint64_t n, gigl_hit* hits, int src_dev) {
    if (!m) return fail(GIGL_ERR_INVALID, "multi handle is NULL");
    if (n == 0) return GIGL_OK;
    if (!o || !d || !hits) return fail(GIGL_ERR_INVALID, "NULL ray or output pointer");
    if (n > (uint64_t)kMaxRays * m->reps.size()) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 rays per device");
    std::lock_guard<std::mutex> lk(m->mu);
    const int k_n = (int)m->reps.size();
    std::vector<int> rcs(k_n, GIGL_OK);
    std::vector<std::string> errs(k_n);
    std::vector<std::thread> th;
    for (int k = 0; k < k_n; ++k) {
        uint64_t lo, hi;
        range_of(n, k_n, k, lo, hi);
        th.emplace_back([&, k, lo, hi]() {
            rcs[k] = run_range(m->reps[k], k == 0, (const char*)o, (const char*)d, (char*)hits, lo, hi, src_dev, errs[k]);
        });
    }
    for (auto& t : th) t.join();
    for (int k = 0; k < k_n; ++k)
        if (rcs[k] != GIGL_OK) return fail(rcs[k], "device " + std::to_string(m->reps[k].scene->device) + ": " + errs[k]);
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
        if (hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking) != hipSuccess)
            rc = fail(GIGL_ERR_HIP, "hipStreamCreateWithFlags");
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
        if (!R.scene) continue;
        DeviceGuard g(R.scene->device);
        if (R.stream) { (void)hipStreamSynchronize(R.stream); (void)hipStreamDestroy(R.stream); }
        if (R.d_o) (void)hipFree(R.d_o);
        if (R.d_d) (void)hipFree(R.d_d);
        if (R.d_hits) (void)hipFree(R.d_hits);
        if (R.owned) (void)gigl_scene_destroy(R.scene);
    }
    delete m;
    return GIGL_OK;
}

int gigl_multi_device_count(const gigl_multi* m, int* count) {
    if (!m || !count) return fail(GIGL_ERR_INVALID, "NULL argument");
    *count = (int)m->reps.size();
    return GIGL_OK;
}

int gigl_multi_trace_closest_device(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n, gigl_hit* hits) {
    if (!m) return fail(GIGL_ERR_INVALID, "multi handle is NULL");
    return run_all(m, o_tmin, d_tmax, n, hits, m->reps[0].scene->device);
}

int gigl_multi_trace_closest(gigl_multi* m, const float* o_tmin, const float* d_tmax, uint64_t n, gigl_hit* hits) {
    return run_all(m, o_tmin, d_tmax, n, hits, -1);
}

}  // extern "C"
