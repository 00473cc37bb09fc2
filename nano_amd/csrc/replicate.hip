// replicate.hip -- one device-resident copy of a model's parameter bytes per GPU of the node, from ONE host upload ("RCCL
// broadcast(weights) once at load"; the reference has no counterpart -- it is a single-context CPU engine).
//
//   nano_hip_blob_share()    host bytes -> the root device (the only PCIe transfer), then to every other listed device
//                              * over xGMI with RCCL: ncclCommInitAll over the distinct devices + one grouped ncclBroadcast
//                                (librccl.so is opened lazily with dlopen -- a single-GPU user never loads it);
//                              * or hipMemcpyPeerAsync per device (NANO_REPLICATE_VIA=peer, or when RCCL is not available);
//                              * or a host upload per device (NANO_REPLICATE_VIA=host: round 3's behaviour).
//   nano_hip_model_create_ex(..., params_on_device = 1) then builds each replica from its device's copy (host/nano_engine.c
//   nano_context_replicate), and nano_hip_blob_release() frees the copies.
// Errors name the device they happened on.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library itself is dlopen()ed
#include "../../include/nano_mi355x.h"

extern "C" void nano_hip_set_error_(const char *msg);

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { x.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (x.h) break; }
        if (!x.h) return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.h, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(x.h, "ncclBroadcast"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.h, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.h, "ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.h, "ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.Broadcast && x.GroupStart && x.GroupEnd && x.GetErrorString;
        return x;
    }();
    return r;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(int code, const char *fmt, int dev, const char *what) {
    char b[400];
    snprintf(b, sizeof b, fmt, dev, what);
    nano_hip_set_error_(b);
    return code;
}

}  // namespace

struct NanoBlobShare {
    size_t bytes = 0;
    int root = 0;
    std::vector<int> udev;              // distinct devices, root first
    std::vector<void *> uptr;           // their copies
    std::vector<int> target;            // index into udev for every requested device
    double upload_s = 0, share_s = 0;
    std::string how;
};

extern "C" void nano_hip_blob_release(NanoBlobShare *s) {
    if (!s) return;
    for (size_t i = 0; i < s->uptr.size(); i++)
        if (s->uptr[i]) { (void)hipSetDevice(s->udev[i]); (void)hipFree(s->uptr[i]); }
    delete s;
}

// the copy requested device i was built from is no longer needed by it: freed once no later target shares it (a device listed twice --
// two replicas on one GPU -- keeps its copy until the second one is built).  Keeps the transient footprint at ONE extra copy per device.
extern "C" void nano_hip_blob_done(NanoBlobShare *s, int i) {
    if (!s || i < 0 || (size_t)i >= s->target.size()) return;
    const int k = s->target[(size_t)i];
    s->target[(size_t)i] = -1;
    for (int t : s->target) if (t == k) return;
    if (s->uptr[(size_t)k]) { (void)hipSetDevice(s->udev[(size_t)k]); (void)hipFree(s->uptr[(size_t)k]); s->uptr[(size_t)k] = nullptr; }
}

extern "C" const void *nano_hip_blob_ptr(const NanoBlobShare *s, int i) {
    if (!s || i < 0 || (size_t)i >= s->target.size() || s->target[(size_t)i] < 0) return nullptr;
    return s->uptr[(size_t)s->target[(size_t)i]];
}

extern "C" void nano_hip_blob_stats(const NanoBlobShare *s, double *upload_s, double *share_s, char *how, size_t cap) {
    if (!s) return;
    if (upload_s) *upload_s = s->upload_s;
    if (share_s) *share_s = s->share_s;
    if (how && cap) { strncpy(how, s->how.c_str(), cap - 1); how[cap - 1] = 0; }
}

extern "C" int nano_hip_blob_share(NanoBlobShare **out, const void *host, size_t bytes, int root_device, const int *devices, int n) {
    if (!out || !host || !bytes || !devices || n <= 0) { nano_hip_set_error_("nano_hip_blob_share: null / empty argument"); return NANO_HIP_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { nano_hip_set_error_("no HIP device visible (this backend has no CPU fallback)"); return NANO_HIP_ENODEV; }
    if (root_device < 0 || root_device >= ndev) return fail(NANO_HIP_EINVAL, "root device %d out of range%s", root_device, "");
    NanoBlobShare *s = new NanoBlobShare();
    s->bytes = bytes; s->root = root_device;
    s->udev.push_back(root_device);
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) { nano_hip_blob_release(s); return fail(NANO_HIP_EINVAL, "device %d out of range%s", devices[i], ""); }
        size_t k = 0;
        while (k < s->udev.size() && s->udev[k] != devices[i]) k++;
        if (k == s->udev.size()) s->udev.push_back(devices[i]);
        s->target.push_back((int)k);
    }
    s->uptr.assign(s->udev.size(), nullptr);
    const char *via_env = getenv("NANO_REPLICATE_VIA");
    const std::string via = via_env ? via_env : "";
    // every distinct device gets its buffer; NANO_REPLICATE_VIA=rccl also gives the ROOT a second buffer and broadcasts into it, so that
    // the RCCL path runs (a 1-rank communicator) on a single-GPU box too -- what the tests can exercise without a multi-GPU node
    // (NANO_REPLICATE_VIA=peer on one device likewise: host -> a staging buffer, hipMemcpyPeer with source device == destination device)
    const bool force_rccl = via == "rccl", force_peer = via == "peer";
    for (size_t k = 0; k < s->udev.size(); k++) {
        if (hipSetDevice(s->udev[k]) != hipSuccess || hipMalloc(&s->uptr[k], bytes) != hipSuccess) {
            const int d = s->udev[k];
            nano_hip_blob_release(s);
            return fail(NANO_HIP_ENOMEM, "device %d: hipMalloc of the parameter copy failed%s", d, "");
        }
    }
    // ---- the one PCIe transfer -------------------------------------------------------------------------------------------------
    double t0 = now_s();
    void *stage = nullptr;              // (forced RCCL on one device: host -> stage, RCCL stage -> the root's copy)
    (void)hipSetDevice(root_device);
    if ((force_rccl || force_peer) && s->udev.size() == 1) {
        if (hipMalloc(&stage, bytes) != hipSuccess) { nano_hip_blob_release(s); return fail(NANO_HIP_ENOMEM, "device %d: hipMalloc of the staging copy failed%s", root_device, ""); }
    }
    hipError_t e = hipMemcpy(stage ? stage : s->uptr[0], host, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (stage) (void)hipFree(stage); nano_hip_blob_release(s); return fail(NANO_HIP_ERUNTIME, "device %d: upload of the parameters failed: %s", root_device, hipGetErrorString(e)); }
    s->upload_s = now_s() - t0;
    t0 = now_s();
    const size_t nu = s->udev.size();
    if (nu == 1 && !stage) { s->how = "single device"; *out = s; return NANO_HIP_OK; }

    // ---- to the other devices ----------------------------------------------------------------------------------------------------
    bool done = false;
    if (via != "peer" && via != "host" && rccl().ok) {
        Rccl &R = rccl();
        std::vector<ncclComm_t> comms(nu);
        std::vector<hipStream_t> st(nu, nullptr);
        ncclResult_t r = R.CommInitAll(comms.data(), (int)nu, s->udev.data());
        if (r == ncclSuccess) {
            bool ok = true;
            for (size_t k = 0; k < nu && ok; k++) ok = hipSetDevice(s->udev[k]) == hipSuccess && hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) == hipSuccess;
            int bad = -1; const char *why = "";
            if (ok) {
                R.GroupStart();
                for (size_t k = 0; k < nu; k++) {       // rank k = device udev[k]; root = rank 0
                    const void *send = k == 0 ? (stage ? stage : s->uptr[0]) : s->uptr[k];
                    r = R.Broadcast(send, s->uptr[k], bytes, ncclUint8, 0, comms[k], st[k]);
                    if (r != ncclSuccess && bad < 0) { bad = s->udev[k]; why = R.GetErrorString(r); }
                }
                r = R.GroupEnd();
                if (r != ncclSuccess && bad < 0) { bad = root_device; why = R.GetErrorString(r); }
                for (size_t k = 0; k < nu; k++) {
                    (void)hipSetDevice(s->udev[k]);
                    const hipError_t se = hipStreamSynchronize(st[k]);
                    if (se != hipSuccess && bad < 0) { bad = s->udev[k]; why = hipGetErrorString(se); }
                }
            } else { bad = root_device; why = "stream creation failed"; }
            for (size_t k = 0; k < nu; k++) { if (st[k]) { (void)hipSetDevice(s->udev[k]); (void)hipStreamDestroy(st[k]); } R.CommDestroy(comms[k]); }
            if (bad >= 0) {
                if (stage) (void)hipFree(stage);
                nano_hip_blob_release(s);
                return fail(NANO_HIP_ERUNTIME, "device %d: RCCL broadcast of the parameters failed: %s", bad, why);
            }
            done = true;
            char b[64]; snprintf(b, sizeof b, "rccl broadcast over %zu device(s)", nu);
            s->how = b;
        } else if (force_rccl) {
            if (stage) (void)hipFree(stage);
            nano_hip_blob_release(s);
            return fail(NANO_HIP_ERUNTIME, "device %d: ncclCommInitAll failed: %s", root_device, R.GetErrorString(r));
        }
    } else if (force_rccl) {
        if (stage) (void)hipFree(stage);
        nano_hip_blob_release(s);
        return fail(NANO_HIP_ERUNTIME, "device %d: NANO_REPLICATE_VIA=rccl but librccl.so could not be loaded%s", root_device, "");
    }
    if (!done && via != "host") {       // xGMI peer copies, one per device
        bool ok = true; int bad = -1; const char *why = "";
        if (stage) {                    // one device, forced: the staging copy -> the device's own copy through the peer-copy call
            (void)hipSetDevice(root_device);
            const hipError_t pe = hipMemcpyPeer(s->uptr[0], root_device, stage, root_device, bytes);
            if (pe != hipSuccess) { ok = false; bad = root_device; why = hipGetErrorString(pe); }
        }
        for (size_t k = 1; k < nu && ok; k++) {
            (void)hipSetDevice(s->udev[k]);
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, s->udev[k], root_device);
            if (can) (void)hipDeviceEnablePeerAccess(root_device, 0);          // (already enabled: an error we do not care about)
            (void)hipGetLastError();
            const hipError_t pe = hipMemcpyPeer(s->uptr[k], s->udev[k], s->uptr[0], root_device, bytes);
            if (pe != hipSuccess) { ok = false; bad = s->udev[k]; why = hipGetErrorString(pe); }
        }
        if (ok) { done = true; s->how = "hipMemcpyPeer per device"; }
        else if (via == "peer") { if (stage) (void)hipFree(stage); nano_hip_blob_release(s); return fail(NANO_HIP_ERUNTIME, "device %d: peer copy of the parameters failed: %s", bad, why); }
    }
    if (stage) { (void)hipSetDevice(root_device); (void)hipFree(stage); stage = nullptr; }
    if (!done) {                        // host upload per device (round 3's way)
        for (size_t k = 1; k < nu; k++) {
            (void)hipSetDevice(s->udev[k]);
            e = hipMemcpy(s->uptr[k], host, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { const int d = s->udev[k]; nano_hip_blob_release(s); return fail(NANO_HIP_ERUNTIME, "device %d: upload of the parameters failed: %s", d, hipGetErrorString(e)); }
        }
        s->how = "host upload per device";
    }
    s->share_s = now_s() - t0;
    (void)hipSetDevice(root_device);
    *out = s;
    return NANO_HIP_OK;
}
