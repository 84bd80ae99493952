// rtd_plan.hip — the reference-shaped call on several GPUs of one process (include/rtd.h, rtd_plan_*).
//
// The reference's beam loop (src/kernel_wrapper.cu:601) is sequential on one GPU; its beams share nothing but the read-only
// CT / LUTs and the final `+=` into the dose volume (:92). Here one host thread drives one device:
//   1. every device holds ITS z-slab of the dose volume, of which only the block the plan can change travels: the bounding box of
//      the beams' dose boxes inside the slab is uploaded once the boxes are known (after step 2) and downloaded at the end — the
//      voxels outside it are neither read nor written by any transfer (kernel_wrapper.cu:542 / :1318 move the whole volume both
//      ways: 2 x 537 MB for a 512^3 grid, 19 of the 30 ms of a one-field call here before this change);
//   2. beams are dealt round-robin; a device computes the beam's-eye-view (BEV) dose of its beams and packs each into a
//      message [state record | non-zero block of the BEV cube] (~10 MB for a 512^3 field);
//   3. every device pulls the messages of the other devices with peer copies over xGMI — the only inter-GPU traffic;
//   4. every device runs the fan -> dose transfer (primTransfDiv, :69-97) of ALL beams, in beam order, restricted to its
//      slab: each voxel receives the same `+=` sequence as in the sequential loop, so the result is bit-identical to rtd_compute;
//   5. every device downloads that block of its slab into the caller's buffer.
// Built only on the public C ABI of the engine plus the HIP runtime (threads, peer copies).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is opened on demand (no link-time dependency)

#include "../../include/rtd.h"

namespace {

// RCCL as the transport of the slab exchange (RTD_PLAN_TRANSPORT=rccl): one communicator per device of this process
// (ncclCommInitAll), one ncclBroadcast per beam from the device that computed it — every device thread makes the call on its own
// communicator and stream. librccl is dlopen'ed when asked for, so the default transport (peer copies) and every other entry point
// of the engine carry no dependency on it (a process may already hold another copy, e.g. PyTorch's).
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("RTD_PLAN_TRANSPORT=rccl: cannot open librccl: ") + dlerror(); return false; }
        commInitAll = reinterpret_cast<decltype(commInitAll)>(dlsym(lib, "ncclCommInitAll"));
        commDestroy = reinterpret_cast<decltype(commDestroy)>(dlsym(lib, "ncclCommDestroy"));
        broadcast = reinterpret_cast<decltype(broadcast)>(dlsym(lib, "ncclBroadcast"));
        errorString = reinterpret_cast<decltype(errorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!commInitAll || !commDestroy || !broadcast || !errorString) { err = "RTD_PLAN_TRANSPORT=rccl: librccl lacks an entry point"; return false; }
        return true;
    }
};

// reusable barrier of the device threads (std::barrier needs C++20)
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }
private:
    std::mutex m_; std::condition_variable cv_; int n_, count_ = 0, gen_ = 0;
};

struct DevSlot {
    int device = 0;
    rtd_handle h = nullptr;
    hipStream_t copyStream = nullptr;     // pulls of the other devices' messages, upload of the dose block
    hipEvent_t uploaded = nullptr;        // the dose block is on the device (the transfers wait for it)
    float* dSlab = nullptr; size_t slabCap = 0;
    std::vector<void*> msgOut; std::vector<size_t> msgOutCap;   // messages of the beams this device computes
    std::vector<void*> msgIn; std::vector<size_t> msgInCap;     // pulled copies of the other devices' messages
    int status = RTD_OK;
    std::string error;
    float ms[5] = {0, 0, 0, 0, 0};        // upload, bev, exchange, transfer, download
};

// Copies the inclusive index box [lo, hi] of a [z][y][x] float volume between the caller's host buffer and a device slab that
// holds the slices from z0 on (same row and plane pitch): rows of the box when it is narrow, whole x rows (one contiguous run per
// plane) when it spans most of the row anyway.
hipError_t copyBox(bool toDevice, float* host, float* dSlab, const uint32_t dims[3], int z0, const int lo[3], const int hi[3], hipStream_t stream) {
    const size_t nx = dims[0], ny = dims[1];
    int x0 = lo[0], x1 = hi[0];
    if ((size_t)(x1 - x0 + 1) * 2 >= nx) { x0 = 0; x1 = (int)nx - 1; }
    hipMemcpy3DParms p;
    std::memset(&p, 0, sizeof p);
    hipPitchedPtr hp = make_hipPitchedPtr(host, nx * sizeof(float), nx, ny);
    hipPitchedPtr dp = make_hipPitchedPtr(dSlab, nx * sizeof(float), nx, ny);
    const hipPos hpos = make_hipPos((size_t)x0 * sizeof(float), (size_t)lo[1], (size_t)lo[2]);
    const hipPos dpos = make_hipPos((size_t)x0 * sizeof(float), (size_t)lo[1], (size_t)(lo[2] - z0));
    p.srcPtr = toDevice ? hp : dp; p.srcPos = toDevice ? hpos : dpos;
    p.dstPtr = toDevice ? dp : hp; p.dstPos = toDevice ? dpos : hpos;
    p.extent = make_hipExtent((size_t)(x1 - x0 + 1) * sizeof(float), (size_t)(hi[1] - lo[1] + 1), (size_t)(hi[2] - lo[2] + 1));
    p.kind = toDevice ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    return hipMemcpy3DAsync(&p, stream);
}

double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct rtd_plan_s {
    std::vector<DevSlot> dev;
    std::string error;
    // RCCL transport (optional)
    bool useRccl = false;
    bool selfMessages = false;             // RTD_PLAN_SELF_MESSAGES: owners, too, transfer from the exchanged copy (exercises the transport on one GPU)
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
};

// the same call on every device, in parallel (uploads of the replicated inputs overlap)
template <typename F>
static int onAllDevices(rtd_plan_t p, F fn) {
    std::vector<std::thread> th;
    for (size_t d = 0; d < p->dev.size(); ++d)
        th.emplace_back([&, d] {
            DevSlot& s = p->dev[d];
            s.status = fn(s);
            if (s.status != RTD_OK) s.error = rtd_last_error(s.h);
        });
    for (auto& t : th) t.join();
    for (DevSlot& s : p->dev) if (s.status != RTD_OK) { p->error = s.error; return s.status; }
    return RTD_OK;
}

extern "C" {

const char* rtd_plan_last_error(rtd_plan_t p) { return p ? p->error.c_str() : "null plan"; }

int rtd_plan_create(const int* device_ids, int n_devices, rtd_plan_t* out) {
    if (!device_ids || n_devices <= 0 || !out) return RTD_ERR_INVALID_ARG;
    *out = nullptr;
    auto* p = new rtd_plan_s();
    p->dev.resize((size_t)n_devices);
    for (int d = 0; d < n_devices; ++d) {
        DevSlot& s = p->dev[(size_t)d];
        s.device = device_ids[d];
        const int st = rtd_create(device_ids[d], &s.h);
        if (st != RTD_OK) { rtd_plan_destroy(p); return st; }       // message: rtd_global_error()
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.copyStream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming) != hipSuccess) {
            rtd_plan_destroy(p);
            return RTD_ERR_HIP;
        }
    }
    // peer access between distinct devices (xGMI): a failed enable only means the copies are staged by the runtime
    for (int a = 0; a < n_devices; ++a)
        for (int b = 0; b < n_devices; ++b) {
            const int da = p->dev[(size_t)a].device, db = p->dev[(size_t)b].device;
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                (void)hipSetDevice(da);
                if (hipDeviceEnablePeerAccess(db, 0) != hipSuccess) (void)hipGetLastError();   // already enabled is fine
            }
        }
    // transport of the slab exchange: peer copies (default) or RCCL
    const char* tr = std::getenv("RTD_PLAN_TRANSPORT");
    if (tr && std::string(tr) == "rccl") {
        for (int a = 0; a < n_devices; ++a)
            for (int b = a + 1; b < n_devices; ++b)
                if (device_ids[a] == device_ids[b]) { rtd_plan_destroy(p); return RTD_ERR_INVALID_ARG; }   // RCCL: one rank per GPU
        if (!p->rccl.load(p->error)) { rtd_plan_destroy(p); return RTD_ERR_IO; }
        p->comms.assign((size_t)n_devices, nullptr);
        const ncclResult_t r = p->rccl.commInitAll(p->comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) { p->comms.clear(); rtd_plan_destroy(p); return RTD_ERR_HIP; }
        p->useRccl = true;
        const char* sm = std::getenv("RTD_PLAN_SELF_MESSAGES");
        p->selfMessages = sm && sm[0] == '1';
    }
    *out = p;
    return RTD_OK;
}

int rtd_plan_destroy(rtd_plan_t p) {
    if (!p) return RTD_ERR_INVALID_ARG;
    for (ncclComm_t c : p->comms) if (c) (void)p->rccl.commDestroy(c);
    p->comms.clear();
    for (DevSlot& s : p->dev) {
        if (!s.h) continue;
        (void)hipSetDevice(s.device);
        if (s.copyStream) { (void)hipStreamSynchronize(s.copyStream); (void)hipStreamDestroy(s.copyStream); }
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        (void)rtd_sync(s.h);
        if (s.dSlab) (void)rtd_device_free(s.h, s.dSlab);
        for (void* m : s.msgOut) if (m) (void)rtd_device_free(s.h, m);
        for (void* m : s.msgIn) if (m) (void)rtd_device_free(s.h, m);
        (void)rtd_destroy(s.h);
    }
    delete p;
    return RTD_OK;
}

int rtd_plan_set_options(rtd_plan_t p, const rtd_options* opt) {
    if (!p || !opt) return RTD_ERR_INVALID_ARG;
    return onAllDevices(p, [&](DevSlot& s) { return rtd_set_options(s.h, opt); });
}
int rtd_plan_set_luts(rtd_plan_t p, const rtd_luts* l) {
    if (!p || !l) return RTD_ERR_INVALID_ARG;
    return onAllDevices(p, [&](DevSlot& s) { return rtd_set_luts(s.h, l); });
}
int rtd_plan_load_luts_dir(rtd_plan_t p, const char* dir, int water) {
    if (!p || !dir) return RTD_ERR_INVALID_ARG;
    return onAllDevices(p, [&](DevSlot& s) { return rtd_load_luts_dir(s.h, dir, water); });
}
int rtd_plan_set_ct(rtd_plan_t p, const float* hu, const uint32_t dims[3]) {
    if (!p || !hu || !dims) return RTD_ERR_INVALID_ARG;
    return onAllDevices(p, [&](DevSlot& s) { return rtd_set_ct(s.h, hu, dims); });
}
int rtd_plan_set_ct_deferred(rtd_plan_t p, const float* hu, const uint32_t dims[3]) {
    if (!p || !hu || !dims) return RTD_ERR_INVALID_ARG;
    return onAllDevices(p, [&](DevSlot& s) { return rtd_set_ct_deferred(s.h, hu, dims); });
}

int rtd_plan_compute(rtd_plan_t p, const rtd_beam* beams, int n_beams, float* dose_inout, const uint32_t dose_dims[3],
                     rtd_timing* per_beam, rtd_plan_timing* plan_timing) {
    if (!p || !beams || n_beams < 0 || !dose_inout || !dose_dims || !dose_dims[0] || !dose_dims[1] || !dose_dims[2]) return RTD_ERR_INVALID_ARG;
    const int D = (int)p->dev.size();
    const size_t nxy = (size_t)dose_dims[0] * dose_dims[1];
    const int nz = (int)dose_dims[2];
    const double t0 = nowMs();
    Barrier bar(D);
    // shared between the threads, written before a barrier and read after it
    std::vector<size_t> msgBytes((size_t)n_beams, 0);
    std::vector<void*> msgPtr((size_t)n_beams, nullptr);
    std::vector<int> boxLo((size_t)n_beams * 3, 0), boxHi((size_t)n_beams * 3, -1);   // the beams' dose boxes (from their owners)
    std::vector<std::atomic<int>> failed((size_t)D);
    for (auto& v : failed) v.store(0);
    auto anyFailed = [&] { for (auto& v : failed) if (v.load()) return true; return false; };
    // One answer for all device threads: the flags are read between two barriers, where nobody writes them. Whether the threads go
    // on to a step in which they depend on each other (the exchange: a thread that skips or leaves an ncclBroadcast sequence its
    // peers are inside of leaves them blocked forever) is decided with this, never with a thread's own view.
    auto agreeOk = [&] { bar.wait(); const bool ok = !anyFailed(); bar.wait(); return ok; };

    auto worker = [&](int d) {
        DevSlot& s = p->dev[(size_t)d];
        s.status = RTD_OK; s.error.clear();
        for (float& m : s.ms) m = 0.0f;
        auto bad = [&](int st) { if (st != RTD_OK && s.status == RTD_OK) { s.status = st; s.error = rtd_last_error(s.h); failed[(size_t)d].store(1); } return st != RTD_OK; };
        auto badHip = [&](hipError_t e) { if (e != hipSuccess && s.status == RTD_OK) { s.status = RTD_ERR_HIP; s.error = std::string("HIP error: ") + hipGetErrorString(e); failed[(size_t)d].store(1); } return e != hipSuccess; };
        (void)hipSetDevice(s.device);
        hipStream_t stream = (hipStream_t)rtd_stream(s.h);
        // ---- 1. this device's z-slab of the volume (allocation only: what travels is decided after step 2) ----
        const int z0 = (int)((long long)nz * d / D), z1 = (int)((long long)nz * (d + 1) / D) - 1;   // inclusive
        const size_t slabN = (size_t)std::max(z1 - z0 + 1, 0) * nxy;
        double t = nowMs();
        if (slabN > s.slabCap) {
            if (s.dSlab) (void)rtd_device_free(s.h, s.dSlab);
            s.dSlab = nullptr; s.slabCap = 0;
            void* q = nullptr;
            if (!bad(rtd_device_alloc(s.h, slabN * sizeof(float), &q))) { s.dSlab = (float*)q; s.slabCap = slabN; }
        }
        // (kernels index the volume by absolute z: the base pointer is shifted so that slice z0 is the slab's first)
        float* doseBase = s.dSlab ? s.dSlab - (size_t)z0 * nxy : nullptr;
        const int32_t clipLo[3] = {0, 0, z0}, clipHi[3] = {(int32_t)dose_dims[0] - 1, (int32_t)dose_dims[1] - 1, z1};
        s.ms[0] = (float)(nowMs() - t);

        // ---- 2. BEV dose of this device's beams; with more than one device, packed for the others ----
        t = nowMs();
        std::vector<rtd_field> mine((size_t)n_beams, nullptr);
        size_t slot = 0;
        for (int i = d; i < n_beams && s.status == RTD_OK; i += D, ++slot) {
            if (bad(rtd_field_create(s.h, &beams[i], dose_dims, &mine[(size_t)i]))) break;
            if (bad(rtd_field_compute_bev(s.h, mine[(size_t)i]))) break;
            size_t bytes = 0;
            rtd_field_info fi;
            if (bad(rtd_field_wait_plan(s.h, mine[(size_t)i], &fi, &bytes))) break;   // the superposition is still running
            for (int a = 0; a < 3; ++a) { boxLo[(size_t)i * 3 + a] = fi.dose_box_min[a]; boxHi[(size_t)i * 3 + a] = fi.dose_box_max[a]; }
            if (D == 1 && !(p->useRccl && p->selfMessages)) continue;
            if (slot >= s.msgOut.size()) { s.msgOut.push_back(nullptr); s.msgOutCap.push_back(0); }
            if (bytes > s.msgOutCap[slot]) {
                if (s.msgOut[slot]) (void)rtd_device_free(s.h, s.msgOut[slot]);
                s.msgOut[slot] = nullptr; s.msgOutCap[slot] = 0;
                if (bad(rtd_device_alloc(s.h, bytes, &s.msgOut[slot]))) break;
                s.msgOutCap[slot] = bytes;
            }
            if (bad(rtd_field_export_bev(s.h, mine[(size_t)i], s.msgOut[slot], s.msgOutCap[slot]))) break;
            msgBytes[(size_t)i] = bytes; msgPtr[(size_t)i] = s.msgOut[slot];
        }
        if (s.status == RTD_OK && (D > 1 || (p->useRccl && p->selfMessages))) bad(rtd_sync(s.h));   // the messages are complete before they travel
        s.ms[1] = (float)(nowMs() - t);
        const bool bevOk = agreeOk();                                 // (also: boxes, message sizes and pointers of all beams are published)

        // ---- 1b. upload the block of the slab that the plan can change: bounding box of all beams' dose boxes, cut to the slab ----
        t = nowMs();
        int blkLo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, blkHi[3] = {-1, -1, -1};
        for (int i = 0; i < n_beams; ++i) {
            const int* lo = &boxLo[(size_t)i * 3]; const int* hi = &boxHi[(size_t)i * 3];
            if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) continue;
            for (int a = 0; a < 3; ++a) { blkLo[a] = std::min(blkLo[a], lo[a]); blkHi[a] = std::max(blkHi[a], hi[a]); }
        }
        blkLo[2] = std::max(blkLo[2], z0); blkHi[2] = std::min(blkHi[2], z1);
        const bool haveBlock = slabN && blkHi[0] >= blkLo[0] && blkHi[1] >= blkLo[1] && blkHi[2] >= blkLo[2];
        // on the copy stream, so that it runs beside the superposition kernels still in flight; the transfers wait for its event
        if (bevOk && haveBlock) {
            if (!badHip(copyBox(true, dose_inout, s.dSlab, dose_dims, z0, blkLo, blkHi, s.copyStream)) && !badHip(hipEventRecord(s.uploaded, s.copyStream)))
                badHip(hipStreamWaitEvent(stream, s.uploaded, 0));
        }
        s.ms[0] += (float)(nowMs() - t);

        // ---- 3. exchange of the messages: peer copies over xGMI (same-device slots are read in place), or RCCL broadcasts ----
        t = nowMs();
        std::vector<const void*> slabOf((size_t)n_beams, nullptr);
        std::vector<rtd_field> theirs((size_t)n_beams, nullptr);
        std::vector<int> inSlot((size_t)n_beams, -1);
        const bool viaMessage = p->useRccl && p->selfMessages;         // owners read the exchanged copy too
        const bool exchange = D > 1 || viaMessage;
        // 3a. every receive buffer is allocated BEFORE the first collective, then the threads agree: all enter the exchange or none
        if (bevOk && exchange) {
            size_t in = 0;
            for (int i = 0; i < n_beams && s.status == RTD_OK; ++i) {
                const int owner = i % D;
                if (owner == d && !viaMessage && !p->useRccl) continue;
                const DevSlot& o = p->dev[(size_t)owner];
                if (!p->useRccl && o.device == s.device) { slabOf[(size_t)i] = msgPtr[(size_t)i]; continue; }
                if (in >= s.msgIn.size()) { s.msgIn.push_back(nullptr); s.msgInCap.push_back(0); }
                if (msgBytes[(size_t)i] > s.msgInCap[in]) {
                    if (s.msgIn[in]) (void)rtd_device_free(s.h, s.msgIn[in]);
                    s.msgIn[in] = nullptr; s.msgInCap[in] = 0;
                    if (bad(rtd_device_alloc(s.h, msgBytes[(size_t)i], &s.msgIn[in]))) break;
                    s.msgInCap[in] = msgBytes[(size_t)i];
                }
                inSlot[(size_t)i] = (int)in;
                ++in;
            }
        }
        const bool exchangeOk = agreeOk() && bevOk;
        // 3b. the exchange itself; with RCCL no thread leaves the sequence of broadcasts early (an error is recorded, the calls go on)
        if (exchangeOk && exchange) {
            for (int i = 0; i < n_beams; ++i) {
                const int in = inSlot[(size_t)i];
                if (in < 0) continue;
                const int owner = i % D;
                const DevSlot& o = p->dev[(size_t)owner];
                if (p->useRccl) {
                    // every device makes the call for every beam, in beam order: root = the device that computed it
                    const ncclResult_t r = p->rccl.broadcast(owner == d ? msgPtr[(size_t)i] : s.msgIn[(size_t)in], s.msgIn[(size_t)in], msgBytes[(size_t)i], ncclUint8, owner,
                                                             p->comms[(size_t)d], s.copyStream);
                    if (r != ncclSuccess && s.status == RTD_OK) { s.status = RTD_ERR_HIP; s.error = std::string("RCCL: ") + p->rccl.errorString(r); failed[(size_t)d].store(1); }
                } else if (s.status == RTD_OK) badHip(hipMemcpyPeerAsync(s.msgIn[(size_t)in], s.device, msgPtr[(size_t)i], o.device, msgBytes[(size_t)i], s.copyStream));
                slabOf[(size_t)i] = s.msgIn[(size_t)in];
            }
            badHip(hipStreamSynchronize(s.copyStream));
        }
        s.ms[2] = (float)(nowMs() - t);

        // ---- 4. transfers of ALL beams into this device's slab, in beam order (= the reference's `+=` order) ----
        t = nowMs();
        const bool transferOk = agreeOk() && exchangeOk;
        if (transferOk && slabN) {
            for (int i = 0; i < n_beams && s.status == RTD_OK; ++i) {
                if (i % D == d && !viaMessage) { bad(rtd_field_transfer(s.h, mine[(size_t)i], doseBase, clipLo, clipHi)); continue; }
                if (bad(rtd_field_create_remote(s.h, &beams[i], dose_dims, &theirs[(size_t)i]))) break;
                if (bad(rtd_field_attach_bev(s.h, theirs[(size_t)i], slabOf[(size_t)i]))) break;
                bad(rtd_field_transfer(s.h, theirs[(size_t)i], doseBase, clipLo, clipHi));
            }
        }
        // device-side errors (radius overflow) and per-beam timing come from the device that computed the beam
        for (int i = d; i < n_beams; i += D)
            if (mine[(size_t)i] && s.status == RTD_OK) bad(rtd_field_finish(s.h, mine[(size_t)i], per_beam ? &per_beam[i] : nullptr, nullptr));
        if (s.status == RTD_OK) bad(rtd_sync(s.h));
        s.ms[3] = (float)(nowMs() - t);
        const bool allOk = agreeOk();                                 // (also: nobody frees a message another device still reads)

        // ---- 5. download this device's slab (:1318) — only when every device succeeded: the volume is all-or-nothing ----
        t = nowMs();
        if (allOk && haveBlock) {
            if (!badHip(copyBox(false, dose_inout, s.dSlab, dose_dims, z0, blkLo, blkHi, stream)))
                badHip(hipStreamSynchronize(stream));
        }
        s.ms[4] = (float)(nowMs() - t);
        for (rtd_field f : theirs) if (f) (void)rtd_field_destroy(s.h, f);
        for (rtd_field f : mine) if (f) (void)rtd_field_release(s.h, f);   // workspaces stay with the handle for the next call
    };

    std::vector<std::thread> th;
    for (int d = 1; d < D; ++d) th.emplace_back(worker, d);
    worker(0);
    for (auto& t : th) t.join();
    int status = RTD_OK;
    for (DevSlot& s : p->dev) if (s.status != RTD_OK && status == RTD_OK) { status = s.status; p->error = s.error; }
    if (plan_timing) {
        std::memset(plan_timing, 0, sizeof *plan_timing);
        plan_timing->total_ms = (float)(nowMs() - t0);
        plan_timing->n_devices = D;
        for (const DevSlot& s : p->dev) {
            plan_timing->upload_ms = std::max(plan_timing->upload_ms, s.ms[0]);
            plan_timing->bev_ms = std::max(plan_timing->bev_ms, s.ms[1]);
            plan_timing->exchange_ms = std::max(plan_timing->exchange_ms, s.ms[2]);
            plan_timing->transfer_ms = std::max(plan_timing->transfer_ms, s.ms[3]);
            plan_timing->download_ms = std::max(plan_timing->download_ms, s.ms[4]);
        }
    }
    return status;
}

}  // extern "C"
