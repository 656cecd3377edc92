// icet_amd/csrc/icet_multi.hip -- the batched-pairs case over several GPUs of one node, behind the C ABI (include/icet_hip.h).
//
// The path shards across PAIRS only (each pair is an independent ICET object in the reference; the only reduction inside a pair
// is V -> 1 of 27 floats): pair k goes to device_ids[k mod n_devices] (BASELINE.json configs[3]), every device runs its own
// context on its own PERSISTENT host thread (created with the handle: hipSetDevice is per thread, and a thread per call cost
// ~50 us x devices on a 2.5 ms step), and the 48 result floats per pair are gathered into one buffer -- host memory for the
// host-pointer entry, HBM of the first device for the device-resident one.  No data-path collective.  The gather of the
// device-resident entry has two forms (icet_multi_set_option "gather"):
//   0  peer copies over xGMI: one strided hipMemcpy2DAsync per device on that device's stream (peer access is enabled between
//      device_ids[0] and every other device when the handle is created) -- 192 B per pair, the default;
//   1  RCCL: ncclCommInitAll over the handle's devices and ONE ncclAllGather of the (padded) shards, then the round-robin
//      interleave is undone on device_ids[0] -- what BASELINE.json's north_star names.  librccl is dlopen'ed on first use, so a
//      caller that never asks for it pays neither its load time nor its start-up.
// A process that already runs one rank per GPU uses torch.distributed / RCCL for the same gather (icet_amd/dist.py); this entry is
// for a single-process C++ caller, which is what the reference's nodes are.
#include "../../include/icet_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "icet_multi_sched.h"

namespace {

using icet_sched::Worker;      // one persistent host thread per device entry with a FIFO of jobs, and the two-phase protocol of a call: icet_multi_sched.h

// The handful of RCCL entry points the gather needs, resolved from librccl.so.1 on first use.
struct Rccl {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string& err) {
        if (so) return true;
        so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!so) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(so, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(so, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(so, "ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(so, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) { err = "librccl lacks ncclCommInitAll / ncclAllGather"; dlclose(so); so = nullptr; return false; }
        return true;
    }
};

}  // namespace

struct icet_multi {
    std::vector<int> dev;
    std::vector<icet_ctx*> ctx;
    std::vector<float*> d_part;          // per device: results of its pairs (device-resident entry), cap x 48, then cap x 6 of X0
    std::vector<int32_t> cap_part;
    icet_sched::Sched sched;             // the device threads and the record of their failures (icet_multi_sched.h)
    hipEvent_t ev_producer = nullptr;    // recorded on the caller's stream (device_ids[0]); every device's stream waits for it
    int gather_mode = 0;                 // 0 peer copies, 1 RCCL all-gather
    Rccl rccl;
    std::vector<ncclComm_t> comms;       // one per device entry once gather_mode 1 has been used
    std::vector<float*> d_full;          // per device: receive buffer of the all-gather (n_devices x m x 48)
    std::vector<int32_t> cap_full;
    // asynchronous calls: the first failure of every device since the last icet_multi_sync, and one event per device recorded behind
    // the last gather it enqueued
    std::vector<hipEvent_t> ev_done;
    std::string err;
};

namespace {

void set_err(icet_multi* m, int d, icet_ctx* c, const char* what) {
    m->err = std::string(what) + " (device " + std::to_string(m->dev[d]) + "): " + (c ? icet_last_error(c) : "no context");
}

// run fn(d) on every device's worker and wait for all of them
template <typename F> void run_all(icet_multi* m, F fn) {
    m->sched.run_all(fn);
}

bool ids_distinct(const icet_multi* m) {
    for (size_t i = 0; i < m->dev.size(); i++) for (size_t j = i + 1; j < m->dev.size(); j++) if (m->dev[i] == m->dev[j]) return false;
    return true;
}

icet_status ensure_comms(icet_multi* m) {
    if (!m->comms.empty()) return ICET_OK;
    if (!ids_distinct(m)) { m->err = "RCCL gather needs distinct devices (one rank per GPU)"; return ICET_ERR_UNSUPPORTED; }
    if (!m->rccl.load(m->err)) return ICET_ERR_UNSUPPORTED;
    const int D = (int)m->dev.size();
    std::vector<ncclComm_t> comms(D, nullptr);
    ncclResult_t r = m->rccl.CommInitAll(comms.data(), D, m->dev.data());
    if (r != ncclSuccess) { m->err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r); return ICET_ERR_HIP; }
    m->comms = comms;
    return ICET_OK;
}

}  // namespace

extern "C" {

icet_status icet_multi_create(icet_multi** out, const int32_t* device_ids, int32_t n_devices) {
    if (!out) return ICET_ERR_BAD_ARG;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return ICET_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ICET_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; i++) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return ICET_ERR_NO_DEVICE;
    }
    // (a device may appear more than once: each entry gets its own context, host thread and result buffer on that device -- two
    // shards sharing one GPU.  Of no use in production, but it is how the N > 1 sharding, X0 scatter and result gather are exercised
    // on a one-GPU box: tests/test_gpu_parity.py)
    icet_multi* m = new (std::nothrow) icet_multi();
    if (!m) return ICET_ERR_NOMEM;
    try {
        m->dev.assign(device_ids, device_ids + n_devices);
        m->ctx.assign(n_devices, nullptr); m->d_part.assign(n_devices, nullptr); m->cap_part.assign(n_devices, 0);
        m->d_full.assign(n_devices, nullptr); m->cap_full.assign(n_devices, 0);
        m->ev_done.assign(n_devices, nullptr);
    } catch (...) { icet_multi_destroy(m); return ICET_ERR_NOMEM; }
    if (!m->sched.start(n_devices)) { icet_multi_destroy(m); return ICET_ERR_NOMEM; }      // (std::system_error / bad_alloc: the threads that started have been joined)
    for (int i = 0; i < n_devices; i++) {
        icet_status s = icet_create(&m->ctx[i], device_ids[i], nullptr);
        if (s != ICET_OK) { icet_multi_destroy(m); return s; }
    }
    // Peer access between the gathering device and every other one, both ways (X0 is read from it, results are written to it).
    // Where the platform offers none the copies below still work (the runtime stages them through the host); nothing to report.
    for (int i = 1; i < n_devices; i++) {
        const int a = device_ids[0], b = device_ids[i];
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess) { hipError_t e = hipDeviceEnablePeerAccess(b, 0); (void)e; }
        can = 0;
        if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can && hipSetDevice(b) == hipSuccess) { hipError_t e = hipDeviceEnablePeerAccess(a, 0); (void)e; }
        (void)hipGetLastError();                                   // hipErrorPeerAccessAlreadyEnabled is fine
    }
    for (int i = 0; i < n_devices; i++)
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming) != hipSuccess) { icet_multi_destroy(m); return ICET_ERR_HIP; }
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipEventCreateWithFlags(&m->ev_producer, hipEventDisableTiming) != hipSuccess) { icet_multi_destroy(m); return ICET_ERR_HIP; }
    *out = m;
    return ICET_OK;
}

icet_status icet_multi_destroy(icet_multi* m) {
    if (!m) return ICET_ERR_BAD_ARG;
    m->sched.stop();
    for (size_t i = 0; i < m->comms.size(); i++) if (m->comms[i]) (void)m->rccl.CommDestroy(m->comms[i]);
    for (size_t i = 0; i < m->ctx.size(); i++) {
        (void)hipSetDevice(m->dev[i]);
        if (m->d_part[i]) (void)hipFree(m->d_part[i]);
        if (i < m->d_full.size() && m->d_full[i]) (void)hipFree(m->d_full[i]);
        if (i < m->ev_done.size() && m->ev_done[i]) (void)hipEventDestroy(m->ev_done[i]);
        if (m->ctx[i]) (void)icet_destroy(m->ctx[i]);
    }
    if (m->ev_producer) { (void)hipSetDevice(m->dev[0]); (void)hipEventDestroy(m->ev_producer); }
    delete m;
    return ICET_OK;
}

const char* icet_multi_last_error(const icet_multi* m) { return m ? m->err.c_str() : "null handle"; }
int32_t icet_multi_devices(const icet_multi* m) { return m ? (int32_t)m->dev.size() : 0; }
icet_ctx* icet_multi_context(icet_multi* m, int32_t i) { return (m && i >= 0 && i < (int32_t)m->ctx.size()) ? m->ctx[i] : nullptr; }

icet_status icet_multi_set_option(icet_multi* m, const char* name, double value) {
    if (!m || !name) return ICET_ERR_BAD_ARG;
    const std::string k(name);
    if (k == "gather") {
        const int v = (int)value;
        if (v != 0 && v != 1) { m->err = "gather: 0 (peer copies) or 1 (RCCL all-gather)"; return ICET_ERR_BAD_ARG; }
        if (v == 1) { icet_status s = ensure_comms(m); if (s != ICET_OK) return s; }
        m->gather_mode = v;
        return ICET_OK;
    }
    // everything else is a per-context knob: apply it to every device's context
    for (size_t i = 0; i < m->ctx.size(); i++) {
        icet_status s = icet_set_option(m->ctx[i], name, value);
        if (s != ICET_OK) { set_err(m, (int)i, m->ctx[i], "icet_set_option"); return s; }
    }
    return ICET_OK;
}

// N independent pairs from HOST memory.  Pair k runs on device k mod n_devices; each device's share goes through icet_solve_batch
// on that device's context from its own host thread (hipSetDevice is per thread), and the threads write disjoint rows of the
// caller's arrays, so the "gather" is the return of the threads.
icet_status icet_multi_solve_batch(icet_multi* m, const icet_params* p, int32_t n_pairs,
                                   const float* const* scan1, const int64_t* n1, const float* const* scan2, const int64_t* n2,
                                   const float* x0, float* x_out, float* pred_stds_out, float* cov_out) {
    if (!m) return ICET_ERR_BAD_ARG;
    if (!p || n_pairs < 0 || (n_pairs > 0 && (!scan1 || !n1 || !scan2 || !n2 || !x_out || !pred_stds_out))) { m->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    if (n_pairs == 0) return ICET_OK;
    const int D = (int)m->dev.size();
    std::vector<icet_status> st;
    try { st.assign(D, ICET_OK); } catch (...) { m->err = "out of host memory"; return ICET_ERR_NOMEM; }
    icet_status* stp = st.data();
    try {
        run_all(m, [=](int d) {
            try {
                std::vector<const float*> a, b; std::vector<int64_t> na, nb; std::vector<int> idx;
                for (int k = d; k < n_pairs; k += D) { a.push_back(scan1[k]); b.push_back(scan2[k]); na.push_back(n1[k]); nb.push_back(n2[k]); idx.push_back(k); }
                const int np = (int)idx.size();
                if (!np) return;
                std::vector<float> xin, xo((size_t)np * 6), po((size_t)np * 6), co(cov_out ? (size_t)np * 36 : 0);
                if (x0) { xin.resize((size_t)np * 6); for (int j = 0; j < np; j++) std::memcpy(&xin[6 * j], x0 + 6 * (size_t)idx[j], 6 * sizeof(float)); }
                stp[d] = icet_solve_batch(m->ctx[d], p, np, a.data(), na.data(), b.data(), nb.data(), x0 ? xin.data() : nullptr, xo.data(), po.data(),
                                          cov_out ? co.data() : nullptr);
                if (stp[d] != ICET_OK) return;
                for (int j = 0; j < np; j++) {
                    std::memcpy(x_out + 6 * (size_t)idx[j], &xo[6 * j], 6 * sizeof(float));
                    std::memcpy(pred_stds_out + 6 * (size_t)idx[j], &po[6 * j], 6 * sizeof(float));
                    if (cov_out) std::memcpy(cov_out + 36 * (size_t)idx[j], &co[36 * j], 36 * sizeof(float));
                }
            } catch (...) { stp[d] = ICET_ERR_NOMEM; }
        });
    } catch (...) {                                     // posting a job allocates (std::function): run_all has waited for the ones already posted
        m->err = "cannot hand the work to the device threads"; return ICET_ERR_NOMEM;
    }
    for (int d = 0; d < D; d++) if (st[d] != ICET_OK) { set_err(m, d, m->ctx[d], "icet_solve_batch"); return st[d]; }
    return ICET_OK;
}

// N independent pairs RESIDENT IN HBM: scan1[k] / scan2[k] live on device k mod n_devices (the caller placed them there, e.g. the
// driver of a sensor rig feeding each GPU its share); d_x0 (n_pairs x 6, may be NULL) and d_out (n_pairs x 48) live on device 0 of
// the handle.  Each device solves its share into a local buffer; the rows are then gathered into d_out (peer copies or RCCL, see
// the head of this file), ordered after the solve on that device's stream.
// ORDERING.  The devices' streams are the contexts' own; they know nothing about the stream that produced d_x0 / the scans or that
// last used d_out.  `producer_stream` (a hipStream_t of device_ids[0] passed as void*, NULL = none) closes that gap for work queued
// on ONE stream: an event recorded on it when the call starts is waited for by every device's stream.  Anything else -- scans
// written on other devices' streams -- must have completed before the call.
// ASYNCHRONOUS FORM.  icet_multi_solve_batch_device_async hands every device's share to its host thread and returns at once: the
// threads enqueue the solve and the gather on their streams and record an event behind them; nothing waits for the device.  Calls queue
// up behind each other (per device: one FIFO of jobs, one stream).  icet_multi_sync waits for the threads and for the events and
// reports the first failure since the last sync.  The synchronous entries are async + sync.
// Two phases per call, because a rank that skipped the RCCL collective would hang the others: (1) everything that can fail before the
// collective for reasons of the HOST (device selection, buffer growth) runs first on every thread and is checked on the calling
// thread -- it allocates only when a capacity grows; (2) only then are the solve + gather jobs posted, and a job whose solve fails
// still enters the collective with what it has.
static icet_status multi_enqueue(icet_multi* m, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                 const float* d_x0, float* d_out, void* producer_stream) {
    if (!m) return ICET_ERR_BAD_ARG;
    if (!p || n_pairs < 0 || (n_pairs > 0 && (!scan1 || !scan2 || !d_out))) { m->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    if (n_pairs == 0) return ICET_OK;
    const int D = (int)m->dev.size();
    const int mode = m->gather_mode;
    if (mode == 1) { icet_status s = ensure_comms(m); if (s != ICET_OK) return s; }
    const bool wait_producer = producer_stream != nullptr;
    if (wait_producer) {
        if (hipSetDevice(m->dev[0]) != hipSuccess || hipEventRecord(m->ev_producer, reinterpret_cast<hipStream_t>(producer_stream)) != hipSuccess) {
            m->err = "cannot record an event on producer_stream (is it a stream of device_ids[0]?)"; (void)hipGetLastError(); return ICET_ERR_BAD_ARG;
        }
    }
    const int mrows = (n_pairs + D - 1) / D;              // largest share: the all-gather's (padded) count per rank
    // ---- phase 1: buffers (only when a capacity grows; earlier jobs of the device must have drained before its buffers move) ----
    bool grow = false;
    for (int d = 0; d < D; d++) {
        const int np = (n_pairs - d + D - 1) / D;
        if ((mode == 1 ? mrows : np) > m->cap_part[d] || (mode == 1 && D * mrows > m->cap_full[d])) grow = true;
    }
    if (grow) {
        int fail_status = 0;
        const int bad = m->sched.prepare_all([=](int d) -> int {
            const int np = (n_pairs - d + D - 1) / D;
            const int need = mode == 1 ? mrows : np;
            if (hipSetDevice(m->dev[d]) != hipSuccess) return (int)ICET_ERR_NO_DEVICE;
            hipStream_t s = reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d]));
            if (hipStreamSynchronize(s) != hipSuccess) return (int)ICET_ERR_HIP;
            if (need > m->cap_part[d]) {
                if (m->d_part[d]) { (void)hipFree(m->d_part[d]); m->d_part[d] = nullptr; m->cap_part[d] = 0; }
                // results (48 floats) and the share's X0 (6 floats) per pair
                if (hipMalloc(reinterpret_cast<void**>(&m->d_part[d]), sizeof(float) * 54 * (size_t)need) != hipSuccess) return (int)ICET_ERR_NOMEM;
                (void)hipMemsetAsync(m->d_part[d], 0, sizeof(float) * 54 * (size_t)need, s);
                m->cap_part[d] = need;
            }
            if (mode == 1 && D * mrows > m->cap_full[d]) {
                if (m->d_full[d]) { (void)hipFree(m->d_full[d]); m->d_full[d] = nullptr; m->cap_full[d] = 0; }
                if (hipMalloc(reinterpret_cast<void**>(&m->d_full[d]), sizeof(float) * 48 * (size_t)D * mrows) != hipSuccess) return (int)ICET_ERR_NOMEM;
                m->cap_full[d] = D * mrows;
            }
            return 0;
        }, &fail_status);
        if (bad == -2) { m->err = "cannot hand the work to the device threads"; return ICET_ERR_NOMEM; }
        if (bad >= 0) { m->err = "device " + std::to_string(m->dev[bad]) + ": cannot grow the result buffers"; return (icet_status)fail_status; }
    }
    // ---- phase 2: solve + gather, asynchronously.  The shares are cut here (the caller's descriptor arrays need not outlive the call) ----
    const icet_params prm = *p;
    try {
        auto as = std::make_shared<std::vector<std::vector<icet_dev_scan>>>(D), bs = std::make_shared<std::vector<std::vector<icet_dev_scan>>>(D);
        for (int k = 0; k < n_pairs; k++) { (*as)[k % D].push_back(scan1[k]); (*bs)[k % D].push_back(scan2[k]); }
        const bool posted = m->sched.post_all(
            // the rank's share: its rows of X0 (rows d, d + D, ... of the buffer on device 0: a strided peer copy; same device when d == 0), then the solve
            [m, D, prm, d_x0, wait_producer, as, bs](int d, std::string& why) -> int {
                const std::vector<icet_dev_scan>& a = (*as)[d]; const std::vector<icet_dev_scan>& b = (*bs)[d];
                const int np = (int)a.size();
                hipError_t e = hipSetDevice(m->dev[d]);
                hipStream_t s = reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d]));
                float* part = m->d_part[d]; float* px0 = part + 48 * (size_t)m->cap_part[d];
                if (e == hipSuccess && wait_producer) e = hipStreamWaitEvent(s, m->ev_producer, 0);
                if (e == hipSuccess && d_x0 && np) e = hipMemcpy2DAsync(px0, 6 * sizeof(float), d_x0 + 6 * (size_t)d, 6 * sizeof(float) * D, 6 * sizeof(float), np, hipMemcpyDefault, s);
                if (e != hipSuccess) { why = hipGetErrorString(e); return (int)ICET_ERR_HIP; }
                if (np) { const icet_status st = icet_solve_batch_device(m->ctx[d], &prm, np, a.data(), b.data(), d_x0 ? px0 : nullptr, part); if (st != ICET_OK) { why = icet_last_error(m->ctx[d]); return (int)st; } }
                return 0;
            },
            // the gather: entered whatever the solve returned when it is a collective (RCCL); the peer copy of a failed share is skipped
            [m, D, mode, mrows, n_pairs, d_out, as](int d, int solve_status, std::string& why) -> int {
                const int np = (int)(*as)[d].size();
                hipStream_t s = reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d]));
                float* part = m->d_part[d];
                hipError_t e = hipSuccess;
                if (mode == 1) {
                    ncclResult_t r = m->rccl.AllGather(part, m->d_full[d], (size_t)mrows * 48, ncclFloat, m->comms[d], s);
                    if (r != ncclSuccess) { why = std::string("ncclAllGather: ") + m->rccl.GetErrorString(r); return (int)ICET_ERR_HIP; }
                    if (d == 0) {
                        // undo the round-robin interleave on the gathering device: block r of the receive buffer holds pairs r, r + D, ...
                        for (int r2 = 0; r2 < D && e == hipSuccess; r2++) {
                            const int rows = (n_pairs - r2 + D - 1) / D;
                            if (rows > 0) e = hipMemcpy2DAsync(d_out + 48 * (size_t)r2, 48 * sizeof(float) * D, m->d_full[0] + 48 * (size_t)r2 * mrows, 48 * sizeof(float),
                                                               48 * sizeof(float), rows, hipMemcpyDeviceToDevice, s);
                        }
                    }
                } else if (np && solve_status == 0) {
                    e = hipMemcpy2DAsync(d_out + 48 * (size_t)d, 48 * sizeof(float) * D, part, 48 * sizeof(float), 48 * sizeof(float), np, hipMemcpyDefault, s);
                }
                if (e != hipSuccess) { why = hipGetErrorString(e); return (int)ICET_ERR_HIP; }
                return 0;
            },
            [m](int d, std::string& why) -> int {
                const hipError_t e = hipEventRecord(m->ev_done[d], reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d])));
                if (e != hipSuccess) { why = hipGetErrorString(e); return (int)ICET_ERR_HIP; }
                return 0;
            });
        if (!posted) { m->err = "cannot hand the work to the device threads"; return ICET_ERR_NOMEM; }      // (the jobs already posted still run and are collected by the sync)
    } catch (...) {                                     // cutting the shares allocates
        m->err = "cannot hand the work to the device threads"; return ICET_ERR_NOMEM;
    }
    return ICET_OK;
}

icet_status icet_multi_sync(icet_multi* m) {
    if (!m) return ICET_ERR_BAD_ARG;
    int status = 0; std::string why;
    const int bad = m->sched.sync([m](int d, std::string& msg) -> int {      // every job has enqueued its work and recorded its event: drain the device
        hipError_t e = hipSetDevice(m->dev[d]);
        if (e == hipSuccess) e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d])));
        if (e != hipSuccess) { msg = hipGetErrorString(e); return (int)ICET_ERR_HIP; }
        return 0;
    }, &status, &why);
    if (bad < 0) return ICET_OK;
    m->err = "device " + std::to_string(m->dev[bad]) + ": " + why;
    return (icet_status)status;
}

icet_status icet_multi_solve_batch_device_async(icet_multi* m, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                                const float* d_x0, float* d_out, void* producer_stream) {
    return multi_enqueue(m, p, n_pairs, scan1, scan2, d_x0, d_out, producer_stream);
}

icet_status icet_multi_solve_batch_device_after(icet_multi* m, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                                const float* d_x0, float* d_out, void* producer_stream) {
    const icet_status s = multi_enqueue(m, p, n_pairs, scan1, scan2, d_x0, d_out, producer_stream);
    if (s != ICET_OK) { if (m) (void)icet_multi_sync(m); return s; }
    return n_pairs > 0 ? icet_multi_sync(m) : ICET_OK;
}

icet_status icet_multi_solve_batch_device(icet_multi* m, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                          const float* d_x0, float* d_out) {
    return icet_multi_solve_batch_device_after(m, p, n_pairs, scan1, scan2, d_x0, d_out, nullptr);
}

}  // extern "C"
