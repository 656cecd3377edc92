// icet_amd/csrc/icet_multi.hip -- the batched-pairs case over several GPUs of one node, behind the C ABI (include/icet_hip.h).
//
// The path shards across PAIRS only (each pair is an independent ICET object in the reference; the only reduction inside a pair
// is V -> 1 of 27 floats): pair k goes to device_ids[k mod n_devices] (BASELINE.json configs[3]), every device runs its own
// context on its own host thread, and the 48 result floats per pair are gathered into one buffer -- host memory for the host-pointer
// entry, HBM of the first device (peer copies over xGMI, one per device) for the device-resident one.  No data-path collective: a
// process that already runs one rank per GPU uses torch.distributed / RCCL for the same gather (icet_amd/dist.py); this entry is
// for a single-process C++ caller, which is what the reference's nodes are.
#include "../../include/icet_hip.h"

#include <hip/hip_runtime.h>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct icet_multi {
    std::vector<int> dev;
    std::vector<icet_ctx*> ctx;
    std::vector<float*> d_part;          // per device: results of its pairs (device-resident entry), n_cap x 48
    std::vector<int32_t> cap_part;
    std::string err;
};

namespace {

void set_err(icet_multi* m, int d, icet_ctx* c, const char* what) {
    m->err = std::string(what) + " (device " + std::to_string(m->dev[d]) + "): " + (c ? icet_last_error(c) : "no context");
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
    } catch (const std::bad_alloc&) { delete m; return ICET_ERR_NOMEM; }
    for (int i = 0; i < n_devices; i++) {
        icet_status s = icet_create(&m->ctx[i], device_ids[i], nullptr);
        if (s != ICET_OK) { icet_multi_destroy(m); return s; }
    }
    *out = m;
    return ICET_OK;
}

icet_status icet_multi_destroy(icet_multi* m) {
    if (!m) return ICET_ERR_BAD_ARG;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        if (m->d_part[i]) { (void)hipSetDevice(m->dev[i]); (void)hipFree(m->d_part[i]); }
        if (m->ctx[i]) (void)icet_destroy(m->ctx[i]);
    }
    delete m;
    return ICET_OK;
}

const char* icet_multi_last_error(const icet_multi* m) { return m ? m->err.c_str() : "null handle"; }
int32_t icet_multi_devices(const icet_multi* m) { return m ? (int32_t)m->dev.size() : 0; }
icet_ctx* icet_multi_context(icet_multi* m, int32_t i) { return (m && i >= 0 && i < (int32_t)m->ctx.size()) ? m->ctx[i] : nullptr; }

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
    std::vector<icet_status> st(D, ICET_OK);
    try {
        std::vector<std::thread> th;
        for (int d = 0; d < D; d++) {
            th.emplace_back([&, d]() {
                try {
                    std::vector<const float*> a, b; std::vector<int64_t> na, nb; std::vector<int> idx;
                    for (int k = d; k < n_pairs; k += D) { a.push_back(scan1[k]); b.push_back(scan2[k]); na.push_back(n1[k]); nb.push_back(n2[k]); idx.push_back(k); }
                    const int np = (int)idx.size();
                    if (!np) return;
                    std::vector<float> xin, xo((size_t)np * 6), po((size_t)np * 6), co(cov_out ? (size_t)np * 36 : 0);
                    if (x0) { xin.resize((size_t)np * 6); for (int j = 0; j < np; j++) std::memcpy(&xin[6 * j], x0 + 6 * (size_t)idx[j], 6 * sizeof(float)); }
                    st[d] = icet_solve_batch(m->ctx[d], p, np, a.data(), na.data(), b.data(), nb.data(), x0 ? xin.data() : nullptr, xo.data(), po.data(),
                                             cov_out ? co.data() : nullptr);
                    if (st[d] != ICET_OK) return;
                    for (int j = 0; j < np; j++) {
                        std::memcpy(x_out + 6 * (size_t)idx[j], &xo[6 * j], 6 * sizeof(float));
                        std::memcpy(pred_stds_out + 6 * (size_t)idx[j], &po[6 * j], 6 * sizeof(float));
                        if (cov_out) std::memcpy(cov_out + 36 * (size_t)idx[j], &co[36 * j], 36 * sizeof(float));
                    }
                } catch (const std::bad_alloc&) { st[d] = ICET_ERR_NOMEM; }
            });
        }
        for (auto& t : th) t.join();
    } catch (...) { m->err = "cannot start a host thread per device"; return ICET_ERR_NOMEM; }
    for (int d = 0; d < D; d++) if (st[d] != ICET_OK) { set_err(m, d, m->ctx[d], "icet_solve_batch"); return st[d]; }
    return ICET_OK;
}

// N independent pairs RESIDENT IN HBM: scan1[k] / scan2[k] live on device k mod n_devices (the caller placed them there, e.g. the
// driver of a sensor rig feeding each GPU its share); d_x0 (n_pairs x 6, may be NULL) and d_out (n_pairs x 48) live on device 0 of
// the handle.  Each device solves its share into a local buffer; the rows are then gathered into d_out with one strided peer copy
// per device (hipMemcpy2DAsync over xGMI: 192 B per pair), ordered after the solve on that device's stream.  Returns when the
// gather has completed (the call synchronises every device's stream).
icet_status icet_multi_solve_batch_device(icet_multi* m, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                          const float* d_x0, float* d_out) {
    if (!m) return ICET_ERR_BAD_ARG;
    if (!p || n_pairs < 0 || (n_pairs > 0 && (!scan1 || !scan2 || !d_out))) { m->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    if (n_pairs == 0) return ICET_OK;
    const int D = (int)m->dev.size();
    std::vector<icet_status> st(D, ICET_OK);
    std::vector<std::string> herr(D);
    try {
        std::vector<std::thread> th;
        for (int d = 0; d < D; d++) {
            th.emplace_back([&, d]() {
                try {
                    std::vector<icet_dev_scan> a, b;
                    for (int k = d; k < n_pairs; k += D) { a.push_back(scan1[k]); b.push_back(scan2[k]); }
                    const int np = (int)a.size();
                    if (!np) return;
                    if (hipSetDevice(m->dev[d]) != hipSuccess) { st[d] = ICET_ERR_NO_DEVICE; return; }
                    hipStream_t s = reinterpret_cast<hipStream_t>(icet_stream(m->ctx[d]));
                    if (np > m->cap_part[d]) {
                        if (m->d_part[d]) { (void)hipFree(m->d_part[d]); m->d_part[d] = nullptr; m->cap_part[d] = 0; }
                        // results (48 floats) and the share's X0 (6 floats) per pair
                        if (hipMalloc(reinterpret_cast<void**>(&m->d_part[d]), sizeof(float) * 54 * (size_t)np) != hipSuccess) { st[d] = ICET_ERR_NOMEM; return; }
                        m->cap_part[d] = np;
                    }
                    float* part = m->d_part[d]; float* px0 = part + 48 * (size_t)m->cap_part[d];
                    hipError_t e = hipSuccess;
                    // this device's rows of X0: rows d, d + D, ... of the buffer on device 0 (a strided peer copy; same device when d == 0)
                    if (d_x0) e = hipMemcpy2DAsync(px0, 6 * sizeof(float), d_x0 + 6 * (size_t)d, 6 * sizeof(float) * D, 6 * sizeof(float), np, hipMemcpyDefault, s);
                    if (e != hipSuccess) { herr[d] = hipGetErrorString(e); st[d] = ICET_ERR_HIP; return; }
                    st[d] = icet_solve_batch_device(m->ctx[d], p, np, a.data(), b.data(), d_x0 ? px0 : nullptr, part);
                    if (st[d] != ICET_OK) return;
                    e = hipMemcpy2DAsync(d_out + 48 * (size_t)d, 48 * sizeof(float) * D, part, 48 * sizeof(float), 48 * sizeof(float), np, hipMemcpyDefault, s);
                    if (e == hipSuccess) e = hipStreamSynchronize(s);
                    if (e != hipSuccess) { herr[d] = hipGetErrorString(e); st[d] = ICET_ERR_HIP; }
                } catch (const std::bad_alloc&) { st[d] = ICET_ERR_NOMEM; }
            });
        }
        for (auto& t : th) t.join();
    } catch (...) { m->err = "cannot start a host thread per device"; return ICET_ERR_NOMEM; }
    for (int d = 0; d < D; d++) if (st[d] != ICET_OK) {
        if (!herr[d].empty()) m->err = "gather (device " + std::to_string(m->dev[d]) + "): " + herr[d]; else set_err(m, d, m->ctx[d], "icet_solve_batch_device");
        return st[d];
    }
    return ICET_OK;
}

}  // extern "C"
