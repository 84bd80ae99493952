// rtd_engine.hip — host side of the MI355X dose engine and its C ABI (include/rtd.h).
//
// Replaces the orchestration of cudaWrapperProtons (reference src/kernel_wrapper.cu:381-1369):
//   * one handle = one device + one stream + resident CT and LUTs (the reference re-uploads per call, :418-537);
//   * one field object = host geometry of a beam (:612-663) + a workspace allocated once (the reference does
//     ~20 cudaMalloc/cudaFree per beam, :685-734, :1265-1281);
//   * rtd_field_compute = a fixed sequence of asynchronous launches; every scalar the reference reads back
//     to the host between kernels (:783,787,790,954,963) stays in device memory (k_plan, k_ks_plan).
// There is no CPU fallback: without a HIP device every entry point fails with RTD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rtd.h"
#include "rtd_geometry.hpp"
#include "rtd_kernels.hpp"
#include "rtd_sweep.hpp"
#include "rtd_sweep_big.hpp"
#include "rtd_uniform.hpp"

using namespace rtd;

namespace {

thread_local std::string g_globalError;

// Lane direction of the tracer and of the dose transfer: the kernels that lay their lanes along the beam / along another dose
// axis carry a fixed cost (LDS transposition, longer position prefix), so they take over only once the memory axis moves this
// much faster along their direction than along the plain one. Measured crossovers on the 512^3 field: tracer ~35 degrees
// (0.130 vs 0.135 ms at 30, 0.166 vs 0.128 at 45), transfer ~38 degrees (0.112 vs 0.125 at 30, 0.142 vs 0.130 at 45).
constexpr float kTransferAxisRatio = 0.8f;

struct rtd_field_impl;

struct rtd_handle_impl {
    std::vector<rtd_field_impl*> fieldCache;   // released field objects whose device workspace the next field of the same shape takes over
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    rtd_options opt{};
    int numCUs = 256;             // compute units of the device (grid size of the grid-stride kernels)
    bool scanLdsSet = false;      // dynamic-LDS cap of k_trace_scan raised (once per handle)
    size_t traceTLds = 0;         // dynamic-LDS cap set for k_trace_sample_t so far
    bool uni4LdsSet = false;      // ... for k_superpose_uniform4
    bool sweepLdsSet = false;     // ... for k_superpose_sweep
    size_t traceDLds = 0;         // ... for k_trace_sample_d
    bool sweepBigLdsSet = false;  // ... for k_superpose_sweep_big
    unsigned inputEpoch = 0;      // bumped whenever CT, LUTs or options change (fields re-test what they learned about their input)
    // LUTs
    bool haveLuts = false;
    std::vector<float> energiesPerU, peakDepths, scaleFacts;
    float densityScale = 0, spScale = 0, rrlScale = 0;
    float* dLutBlock = nullptr;   // one allocation: cidd | density | sp | rrl (| nuclear weight | nuclear sigma^2)
    size_t lutBlockFloats = 0;
    LutView lut{};
    // CT
    const float* dCt = nullptr;
    float* dCtOwned = nullptr;
    size_t ctOwnedVoxels = 0;     // size of dCtOwned: a CT of the same size is uploaded in place (no free + malloc of the volume)
    uint32_t ctDims[3] = {0, 0, 0};
    const float* ctHost = nullptr;                 // rtd_set_ct_deferred: the caller's volume, uploaded box by box as fields need it
    struct CtBox { std::array<int, 6> box; hipEvent_t done; hipStream_t stream; };
    std::vector<CtBox> ctBoxes;                    // boxes of ctHost already on the device (x0, y0, z0, x1, y1, z1 inclusive), each with the
                                                   // event of its upload and the stream it was issued on (a consumer on another stream waits for it)
    void clearCtBoxes() { for (auto& b : ctBoxes) if (b.done) (void)hipEventDestroy(b.done); ctBoxes.clear(); }
};

// what the device allocations of a field depend on: a released workspace is reused by a field with the same signature
// (W and H separately, not only R = W * H: the padded BEV cube is (W + 64) x (H + 64) x S and the superposition's hand-off slots and
//  node counters scale with ceil(bevW / 64) * ceil(bevH / 32) — a 64 x 32 and a 32 x 64 ray grid need different sizes)
struct AllocSig {
    size_t W = 0, H = 0, S = 0, L = 0, nSpot = 0, nInterm = 0, G = 0, Gs = 0, tileRadWords = 0;
    bool operator==(const AllocSig& o) const {
        return W == o.W && H == o.H && S == o.S && L == o.L && nSpot == o.nSpot && nInterm == o.nInterm && G == o.G && Gs == o.Gs &&
               tileRadWords == o.tileRadWords;
    }
};

struct rtd_field_impl {
    AllocSig sig;
    FieldConst fc{};
    TracerParams tracer{};
    FillGeom fillGeom{};
    FromFan rayIdxToDoseIdx{};
    TransferParams transfer0{};
    int traceMode = 0;              // tracer: 0 lanes across the rays, 1 along the beam (CT x runs along it), 2 along diagonals of (ray, step) (oblique beams)
    int traceDiagB = 0;             // ... mode 2: steps per ray along a diagonal
    int transferMode = 0;           // transfer kernel: lanes of the BEV gathers along dose x (0), y (1) or z (2)
    uint32_t doseDims[3] = {0, 0, 0};
    size_t R = 0;
    // device workspace
    float *dSpotWeights = nullptr, *dConvInterm = nullptr, *dRayWeights = nullptr;
    float *dDensity = nullptr, *dWepl = nullptr, *dRrl = nullptr, *dIdd = nullptr, *dRSigma = nullptr, *dBev = nullptr, *dBevPart = nullptr;
    int* dNodeCount = nullptr;   // [output tile][step][32] arrival counters of the superposition's reduction tree (all zero between launches)
    int *dFirstInside = nullptr, *dFirstOutside = nullptr, *dFirstPassive = nullptr, *dWeplMin = nullptr;
    float* dBlockWeplMin = nullptr;   // [R/64][S] per scan block and step: smallest WEPL of the block's 64 rays
    KsPlanArgs* dKsArgs = nullptr;    // the plan's arguments for a launch that plans for itself (k_superpose_sweep<true>): written at each such launch
    float* dSegPos = nullptr;         // [S / kTraceSeg + 1][3][R] sample positions at the segment boundaries of k_trace_sample (walked once, at creation)
    unsigned char* dTileRad = nullptr;
    size_t tileRadWords = 0;
    LayerPlan* dLayers = nullptr;
    float* dStepTab = nullptr;
    int* dActive = nullptr;      // [L][S][4] minima of (x, y, -x, -y) over rays with dose > 0
    unsigned int *dSigMin = nullptr, *dSigMax = nullptr;   // [L][S] bits of the smallest / largest tile-uniform sigma^2 (uniform-sigma detection)
    bool uniformEligible = false; // the separable superposition may take the field (no nuclear halo, BEV height within its accumulators)
    int uniformHint = -1;         // what the last finished compute found: 0 heterogeneous, 1 one sigma per slice, -1 unknown
    unsigned hintEpoch = 0;       // ... under this handle->inputEpoch
    unsigned launchEpoch = 0;     // handle->inputEpoch when the compute in flight was launched (what its findings are valid for)
    bool launchedKnownUniform = false;   // the compute in flight skipped the general superposition kernel on the strength of the hint
    bool triedUniform = false;    // the compute in flight ran the detection
    // NUCLEAR_CORR (default off): the halo on the spot-resolution grid
    int* dNucSpotIdx = nullptr; float *dNucRayWeights = nullptr, *dNucIdd = nullptr, *dNucRs = nullptr, *dNucBev = nullptr;
    int* dNucEffT = nullptr;
    FieldState* dStateNuc = nullptr;
    FromFan nucIdxToDoseIdx{};
    TransferParams transfer0Nuc{};
    int transferModeNuc = 0;
    long long* dFillDbg = nullptr; size_t fillDbgN = 0;   // RTD_FILL_DEBUG: per-block clock stamps of k_fill (diagnostics)
    long long* dUniDbg = nullptr; size_t uniDbgN = 0;     // RTD_UNIFORM_DEBUG: per-block clock stamps of k_superpose_uniform4 (diagnostics)
    long long* dSweepDbg = nullptr; size_t sweepDbgN = 0; // RTD_SWEEP_DEBUG: per-block clock stamps of k_superpose_sweep (diagnostics)
    long long* dSweepBigDbg = nullptr; size_t sweepBigDbgN = 0; // ... and of k_superpose_sweep_big
    long long* dScanDbg = nullptr; size_t scanDbgN = 0;         // RTD_SCAN_DEBUG: ... of k_trace_scan
    FieldState* dState = nullptr;
    FieldState* hState = nullptr;      // pinned host mirror of *dState (written by k_ks_plan), and its device-side address
    FieldState* dHostState = nullptr;
    std::vector<LayerPlan> hLayers;
    hipEvent_t ev[9] = {};       // 0..6 stage ends, 7 / 8 stop / start of k_superpose_mfma
    bool selfPlanned = false;    // the last compute had no k_ks_plan launch: block 0 of k_superpose_sweep's launch was the plan (ev[4] not recorded)
    bool computed = false;       // the BEV dose and the state record of the last rtd_field_compute[_bev] exist (or a slab is attached)
    bool transferred = false;    // a transfer has been launched since (ev[6] is recorded)
    bool remote = false;         // geometry only: the BEV slab comes from another GPU (rtd_field_attach_bev)
    const unsigned char* attached = nullptr;   // remote: the packed message [FieldState | slab]
    int ksGroups = 14;   // layer groups of the superposition (partial BEV buffers); RTD_KS_GROUPS overrides
    // k_superpose_sweep (rtd_sweep.hpp): layer groups, patches of the ray grid, partial tiles [step][patch][group][96 x 96], arrival counters [step]
    int swGroups = 4, swPX = 1, swPY = 1;
    float* dSwSlots = nullptr; int* dSwCount = nullptr;
    // k_superpose_sweep_big (rtd_sweep_big.hpp), the sources of batch radius 17 .. 32: its own layer groups, partial tiles [step][patch][group][128 x 128], counters
    int bgGroups = kBgMaxGroups;
    float* dSwSlotsBig = nullptr; int* dSwCountBig = nullptr;
    int radiusHint = -1;          // largest batch radius the last finished compute found (-1 unknown), under hintEpoch like uniformHint
    bool sweepEnabled = true;     // RTD_NO_SWEEP: every field through k_superpose_mfma
};

#define RTD_HIP(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof buf_, "HIP error: %s %s %d", hipGetErrorString(e_), __FILE__, __LINE__); \
            (h)->error = buf_;                                                                   \
            return RTD_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

int fail(rtd_handle_impl* h, int code, const std::string& msg) { h->error = msg; return code; }

Affine toAffine(const rtd_affine& a) {
    Affine r;
    r.m.r0 = v3(a.m[0], a.m[1], a.m[2]); r.m.r1 = v3(a.m[3], a.m[4], a.m[5]); r.m.r2 = v3(a.m[6], a.m[7], a.m[8]);
    r.v = v3(a.v[0], a.v[1], a.v[2]);
    return r;
}
IdxTransform toIdx(const rtd_idx_transform& t) {
    IdxTransform r; r.delta = v3(t.delta[0], t.delta[1], t.delta[2]); r.offset = v3(t.offset[0], t.offset[1], t.offset[2]);
    return r;
}

template <typename T>
int devAlloc(rtd_handle_impl* h, T** p, size_t n) {
    RTD_HIP(h, hipMalloc((void**)p, n * sizeof(T)));
    return RTD_OK;
}

// LUT text layout of the reference (energy_reader.cpp:12-101): "N scale" header then N values.
bool readTokens(const std::string& path, std::vector<double>& out) {
    std::ifstream f(path.c_str());
    if (!f) return false;
    double v;
    while (f >> v) out.push_back(v);
    return true;
}

void fillInfo(const rtd_field_impl* f, const FieldState& st, rtd_field_info* info) {
    std::memset(info, 0, sizeof *info);
    info->ray_dims[0] = f->fc.W; info->ray_dims[1] = f->fc.H; info->ray_dims[2] = f->fc.L;
    for (int i = 0; i < 3; ++i) { info->ray_offset[i] = f->fc.rayOffset[i]; info->ray_res[i] = f->fc.rayRes[i]; }
    info->beam_first_inside = st.beamFirstInside; info->beam_first_outside = st.beamFirstOutside;
    info->beam_first_guaranteed_passive = st.firstGuaranteedPassive;
    info->beam_first_calculated_passive = st.firstCalculatedPassive;
    for (int i = 0; i < 3; ++i) { info->bbox_min[i] = st.bboxMin[i]; info->bbox_max[i] = st.bboxMax[i]; }
    for (int i = 0; i < 3; ++i) { info->dose_box_min[i] = st.tboxMin[i]; info->dose_box_max[i] = st.tboxMax[i]; }
    // NUCLEAR_CORR: the halo's slice reaches further sideways than the primary's box and its own box lives on the device only:
    // report the whole grid (callers that move only the box across PCIe then move everything, as the reference does)
    if (f->fc.nuclearCorr && !f->remote) for (int i = 0; i < 3; ++i) { info->dose_box_min[i] = 0; info->dose_box_max[i] = (int32_t)f->doseDims[i] - 1; }
    info->live_steps = st.liveSteps; info->max_radius = st.maxRadius;
    info->uniform_sigma = st.uniformField;
}

}  // namespace

// Launch with optional start / stop events taken from the kernel's own dispatch timestamps (hipExtLaunchKernelGGL): no
// event packets between kernels. (Measured alternative: plain launches bracketed by hipEventRecord, +25 us per field.)
template <typename K, typename... Args>
static void launchK(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipEvent_t startEv, hipEvent_t stopEv, Args... args) {
    hipExtLaunchKernelGGL(kernel, grid, block, lds, s, startEv, stopEv, 0, args...);
}

extern "C" {

uint32_t rtd_abi_version(void) { return RTD_ABI_VERSION; }

void rtd_default_options(rtd_options* o) {   // CMakeLists.txt:36-79
    std::memset(o, 0, sizeof *o);
    o->dose_to_water = 1; o->nozzle = 1;
    o->bp_depth_cutoff = 1.05f; o->conv_sigma_cutoff = 3.0f; o->ks_sigma_cutoff = 3.0f; o->ray_weight_cutoff = 1.0f;
    o->fine_grained_timing = 0;
}

const char* rtd_global_error(void) { return g_globalError.c_str(); }

int rtd_create(int device_id, rtd_handle* out) {
    if (!out) return RTD_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_globalError = "no HIP device available: the dose engine has no CPU fallback";
        return RTD_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) { g_globalError = "device id out of range"; return RTD_ERR_INVALID_ARG; }
    auto* h = new rtd_handle_impl();
    h->device = device_id;
    rtd_default_options(&h->opt);
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking) != hipSuccess) {
        g_globalError = "hipSetDevice / hipStreamCreate failed";
        delete h;
        return RTD_ERR_HIP;
    }
    h->stream = h->ownStream;
    if (hipDeviceGetAttribute(&h->numCUs, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || h->numCUs <= 0) h->numCUs = 256;
    *out = reinterpret_cast<rtd_handle>(h);
    return RTD_OK;
}

int rtd_destroy(rtd_handle hh) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    while (!h->fieldCache.empty()) { rtd_field_impl* c = h->fieldCache.back(); h->fieldCache.pop_back(); rtd_field_destroy(hh, reinterpret_cast<rtd_field>(c)); }
    h->clearCtBoxes();
    if (h->dLutBlock) (void)hipFree(h->dLutBlock);
    if (h->dCtOwned) (void)hipFree(h->dCtOwned);
    if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
    delete h;
    return RTD_OK;
}

const char* rtd_last_error(rtd_handle hh) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    return h ? h->error.c_str() : "null handle";
}

int rtd_set_options(rtd_handle hh, const rtd_options* opt) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (h) ++h->inputEpoch;
    if (!h || !opt) return RTD_ERR_INVALID_ARG;
    h->opt = *opt;
    return RTD_OK;
}

void* rtd_stream(rtd_handle hh) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    return h ? (void*)h->stream : nullptr;
}

int rtd_set_stream(rtd_handle hh, void* s) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    h->stream = s ? (hipStream_t)s : h->ownStream;
    return RTD_OK;
}

int rtd_sync(rtd_handle hh) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipStreamSynchronize(h->stream));
    return RTD_OK;
}

int rtd_host_register(void* p, size_t bytes) {   // host_image_3d.cuh:23-32
    if (!p || !bytes) return RTD_ERR_INVALID_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); g_globalError = "hipHostRegister failed"; return RTD_ERR_HIP; }
    return RTD_OK;
}
int rtd_host_unregister(void* p) {               // host_image_3d.cuh:45-48
    if (!p) return RTD_ERR_INVALID_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); g_globalError = "hipHostUnregister failed"; return RTD_ERR_HIP; }
    return RTD_OK;
}

int rtd_device_alloc(rtd_handle hh, size_t bytes, void** p) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h || !p) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipSetDevice(h->device));
    RTD_HIP(h, hipMalloc(p, bytes));
    return RTD_OK;
}
int rtd_device_free(rtd_handle hh, void* p) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipFree(p));
    return RTD_OK;
}
int rtd_device_zero(rtd_handle hh, void* p, size_t bytes) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipMemsetAsync(p, 0, bytes, h->stream));
    return RTD_OK;
}
int rtd_copy_to_device(rtd_handle hh, void* d, const void* s, size_t bytes) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, h->stream));
    RTD_HIP(h, hipStreamSynchronize(h->stream));
    return RTD_OK;
}
int rtd_copy_to_host(rtd_handle hh, void* d, const void* s, size_t bytes) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, h->stream));
    RTD_HIP(h, hipStreamSynchronize(h->stream));
    return RTD_OK;
}

int rtd_set_luts(rtd_handle hh, const rtd_luts* l) {   // kernel_wrapper.cu:453-537
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (h) ++h->inputEpoch;
    if (!h || !l) return RTD_ERR_INVALID_ARG;
    if (l->n_energies <= 0 || l->n_energy_samples <= 0 || l->n_density_samples <= 0 || l->n_sp_samples <= 0 ||
        l->n_rrl_samples <= 0 || !l->energies_per_u || !l->peak_depths || !l->scale_facts || !l->cidd_matrix ||
        !l->density_vector || !l->sp_vector || !l->rrl_vector)
        return fail(h, RTD_ERR_INVALID_ARG, "rtd_set_luts: empty or null table");
    RTD_HIP(h, hipSetDevice(h->device));
    const size_t nC = (size_t)l->n_energies * l->n_energy_samples, nD = l->n_density_samples, nS = l->n_sp_samples, nR = l->n_rrl_samples;
    const bool nuc = l->nuc_weight_matrix && l->nuc_sq_sigma_matrix;    // NUCLEAR_CORR tables (energy_struct.h:33-36), optional
    const size_t lutFloats = nC + nD + nS + nR + (nuc ? 2 * nC : 0);
    RTD_HIP(h, hipStreamSynchronize(h->stream));                     // no kernel still reads the tables that are replaced
    if (h->dLutBlock && h->lutBlockFloats != lutFloats) { RTD_HIP(h, hipFree(h->dLutBlock)); h->dLutBlock = nullptr; }
    if (!h->dLutBlock) { RTD_HIP(h, hipMalloc((void**)&h->dLutBlock, lutFloats * sizeof(float))); h->lutBlockFloats = lutFloats; }
    float* p = h->dLutBlock;
    RTD_HIP(h, hipMemcpy(p, l->cidd_matrix, nC * 4, hipMemcpyHostToDevice)); h->lut.cidd = p; p += nC;
    RTD_HIP(h, hipMemcpy(p, l->density_vector, nD * 4, hipMemcpyHostToDevice)); h->lut.density = p; p += nD;
    RTD_HIP(h, hipMemcpy(p, l->sp_vector, nS * 4, hipMemcpyHostToDevice)); h->lut.sp = p; p += nS;
    RTD_HIP(h, hipMemcpy(p, l->rrl_vector, nR * 4, hipMemcpyHostToDevice)); h->lut.rrl = p; p += nR;
    h->lut.nucWeight = nullptr; h->lut.nucSqSigma = nullptr;
    if (nuc) {
        RTD_HIP(h, hipMemcpy(p, l->nuc_weight_matrix, nC * 4, hipMemcpyHostToDevice)); h->lut.nucWeight = p; p += nC;
        RTD_HIP(h, hipMemcpy(p, l->nuc_sq_sigma_matrix, nC * 4, hipMemcpyHostToDevice)); h->lut.nucSqSigma = p;
    }
    h->lut.nSamples = l->n_energy_samples; h->lut.nEnergies = l->n_energies;
    h->lut.nDensity = (int)nD; h->lut.nSp = (int)nS; h->lut.nRrl = (int)nR;
    h->energiesPerU.assign(l->energies_per_u, l->energies_per_u + l->n_energies);
    h->peakDepths.assign(l->peak_depths, l->peak_depths + l->n_energies);
    h->scaleFacts.assign(l->scale_facts, l->scale_facts + l->n_energies);
    h->densityScale = l->density_scale_fact; h->spScale = l->sp_scale_fact; h->rrlScale = l->rrl_scale_fact;
    h->haveLuts = true;
    return RTD_OK;
}

int rtd_load_luts_dir(rtd_handle hh, const char* dir, int water_cube_test) {   // energy_reader.cpp:12-101
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h || !dir) return RTD_ERR_INVALID_ARG;
    std::string d(dir);
    if (!d.empty() && d.back() != '/') d += '/';
    std::vector<double> t;
    if (!readTokens(d + "proton_cumul_ddd_data.txt", t) || t.size() < 2)
        return fail(h, RTD_ERR_IO, "Failed to open " + d + "proton_cumul_ddd_data.txt");
    const int nS = (int)t[0], nE = (int)t[1];
    if (nS <= 0 || nE <= 0 || t.size() < 2 + 3 * (size_t)nE + (size_t)nS * nE)
        return fail(h, RTD_ERR_IO, "Truncated " + d + "proton_cumul_ddd_data.txt");
    std::vector<float> e(nE), p(nE), s(nE), m((size_t)nS * nE);
    size_t o = 2;
    for (int i = 0; i < nE; ++i) e[i] = (float)t[o++];
    for (int i = 0; i < nE; ++i) p[i] = (float)t[o++];
    for (int i = 0; i < nE; ++i) s[i] = (float)t[o++];
    for (size_t i = 0; i < m.size(); ++i) m[i] = (float)t[o++];
    auto one = [&](const std::string& name, int& n, float& scale, std::vector<float>& v) -> bool {
        std::vector<double> tt;
        if (!readTokens(d + name, tt) || tt.size() < 2) return false;
        n = (int)tt[0]; scale = (float)tt[1];
        if (n <= 0 || tt.size() < 2 + (size_t)n) return false;
        v.resize(n);
        for (int i = 0; i < n; ++i) v[i] = (float)tt[2 + i];
        return true;
    };
    rtd_luts l{};
    std::vector<float> dv, sv, rv;
    if (!one("density_Schneider2000_adj.txt", l.n_density_samples, l.density_scale_fact, dv))
        return fail(h, RTD_ERR_IO, "Failed to open " + d + "density_Schneider2000_adj.txt");
    if (!one("HU_to_SP_H&N_adj.txt", l.n_sp_samples, l.sp_scale_fact, sv))
        return fail(h, RTD_ERR_IO, "Failed to open " + d + "HU_to_SP_H&N_adj.txt");
    const char* rname = water_cube_test ? "radiation_length_inc_water.txt" : "radiation_length.txt";
    if (!one(rname, l.n_rrl_samples, l.rrl_scale_fact, rv))
        return fail(h, RTD_ERR_IO, "Failed to open " + d + rname);
    // NUCLEAR_CORR: the variant's table, checked against the cumulative-IDD table like the reference does (energy_reader.cpp:103-162)
    std::vector<float> nw, nq;
    if (h->opt.nuclear_corr != RTD_NUC_OFF) {
        const char* nname = h->opt.nuclear_corr == RTD_NUC_SOUKUP ? "nuclear_weights_and_sigmas_Soukup.txt"
                          : h->opt.nuclear_corr == RTD_NUC_FLUKA ? "nuclear_weights_and_sigmas_Fluka.txt" : "nuclear_weights_and_sigmas_fit.txt";
        std::vector<double> tt;
        if (!readTokens(d + nname, tt) || tt.size() < 2) return fail(h, RTD_ERR_IO, "Failed to open " + d + nname);
        if ((int)tt[0] != nS || (int)tt[1] != nE)
            return fail(h, RTD_ERR_IO, std::string("Number of samples or energies in ") + nname + " different from proton_cumul_ddd_data.txt");
        if (tt.size() < 2 + 3 * (size_t)nE + 2 * (size_t)nS * nE) return fail(h, RTD_ERR_IO, "Truncated " + d + nname);
        size_t q = 2;
        const std::vector<float>* axes[3] = { &e, &p, &s };
        const char* what[3] = { "Energies", "Peak depths", "Scale facts" };
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < nE; ++i)
                if (std::fabs((*axes[a])[i] - (float)tt[q++]) > 0.01f)
                    return fail(h, RTD_ERR_IO, std::string(what[a]) + " in " + nname + " different from proton_cumul_ddd_data.txt");
        nw.resize((size_t)nS * nE); nq.resize((size_t)nS * nE);
        for (auto& v : nw) v = (float)tt[q++];
        for (auto& v : nq) v = (float)tt[q++];
        l.nuc_weight_matrix = nw.data(); l.nuc_sq_sigma_matrix = nq.data();
    }
    l.n_energy_samples = nS; l.n_energies = nE;
    l.energies_per_u = e.data(); l.peak_depths = p.data(); l.scale_facts = s.data(); l.cidd_matrix = m.data();
    l.density_vector = dv.data(); l.sp_vector = sv.data(); l.rrl_vector = rv.data();
    return rtd_set_luts(hh, &l);
}

int rtd_set_ct_device(rtd_handle hh, const float* dev, const uint32_t dims[3]) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (h) ++h->inputEpoch;
    if (!h || !dev || !dims || !dims[0] || !dims[1] || !dims[2]) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipSetDevice(h->device));
    // (device-wide: the handle may have had kernels on other streams, rtd_set_stream, that still read the volume)
    if (h->dCtOwned) { RTD_HIP(h, hipDeviceSynchronize()); RTD_HIP(h, hipFree(h->dCtOwned)); h->dCtOwned = nullptr; h->ctOwnedVoxels = 0; }
    h->dCt = dev;
    h->ctHost = nullptr; h->clearCtBoxes();
    std::memcpy(h->ctDims, dims, sizeof h->ctDims);
    return RTD_OK;
}

int rtd_set_ct(rtd_handle hh, const float* host, const uint32_t dims[3]) {   // kernel_wrapper.cu:420-451
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (h) ++h->inputEpoch;
    if (!h || !host || !dims || !dims[0] || !dims[1] || !dims[2]) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipSetDevice(h->device));
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    RTD_HIP(h, hipDeviceSynchronize());                              // no kernel, on any stream the handle has used, still reads the volume that is replaced
    if (h->dCtOwned && h->ctOwnedVoxels != n) { RTD_HIP(h, hipFree(h->dCtOwned)); h->dCtOwned = nullptr; }
    if (!h->dCtOwned) { RTD_HIP(h, hipMalloc((void**)&h->dCtOwned, n * sizeof(float))); h->ctOwnedVoxels = n; }
    RTD_HIP(h, hipMemcpy(h->dCtOwned, host, n * sizeof(float), hipMemcpyHostToDevice));
    h->dCt = h->dCtOwned;
    h->ctHost = nullptr; h->clearCtBoxes();
    std::memcpy(h->ctDims, dims, sizeof h->ctDims);
    return RTD_OK;
}

// rtd_set_ct without the copy: the volume stays with the caller and every field uploads, before its tracer runs, the box of it that
// its rays can sample (ensureCtBox). A beam reads ~10 % of a 512^3 CT; the reference binds the whole volume (:420-451).
int rtd_set_ct_deferred(rtd_handle hh, const float* host, const uint32_t dims[3]) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (h) ++h->inputEpoch;
    if (!h || !host || !dims || !dims[0] || !dims[1] || !dims[2]) return RTD_ERR_INVALID_ARG;
    RTD_HIP(h, hipSetDevice(h->device));
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    RTD_HIP(h, hipDeviceSynchronize());
    if (h->dCtOwned && h->ctOwnedVoxels != n) { RTD_HIP(h, hipFree(h->dCtOwned)); h->dCtOwned = nullptr; }
    if (!h->dCtOwned) { RTD_HIP(h, hipMalloc((void**)&h->dCtOwned, n * sizeof(float))); h->ctOwnedVoxels = n; }
    h->dCt = h->dCtOwned;
    h->ctHost = host; h->clearCtBoxes();
    std::memcpy(h->ctDims, dims, sizeof h->ctDims);
    return RTD_OK;
}

int rtd_field_destroy(rtd_handle hh, rtd_field ff) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f) return RTD_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = { f->dSpotWeights, f->dConvInterm, f->dRayWeights, f->dDensity, f->dWepl, f->dRrl, f->dIdd, f->dRSigma, f->dBev, f->dBevPart, f->dNodeCount, f->dSwSlots, f->dSwCount, f->dSwSlotsBig, f->dSwCountBig,
                     f->dFirstInside, f->dFirstOutside, f->dFirstPassive, f->dWeplMin, f->dBlockWeplMin, f->dSegPos, f->dKsArgs, f->dTileRad,
                     f->dLayers, f->dState, f->dStepTab, f->dActive, f->dSigMin, f->dSigMax, f->dFillDbg, f->dSweepDbg, f->dSweepBigDbg, f->dScanDbg, f->dUniDbg,
                     f->dNucSpotIdx, f->dNucRayWeights, f->dNucIdd, f->dNucRs, f->dNucBev, f->dNucEffT, f->dStateNuc };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (f->hState) (void)hipHostFree(f->hState);
    for (auto& e : f->ev) if (e) (void)hipEventDestroy(e);
    delete f;
    return RTD_OK;
}

// Gives the field's device workspace back to the handle: the next rtd_field_create of the same shape (ray grid, steps, layers,
// spot map) takes it over instead of allocating (the reference mallocs and frees ~20 buffers per beam, :685-734, :1265-1281).
int rtd_field_release(rtd_handle hh, rtd_field ff) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f) return RTD_ERR_INVALID_ARG;
    if (f->remote || f->fc.nuclearCorr || h->fieldCache.size() >= 4) return rtd_field_destroy(hh, ff);
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);      // its kernels have drained: the next owner uploads with plain copies
    f->computed = false; f->transferred = false;
    h->fieldCache.push_back(f);
    return RTD_OK;
}

// Host geometry of one beam (kernel_wrapper.cu:612-663, 829-838) + workspace allocation + spot-weight upload (:851).
// remote: geometry only — the field's BEV slab is computed on another GPU and attached (rtd_field_attach_bev).
static int createField(rtd_handle hh, const rtd_beam* b, const uint32_t dose_dims[3], bool remote, rtd_field* out) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h || !b || !dose_dims || !out) return RTD_ERR_INVALID_ARG;
    *out = nullptr;
    if (!remote && (!h->haveLuts || !h->dCt)) return fail(h, RTD_ERR_NOT_READY, "rtd_field_create: set LUTs and CT first");
    if (b->n_layers == 0) return fail(h, RTD_ERR_INVALID_ARG, "Empty list");   // findMax on an empty vector, vector_find.h:24
    if (!b->spot_weights || !b->energies || !b->spot_sigmas || b->spot_nx == 0 || b->spot_ny == 0 || b->tracer_steps == 0)
        return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_create: null or empty beam field");
    if (!(b->ray_spacing[0] > 0.0f) || !(b->ray_spacing[1] > 0.0f) || !(b->spot_idx_to_gantry.delta[0] > 0.0f) ||
        !(b->spot_idx_to_gantry.delta[1] > 0.0f) || b->spot_idx_to_gantry.delta[2] == 0.0f)
        return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_create: ray spacing and spot pitch must be positive, step length non-zero");
    RTD_HIP(h, hipSetDevice(h->device));
    const rtd_options& opt = h->opt;
    const int L = (int)b->n_layers, S = (int)b->tracer_steps;

    float maxSx = b->spot_sigmas[0], maxSy = b->spot_sigmas[1];
    for (int i = 1; i < L; ++i) { maxSx = std::max(maxSx, b->spot_sigmas[2 * i]); maxSy = std::max(maxSy, b->spot_sigmas[2 * i + 1]); }
    const IdxTransform sitg = toIdx(b->spot_idx_to_gantry);
    const Vec3 res = v3(b->ray_spacing[0], b->ray_spacing[1], sitg.delta.z);                                  // :623
    const float cc = opt.conv_sigma_cutoff;
    const int lSteps = (int)std::ceil((sitg.offset.x - (cc * maxSx + 0.5f * res.x)) / res.x);                  // :650-653
    const int bSteps = (int)std::ceil((sitg.offset.y - (cc * maxSy + 0.5f * res.y)) / res.y);
    const int rSteps = (int)std::floor(((float)(b->spot_nx - 1) * sitg.delta.x + sitg.offset.x + (cc * maxSx + 0.5f * res.x)) / res.x);
    const int tSteps = (int)std::floor(((float)(b->spot_ny - 1) * sitg.delta.y + sitg.offset.y + (cc * maxSy + 0.5f * res.y)) / res.y);
    const Vec3 off = v3(res.x * (float)lSteps, res.y * (float)bSteps, sitg.offset.z);                          // :654
    const int W = roundTo(rSteps - lSteps + 1, kSuperpTileX), H = roundTo(tSteps - bSteps + 1, kSuperpTileY);   // :659
    if (W <= 0 || H <= 0) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_create: empty ray grid");
    if (W > 4095 || H > 4095) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_create: ray grid larger than 4095 x 4095");
    const int tilesX = W / kSuperpTileX, tilesY = H / kSuperpTileY;
    if (L > kMaxLayers || S > kMaxSteps)
        return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_create: more than 256 layers or 4096 steps");

    auto* f = new rtd_field_impl();
    f->remote = remote;
    FieldConst& fc = f->fc;
    fc.W = W; fc.H = H; fc.L = L; fc.S = S; fc.bevW = W + 2 * kMaxSuperpR; fc.bevH = H + 2 * kMaxSuperpR;
    fc.tilesX = tilesX; fc.tilesY = tilesY;
    fc.rayRes[0] = res.x; fc.rayRes[1] = res.y; fc.rayRes[2] = res.z;
    fc.rayOffset[0] = off.x; fc.rayOffset[1] = off.y; fc.rayOffset[2] = off.z;
    fc.sourceDist[0] = b->source_dist[0]; fc.sourceDist[1] = b->source_dist[1];
    fc.spotNx = (int)b->spot_nx; fc.spotNy = (int)b->spot_ny;
    fc.spotDelta[0] = sitg.delta.x; fc.spotDelta[1] = sitg.delta.y; fc.spotDelta[2] = sitg.delta.z;
    fc.spotOffset[0] = sitg.offset.x; fc.spotOffset[1] = sitg.offset.y; fc.spotOffset[2] = sitg.offset.z;
    fc.bpDepthCutoff = opt.bp_depth_cutoff; fc.convSigmaCutoff = opt.conv_sigma_cutoff;
    fc.ksSigmaCutoff = opt.ks_sigma_cutoff; fc.rayWeightCutoff = opt.ray_weight_cutoff;
    fc.doseToWater = opt.dose_to_water; fc.nozzle = opt.nozzle;
    fc.nuclearCorr = remote ? 0 : opt.nuclear_corr;
    f->uniformEligible = !remote && !fc.nuclearCorr && fc.bevH <= kUniMaxBevH && L <= 256 && std::getenv("RTD_NO_UNIFORM_PATH") == nullptr;
    fc.nucW = fc.nuclearCorr ? roundTo((int)b->spot_nx, kSuperpTileX) : 0;                                     // :667
    fc.nucH = fc.nuclearCorr ? roundTo((int)b->spot_ny, kSuperpTileY) : 0;
    fc.spotDist = sitg.delta.x / b->ray_spacing[0];                                                            // spotDistInRays, :922
    if (fc.nuclearCorr && (!h->lut.nucWeight || !h->lut.nucSqSigma)) {
        delete f;
        return fail(h, RTD_ERR_INVALID_ARG, "nuclear_corr is set but the LUTs carry no nuclear tables");
    }
    std::memcpy(f->doseDims, dose_dims, sizeof f->doseDims);
    f->R = (size_t)W * H;

    IdxTransform primRayIdxToGantry; primRayIdxToGantry.delta = res; primRayIdxToGantry.offset = off;          // :656
    FromFan rayIdxToImIdx; rayIdxToImIdx.fitf = primRayIdxToGantry; rayIdxToImIdx.gtii = toAffine(b->gantry_to_im_idx);
    rayIdxToImIdx.dist.x = b->source_dist[0]; rayIdxToImIdx.dist.y = b->source_dist[1];                        // :657
    f->tracer = makeTracerParams(h->densityScale, h->spScale, (unsigned int)S, rayIdxToImIdx);                 // :766
    {
        // Three sampling kernels, by the lanes' direction: across the rays (k_trace_sample), along the beam (k_trace_sample_t), along
        // the diagonal (ray + j, step + b j) of the (ray, step) plane whose samples stay closest to one CT slice (k_trace_sample_d).
        // What decides is the drift across CT slices (and, weakly, rows) per lane — rays: coefIdxI.z, steps: coefOffset.z * delta.z,
        // a diagonal: their sum with b — with the kernels' measured costs on the 512^3 bench field (us, tracer stage):
        //   across  57 + 65 drift   (0 deg 57, 20 deg 97, 30 deg 123, 45 deg 149)
        //   along   91 + 22 drift   (90 deg 91, 75 deg 105, 60 deg 116, 45 deg 121)
        //   diagonal 79 + 11 drift  (45 deg 79, 30 deg 88, 20 deg 94, 0 deg 101; 60 deg 87, 80 deg 97)
        {
            const float rayZ = f->tracer.coefIdxI.z, stepZ = f->tracer.coefOffset.z * f->tracer.delta.z;
            const float rayY = f->tracer.coefIdxI.y, stepY = f->tracer.coefOffset.y * f->tracer.delta.z;
            auto drift = [&](float rz, float ry) { return std::fabs(rz) + 0.05f * std::fabs(ry); };
            int bestB = 0; float bestD = 1e30f;
            for (int bb = -3; bb <= 3; ++bb) {
                if (bb == 0) continue;
                const float d = drift(rayZ + bb * stepZ, rayY + bb * stepY);
                if (d < bestD) { bestD = d; bestB = bb; }
            }
            f->traceDiagB = bestB;
            const float costAcross = 57.0f + 65.0f * drift(rayZ, rayY), costAlong = 91.0f + 22.0f * drift(stepZ, stepY);
            const float costDiag = (W % kTdRays == 0) ? 79.0f + 11.0f * bestD : 1e30f;
            f->traceMode = costAcross <= costAlong && costAcross <= costDiag ? 0 : (costAlong <= costDiag ? 1 : 2);
        }
        if (const char* v = std::getenv("RTD_TRACE_MODE")) f->traceMode = std::atoi(v);   // diagnostics: force the plain (0) / along-beam (1) / diagonal (2) sampling kernel
        if (const char* v = std::getenv("RTD_TRACE_DIAG_B")) f->traceDiagB = std::atoi(v);
        if (f->traceMode == 2 && (f->traceDiagB == 0 || W % kTdRays != 0)) f->traceMode = 0;
    }
    f->fillGeom = makeFillGeom(h->rrlScale, rayIdxToImIdx);                                                    // :925
    f->rayIdxToDoseIdx = rayIdxToImIdx; f->rayIdxToDoseIdx.gtii = toAffine(b->gantry_to_dose_idx);             // :1185
    f->transfer0 = makeTransferParams(invertAndShift(f->rayIdxToDoseIdx, v3((float)kMaxSuperpR, (float)kMaxSuperpR, 0.0f)));  // :1213 (z shift on device)
    {
        const float ax = std::fabs(f->transfer0.coefIdxI.x), ay = std::fabs(f->transfer0.coefIdxJ.x), az = std::fabs(f->transfer0.inc.x);
        const int other = ay >= az ? 1 : 2;                           // the axis besides x that moves fastest along BEV x
        const float ao = std::max(ay, az);
        f->transferMode = ao > kTransferAxisRatio * ax ? other : 0;
    }

    if (remote) {
        hipError_t e = hipEventCreate(&f->ev[0]);
        if (e == hipSuccess) e = hipEventCreate(&f->ev[6]);
        if (e != hipSuccess) { h->error = std::string("HIP error: ") + hipGetErrorString(e); rtd_field_destroy(hh, reinterpret_cast<rtd_field>(f)); return RTD_ERR_HIP; }
        *out = reinterpret_cast<rtd_field>(f);
        return RTD_OK;
    }
    // per-layer beam-model tables (:792-794, :829-838)
    const int nE = (int)h->energiesPerU.size();
    float maxEnergy = b->energies[0];
    for (int i = 1; i < L; ++i) maxEnergy = std::max(maxEnergy, b->energies[i]);
    fc.maxPeakDepth = vectorInterpolate(h->peakDepths.data(), nE, findDecimalOrdered(h->energiesPerU.data(), nE, maxEnergy));
    f->hLayers.resize(L);
    for (int l = 0; l < L; ++l) {
        LayerPlan& p = f->hLayers[l];
        std::memset(&p, 0, sizeof p);
        p.energyIdx = findDecimalOrdered(h->energiesPerU.data(), nE, b->energies[l]);
        p.energyScaleFact = vectorInterpolate(h->scaleFacts.data(), nE, p.energyIdx);
        p.peakDepth = vectorInterpolate(h->peakDepths.data(), nE, p.energyIdx);
        p.spotSigmaX = b->spot_sigmas[2 * l]; p.spotSigmaY = b->spot_sigmas[2 * l + 1];
        Vec2 c = sigmaSqAirCoefs(p.peakDepth, opt.nozzle);
        p.airCoefA = c.x; p.airCoefB = c.y;
        const float relStepLenSq = 1.0f;                                                                       // fill_idd_and_sigma_params.cu:28-40
        p.sigmaSqAirQuad = c.x * relStepLenSq * res.z * res.z;
        p.sigmaSqAirLin = 2.0f * c.x * relStepLenSq * res.z * off.z + c.y * res.z;
        for (int i = 0; i < kMaxSuperpR + 2; ++i) p.effRad[i] = i;
    }

    if (const char* v = std::getenv("RTD_KS_GROUPS")) f->ksGroups = std::max(1, std::min(kKsMaxGroups, std::atoi(v)));
    f->ksGroups = std::min(f->ksGroups, L);
    {   // keep the partial BEV buffers below ~4 GiB for large ray grids (G only trades parallelism for memory)
        const size_t sliceBytes = (size_t)fc.bevW * fc.bevH * (size_t)S * sizeof(float);
        const size_t cap = (size_t)4 << 30;
        f->ksGroups = (int)std::max<size_t>(1, std::min<size_t>((size_t)f->ksGroups, cap / std::max<size_t>(sliceBytes, 1)));
    }
    // workspace (the reference's per-beam cudaMallocs, :685-734, :804-808): taken over from a released field of the same shape
    // when there is one (rtd_field_release), so a plan of similar beams allocates once
    const size_t R = f->R, P = (size_t)fc.bevW * fc.bevH;
    const size_t nSpot = (size_t)b->spot_nx * b->spot_ny * L;
    f->tileRadWords = ((size_t)L * S * tilesX * tilesY + 3) / 4;      // filled as 32-bit words by k_reset
    f->sig.W = (size_t)W; f->sig.H = (size_t)H; f->sig.S = (size_t)S; f->sig.L = (size_t)L; f->sig.nSpot = nSpot; f->sig.nInterm = (size_t)W * b->spot_ny * L;
    f->sig.G = (size_t)f->ksGroups; f->sig.tileRadWords = f->tileRadWords;
    // sweep: 4 layer groups unless told otherwise; at most 64 layers per group; partial tiles below ~2 GiB
    f->sweepEnabled = std::getenv("RTD_NO_SWEEP") == nullptr;
    if (const char* v = std::getenv("RTD_SW_GROUPS")) f->swGroups = std::atoi(v);
    f->swGroups = std::max(std::max(1, (L + kSwMaxLay - 1) / kSwMaxLay), std::min(std::min(f->swGroups, kSwMaxGroups), L));
    f->swPX = (W + kSwPatch - 1) / kSwPatch; f->swPY = (H + kSwPatchRows - 1) / kSwPatchRows;
    while (f->swGroups > std::max(1, (L + kSwMaxLay - 1) / kSwMaxLay) &&
           (size_t)S * f->swPX * f->swPY * f->swGroups * kSwSlot * sizeof(float) > ((size_t)2 << 30)) --f->swGroups;
    f->sig.Gs = (size_t)f->swGroups;
    f->bgGroups = std::max(1, std::min(kBgMaxGroups, L));
    while (f->bgGroups > std::max(1, (L + kSwMaxLay - 1) / kSwMaxLay) &&       // (a group holds at most kSwMaxLay layers: L <= 256 needs up to 4)
           (size_t)S * f->swPX * f->swPY * f->bgGroups * kBgSlot * sizeof(float) > ((size_t)1 << 30)) --f->bgGroups;
    if (f->sweepEnabled) f->sig.G = 0;      // (the partial BEV buffers of k_superpose_mfma exist only without the sweep)
    rtd_field_impl* husk = nullptr;
    for (size_t i = 0; i < h->fieldCache.size(); ++i)
        if (h->fieldCache[i]->sig == f->sig) { husk = h->fieldCache[i]; h->fieldCache.erase(h->fieldCache.begin() + (long)i); break; }
    if (husk) {
        f->dSpotWeights = husk->dSpotWeights; f->dConvInterm = husk->dConvInterm; f->dRayWeights = husk->dRayWeights;
        f->dDensity = husk->dDensity; f->dWepl = husk->dWepl; f->dRrl = husk->dRrl; f->dIdd = husk->dIdd; f->dRSigma = husk->dRSigma;
        f->dBev = husk->dBev; f->dBevPart = husk->dBevPart; f->dNodeCount = husk->dNodeCount; f->dSwSlots = husk->dSwSlots; f->dSwCount = husk->dSwCount; f->dSwSlotsBig = husk->dSwSlotsBig; f->dSwCountBig = husk->dSwCountBig; f->dFirstInside = husk->dFirstInside; f->dFirstOutside = husk->dFirstOutside;
        f->dFirstPassive = husk->dFirstPassive; f->dWeplMin = husk->dWeplMin; f->dBlockWeplMin = husk->dBlockWeplMin; f->dSegPos = husk->dSegPos; f->dKsArgs = husk->dKsArgs; f->dTileRad = husk->dTileRad; f->dLayers = husk->dLayers;
        f->dState = husk->dState; f->dStepTab = husk->dStepTab; f->dActive = husk->dActive; f->dSigMin = husk->dSigMin; f->dSigMax = husk->dSigMax; f->hState = husk->hState; f->dHostState = husk->dHostState;
        for (int i = 0; i < 9; ++i) f->ev[i] = husk->ev[i];
        delete husk;
    }
    const bool fresh = husk == nullptr;
    int st = RTD_OK;
    auto A = [&](auto** p, size_t n) { if (st == RTD_OK && fresh) st = devAlloc(h, p, n); };
    A(&f->dSpotWeights, nSpot); A(&f->dConvInterm, (size_t)W * b->spot_ny * L); A(&f->dRayWeights, R * L);
    A(&f->dDensity, R * S); A(&f->dWepl, R * S); A(&f->dRrl, R * S); A(&f->dIdd, R * S * L); A(&f->dRSigma, R * S * L); A(&f->dBev, P * S);
    const size_t nOutTiles = (size_t)((fc.bevW + kKsTileX - 1) / kKsTileX) * ((fc.bevH + kKsTileY - 1) / kKsTileY);
    if (!f->sweepEnabled) { A(&f->dBevPart, nOutTiles * kKsTileX * kKsTileY * S * f->ksGroups); A(&f->dNodeCount, nOutTiles * S * 32); }
    else {
        A(&f->dSwSlots, (size_t)S * f->swPX * f->swPY * f->swGroups * kSwSlot); A(&f->dSwCount, (size_t)S);
        A(&f->dSwSlotsBig, (size_t)S * f->swPX * f->swPY * f->bgGroups * kBgSlot); A(&f->dSwCountBig, (size_t)S);
    }
    A(&f->dFirstInside, R); A(&f->dFirstOutside, R); A(&f->dFirstPassive, R * L); A(&f->dWeplMin, (size_t)S); A(&f->dBlockWeplMin, (R / 64) * (size_t)S); A(&f->dSegPos, ((size_t)S / kTraceSeg + 1) * 3 * R); A(&f->dKsArgs, (size_t)1);
    A(&f->dTileRad, f->tileRadWords * 4); A(&f->dLayers, (size_t)L); A(&f->dState, (size_t)1); A(&f->dStepTab, (size_t)2 * S); A(&f->dActive, (size_t)4 * L * S); A(&f->dSigMin, (size_t)L * S); A(&f->dSigMax, (size_t)L * S);
    if (st != RTD_OK) { rtd_field_destroy(hh, reinterpret_cast<rtd_field>(f)); return st; }
    hipError_t e = hipMemcpy(f->dSpotWeights, b->spot_weights, nSpot * sizeof(float), hipMemcpyHostToDevice);   // :851
    if (e == hipSuccess) e = hipMemcpy(f->dLayers, f->hLayers.data(), (size_t)L * sizeof(LayerPlan), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(f->dState, 0, sizeof(FieldState));
    if (e == hipSuccess) {   // the sample positions at the segment boundaries of k_trace_sample: geometry only, walked once
        k_trace_segpos<<<(unsigned)((R + 255) / 256), 256, 0, h->stream>>>(f->tracer, fc.W, (int)R, f->dSegPos);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess && fresh) e = hipHostMalloc((void**)&f->hState, sizeof(FieldState), hipHostMallocMapped);
    if (e == hipSuccess && fresh) e = hipHostGetDevicePointer((void**)&f->dHostState, f->hState, 0);
    if (e == hipSuccess) std::memset(f->hState, 0, sizeof(FieldState));
    if (e == hipSuccess) {   // the plan's arguments as a self-planning sweep launch reads them (constant for the field: such a launch never tries the uniform path)
        const KsPlanArgs ksSelf{f->dState, f->dLayers, f->rayIdxToDoseIdx, f->transfer0, (int)f->doseDims[0], (int)f->doseDims[1], (int)f->doseDims[2],
                                f->ksGroups, f->swGroups, f->dHostState, nullptr, (const unsigned int*)f->dSigMin, (const unsigned int*)f->dSigMax,
                                0, f->sweepEnabled ? kSwMaxR : -1, f->bgGroups};
        e = hipMemcpy(f->dKsArgs, &ksSelf, sizeof ksSelf, hipMemcpyHostToDevice);
    }
    {
        std::vector<float> tab(2 * (size_t)S);
        for (int k = 0; k < S; ++k) {
            Vec2 vw = f->fillGeom.voxelWidth((unsigned)k);
            tab[2 * k] = 0.5f * (vw.x + vw.y);
            tab[2 * k + 1] = f->fillGeom.stepVol((unsigned)k);
        }
        if (e == hipSuccess) e = hipMemcpy(f->dStepTab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (fc.nuclearCorr && e == hipSuccess) {
        // NUCLEAR_CORR set-up (kernel_wrapper.cu:736-751, 858-892): spot index of every ray, padded spot weights, halo buffers
        const size_t nucR = (size_t)fc.nucW * fc.nucH, bevN = (size_t)(fc.nucW + 2 * kMaxSuperpR) * (fc.nucH + 2 * kMaxSuperpR);
        const size_t nTn = (size_t)(fc.nucW / kSuperpTileX) * (fc.nucH / kSuperpTileY);
        std::vector<int> spotIdx(R, -1);
        for (unsigned int sy = 0; sy < b->spot_ny; ++sy) {
            const float gy = (float)sy * sitg.delta.y + sitg.offset.y;
            const int ry = (int)std::round((gy - off.y) / res.y);
            for (unsigned int sx = 0; sx < b->spot_nx; ++sx) {
                const float gx = (float)sx * sitg.delta.x + sitg.offset.x;
                const int rx = (int)std::round((gx - off.x) / res.x);
                if (rx >= 0 && rx < W && ry >= 0 && ry < H) spotIdx[(size_t)W * ry + rx] = fc.nucW * (int)sy + (int)sx;
            }
        }
        std::vector<float> padded(nucR * (size_t)L, 0.0f);               // extendAndPadd, :51-66
        for (int z = 0; z < L; ++z) for (unsigned int y = 0; y < b->spot_ny; ++y) for (unsigned int x = 0; x < b->spot_nx; ++x)
            padded[(size_t)z * nucR + (size_t)y * fc.nucW + x] = b->spot_weights[((size_t)z * b->spot_ny + y) * b->spot_nx + x];
        int stn = RTD_OK;
        auto An = [&](auto** p, size_t n) { if (stn == RTD_OK) stn = devAlloc(h, p, n); };
        An(&f->dNucSpotIdx, R); An(&f->dNucRayWeights, nucR * L); An(&f->dNucIdd, nucR * L); An(&f->dNucRs, nucR * L);
        An(&f->dNucBev, bevN); An(&f->dNucEffT, nTn * L); An(&f->dStateNuc, (size_t)1);
        if (stn != RTD_OK) { rtd_field_destroy(hh, reinterpret_cast<rtd_field>(f)); return stn; }
        e = hipMemcpy(f->dNucSpotIdx, spotIdx.data(), R * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(f->dNucRayWeights, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(f->dStateNuc, 0, sizeof(FieldState));
        // the halo cube lives on the spot grid: its own fan transform (:1221) and transfer parameters (:1245, shift by -entry step = 0)
        f->nucIdxToDoseIdx.fitf = sitg; f->nucIdxToDoseIdx.gtii = toAffine(b->gantry_to_dose_idx);
        f->nucIdxToDoseIdx.dist.x = b->source_dist[0]; f->nucIdxToDoseIdx.dist.y = b->source_dist[1];
        f->transfer0Nuc = makeTransferParams(invertAndShift(f->nucIdxToDoseIdx, v3((float)kMaxSuperpR, (float)kMaxSuperpR, 0.0f)));
        const float ax = std::fabs(f->transfer0Nuc.coefIdxI.x), ay = std::fabs(f->transfer0Nuc.coefIdxJ.x), az = std::fabs(f->transfer0Nuc.inc.x);
        f->transferModeNuc = std::max(ay, az) > kTransferAxisRatio * ax ? (ay >= az ? 1 : 2) : 0;
    }
    if (fresh) for (auto& ev : f->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e != hipSuccess) { h->error = std::string("HIP error (field set-up): ") + hipGetErrorString(e); rtd_field_destroy(hh, reinterpret_cast<rtd_field>(f)); return RTD_ERR_HIP; }
    // (the transfer reads the slices [entry, passive) only, and the superposition's reduce writes every pixel of those: slices
    //  outside hold stale values that nothing samples; a fresh buffer is cleared once so that a fetch of "bev" reads zeros there)
    if (fresh) e = hipMemset(f->dBev, 0, P * (size_t)S * sizeof(float));
    if (fresh && e == hipSuccess && f->dNodeCount) e = hipMemset(f->dNodeCount, 0, nOutTiles * (size_t)S * 32 * sizeof(int));
    if (fresh && e == hipSuccess && f->dSwCount) e = hipMemset(f->dSwCount, 0, (size_t)S * sizeof(int));
    if (fresh && e == hipSuccess && f->dSwCountBig) e = hipMemset(f->dSwCountBig, 0, (size_t)S * sizeof(int));
    if (e != hipSuccess) { h->error = std::string("HIP error (clearing the BEV buffer / node counters): ") + hipGetErrorString(e); rtd_field_destroy(hh, reinterpret_cast<rtd_field>(f)); return RTD_ERR_HIP; }
    *out = reinterpret_cast<rtd_field>(f);
    return RTD_OK;
}

int rtd_field_create(rtd_handle hh, const rtd_beam* b, const uint32_t dose_dims[3], rtd_field* out) { return createField(hh, b, dose_dims, false, out); }
int rtd_field_create_remote(rtd_handle hh, const rtd_beam* b, const uint32_t dose_dims[3], rtd_field* out) { return createField(hh, b, dose_dims, true, out); }

// The beam loop body as launches only (kernel_wrapper.cu:766-1218). Asynchronous on the handle's stream.
// Part 1: everything up to the beam's-eye-view dose (:766-1105).
// Deferred CT (rtd_set_ct_deferred): the index box of the volume that the field's tracer can sample — the positions
// start(i, j) + k * inc(i, j) are multilinear in (i, j, k), so their extremes lie at the 8 corners of the ray grid x step range;
// +-2 voxels cover the interpolation neighbours and the rounding of the accumulated walk — is uploaded unless a box already on the
// device contains it. Asynchronous, on the handle's stream, in front of the tracer.
static int ensureCtBox(rtd_handle_impl* h, rtd_field_impl* f) {
    if (!h->ctHost) return RTD_OK;
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    const int is[2] = {0, f->fc.W - 1}, js[2] = {0, f->fc.H - 1};
    const double ks[2] = {0.0, (double)(f->fc.S - 1)};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 2; ++c) {
        const Vec3 st = f->tracer.getStart(is[a], js[b]), inc = f->tracer.getInc(is[a], js[b]);
        const double p[3] = {st.x + ks[c] * inc.x, st.y + ks[c] * inc.y, st.z + ks[c] * inc.z};
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[d]); }
    }
    std::array<int, 6> box;
    for (int d = 0; d < 3; ++d) {
        if (!(lo[d] == lo[d]) || !(hi[d] == hi[d])) { lo[d] = 0; hi[d] = (double)h->ctDims[d]; }     // NaN geometry: the whole axis
        const double a = std::floor(lo[d]) - 2.0, b = std::floor(hi[d]) + 3.0;
        box[d] = (int)std::max(a, 0.0);
        box[3 + d] = (int)std::min(b, (double)h->ctDims[d] - 1.0);
        if (box[3 + d] < box[d]) return RTD_OK;                       // the beam misses the volume: every sample is BORDER zero
    }
    for (const auto& cb : h->ctBoxes) {
        const auto& u = cb.box;
        if (u[0] <= box[0] && u[1] <= box[1] && u[2] <= box[2] && u[3] >= box[3] && u[4] >= box[4] && u[5] >= box[5]) {
            // uploaded on another stream (rtd_set_stream in between): this field's tracer is ordered behind that copy
            if (cb.stream != h->stream) RTD_HIP(h, hipStreamWaitEvent(h->stream, cb.done, 0));
            return RTD_OK;
        }
    }
    const size_t nx = h->ctDims[0], ny = h->ctDims[1];
    int x0 = box[0], x1 = box[3];
    if ((size_t)(x1 - x0 + 1) * 2 >= nx) { x0 = 0; x1 = (int)nx - 1; box[0] = x0; box[3] = x1; }   // wide boxes travel as whole rows
    hipMemcpy3DParms p;
    std::memset(&p, 0, sizeof p);
    p.srcPtr = make_hipPitchedPtr(const_cast<float*>(h->ctHost), nx * sizeof(float), nx, ny);
    p.dstPtr = make_hipPitchedPtr(h->dCtOwned, nx * sizeof(float), nx, ny);
    p.srcPos = p.dstPos = make_hipPos((size_t)x0 * sizeof(float), (size_t)box[1], (size_t)box[2]);
    p.extent = make_hipExtent((size_t)(x1 - x0 + 1) * sizeof(float), (size_t)(box[4] - box[1] + 1), (size_t)(box[5] - box[2] + 1));
    p.kind = hipMemcpyHostToDevice;
    RTD_HIP(h, hipMemcpy3DAsync(&p, h->stream));
    rtd_handle_impl::CtBox cb{box, nullptr, h->stream};
    RTD_HIP(h, hipEventCreateWithFlags(&cb.done, hipEventDisableTiming));
    h->ctBoxes.push_back(cb);
    RTD_HIP(h, hipEventRecord(cb.done, h->stream));
    return RTD_OK;
}

int rtd_field_compute_bev(rtd_handle hh, rtd_field ff) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f) return RTD_ERR_INVALID_ARG;
    if (f->remote) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_compute_bev: a remote field has no workspace (attach a slab instead)");
    if (!h->dCt || !h->haveLuts) return fail(h, RTD_ERR_NOT_READY, "rtd_field_compute: set LUTs and CT first");
    RTD_HIP(h, hipSetDevice(h->device));   // one host thread may drive handles on several devices
    { const int st = ensureCtBox(h, f); if (st != RTD_OK) return st; }
    // the uniform-sigma detection and kernel are skipped for a field that was found heterogeneous under the same CT / LUTs / options
    const bool tryUniform = f->uniformEligible && !(f->uniformHint == 0 && f->hintEpoch == h->inputEpoch);
    f->triedUniform = tryUniform;
    f->launchEpoch = h->inputEpoch;
    // ... and a field that was found uniform under the same inputs will be found uniform again (the test is exact arithmetic on
    // the same values): the general superposition kernel, all of whose ~10^5 blocks would only look at the flag and leave, is not launched
    const bool knownUniform = tryUniform && f->uniformHint == 1 && f->hintEpoch == h->inputEpoch;
    f->launchedKnownUniform = knownUniform;
    const FieldConst& fc = f->fc;
    hipStream_t s = h->stream;
    const bool timing = h->opt.fine_grained_timing != 0;
    const dim3 blk(kSuperpTileX, kSuperpTileY);
    const dim3 rayGrid(fc.W / kSuperpTileX, fc.H / kSuperpTileY);

    // Stage boundaries are the start / stop timestamps of the kernels themselves (hipExtLaunchKernelGGL), not event
    // packets between them: no barrier packet and no idle gap is inserted into the stream by the timing.
    auto ev = [&](int i) -> hipEvent_t { return timing ? f->ev[i] : nullptr; };
    const size_t lutLds = (size_t)(h->lut.nDensity + h->lut.nSp) * sizeof(float);
    // dIdd doubles as the HU scratch of the tracer (it is written by k_fill only afterwards)
    const size_t tLds = lutLds + (size_t)3 * kTrRays * kTrPitch * sizeof(float);
    const size_t dLds = lutLds + (size_t)3 * kTdSteps * kTdPitch * sizeof(float);
    if (f->traceMode == 2 && dLds <= 150 * 1024) {
        if (h->traceDLds < dLds) {
            RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_trace_sample_d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dLds));
            h->traceDLds = dLds;
        }
        launchK(k_trace_sample_d, dim3((unsigned)(fc.W / kTdRays), (unsigned)fc.H, (unsigned)((fc.S + kTdSteps - 1) / kTdSteps)), dim3(kTdThreads), dLds, s, f->ev[0], nullptr,
                (const float*)h->dCt, (int)h->ctDims[0], (int)h->ctDims[1], (int)h->ctDims[2], h->lut, f->tracer, fc.W, fc.H, f->dDensity, f->dWepl, f->dIdd,
                f->dRrl, h->rrlScale, f->dState, (const float*)f->dSegPos, f->traceDiagB);
    } else if (f->traceMode == 1 && tLds <= 144 * 1024) {
        // the beam runs along the CT x axis: lanes on consecutive steps of one ray (see k_trace_sample_t); 16 rays per block
        // measured best (4 .. 12 rays: 0.107 - 0.133 ms for the stage, 16: 0.100 ms)
        if (h->traceTLds < tLds) {
            RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_trace_sample_t), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tLds));
            h->traceTLds = tLds;
        }
        launchK(k_trace_sample_t, dim3((unsigned)((f->R + kTrRays - 1) / kTrRays)), dim3(64, kTrRays), tLds, s, f->ev[0], nullptr,
                (const float*)h->dCt, (int)h->ctDims[0], (int)h->ctDims[1], (int)h->ctDims[2], h->lut, f->tracer, fc.W, fc.H, f->dDensity, f->dWepl, f->dIdd,
                f->dRrl, h->rrlScale, f->dState);
    } else {
        launchK(k_trace_sample, dim3((unsigned)(f->R / 256), (fc.S + kTraceSeg * kTraceSegsPerBlock - 1) / (kTraceSeg * kTraceSegsPerBlock)), dim3(256, kTraceSegsPerBlock), lutLds, s, f->ev[0], nullptr,
                (const float*)h->dCt, (int)h->ctDims[0], (int)h->ctDims[1], (int)h->ctDims[2], h->lut, f->tracer, fc.W, fc.H, f->dDensity, f->dWepl, f->dIdd,
                f->dRrl, h->rrlScale, f->dState, (const float*)f->dSegPos);
    }
    constexpr size_t scanLds = 2 * 2 * kScanChunk * 64 * sizeof(float);   // two buffers of 64 KiB: above the 64 KiB default cap of dynamic LDS
    if (!h->scanLdsSet) {
        RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_trace_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scanLds));
        h->scanLdsSet = true;
    }
    if (!f->dScanDbg && std::getenv("RTD_SCAN_DEBUG")) {
        f->scanDbgN = (size_t)8 * (f->R / 64);
        RTD_HIP(h, hipMalloc((void**)&f->dScanDbg, f->scanDbgN * sizeof(long long)));
        RTD_HIP(h, hipMemset(f->dScanDbg, 0, f->scanDbgN * sizeof(long long)));
    }
    const ResetJob resetJob{f->dLayers, fc.L, reinterpret_cast<unsigned int*>(f->dTileRad), f->tileRadWords, f->dActive, (size_t)4 * fc.L * fc.S,
                            f->dNucIdd, f->dNucRs, fc.nuclearCorr ? (size_t)fc.nucW * fc.nucH * fc.L : (size_t)0,
                            f->dSigMin, f->dSigMax, (size_t)fc.L * fc.S, f->dScanDbg};
    launchK(k_trace_scan, dim3((unsigned)(f->R / 64)), dim3(64, kScanWaves), scanLds, s, nullptr, ev(1), (const float*)f->dIdd, f->dWepl, fc.W, fc.H,
            (unsigned)fc.S, f->dFirstInside, f->dFirstOutside, f->dState, f->dBlockWeplMin, resetJob);
    if (fc.spotNy <= kPlanConvMaxRows && std::getenv("RTD_SEPARATE_PLAN") == nullptr) {
        // the plan and the spot -> ray convolution in one launch (k_plan_conv): neither reads what the other writes
        launchK(k_plan_conv, dim3(fc.W / 32, (fc.H / 8 + 3) / 4, fc.L + 1), dim3(1024), (size_t)4 * fc.spotNy * 32 * sizeof(float), s, nullptr, ev(2),
                (const float*)f->dSpotWeights, f->dRayWeights, f->dLayers, f->dState, (const float*)f->dBlockWeplMin, (int)(f->R / 64), f->dWeplMin, fc);
    } else {
    k_plan<<<1, 1024, 0, s>>>(f->dState, f->dLayers, (const float*)f->dBlockWeplMin, (int)(f->R / 64), f->dWeplMin, fc);
    if (fc.spotNy <= kConvMaxRows) {
        // both passes in one launch, the x pass staged in LDS (k_conv)
        launchK(k_conv, dim3(fc.W / 32, fc.H / 8, fc.L), blk, (size_t)fc.spotNy * 32 * sizeof(float), s, nullptr, ev(2), (const float*)f->dSpotWeights,
                f->dRayWeights, (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc);
    } else {
        k_conv_x<<<dim3(fc.W / 32, (fc.spotNy + 7) / 8, fc.L), blk, 0, s>>>(f->dSpotWeights, f->dConvInterm, f->dLayers, f->dState, fc);
        launchK(k_conv_y, dim3(fc.W / 32, fc.H / 8, fc.L), blk, 0, s, nullptr, ev(2), (const float*)f->dConvInterm, f->dRayWeights,
                (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc);
    }
    }
    {
        const size_t fillLds = (size_t)(2 * h->lut.nSamples) * sizeof(float);   // the layer's two cumulative-IDD rows
        const dim3 fillGrid(2 * rayGrid.x * rayGrid.y * fc.L);          // (layer, tile, role) items: sigma walk and dose walk of every tile; placement is decided in the kernel
        const dim3 fillBlk = blk;
        const NucFill nucFill{f->dNucSpotIdx, f->dNucRayWeights, f->dNucIdd, f->dNucRs};
        if (!f->dFillDbg && std::getenv("RTD_FILL_DEBUG")) {
            f->fillDbgN = (size_t)4 * fillGrid.x;
            RTD_HIP(h, hipMalloc((void**)&f->dFillDbg, f->fillDbgN * sizeof(long long)));
        }
        auto launchFill = [&](auto kern, size_t lds) {
            launchK(kern, fillGrid, fillBlk, lds, s, nullptr, ev(3), (const float*)f->dDensity, (const float*)f->dWepl, (const float*)f->dRrl, f->dIdd,
                    f->dRSigma, (const float*)f->dRayWeights, (const int*)f->dFirstInside, (const int*)f->dFirstOutside,
                    f->dFirstPassive, f->dTileRad, f->dLayers, f->dState, h->lut, f->fillGeom, fc, (const float*)f->dStepTab, f->dActive, h->numCUs, f->dFillDbg, nucFill, f->dSigMin, f->dSigMax, tryUniform ? 1 : 0);
        };
        const bool ldsLut = fillLds <= 56 * 1024;     // (+ ~1 KiB of static arrays: stays under the 64 KiB default cap of a block's LDS)
        constexpr size_t sigLds = (size_t)2 * kFillBatch * 256 * sizeof(float);   // the sigma walk's exchange buffers share the dynamic LDS with the dose walk's LUT rows
        const size_t dynLds = std::max(sigLds, ldsLut ? fillLds : (size_t)0);
        if (fc.nuclearCorr) { if (ldsLut) launchFill((k_fill<true, true>), dynLds); else launchFill((k_fill<false, true>), dynLds); }
        else { if (ldsLut) launchFill((k_fill<true, false>), dynLds); else launchFill((k_fill<false, false>), dynLds); }
    }
    if (fc.nuclearCorr) {
        // the halo's plan runs first: its radius overflow (kernel_wrapper.cu:984) is reported in the primary state, which k_ks_plan mirrors
        k_nuc_plan<<<1, 256, 0, s>>>(f->dState, f->dStateNuc, (const LayerPlan*)f->dLayers, (const float*)f->dNucRs, f->dNucEffT, fc,
                                     f->nucIdxToDoseIdx, f->transfer0Nuc, (int)f->doseDims[0], (int)f->doseDims[1], (int)f->doseDims[2]);
    }
    const KsPlanArgs ksArgs{f->dState, f->dLayers, f->rayIdxToDoseIdx, f->transfer0, (int)f->doseDims[0], (int)f->doseDims[1], (int)f->doseDims[2],
                            f->ksGroups, f->swGroups, f->dHostState, f->dStateNuc, (const unsigned int*)f->dSigMin, (const unsigned int*)f->dSigMax,
                            tryUniform ? 1 : 0, f->sweepEnabled ? kSwMaxR : -1, f->bgGroups};
    // Once the host knows that the field is not a uniform-sigma one (and without the halo), the sweep's launch plans for itself
    // (k_superpose_sweep<true>: its block 0 is the plan): one launch and its gap less on the critical path.
    const bool selfPlan = f->sweepEnabled && !tryUniform && !fc.nuclearCorr && std::getenv("RTD_SEPARATE_KS_PLAN") == nullptr;
    f->selfPlanned = selfPlan;
    // (a field that may be a uniform-sigma one has its 2 x L x S sigma extremes compared by this one block: 1024 threads make that
    //  three memory round trips instead of ten)
    if (!selfPlan) launchK(k_ks_plan, dim3(1), dim3(tryUniform ? 1024 : 256), 0, s, nullptr, f->ev[4], ksArgs, fc);
    if (fc.nuclearCorr) {
        const int nPix = (fc.nucW + 2 * kMaxSuperpR) * (fc.nucH + 2 * kMaxSuperpR);
        k_nuc_superpose<<<(nPix + 255) / 256, 256, 0, s>>>((const float*)f->dNucIdd, (const float*)f->dNucRs, (const int*)f->dNucEffT,
                                                           (const FieldState*)f->dStateNuc, fc, f->dNucBev);
    }
    hipEvent_t ksStart = ev(8);
    if (tryUniform) {
        // A field with one sigma per slice (water) is superposed as a separable convolution; whether this field is one is known
        // on the device only (FieldState::uniformField): the launch returns at once otherwise, k_superpose_mfma below when it is.
        // A small persistent grid, so that the empty launch of a heterogeneous field costs next to nothing.
        const int nYB = (fc.bevH + 15) / 16;
        const bool u4 = fc.W <= 16 * (kU2XB - 4) && fc.W % 4 == 0 && std::getenv("RTD_UNIFORM_V2") == nullptr;
        if (u4) {
            // (rtd_uniform.hpp: a block per four row blocks of a slice, the rows within their reach staged layer by layer)
            const int nPartsU4 = (nYB + kU4RB - 1) / kU4RB;
            if (!h->uni4LdsSet) {
                RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_superpose_uniform4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)((kU4Rows + kU4Slack) * (16 * (kU2XB - 4) + 16) * sizeof(float))));
                h->uni4LdsSet = true;
            }
            if (!f->dUniDbg && std::getenv("RTD_UNIFORM_DEBUG")) {
                f->uniDbgN = (size_t)16 * fc.S * nPartsU4;
                RTD_HIP(h, hipMalloc((void**)&f->dUniDbg, f->uniDbgN * sizeof(long long)));
                RTD_HIP(h, hipMemset(f->dUniDbg, 0, f->uniDbgN * sizeof(long long)));
            }
            launchK(k_superpose_uniform4, dim3((unsigned)(fc.S * nPartsU4)), dim3(64 * kU4Waves), (size_t)(kU4Rows + kU4Slack) * (fc.W + 16) * sizeof(float), s, ksStart,
                    knownUniform ? f->ev[5] : nullptr, (const float*)f->dIdd, (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc,
                    (const unsigned int*)f->dSigMin, (const float*)f->dStepTab, f->dBev, f->dUniDbg);
        } else {
            // (rtd_uniform.hpp: one wave per 16 rows x 192 columns of a slice, no staging, no barrier in its loop)
            const int nXS = (fc.bevW + 16 * kU2XB - 1) / (16 * kU2XB), nParts = ((fc.bevH + 15) / 16 + 3) / 4;
            launchK(k_superpose_uniform2, dim3((unsigned)(fc.S * nXS * nParts)), dim3(256), 0, s, ksStart, knownUniform ? f->ev[5] : nullptr, (const float*)f->dIdd,
                    (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc, (const unsigned int*)f->dSigMin, (const float*)f->dStepTab, f->dBev, nXS);
        }
        ksStart = nullptr;
    }
    // The general superposition is the row sweep in two launches: k_superpose_sweep for the sources whose batch radius is within its
    // reach (<= 16; it writes every slice), then k_superpose_sweep_big for the rest (17 .. 32), added to the slices. Whether a field has
    // such a rest is known on the device (FieldState::maxRadius, k_ks_plan): the second launch returns at once if not — and is left out
    // once a finished compute has told the host, under the same CT / LUTs / options, that it does not. (RTD_NO_SWEEP: k_superpose_mfma,
    // round 2's output-stationary kernel, takes everything — kept as a second implementation the tests compare the sweep with.)
    const bool radiusKnown = f->radiusHint >= 0 && f->hintEpoch == h->inputEpoch;
    const bool runSweep = !knownUniform && f->sweepEnabled;
    const bool runBig = runSweep && !(radiusKnown && f->radiusHint <= kSwMaxR);
    const bool runMfma = !knownUniform && !f->sweepEnabled;
    if (runSweep) {
        constexpr size_t swLds = (size_t)kSwLdsWords * sizeof(float);
        static_assert(sizeof(KsPlanLds) <= swLds, "the plan block's LDS is the front of the sweep's");
        if (!h->sweepLdsSet) {
            RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_superpose_sweep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)swLds));
            RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_superpose_sweep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)swLds));
            h->sweepLdsSet = true;
        }
        const unsigned nSwBlocks = (unsigned)(fc.S * f->swPX * f->swPY * f->swGroups) + (selfPlan ? 1u : 0u);
        if (!f->dSweepDbg && std::getenv("RTD_SWEEP_DEBUG")) {
            f->sweepDbgN = (size_t)(8 + 4 * 16) * (fc.S * f->swPX * f->swPY * f->swGroups + 1);
            RTD_HIP(h, hipMalloc((void**)&f->dSweepDbg, f->sweepDbgN * sizeof(long long)));
            RTD_HIP(h, hipMemset(f->dSweepDbg, 0, f->sweepDbgN * sizeof(long long)));
        }
        auto launchSweep = [&](auto kern) {
            launchK(kern, dim3(nSwBlocks), dim3(64 * kSwWaves), swLds, s, ksStart, runBig ? nullptr : f->ev[5],
                    (const float*)f->dIdd, (const float*)f->dRSigma, (const unsigned char*)f->dTileRad, (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc,
                    f->swGroups, f->swPX, f->swPY, (const int*)f->dActive, f->dSwSlots, f->dSwCount, f->dBev, f->dSweepDbg, (const KsPlanArgs*)f->dKsArgs);
        };
        if (selfPlan) launchSweep(k_superpose_sweep<true>); else launchSweep(k_superpose_sweep<false>);
        ksStart = nullptr;
    }
    if (runBig) {
        constexpr size_t bgLds = (size_t)kBgLdsWords * sizeof(float);
        if (!h->sweepBigLdsSet) {
            RTD_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_superpose_sweep_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bgLds));
            h->sweepBigLdsSet = true;
        }
        if (!f->dSweepBigDbg && std::getenv("RTD_SWEEP_DEBUG")) {
            f->sweepBigDbgN = (size_t)48 * fc.S * f->swPX * f->swPY * f->bgGroups;
            RTD_HIP(h, hipMalloc((void**)&f->dSweepBigDbg, f->sweepBigDbgN * sizeof(long long)));
            RTD_HIP(h, hipMemset(f->dSweepBigDbg, 0, f->sweepBigDbgN * sizeof(long long)));
        }
        launchK(k_superpose_sweep_big, dim3((unsigned)(fc.S * f->swPX * f->swPY * f->bgGroups)), dim3(64 * kSwWaves), bgLds, s, nullptr, f->ev[5],
                (const float*)f->dIdd, (const float*)f->dRSigma, (const unsigned char*)f->dTileRad, (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc,
                f->bgGroups, f->swPX, f->swPY, (const int*)f->dActive, f->dSwSlotsBig, f->dSwCountBig, f->dBev, f->dSweepBigDbg);
    }
    if (runMfma) {
        const int nTX = (fc.bevW + kKsTileX - 1) / kKsTileX, nTY = (fc.bevH + kKsTileY - 1) / kKsTileY;
        const int G = f->ksGroups;
        const int nItems = fc.S * G * nTY * nTX;
        // few layers -> few, long work items: deal each item's chunks to 2 or 4 waves (the live items are a fraction of nItems)
        const int split = nItems >= 48 * 1024 ? 1 : (nItems >= 20 * 1024 ? 2 : 4);
        auto launchKs = [&](auto kernel) {
            launchK(kernel, dim3(nItems), dim3(64 * split), 0, s, ksStart, f->ev[5], (const float*)f->dIdd, (const float*)f->dRSigma,
                    f->dBevPart, (const unsigned char*)f->dTileRad, (const LayerPlan*)f->dLayers, (const FieldState*)f->dState, fc, nTX, nTY, G,
                    (const int*)f->dActive, f->dBev, f->dNodeCount, -1);
        };
        if (split == 1) launchKs(k_superpose_mfma<1>); else if (split == 2) launchKs(k_superpose_mfma<2>); else launchKs(k_superpose_mfma<4>);
    }
    RTD_HIP(h, hipGetLastError());
    f->computed = true;
    f->transferred = false;
    return RTD_OK;
}

static ClipBox makeClip(const int32_t* lo, const int32_t* hi) {
    ClipBox c;
    for (int i = 0; i < 3; ++i) { c.lo[i] = lo ? lo[i] : -0x40000000; c.hi[i] = hi ? hi[i] : 0x40000000; }
    return c;
}

// Part 2: fan -> dose-grid transfer (:1185-1218) of the field's BEV dose — its own, or the slab another GPU exported —
// into dev_dose, optionally restricted to a box of the dose grid (a GPU's slab of the plan's volume).
static int transferImpl(rtd_handle hh, rtd_field ff, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3], bool init);
int rtd_field_transfer(rtd_handle hh, rtd_field ff, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]) {
    return transferImpl(hh, ff, dev_dose, clip_min, clip_max, false);
}
// The same transfer for the FIRST field into a volume that is zero everywhere except possibly inside this field's dose box: the
// box is written (dose or zero), not accumulated into — no separate clear, no read of the old values.
int rtd_field_transfer_init(rtd_handle hh, rtd_field ff, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]) {
    return transferImpl(hh, ff, dev_dose, clip_min, clip_max, true);
}
static int transferImpl(rtd_handle hh, rtd_field ff, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3], bool init) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f || !dev_dose) return RTD_ERR_INVALID_ARG;
    if (!f->computed) return fail(h, RTD_ERR_NOT_READY, "rtd_field_transfer: no BEV dose (compute the field or attach a slab first)");
    RTD_HIP(h, hipSetDevice(h->device));
    const FieldConst& fc = f->fc;
    hipStream_t s = h->stream;
    const dim3 blk(kSuperpTileX, kSuperpTileY);
    const ClipBox clip = makeClip(clip_min, clip_max);
    const float* bev = f->remote ? reinterpret_cast<const float*>(f->attached + kPackHeader) : f->dBev;
    const FieldState* st = f->remote ? reinterpret_cast<const FieldState*>(f->attached) : f->dState;
    // depth of a brick along the axis a thread walks: 16 for a whole dose box; a clipped transfer (a GPU's slab of a multi-GPU
    // plan) has a fraction of the bricks and is latency-bound on the 4 rounds of a 16-deep brick: 4 there (measured on the four
    // quarter-slab transfers of the bench plan: 36 -> 32 us plain, 47 -> 39 us transposed)
    const int zChunk = (clip_min && clip_max) ? 4 : 16;
    {
        // grid-stride over the bricks of the device-side box; never more blocks than bricks of the whole volume
        const size_t allBricks = (size_t)((f->doseDims[0] + 31) / 32) * ((f->doseDims[1] + 7) / 8) * ((f->doseDims[2] + zChunk - 1) / zChunk);
        const unsigned tg = (unsigned)std::min<size_t>(allBricks, (size_t)h->numCUs * 8 * 4);
        const bool halo = !f->remote && fc.nuclearCorr != 0;
        auto launchT = [&](auto kern, const float* slab, const FieldState* state, hipEvent_t startEv, hipEvent_t stopEv) {
            launchK(kern, dim3(tg), blk, 0, s, startEv, stopEv, dev_dose, (int)f->doseDims[0], (int)f->doseDims[1],
                    (int)f->doseDims[2], slab, state, fc, zChunk, clip);
        };
        // lanes run along the dose axis that moves fastest along BEV x, so that the gathers stay within few BEV rows
        hipEvent_t e0 = f->remote ? f->ev[0] : nullptr, e1 = halo ? nullptr : f->ev[6];
        if (init && halo) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_transfer_init: not with nuclear_corr (the halo's box differs from the primary's)");
        if (init) {
            switch (f->transferMode) {
                case 0: launchT((k_transfer<true>), bev, st, e0, e1); break;
                case 1: launchT((k_transfer_t<1, true>), bev, st, e0, e1); break;
                default: launchT((k_transfer_t<2, true>), bev, st, e0, e1); break;
            }
        } else {
            switch (f->transferMode) {
                case 0: launchT((k_transfer<false>), bev, st, e0, e1); break;
                case 1: launchT((k_transfer_t<1, false>), bev, st, e0, e1); break;
                default: launchT((k_transfer_t<2, false>), bev, st, e0, e1); break;
            }
        }
        if (halo) {   // NUCLEAR_CORR: nucTransfDiv (kernel_wrapper.cu:100-127, launch :1221-1254) after the primary transfer, like the reference
            switch (f->transferModeNuc) {
                case 0: launchT((k_transfer<false>), (const float*)f->dNucBev, (const FieldState*)f->dStateNuc, nullptr, f->ev[6]); break;
                case 1: launchT((k_transfer_t<1, false>), (const float*)f->dNucBev, (const FieldState*)f->dStateNuc, nullptr, f->ev[6]); break;
                default: launchT((k_transfer_t<2, false>), (const float*)f->dNucBev, (const FieldState*)f->dStateNuc, nullptr, f->ev[6]); break;
            }
        }
    }
    RTD_HIP(h, hipGetLastError());
    f->transferred = true;
    return RTD_OK;
}

// Several fields (own BEV doses and / or attached slabs) into one box of the dose grid in one launch: every voxel of the
// box is written with 0 + field 0 + field 1 + ... (k_transfer_multi) — the loop of rtd_field_transfer over the fields into a
// zeroed box, bit for bit, without its read-modify-write passes. The timing / completion events go to the last own field
// of the list (the first field if all are remote): rtd_field_finish of THAT field waits for the launch.
int rtd_fields_transfer_init(rtd_handle hh, const rtd_field* fields, uint32_t n_fields, float* dev_dose, const int32_t box_min[3],
                             const int32_t box_max[3]) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h || !fields || !dev_dose || n_fields == 0) return RTD_ERR_INVALID_ARG;
    if (n_fields > (uint32_t)kMultiMaxFields) return fail(h, RTD_ERR_INVALID_ARG, "rtd_fields_transfer_init: more than 16 fields in one call");
    RTD_HIP(h, hipSetDevice(h->device));
    MultiFields mf;
    std::memset(&mf, 0, sizeof mf);
    mf.n = (int)n_fields;
    rtd_field_impl* lead = nullptr;
    uint32_t dims[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n_fields; ++i) {
        auto* f = reinterpret_cast<rtd_field_impl*>(fields[i]);
        if (!f) return RTD_ERR_INVALID_ARG;
        if (!f->computed) return fail(h, RTD_ERR_NOT_READY, "rtd_fields_transfer_init: a field has no BEV dose (compute it or attach a slab first)");
        if (!f->remote && f->fc.nuclearCorr) return fail(h, RTD_ERR_INVALID_ARG, "rtd_fields_transfer_init: not with nuclear_corr (the halo is transferred separately)");
        if (i == 0) std::memcpy(dims, f->doseDims, sizeof dims);
        else if (std::memcmp(dims, f->doseDims, sizeof dims) != 0) return fail(h, RTD_ERR_INVALID_ARG, "rtd_fields_transfer_init: fields of different dose grids");
        mf.bev[i] = f->remote ? reinterpret_cast<const float*>(f->attached + kPackHeader) : f->dBev;
        mf.st[i] = f->remote ? reinterpret_cast<const FieldState*>(f->attached) : f->dState;
        mf.mode[i] = f->transferMode;
        if (!f->remote || !lead) lead = f;                            // the last own field, else the first field
    }
    ClipBox box;
    for (int a = 0; a < 3; ++a) {
        box.lo[a] = box_min ? std::max(box_min[a], 0) : 0;
        box.hi[a] = box_max ? std::min(box_max[a], (int32_t)dims[a] - 1) : (int32_t)dims[a] - 1;
        if (box.hi[a] < box.lo[a]) return RTD_OK;                    // empty box: nothing to write
    }
    const size_t bricks = (size_t)((box.hi[0] - box.lo[0]) / 16 + 1) * ((box.hi[1] - box.lo[1]) / 16 + 1) * ((box.hi[2] - box.lo[2]) / 16 + 1);
    const unsigned g = (unsigned)std::min<size_t>(bricks, (size_t)h->numCUs * 8 * 4);
    for (uint32_t i = 0; i < n_fields; ++i) reinterpret_cast<rtd_field_impl*>(fields[i])->transferred = false;
    launchK(k_transfer_multi, dim3(g), dim3(256), 0, h->stream, lead->remote ? lead->ev[0] : nullptr, lead->ev[6], dev_dose, (int)dims[0], (int)dims[1],
            (int)dims[2], mf, box);
    RTD_HIP(h, hipGetLastError());
    lead->transferred = true;
    return RTD_OK;
}

int rtd_field_compute(rtd_handle hh, rtd_field ff, float* dev_dose) {
    if (!dev_dose) return RTD_ERR_INVALID_ARG;
    const int st = rtd_field_compute_bev(hh, ff);
    return st != RTD_OK ? st : rtd_field_transfer(hh, ff, dev_dose, nullptr, nullptr);
}

int rtd_field_clear_dose_box(rtd_handle hh, rtd_field ff, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f || !dev_dose) return RTD_ERR_INVALID_ARG;
    if (!f->computed) return fail(h, RTD_ERR_NOT_READY, "rtd_field_clear_dose: field not computed");
    RTD_HIP(h, hipSetDevice(h->device));
    const int zChunk = 16;
    const size_t allBricks = (size_t)((f->doseDims[0] + 31) / 32) * ((f->doseDims[1] + 7) / 8) * ((f->doseDims[2] + zChunk - 1) / zChunk);
    const unsigned g = (unsigned)std::min<size_t>(allBricks, (size_t)h->numCUs * 8 * 4);
    const FieldState* st = f->remote ? reinterpret_cast<const FieldState*>(f->attached) : f->dState;
    k_clear_box<<<g, dim3(kSuperpTileX, kSuperpTileY), 0, h->stream>>>(dev_dose, (int)f->doseDims[0], (int)f->doseDims[1], st, zChunk,
                                                                       makeClip(clip_min, clip_max));
    if (!f->remote && f->fc.nuclearCorr)     // the halo's dose box (its slice reaches further sideways than the primary's)
        k_clear_box<<<g, dim3(kSuperpTileX, kSuperpTileY), 0, h->stream>>>(dev_dose, (int)f->doseDims[0], (int)f->doseDims[1],
                                                                           (const FieldState*)f->dStateNuc, zChunk, makeClip(clip_min, clip_max));
    RTD_HIP(h, hipGetLastError());
    return RTD_OK;
}

int rtd_field_clear_dose(rtd_handle hh, rtd_field ff, float* dev_dose) { return rtd_field_clear_dose_box(hh, ff, dev_dose, nullptr, nullptr); }

// ---- multi-GPU plans: a field's BEV slab travels, the receiving GPU transfers it into its slab of the dose volume ----

// Waits for the field's device-side plan only (k_ks_plan: entry / passive steps, the BEV rectangle that carries dose, the dose
// box) while the superposition still runs, and reports the size of the message rtd_field_export_bev will write.
int rtd_field_wait_plan(rtd_handle hh, rtd_field ff, rtd_field_info* info, size_t* packed_bytes) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f) return RTD_ERR_INVALID_ARG;
    if (!f->computed || f->remote) return fail(h, RTD_ERR_NOT_READY, "rtd_field_wait_plan: field not computed on this handle");
    RTD_HIP(h, hipSetDevice(h->device));
    RTD_HIP(h, hipEventSynchronize(f->selfPlanned ? f->ev[5] : f->ev[4]));   // (a launch that planned for itself: its plan is complete when the superposition is)
    const FieldState st = *f->hState;                                // mirrored by the plan into pinned host memory
    // (the finding belongs to the inputs the compute was LAUNCHED under: CT, LUTs or options may have changed since)
    if (f->triedUniform) { f->uniformHint = st.uniformField ? 1 : 0; f->hintEpoch = f->launchEpoch; }
    else if (f->hintEpoch != f->launchEpoch) { f->uniformHint = -1; f->hintEpoch = f->launchEpoch; }
    f->radiusHint = (st.errorFlags || st.empty) ? -1 : st.maxRadius;   // (valid under hintEpoch, like the uniform hint)
    if (f->launchedKnownUniform && !st.uniformField && !st.errorFlags && !st.empty)
        return fail(h, RTD_ERR_NOT_READY, "the field was launched as a uniform-sigma field but is not one: its inputs were modified in place; call rtd_set_ct* again and recompute");
    if (info) fillInfo(f, st, info);
    if (packed_bytes) {
        const int nz = std::max(st.firstCalculatedPassive - st.beamFirstInside, 0);
        const int x0 = std::max(st.bevLo[0] - 1, 0) & ~3, x1 = std::min(st.bevHi[0] + 1, f->fc.bevW - 1);
        const int y0 = std::max(st.bevLo[1] - 1, 0), y1 = std::min(st.bevHi[1] + 1, f->fc.bevH - 1);
        const bool none = nz == 0 || x1 < x0 || y1 < y0;
        *packed_bytes = (size_t)kPackHeader + (none ? 0 : (size_t)nz * (y1 - y0 + 1) * ((x1 - x0 + 4) / 4) * 16);
    }
    return RTD_OK;
}

size_t rtd_bev_message_bound(rtd_handle hh, rtd_field ff) {
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    (void)hh;
    return f ? (size_t)kPackHeader + (size_t)f->fc.bevW * f->fc.bevH * (size_t)f->fc.S * sizeof(float) : 0;
}

int rtd_field_export_bev(rtd_handle hh, rtd_field ff, void* dev_buf, size_t capacity) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f || !dev_buf || capacity < (size_t)kPackHeader) return RTD_ERR_INVALID_ARG;
    if (!f->computed || f->remote) return fail(h, RTD_ERR_NOT_READY, "rtd_field_export_bev: field not computed on this handle");
    if (f->fc.nuclearCorr) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_export_bev: the halo slab of nuclear_corr is not exported (one GPU per field only)");
    RTD_HIP(h, hipSetDevice(h->device));
    k_pack_bev<<<dim3((unsigned)h->numCUs * 4), dim3(256), 0, h->stream>>>((const float*)f->dBev, (const FieldState*)f->dState, f->fc,
                                                                           reinterpret_cast<unsigned char*>(dev_buf), capacity);
    RTD_HIP(h, hipGetLastError());
    return RTD_OK;
}

int rtd_field_attach_bev(rtd_handle hh, rtd_field ff, const void* dev_buf) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f || !dev_buf) return RTD_ERR_INVALID_ARG;
    if (!f->remote) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_attach_bev: not a remote field (rtd_field_create_remote)");
    f->attached = reinterpret_cast<const unsigned char*>(dev_buf);
    f->computed = true;
    f->transferred = false;
    return RTD_OK;
}

int rtd_field_finish(rtd_handle hh, rtd_field ff, rtd_timing* timing, rtd_field_info* info) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f) return RTD_ERR_INVALID_ARG;
    if (!f->computed) return fail(h, RTD_ERR_NOT_READY, "rtd_field_finish: field not computed");
    // Wait for THIS field's last kernel only (not for the whole stream): a caller that alternates two fields can launch the
    // next plan before finishing the previous one, and the device never idles on the host's bookkeeping. The state record was
    // mirrored into pinned host memory by k_ks_plan: no copy is issued here.
    RTD_HIP(h, hipSetDevice(h->device));
    if (f->remote) {
        // a slab from another GPU: the state record is the message header (device memory) — copy it once the transfer is done
        if (f->transferred) RTD_HIP(h, hipEventSynchronize(f->ev[6]));
        FieldState st;
        RTD_HIP(h, hipMemcpy(&st, f->attached, sizeof st, hipMemcpyDeviceToHost));
        if (timing) {
            std::memset(timing, 0, sizeof *timing);
            if (f->transferred) { RTD_HIP(h, hipEventElapsedTime(&timing->transforming_ms, f->ev[0], f->ev[6])); timing->total_ms = timing->transforming_ms; }
        }
        if (info) fillInfo(f, st, info);
        if (st.errorFlags & kErrPackOverflow) return fail(h, RTD_ERR_INVALID_ARG, "BEV message buffer too small for the exported slab");
        if (st.errorFlags & kErrRadiusOverflow)
            return fail(h, RTD_ERR_RADIUS_OVERFLOW, "Found larger than allowed kernel superposition radius");
        return RTD_OK;
    }
    const int last = f->transferred ? 6 : 5;                         // BEV only: the superposition's reduce is the last kernel
    RTD_HIP(h, hipEventSynchronize(f->ev[last]));
    const FieldState st = *f->hState;                                // mirrored by k_ks_plan into pinned host memory
    if (f->triedUniform) { f->uniformHint = st.uniformField ? 1 : 0; f->hintEpoch = f->launchEpoch; }
    else if (f->hintEpoch != f->launchEpoch) { f->uniformHint = -1; f->hintEpoch = f->launchEpoch; }
    f->radiusHint = (st.errorFlags || st.empty) ? -1 : st.maxRadius;   // (valid under hintEpoch, like the uniform hint)
    // A compute that skipped the general kernel (hint: uniform) on a field the device then found heterogeneous has written no BEV
    // dose: only possible when the caller changed a bound device volume in place (rtd_set_ct_device) without telling the handle.
    if (f->launchedKnownUniform && !st.uniformField && !st.errorFlags && !st.empty)
        return fail(h, RTD_ERR_NOT_READY, "the field was launched as a uniform-sigma field but is not one: its inputs were modified in place; call rtd_set_ct* again and recompute");
    if (timing) {
        std::memset(timing, 0, sizeof *timing);
        RTD_HIP(h, hipEventElapsedTime(&timing->total_ms, f->ev[0], f->ev[last]));
        if (h->opt.fine_grained_timing) {
            RTD_HIP(h, hipEventElapsedTime(&timing->raytracing_ms, f->ev[0], f->ev[1]));
            RTD_HIP(h, hipEventElapsedTime(&timing->prepare_energy_loop_ms, f->ev[1], f->ev[2]));
            RTD_HIP(h, hipEventElapsedTime(&timing->fill_idd_sigma_ms, f->ev[2], f->ev[3]));
            hipEvent_t planEnd = f->selfPlanned ? f->ev[8] : f->ev[4];   // (self-planned: the plan is inside the superposition launch)
            RTD_HIP(h, hipEventElapsedTime(&timing->prepare_superp_ms, f->ev[3], planEnd));
            RTD_HIP(h, hipEventElapsedTime(&timing->superp_ms, planEnd, f->ev[5]));
            RTD_HIP(h, hipEventElapsedTime(&timing->superp_kernel_ms, f->ev[8], f->ev[5]));
            if (f->transferred) RTD_HIP(h, hipEventElapsedTime(&timing->transforming_ms, f->ev[5], f->ev[6]));
        }
        timing->superp_launches = 1;   // k_superpose_mfma, all layers and radii (the reference: up to 33 launches per layer)
        timing->ray_dims[0] = (uint32_t)f->fc.W; timing->ray_dims[1] = (uint32_t)f->fc.H;
        timing->steps = (uint32_t)f->fc.S; timing->n_layers = (uint32_t)f->fc.L;
        timing->transfer_voxels = 1;
        for (int i = 0; i < 3; ++i) timing->transfer_voxels *= (int64_t)std::max(st.tboxMax[i] - st.tboxMin[i] + 1, 0);
    }
    if (info) fillInfo(f, st, info);
    if (st.errorFlags & kErrRadiusOverflow)
        return fail(h, RTD_ERR_RADIUS_OVERFLOW, "Found larger than allowed kernel superposition radius");   // kernel_wrapper.cu:965
    return RTD_OK;
}

int rtd_field_fetch(rtd_handle hh, rtd_field ff, const char* name, void* host_out, size_t bytes, size_t* bytes_needed) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    auto* f = reinterpret_cast<rtd_field_impl*>(ff);
    if (!h || !f || !name) return RTD_ERR_INVALID_ARG;
    const FieldConst& fc = f->fc;
    const size_t R = f->R, S = fc.S, L = fc.L, tiles = (size_t)fc.tilesX * fc.tilesY;
    const void* src = nullptr; size_t n = 0;
    std::string nm(name);
    std::vector<char> staging;
    RTD_HIP(h, hipStreamSynchronize(h->stream));
    if (nm == "density") { src = f->dDensity; n = 4 * R * S; }
    else if (nm == "wepl") { src = f->dWepl; n = 4 * R * S; }
    else if (nm == "first_inside") { src = f->dFirstInside; n = 4 * R; }
    else if (nm == "first_outside") { src = f->dFirstOutside; n = 4 * R; }
    else if (nm == "wepl_min") { src = f->dWeplMin; n = 4 * S; }
    else if (nm == "ray_weights") { src = f->dRayWeights; n = 4 * R * L; }
    else if (nm == "idd") { src = f->dIdd; n = 4 * R * S * L; }
    else if (nm == "rsigma") { src = f->dRSigma; n = 4 * R * S * L; }
    else if (nm == "first_passive") { src = f->dFirstPassive; n = 4 * R * L; }
    else if (nm == "tile_radius") { src = f->dTileRad; n = L * S * tiles; }
    else if (nm == "bev") { src = f->dBev; n = 4 * (size_t)fc.bevW * fc.bevH * S; }
    else if (nm == "fill_debug" && f->dFillDbg) { src = f->dFillDbg; n = f->fillDbgN * sizeof(long long); }
    else if (nm == "uniform_debug" && f->dUniDbg) { src = f->dUniDbg; n = f->uniDbgN * sizeof(long long); }
    else if (nm == "sweep_debug" && f->dSweepDbg) { src = f->dSweepDbg; n = f->sweepDbgN * sizeof(long long); }
    else if (nm == "sweep_big_debug" && f->dSweepBigDbg) { src = f->dSweepBigDbg; n = f->sweepBigDbgN * sizeof(long long); }
    else if (nm == "scan_debug" && f->dScanDbg) { src = f->dScanDbg; n = f->scanDbgN * sizeof(long long); }
    else if (nm == "eff_radius" || nm == "layer_plan") {
        std::vector<LayerPlan> lp(L);
        RTD_HIP(h, hipMemcpy(lp.data(), f->dLayers, L * sizeof(LayerPlan), hipMemcpyDeviceToHost));
        if (nm == "eff_radius") {
            n = 4 * L * (kMaxSuperpR + 2); staging.resize(n);
            for (size_t l = 0; l < L; ++l) std::memcpy(staging.data() + l * 4 * (kMaxSuperpR + 2), lp[l].effRad, 4 * (kMaxSuperpR + 2));
        } else {
            n = 4 * L * 8; staging.resize(n);
            for (size_t l = 0; l < L; ++l) {
                float v[8] = { lp[l].energyIdx, lp[l].energyScaleFact, lp[l].peakDepth, lp[l].entrySigmaX, lp[l].entrySigmaY,
                               (float)lp[l].afterLast, (float)lp[l].layerFirstPassive, 0.0f };
                std::memcpy(staging.data() + l * 32, v, 32);
            }
        }
    } else return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_fetch: unknown name " + nm);
    if (bytes_needed) *bytes_needed = n;
    if (!host_out) return RTD_OK;
    if (bytes < n) return fail(h, RTD_ERR_INVALID_ARG, "rtd_field_fetch: buffer too small");
    if (!staging.empty()) std::memcpy(host_out, staging.data(), n);
    else RTD_HIP(h, hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost));
    return RTD_OK;
}

// The reference-shaped call (kernel_wrapper.cu:381-1369): dose up (:542), beam loop (:601), dose down (:1318).
int rtd_compute(rtd_handle hh, const rtd_beam* beams, int n_beams, float* dose_inout, const uint32_t dose_dims[3], rtd_timing* timing) {
    auto* h = reinterpret_cast<rtd_handle_impl*>(hh);
    if (!h || !beams || n_beams < 0 || !dose_inout || !dose_dims) return RTD_ERR_INVALID_ARG;
    if (!h->dCt || !h->haveLuts) return fail(h, RTD_ERR_NOT_READY, "rtd_compute: set LUTs and CT first");
    RTD_HIP(h, hipSetDevice(h->device));
    const size_t nx = dose_dims[0], ny = dose_dims[1];
    const size_t n = nx * ny * dose_dims[2];
    float* dDose = nullptr;
    RTD_HIP(h, hipMalloc((void**)&dDose, n * sizeof(float)));
    // Every beam up to its BEV dose first; then the block of the dose volume that the beams can change — the bounding box of their
    // dose boxes — goes up, the transfers accumulate into it in beam order, and the same block comes down: the voxels outside
    // it are neither read nor written (the reference moves the whole volume both ways, :542 / :1318).
    std::vector<rtd_field> fields((size_t)n_beams, nullptr);
    int st = RTD_OK;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    for (int i = 0; i < n_beams && st == RTD_OK; ++i) {
        st = rtd_field_create(hh, &beams[i], dose_dims, &fields[(size_t)i]);
        if (st == RTD_OK) st = rtd_field_compute_bev(hh, fields[(size_t)i]);
        rtd_field_info fi;
        if (st == RTD_OK) st = rtd_field_wait_plan(hh, fields[(size_t)i], &fi, nullptr);
        if (st != RTD_OK) break;
        if (fi.dose_box_max[0] < fi.dose_box_min[0] || fi.dose_box_max[1] < fi.dose_box_min[1] || fi.dose_box_max[2] < fi.dose_box_min[2]) continue;
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (int)fi.dose_box_min[a]); hi[a] = std::max(hi[a], (int)fi.dose_box_max[a]); }
    }
    const bool haveBlock = hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];
    hipMemcpy3DParms p;
    std::memset(&p, 0, sizeof p);
    if (haveBlock) {
        if ((size_t)(hi[0] - lo[0] + 1) * 2 >= nx) { lo[0] = 0; hi[0] = (int)nx - 1; }   // wide blocks travel as whole rows
        p.srcPtr = make_hipPitchedPtr(dose_inout, nx * sizeof(float), nx, ny);
        p.dstPtr = make_hipPitchedPtr(dDose, nx * sizeof(float), nx, ny);
        p.srcPos = p.dstPos = make_hipPos((size_t)lo[0] * sizeof(float), (size_t)lo[1], (size_t)lo[2]);
        p.extent = make_hipExtent((size_t)(hi[0] - lo[0] + 1) * sizeof(float), (size_t)(hi[1] - lo[1] + 1), (size_t)(hi[2] - lo[2] + 1));
        p.kind = hipMemcpyHostToDevice;
        if (st == RTD_OK) { const hipError_t e = hipMemcpy3DAsync(&p, h->stream); if (e != hipSuccess) { h->error = hipGetErrorString(e); st = RTD_ERR_HIP; } }
    }
    for (int i = 0; i < n_beams && st == RTD_OK; ++i) st = rtd_field_transfer(hh, fields[(size_t)i], dDose, nullptr, nullptr);
    // device-side errors (radius overflow) surface here; the caller's volume is written only when every beam succeeded
    for (int i = 0; i < n_beams && st == RTD_OK; ++i) st = rtd_field_finish(hh, fields[(size_t)i], timing ? &timing[i] : nullptr, nullptr);
    if (st == RTD_OK && haveBlock) {
        std::swap(p.srcPtr, p.dstPtr);
        p.kind = hipMemcpyDeviceToHost;
        hipError_t e = hipMemcpy3DAsync(&p, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { h->error = hipGetErrorString(e); st = RTD_ERR_HIP; }
    }
    (void)hipStreamSynchronize(h->stream);
    const std::string keep = h->error;
    for (rtd_field f : fields) if (f) rtd_field_release(hh, f);       // workspaces stay with the handle for the next call
    h->error = keep;
    (void)hipFree(dDose);
    return st;
}

}  // extern "C"
