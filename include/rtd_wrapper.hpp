// rtd_wrapper.hpp — C++ shim with the reference's entry point on top of the C ABI (include/rtd.h).
//
//   void cudaWrapperProtons(HostPinnedImage3D<float>* imVol, HostPinnedImage3D<float>* doseVol,
//                           const std::vector<BeamSettings> beams, const EnergyStruct iddData,
//                           std::ostream& outStream);                    (reference src/kernel_wrapper.cuh:161)
//
// The template is written against the ACCESSOR NAMES of the reference's own host types
// (src/host_image_3d.cuh, src/beam_settings.h, src/energy_struct.h, src/float3_*_transform.cuh), so inside the
// reference tree it compiles against those headers unchanged; outside it compiles against the stand-alone
// mirrors in include/rtd_types.hpp. It marshals everything into the PODs of rtd.h and calls rtd_plan_compute.
//
// Behaviour kept from the reference: dose is accumulated into doseVol's buffer; failures throw
// std::runtime_error (cuda_errchk.cu:11-22; the radius overflow text of kernel_wrapper.cu:965); the log text goes to
// outStream: "Total global execution time (excluding GPU initialisation)" (:1360) measures what the reference measures —
// from after context creation (:410-414) over the CT / LUT upload, the beam loop and the dose download to the last free
// (:1356-1360) — and with options->fine_grained_timing the per-field bucket lines of :598,1298-1307,1325,1349-1352.
// New: gpuIds lists the devices to use (the reference parses --gpu_id and ignores it): with several ids the beams are
// computed on several GPUs (rtd_plan_*, include/rtd.h), with the same result bit for bit.
// Behaviour NOT kept unless `referenceOwnership` is true: the reference deletes imVol, doseVol and every
// beam's weight image and resets the device before returning (kernel_wrapper.cu:856,1366-1368).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "rtd.h"

namespace rtd_shim {

template <class Affine>
inline rtd_affine toAffine(const Affine& t) {
    rtd_affine a;
    const auto m = t.getMatrix();
    const auto r0 = m.row0(), r1 = m.row1(), r2 = m.row2();
    const auto v = t.getOffset();
    a.m[0] = r0.x; a.m[1] = r0.y; a.m[2] = r0.z;
    a.m[3] = r1.x; a.m[4] = r1.y; a.m[5] = r1.z;
    a.m[6] = r2.x; a.m[7] = r2.y; a.m[8] = r2.z;
    a.v[0] = v.x; a.v[1] = v.y; a.v[2] = v.z;
    return a;
}

template <class Idx>
inline rtd_idx_transform toIdx(const Idx& t) {
    rtd_idx_transform r;
    const auto d = t.getDelta(), o = t.getOffset();
    r.delta[0] = d.x; r.delta[1] = d.y; r.delta[2] = d.z;
    r.offset[0] = o.x; r.offset[1] = o.y; r.offset[2] = o.z;
    return r;
}

// the NUCLEAR_CORR members exist in the reference's EnergyStruct only when it is built with that option (energy_struct.h:33-36)
template <class Energy, class = void> struct HasNuclearTables : std::false_type {};
template <class Energy> struct HasNuclearTables<Energy, decltype((void)std::declval<const Energy&>().nucWeightMatrix, void())> : std::true_type {};
template <class Energy>
inline void nuclearTables(const Energy& e, rtd_luts& l, std::true_type) {
    if (!e.nucWeightMatrix.empty() && !e.nucSqSigmaMatrix.empty()) { l.nuc_weight_matrix = e.nucWeightMatrix.data(); l.nuc_sq_sigma_matrix = e.nucSqSigmaMatrix.data(); }
}
template <class Energy>
inline void nuclearTables(const Energy&, rtd_luts&, std::false_type) {}

template <class Energy>
inline rtd_luts toLuts(const Energy& e) {
    rtd_luts l{};
    nuclearTables(e, l, HasNuclearTables<Energy>{});
    l.n_energy_samples = e.nEnergySamples; l.n_energies = e.nEnergies;
    l.energies_per_u = e.energiesPerU.data(); l.peak_depths = e.peakDepths.data(); l.scale_facts = e.scaleFacts.data();
    l.cidd_matrix = e.ciddMatrix.data();
    l.n_density_samples = e.nDensitySamples; l.density_scale_fact = e.densityScaleFact; l.density_vector = e.densityVector.data();
    l.n_sp_samples = e.nSpSamples; l.sp_scale_fact = e.spScaleFact; l.sp_vector = e.spVector.data();
    l.n_rrl_samples = e.nRRlSamples; l.rrl_scale_fact = e.rRlScaleFact; l.rrl_vector = e.rRlVector.data();
    return l;
}

// RTD_DUMP_CALL=<file>: the marshalled call (CT, dims, every rtd_beam with its arrays) is written to <file> before the
// computation, so that a checker can be run on exactly what crossed the boundary (tests/test_plan_import.py).
inline void dumpCall(const char* path, const float* ct, const uint32_t imDims[3], const uint32_t doseDims[3], const std::vector<rtd_beam>& beams) {
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return;
    const uint32_t head[8] = { 0x52544443u /* "RTDC" */, imDims[0], imDims[1], imDims[2], doseDims[0], doseDims[1], doseDims[2], (uint32_t)beams.size() };
    std::fwrite(head, sizeof head, 1, f);
    std::fwrite(ct, sizeof(float), (size_t)imDims[0] * imDims[1] * imDims[2], f);
    for (const rtd_beam& b : beams) {
        const uint32_t dims[4] = { b.spot_nx, b.spot_ny, b.n_layers, b.tracer_steps };
        std::fwrite(dims, sizeof dims, 1, f);
        std::fwrite(b.ray_spacing, sizeof(float), 2, f);
        std::fwrite(b.source_dist, sizeof(float), 2, f);
        std::fwrite(&b.spot_idx_to_gantry, sizeof b.spot_idx_to_gantry, 1, f);
        std::fwrite(&b.gantry_to_im_idx, sizeof b.gantry_to_im_idx, 1, f);
        std::fwrite(&b.gantry_to_dose_idx, sizeof b.gantry_to_dose_idx, 1, f);
        std::fwrite(b.energies, sizeof(float), b.n_layers, f);
        std::fwrite(b.spot_sigmas, sizeof(float), 2 * (size_t)b.n_layers, f);
        std::fwrite(b.spot_weights, sizeof(float), (size_t)b.spot_nx * b.spot_ny * b.n_layers, f);
    }
    std::fclose(f);
}

struct PlanGuard {
    rtd_plan_t p = nullptr;
    ~PlanGuard() { if (p) rtd_plan_destroy(p); }
};

inline double wallMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace rtd_shim

template <class Image, class Beam, class Energy>
void cudaWrapperProtons(Image* const imVol, Image* const doseVol, std::vector<Beam> beams, const Energy& iddData,
                        std::ostream& outStream, const std::vector<int>& gpuIds, const rtd_options* options = nullptr,
                        bool referenceOwnership = false) {
    using namespace rtd_shim;
    PlanGuard g;
    if (gpuIds.empty()) throw std::runtime_error("cudaWrapperProtons: empty device list");
    if (rtd_plan_create(gpuIds.data(), (int)gpuIds.size(), &g.p) != RTD_OK) throw std::runtime_error(rtd_global_error());
    auto check = [&](int st) { if (st != RTD_OK) throw std::runtime_error(rtd_plan_last_error(g.p)); };
    const bool fine = options && options->fine_grained_timing;
    const double tStart = wallMs();                                  // device contexts exist: kernel_wrapper.cu:410-414
    if (options) check(rtd_plan_set_options(g.p, options));
    const rtd_luts luts = toLuts(iddData);
    check(rtd_plan_set_luts(g.p, &luts));
    const uint32_t imDims[3] = { imVol->getDims().x, imVol->getDims().y, imVol->getDims().z };
    // (deferred: imVol outlives the call, so each beam uploads only the box of the CT its rays cross, in front of its tracer)
    check(rtd_plan_set_ct_deferred(g.p, imVol->getImData(), imDims));
    const double tBound = wallMs();
    if (fine) outStream << "    Copy data to GPU and bind to textures: " << (tBound - tStart) << " ms\n\n";   // :598

    std::vector<rtd_beam> pods(beams.size());
    std::vector<std::vector<float>> sigmas(beams.size());
    for (size_t i = 0; i < beams.size(); ++i) {
        Beam& b = beams[i];
        rtd_beam& p = pods[i];
        p.spot_weights = b.getWeights()->getImData();
        p.spot_nx = b.getWeights()->getDims().x; p.spot_ny = b.getWeights()->getDims().y; p.n_layers = b.getWeights()->getDims().z;
        p.energies = b.getEnergies().data();
        for (const auto& s : b.getSpotSigmas()) { sigmas[i].push_back(s.x); sigmas[i].push_back(s.y); }
        p.spot_sigmas = sigmas[i].data();
        p.ray_spacing[0] = b.getRaySpacing().x; p.ray_spacing[1] = b.getRaySpacing().y;
        p.tracer_steps = b.getSteps();
        p.source_dist[0] = b.getSourceDist().x; p.source_dist[1] = b.getSourceDist().y;
        p.spot_idx_to_gantry = toIdx(b.getSpotIdxToGantry());
        p.gantry_to_im_idx = toAffine(b.getGantryToImIdx());
        p.gantry_to_dose_idx = toAffine(b.getGantryToDoseIdx());
        if (b.getEnergies().size() != p.n_layers || b.getSpotSigmas().size() != p.n_layers)
            throw std::runtime_error("BeamSettings: energies / sigmas / weight layers disagree");
    }
    const uint32_t doseDims[3] = { doseVol->getDims().x, doseVol->getDims().y, doseVol->getDims().z };
    if (const char* dump = std::getenv("RTD_DUMP_CALL")) dumpCall(dump, imVol->getImData(), imDims, doseDims, pods);
    std::vector<rtd_timing> timing(beams.size() ? beams.size() : 1);
    rtd_plan_timing pt{};
    check(rtd_plan_compute(g.p, pods.data(), (int)pods.size(), doseVol->getImData(), doseDims, timing.data(), &pt));
    const double tEnd = wallMs();
    if (fine) {
        // the reference's FINE_GRAINED_TIMING text (kernel_wrapper.cu:604,1298-1307,1325,1349-1352). Buckets that do not exist
        // here (per-beam allocation, the BEV -> texture copy) are reported as 0; the stage times are device times of the kernels.
        float approx = (float)(tBound - tStart);
        for (size_t i = 0; i < beams.size(); ++i) {
            const rtd_timing& t = timing[i];
            outStream << "    Calculating field no. " << i << "\n\n";
            outStream << "        Allocate memory and set up parameters: " << 0.0f << " ms\n";
            outStream << "        Time to trace " << t.ray_dims[0] << "x" << t.ray_dims[1] << " rays " << t.steps << " steps: " << t.raytracing_ms << " ms\n";
            outStream << "        Time preparing data for loop over energies: " << t.prepare_energy_loop_ms << " ms\n";
            outStream << "        Time depositing IDD and calculating sigma " << t.n_layers << " time(s): " << t.fill_idd_sigma_ms << " ms\n";
            outStream << "        Time preparing for superposition " << t.n_layers << " time(s): " << t.prepare_superp_ms << " ms\n";
            outStream << "        Time executing superposition " << t.n_layers << " time(s): " << t.superp_ms << " ms\n";
            outStream << "        Copy dose distribution to texture memory: " << 0.0f << " ms\n";
            outStream << "        Kernel time to transform " << t.transfer_voxels << " voxels: " << t.transforming_ms << " ms\n\n";
            approx += t.total_ms;
        }
        outStream << "    Time to copy dose back to host: " << pt.download_ms << " ms\n";
        outStream << "    Time spent freeing memory: " << 0.0f << " ms.\n\n";
        approx += pt.upload_ms + pt.download_ms;
        outStream << "    Approximate total execution time (excluding GPU initialisation): " << approx << " ms.\n";
        outStream << "    (Remove FINE_GRAINED_TIMING flag for more accurate total time, reports up to 30 ms longer execution time)\n\n";
    } else {
        outStream << "    Total global execution time (excluding GPU initialisation): " << (tEnd - tStart) << " ms.\n\n";
    }
    if (referenceOwnership) {   // kernel_wrapper.cu:856,1366-1367
        for (auto& b : beams) delete b.getWeights();
        delete imVol;
        delete doseVol;
    }
}

// The reference's signature plus a single device id (its --gpu_id, config.cpp:13-15).
template <class Image, class Beam, class Energy>
void cudaWrapperProtons(Image* const imVol, Image* const doseVol, std::vector<Beam> beams, const Energy& iddData,
                        std::ostream& outStream, int gpuId = 0, const rtd_options* options = nullptr,
                        bool referenceOwnership = false) {
    cudaWrapperProtons(imVol, doseVol, std::move(beams), iddData, outStream, std::vector<int>{gpuId}, options, referenceOwnership);
}
