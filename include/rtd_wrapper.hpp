// rtd_wrapper.hpp — C++ shim with the reference's entry point on top of the C ABI (include/rtd.h).
//
//   void cudaWrapperProtons(HostPinnedImage3D<float>* imVol, HostPinnedImage3D<float>* doseVol,
//                           const std::vector<BeamSettings> beams, const EnergyStruct iddData,
//                           std::ostream& outStream);                    (reference src/kernel_wrapper.cuh:161)
//
// The template is written against the ACCESSOR NAMES of the reference's own host types
// (src/host_image_3d.cuh, src/beam_settings.h, src/energy_struct.h, src/float3_*_transform.cuh), so inside the
// reference tree it compiles against those headers unchanged; outside it compiles against the stand-alone
// mirrors in include/rtd_types.hpp. It marshals everything into the PODs of rtd.h and calls rtd_compute.
//
// Behaviour kept from the reference: dose is accumulated into doseVol's buffer; failures throw
// std::runtime_error (cuda_errchk.cu:11-22; the radius overflow text of kernel_wrapper.cu:965); the timing
// line "Total global execution time (excluding GPU initialisation)" goes to outStream (:1360).
// Behaviour NOT kept unless `referenceOwnership` is true: the reference deletes imVol, doseVol and every
// beam's weight image and resets the device before returning (kernel_wrapper.cu:856,1366-1368).
#pragma once

#include <ostream>
#include <stdexcept>
#include <string>
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

template <class Energy>
inline rtd_luts toLuts(const Energy& e) {
    rtd_luts l{};
    l.n_energy_samples = e.nEnergySamples; l.n_energies = e.nEnergies;
    l.energies_per_u = e.energiesPerU.data(); l.peak_depths = e.peakDepths.data(); l.scale_facts = e.scaleFacts.data();
    l.cidd_matrix = e.ciddMatrix.data();
    l.n_density_samples = e.nDensitySamples; l.density_scale_fact = e.densityScaleFact; l.density_vector = e.densityVector.data();
    l.n_sp_samples = e.nSpSamples; l.sp_scale_fact = e.spScaleFact; l.sp_vector = e.spVector.data();
    l.n_rrl_samples = e.nRRlSamples; l.rrl_scale_fact = e.rRlScaleFact; l.rrl_vector = e.rRlVector.data();
    return l;
}

struct HandleGuard {
    rtd_handle h = nullptr;
    ~HandleGuard() { if (h) rtd_destroy(h); }
};

}  // namespace rtd_shim

template <class Image, class Beam, class Energy>
void cudaWrapperProtons(Image* const imVol, Image* const doseVol, std::vector<Beam> beams, const Energy& iddData,
                        std::ostream& outStream, int gpuId = 0, const rtd_options* options = nullptr,
                        bool referenceOwnership = false) {
    using namespace rtd_shim;
    HandleGuard g;
    if (rtd_create(gpuId, &g.h) != RTD_OK) throw std::runtime_error(rtd_global_error());
    auto check = [&](int st) { if (st != RTD_OK) throw std::runtime_error(rtd_last_error(g.h)); };
    if (options) check(rtd_set_options(g.h, options));
    const rtd_luts luts = toLuts(iddData);
    check(rtd_set_luts(g.h, &luts));
    const uint32_t imDims[3] = { imVol->getDims().x, imVol->getDims().y, imVol->getDims().z };
    check(rtd_set_ct(g.h, imVol->getImData(), imDims));

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
    std::vector<rtd_timing> timing(beams.size() ? beams.size() : 1);
    check(rtd_compute(g.h, pods.data(), (int)pods.size(), doseVol->getImData(), doseDims, timing.data()));
    float total = 0.0f;
    for (size_t i = 0; i < beams.size(); ++i) total += timing[i].total_ms;
    outStream << "    Total global execution time (excluding GPU initialisation): " << total << " ms.\n\n";
    if (referenceOwnership) {   // kernel_wrapper.cu:856,1366-1367
        for (auto& b : beams) delete b.getWeights();
        delete imVol;
        delete doseVol;
    }
}
