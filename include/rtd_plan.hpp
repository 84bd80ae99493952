// rtd_plan.hpp — RT ion plan spot list -> BeamSettings.
//
// The reference reads the spot sequence of a field from the plan — per spot (E, X, Y, FWHMx, FWHMy, meterset), grouped in
// energy layers (src/main.cu:150-181) — prints it, and stops there: its BeamSettings is then built from an all-zero
// weight array and empty energy / sigma vectors (main.cu:184-197, "to be deleted by cudaWrapperProtons"). This header
// is the missing step, written against the same host types (include/rtd_types.hpp here; beam_settings.h,
// float3_affine_transform.cuh, float3_idx_transform.cuh in the reference tree):
//
//   spots of a layer  ->  one weight map on a regular spot grid  (BeamSettings weights [layer][y][x], x fastest, beam_settings.h:21)
//   layer energies    ->  beamEnergies (MeV/u), in sequence order
//   spot FWHM         ->  spotSigmas = FWHM / (2 sqrt(2 ln 2)), meterset-weighted mean per layer (mm, at iso in air)
//   spot grid         ->  spotIdxToGantry  (delta = pitch x, pitch y, -step length; offset = first spot x, y, start depth)
//   gantry angle, isocentre, image geometry -> gantryToImIdx = worldToImIdx o gantryToWorld  (main.cu:52-57)
//
// Conventions (the reference never got far enough to fix them; these are the ones of this repository's test scenarios,
// SURVEY C4): gantry coordinates have the beam along -z starting at z = startDepth, spot (x, y) in the plane through the
// isocentre; gantryToWorld is the rotation about the world Y axis by the gantry angle followed by the translation to the
// isocentre.
#pragma once
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtd_types.hpp"

namespace rtd_plan {

using namespace rtd_types;

struct Spot {                       // one control-point spot, as printed by the reference (main.cu:169-175): (E,X,Y,Sx,Sy,W)
    float energy;                   // MeV/u
    float x, y;                     // mm in the isocentre plane, gantry coordinates
    float fwhmX, fwhmY;             // mm, in air at the isocentre
    float meterset;                 // weight
};

struct FieldGeometry {
    float gantryAngleDeg = 0.0f;
    float3 isocenter{0.0f, 0.0f, 0.0f};                               // world mm
    float2 sourceDist{INFINITY, INFINITY};                            // virtual source-axis distances (mm); inf = parallel
    float2 raySpacing{1.0f, 1.0f};                                    // mm at the isocentre
    unsigned int tracerSteps = 512;
    float stepLength = 1.0f;                                          // mm per tracer step (beam travels along -z)
    float startDepth = 128.0f;                                        // gantry z of tracer step 0 (mm upstream of the isocentre)
    bool hasGantryToWorld = false;                                    // true: gantryToWorld below replaces the rotation-about-Y convention
    Float3AffineTransform gantryToWorld;                              //       (rtd_dicom.hpp sets the IEC 61217 / DICOM patient one)
};

// Owns the arrays a BeamSettings points to.
struct BuiltField {
    std::vector<float> weights;                                       // [layer][ny][nx]
    uint3 dims{0, 0, 0};                                              // (nx, ny, layers)
    std::vector<float> energies;
    std::vector<float2> sigmas;
    Float3IdxTransform spotIdxToGantry;
    Float3AffineTransform gantryToImIdx, gantryToDoseIdx;
    FieldGeometry geo;
    // the view the dose engine takes; `image` must outlive the BeamSettings
    BeamSettings beamSettings(HostPinnedImage3D<float>* image) const {
        return BeamSettings(image, energies, sigmas, geo.raySpacing, geo.tracerSteps, geo.sourceDist, spotIdxToGantry, gantryToImIdx,
                            gantryToDoseIdx);
    }
};

namespace detail {
// positions on a lattice first + i * pitch: returns (first, pitch, count); throws if a position is off the lattice
inline void lattice(std::vector<float> pos, const char* axis, float& first, float& pitch, unsigned int& count) {
    std::sort(pos.begin(), pos.end());
    std::vector<float> uniq;
    for (float p : pos) if (uniq.empty() || std::fabs(p - uniq.back()) > 1e-3f) uniq.push_back(p);
    first = uniq.front();
    if (uniq.size() == 1) { pitch = 1.0f; count = 1; return; }
    pitch = uniq[1] - uniq[0];
    for (size_t i = 2; i < uniq.size(); ++i) pitch = std::min(pitch, uniq[i] - uniq[i - 1]);
    const float span = uniq.back() - first;
    count = (unsigned int)std::lround(span / pitch) + 1;
    pitch = count > 1 ? span / float(count - 1) : pitch;              // spread rounding over the whole span
    for (float p : uniq) {
        const float idx = (p - first) / pitch;
        if (std::fabs(idx - std::round(idx)) > 1e-2f)
            throw std::runtime_error(std::string("spot positions are not on a regular grid along ") + axis);
    }
}
inline Matrix3x3 rotationY(float deg) {
    double a = (double)deg * 3.14159265358979323846 / 180.0, c = std::cos(a), s = std::sin(a);
    if (std::fabs(c - std::round(c)) < 1e-12) c = std::round(c);     // exact quarter turns stay axis-aligned
    if (std::fabs(s - std::round(s)) < 1e-12) s = std::round(s);
    return Matrix3x3(make_float3((float)c, 0.0f, (float)s), make_float3(0.0f, 1.0f, 0.0f), make_float3((float)-s, 0.0f, (float)c));
}
}  // namespace detail

// Spots in delivery order (layers are runs of equal energy, as in the plan's control-point sequence).
inline BuiltField buildField(const std::vector<Spot>& spots, const FieldGeometry& geo, const Float3AffineTransform& imIdxToWorld,
                             const Float3AffineTransform& doseIdxToWorld) {
    if (spots.empty()) throw std::runtime_error("empty spot list");
    BuiltField f;
    f.geo = geo;
    // layers: runs of equal energy in sequence order
    std::vector<unsigned int> layerOf(spots.size());
    for (size_t i = 0; i < spots.size(); ++i) {
        if (!(spots[i].energy > 0.0f) || !(spots[i].meterset >= 0.0f) || !(spots[i].fwhmX > 0.0f) || !(spots[i].fwhmY > 0.0f))
            throw std::runtime_error("spot " + std::to_string(i) + ": energy and FWHM must be positive, meterset non-negative");
        if (f.energies.empty() || std::fabs(spots[i].energy - f.energies.back()) > 1e-4f * f.energies.back()) f.energies.push_back(spots[i].energy);
        layerOf[i] = (unsigned int)f.energies.size() - 1;
    }
    const unsigned int nLayers = (unsigned int)f.energies.size();
    // one regular grid for all layers of the field
    std::vector<float> xs(spots.size()), ys(spots.size());
    for (size_t i = 0; i < spots.size(); ++i) { xs[i] = spots[i].x; ys[i] = spots[i].y; }
    float x0, y0, px, py; unsigned int nx, ny;
    detail::lattice(xs, "x", x0, px, nx);
    detail::lattice(ys, "y", y0, py, ny);
    f.dims = make_uint3(nx, ny, nLayers);
    f.weights.assign((size_t)nx * ny * nLayers, 0.0f);
    std::vector<double> wSum(nLayers, 0.0), sx(nLayers, 0.0), sy(nLayers, 0.0);
    std::vector<unsigned int> cnt(nLayers, 0);
    for (size_t i = 0; i < spots.size(); ++i) {
        const unsigned int l = layerOf[i];
        const unsigned int ix = (unsigned int)std::lround((spots[i].x - x0) / px), iy = (unsigned int)std::lround((spots[i].y - y0) / py);
        f.weights[((size_t)l * ny + iy) * nx + ix] += spots[i].meterset;
        wSum[l] += spots[i].meterset; sx[l] += (double)spots[i].meterset * spots[i].fwhmX; sy[l] += (double)spots[i].meterset * spots[i].fwhmY;
        ++cnt[l];
    }
    const double fwhmToSigma = 1.0 / (2.0 * std::sqrt(2.0 * std::log(2.0)));
    f.sigmas.resize(nLayers);
    for (unsigned int l = 0; l < nLayers; ++l) {
        double mx, my;
        if (wSum[l] > 0.0) { mx = sx[l] / wSum[l]; my = sy[l] / wSum[l]; }
        else {                                                        // a layer of zero weights: plain mean
            mx = my = 0.0;
            for (size_t i = 0; i < spots.size(); ++i) if (layerOf[i] == l) { mx += spots[i].fwhmX; my += spots[i].fwhmY; }
            mx /= cnt[l]; my /= cnt[l];
        }
        f.sigmas[l] = make_float2((float)(mx * fwhmToSigma), (float)(my * fwhmToSigma));
    }
    f.spotIdxToGantry = Float3IdxTransform(make_float3(px, py, -geo.stepLength), make_float3(x0, y0, geo.startDepth));
    const Float3AffineTransform gantryToWorld = geo.hasGantryToWorld ? geo.gantryToWorld
                                                                      : Float3AffineTransform(detail::rotationY(geo.gantryAngleDeg), geo.isocenter);
    f.gantryToImIdx = concatFloat3AffineTransform(gantryToWorld, imIdxToWorld.inverse());       // main.cu:55-57
    f.gantryToDoseIdx = concatFloat3AffineTransform(gantryToWorld, doseIdxToWorld.inverse());
    return f;
}

}  // namespace rtd_plan
