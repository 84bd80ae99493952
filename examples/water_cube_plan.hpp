// water_cube_plan.hpp — the reference's WATER_CUBE_TEST plan (src/main.cu:39-99,192-216), shared by the two example drivers:
// builds the water cube, one G000 field of 33x33 spots x nLayers layers, calls cudaWrapperProtons (the C++ shim over the C ABI),
// writes <outDir>/dose.dat (raw float32, x fastest) and prints "Max:" like the reference. Throws std::runtime_error.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>

#include <sstream>

#include "rtd_dicom.hpp"
#include "rtd_plan.hpp"
#include "rtd_types.hpp"
#include "rtd_wrapper.hpp"

using namespace rtd_types;

static float findDecimalOrdered(const std::vector<float>& l, float v) {   // vector_find.h:128-144
    if (v >= l.back()) return float(l.size() - 1);
    if (v < l.front()) return 0.0f;
    size_t lo = 0, hi = l.size() - 1;
    while (hi - lo > 1) { size_t p = (hi + lo) / 2; if (l[p] <= v) lo = p; else hi = p; }
    return float(lo) + (v - l[lo]) / (l[lo + 1] - l[lo]);
}
static float vectorInterpolate(const std::vector<float>& l, float idx) {   // vector_interpolate.h:17-30
    if (idx <= 0.0f) return l.front();
    if (idx >= float(l.size() - 1)) return l.back();
    float ip; float d = std::modf(idx, &ip);
    size_t f = (size_t)ip;
    return l[f] + (l[f + 1] - l[f]) * d;
}


// The water cube and its image geometry (main.cu:39-43).
struct WaterCube {
    uint3 dim;
    std::vector<float> image, dose;
    Float3AffineTransform imIdxToWorld;
    explicit WaterCube(unsigned int n)
        : dim(make_uint3(n, n, n)), image((size_t)n * n * n, 1000.0f), dose((size_t)n * n * n, 0.0f),
          imIdxToWorld(Matrix3x3(256.0f / float(n), 256.0f / float(n), 256.0f / float(n)), make_float3(-128.0f, -128.0f, -256.0f + 150.0f)) {}
};

inline void writeDoseAndReport(const WaterCube& w, const std::string& outDir) {   // main.cu:211-216
    std::ofstream fout((outDir + "/dose.dat").c_str(), std::ios::out | std::ios::binary);
    fout.write(reinterpret_cast<const char*>(w.dose.data()), w.dose.size() * sizeof(float));
    fout.close();
    std::cout << "Written " << outDir << "/dose.dat with size " << w.dim.x << "x" << w.dim.y << "x" << w.dim.z << "\n\n";
    std::cout << "Max:" << *std::max_element(w.dose.begin(), w.dose.end()) << std::endl;
}

// The reference's WATER_CUBE_TEST field as a plan spot list (E, x, y, FWHM, meterset), layer by layer (main.cu:61-99): what an
// RT ion plan of that field would hold; rtd_plan::buildField turns it back into the BeamSettings of the built-in plan.
inline std::vector<rtd_plan::Spot> waterCubeSpots(const EnergyStruct& ciddData, unsigned int nLayers) {
    std::mt19937 rng(1234);   // the reference uses unseeded rand() (main.cu:80)
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<float> w((size_t)33 * 33 * nLayers);
    for (auto& v : w) v = 90.0f + 10.0f * u(rng);
    std::vector<rtd_plan::Spot> spots;
    float currentEnergy = 118.12f;
    const float energyStep = nLayers > 1 ? (172.51f - currentEnergy) / float(nLayers - 1) : 0.0f;
    const float sigmaToFwhm = (float)(2.0 * std::sqrt(2.0 * std::log(2.0)));
    for (unsigned int l = 0; l < nLayers; ++l) {
        const float peakDepth = vectorInterpolate(ciddData.peakDepths, findDecimalOrdered(ciddData.energiesPerU, currentEnergy));
        const float sigma = 2.3f + 290.0f / (peakDepth + 15.0f);
        for (unsigned int iy = 0; iy < 33; ++iy)
            for (unsigned int ix = 0; ix < 33; ++ix)
                spots.push_back(rtd_plan::Spot{currentEnergy, -48.0f + 3.0f * float(ix), -48.0f + 3.0f * float(iy), sigma * sigmaToFwhm,
                                               sigma * sigmaToFwhm, w[((size_t)l * 33 + iy) * 33 + ix]});
        currentEnergy += energyStep;
    }
    return spots;
}

// Spot-list text file: "key = value" lines (gantry_angle, isocenter x y z, source_dist x y, ray_spacing x y, tracer_steps,
// step_length, start_depth), '#' comments, and one "E x y fwhm_x fwhm_y meterset" row per spot in delivery order.
inline void writeSpotList(const std::string& path, const std::vector<rtd_plan::Spot>& spots, const rtd_plan::FieldGeometry& g) {
    std::ofstream o(path.c_str());
    if (!o) throw std::runtime_error("cannot write " + path);
    o.precision(9);
    o << "# E[MeV/u] x[mm] y[mm] fwhm_x[mm] fwhm_y[mm] meterset\n";
    o << "gantry_angle = " << g.gantryAngleDeg << "\nisocenter = " << g.isocenter.x << " " << g.isocenter.y << " " << g.isocenter.z << "\n";
    o << "source_dist = " << g.sourceDist.x << " " << g.sourceDist.y << "\n";
    for (const auto& s : spots) o << s.energy << " " << s.x << " " << s.y << " " << s.fwhmX << " " << s.fwhmY << " " << s.meterset << "\n";
}
inline void readSpotList(const std::string& path, std::vector<rtd_plan::Spot>& spots, rtd_plan::FieldGeometry& g) {
    std::ifstream in(path.c_str());
    if (!in) throw std::runtime_error("Failed to open " + path);
    auto num = [&](const std::string& t) -> float {
        if (t == "inf" || t == "+inf" || t == "Inf" || t == "INF") return std::numeric_limits<float>::infinity();
        size_t used = 0; float v = 0.0f;
        try { v = std::stof(t, &used); } catch (...) { used = 0; }
        if (used != t.size()) throw std::runtime_error(path + ": not a number: " + t);
        return v;
    };
    std::string line;
    int lineNo = 0;
    while (std::getline(in, line)) {
        ++lineNo;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        const size_t eq = line.find('=');
        std::istringstream is(eq == std::string::npos ? line : line.substr(eq + 1));
        std::vector<float> v; std::string t;
        while (is >> t) v.push_back(num(t));
        if (eq == std::string::npos) {
            if (v.empty()) continue;
            if (v.size() != 6) throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": a spot row has 6 numbers (E x y fwhm_x fwhm_y meterset)");
            spots.push_back(rtd_plan::Spot{v[0], v[1], v[2], v[3], v[4], v[5]});
            continue;
        }
        std::string key = line.substr(0, eq);
        key.erase(0, key.find_first_not_of(" \t")); key.erase(key.find_last_not_of(" \t") + 1);
        auto need = [&](size_t n) { if (v.size() != n) throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": " + key + " takes " + std::to_string(n) + " value(s)"); };
        if (key == "gantry_angle") { need(1); g.gantryAngleDeg = v[0]; }
        else if (key == "isocenter") { need(3); g.isocenter = make_float3(v[0], v[1], v[2]); }
        else if (key == "source_dist") { need(2); g.sourceDist = make_float2(v[0], v[1]); }
        else if (key == "ray_spacing") { need(2); g.raySpacing = make_float2(v[0], v[1]); }
        else if (key == "tracer_steps") { need(1); g.tracerSteps = (unsigned int)v[0]; }
        else if (key == "step_length") { need(1); g.stepLength = v[0]; }
        else if (key == "start_depth") { need(1); g.startDepth = v[0]; }
        else throw std::runtime_error(path + ":" + std::to_string(lineNo) + ": unknown key " + key);
    }
}

// Dose of the named beams of an RT Ion Plan on a DICOM CT series (what the reference's non-WATER_CUBE_TEST build is meant to do,
// main.cu:100-216): CT -> HU+1000 + imIdxToWorld (dicom_reader.cpp), beams -> BeamSettings (rtd_plan.hpp), dose grid = CT grid.
// startDepth / tracerSteps <= 0 mean "cover the CT along the beam axis".
inline void runDicomPlan(const std::string& lutDir, const std::string& outDir, const std::string& ctDir, const std::string& planFile,
                         const std::vector<std::string>& beamNamesIn, const std::vector<int>& gpuIds, float startDepth, int tracerSteps, std::vector<float>* doseOut = nullptr,
                         const rtd_options* opt = nullptr) {
    clock_t t = clock();
    EnergyStruct ciddData = energyReader(lutDir, /*waterCubeTest=*/false);
    std::cout << "Read energy matrix: " << static_cast<float>(clock() - t) / CLOCKS_PER_SEC << " seconds.\n\n";
    t = clock();
    rtd_dicom::CtVolume ct = rtd_dicom::readCtSeries(ctDir);
    std::cout << "Read image: " << static_cast<float>(clock() - t) / CLOCKS_PER_SEC << " seconds (series " << ct.seriesUid << ", " << ct.dim.x << "x" << ct.dim.y
              << "x" << ct.dim.z << ")\n\n";
    if (!ct.patientPosition.empty() && ct.patientPosition != "HFS") throw std::runtime_error("patient position " + ct.patientPosition + " is not supported (HFS only)");
    const rtd_dicom::File plan = rtd_dicom::readFile(planFile);
    std::vector<float> dose(ct.huPlus1000.size(), 0.0f);
    HostPinnedImage3D<float> doseVol(dose.data(), ct.dim), imVol(ct.huPlus1000.data(), ct.dim);
    std::vector<rtd_plan::BuiltField> fields;
    std::vector<std::vector<float>> weightStore;
    std::vector<std::unique_ptr<HostPinnedImage3D<float>>> weightImages;
    std::vector<BeamSettings> beams;
    for (size_t i = 0; i < beamNamesIn.size(); ++i) {
        std::cout << "Loading field " << i << " corresponding to beamname " << beamNamesIn[i] << std::endl;                    // main.cu:122
        rtd_dicom::PlanBeam b = rtd_dicom::readPlanBeam(plan, beamNamesIn[i]);
        std::cout << "Angles: " << b.beamLimitingDeviceAngleDeg << " " << b.geo.gantryAngleDeg << " " << b.patientSupportAngleDeg << " deg" << std::endl;  // :151
        std::cout << "IsoCenter: " << b.geo.isocenter.x << " " << b.geo.isocenter.y << " " << b.geo.isocenter.z << " mm" << std::endl;
        std::cout << "ImgPixels: " << ct.dim.x << " " << ct.dim.y << " " << ct.dim.z << std::endl;
        b.geo.hasGantryToWorld = true;
        b.geo.gantryToWorld = rtd_dicom::gantryToPatientHfs(b.geo.gantryAngleDeg, b.patientSupportAngleDeg, b.geo.isocenter);
        rtd_dicom::tracerRange(ct, b.geo.gantryToWorld, b.geo.stepLength, b.geo.startDepth, b.geo.tracerSteps);
        if (startDepth > 0.0f) b.geo.startDepth = startDepth;
        if (tracerSteps > 0) b.geo.tracerSteps = (unsigned int)tracerSteps;
        fields.push_back(rtd_plan::buildField(b.spots, b.geo, ct.imIdxToWorld, ct.imIdxToWorld));
        std::cout << b.spots.size() << " spots in " << fields.back().dims.z << " layer(s) on a " << fields.back().dims.x << "x" << fields.back().dims.y
                  << " spot grid; tracer: " << b.geo.tracerSteps << " steps of " << b.geo.stepLength << " mm from " << b.geo.startDepth << " mm\n\n";
    }
    for (auto& f : fields) {
        weightStore.push_back(f.weights);
        weightImages.emplace_back(new HostPinnedImage3D<float>(weightStore.back().data(), f.dims));
        beams.push_back(f.beamSettings(weightImages.back().get()));
    }
    std::cout << "Executing code on GPU...\n\n";
    cudaWrapperProtons(&imVol, &doseVol, beams, ciddData, std::cout, gpuIds, opt);
    std::cout << "Done!\n\n";
    std::ofstream fout((outDir + "/dose.dat").c_str(), std::ios::out | std::ios::binary);                                       // main.cu:211-216
    fout.write(reinterpret_cast<const char*>(dose.data()), dose.size() * sizeof(float));
    fout.close();
    std::cout << "Written " << outDir << "/dose.dat with size " << ct.dim.x << "x" << ct.dim.y << "x" << ct.dim.z << "\n\n";
    std::cout << "Max:" << *std::max_element(dose.begin(), dose.end()) << std::endl;
    if (doseOut) *doseOut = dose;
}

// Dose of a plan given as a spot list on the water cube (CT input proper is SURVEY section 8 row f3).
inline void runSpotListOnWaterCube(const std::string& lutDir, const std::string& outDir, unsigned int n, const std::string& spotFile, const std::vector<int>& gpuIds, const rtd_options* opt = nullptr) {
    EnergyStruct ciddData = energyReader(lutDir, /*waterCubeTest=*/true);
    WaterCube w(n);
    std::vector<rtd_plan::Spot> spots;
    rtd_plan::FieldGeometry geo;
    readSpotList(spotFile, spots, geo);
    const rtd_plan::BuiltField f = rtd_plan::buildField(spots, geo, w.imIdxToWorld, w.imIdxToWorld);
    std::cout << "Plan: " << spots.size() << " spots in " << f.dims.z << " layer(s) on a " << f.dims.x << "x" << f.dims.y << " spot grid, pitch "
              << f.spotIdxToGantry.getDelta().x << " x " << f.spotIdxToGantry.getDelta().y << " mm, gantry " << geo.gantryAngleDeg << " deg\n\n";
    HostPinnedImage3D<float> doseVol(w.dose.data(), w.dim), imVol(w.image.data(), w.dim);
    std::vector<float> weights = f.weights;
    HostPinnedImage3D<float> spotWeights(weights.data(), f.dims);
    std::vector<BeamSettings> beams;
    beams.push_back(f.beamSettings(&spotWeights));
    std::cout << "Executing code on GPU...\n\n";
    cudaWrapperProtons(&imVol, &doseVol, beams, ciddData, std::cout, gpuIds, opt);
    std::cout << "Done!\n\n";
    writeDoseAndReport(w, outDir);
}

inline void runWaterCube(const std::string& lutDir, const std::string& outDir, unsigned int n, unsigned int nLayers, const std::vector<int>& gpuIds, const rtd_options* opt = nullptr) {
    EnergyStruct ciddData = energyReader(lutDir, /*waterCubeTest=*/true);
    const uint3 dim = make_uint3(n, n, n);
    const size_t N = (size_t)n * n * n;
    const float voxel = 256.0f / float(n);
    std::vector<float> imageData(N, 1000.0f), doseData(N, 0.0f);
    Float3AffineTransform imIdxToWorld(Matrix3x3(voxel, voxel, voxel), make_float3(-128.0f, -128.0f, -256.0f + 150.0f));   // main.cu:43
    Float3AffineTransform worldToImIdx = imIdxToWorld.inverse();
    const float fInf = std::numeric_limits<float>::infinity();
    const float2 sourceDist = make_float2(fInf, fInf);
    Float3AffineTransform gantryToImIdx = concatFloat3AffineTransform(Float3AffineTransform().inverse(), worldToImIdx);      // main.cu:55-57
    Float3IdxTransform fanIdxToFan(make_float3(3.0f, 3.0f, -1.0f), make_float3(-48.0f, -48.0f, 128.0f));                     // main.cu:62

    HostPinnedImage3D<float> doseVol(doseData.data(), dim), imVol(imageData.data(), dim);
    const uint3 beamDim = make_uint3(33, 33, nLayers);
    std::vector<float> beamData((size_t)33 * 33 * nLayers);
    std::mt19937 rng(1234);   // the reference uses unseeded rand() (main.cu:80)
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    for (auto& w : beamData) w = 90.0f + 10.0f * u(rng);
    HostPinnedImage3D<float> spotWeights(beamData.data(), beamDim);

    float currentEnergy = 118.12f;
    const float lastEnergy = 172.51f;
    const float energyStep = nLayers > 1 ? (lastEnergy - currentEnergy) / float(nLayers - 1) : 0.0f;
    std::vector<float> energiesPerU(nLayers);
    std::vector<float2> sigmas(nLayers);
    for (unsigned int i = 0; i < nLayers; ++i) {
        energiesPerU[i] = currentEnergy;
        const float peakDepth = vectorInterpolate(ciddData.peakDepths, findDecimalOrdered(ciddData.energiesPerU, currentEnergy));
        sigmas[i].x = sigmas[i].y = 2.3f + 290.0f / (peakDepth + 15.0f);   // main.cu:92-95
        currentEnergy += energyStep;
    }
    std::vector<BeamSettings> beams;
    beams.push_back(BeamSettings(&spotWeights, energiesPerU, sigmas, make_float2(1.0f, 1.0f), 512, sourceDist, fanIdxToFan,
                                 gantryToImIdx, gantryToImIdx));                                                             // main.cu:192-197
    std::cout << "Executing code on GPU...\n\n";
    cudaWrapperProtons(&imVol, &doseVol, beams, ciddData, std::cout, gpuIds, opt);
    std::cout << "Done!\n\n";
    std::ofstream fout((outDir + "/dose.dat").c_str(), std::ios::out | std::ios::binary);
    fout.write(reinterpret_cast<const char*>(doseData.data()), doseData.size() * sizeof(float));
    fout.close();
    std::cout << "Written " << outDir << "/dose.dat with size " << dim.x << "x" << dim.y << "x" << dim.z << "\n\n";
    std::cout << "Max:" << *std::max_element(doseData.begin(), doseData.end()) << std::endl;
}
