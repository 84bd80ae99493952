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


inline void runWaterCube(const std::string& lutDir, const std::string& outDir, unsigned int n, unsigned int nLayers, int gpuId) {
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
    cudaWrapperProtons(&imVol, &doseVol, beams, ciddData, std::cout, gpuId);
    std::cout << "Done!\n\n";
    std::ofstream fout((outDir + "/dose.dat").c_str(), std::ios::out | std::ios::binary);
    fout.write(reinterpret_cast<const char*>(doseData.data()), doseData.size() * sizeof(float));
    fout.close();
    std::cout << "Written " << outDir << "/dose.dat with size " << dim.x << "x" << dim.y << "x" << dim.z << "\n\n";
    std::cout << "Max:" << *std::max_element(doseData.begin(), doseData.end()) << std::endl;
}
