// rtd_types.hpp — stand-alone mirrors of the reference's host value types, for C++ callers that do not have the
// reference tree (or CUDA's float2/float3/uint3) at hand. Same class names, constructors and accessors:
//   Matrix3x3 (src/matrix_3x3.cuh), Float3AffineTransform (src/float3_affine_transform.cuh),
//   Float3IdxTransform (src/float3_idx_transform.cuh), HostPinnedImage3D<T> (src/host_image_3d.cuh: non-owning, the
//   buffer is page-locked for the lifetime of the object like the reference's cudaHostRegister), BeamSettings (src/beam_settings.h),
//   EnergyStruct (src/energy_struct.h), energyReader (src/energy_reader.h; text layout of SURVEY Appendix A).
#pragma once

#include <cmath>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtd.h"

namespace rtd_types {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct uint3 { unsigned int x, y, z; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { return uint3{x, y, z}; }

class Matrix3x3 {
public:
    Matrix3x3(float3 a0, float3 a1, float3 a2) : r0(a0), r1(a1), r2(a2) {}
    Matrix3x3(float s00, float s11, float s22) : r0{s00, 0, 0}, r1{0, s11, 0}, r2{0, 0, s22} {}
    float3 row0() const { return r0; }
    float3 row1() const { return r1; }
    float3 row2() const { return r2; }
    float3 operator*(float3 a) const {
        return float3{r0.x * a.x + r0.y * a.y + r0.z * a.z, r1.x * a.x + r1.y * a.y + r1.z * a.z, r2.x * a.x + r2.y * a.y + r2.z * a.z};
    }
    Matrix3x3 operator*(const Matrix3x3& m) const {
        auto col = [&](int c) { return float3{c == 0 ? m.r0.x : c == 1 ? m.r0.y : m.r0.z, c == 0 ? m.r1.x : c == 1 ? m.r1.y : m.r1.z,
                                               c == 0 ? m.r2.x : c == 1 ? m.r2.y : m.r2.z}; };
        auto d = [](float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
        return Matrix3x3(float3{d(r0, col(0)), d(r0, col(1)), d(r0, col(2))}, float3{d(r1, col(0)), d(r1, col(1)), d(r1, col(2))},
                         float3{d(r2, col(0)), d(r2, col(1)), d(r2, col(2))});
    }
    float det() const { return r0.x * (r1.y * r2.z - r1.z * r2.y) - r0.y * (r1.x * r2.z - r1.z * r2.x) + r0.z * (r1.x * r2.y - r1.y * r2.x); }
    Matrix3x3 inverse() const {
        const float s = (float)(1.0 / (double)det());
        return Matrix3x3(float3{(r1.y * r2.z - r1.z * r2.y) * s, (r0.z * r2.y - r0.y * r2.z) * s, (r0.y * r1.z - r0.z * r1.y) * s},
                         float3{(r1.z * r2.x - r1.x * r2.z) * s, (r0.x * r2.z - r0.z * r2.x) * s, (r0.z * r1.x - r0.x * r1.z) * s},
                         float3{(r1.x * r2.y - r1.y * r2.x) * s, (r0.y * r2.x - r0.x * r2.y) * s, (r0.x * r1.y - r0.y * r1.x) * s});
    }
private:
    float3 r0, r1, r2;
};

class Float3AffineTransform {
public:
    Float3AffineTransform() : m(1.0f, 1.0f, 1.0f), v{0, 0, 0} {}
    Float3AffineTransform(const Matrix3x3& mIn, float3 vIn) : m(mIn), v(vIn) {}
    float3 transformPoint(float3 p) const { float3 r = m * p; return float3{r.x + v.x, r.y + v.y, r.z + v.z}; }
    Float3AffineTransform inverse() const {
        Matrix3x3 mi = m.inverse();
        float3 t = mi * float3{-v.x, -v.y, -v.z};
        return Float3AffineTransform(mi, t);
    }
    Matrix3x3 getMatrix() const { return m; }
    float3 getOffset() const { return v; }
    friend Float3AffineTransform concatFloat3AffineTransform(const Float3AffineTransform& t1, const Float3AffineTransform& t2) {
        float3 t = t2.m * t1.v;   // apply t1 then t2 (float3_affine_transform.cu:42-45)
        return Float3AffineTransform(t2.m * t1.m, float3{t.x + t2.v.x, t.y + t2.v.y, t.z + t2.v.z});
    }
private:
    Matrix3x3 m;
    float3 v;
};

class Float3IdxTransform {
public:
    Float3IdxTransform() : delta{1, 1, 1}, offset{0, 0, 0} {}
    Float3IdxTransform(float3 d, float3 o) : delta(d), offset(o) {}
    float3 getDelta() const { return delta; }
    float3 getOffset() const { return offset; }
private:
    float3 delta, offset;
};

template <typename T>
class HostPinnedImage3D {
public:
    // page-locks the caller's buffer (host_image_3d.cuh:23-32); without a GPU the registration fails and the image is pageable
    HostPinnedImage3D(T* imagePtr, uint3 dimensions) : imPtr(imagePtr), dims(dimensions) {
        pinned = imagePtr && rtd_host_register((void*)imPtr, (size_t)dims.x * dims.y * dims.z * sizeof(T)) == RTD_OK;
    }
    ~HostPinnedImage3D() { if (pinned) rtd_host_unregister((void*)imPtr); }   // host_image_3d.cuh:45-48
    HostPinnedImage3D(const HostPinnedImage3D&) = delete;
    HostPinnedImage3D& operator=(const HostPinnedImage3D&) = delete;
    T* getImData() const { return imPtr; }
    uint3 getDims() const { return dims; }
    bool isPinned() const { return pinned; }
private:
    T* const imPtr;     // not owned
    const uint3 dims;
    bool pinned = false;
};

class BeamSettings {   // src/beam_settings.h:31,101-109
public:
    BeamSettings(HostPinnedImage3D<float>* spotWeights, const std::vector<float>& beamEnergies, const std::vector<float2>& spotSigmas,
                 float2 raySpacing, unsigned int tracerSteps, float2 sourceDist, Float3IdxTransform spotIdxToGantry,
                 const Float3AffineTransform& gantryToImIdx, const Float3AffineTransform& gantryToDoseIdx)
        : weightMaps_(spotWeights), layerEnergies_(beamEnergies), layerSigmas_(spotSigmas), rayPitch_(raySpacing), nSteps_(tracerSteps),
          sad_(sourceDist), spotGrid_(spotIdxToGantry), toImage_(gantryToImIdx), toDose_(gantryToDoseIdx) {}
    HostPinnedImage3D<float>* getWeights() { return weightMaps_; }
    std::vector<float>& getEnergies() { return layerEnergies_; }
    std::vector<float2>& getSpotSigmas() { return layerSigmas_; }
    float2 getRaySpacing() const { return rayPitch_; }
    unsigned int getSteps() const { return nSteps_; }
    float2 getSourceDist() const { return sad_; }
    Float3IdxTransform getSpotIdxToGantry() const { return spotGrid_; }
    Float3AffineTransform getGantryToImIdx() const { return toImage_; }
    Float3AffineTransform getGantryToDoseIdx() const { return toDose_; }
private:
    HostPinnedImage3D<float>* weightMaps_;      // [layer][ny][nx] particle numbers, not owned
    std::vector<float> layerEnergies_;          // MeV/u per layer
    std::vector<float2> layerSigmas_;           // spot sigma (x, y) per layer, mm at iso in air
    float2 rayPitch_;                           // mm between rays at iso
    unsigned int nSteps_;
    float2 sad_;                                // apparent source distances, mm (may be +inf)
    Float3IdxTransform spotGrid_;
    Float3AffineTransform toImage_, toDose_;
};

struct EnergyStruct {   // src/energy_struct.h:13-31
    int nEnergySamples = 0, nEnergies = 0;
    std::vector<float> energiesPerU, peakDepths, scaleFacts, ciddMatrix;
    int nDensitySamples = 0; float densityScaleFact = 0; std::vector<float> densityVector;
    int nSpSamples = 0; float spScaleFact = 0; std::vector<float> spVector;
    int nRRlSamples = 0; float rRlScaleFact = 0; std::vector<float> rRlVector;
    std::vector<float> nucWeightMatrix, nucSqSigmaMatrix;   // NUCLEAR_CORR only (energy_struct.h:33-36); empty otherwise
};

// energyReader(dataPath) (src/energy_reader.cpp:12-162); waterCubeTest selects the *_inc_water radiation-length file, nuclearCorr
// (RTD_NUC_*, include/rtd.h) the NUCLEAR_CORR table to read and check as well (:103-162).
inline EnergyStruct energyReader(const std::string& dataPath, bool waterCubeTest = false, int nuclearCorr = 0) {
    EnergyStruct e;
    auto open = [&](const std::string& name) {
        std::ifstream f((dataPath + name).c_str());
        if (!f) throw std::runtime_error("Failed to open " + dataPath + name);
        return f;
    };
    {
        std::ifstream f = open("proton_cumul_ddd_data.txt");
        f >> e.nEnergySamples >> e.nEnergies;
        e.energiesPerU.resize(e.nEnergies); e.peakDepths.resize(e.nEnergies); e.scaleFacts.resize(e.nEnergies);
        e.ciddMatrix.resize((size_t)e.nEnergySamples * e.nEnergies);
        for (auto& v : e.energiesPerU) f >> v;
        for (auto& v : e.peakDepths) f >> v;
        for (auto& v : e.scaleFacts) f >> v;
        for (auto& v : e.ciddMatrix) f >> v;
        if (!f) throw std::runtime_error("Truncated " + dataPath + "proton_cumul_ddd_data.txt");
    }
    auto one = [&](const std::string& name, int& n, float& scale, std::vector<float>& vec) {
        std::ifstream f = open(name);
        f >> n >> scale;
        vec.resize(n);
        for (auto& v : vec) f >> v;
        if (!f) throw std::runtime_error("Truncated " + dataPath + name);
    };
    one("density_Schneider2000_adj.txt", e.nDensitySamples, e.densityScaleFact, e.densityVector);
    one("HU_to_SP_H&N_adj.txt", e.nSpSamples, e.spScaleFact, e.spVector);
    one(waterCubeTest ? "radiation_length_inc_water.txt" : "radiation_length.txt", e.nRRlSamples, e.rRlScaleFact, e.rRlVector);
    if (nuclearCorr != RTD_NUC_OFF) {
        const std::string name = nuclearCorr == RTD_NUC_SOUKUP ? "nuclear_weights_and_sigmas_Soukup.txt"
                               : nuclearCorr == RTD_NUC_FLUKA ? "nuclear_weights_and_sigmas_Fluka.txt" : "nuclear_weights_and_sigmas_fit.txt";
        std::ifstream f = open(name);
        int nS = 0, nE = 0;
        f >> nS >> nE;
        if (nS != e.nEnergySamples || nE != e.nEnergies)
            throw std::runtime_error("Number of samples or energies in " + name + " different from proton_cumul_ddd_data.txt");
        const std::vector<float>* axes[3] = { &e.energiesPerU, &e.peakDepths, &e.scaleFacts };
        const char* what[3] = { "Energies", "Peak depths", "Scale facts" };
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < nE; ++i) {
                float v = 0;
                f >> v;
                if (std::fabs((*axes[a])[i] - v) > 0.01f) throw std::runtime_error(std::string(what[a]) + " in " + name + " different from proton_cumul_ddd_data.txt");
            }
        e.nucWeightMatrix.resize((size_t)nS * nE); e.nucSqSigmaMatrix.resize((size_t)nS * nE);
        for (auto& v : e.nucWeightMatrix) f >> v;
        for (auto& v : e.nucSqSigmaMatrix) f >> v;
        if (!f) throw std::runtime_error("Truncated " + dataPath + name);
    }
    return e;
}

}  // namespace rtd_types
