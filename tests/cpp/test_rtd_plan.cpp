// Host-only checks of include/rtd_plan.hpp (spot list -> BeamSettings); built and run by tests/test_plan_import.py.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "water_cube_plan.hpp"   // waterCubeSpots + the built-in plan's formulas

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static bool close(float a, float b, float rel = 2e-6f) { return std::fabs(a - b) <= rel * std::max(std::fabs(a), std::fabs(b)) + 1e-12f; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const EnergyStruct luts = energyReader(argv[1], true);
    const unsigned int L = 3;
    // 1. the reference's water-cube field written as a plan spot list comes back as the built-in BeamSettings
    const std::vector<rtd_plan::Spot> spots = waterCubeSpots(luts, L);
    CHECK(spots.size() == 33u * 33u * L);
    const Float3AffineTransform imIdxToWorld(Matrix3x3(1.0f, 1.0f, 1.0f), make_float3(-128.0f, -128.0f, -256.0f + 150.0f));
    rtd_plan::FieldGeometry geo;
    const rtd_plan::BuiltField f = rtd_plan::buildField(spots, geo, imIdxToWorld, imIdxToWorld);
    CHECK(f.dims.x == 33 && f.dims.y == 33 && f.dims.z == L);
    CHECK(close(f.spotIdxToGantry.getDelta().x, 3.0f) && close(f.spotIdxToGantry.getDelta().y, 3.0f) && f.spotIdxToGantry.getDelta().z == -1.0f);
    CHECK(f.spotIdxToGantry.getOffset().x == -48.0f && f.spotIdxToGantry.getOffset().y == -48.0f && f.spotIdxToGantry.getOffset().z == 128.0f);
    for (size_t i = 0; i < spots.size(); ++i) CHECK(f.weights[i] == spots[i].meterset);            // same order: layer, y, x
    float e = 118.12f;
    for (unsigned int l = 0; l < L; ++l) {
        CHECK(f.energies[l] == e);
        const float peak = vectorInterpolate(luts.peakDepths, findDecimalOrdered(luts.energiesPerU, e));
        const float sigma = 2.3f + 290.0f / (peak + 15.0f);                                          // main.cu:92-95
        CHECK(close(f.sigmas[l].x, sigma) && close(f.sigmas[l].y, sigma));
        e += (172.51f - 118.12f) / float(L - 1);
    }
    const Float3AffineTransform ref = concatFloat3AffineTransform(Float3AffineTransform().inverse(), imIdxToWorld.inverse());   // main.cu:55-57
    CHECK(f.gantryToImIdx.getMatrix().row0().x == ref.getMatrix().row0().x && f.gantryToImIdx.getOffset().z == ref.getOffset().z);
    CHECK(f.gantryToImIdx.getOffset().x == ref.getOffset().x && f.gantryToImIdx.getMatrix().row2().z == ref.getMatrix().row2().z);
    // 2. ragged plan: layers with different spot subsets share one grid; missing spots are zero weights; repeats accumulate
    std::vector<rtd_plan::Spot> rag = {
        {100.0f, -5.0f, 0.0f, 10.0f, 12.0f, 2.0f}, {100.0f, 5.0f, 0.0f, 10.0f, 12.0f, 6.0f},
        {110.0f, 0.0f, 2.5f, 8.0f, 8.0f, 1.0f}, {110.0f, 0.0f, 2.5f, 8.0f, 8.0f, 0.5f}, {110.0f, 5.0f, -2.5f, 16.0f, 8.0f, 1.5f},
    };
    const rtd_plan::BuiltField r = rtd_plan::buildField(rag, geo, imIdxToWorld, imIdxToWorld);
    CHECK(r.dims.x == 3 && r.dims.y == 3 && r.dims.z == 2);                                       // x in {-5,0,5}, y in {-2.5,0,2.5}
    CHECK(r.weights[(0 * 3 + 1) * 3 + 0] == 2.0f && r.weights[(0 * 3 + 1) * 3 + 2] == 6.0f && r.weights[(0 * 3 + 1) * 3 + 1] == 0.0f);
    CHECK(r.weights[(1 * 3 + 2) * 3 + 1] == 1.5f && r.weights[(1 * 3 + 0) * 3 + 2] == 1.5f);
    const float k = 1.0f / (2.0f * std::sqrt(2.0f * std::log(2.0f)));
    CHECK(close(r.sigmas[0].x, 10.0f * k, 1e-5f) && close(r.sigmas[0].y, 12.0f * k, 1e-5f));
    CHECK(close(r.sigmas[1].x, (1.5f * 8.0f + 1.5f * 16.0f) / 3.0f * k, 1e-5f) && close(r.sigmas[1].y, 8.0f * k, 1e-5f));   // meterset-weighted
    CHECK(r.energies[0] == 100.0f && r.energies[1] == 110.0f);
    // 3. gantry at 90 degrees about world Y: the beam axis (gantry -z) points along world -x; isocentre translation applied
    geo.gantryAngleDeg = 90.0f; geo.isocenter = make_float3(10.0f, 20.0f, 30.0f);
    const rtd_plan::BuiltField g = rtd_plan::buildField(rag, geo, Float3AffineTransform(), Float3AffineTransform());
    const float3 p = g.gantryToImIdx.transformPoint(make_float3(0.0f, 0.0f, 100.0f));               // 100 mm upstream of the isocentre
    CHECK(close(p.x, 110.0f) && close(p.y, 20.0f) && std::fabs(p.z - 30.0f) < 1e-4f);
    // 4. errors
    std::vector<rtd_plan::Spot> off = rag; off.push_back({110.0f, 1.3f, 0.0f, 8.0f, 8.0f, 1.0f});
    bool threw = false;
    try { rtd_plan::buildField(off, geo, imIdxToWorld, imIdxToWorld); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { rtd_plan::buildField({}, geo, imIdxToWorld, imIdxToWorld); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    // 5. spot-list text round trip
    const std::string path = std::string(argv[2]) + "/spots.txt";
    rtd_plan::FieldGeometry g2; g2.gantryAngleDeg = 37.5f; g2.isocenter = make_float3(1.0f, -2.0f, 3.5f); g2.sourceDist = make_float2(2000.0f, 2500.0f);
    writeSpotList(path, spots, g2);
    std::vector<rtd_plan::Spot> back; rtd_plan::FieldGeometry g3;
    readSpotList(path, back, g3);
    CHECK(back.size() == spots.size() && g3.gantryAngleDeg == 37.5f && g3.isocenter.z == 3.5f && g3.sourceDist.y == 2500.0f);
    for (size_t i = 0; i < spots.size(); ++i) CHECK(back[i].energy == spots[i].energy && back[i].meterset == spots[i].meterset && back[i].fwhmX == spots[i].fwhmX);
    std::puts("rtd_plan ok");
    return 0;
}
