// Host-only driver for include/rtd_dicom.hpp: reads a CT directory and an RT Ion Plan beam and dumps what it parsed as raw
// little-endian files for tests/test_dicom_input.py to compare with what the fixture writer put in.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "rtd_dicom.hpp"

using namespace rtd_types;

template <typename T>
static void dump(const std::string& path, const std::vector<T>& v) {
    std::ofstream o(path.c_str(), std::ios::binary);
    o.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(T));
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s ct_dir rtplan beam out_dir\n", argv[0]); return 2; }
    try {
        const std::string out = argv[4];
        const rtd_dicom::CtVolume ct = rtd_dicom::readCtSeries(argv[1]);
        dump(out + "/ct.bin", ct.huPlus1000);
        const Matrix3x3 m = ct.imIdxToWorld.getMatrix();
        const float3 o = ct.imIdxToWorld.getOffset();
        std::vector<float> geo = {float(ct.dim.x), float(ct.dim.y), float(ct.dim.z), m.row0().x, m.row0().y, m.row0().z, m.row1().x, m.row1().y, m.row1().z,
                                  m.row2().x, m.row2().y, m.row2().z, o.x, o.y, o.z};
        dump(out + "/ct_geo.bin", geo);
        const rtd_dicom::File plan = rtd_dicom::readFile(argv[2]);
        std::cout << "beams:";
        for (const auto& n : rtd_dicom::beamNames(plan)) std::cout << " " << n;
        std::cout << "\n";
        const rtd_dicom::PlanBeam b = rtd_dicom::readPlanBeam(plan, argv[3]);
        std::vector<float> sp;
        for (const auto& s : b.spots) { sp.push_back(s.energy); sp.push_back(s.x); sp.push_back(s.y); sp.push_back(s.fwhmX); sp.push_back(s.fwhmY); sp.push_back(s.meterset); }
        dump(out + "/spots.bin", sp);
        std::vector<float> bg = {b.geo.gantryAngleDeg, b.patientSupportAngleDeg, b.beamLimitingDeviceAngleDeg, b.geo.isocenter.x, b.geo.isocenter.y, b.geo.isocenter.z,
                                 b.geo.sourceDist.x, b.geo.sourceDist.y, float(b.nLayers)};
        dump(out + "/beam_geo.bin", bg);
        // geometry conventions: where the source (0, 0, +1000 in gantry coordinates) and the gantry axes land in patient coordinates
        std::vector<float> conv;
        const float angles[5][2] = {{0, 0}, {90, 0}, {180, 0}, {270, 0}, {0, 90}};
        for (const auto& a : angles) {
            const Float3AffineTransform g = rtd_dicom::gantryToPatientHfs(a[0], a[1], make_float3(0, 0, 0));
            for (const float3 p : {make_float3(0, 0, 1000), make_float3(1, 0, 0), make_float3(0, 1, 0)}) {
                const float3 q = g.transformPoint(p);
                conv.push_back(q.x); conv.push_back(q.y); conv.push_back(q.z);
            }
        }
        dump(out + "/conventions.bin", conv);
        float sd; unsigned int st;
        rtd_dicom::tracerRange(ct, rtd_dicom::gantryToPatientHfs(b.geo.gantryAngleDeg, b.patientSupportAngleDeg, b.geo.isocenter), 1.0f, sd, st);
        dump(out + "/tracer_range.bin", std::vector<float>{sd, float(st)});
        std::cout << "series " << ct.seriesUid << " " << ct.dim.x << "x" << ct.dim.y << "x" << ct.dim.z << " spots " << b.spots.size() << " layers " << b.nLayers << "\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
