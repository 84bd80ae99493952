// rtd_dicom.hpp — minimal DICOM input for the dose engine: a CT series and an RT Ion Plan, without
// ITK / GDCM / dicom-interface (none of which exist on the target image).
//
// What the reference gets from those libraries and what is rebuilt here:
//   * itk_reader (src/dicom_reader.cpp:15-129): first series of a directory, slices stacked along the slice normal, pixels
//     rescaled to HU as short, +1000, and the affine  imIdxToWorld = Direction * diag(Spacing), Origin  (:117-128);
//   * the plan part of main.cu (:105-181, through topasmc/dicom-interface): per beam the control-point sequence with
//     nominal energy, scan-spot positions, meterset weights and spot size, gantry / couch angles and isocentre — printed there,
//     turned into BeamSettings here by rtd_plan.hpp.
// Scope of the parser: DICOM Part 10 files, transfer syntaxes Implicit VR Little Endian (1.2.840.10008.1.2) and Explicit VR
// Little Endian (1.2.840.10008.1.2.1), uncompressed 16-bit pixel data, sequences of defined or undefined length. Anything
// else (big endian, deflate, encapsulated / compressed pixel data) is rejected with a message.
// Parity: there is no ITK here to compare with, so this reader is pinned by its own writer-side fixtures only
// (tests/dicom_fixture.py); DESIGN.md says so.
#pragma once
#include <dirent.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>

#include "rtd_plan.hpp"

namespace rtd_dicom {

using namespace rtd_types;

struct Element;
using Dataset = std::map<uint32_t, Element>;                         // key = (group << 16) | element
struct Element {
    char vr[3] = {'U', 'N', 0};
    std::string value;                                                // raw little-endian bytes
    std::vector<Dataset> items;                                       // for sequences
};
inline uint32_t tag(uint16_t g, uint16_t e) { return ((uint32_t)g << 16) | e; }

namespace detail {

struct Reader {
    const std::string& buf;
    size_t pos;
    bool explicitVr;
    const std::string& name;
    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error(name + ": " + what); }
    void need(size_t n) const { if (pos + n > buf.size()) fail("truncated DICOM stream"); }
    uint16_t u16() { need(2); uint16_t v; std::memcpy(&v, &buf[pos], 2); pos += 2; return v; }
    uint32_t u32() { need(4); uint32_t v; std::memcpy(&v, &buf[pos], 4); pos += 4; return v; }
};

inline bool longVr(const char* vr) {
    static const char* l[] = {"OB", "OD", "OF", "OL", "OV", "OW", "SQ", "UC", "UN", "UR", "UT", "SV", "UV"};
    for (const char* v : l) if (vr[0] == v[0] && vr[1] == v[1]) return true;
    return false;
}
// Implicit VR: the value representation comes from the data dictionary; these are the binary-valued attributes this reader
// decodes (everything else it reads is a decimal / integer / code string, which needs no dictionary)
inline const char* dictionaryVr(uint32_t t) {
    switch (t) {
        case 0x00280002u: case 0x00280010u: case 0x00280011u: case 0x00280100u: case 0x00280101u: case 0x00280102u: case 0x00280103u:
            return "US";                                              // samples, rows, columns, bits allocated/stored/high, representation
        case 0x300A0394u: case 0x300A0396u: case 0x300A0398u: case 0x300A030Au:
            return "FL";                                              // scan spot position map, meterset weights, spot size, VSAD
        case 0x7FE00010u: return "OW";
        default: return nullptr;
    }
}
// the few sequences this reader descends into when the VR is implicit
inline bool knownSequence(uint32_t t) {
    switch (t) {
        case 0x300A03A2u: case 0x300A03A8u: case 0x300A00B0u: case 0x300A0070u: case 0x300C0004u: case 0x300A0010u:
        case 0x300A03A4u: case 0x300A03ACu: case 0x300A0314u: case 0x300A0360u: case 0x300A0342u: case 0x300A00B6u:
            return true;
        default: return false;
    }
}

void parseDataset(Reader& r, size_t end, Dataset& out, bool stopAtItemDelim, int depth);

inline void parseSequence(Reader& r, uint32_t length, Element& el, int depth) {
    if (depth > 16) r.fail("sequences nested too deeply");
    const bool undefinedLen = length == 0xFFFFFFFFu;
    const size_t end = undefinedLen ? r.buf.size() : r.pos + length;
    if (end > r.buf.size()) r.fail("sequence longer than the file");
    while (r.pos < end) {
        const uint16_t g = r.u16(), e = r.u16();
        const uint32_t len = r.u32();
        if (g == 0xFFFE && e == 0xE0DD) { if (!undefinedLen) r.fail("unexpected sequence delimiter"); return; }
        if (!(g == 0xFFFE && e == 0xE000)) r.fail("expected an item in a sequence");
        el.items.emplace_back();
        if (len == 0xFFFFFFFFu) parseDataset(r, r.buf.size(), el.items.back(), true, depth + 1);
        else { if (r.pos + len > r.buf.size()) r.fail("item longer than the file"); parseDataset(r, r.pos + len, el.items.back(), false, depth + 1); }
    }
    if (undefinedLen) r.fail("sequence without delimiter");
}

inline void parseDataset(Reader& r, size_t end, Dataset& out, bool stopAtItemDelim, int depth) {
    while (r.pos < end) {
        const uint16_t g = r.u16(), e = r.u16();
        if (g == 0xFFFE && e == 0xE00D) { r.u32(); if (!stopAtItemDelim) r.fail("unexpected item delimiter"); return; }
        Element el;
        uint32_t len;
        if (r.explicitVr) {
            r.need(2);
            el.vr[0] = r.buf[r.pos]; el.vr[1] = r.buf[r.pos + 1]; r.pos += 2;
            if (longVr(el.vr)) { r.u16(); len = r.u32(); } else len = r.u16();
        } else {
            len = r.u32();
            if (knownSequence(tag(g, e)) || len == 0xFFFFFFFFu) { el.vr[0] = 'S'; el.vr[1] = 'Q'; }
            else if (const char* dv = dictionaryVr(tag(g, e))) { el.vr[0] = dv[0]; el.vr[1] = dv[1]; }
        }
        if (el.vr[0] == 'S' && el.vr[1] == 'Q') parseSequence(r, len, el, depth);
        else {
            if (len == 0xFFFFFFFFu) r.fail("encapsulated (compressed) pixel data is not supported");
            r.need(len);
            el.value.assign(r.buf, r.pos, len);
            r.pos += len;
        }
        out[tag(g, e)] = std::move(el);
    }
    if (stopAtItemDelim) r.fail("item without delimiter");
}

inline std::string trimmed(const std::string& s) {
    size_t a = 0, b = s.size();
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\0')) --b;
    while (a < b && s[a] == ' ') ++a;
    return s.substr(a, b - a);
}

}  // namespace detail

struct File {
    std::string path, transferSyntax;
    Dataset meta, data;
    const Element* find(uint32_t t) const { auto it = data.find(t); return it == data.end() ? nullptr : &it->second; }
};

inline std::string str(const Element* e) { return e ? detail::trimmed(e->value) : std::string(); }
// decimal / integer strings ("a\b\c") or binary FL / FD / US / SS / UL / SL values
inline std::vector<double> numbers(const Element* e) {
    std::vector<double> v;
    if (!e) return v;
    const std::string vr(e->vr, 2);
    auto bin = [&](auto sample) { using T = decltype(sample); for (size_t i = 0; i + sizeof(T) <= e->value.size(); i += sizeof(T)) { T x; std::memcpy(&x, &e->value[i], sizeof(T)); v.push_back((double)x); } };
    if (vr == "FL") bin(float()); else if (vr == "FD") bin(double()); else if (vr == "US") bin(uint16_t()); else if (vr == "SS") bin(int16_t());
    else if (vr == "UL") bin(uint32_t()); else if (vr == "SL") bin(int32_t());
    else {
        std::string s = detail::trimmed(e->value), cur;
        for (size_t i = 0; i <= s.size(); ++i) {
            if (i == s.size() || s[i] == '\\') { if (!detail::trimmed(cur).empty()) v.push_back(std::strtod(cur.c_str(), nullptr)); cur.clear(); }
            else cur += s[i];
        }
    }
    return v;
}

inline File readFile(const std::string& path, bool headerOnly = false) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) throw std::runtime_error("Failed to open " + path);
    std::string buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (buf.size() < 132 || buf.compare(128, 4, "DICM") != 0) throw std::runtime_error(path + ": not a DICOM Part 10 file");
    File f;
    f.path = path;
    // file meta information: explicit VR little endian, group 0002 only
    detail::Reader mr{buf, 132, true, path};
    size_t metaEnd = buf.size();
    {
        detail::Reader probe{buf, 132, true, path};
        const uint16_t g = probe.u16(), e = probe.u16();
        if (g == 0x0002 && e == 0x0000) { probe.pos += 2; const uint16_t l = probe.u16(); if (l == 4) { const uint32_t groupLen = probe.u32(); metaEnd = probe.pos + groupLen; } }
    }
    if (metaEnd == buf.size()) {                                       // no group length: walk group 0002 elements
        detail::Reader w{buf, 132, true, path};
        while (w.pos + 4 <= buf.size()) {
            const size_t at = w.pos;
            if (w.u16() != 0x0002) { metaEnd = at; break; }
            w.u16();
            char vr[3] = {buf[w.pos], buf[w.pos + 1], 0}; w.pos += 2;
            uint32_t len; if (detail::longVr(vr)) { w.u16(); len = w.u32(); } else len = w.u16();
            w.pos += len;
        }
    }
    detail::parseDataset(mr, metaEnd, f.meta, false, 0);
    auto ts = f.meta.find(tag(0x0002, 0x0010));
    f.transferSyntax = ts == f.meta.end() ? "1.2.840.10008.1.2" : detail::trimmed(ts->second.value);
    bool explicitVr;
    if (f.transferSyntax == "1.2.840.10008.1.2") explicitVr = false;
    else if (f.transferSyntax == "1.2.840.10008.1.2.1") explicitVr = true;
    else throw std::runtime_error(path + ": transfer syntax " + f.transferSyntax + " is not supported (uncompressed little endian only)");
    detail::Reader dr{buf, metaEnd, explicitVr, path};
    (void)headerOnly;
    detail::parseDataset(dr, buf.size(), f.data, false, 0);
    return f;
}

// ---- CT series -> HU+1000 floats and imIdxToWorld (dicom_reader.cpp:15-129) ----
struct CtVolume {
    uint3 dim{0, 0, 0};
    std::vector<float> huPlus1000;                                    // x fastest, then y, then slice
    Float3AffineTransform imIdxToWorld;
    std::string seriesUid, patientPosition;                          // (0018,5100), e.g. HFS
    float3 spacing{0, 0, 0};
};

inline CtVolume readCtSeries(const std::string& dir) {
    DIR* d = ::opendir(dir.c_str());
    if (!d) throw std::runtime_error("Failed to open directory " + dir);
    std::vector<std::string> names;
    while (dirent* e = ::readdir(d)) { if (e->d_name[0] != '.') names.push_back(e->d_name); }
    ::closedir(d);
    std::sort(names.begin(), names.end());
    struct Slice { File f; double pos[3]; double along; };
    std::map<std::string, std::vector<Slice>> series;                  // SeriesInstanceUID -> slices
    std::vector<std::string> order;
    for (const auto& n : names) {
        File f;
        try { f = readFile(dir + "/" + n); } catch (const std::runtime_error&) { continue; }   // not DICOM: skipped like GDCMSeriesFileNames does
        if (!f.find(tag(0x7FE0, 0x0010)) || !f.find(tag(0x0020, 0x0032))) continue;               // no pixels / no position: not an image slice
        const std::string uid = str(f.find(tag(0x0020, 0x000E)));
        if (!series.count(uid)) order.push_back(uid);
        const std::vector<double> p = numbers(f.find(tag(0x0020, 0x0032)));
        if (p.size() != 3) throw std::runtime_error(f.path + ": ImagePositionPatient needs 3 values");
        series[uid].push_back(Slice{std::move(f), {p[0], p[1], p[2]}, 0.0});
    }
    if (order.empty()) throw std::runtime_error("The directory " + dir + " contains no DICOM Series");
    std::sort(order.begin(), order.end());                             // the reference takes *seriesUID.begin()
    CtVolume ct;
    ct.seriesUid = order.front();
    std::vector<Slice>& sl = series[ct.seriesUid];
    const File& f0 = sl.front().f;
    const std::vector<double> iop = numbers(f0.find(tag(0x0020, 0x0037))), ps = numbers(f0.find(tag(0x0028, 0x0030)));
    const std::vector<double> rows = numbers(f0.find(tag(0x0028, 0x0010))), cols = numbers(f0.find(tag(0x0028, 0x0011)));
    if (iop.size() != 6 || ps.size() != 2 || rows.size() != 1 || cols.size() != 1) throw std::runtime_error(f0.path + ": incomplete image geometry");
    const double rx[3] = {iop[0], iop[1], iop[2]}, cx[3] = {iop[3], iop[4], iop[5]};   // direction of increasing column / row index
    const double nrm[3] = {rx[1] * cx[2] - rx[2] * cx[1], rx[2] * cx[0] - rx[0] * cx[2], rx[0] * cx[1] - rx[1] * cx[0]};
    for (auto& s : sl) s.along = s.pos[0] * nrm[0] + s.pos[1] * nrm[1] + s.pos[2] * nrm[2];
    std::stable_sort(sl.begin(), sl.end(), [](const Slice& a, const Slice& b) { return a.along < b.along; });
    const unsigned int nx = (unsigned int)cols[0], ny = (unsigned int)rows[0], nz = (unsigned int)sl.size();
    double dz = 1.0;
    if (nz > 1) {
        dz = (sl.back().along - sl.front().along) / double(nz - 1);
        for (unsigned int k = 1; k < nz; ++k)
            if (std::fabs((sl[k].along - sl[k - 1].along) - dz) > 1e-3 * std::fabs(dz) + 1e-4) throw std::runtime_error(dir + ": slices are not equally spaced");
    } else {
        const std::vector<double> th = numbers(f0.find(tag(0x0018, 0x0050)));
        if (!th.empty()) dz = th[0];
    }
    if (nx == 0 || ny == 0 || f0.find(tag(0x7FE0, 0x0010))->value.size() < (size_t)nx * ny * 2)                 // before any allocation sized by the header
        throw std::runtime_error(f0.path + ": pixel data shorter than rows x columns");
    ct.dim = make_uint3(nx, ny, nz);
    ct.patientPosition = str(f0.find(tag(0x0018, 0x5100)));
    ct.huPlus1000.resize((size_t)nx * ny * nz);
    for (unsigned int k = 0; k < nz; ++k) {
        const File& f = sl[k].f;
        const std::vector<double> r = numbers(f.find(tag(0x0028, 0x0010))), c = numbers(f.find(tag(0x0028, 0x0011)));
        if (r.size() != 1 || c.size() != 1 || (unsigned int)r[0] != ny || (unsigned int)c[0] != nx) throw std::runtime_error(f.path + ": slice size differs within the series");
        const std::vector<double> ba = numbers(f.find(tag(0x0028, 0x0100))), pr = numbers(f.find(tag(0x0028, 0x0103)));
        if (ba.size() != 1 || (int)ba[0] != 16) throw std::runtime_error(f.path + ": only 16-bit pixel data is supported");
        const bool isSigned = !pr.empty() && (int)pr[0] == 1;
        const std::vector<double> sv = numbers(f.find(tag(0x0028, 0x1053))), iv = numbers(f.find(tag(0x0028, 0x1052)));
        const double slope = sv.empty() ? 1.0 : sv[0], intercept = iv.empty() ? 0.0 : iv[0];
        const std::string& px = f.find(tag(0x7FE0, 0x0010))->value;
        if (px.size() < (size_t)nx * ny * 2) throw std::runtime_error(f.path + ": pixel data shorter than rows x columns");
        float* out = &ct.huPlus1000[(size_t)k * nx * ny];
        for (size_t i = 0; i < (size_t)nx * ny; ++i) {
            uint16_t raw; std::memcpy(&raw, &px[2 * i], 2);
            const double stored = isSigned ? (double)(int16_t)raw : (double)raw;
            const short hu = (short)(stored * slope + intercept);     // ITK hands the series reader rescaled shorts (dicom_reader.cpp:17,106)
            out[i] = float(hu + 1000);                                // HUOFFSET, dicom_reader.cpp:24,106
        }
    }
    // imIdxToWorld = Direction * diag(Spacing), Origin  (dicom_reader.cpp:117-128); Direction columns = row dir, column dir, normal
    ct.spacing = make_float3((float)ps[1], (float)ps[0], (float)dz);   // PixelSpacing = (row spacing, column spacing)
    const Matrix3x3 dirM(make_float3((float)rx[0], (float)cx[0], (float)nrm[0]), make_float3((float)rx[1], (float)cx[1], (float)nrm[1]),
                         make_float3((float)rx[2], (float)cx[2], (float)nrm[2]));
    ct.imIdxToWorld = Float3AffineTransform(dirM * Matrix3x3(ct.spacing.x, ct.spacing.y, ct.spacing.z),
                                            make_float3((float)sl.front().pos[0], (float)sl.front().pos[1], (float)sl.front().pos[2]));
    return ct;
}

// ---- RT Ion Plan -> spots + geometry of one beam (main.cu:105-181) ----
struct PlanBeam {
    std::string name;
    std::vector<rtd_plan::Spot> spots;                                // delivery order
    rtd_plan::FieldGeometry geo;                                       // gantry angle, isocentre, source distances filled in
    float patientSupportAngleDeg = 0.0f, beamLimitingDeviceAngleDeg = 0.0f;
    unsigned int nLayers = 0;
};

inline std::vector<std::string> beamNames(const File& plan) {
    std::vector<std::string> out;
    if (const Element* seq = plan.find(tag(0x300A, 0x03A2)))
        for (const Dataset& b : seq->items) { auto it = b.find(tag(0x300A, 0x00C2)); out.push_back(it == b.end() ? std::string() : detail::trimmed(it->second.value)); }
    return out;
}

inline PlanBeam readPlanBeam(const File& plan, const std::string& beamName) {
    const std::string modality = str(plan.find(tag(0x0008, 0x0060)));
    if (modality != "RTPLAN") throw std::runtime_error(plan.path + ": Unknown modality " + modality);   // main.cu:143-145
    const Element* seq = plan.find(tag(0x300A, 0x03A2));                // IonBeamSequence
    if (!seq) throw std::runtime_error(plan.path + ": no IonBeamSequence (not an RT Ion Plan)");
    auto get = [](const Dataset& d, uint16_t g, uint16_t e) -> const Element* { auto it = d.find(tag(g, e)); return it == d.end() ? nullptr : &it->second; };
    for (const Dataset& b : seq->items) {
        if (str(get(b, 0x300A, 0x00C2)) != beamName) continue;
        PlanBeam out;
        out.name = beamName;
        const std::vector<double> vsad = numbers(get(b, 0x300A, 0x030A));   // VirtualSourceAxisDistances
        if (vsad.size() == 2) out.geo.sourceDist = make_float2((float)vsad[0], (float)vsad[1]);
        const Element* cps = get(b, 0x300A, 0x03A8);                    // IonControlPointSequence
        if (!cps || cps->items.empty()) throw std::runtime_error(plan.path + ": beam " + beamName + " has no control points");
        float energy = 0.0f, fwhm[2] = {0.0f, 0.0f};
        bool first = true;
        float lastLayerEnergy = -1.0f;
        for (const Dataset& cp : cps->items) {
            // values persist from control point to control point when absent (DICOM "changes only" encoding)
            const std::vector<double> en = numbers(get(cp, 0x300A, 0x0114));
            if (!en.empty()) energy = (float)en[0];
            const std::vector<double> ss = numbers(get(cp, 0x300A, 0x0398));
            if (ss.size() == 2) { fwhm[0] = (float)ss[0]; fwhm[1] = (float)ss[1]; }
            if (first) {                                               // angles and isocentre of the first control point (main.cu:133-150)
                const std::vector<double> ga = numbers(get(cp, 0x300A, 0x011E)), pa = numbers(get(cp, 0x300A, 0x0122)), ca = numbers(get(cp, 0x300A, 0x0120));
                const std::vector<double> iso = numbers(get(cp, 0x300A, 0x012C));
                if (!ga.empty()) out.geo.gantryAngleDeg = (float)ga[0];
                if (!pa.empty()) out.patientSupportAngleDeg = (float)pa[0];
                if (!ca.empty()) out.beamLimitingDeviceAngleDeg = (float)ca[0];
                if (iso.size() == 3) out.geo.isocenter = make_float3((float)iso[0], (float)iso[1], (float)iso[2]);
                first = false;
            }
            const std::vector<double> pos = numbers(get(cp, 0x300A, 0x0394)), w = numbers(get(cp, 0x300A, 0x0396));
            if (pos.empty() || w.empty()) continue;
            if (pos.size() != 2 * w.size()) throw std::runtime_error(plan.path + ": scan spot positions and weights disagree in beam " + beamName);
            double sum = 0.0;
            for (double x : w) sum += x;
            if (sum <= 0.0) continue;                                  // the closing control point of a layer carries zero weights
            if (!(energy > 0.0f) || !(fwhm[0] > 0.0f) || !(fwhm[1] > 0.0f)) throw std::runtime_error(plan.path + ": control point without energy or spot size in beam " + beamName);
            if (energy != lastLayerEnergy) { ++out.nLayers; lastLayerEnergy = energy; }
            for (size_t i = 0; i < w.size(); ++i)
                out.spots.push_back(rtd_plan::Spot{energy, (float)pos[2 * i], (float)pos[2 * i + 1], fwhm[0], fwhm[1], (float)w[i]});
        }
        if (out.spots.empty()) throw std::runtime_error(plan.path + ": beam " + beamName + " has no spots");
        return out;
    }
    throw std::runtime_error(plan.path + ": no beam named " + beamName);
}

// ---- geometry of a plan beam in the DICOM patient coordinate system ----
// IEC 61217 chain for a head-first supine patient: gantry system (Z_g from the isocentre towards the source, spot positions in
// X_g / Y_g) --R_y(gantry angle)--> fixed system (X_f to the right seen from the front, Y_f towards the gantry, Z_f up)
// --R_z(-patient support angle)--> table --> DICOM patient LPS (x = X, y = -Z, z = Y). At gantry 0 the beam travels from
// anterior to posterior (+y), at 90 degrees it enters from the patient's left (travels -x).
inline Float3AffineTransform gantryToPatientHfs(float gantryDeg, float supportDeg, float3 isocenter) {
    auto snap = [](double v) { return std::fabs(v - std::round(v)) < 1e-12 ? std::round(v) : v; };
    const double g = (double)gantryDeg * 3.14159265358979323846 / 180.0, t = -(double)supportDeg * 3.14159265358979323846 / 180.0;
    const double cg = snap(std::cos(g)), sg = snap(std::sin(g)), ct = snap(std::cos(t)), st = snap(std::sin(t));
    const Matrix3x3 ry(make_float3((float)cg, 0.0f, (float)sg), make_float3(0.0f, 1.0f, 0.0f), make_float3((float)-sg, 0.0f, (float)cg));
    const Matrix3x3 rz(make_float3((float)ct, (float)-st, 0.0f), make_float3((float)st, (float)ct, 0.0f), make_float3(0.0f, 0.0f, 1.0f));
    const Matrix3x3 fixedToPatient(make_float3(1.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, -1.0f), make_float3(0.0f, 1.0f, 0.0f));
    return Float3AffineTransform(fixedToPatient * (rz * ry), isocenter);
}

// Tracer range that covers the CT along the beam axis: step 0 just upstream of the volume, enough steps to leave it.
inline void tracerRange(const CtVolume& ct, const Float3AffineTransform& gantryToWorld, float stepLength, float& startDepth, unsigned int& steps) {
    const Float3AffineTransform worldToGantry = gantryToWorld.inverse();
    float zMin = INFINITY, zMax = -INFINITY;
    for (int c = 0; c < 8; ++c) {
        const float3 idx = make_float3((c & 1) ? float(ct.dim.x) - 0.5f : -0.5f, (c & 2) ? float(ct.dim.y) - 0.5f : -0.5f, (c & 4) ? float(ct.dim.z) - 0.5f : -0.5f);
        const float3 g = worldToGantry.transformPoint(ct.imIdxToWorld.transformPoint(idx));
        zMin = std::min(zMin, g.z); zMax = std::max(zMax, g.z);
    }
    startDepth = std::ceil(zMax) + stepLength;
    steps = (unsigned int)std::min(4096.0f, std::ceil((startDepth - zMin) / stepLength) + 1.0f);
}

}  // namespace rtd_dicom
