// ref_wrap.cpp — extern "C" doorways into the REFERENCE's own host code, compiled from
// /root/reference/src where it lies (see oracle/Makefile, target `ref`). Used only to pin the oracle
// (tests/golden generation). This file contains no reference code: it includes the reference headers
// by name and forwards calls.
#include <cstring>
#include <string>
#include <vector>
#include "energy_reader.h"
#include "energy_struct.h"
#include "cpu_convolution_1d.h"
#include "vector_find.h"
#include "vector_interpolate.h"

static EnergyStruct g_es;

extern "C" {

int ref_energy_reader(const char* dir) {
    try { g_es = energyReader(std::string(dir)); } catch (...) { return -1; }
    return 0;
}
// which: 0 energiesPerU 1 peakDepths 2 scaleFacts 3 ciddMatrix 4 density 5 sp 6 rRl
long ref_energy_size(int which) {
    switch (which) {
        case 0: return (long)g_es.energiesPerU.size(); case 1: return (long)g_es.peakDepths.size();
        case 2: return (long)g_es.scaleFacts.size(); case 3: return (long)g_es.ciddMatrix.size();
        case 4: return (long)g_es.densityVector.size(); case 5: return (long)g_es.spVector.size();
        case 6: return (long)g_es.rRlVector.size();
#ifdef NUCLEAR_CORR
        case 7: return (long)g_es.nucWeightMatrix.size(); case 8: return (long)g_es.nucSqSigmaMatrix.size();
#endif
    }
    return -1;
}
void ref_energy_copy(int which, float* dst) {
    const std::vector<float>* v = nullptr;
    switch (which) {
        case 0: v = &g_es.energiesPerU; break; case 1: v = &g_es.peakDepths; break; case 2: v = &g_es.scaleFacts; break;
        case 3: v = &g_es.ciddMatrix; break; case 4: v = &g_es.densityVector; break; case 5: v = &g_es.spVector; break;
        case 6: v = &g_es.rRlVector; break;
#ifdef NUCLEAR_CORR
        case 7: v = &g_es.nucWeightMatrix; break; case 8: v = &g_es.nucSqSigmaMatrix; break;
#endif
    }
    if (v) std::memcpy(dst, v->data(), v->size() * sizeof(float));
}
void ref_energy_scalars(int* nEnergySamples, int* nEnergies, int* nDensity, float* densityScale, int* nSp, float* spScale,
                        int* nRRl, float* rRlScale) {
    *nEnergySamples = g_es.nEnergySamples; *nEnergies = g_es.nEnergies; *nDensity = g_es.nDensitySamples;
    *densityScale = g_es.densityScaleFact; *nSp = g_es.nSpSamples; *spScale = g_es.spScaleFact;
    *nRRl = g_es.nRRlSamples; *rRlScale = g_es.rRlScaleFact;
}

float ref_find_max(const float* l, int n) { return findMax<float>(std::vector<float>(l, l + n)); }
int ref_find_first_larger_ordered(const float* l, int n, float v) { return findFirstLargerOrdered<float>(std::vector<float>(l, l + n), v); }
int ref_find_last_smaller_or_eq_ordered(const float* l, int n, float v) { return findLastSmallerOrEqOrdered<float>(std::vector<float>(l, l + n), v); }
float ref_find_decimal_ordered(const float* l, int n, float v) { return findDecimalOrdered<float, float>(std::vector<float>(l, l + n), v); }
float ref_vector_interpolate(const float* l, int n, float idx) { return vectorInterpolate<float, float>(std::vector<float>(l, l + n), idx); }

void ref_x_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth, unsigned int outWidth,
                    unsigned int height, int inOutOffset) { xConvCpu(in, out, rSigmaEff, rad, inWidth, outWidth, height, inOutOffset); }
void ref_x_conv_cpu_scat(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth, unsigned int outWidth,
                         unsigned int height, unsigned int inOutOffset) { xConvCpuScat(in, out, rSigmaEff, rad, inWidth, outWidth, height, inOutOffset); }
void ref_y_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inHeight, unsigned int width,
                    int inOutOffset) { yConvCpu(in, out, rSigmaEff, rad, inHeight, width, inOutOffset); }

}
