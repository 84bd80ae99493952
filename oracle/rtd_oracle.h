/*
 * rtd_oracle.h — prototypes of the CPU oracle (TEST INFRASTRUCTURE ONLY, see rtd_oracle.c).
 */
#ifndef RTD_ORACLE_H
#define RTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/rtd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_field_s* orc_field;

void orc_set_threads(int n);
void orc_set_weight_bits(int bits);
int orc_get_max_threads(void);
/* distinct CT voxels read by the tracer of the fields run between start and stop (N_fp of SURVEY.md 8(d)) */
void orc_footprint_start(size_t nVoxels);
long long orc_footprint_stop(void);
float orc_pow_det(float x, float y);
float orc_erf_det(float x);

/* host helpers (vector_find.h, vector_interpolate.h) */
float orc_find_max(const float* list, int n);
int orc_find_first_larger_ordered(const float* list, int n, float value);
int orc_find_last_smaller_or_eq_ordered(const float* list, int n, float value);
float orc_find_decimal_ordered(const float* list, int n, float value);
float orc_vector_interpolate(const float* list, int n, float idx);

/* transforms and parameter probes */
void orc_affine_inverse(const rtd_affine* in, rtd_affine* out);
void orc_affine_concat(const rtd_affine* t1, const rtd_affine* t2, rtd_affine* out);
void orc_from_fan_point(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                        const float in[3], float out[3]);
void orc_to_fan_point(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                      const float shift[3], const float in[3], float out[3]);
void orc_transfer_fan_idx(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                          const float shift[3], int i, int j, int k, float out[3]);
void orc_tracer_probe(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                      int i, int j, float start[3], float inc[3], float* stepLen);
void orc_fill_probe(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                    float peakDepth, int nozzle, unsigned int k, float* stepVol, float voxelWidth[2],
                    float* sigmaSqAirLin, float* sigmaSqAirQuad);
float orc_sample1d(const float* t, int n, float p);
float orc_sample2d(const float* t, int ncol, int nrow, float px, float py);
float orc_sample3d(const float* vol, int nx, int ny, int nz, float px, float py, float pz);
void orc_erf_diffs(float rSigmaEff, int rad, float* out);
int orc_batch_radii(const int ctrs[34], int effRad[34]);
int orc_ray_grid(const rtd_beam* b, const rtd_options* opt, unsigned int rayDims[3], float rayRes[3], float rayOffset[3]);

/* one field / all beams */
int orc_field_run(const rtd_luts* l, const float* ct, const uint32_t ctDims[3], const rtd_beam* b,
                  float* dose, const uint32_t doseDims[3], const rtd_options* opt, int keepLayers, orc_field* out);
int orc_compute(const rtd_luts* l, const float* ct, const uint32_t ctDims[3], const rtd_beam* beams, int nBeams,
                float* dose, const uint32_t doseDims[3], const rtd_options* opt);
const char* orc_field_error(orc_field f);
void orc_field_info(orc_field f, rtd_field_info* out);
const void* orc_field_get(orc_field f, const char* name, size_t* bytes);
void orc_field_free(orc_field f);

/* uniform-sigma separable convolution (cpu_convolution_1d.cpp) */
void orc_x_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth,
                    unsigned int outWidth, unsigned int height, int inOutOffset);
void orc_x_conv_cpu_scat(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth,
                         unsigned int outWidth, unsigned int height, unsigned int inOutOffset);
void orc_y_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inHeight,
                    unsigned int width, int inOutOffset);

/* LUT text layout (energy_reader.cpp) */
int orc_read_luts(const char* dir, int waterCubeTest, rtd_luts* out);
int orc_read_luts_nuc(const char* dir, int waterCubeTest, int variant, rtd_luts* out);
void orc_luts_free(rtd_luts* l);

/* gamma index */
double orc_gamma_pass_rate(const float* ref, const float* eval, const uint32_t dims[3], const float spacing[3],
                           float ddFrac, float dtaMm, float thresholdFrac, int64_t* nEval, float* maxGamma);

#ifdef __cplusplus
}
#endif
#endif
