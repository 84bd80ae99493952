/*
 * rtd.h — C ABI of the MI355X pencil-beam proton dose engine.
 *
 * This is the drop-in boundary for the reference's single hot-path entry point
 *
 *     void cudaWrapperProtons(HostPinnedImage3D<float>* imVol, HostPinnedImage3D<float>* doseVol,
 *                             const std::vector<BeamSettings> beams, const EnergyStruct iddData,
 *                             std::ostream& outStream);            (reference src/kernel_wrapper.cuh:161)
 *
 * Every struct below is the plain-C image of one reference host type; every function cites the
 * part of the reference it replaces. No C++/STL/torch types cross this boundary: plain pointers,
 * sizes and PODs only. All arrays are x-fastest ("[z][y][x]"), float32, exactly as in the reference.
 *
 * Threading: one handle per host thread; a handle owns one HIP device, one stream and all device
 * memory it allocates. Functions return RTD_OK (0) or a negative rtd_status; rtd_last_error()
 * returns the message (the reference throws std::runtime_error / const char* instead,
 * src/cuda_errchk.cu:11-22, src/kernel_wrapper.cu:965).
 */
#ifndef RTD_H
#define RTD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTD_ABI_VERSION 3

typedef enum rtd_status {
    RTD_OK = 0,
    RTD_ERR_INVALID_ARG = -1,     /* null pointer, zero dimension, empty energy list (vector_find.h:24) */
    RTD_ERR_HIP = -2,             /* a HIP runtime call or kernel launch failed (cuda_errchk.cu:11-22) */
    RTD_ERR_RADIUS_OVERFLOW = -3, /* "Found larger than allowed kernel superposition radius" (kernel_wrapper.cu:965) */
    RTD_ERR_NOT_READY = -4,       /* CT or LUTs not set before compute */
    RTD_ERR_IO = -5,              /* LUT directory unreadable (energy_reader.cpp:21-24) */
    RTD_ERR_NO_DEVICE = -6        /* no HIP device: the engine has no CPU fallback */
} rtd_status;

/* Float3AffineTransform (src/float3_affine_transform.cuh): y = M x + v, M row-major. */
typedef struct rtd_affine {
    float m[9];
    float v[3];
} rtd_affine;

/* Float3IdxTransform (src/float3_idx_transform.cuh): y = x*delta + offset, component-wise. */
typedef struct rtd_idx_transform {
    float delta[3];
    float offset[3];
} rtd_idx_transform;

/*
 * BeamSettings (src/beam_settings.h:101-109), all nine fields.
 * spot_weights: [n_layers][spot_ny][spot_nx] particle numbers (beam_settings.h:21). Caller-owned; the
 * engine never frees it (the reference deletes it, kernel_wrapper.cu:856 — the C++ shim can mimic that).
 */
typedef struct rtd_beam {
    const float* spot_weights;
    uint32_t spot_nx, spot_ny, n_layers;
    const float* energies;            /* [n_layers] MeV/u */
    const float* spot_sigmas;         /* [n_layers][2] (sigma_x, sigma_y) mm at iso in air */
    float ray_spacing[2];             /* mm between adjacent rays at iso */
    uint32_t tracer_steps;            /* number of ray-trace steps */
    float source_dist[2];             /* apparent source-to-iso distance in x and y, mm; may be +inf */
    rtd_idx_transform spot_idx_to_gantry; /* delta.z = (negative) step length, offset.z = start depth */
    rtd_affine gantry_to_im_idx;      /* gantry mm -> CT voxel index */
    rtd_affine gantry_to_dose_idx;    /* gantry mm -> dose voxel index */
} rtd_beam;

/* EnergyStruct (src/energy_struct.h:13-31). cidd_matrix is [n_energies][n_energy_samples]. */
typedef struct rtd_luts {
    int32_t n_energy_samples;
    int32_t n_energies;
    const float* energies_per_u;   /* [n_energies] ascending */
    const float* peak_depths;      /* [n_energies] mm */
    const float* scale_facts;      /* [n_energies] samples per mm */
    const float* cidd_matrix;      /* [n_energies*n_energy_samples] cumulative IDD */
    int32_t n_density_samples;
    float density_scale_fact;
    const float* density_vector;   /* mass density vs (HU+1000)*scale */
    int32_t n_sp_samples;
    float sp_scale_fact;
    const float* sp_vector;        /* relative stopping power vs (HU+1000)*scale */
    int32_t n_rrl_samples;
    float rrl_scale_fact;
    const float* rrl_vector;       /* 1/X0 per unit density vs density*scale */
    /* NUCLEAR_CORR only (energy_struct.h:33-36, energy_reader.cpp:103-162); may be NULL when options.nuclear_corr == 0 */
    const float* nuc_weight_matrix;   /* [n_energies*n_energy_samples] fraction of the dose in the nuclear halo        */
    const float* nuc_sq_sigma_matrix; /* [n_energies*n_energy_samples] squared sigma of the halo, mm^2                  */
} rtd_luts;

/*
 * The reference's compile-time physics switches (CMakeLists.txt:36-79) as run-time options.
 * rtd_default_options() returns the reference defaults.
 */
typedef struct rtd_options {
    int32_t dose_to_water;      /* DOSE_TO_WATER (ON)  kernel_wrapper.cu:314-318 */
    int32_t nozzle;             /* NOZZLE (ON); 0 = NO_NOZZLE  fill_idd_and_sigma_params.cu:74-83 */
    float bp_depth_cutoff;      /* BP_DEPTH_CUTOFF   1.05 */
    float conv_sigma_cutoff;    /* CONV_SIGMA_CUTOFF 3.0  */
    float ks_sigma_cutoff;      /* KS_SIGMA_CUTOFF   3.0  */
    float ray_weight_cutoff;    /* RAY_WEIGHT_CUTOFF 1.0  */
    int32_t fine_grained_timing;/* FINE_GRAINED_TIMING: per-stage hipEvent buckets */
    int32_t nuclear_corr;       /* NUCLEAR_CORR (OFF): RTD_NUC_* below. CMakeLists.txt:57-69                             */
    int32_t reserved[3];
} rtd_options;

/*
 * NUCLEAR_CORR: a second, broad Gaussian per spot (the nuclear halo) on a grid at spot resolution. The reference ships this
 * path switched OFF and unfinished: the constants of the variants carry the comment "CORRECT ALL THESE"
 * (kernel_wrapper.cu:230), and the fill kernel is constructed with a nuclear memory step of 0 (:925, 7th argument), so every
 * step of a ray overwrites the SAME nuclear voxel (:367-373) and the halo reaches the dose only through BEV slice 0, i.e. only
 * when the beam starts inside the patient. This engine restates what that code does, quirk included (the primary dose loses the
 * nuclear fraction; the halo deposit is whatever slice 0 receives); it does not repair it.
 */
enum { RTD_NUC_OFF = 0, RTD_NUC_SOUKUP = 1, RTD_NUC_FLUKA = 2, RTD_NUC_GAUSS_FIT = 3 };

/*
 * Per-field timing record; bucket names follow the reference's FINE_GRAINED_TIMING printout
 * (kernel_wrapper.cu:1298-1307). All values in milliseconds, measured with hipEvents on the
 * handle's stream. Only filled when options.fine_grained_timing != 0 (else total_ms only).
 */
typedef struct rtd_timing {
    float raytracing_ms;          /* "Time to trace ... rays"                    */
    float prepare_energy_loop_ms; /* plan + BEV zero + spot->ray convolution     */
    float fill_idd_sigma_ms;      /* "Time depositing IDD and calculating sigma" */
    float prepare_superp_ms;      /* tile radius classification + batching       */
    float superp_ms;              /* "Time executing superposition"              */
    float transforming_ms;        /* "Kernel time to transform ... voxels"       */
    float total_ms;               /* first kernel to last kernel of the field    */
    int32_t superp_launches;      /* number of superposition kernel launches     */
    float superp_kernel_ms;       /* the dominant kernel alone (k_superpose_mfma), inside superp_ms */
    uint32_t ray_dims[2];         /* what the reference's timing lines quote: "trace WxH rays S steps", "N time(s)" per   */
    uint32_t steps;               /* layer, "transform N voxels" (kernel_wrapper.cu:1298-1307)                           */
    uint32_t n_layers;
    int64_t transfer_voxels;      /* voxels of the dose box the transfer visited */
    int32_t reserved[2];
} rtd_timing;

/* Geometry and cut-off scalars of the last computed field (for logs, tests and the roofline model). */
typedef struct rtd_field_info {
    uint32_t ray_dims[3];         /* primRayDims (W, H, L)  kernel_wrapper.cu:659 */
    float ray_offset[3];          /* primRayOffset          kernel_wrapper.cu:654 */
    float ray_res[3];             /* primRayRes             kernel_wrapper.cu:623 */
    int32_t beam_first_inside;    /* kernel_wrapper.cu:781-784 */
    int32_t beam_first_outside;   /* kernel_wrapper.cu:785-787 */
    int32_t beam_first_guaranteed_passive; /* kernel_wrapper.cu:796 */
    int32_t beam_first_calculated_passive; /* kernel_wrapper.cu:955-957 */
    int32_t bbox_min[3];          /* kernel_wrapper.cu:1207 */
    int32_t bbox_max[3];          /* kernel_wrapper.cu:1208 */
    int64_t live_steps;           /* sum over layers of (layerFirstPassive - beamFirstInside) */
    int32_t max_radius;           /* largest tile radius over all layers */
    int32_t dose_box_min[3];      /* sub-box of bbox that this field can have changed: the image of the BEV rectangle that */
    int32_t dose_box_max[3];      /* carries dose (what rtd_field_clear_dose clears; what has to cross PCIe); with nuclear_corr
                                     the whole grid (the halo's own box is wider and known on the device only)             */
    int32_t uniform_sigma;        /* 1: every (layer, step) slice had one sigma over its live rays (a water phantom) and the
                                     superposition ran as a separable convolution (k_superpose_uniform); decided on the device */
} rtd_field_info;

typedef struct rtd_handle_s* rtd_handle;
typedef struct rtd_field_s* rtd_field;

uint32_t rtd_abi_version(void);
void rtd_default_options(rtd_options* out);

/* Replaces cudaFree(0) context init (kernel_wrapper.cu:388); --gpu_id is finally honoured (config.cpp:13-15). */
int rtd_create(int device_id, rtd_handle* out);
int rtd_destroy(rtd_handle h);
const char* rtd_last_error(rtd_handle h);
/* Message for a failure that happened before a handle existed (rtd_create). */
const char* rtd_global_error(void);

int rtd_set_options(rtd_handle h, const rtd_options* opt);

/* LUT upload, replaces the texture set-up (kernel_wrapper.cu:453-537). Arrays are copied. */
int rtd_set_luts(rtd_handle h, const rtd_luts* luts);
/* Reads the reference's LUT text layout from a directory (energy_reader.cpp:12-101) and uploads it.
 * water_cube_test != 0 selects radiation_length_inc_water.txt (energy_reader.cpp:77-93). With options.nuclear_corr set (before
 * this call) the variant's nuclear_weights_and_sigmas_*.txt is read and checked against the IDD table too (:103-162). */
int rtd_load_luts_dir(rtd_handle h, const char* dir, int water_cube_test);

/* CT upload, replaces cudaMemcpy3D into the 3-D texture (kernel_wrapper.cu:420-451).
 * hu_plus_1000: [dims[2]][dims[1]][dims[0]] float, host memory. */
int rtd_set_ct(rtd_handle h, const float* hu_plus_1000, const uint32_t dims[3]);
/* rtd_set_ct without the copy: hu_plus_1000 stays with the caller — valid and unchanged until the last compute that uses it has
 * been finished — and every field uploads, in front of its tracer, only the index box of the volume its rays can sample (a beam
 * of the 512^3 bench plan reads a tenth of the CT; the upload of all 537 MB was 9.4 of the 13 ms of a one-field call). */
int rtd_set_ct_deferred(rtd_handle h, const float* hu_plus_1000, const uint32_t dims[3]);
/* Same, for a CT already resident on this handle's device (not copied, not owned). */
int rtd_set_ct_device(rtd_handle h, const float* dev_hu_plus_1000, const uint32_t dims[3]);

/*
 * The reference-shaped call: for every beam run the whole path and ACCUMULATE into dose_inout
 * (host memory, [dims[2]][dims[1]][dims[0]]): upload dose (kernel_wrapper.cu:542), beam loop (:601-1312),
 * download (:1318). timing may be NULL; else it must point to n_beams records.
 */
int rtd_compute(rtd_handle h, const rtd_beam* beams, int n_beams, float* dose_inout,
                const uint32_t dose_dims[3], rtd_timing* timing);

/*
 * Split form for callers that keep data resident (benchmarks, multi-GPU plans):
 * rtd_field_create   host geometry + workspace allocation + spot-weight upload for one beam
 *                    (kernel_wrapper.cu:612-734, 851-852);
 * rtd_field_compute  launches every kernel of the field on the handle's stream and accumulates into
 *                    dev_dose (device memory of this handle's device). Asynchronous: no host sync,
 *                    no allocation, so it can be captured into a hipGraph;
 * rtd_field_finish   waits for the field's last launch (not for later work on the stream) and reports device-side
 *                    errors (radius overflow), timing and geometry of that launch;
 * rtd_field_clear_dose  zeroes, on the stream, exactly the voxels of dev_dose that the field's last
 *                    rtd_field_compute could have changed (its device-side dose box). A plan loop that starts
 *                    every iteration from an all-zero volume (the reference uploads a zero dose image per call,
 *                    main.cu:192-206, kernel_wrapper.cu:542) restores it with this instead of clearing all of it.
 */
int rtd_field_create(rtd_handle h, const rtd_beam* beam, const uint32_t dose_dims[3], rtd_field* out);
int rtd_field_compute(rtd_handle h, rtd_field f, float* dev_dose);
int rtd_field_finish(rtd_handle h, rtd_field f, rtd_timing* timing, rtd_field_info* info);
int rtd_field_clear_dose(rtd_handle h, rtd_field f, float* dev_dose);
int rtd_field_destroy(rtd_handle h, rtd_field f);
/* Like rtd_field_destroy, but the device workspace stays with the handle and is taken over by the next rtd_field_create of
 * the same shape (ray grid, steps, layers, spot map): a plan of similar beams allocates once, where the reference mallocs and
 * frees ~20 buffers per beam (kernel_wrapper.cu:685-734, 1265-1281). rtd_compute uses it between its beams. */
int rtd_field_release(rtd_handle h, rtd_field f);

/*
 * The two halves of rtd_field_compute, and the pieces a multi-GPU plan is made of. The beam loop of the reference
 * (kernel_wrapper.cu:601) shares nothing between beams but the final `+=` into the dose volume (:92), so fields shard one
 * per GPU; what then has to cross the xGMI links is NOT the dose: the beam's-eye-view (BEV) dose of a field — the cube the
 * reference copies into a 3-D texture before primTransfDiv samples it (:1107-1141) — is ~10 MB where the dose box it
 * turns into is 60-83 MB (512^3 grid). So a field's BEV slab travels, and every GPU runs the transfer of EVERY field into its
 * own slab of the dose volume:
 * rtd_field_compute_bev   all kernels up to the BEV dose (:766-1105); asynchronous;
 * rtd_field_transfer      fan -> dose-grid transfer (primTransfDiv, :69-97) of the field's BEV dose into dev_dose, restricted
 *                         to the inclusive dose-index box [clip_min, clip_max] (NULL = the whole grid); asynchronous; may be
 *                         called several times (several volumes / boxes) for one BEV dose;
 * rtd_field_wait_plan     waits until the field's device-side plan is known (entry / passive steps, BEV rectangle, dose
 *                         box) — the superposition may still be running — and returns the size of the message below;
 * rtd_field_export_bev    packs [state record | non-zero block of the BEV dose] into dev_buf (device memory, capacity
 *                         bytes; rtd_bev_message_bound() is always enough); asynchronous, after rtd_field_compute_bev;
 * rtd_field_create_remote a field object for a beam computed on ANOTHER GPU: host geometry only, no workspace
 *                         (needs neither LUTs nor CT on this handle);
 * rtd_field_attach_bev    the remote field samples the message at dev_buf (device memory of this handle's device, not
 *                         copied: it must stay valid until the transfers that use it have run); then rtd_field_transfer,
 *                         rtd_field_clear_dose_box and rtd_field_finish work on it as on a local field.
 * rtd_field_clear_dose_box zeroes the part of the field's dose box inside [clip_min, clip_max].
 */
int rtd_field_compute_bev(rtd_handle h, rtd_field f);
int rtd_field_transfer(rtd_handle h, rtd_field f, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]);
/* rtd_field_transfer for the FIRST field of a plan, into a volume that is zero everywhere except possibly inside this
 * field's dose box (e.g. the volume the same field was transferred into by the previous plan): the box is written — dose
 * or zero — instead of accumulated into, which replaces rtd_field_clear_dose + the read half of the read-modify-write. */
int rtd_field_transfer_init(rtd_handle h, rtd_field f, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]);
/* Several fields — own BEV doses and / or attached slabs — into ONE box of the dose grid in one launch: every voxel of the
 * inclusive box [box_min, box_max] (NULL = the whole grid) is WRITTEN with 0 + field 0 + field 1 + ..., the positive samples
 * in list order: bit for bit what rtd_field_transfer of each field in turn accumulates into a zeroed box (the beam loop of
 * kernel_wrapper.cu:601 with its primTransfDiv launches, :1216), without the read-modify-write passes and without a clear.
 * Voxels outside the box are not touched: a GPU of a multi-GPU plan passes its slab of the volume, cut to the bounding box of
 * the fields' dose boxes. At most 16 fields; not with nuclear_corr. rtd_field_finish of the LAST own field of the list (the
 * first field if all are remote) waits for the launch and reports its duration as transforming_ms. */
int rtd_fields_transfer_init(rtd_handle h, const rtd_field* fields, uint32_t n_fields, float* dev_dose, const int32_t box_min[3],
                             const int32_t box_max[3]);
int rtd_field_wait_plan(rtd_handle h, rtd_field f, rtd_field_info* info, size_t* packed_bytes);
size_t rtd_bev_message_bound(rtd_handle h, rtd_field f);
int rtd_field_export_bev(rtd_handle h, rtd_field f, void* dev_buf, size_t capacity);
int rtd_field_create_remote(rtd_handle h, const rtd_beam* beam, const uint32_t dose_dims[3], rtd_field* out);
int rtd_field_attach_bev(rtd_handle h, rtd_field remote_field, const void* dev_buf);
int rtd_field_clear_dose_box(rtd_handle h, rtd_field f, float* dev_dose, const int32_t clip_min[3], const int32_t clip_max[3]);

/* Page-locks / unlocks a caller-owned host buffer so that rtd_set_ct / rtd_compute copy at full PCIe rate — what
 * HostPinnedImage3D's constructor and destructor do with cudaHostRegister (host_image_3d.cuh:23-32, 45-48). */
int rtd_host_register(void* host_ptr, size_t bytes);
int rtd_host_unregister(void* host_ptr);

/* Device buffers owned by the handle (so a C caller needs no HIP headers). */
int rtd_device_alloc(rtd_handle h, size_t bytes, void** dev_ptr);
int rtd_device_free(rtd_handle h, void* dev_ptr);
int rtd_device_zero(rtd_handle h, void* dev_ptr, size_t bytes);
int rtd_copy_to_device(rtd_handle h, void* dev_dst, const void* host_src, size_t bytes);
int rtd_copy_to_host(rtd_handle h, void* host_dst, const void* dev_src, size_t bytes);
int rtd_sync(rtd_handle h);
/* The hipStream_t the handle launches on (as void*), for event timing by the caller. */
void* rtd_stream(rtd_handle h);
/* Use an externally created hipStream_t (e.g. torch's current stream); NULL restores the handle's own. */
int rtd_set_stream(rtd_handle h, void* hip_stream);

/*
 * Introspection for parity tests: copy a named intermediate of the last rtd_field_compute to host.
 * Names: "density" "wepl" [S][H][W] float; "first_inside" "first_outside" [H][W] int32;
 * "wepl_min" [S] float; "ray_weights" [L][H][W] float; "idd" "rsigma" [L][S][H][W] float;
 * "first_passive" [L][H][W] int32; "tile_radius" [L][S][tilesY][tilesX] uint8 (0xFF = not classified);
 * "eff_radius" [L][34] int32 (batch radius per tile radius); "bev" [S][H+64][W+64] float;
 * "layer_plan" [L][8] float (energyIdx, scaleFact, peakDepth, entrySigmaX, entrySigmaY, afterLast, 0, 0).
 * Returns the number of bytes the buffer holds via *bytes_needed when host_out is NULL.
 */
int rtd_field_fetch(rtd_handle h, rtd_field f, const char* name, void* host_out, size_t bytes,
                    size_t* bytes_needed);

/*
 * ---- Multi-GPU plans behind the boundary (SURVEY.md 8(b) "Threading": one handle and one host thread per device) ----
 *
 * rtd_plan is the reference-shaped call on several GPUs of one process: the 4-beam cudaWrapperProtons of the C++ shim uses
 * 4 GPUs. Beams are dealt round-robin to the devices; every device computes the BEV dose of its beams, the packed BEV
 * slabs (~10 MB each) are pulled by the other devices over xGMI (peer copies), and every device transfers ALL beams, in
 * beam order, into its own z-slab of the dose volume — uploaded from and downloaded to the caller's host buffer by that
 * device alone, so the PCIe legs (the bulk of the reference's "total global execution time", kernel_wrapper.cu:410-414,
 * 1356-1360) run in parallel as well. No dose data crosses between GPUs, and every voxel sees the same `+=` order as on one
 * GPU: the result equals rtd_compute bit for bit.
 * device_ids may repeat an id (several handles on one GPU) — used by the tests on one-GPU machines.
 * Environment: RTD_PLAN_TRANSPORT=rccl moves the slabs with ncclBroadcast (communicators from ncclCommInitAll, librccl opened on
 * demand; distinct device ids required) instead of peer copies.
 */
typedef struct rtd_plan_s* rtd_plan_t;

typedef struct rtd_plan_timing {
    float total_ms;        /* wall clock of rtd_plan_compute: dose up, all beams, dose down ("Total global execution time
                              (excluding GPU initialisation)", kernel_wrapper.cu:1356-1360)                              */
    float upload_ms;       /* slowest device: slab allocation + issuing the upload of the dose block the plan can change (the
                              copy itself runs beside the kernels; the transfers wait for it)                                */
    float bev_ms;          /* slowest device: its beams up to the BEV dose, slabs exported                               */
    float exchange_ms;     /* slowest device: pulling the other devices' slabs                                          */
    float transfer_ms;     /* slowest device: transfers of all beams into its slab (includes waiting for the upload)    */
    float download_ms;     /* slowest device: the changed block of its slab, device -> host                             */
    int32_t n_devices;
    int32_t reserved[3];
} rtd_plan_timing;

int rtd_plan_create(const int* device_ids, int n_devices, rtd_plan_t* out);
int rtd_plan_destroy(rtd_plan_t p);
const char* rtd_plan_last_error(rtd_plan_t p);
int rtd_plan_set_options(rtd_plan_t p, const rtd_options* opt);
int rtd_plan_set_luts(rtd_plan_t p, const rtd_luts* luts);                          /* replicated on every device */
int rtd_plan_load_luts_dir(rtd_plan_t p, const char* dir, int water_cube_test);
int rtd_plan_set_ct(rtd_plan_t p, const float* hu_plus_1000, const uint32_t dims[3]);            /* replicated, uploads in parallel */
int rtd_plan_set_ct_deferred(rtd_plan_t p, const float* hu_plus_1000, const uint32_t dims[3]);   /* rtd_set_ct_deferred on every device */
/* per_beam_timing: NULL or n_beams records (filled by the device that computed the beam). */
int rtd_plan_compute(rtd_plan_t p, const rtd_beam* beams, int n_beams, float* dose_inout, const uint32_t dose_dims[3],
                     rtd_timing* per_beam_timing, rtd_plan_timing* plan_timing);

#ifdef __cplusplus
}
#endif

#endif /* RTD_H */
