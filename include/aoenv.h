/*
 * aoenv.h -- C ABI of the MI355X-native batched adaptive-optics environment (libaoenv.so).
 *
 * The reference (artiom-matvei/RLAO) is pure Python: there is no native FFI to mirror.  Each entry
 * point below names the reference *Python* interface it replaces, so that a maintainer can bind the
 * library behind drl4ao's gym-style env (INTEGRATION.md shows the ctypes stub):
 *
 *   OOPAO/  = drl4ao/AO_OOPAO/OOPAO/          MAIN/ = drl4ao/MAIN_CODE/
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, non-zero on error and never throws or
 *     aborts; aoenv_last_error() returns a thread-local description of the last failure.
 *   - "d_" pointers are DEVICE pointers owned by the caller (PyTorch tensors, hipMalloc, ...), laid
 *     out contiguously, element type = the environment's dtype (AOENV_F32 float / AOENV_F64 double).
 *   - "h_" pointers are HOST pointers; constants are always handed over as float64 / int32 / uint8 and
 *     converted to the environment's dtype on upload.
 *   - stream arguments are hipStream_t passed as void* (NULL = the default stream).  All work of a
 *     call is enqueued on that stream; no call synchronises the device except aoenv_download().
 *   - one host thread per AoEnv; one AoEnv per GPU shard of independent AO loops ("envs").
 */
#ifndef AOENV_H
#define AOENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOENV_ABI_VERSION 6

enum { AOENV_F32 = 0, AOENV_F64 = 1 };
enum { AOENV_WFS_SH = 0, AOENV_WFS_PYRAMID = 1 };

/* Geometry and loop constants of one shard.  Filled by the host (rlao_amd/calib.py) from the same
 * parameter-file keys the reference uses (MAIN/Conf/parameterFile_oopao_parser.py:19-79). */
typedef struct AoEnv AoEnv;   /* opaque */

typedef struct AoCfg {
    int32_t abi_version;     /* = AOENV_ABI_VERSION */
    int32_t dtype;           /* AOENV_F32 | AOENV_F64: arithmetic type of device state and kernels */
    int32_t n_env;           /* independent AO loops stepped in lock-step on this GPU */
    int32_t resolution;      /* R: telescope pupil pixels across the diameter (OOPAO/Telescope.py:138) */
    int32_t n_layer;         /* turbulence layers; 0 = no atmosphere (calibration shards) */
    int32_t layer_res;       /* N: interior size of a layer screen, N = R + 4 for fov 0 (OOPAO/Atmosphere.py:216-218) */
    int32_t n_inner;         /* 8N-16 : conditioning ring pixels Z (OOPAO/Atmosphere.py:267-274) */
    int32_t n_outer;         /* 4N+4  : regenerated ring pixels X */
    int32_t n_act;           /* actuators across the diameter, nSubap+1 (OOPAO/DeformableMirror.py:288) */
    int32_t n_valid_act;     /* A: controlled actuators */
    int32_t dm_separable;    /* 1: OPD = Gy C Gx^T (Cartesian Gaussian DM, no rotation); 0: dense modes GEMM */
    int32_t wfs_type;        /* AOENV_WFS_SH | AOENV_WFS_PYRAMID */
    int32_t n_subap;         /* lenslets (SH) / pupil samples (Pyramid) across the diameter */
    int32_t n_valid_subap;   /* SH: valid lenslets;  Pyramid: valid pixels per quadrant */
    int32_t n_signal;        /* length of wfs.signal */
    int32_t cam_res;         /* WFS camera frame is cam_res x cam_res */
    int32_t n_loop;          /* length of the total[] / residual[] telemetry (param['nLoop']) */
    int32_t max_group;       /* envs in consecutive groups of max_group share the centroid-threshold
                                maximum (1 in the loop; nMeasurements when emulating the batched
                                interaction-matrix measurement, OOPAO/ShackHartmann.py:605-672) */
    int32_t pyr_n_res;       /* Pyramid: padded FFT size nRes (OOPAO/Pyramid.py:251) */
    int32_t pyr_n_theta;     /* Pyramid: modulation points (1 = unmodulated, OOPAO/Pyramid.py:955, 976) */
    int32_t pyr_centering;   /* Pyramid: 1 = psfCentering (mask on 4 pixels, phasor), 0 = fftshift + 1-pixel mask */
    int32_t pyr_norm_valid;  /* Pyramid: 0 = 'slopesMaps_incidence_flux' (norm = frame.mean()), 1 = 'slopesMaps' */
    int32_t pyr_q_lo;        /* Pyramid: first row/column of quadrants 1 (and of the low side of 2, 4) in the frame */
    int32_t pyr_q_hi;        /* Pyramid: first row/column of the high-side quadrants (grabQuadrant, OOPAO/Pyramid.py:774-790) */
    int32_t layer_res_l[8];  /* per-layer N: 0 = layer_res.  With a field of view (the reference env builds its telescope with
                                fov = 1 arcsec, MAIN/OOPAOEnv/OOPAOEnv.py:129) a layer at altitude h lives on a grid of
                                N_l = ceil(R / D (D + 2 tan(fov / 2) h)) + 4 pixels with ring operators of its own
                                (OOPAO/Atmosphere.py:216-218, 277-286); its n_inner / n_outer are 8 N_l - 16 / 4 N_l + 4.  Layers on
                                different grids run the batched kernels (the fused step kernel needs one grid) */
    double  atm_wavelength;  /* 500e-9: wavelength the screens are expressed at (OOPAO/Atmosphere.py:134) */
    double  src_wavelength;  /* guide-star wavelength (OOPAO/Source.py:102) */
    double  leak;            /* leaky-integrator factor (MAIN/OOPAOEnv/OOPAOEnv.py:69) */
    double  threshold_cog;   /* SH centre-of-gravity threshold (OOPAO/ShackHartmann.py:42) */
} AoCfg;

/* Constant tables, uploaded once per geometry (aoenv_upload).  Element type on the host side in []. */
enum AoConst {
    AOENV_C_PUPIL = 0,       /* [u8  R*R]            Telescope.pupil                                  */
    AOENV_C_AB,              /* [f64 n_outer*(n_inner+n_outer)] rows = [A | B] (OOPAO/Atmosphere.py:284-286) */
    AOENV_C_INNER_IDX,       /* [i32 n_inner]  flat index into the (N+2)^2 screen of each Z pixel, mask order */
    AOENV_C_OUTER_IDX,       /* [i32 n_outer]  flat index of each X pixel, mask order                  */
    AOENV_C_LAYER_WEIGHT,    /* [f64 n_layer]  sqrt(fractionalR0) (OOPAO/Atmosphere.py:450)            */
    AOENV_C_DM_GX,           /* [f64 R*n_act]  separable influence factor along x                      */
    AOENV_C_DM_GY,           /* [f64 R*n_act]  separable influence factor along y                      */
    AOENV_C_DM_MODES,        /* [f64 R*R*A]    dense influence matrix dm.modes (only if !dm_separable)  */
    AOENV_C_ACT_IDX,         /* [i32 A]        iy*n_act+ix of each valid actuator (xvalid,yvalid)       */
    AOENV_C_WFS_AMP,         /* [f64 R*R]      sqrt(src.fluxMap) (x pupilReflectivity)                  */
    AOENV_C_SH_SUBAP_IDX,    /* [i32 n_valid_subap] SH: i*n_subap+j of each valid lenslet;
                                                    Pyramid: r*n_subap+c of each valid pixel of a quadrant (validI4Q) */
    AOENV_C_SH_REF,          /* [f64 2*n_valid_subap] SH: reference centroids (x block then y block);
                                                      Pyramid: referenceSignal_2D at the valid pixels             */
    AOENV_C_WFS_UNITS,       /* [f64 1]        slopes_units                                             */
    AOENV_C_RECON,           /* [f64 A*n_signal] reconstructor = M2C @ calib.M (MAIN/OOPAOEnv/OOPAOEnv.py:381) */
    AOENV_C_PYR_MASK,        /* [f64 nRes*nRes*2] exp(i m) of the pyramid mask, rounded to complex64 (OOPAO/Pyramid.py:323) */
    AOENV_C_PYR_TT,          /* [f64 n_theta*R*R] modulation tip/tilt phases, float32-rounded (OOPAO/Pyramid.py:964-970) */
    AOENV_C_RECON_FACTORS,   /* [f64 K*n_signal + A*K] optional: calib.M (K x nSig) then M2C (A x K) with reconstructor = M2C @ M
                                (MAIN/OOPAOEnv/OOPAOEnv.py:295, 381); enables the fused low-rank tail.  Upload after AOENV_C_RECON */
    AOENV_C_COUNT
};

/* Device buffers that can be inspected / overwritten (aoenv_download / aoenv_upload_state): the
 * environment state of SURVEY.md section 5 "checkpoint / resume" plus the stage boundaries the parity
 * tests compare.  Shapes are per shard, leading dimension n_env unless noted. */
enum AoBuf {
    AOENV_B_SCREEN = 0,      /* [n_layer][n_env][(N_l+2)^2]  layer.mapShift, layer after layer (stored as a torus: no flat device image) */
    AOENV_B_OPD_ATM,         /* [n_env][R*R]   atm.OPD_no_pupil                                          */
    AOENV_B_COEFS,           /* [n_env][A]     dm.coefs                                                  */
    AOENV_B_PHASE,           /* [n_env][R*R]   tel.src.phase (residual, pupil-masked, rad @ src)          */
    AOENV_B_FRAME,           /* [n_env][cam^2] wfs.cam.frame                                             */
    AOENV_B_SIGNAL,          /* [n_env][n_signal] wfs.signal                                             */
    AOENV_B_TOTAL,           /* [n_loop][n_env] env.total  (nm rms)                                      */
    AOENV_B_RESIDUAL,        /* [n_loop][n_env] env.residual (nm rms)                                    */
    AOENV_B_WFS_MAX,         /* [n_env]        max of the valid spot intensities (threshold reference)   */
    AOENV_B_XI,              /* [n_env][n_inner+n_outer] last [Z | xi] operand of the ring extrusion      */
    AOENV_B_MT_STATE,        /* [n_layer][n_env][625] uint32 (whatever the env dtype): the 624 MT19937 state words of
                                the layer's ring RandomState and its position (OOPAO/Atmosphere.py:201, 308)         */
    AOENV_B_COUNTERS,        /* [4] uint32: frame counter of the camera noise streams, 3 reserved                 */
    AOENV_B_DM_PREV,         /* [n_env][A]     env.dm_prev: the leaky integrator's state.  aoenv_step computes
                                dm.coefs = dm_prev * leak + action and copies it back to dm_prev; aoenv_set_coefs (dm.coefs = ...
                                from outside) leaves it alone, as in the reference (MAIN/OOPAOEnv/OOPAOEnv.py:314, 508-509), so
                                the trainers' episode prologue `env.dm.coefs = 0` (MAIN/PO4AO/mbrl.py:50) does not clear it */
    AOENV_B_COUNT
};

const char* aoenv_last_error(void);
int aoenv_abi_version(void);

/* Replaces: OOPAO() + the object construction half of set_params() (MAIN/OOPAOEnv/OOPAOEnv.py:19-73,
 * 121-246): allocates all device state for cfg->n_env loops on HIP device `device`.  The DM starts
 * flat (dm.coefs = 0), the atmosphere OPD at zero, the WFS reference at zero and units at 1. */
int aoenv_create(const AoCfg* cfg, int device, AoEnv** out);
int aoenv_destroy(AoEnv* env);

/* Replaces: the constant members the reference objects compute in their constructors (pupil, layer.A /
 * layer.B, dm.modes, wfs flux / reference / units, env.reconstructor ...).  `kind` is an AoConst;
 * `bytes` must match the table size implied by the AoCfg.  May be called again at any time (e.g. the
 * r0 setter re-uploads [A|B], OOPAO/Atmosphere.py:792-807). */
int aoenv_upload(AoEnv* env, int kind, const void* h_data, size_t bytes);

/* The ring tables AOENV_C_AB / AOENV_C_INNER_IDX / AOENV_C_OUTER_IDX of ONE layer, sized by that layer's grid (AoCfg.layer_res_l):
 * layer.A / layer.B and the masks of OOPAO/Atmosphere.py:262-286.  aoenv_upload() with these kinds serves every layer at once and
 * is only accepted when all layers share one grid. */
int aoenv_upload_layer(AoEnv* env, int kind, int layer, const void* h_data, size_t bytes);

/* Replaces: the windSpeed / windDirection setters (OOPAO/Atmosphere.py:829-873): per layer
 * ratio = (vX, vY) * samplingTime / pixel_size in pixels per frame (OOPAO/Atmosphere.py:362-363).
 * h_ratio is [n_layer][2] float64, shared by all envs of the shard.  `reset_buff` != 0 also clears the
 * sub-pixel accumulator (what notDoneOnce does after generateNewPhaseScreen). */
int aoenv_set_wind(AoEnv* env, const double* h_ratio, int reset_buff);

/* The same setters when every env has its OWN wind (a trainer that draws wind speed / direction per run, e.g.
 * MAIN/integrator_oopao_razor.py:41-44, batched): h_ratio is [n_layer][n_env][2] float64, |ratio| < 1 pixel per frame.
 * The shard switches to per-env clocks for good: accumulators, torus origins and warp taps of every (env, layer) live on
 * the device and are advanced there, by the same arithmetic as the shared host clock (an env stepped by its own clock is
 * bit-identical to a shard stepped with that wind); on every step one launch per layer advances the clocks and prepares the
 * ring operands of the envs that cross a pixel, and the ring GEMM runs over the whole shard.  aoenv_set_wind keeps working
 * afterwards (the same wind for every env); new screens reset accumulators and origins as for the shared clock.
 * Clock state for checkpoints: h_clock [n_layer][n_env][4] float64 = {ratio x, ratio y, buff x, buff y}. */
int aoenv_set_wind_env(AoEnv* env, const double* h_ratio, int reset_buff, void* stream);
int aoenv_get_clock_env(AoEnv* env, double* h_clock);
int aoenv_set_clock_env(AoEnv* env, const double* h_clock);

/* Replaces: atm.generateNewPhaseScreen(seed) (OOPAO/Atmosphere.py:560-592) for every env of the shard.
 *   h_screens [n_env][n_layer][N*N] float64: the new layer.phase screens (rad @ 500 nm), or NULL to keep
 *             the current interior (not with per-env clocks).  Layers on grids of their own (layer_res_l): layer-major
 *             blocks, layer l = [n_env][N_l*N_l];
 *   h_ring_seeds [n_env][n_layer] uint32: seeds of the per-layer ring RandomState (seed + 1000*layer);
 * seeds every MT19937 stream, draws the first ring X = A.Z + B.xi on the device, rebuilds mapShift,
 * clears the sub-pixel accumulators and refreshes atm.OPD (fill_phase_support + set_OPD). */
int aoenv_new_screens(AoEnv* env, const double* h_screens, const uint32_t* h_ring_seeds, void* stream);

/* Same, with the screens themselves generated on the device (OOPAO/phaseStats.py:190-318 ft_sh_phase_screen:
 * FFT screen + 3 sub-harmonic grids, both seeded with the same RandomState as the reference does):
 *   h_screen_seeds [n_env][n_layer] uint32: seed + layer of each layer's own RandomState (OOPAO/Atmosphere.py:574);
 *   r0 [m @ 500 nm], L0 [m], pixel_size = layer.D / layer.resolution [m] (= telescope pixel size for fov 0).
 * MT19937 + legacy Gaussian stream, float64 FFT: agrees with the NumPy generator to ~1e-12 rad. */
int aoenv_new_screens_device(AoEnv* env, const uint32_t* h_screen_seeds, const uint32_t* h_ring_seeds, double r0,
                             double L0, double pixel_size, void* stream);

/* Replaces: atm.update(OPD) with a user-defined OPD (OOPAO/Atmosphere.py:421-425) and
 * tel.OPD = ... in the WFS calibration (OOPAO/ShackHartmann.py:296-297).  h_opd [n_env][R*R] float64
 * (no pupil applied); only meaningful while n_layer == 0 or until the next aoenv_step. */
int aoenv_set_atm_opd(AoEnv* env, const double* h_opd, void* stream);

/* Replaces: dm.coefs = v (OOPAO/DeformableMirror.py:534-570).  h_coefs [n_env][A] float64 or NULL for
 * dm.coefs = 0 (MAIN/PO4AO/mbrl.py:50).  env.dm_prev (AOENV_B_DM_PREV) is not touched. */
int aoenv_set_coefs(AoEnv* env, const double* h_coefs, void* stream);

/* Replaces: tel*dm*wfs with no atmosphere update (MAIN/PO4AO/mbrl.py:52; OOPAO/Telescope.py:533-544,
 * 476-485): residual phase = (atm.OPD_no_pupil + dm.OPD) * pupil, WFS measurement, signal. */
int aoenv_measure(AoEnv* env, void* stream);

/* Replaces: atm.update() on its own (OOPAO/Atmosphere.py:439-477; the open-loop scripts call it between propagations):
 * every layer moves by one frame -- ring extrusions included -- and atm.OPD_no_pupil (AOENV_B_OPD_ATM) and the residual
 * phase are re-derived.  No WFS measurement, no controller.  aoenv_step = this + aoenv_measure + the glue. */
int aoenv_atm_update(AoEnv* env, void* stream);

/* Replaces: env.reset_soft() (MAIN/OOPAOEnv/OOPAOEnv.py:82-86): obs = vec_to_img(-R @ wfs.signal) * 1e6.
 * d_obs [n_env][n_act][n_act]. */
int aoenv_reset_soft(AoEnv* env, void* d_obs, void* stream);

/* Replaces: env.step(i, action) (MAIN/OOPAOEnv/OOPAOEnv.py:485-536; Razor twin OOPAOEnvRazor.py:474-514).
 *   i        frame index: total[i], residual[i] are written (i < n_loop)
 *   d_action [n_env][n_act][n_act]  DM increment image in micrometres
 *   d_obs    [n_env][n_act][n_act]  out: reconstructed residual image (micrometres)
 *   d_frame  [n_env][cam][cam] or NULL  out: wfs.cam.frame (Papyrus 6-tuple) / NULL (Razor 5-tuple)
 *   d_reward [n_env]  out: -||obs||_2        d_strehl [n_env]  out: exp(-var(phase[pupil]))
 * Order of effects as in the reference: turbulence advances, the WFS sees the command latched by the
 * previous call, then dm.coefs <- leak * dm_prev + 1e-6 * img_to_vec(action) and dm_prev <- dm.coefs. */
int aoenv_step(AoEnv* env, int i, const void* d_action, void* d_obs, void* d_frame, void* d_reward,
               void* d_strehl, void* stream);

/* Closed-loop driver of MAIN/integrator_oopao_razor.py:66-91 kept on the device: n_steps iterations of
 * action = gain * obs; obs, reward, strehl = step(i0 + k, action).  d_obs is in/out. */
int aoenv_run_integrator(AoEnv* env, int i0, int n_steps, double gain, void* d_obs, void* d_frame,
                         void* d_reward, void* d_strehl, void* stream);

/* Replaces: tel.computePSF(zeroPaddingFactor) (OOPAO/Telescope.py:258-357) of the current residual phase (tel.src.phase):
 * d_psf [n_env][M][M], M = zero_padding * R (even), env dtype: the short-exposure PSF of every env.  As the reference does for even
 * image sizes (oversampling 2, :303-305), the field is transformed at N = 2 M and |fftshift(fft2(E phasor)) / N|^2 is sum-binned 2 x 2.
 * Science-path rendering (MAIN/OOPAOEnv/OOPAOEnv.py:473-482, integrator_network.py:93-95); not part of the step. */
int aoenv_compute_psf(AoEnv* env, int zero_padding, void* d_psf, void* stream);

/* Replaces: the WFS camera settings wfs.cam.{photonNoise, readoutNoise, QE, darkCurrent, integrationTime, FWC, bits, gain,
 * sensor} (OOPAO/Detector.py:19-60; Papyrus: photonNoise = True, MAIN/OOPAOEnv/OOPAOEnv.py:379; Razor:
 * OOPAOEnvRazor.py:243-250, 333).  The frame of every measurement then goes through integrate() + readout()
 * (Detector.py:232-301) before the slopes are computed.  The reference seeds its noise generators from the wall clock;
 * here every pixel of every frame draws from a counter-based stream keyed by `seed` and indexed by (pixel, env_index_offset
 * + env, frame number): reproducible, and the same for an env wherever it sits in a batch.  NULL = ideal detector.
 * The frame number keeps counting across calls (a camera setting changed in mid-run does not replay earlier noise frames); it
 * restarts at 0 only when `seed` changes, and it is part of the checkpoint (AOENV_B_COUNTERS).  Work is enqueued on `stream`. */
typedef struct AoDetector {
    int32_t photon_noise;    /* cam.photonNoise */
    int32_t bits;            /* ADC bits, 0 = None (needs fwc > 0 when set) */
    int32_t emccd;           /* sensor == 'EMCCD': gain before the read-out noise; CCD / CMOS: after */
    int32_t env_index_offset; /* global index of env 0 of this shard (multi-GPU sharding) */
    double  qe;              /* cam.QE */
    double  dark_electrons;  /* cam.darkCurrent * cam.integrationTime */
    double  fwc;             /* cam.FWC, 0 = None */
    double  gain;            /* cam.gain */
    double  readout_noise;   /* cam.readoutNoise [e- rms] */
    uint64_t seed;
} AoDetector;
int aoenv_set_detector(AoEnv* env, const AoDetector* cfg, void* stream);

/* Episode return on the device (the sum of rewards the trainers accumulate on the host, MAIN/PO4AO/mbrl.py:64-89):
 * d_return [n_env] is a caller-owned device buffer of the env dtype; every aoenv_step / integrator step adds its reward
 * to it.  NULL detaches it.  The caller zeroes it at the start of an episode. */
int aoenv_set_return_accumulator(AoEnv* env, void* d_return);

/* State access (get_state / set_state = checkpoint / resume of the env; also the stage boundaries
 * compared by the parity tests).  The complete loop state is {AOENV_B_SCREEN, aoenv_get_buff, AOENV_B_MT_STATE,
 * AOENV_B_COEFS, AOENV_B_COUNTERS} plus the caller's last observation; aoenv_upload_state(AOENV_B_SCREEN) takes the
 * logical layer.mapShift (as aoenv_download returns it) and re-derives the clip range.  `which` is an AoBuf.  aoenv_buffer returns the device pointer and size in bytes;
 * aoenv_download copies to a host buffer of the env dtype and synchronises the stream. */
int aoenv_buffer(AoEnv* env, int which, void** d_ptr, size_t* bytes);
int aoenv_download(AoEnv* env, int which, void* h_dst, size_t bytes, void* stream);
int aoenv_upload_state(AoEnv* env, int which, const void* h_src, size_t bytes, void* stream);
/* host-side atmosphere clock: h_buff [n_layer][2] float64 (layer.buff, OOPAO/Atmosphere.py:392-404) */
int aoenv_get_buff(AoEnv* env, double* h_buff);
int aoenv_set_buff(AoEnv* env, const double* h_buff);

/* Implementation switches (parity tests compare the specialised kernels with the generic ones). */
enum AoOption {
    AOENV_OPT_FAST_WFS = 0,  /* 1 (default): register-resident 6 px/lenslet SH kernel; 0: generic LDS kernel */
    AOENV_OPT_MFMA_GEMM = 1, /* 1 (default): float32 split-K MFMA contractions; 0: generic tiled VALU kernel */
    AOENV_OPT_STORE_ATM_OPD = 3, /* 1: every step also writes atm.OPD_no_pupil to AOENV_B_OPD_ATM; 0 (default): it is
                                re-derived from the screens when it is downloaded */
    AOENV_OPT_FUSED_TAIL = 4, /* 1 (default): when AOENV_C_RECON_FACTORS is uploaded, SH centroid + low-rank R.s + epilogue run as
                                 one per-env kernel; 0: separate centroid / MFMA GEMM / epilogue kernels */
    AOENV_OPT_FUSED_STEP = 5, /* 1 (default): float32 SH shards with 6 px per lenslet, <= 336 valid lenslets, R <= 128 and uploaded
                                 AOENV_C_RECON_FACTORS run the whole step as ONE kernel, one workgroup per env, every intermediate in LDS;
                                 0: the separate phase / spots / tail kernels */
    AOENV_OPT_DEFER_RING = 6, /* 1 (default): on a step where a layer crosses a pixel, the fused step kernel itself writes the new
                                 border ring of the screen (sum of the ring GEMM's slabs); 0: a separate scatter launch */
    AOENV_OPT_COEFS_IMAGE = 7, /* 1: the per-env part of the DM surface is formed once per step by a small kernel instead of once per
                                 tile of the separate phase kernel: float32 shards Gy.C on the matrix cores (MFMA operand layout),
                                 float64 shards the scattered command image (always on above 1024 actuators: ELT-size DMs);
                                 0 (default below): every tile workgroup does it itself */
    AOENV_OPT_FACTORED_RECON = 8, /* 1 (default): with AOENV_C_RECON_FACTORS uploaded, the batched (non-fused) reconstruction is the
                                 chained product v = M2C (M s) instead of the dense reconstructor; 0: dense */
    AOENV_OPT_RING_LOOKAHEAD = 9, /* 1 (default): float32 fused step, shared clock: the ring pipeline -- the operand [Z | xi] of a layer's NEXT
                                 pixel crossing is put together while the current one is served: xi by extra workgroups of the ring GEMM's
                                 launch, Z by the step kernel once it has written the ring; a crossing step then launches the GEMM and
                                 nothing else in front of the step kernel (bit-identical results).  0: gather + draw in a launch of
                                 their own in front of the GEMM.  [Round-2 note: the same work one crossing ahead on a SECOND STREAM was
                                 slower -- with one 1024-lane workgroup per CU the side stream finds no free CU (5.59 -> 4.97 M env-steps/s)] */
    AOENV_OPT_FAST_TRIG = 2  /* 1 (default): v_sin/v_cos after Cody-Waite reduction in the float32 SH kernel; 0: sincosf */
};
int aoenv_set_option(AoEnv* env, int option, int value);

/* Which path aoenv_step would take with the tables, options and camera as they are now: 1 = the whole step as one kernel
 * (float32 Shack-Hartmann, 6 pixels per lenslet, <= 336 valid lenslets, R <= 128, <= 32 actuators across, <= 52 modes, reconstructor
 * factors uploaded), 0 = the batched kernels (3-5 launches per step, roughly half the speed at small geometries).  A caller
 * outside the envelope is told instead of finding out from a profile. */
int aoenv_fused_step_active(AoEnv* env);

/* Per-kernel timing (bench.py roofline leg).  While enabled, every kernel launch of aoenv_step /
 * aoenv_measure is bracketed by a hipEvent pair recorded on the launch stream; aoenv_profile_read
 * synchronises the stream and returns the summed elapsed milliseconds and the launch count of each
 * AoKernel.  aoenv_profile(env, 0|1) also clears the recorded events. */
enum AoKernel {
    AOENV_K_SHIFT_GATHER = 0, AOENV_K_MT_NORMAL, AOENV_K_GEMM_RING, AOENV_K_SCATTER, AOENV_K_PHASE,
    AOENV_K_SH_SPOTS, AOENV_K_SH_CENTROID, AOENV_K_GEMM_RECON, AOENV_K_RECON_FINISH, AOENV_K_PYRAMID, AOENV_K_SH_TAIL, AOENV_K_ENV_STEP,
    AOENV_K_DETECTOR, AOENV_K_COUNT
};
int aoenv_profile(AoEnv* env, int enable);
int aoenv_profile_read(AoEnv* env, double* h_ms, int32_t* h_count, void* stream);

/* Test hook: draw `n` (even) values of RandomState(seed).normal(size=n) with the device MT19937 +
 * legacy polar generator into h_out (float64), to pin the stream against NumPy. */
int aoenv_test_normal(int device, uint32_t seed, int n, int n_calls, double* h_out);

/* Test hooks of the camera's photon-noise sampler (rlao_amd/csrc/poisson_alias.hpp; replaces the NumPy draw of
 * OOPAO/Detector.py:204-206, whose law -- not whose stream -- is what can be matched: the reference seeds it from the wall clock).
 * aoenv_test_poisson_table: the alias tables as the kernels read them (host only, no GPU needed): h_words = size in 32-bit
 *   words; h_out (cap_words >= that) receives header {fine rows, coarse rows, words, 0}, row descriptors {first entry, kmin << 16 |
 *   cells} and entries {23-bit threshold << 9 | alias}.  A CPU test rebuilds every row's outcome probabilities from them.
 * aoenv_test_poisson: h_out[i] = one draw of Poisson(h_lambda[i]) by the cameras' code path, stream (seed, frame); lmax > 0 lowers
 *   the photon count from which PTRS takes over (a multiple of 32, at least 32), 0 = the table's own end (1024). */
int aoenv_test_poisson_table(uint32_t* h_out, size_t cap_words, size_t* h_words);
int aoenv_test_poisson(int device, const float* h_lambda, int n, uint64_t seed, uint32_t frame, float lmax, float* h_out);

#ifdef __cplusplus
}
#endif
#endif
