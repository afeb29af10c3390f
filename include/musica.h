/*
 * musica.h — C ABI of libmusica_hip.so, the MI355X-native replacement for the
 * MUSICA Laplacian-pyramid contrast-amplification path of the reference
 * (class VulkanProcessing, include/vk_processing.h:26-356 of the reference;
 * dispatch script src/vk_processing.cpp:2104-2601; shaders/X.comp).
 *
 * The reference has no extern "C" surface: its harness crosses a process
 * boundary (`maverick-standalone <raw> <bmp>`, test/standalone/main.cpp:30-87)
 * that makes four C++ calls on VulkanProcessing. Each entry point below names
 * the C++ member it replaces. `bool` becomes `int` (1 = ok, 0 = failed) so the
 * reference's truthiness convention (ASSERT_MSG(call, msg)) carries over; on
 * failure a message "MUSICA ERROR: ..." is written to stderr, mirroring
 * `fprintf(stderr, "VK STATE ERROR: %s\n")` (src/vk_processing.cpp:14-18), and
 * is retrievable with musica_last_error().
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All images are square (the reference never handles W != H,
 * src/vk_processing.cpp:2630-2631), single channel, row-major, dense when they
 * cross the ABI (the library keeps its own pitched layout in HBM).
 */
#ifndef MUSICA_H
#define MUSICA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUSICA_ABI_VERSION 3

/* Hard-coded constants of the reference. */
#define MUSICA_MAX_LEVELS 16          /* reference: 12 + 1 clear buffers, vk_processing.h:67 */
#define MUSICA_MIN_LEVELS 4           /* cnrLevel = coarserLevelsStart = 3 must exist, vk_processing.h:28-29 */
#define MUSICA_NOISE_BINS 2048        /* vk_processing.h:37, noise_hist.comp:6 */
#define MUSICA_GRAD_BINS 1024         /* vk_processing.h:38, gradation_histogram.comp:6 */
#define MUSICA_MAX_POINTS 256         /* contrast_curve_generate.comp:5 */
#define MUSICA_CNR_LEVEL 3            /* vk_processing.h:29 */
#define MUSICA_COARSER_LEVELS_START 3 /* vk_processing.h:28 */
#define MUSICA_OUT_MARGIN 10          /* src/vk_processing.cpp:2607 */
#define MUSICA_CLAHE_TILES 4          /* clahe_histogram.comp:4 */
#define MUSICA_CLAHE_BINS 256         /* clahe_grad_curve.comp:4 */

/* T1: struct HistogramMaxPoint, vk_processing.h:123-126. */
typedef struct musica_hist_max_point {
    uint32_t maxValue;
    uint32_t maxBin;
} musica_hist_max_point;

/* T2: struct ContrastParameters, vk_processing.h:135-138. */
typedef struct musica_contrast_params {
    float lowContrastFactor;
    float highContrastFactor;
} musica_contrast_params;

typedef struct musica_point {
    float x;
    float y;
} musica_point;

/* T3: struct ContrastCurveObj, vk_processing.h:141-149 (2052 bytes). */
typedef struct musica_contrast_curve {
    musica_point points[MUSICA_MAX_POINTS];
    uint32_t pointsCount;
} musica_contrast_curve;

/* T4: struct NoiseReductionParams, vk_processing.h:112-117. */
typedef struct musica_nr_params {
    float lowCnr;
    float lowFactor;
    float highCnr;
    float highFactor;
} musica_nr_params;

/* T5: struct GradCurveObj, vk_processing.h:215-222 (2064 bytes). */
typedef struct musica_grad_curve {
    musica_point points[MUSICA_MAX_POINTS];
    uint32_t pointsCount;
    float t0;
    float ta;
    float t1;
} musica_grad_curve;

/* Per-image summary gathered across GPUs by the batch driver. */
typedef struct musica_stats {
    uint32_t image_id;        /* index inside the batch handed to execute */
    float min_sqrt;           /* final texel of the min chain (min_reduce.comp) */
    float max_sqrt;           /* final texel of the max chain (img_max_reduce.comp) */
    uint32_t noise_max_bin[4];   /* HistogramMaxPoint.maxBin of levels 0..3 */
    uint32_t noise_max_value[4]; /* HistogramMaxPoint.maxValue of levels 0..3 */
    uint32_t grad_max_bin;
    uint32_t grad_max_value;
    float mean_cnr;           /* mean(cnr image) * 256: what test/mean_cnr/script.py:13-24 prints */
    float t0, ta, t1;         /* gradation window, gradation_curve_generate.comp:54-144 */
} musica_stats;

/* Flags for musica_params.flags */
#define MUSICA_FLAG_CLAHE      0x1u /* CLAHE gradation (reference: #ifdef ENABLE_CLAHE, vk_processing.h:13) */
#define MUSICA_FLAG_NO_GRAPH   0x2u /* launch kernels eagerly instead of replaying a captured hipGraph */
#define MUSICA_FLAG_GENERIC_KERNELS 0x4u /* test hook: use the one-thread-per-texel kernels at every level */
#define MUSICA_FLAG_LINEAR     0x10u /* one in-order stream in the reference's submission order (src/vk_processing.cpp:2104-2601), replayed as a
                                        graph, whatever the workload: for contexts that run beside other contexts on one GPU (each then owns
                                        one stream = one hardware queue; musica_pipeline_* alternates steps over three of them) */
#define MUSICA_FLAG_NO_AUTOTUNE 0x8u /* skip the init-time launch-geometry autotune (rows per wavefront stay heuristic) */
#define MUSICA_FLAG_ONE_SHOT   0x40u /* the context will execute once or a few times (musica-standalone: one image per process): no autotune, no graph
                                        capture, ONE stream whatever the workload (creating a second stream costs more than one step saves).
                                        Implies NO_AUTOTUNE and NO_GRAPH; those two flags on their own do NOT change the number of streams. */
#define MUSICA_FLAG_REFERENCE_ORDER 0x20u /* the shaders' literal arithmetic order: img_smooth.comp:32-45, img_smooth_upsampled.comp:32-45
                                        (with the * 4.0 per tap) and img_sdev.comp:17-30 accumulate their 25 taps m (x) outer,
                                        n (y) inner, starting from 0. One thread per texel (as the shaders run), several times
                                        slower than the default separable order; results are bit-identical to the oracle's
                                        MUSICA_ORDER_REFERENCE. The default order is tolerance-close to this one. */

/* Construction parameters: the reference hard-wires these as literals
 * (imageSize = 3072 in test/standalone/main.cpp:31; L = ceil(log2 N) in
 * src/vk_processing.cpp:1989). */
typedef struct musica_params {
    uint32_t image_size; /* N; 16 <= N <= 16384 (a level-0 f32 plane must stay below 2 GiB: 32-bit buffer offsets) */
    uint32_t levels;     /* L; 0 => ceil(log2 N) (reference rule); else 4 <= L <= ceil(log2 N) */
    uint32_t batch;      /* images per execute call, 0 => 1 (reference: 1) */
    int32_t device;      /* HIP device ordinal */
    uint32_t flags;      /* MUSICA_FLAG_* */
} musica_params;

/* The reference's compile-time configuration of the contrast and noise-reduction parameter formulas as runtime values (ABI version 3).
 * Replaces: the private constants of include/vk_processing.h:39-49 (nrHighCnr, nrMaxHighFactor, nrLowCnr, nrMinLowFactor,
 * highContrastMaxReduction, lowContrastMaxEnhancment) and the two #defines of include/vk_processing.h:16-17
 * (LINEAR_LOW_CONTRAST_LEVELS_REDUCTION, LINEAR_HIGH_CONTRAST_LEVELS_REDUCTION) that select the linear instead of the power form of
 * src/vk_processing.cpp:262-293. musica_tunables_default() fills in the reference's values (both #defines are commented out there).
 * coarserLevelsStart = 3 and cnrLevel = 3 (vk_processing.h:28-29) stay compile-time constants here as well
 * (MUSICA_COARSER_LEVELS_START, MUSICA_CNR_LEVEL): they decide which images and launches exist, not a formula. */
typedef struct musica_tunables {
    float nr_high_cnr;                  /* nrHighCnr = 9.0f */
    float nr_max_high_factor;           /* nrMaxHighFactor = 1.2f */
    float nr_low_cnr;                   /* nrLowCnr = 3.0f */
    float nr_min_low_factor;            /* nrMinLowFactor = 0.6f */
    float high_contrast_max_reduction;  /* highContrastMaxReduction = 0.2f */
    float low_contrast_max_enhancement; /* lowContrastMaxEnhancment = 3.0f */
    uint32_t linear_low_contrast;       /* != 0: #define LINEAR_LOW_CONTRAST_LEVELS_REDUCTION  (src/vk_processing.cpp:282-287) */
    uint32_t linear_high_contrast;      /* != 0: #define LINEAR_HIGH_CONTRAST_LEVELS_REDUCTION (src/vk_processing.cpp:262-268) */
} musica_tunables;

typedef struct musica_ctx musica_ctx;

/* Image kinds addressable by musica_get_image / musica_debug_set_image.
 * `level` is the pyramid level whose grid the image lives on. */
typedef enum musica_image_kind {
    MUSICA_IMG_NORMALIZED = 0,   /* normalizedImageState, N x N (level must be 0) */
    MUSICA_IMG_DOWNSAMPLED = 1,  /* downsampledImageStates[level], side S_{level+1} */
    MUSICA_IMG_BANDPASS = 2,     /* bandpassImageStates[level], side S_level */
    MUSICA_IMG_SDEV = 3,         /* sdevImageStates[level] (levels 0..3), side S_level */
    MUSICA_IMG_CNR = 4,          /* cnrImageState, side S_3 (level must be 3) */
    MUSICA_IMG_EXPAND = 5,       /* expandImageStates[L-1-level]: reconstruction at `level`, side S_level */
    MUSICA_IMG_GRADED = 6,       /* gradedImageState, N x N (level must be 0) */
    MUSICA_IMG_RELEVANT = 7,     /* relevantImageState, N x N; recomputed on demand (not stored on the hot path) */
    MUSICA_IMG_LOWPASS = 8,      /* lowpassImageStates[level]; recomputed on demand */
    MUSICA_IMG_EXP_BANDPASS = 9, /* band after contrast curve (+ noise reduction on levels 0,1) as fed to img_addition; recomputed on demand */
    MUSICA_IMG_SQRT = 10,        /* sqrtImageState, N x N; recomputed on demand */
    MUSICA_IMG_CLAHE_GRADED = 11, /* claheGradedImageState, N x N (only with MUSICA_FLAG_CLAHE; src/vk_processing.cpp:432-438) */
    MUSICA_IMG_CONTRAST_BAND = 12, /* expandBandpassImageStates[L-1-level]: band after the contrast curve, BEFORE noise reduction (what
                                    * debugProcess dumps as exp_bandpass_i, src/vk_processing.cpp:2710-2718); recomputed on demand */
    MUSICA_IMG_KIND_COUNT = 13
} musica_image_kind;

/* Pipeline stages runnable one at a time through musica_debug_run_stage
 * (kernel-level parity tests inject an input with musica_debug_set_image and
 * run exactly one stage). Names follow the per-stage timing line of the
 * reference (src/vk_processing.cpp:2585-2595). */
typedef enum musica_stage {
    MUSICA_STAGE_NORM = 0,   /* sqrt, min/max chains, normalize        (.cpp:2182-2222) */
    MUSICA_STAGE_REDUCE = 1, /* smooth, downsample, upsample, smooth_upsampled, difference (.cpp:2233-2273) */
    MUSICA_STAGE_ANALYSIS = 2, /* sdev, noise_hist, hist_max, contrast_curve_generate, cnr (.cpp:2284-2357) */
    MUSICA_STAGE_EXPAND = 3, /* contrast apply, noise reduction, upsample, smooth_upsampled, addition (.cpp:2361-2431) */
    MUSICA_STAGE_GRADATION = 4, /* relevant, grad hist, hist max, curve generate, curve apply (.cpp:2456-2518) */
    MUSICA_STAGE_COUNT = 5
} musica_stage;

/* ---- lifecycle ------------------------------------------------------- */

/* Replaces: VulkanProcessing::VulkanProcessing(VulkanState*) + bool init(uint32_t imageSize,
 * std::vector<VkImageView>*) (vk_processing.h:281-288, src/vk_processing.cpp:1984-2020).
 * Allocates every device buffer once; returns NULL on failure. */
musica_ctx* musica_create(const musica_params* params);
/* The same with the parameter formulas' constants given by the caller (NULL: the reference's). Values the formulas cannot use
 * (nr_high_cnr == nr_low_cnr, non-finite numbers) are refused. */
musica_ctx* musica_create_ex(const musica_params* params, const musica_tunables* tunables);
void musica_tunables_default(musica_tunables* out);
/* the tunables a context was created with */
int musica_get_tunables(const musica_ctx* ctx, musica_tunables* out);

/* Replaces: bool VulkanProcessing::cleanup() (src/vk_processing.cpp:2647-2651). Frees everything. */
void musica_destroy(musica_ctx* ctx);

/* Replaces: uint32_t getImageSize() (vk_processing.h:355). */
uint32_t musica_get_image_size(const musica_ctx* ctx);
uint32_t musica_get_levels(const musica_ctx* ctx);
uint32_t musica_get_batch(const musica_ctx* ctx);
/* Side of pyramid level `level` (S_0 = N, S_{i+1} = ceil(S_i / 2); level may be L for the residual). */
/* 1 when the level-0 expand launch of this context also accumulates the gradation histogram (no separate pass over the
 * reconstructed image; bench.py prices the launch accordingly), else 0. */
int musica_fuses_gradation_histogram(const musica_ctx* ctx);
/* 1 when level 0's smooth + downsample and band-pass image come out of one launch (k_reduce_band_u16; the `reduce_l0` profile
 * family then covers both and `band_l0` stays empty), else 0. */
int musica_fuses_reduce_band(const musica_ctx* ctx);
/* 1 when the expand launches of levels 0 .. 2 compute the 5 x 5 RMS of their band image themselves and the sdev + noise-histogram
 * launches of those levels store no image (whole-step execution only: getters, dumps and the stage entry points produce the stored
 * images on demand, bit-identical); chosen per workload by musica_create, MUSICA_SDEV_IN_EXPAND=0|1 overrides. bench.py prices
 * the launches accordingly. */
int musica_fuses_sdev(const musica_ctx* ctx);
uint32_t musica_get_level_size(const musica_ctx* ctx, uint32_t level);
/* How this context dispatches a step (chosen by musica_create from the batch, the image side, the depth of the pyramid and the
 * flags): *streams = 1 (the reference's one in-order queue), 2 (the analysis launches on a second stream
 * beside the reduce tail: the default for everything but small steps; MUSICA_STREAMS=1|2 overrides); *graph = 1 when steps replay a
 * captured hipGraph, 0 for eager launches. Either pointer may be NULL. Returns 1, 0 for a NULL context. */
int musica_get_dispatch(const musica_ctx* ctx, int* streams, int* graph);

/* ---- the hot path ---------------------------------------------------- */

/* Replaces: bool VulkanProcessing::execute(const uint16_t* imageData) (src/vk_processing.cpp:2104-2601).
 * `pixels`: batch * N * N host uint16, row-major, borrowed for the call
 * (H2D copy inside, as vk_state.cpp:313-342 does). Synchronous: returns after
 * the device finished, like the final vkWaitForFences (.cpp:2535-2536). */
int musica_execute(musica_ctx* ctx, const uint16_t* pixels);

/* Same pipeline with the input already resident in HBM: `d_pixels` is a device
 * pointer to batch * N * N uint16 (dense). Enqueues on the ctx stream and
 * returns without waiting; pair with musica_sync(). The dispatch script is
 * captured into a hipGraph once per distinct `d_pixels` and replayed afterwards;
 * a context keeps the graphs of the 4 most recently used pointers (rotating more
 * than 4 buffers through one context recaptures — a host-side stall of about a
 * millisecond plus a stream drain — every time a pointer comes back). */
int musica_execute_device(musica_ctx* ctx, const uint16_t* d_pixels);

/* Uploads host pixels into the ctx-owned resident input buffer and returns its
 * device pointer (for musica_execute_device). */
int musica_upload(musica_ctx* ctx, const uint16_t* pixels);
const uint16_t* musica_input_device_ptr(musica_ctx* ctx);

/* Blocks until everything enqueued on the ctx stream has finished. */
int musica_sync(musica_ctx* ctx);

/* ---- results --------------------------------------------------------- */

/* D2H of gradedImageState (f32, batch * N * N, dense) — the image
 * saveOutImage reads back (src/vk_processing.cpp:2609-2621). */
int musica_get_graded(musica_ctx* ctx, float* dst);

/* Replaces: bool VulkanProcessing::saveOutImage(std::string) (src/vk_processing.cpp:2603-2645):
 * crop MUSICA_OUT_MARGIN on every side, (uint8_t)(255.0f * v), 24-bpp bottom-up
 * BMP byte-identical to stbi_write_bmp(comp = 1) (stb_image_write.h:492-500).
 * `image_index` selects the image of the batch (reference: 0). */
int musica_save_out_image(musica_ctx* ctx, uint32_t image_index, const char* path);

/* The 8-bit cropped pixels saveOutImage would write, side N - 20, top-down rows. */
int musica_get_out_pixels(musica_ctx* ctx, uint32_t image_index, uint8_t* dst);

/* One dense f32 image of one batch entry; side = musica_image_side(kind, level). */
int musica_get_image(musica_ctx* ctx, uint32_t image_index, musica_image_kind kind, uint32_t level, float* dst);
uint32_t musica_image_side(const musica_ctx* ctx, musica_image_kind kind, uint32_t level);

/* Integer / small-struct state (bit-exact parity objects). */
int musica_get_noise_hist(musica_ctx* ctx, uint32_t image_index, uint32_t level, uint32_t* dst /*2048*/);
int musica_get_grad_hist(musica_ctx* ctx, uint32_t image_index, uint32_t* dst /*1024*/);
int musica_get_noise_hist_max(musica_ctx* ctx, uint32_t image_index, uint32_t level, musica_hist_max_point* dst);
int musica_get_grad_hist_max(musica_ctx* ctx, uint32_t image_index, musica_hist_max_point* dst);
int musica_get_contrast_curve(musica_ctx* ctx, uint32_t image_index, uint32_t level, musica_contrast_curve* dst);
int musica_get_grad_curve(musica_ctx* ctx, uint32_t image_index, musica_grad_curve* dst);
int musica_get_contrast_params(musica_ctx* ctx, uint32_t level, musica_contrast_params* dst);
int musica_get_nr_params(musica_ctx* ctx, uint32_t level /*0..2*/, musica_nr_params* dst);
int musica_get_minmax(musica_ctx* ctx, uint32_t image_index, float* min_sqrt, float* max_sqrt);
int musica_get_stats(musica_ctx* ctx, uint32_t image_index, musica_stats* dst);
/* Writes the musica_stats of every image of the batch (batch * sizeof(musica_stats) bytes,
 * image_id = image_id_base + index) into caller-owned DEVICE memory, asynchronously on the ctx
 * stream: the buffer the multi-GPU batch driver hands to its RCCL all-gather. */
int musica_stats_device(musica_ctx* ctx, void* d_dst, uint32_t image_id_base);
/* The same with image_id = image_id_base + index * image_id_stride: a rank of the batch driver owns the images
 * rank, rank + world, rank + 2 * world, ..., so its rows carry their job-wide ids without a second kernel. */
int musica_stats_device_strided(musica_ctx* ctx, void* d_dst, uint32_t image_id_base, uint32_t image_id_stride);
/* CLAHE state (only with MUSICA_FLAG_CLAHE): 4*4*256 u32 histograms [tx][ty][bin], 4*4*256 curve points. */
int musica_get_clahe_hist(musica_ctx* ctx, uint32_t image_index, uint32_t* dst);
int musica_get_clahe_curves(musica_ctx* ctx, uint32_t image_index, musica_point* dst);

/* Replaces: bool VulkanProcessing::debugProcess() (src/vk_processing.cpp:2661-2809): writes
 * norm.bmp, red_bandpass_i.bmp, red_lowpass_i.bmp, sdev.bmp, cnr.bmp,
 * exp_bandpass_i.bmp, exp_lowpass_i.bmp, relevant.bmp, graded.bmp (same
 * quantisation as VulkanState::downloadAndSaveImage, src/vk_state.cpp:809-855)
 * into `dir`, the two RGBA plots noise_hist.bmp / grad_hist.bmp, and (an addition)
 * noise_hist.csv / grad_hist.csv / grad_curve.csv with the numbers behind them. */
int musica_debug_process(musica_ctx* ctx, uint32_t image_index, const char* dir);
/* The two RGBA plots the reference renders on every execute (#define RENDER_HISTS, include/vk_processing.h:22) and
 * debugProcess writes as noise_hist.bmp / grad_hist.bmp (src/vk_processing.cpp:2758-2806): noise_hist_render.comp on the
 * histogram of cnrLevel (:1260-1266) and gradation_curve_debug_render.comp on the gradation histogram + tone curve
 * (:1668-1675), each one workgroup of 512 invocations on a histRenderWidth x histRenderHeight rgba8 image
 * (include/vk_processing.h:31-32). `rgba`: MUSICA_HIST_RENDER_WIDTH * MUSICA_HIST_RENDER_HEIGHT * 4 bytes, top row first. */
#define MUSICA_HIST_RENDER_WIDTH 512
#define MUSICA_HIST_RENDER_HEIGHT 128
int musica_render_noise_hist(musica_ctx* ctx, uint32_t image_index, uint8_t* rgba);
int musica_render_grad_hist(musica_ctx* ctx, uint32_t image_index, uint8_t* rgba);

/* ---- steps in flight (new, not in the reference) ----------------------- */

/* The reference has one VulkanProcessing and one frame in flight (vkWaitForFences at the end of execute,
 * src/vk_processing.cpp:2535-2536). A pipeline is `depth` contexts of one GPU, each created from `params` with
 * MUSICA_FLAG_LINEAR, whose steps alternate: step s is enqueued (asynchronously) on context s mod depth, so the chip-filling
 * kernels of a step run in the part-idle phases of the steps beside it (three contexts: +23 % throughput at 8 x 2048^2,
 * 2.3 x for one image per step). musica_pipeline_prime() captures every context's graph and chooses WHICH hardware queues
 * stay: it creates one context per queue of the runtime (MUSICA_PIPELINE_QUEUES), times every cyclic window of `depth` of
 * them for `calibration_steps` steps (0: three per context) and destroys the contexts outside the fastest window — two of
 * the four queues of an MI355X do not run side by side. Order of calls: create, upload (or fill every context's input
 * through musica_pipeline_context), prime, then step / sync at will. A context runs its own steps in order: the results of
 * the step a context ran last are valid after musica_sync of that context (or musica_pipeline_sync) and before its next step. */
typedef struct musica_pipeline musica_pipeline;
#define MUSICA_PIPELINE_QUEUES 4
musica_pipeline* musica_pipeline_create(const musica_params* params, uint32_t depth);
musica_pipeline* musica_pipeline_create_ex(const musica_params* params, uint32_t depth, const musica_tunables* tunables);   /* NULL: the reference's constants */
void musica_pipeline_destroy(musica_pipeline* p);
uint32_t musica_pipeline_depth(const musica_pipeline* p);
/* k-th context: before prime() k < max(depth, MUSICA_PIPELINE_QUEUES) (depth 1: one), afterwards k < depth, in step order. */
musica_ctx* musica_pipeline_context(musica_pipeline* p, uint32_t k);
/* The same batch (batch x N x N uint16, host memory) into the input buffer of every context. */
int musica_pipeline_upload(musica_pipeline* p, const uint16_t* pixels);
int musica_pipeline_prime(musica_pipeline* p, uint32_t calibration_steps);
/* ms per step of the windows prime() timed (window k starts at the k-th created context); returns how many (0: none). */
uint32_t musica_pipeline_calibration(const musica_pipeline* p, float* window_ms /* [MUSICA_PIPELINE_QUEUES] or NULL */);
/* Enqueue one step on the next context: d_pixels (device memory, 16-byte aligned) or, when NULL, that context's own input. */
int musica_pipeline_step(musica_pipeline* p, const uint16_t* d_pixels);
/* The context the most recent step was enqueued on. */
musica_ctx* musica_pipeline_last(musica_pipeline* p);
int musica_pipeline_sync(musica_pipeline* p);

/* ---- test / profiling hooks ------------------------------------------ */

/* A sequence of `count` batches (pixels[j]: batch x N x N uint16 in host memory), pipelined: two device input buffers and a
 * copy stream, so the host-to-device copy of batch j + 1 runs under the kernels of batch j — what replaces the reference's
 * staging-buffer upload with three queue-idle waits per image (VulkanState::loadDataToImage, src/vk_state.cpp:313-342).
 * Returns after the last batch has been computed; the context then holds the results of batch count - 1 (every getter works).
 * stats (may be NULL): count * batch rows, row j * batch + i = image i of batch j, image_id = its index in the sequence.
 * Inputs in pinned memory (musica_host_alloc) are copied at the PCIe rate; pageable memory is accepted (slower). */
int musica_execute_stream(musica_ctx* ctx, const uint16_t* const* pixels, uint32_t count, musica_stats* stats);
/* Page-locked host memory for musica_execute_stream / musica_execute inputs (hipHostMalloc / hipHostFree). */
void* musica_host_alloc(musica_ctx* ctx, size_t bytes);
void musica_host_free(musica_ctx* ctx, void* ptr);

/* Overwrites one stored intermediate of one batch entry (dense f32 in). Only
 * kinds the hot path keeps resident are accepted. */
int musica_debug_set_image(musica_ctx* ctx, uint32_t image_index, musica_image_kind kind, uint32_t level, const float* src);
/* Runs exactly one stage of the dispatch script on the current device state. */
int musica_debug_run_stage(musica_ctx* ctx, musica_stage stage);

/* Per-kernel device timing with HIP events on the ctx stream.
 * musica_profile_enable(ctx, mask): mask < 0 brackets every kernel family of every
 * later execute with an event pair, mask > 0 only the families whose bit
 * (1 << musica_kernel_id) is set, 0 switches it off. musica_profile_get returns
 * the mean duration in microseconds and the launch count since the last reset. */
typedef enum musica_kernel_id {
    MUSICA_KERNEL_MINMAX = 0,
    MUSICA_KERNEL_NORMALIZE = 1,
    MUSICA_KERNEL_REDUCE_L0 = 2,   /* fused 5-tap smooth + 2x downsample at level 0 (the metric kernel) */
    MUSICA_KERNEL_REDUCE_REST = 3, /* same kernel, levels >= 1 */
    MUSICA_KERNEL_BAND_L0 = 4,     /* fused upsample + smooth x4 + difference at level 0 */
    MUSICA_KERNEL_BAND_REST = 5,
    MUSICA_KERNEL_SDEV_HIST = 6,
    MUSICA_KERNEL_CURVES = 7,
    MUSICA_KERNEL_CNR = 8,
    MUSICA_KERNEL_EXPAND_L0 = 9,   /* fused contrast apply + NR + upsample + smooth x4 + addition at level 0 */
    MUSICA_KERNEL_EXPAND_REST = 10,
    MUSICA_KERNEL_GRAD_HIST = 11,
    MUSICA_KERNEL_GRAD_CURVE = 12,
    MUSICA_KERNEL_GRAD_APPLY = 13,
    MUSICA_KERNEL_COUNT = 14
} musica_kernel_id;
int musica_profile_enable(musica_ctx* ctx, int mask);
int musica_profile_reset(musica_ctx* ctx);
int musica_profile_get(musica_ctx* ctx, musica_kernel_id id, double* mean_us, uint64_t* launches);

/* Stand-alone launch of the metric kernel (fused smooth + downsample) on a
 * side x side f32 image owned by the caller in device memory (pitch in floats,
 * multiple of 4). Used by bench.py to time the kernel at 4096 x 4096 and by the
 * kernel-level parity tests. d_out has side ceil(side/2), pitch out_pitch. */
int musica_k_reduce(musica_ctx* ctx, const float* d_in, uint32_t side, uint32_t in_pitch,
                    float* d_out, uint32_t out_pitch, uint32_t batch);
/* Times `iters` back-to-back launches of the above with HIP events on the ctx
 * stream; returns mean microseconds per launch in *mean_us. */
int musica_k_reduce_timed(musica_ctx* ctx, const float* d_in, uint32_t side, uint32_t in_pitch,
                          float* d_out, uint32_t out_pitch, uint32_t batch, uint32_t iters, double* mean_us);

/* The same measurement from HBM rather than from the 256 MiB Infinity Cache: launch i uses input plane
 * d_in + (i % nbuf) * in_pitch * side and output plane d_out + (i % nbuf) * out_pitch * ceil(side/2), so with
 * nbuf * 5 * side^2 bytes well above 256 MiB no launch finds its input (or the lines of its output) on the die. */
int musica_k_reduce_timed_rot(musica_ctx* ctx, const float* d_in, uint32_t side, uint32_t in_pitch, float* d_out,
                              uint32_t out_pitch, uint32_t nbuf, uint32_t iters, double* mean_us);
/* Measurement aid, not on the product path: a plain streaming kernel with the metric kernel's traffic shape
 * (reads side^2 f32 with 16-byte loads, writes (side/2)^2 f32 with 16-byte stores, no halo, no arithmetic to
 * speak of), timed the same rotating way — the ceiling `roofline.frac` can be read against. side % 8 == 0,
 * dense rows. */
int musica_k_copy41_timed_rot(musica_ctx* ctx, const float* d_in, uint32_t side, float* d_out, uint32_t nbuf,
                              uint32_t iters, double* mean_us);

/* Self-test of the exact arithmetic shortcuts that lean on a hardware approximation (v_rsq_f32) and therefore
 * cannot be checked on a CPU (csrc/exact_math.h): runs on the ctx device over EVERY float bit pattern and counts
 * disagreements with the literal expressions of the shaders (img_sqrt.comp:15, img_sdev.comp:30,
 * img_normalize.comp:24). mismatches[0]: musica_sqrt vs sqrtf, 2^32 patterns; [1]: the 8-wide grouped form;
 * [2]: the grouped form with a +0 among the eight; [3]: the normalisation of a raw pixel for every
 * (pixel, min, max) triple the chains can produce (65536 x 256 x 256). All four must be 0. */
int musica_selftest_exact_math(musica_ctx* ctx, uint64_t mismatches[4]);

/* Raw device memory helpers so callers without a HIP binding (ctypes tests,
 * bench.py) can stage buffers for the two functions above. */
void* musica_device_alloc(musica_ctx* ctx, size_t bytes);
void musica_device_free(musica_ctx* ctx, void* d_ptr);
int musica_memcpy_h2d(musica_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int musica_memcpy_d2h(musica_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- file formats (host only, no GPU needed) -------------------------- */

/* Raw reader of test/standalone/main.cpp:54-75: file = 256-byte header
 * (ignored) + N*N little-endian uint16, size must match exactly. */
int musica_read_raw(const char* path, uint32_t image_size, uint16_t* dst);
/* 24-bpp BMP writer byte-identical to stbi_write_bmp(path, w, h, 1, data). */
int musica_write_bmp_gray(const char* path, uint32_t w, uint32_t h, const uint8_t* data);
/* 32-bpp BMP writer byte-identical to stbi_write_bmp(path, w, h, 4, data) (V4 header, stb_image_write.h:501-509). */
int musica_write_bmp_rgba(const char* path, uint32_t w, uint32_t h, const uint8_t* data);

/* ---- misc ------------------------------------------------------------ */
const char* musica_last_error(void);
uint32_t musica_abi_version(void);
/* Number of HIP devices visible; 0 when there is none (never falls back to a CPU path). */
int musica_device_count(void);

#ifdef __cplusplus
}
#endif

#endif /* MUSICA_H */
