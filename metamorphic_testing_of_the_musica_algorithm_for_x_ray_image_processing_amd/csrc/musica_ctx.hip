// musica_ctx.hip — host side of libmusica_hip.so: the context that replaces class VulkanProcessing
// (include/vk_processing.h:26-356 of the reference), the dispatch script that replaces
// VulkanProcessing::execute (src/vk_processing.cpp:2104-2601) and the extern "C" boundary of
// include/musica.h. No CPU fallback exists: without a HIP device musica_create fails loudly.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <exception>
#include <string>
#include <vector>

#include "launchers.h"

// host-only helpers implemented in musica_io.cpp
extern "C" int musica_write_bmp_gray(const char* path, uint32_t w, uint32_t h, const uint8_t* data);
extern "C" uint32_t musica_bmp24_header(uint32_t w, uint32_t h, uint8_t hdr[54]);
extern "C" int musica_write_file(const char* path, const uint8_t* bytes, size_t count);

using namespace musica;

// ---- errors ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    fprintf(stderr, "MUSICA ERROR: %s\n", buf);  // cf. "VK STATE ERROR: %s" src/vk_processing.cpp:14-18
    return 0;
}

// No C++ exception may cross the extern "C" boundary (a ctypes / CLI caller would abort): entry points that allocate
// host memory run their body under this guard and report through fail() like every other error.
#define ABI_TRY try {
#define ABI_CATCH(name_)                                                                             \
    }                                                                                                \
    catch (const std::exception& e_) { return fail("%s: %s", name_, e_.what()); }                   \
    catch (...) { return fail("%s: unknown C++ exception", name_); }

#define HIP_OK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fail("%s failed: %s", #call, hipGetErrorString(e_));  \
    } while (0)

// ---- context --------------------------------------------------------------------------
struct ProfSpan {
    int id;
    hipEvent_t a, b;
};

// Captured graphs kept per context, one per distinct input pointer (include/musica.h, musica_execute_device).
constexpr int kGraphSlots = 4;

struct musica_ctx {
    musica_params p;
    musica_tunables tun;     // the constants of the host parameter formulas (musica_create_ex; default: the reference's)
    int N, L, B;
    bool generic;
    int ref_order;           // MUSICA_FLAG_REFERENCE_ORDER: generic kernels in the shaders' literal 25-tap accumulation order
    bool tuning;  // inside autotune(): launches are tagged so profilers keep them apart
    LevelDesc lv[MUSICA_MAX_LEVELS + 1];
    int min_chain_exact;
    int hist_cov;  // (N / 512) * 512
    hipStream_t stream;
    hipStream_t cur;         // stream the run_*_level helpers launch on (stream or side)
    hipStream_t side;        // coarse-level chain runs here, concurrently with the level-0 kernels on `stream`
    hipEvent_t ev_fork, ev_join;
    bool fuse_u16;           // level-0 kernels read the raw uint16 pixels; the normalized image is produced on demand only
    bool grad_one_launch;    // recount + tone curve in one launch behind the fused expand launch (MUSICA_GRAD_ONE_LAUNCH=0: two)
    uint32_t* d_gr_ticket;   // its tickets: [B][kGradTicketStride]
    bool tiny_tail;          // levels of side <= kTailSide in one launch (MUSICA_TINY_TAIL=0: one launch per level and stage)
    bool clahe_one_apply;    // ... and whose two apply passes are one launch (MUSICA_CLAHE_ONE_APPLY=0: k_grad_apply and k_clahe_apply4)
    bool clahe_in_expand;    // ... and whose histogram the level-0 expand launch counts (MUSICA_CLAHE_IN_EXPAND=0: k_clahe_hist)
    bool clahe_raw;          // CLAHE context whose relevant image is computed from the raw pixels (no stored normalized image)
    bool norm_valid;         // d_norm holds the normalized image of the current input
    // hipGraph replay of the two-stream dispatch (captured once per input pointer; MUSICA_FLAG_NO_GRAPH /
    // MUSICA_GRAPH=0 / per-kernel profiling fall back to eager launches)
    bool use_graph;
    hipGraphExec_t graph_exec[kGraphSlots];      // one captured graph per input pointer, the kGraphSlots most recently used (the streaming
    const uint16_t* graph_input[kGraphSlots];    // path alternates between two device input buffers; callers may rotate a few of their own)
    uint64_t graph_used[kGraphSlots];            // launch counter at the slot's last use (least recently used slot is recaptured)
    uint64_t graph_clock;
    int dag;                 // 0: one in-order stream (enqueue_linear); 2: two streams (enqueue_fork: the analysis beside the reduce tail)
    // device state
    uint16_t* d_input;
    uint16_t* d_input2;      // second input buffer of the streaming path (musica_execute_stream), allocated on first use
    hipStream_t copy_stream; // its H2D copies run here, under the previous batch's kernels
    hipEvent_t ev_copied[2], ev_consumed[2];
    const uint16_t* cur_input;
    uint32_t* d_minmax;
    uint32_t* d_mm_slots;      // [B][kMinMaxSlots]: per-block {min | max << 16} of k_minmax_u16
    uint32_t* d_mm_ticket;     // [B]: its arrival counters (self-resetting)
    float* d_norm;
    float* d_down[MUSICA_MAX_LEVELS];
    float* d_band[MUSICA_MAX_LEVELS];
    float* d_recon[MUSICA_MAX_LEVELS];
    float* d_sdev[4];
    uint32_t* d_noise_hist;
    musica_hist_max_point* d_noise_max;
    DevCurve* d_curves;
    DevCurveLut* d_luts;
    musica_contrast_params* d_cparams;
    float* d_cnr;
    uint32_t* d_grad_hist;
    uint32_t* d_grad_hist_b;   // the literal recount of images whose reconstruction holds an exact zero (fused gradation histogram)
    uint32_t* d_gzero;         // [B]: that condition
    int* d_thr090;             // [B]: raw-pixel form of `normalized <= 0.9`
    uint32_t* d_plot;          // one MUSICA_HIST_RENDER_WIDTH x MUSICA_HIST_RENDER_HEIGHT rgba8 image (the RENDER_HISTS plots, on demand)
    double* d_stats_partial;   // [B][kStatsMaxBlocks]: partial sums of the cnr image (k_stats_partial -> k_stats)
    uint16_t* d_le090;         // [B][S1][S0 / 8] or null: its bit image, written by the level-0 reduce + band launch for the level-0 expand launch
    bool fuse_gh;              // the level-0 expand launch accumulates the gradation histogram
    // The expand launches of levels 0 .. 2 compute the 5 x 5 RMS of their band image themselves (k_expand_fast<.., SD>) and the sdev +
    // noise-histogram launches of those levels store nothing: 8 of a step's 48 bytes per input pixel. The whole-step scripts run that way
    // (sd_active); the stage entry points, getters and dumps want the stored images: ensure_sdev() writes them on demand.
    bool sd_fused, sd_active, sdev_stored;
    bool pair_rb_sdev;         // the one-stream script pairs the sdev pass of level i with reduce + band of level i + 1 in one launch (k_rb_sdev); MUSICA_PAIR_RB_SDEV
    bool sdev_one_launch;      // the sdev + noise-histogram passes of levels 0 .. 3 as ONE launch (k_sdev_hist_levels); MUSICA_SDEV_ONE_LAUNCH=0: one launch per marching level + one for the runs
    int rows_rb[MUSICA_MAX_LEVELS];   // its coarse rows per wavefront
    musica_hist_max_point* d_grad_max;
    DevCurve* d_gcurve;
    float* d_graded;
    float* d_scratch;
    musica_stats* d_stats;
    uint32_t* d_clahe_hist;
    musica_point* d_clahe_pts;
    float* d_clahe_graded;
    uint8_t* d_out8;           // saveOutImage's cropped 8-bit pixels of one image (device) and their pinned host copy, allocated on first use
    uint8_t* h_out8;
    uint8_t* h_bmp;            // saveOutImage's whole file image in page-locked memory: 2 bytes of padding, the 54-byte header, then the pixel array the
                               // device writes itself (k_out_bmp24: the array starts on a 4-byte boundary); allocated on first use
    // host parameters (src/vk_processing.cpp:259-297, 321-325)
    musica_contrast_params h_cparams[MUSICA_MAX_LEVELS];
    musica_nr_params h_nr[3];
    // rows each wavefront marches per launch, per level (heuristic, then autotuned at create)
    int rows_expand[MUSICA_MAX_LEVELS], rows_sdev[4];
    // tunables
    int expand_rows, sdev_rows, grad_groups, min_waves;
    // profiling
    uint32_t profiling;  // bit i set: bracket kernel family i with HIP events
    std::vector<ProfSpan> spans;
    size_t spans_used;
    double prof_total_us[MUSICA_KERNEL_COUNT];
    uint64_t prof_count[MUSICA_KERNEL_COUNT];
    bool needs_reset;        // a step failed (launch / sync error): the self-resetting tickets of k_minmax_u16 and k_grad_recount_curve may hold a
                             // partial count, which would leave every later launch without a last-ticket block — zeroed before the next step
    std::vector<void*> allocations;
    // Image lanes (musica_execute of a context with a batch, from page-locked host memory): shallow copies of the context for one or
    // two images each — device pointers moved to those images — whose one-stream script is enqueued behind the host-to-device copy of
    // just those images. Every stage of the path is per image, so nothing changes in the results; the first images' kernels run under
    // the remaining copies. Created on first use.
    std::vector<musica_ctx*> lanes;
    hipStream_t lane_stream[3];
    hipStream_t lane_copy[2];        // the images' copies ([1]: unused; two alternating copy streams made every copy twice as long)
    hipEvent_t lane_done[3], lane_start;
    std::vector<hipEvent_t> img_copied;   // one per image
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// MUSICA_TIMING=1: where the host-side time of create / save goes, one line per phase on stderr (the drop-in CLI is one process per
// image: its wall time is start-up, allocation and file I/O, not the pipeline)
struct Tick {
    bool on;
    std::chrono::high_resolution_clock::time_point t;
    const char* what;
    explicit Tick(const char* w) : on(env_int("MUSICA_TIMING", 0) != 0), t(std::chrono::high_resolution_clock::now()), what(w) {}
    void lap(const char* phase) {
        if (!on) return;
        const auto n = std::chrono::high_resolution_clock::now();
        fprintf(stderr, "[musica timing] %s: %s %.2f ms\n", what, phase, std::chrono::duration<float, std::milli>(n - t).count());
        t = n;
    }
};

template <typename T>
static bool dalloc(musica_ctx* c, T** out, size_t count) {
    void* p = nullptr;
    if (count == 0) count = 1;
    if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) return false;
    // "never-written texels read as 0" (Q2). hipMemset runs on the null stream, which the context's non-blocking streams do not wait for:
    // drain it here, or a buffer allocated on first use (the 8-bit output, the second input buffer) could be zeroed AFTER the first
    // kernel or copy has written it (seen once as zero rows at the top of saveOutImage's pixels)
    if (hipMemset(p, 0, count * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { hipFree(p); return false; }
    c->allocations.push_back(p);
    *out = (T*)p;
    return true;
}

// highContrastFactor / lowContrastFactor per level: src/vk_processing.cpp:259-293, both forms of each (the reference picks one per
// #define LINEAR_*_CONTRAST_LEVELS_REDUCTION, include/vk_processing.h:16-17; musica_tunables carries the choice).
static musica_contrast_params host_contrast_params(uint32_t i, uint32_t levels, const musica_tunables& t) {
    const uint32_t coarserLevelsStart = MUSICA_COARSER_LEVELS_START;
    const float highContrastMaxReduction = t.high_contrast_max_reduction, lowContrastMaxEnhancment = t.low_contrast_max_enhancement;  // vk_processing.h:48-49
    musica_contrast_params cp;
    const uint32_t coarserLevelsCount = levels - coarserLevelsStart;
    if (i < coarserLevelsStart) cp.highContrastFactor = 1.0f;
    else if (t.linear_high_contrast) {
        // :264-268 — divides by (pyramidLevels - coarserLevelsStart - 1): 0 / 0 at L = 4, taken as "no reduction" like the power form's exponent 0
        cp.highContrastFactor = coarserLevelsCount > 1
            ? 1.0f - (float)(i - coarserLevelsStart) * (1.0f - highContrastMaxReduction) / (float)(levels - coarserLevelsStart - 1) : 1.0f;
    } else {
        // the reference divides by (coarserLevelsCount - 1): 0/0 at L = 4 — taken as exponent 0 there
        const float e = coarserLevelsCount > 1 ? (float)(i - coarserLevelsStart) / (float)(coarserLevelsCount - 1) : 0.0f;
        cp.highContrastFactor = powf(highContrastMaxReduction, e);
    }
    if (i >= coarserLevelsStart) cp.lowContrastFactor = 1.0f;
    else if (t.linear_low_contrast) cp.lowContrastFactor = lowContrastMaxEnhancment - (float)i * ((lowContrastMaxEnhancment - 1.0f) / (float)coarserLevelsStart);   // :284-286
    else cp.lowContrastFactor = powf(lowContrastMaxEnhancment, 1.0f - ((float)i / (float)coarserLevelsStart));   // :289-291
    return cp;
}

// src/vk_processing.cpp:321-325; the buffer bound to band level l is index l (:1518-1520).
static musica_nr_params host_nr_params(uint32_t i, const musica_tunables& t) {
    const float nrHighCnr = t.nr_high_cnr, nrMaxHighFactor = t.nr_max_high_factor, nrLowCnr = t.nr_low_cnr, nrMinLowFactor = t.nr_min_low_factor;  // vk_processing.h:39-42
    musica_nr_params q;
    q.highCnr = nrHighCnr;
    q.highFactor = nrMaxHighFactor - (nrMaxHighFactor - 1.0f) * ((float)i / (float)MUSICA_CNR_LEVEL);
    q.lowCnr = nrLowCnr;
    q.lowFactor = nrMinLowFactor + (1.0f - nrMinLowFactor) * ((float)i / (float)MUSICA_CNR_LEVEL);
    return q;
}

static void autotune(musica_ctx* c);
static bool rb_level(const musica_ctx* c, int i);
static void copy_rows(musica_ctx* dst, const musica_ctx* src) {
    if (dst == src) return;
    memcpy(dst->rows_expand, src->rows_expand, sizeof(src->rows_expand));
    memcpy(dst->rows_sdev, src->rows_sdev, sizeof(src->rows_sdev));
    memcpy(dst->rows_rb, src->rows_rb, sizeof(src->rows_rb));
}

static uint32_t cnr_scale(int S, int cnrS) { return (uint32_t)ceilf((float)S / (float)cnrS); }  // noise_reduction.comp:38

static int g_min_waves = 2048;
static int pick_rows(int dflt, int min_rows, int S, int rows_total, int batch, int min_waves = 0) {
    int rpw = dflt;
    const int strips = (S + kStripCols - 1) / kStripCols;
    if (min_waves <= 0) min_waves = g_min_waves;
    while (rpw > min_rows) {
        const long waves = (long)strips * ((rows_total + rpw - 1) / rpw) * batch;
        if (waves >= min_waves) break;
        rpw /= 2;
    }
    return rpw < min_rows ? min_rows : rpw;
}
// Rows per wavefront of the sdev + noise-histogram launch of level i; 0 = one 16-row run per workgroup (k_sdev_hist_run), the
// form for every launch that cannot fill the chip with 16-row marches (MUSICA_SDEV_RUN=0 / 1: never / always).
static int sdev_rows_default(const musica_ctx* c, int i, int batch) {
    const int mode = env_int("MUSICA_SDEV_RUN", -1);
    // 16-row marches of this launch: with fewer than one per SIMD (1024) each walks its 20 dependent row trips alone (17 - 18 us
    // whatever the size); the run form costs 9 - 13 us there but reads its input twice: slower once the marches fill the chip
    // (8 x 2048^2, every kernel alone: level 0 74.7 against 61.5 us, level 1 27.4 / 23.6, level 2 13.1 / 18.8, level 3 9.9 / 18.1)
    const long marches = (long)((c->lv[i].S + kStripCols - 1) / kStripCols) * ((c->lv[i].S + kHistArea - 1) / kHistArea) * batch;
    if (mode == 1 || (mode < 0 && marches < 1024)) return 0;
    return pick_rows(c->sdev_rows, 16, c->lv[i].S, c->lv[i].S, batch);
}

extern "C" {

uint32_t musica_abi_version(void) { return MUSICA_ABI_VERSION; }
const char* musica_last_error(void) { return g_last_error.c_str(); }

int musica_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void musica_destroy(musica_ctx* c) {
    if (!c) return;
    hipSetDevice(c->p.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (int k = 0; k < 3; k++) if (c->lane_stream[k]) hipStreamSynchronize(c->lane_stream[k]);
    for (musica_ctx* v : c->lanes) {
        for (auto& sp : v->spans) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
        for (int k = 0; k < kGraphSlots; k++) if (v->graph_exec[k]) hipGraphExecDestroy(v->graph_exec[k]);
        delete v;
    }
    for (int k = 0; k < 3; k++) {
        if (c->lane_stream[k]) { hipStreamSynchronize(c->lane_stream[k]); hipStreamDestroy(c->lane_stream[k]); }
        if (c->lane_done[k]) hipEventDestroy(c->lane_done[k]);
    }
    if (c->lane_start) hipEventDestroy(c->lane_start);
    for (int k = 0; k < 2; k++) if (c->lane_copy[k]) { hipStreamSynchronize(c->lane_copy[k]); hipStreamDestroy(c->lane_copy[k]); }
    for (hipEvent_t e : c->img_copied) hipEventDestroy(e);
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (int k = 0; k < 2; k++) {
        if (c->ev_copied[k]) hipEventDestroy(c->ev_copied[k]);
        if (c->ev_consumed[k]) hipEventDestroy(c->ev_consumed[k]);
    }
    for (auto& s : c->spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (void* p : c->allocations) hipFree(p);
    if (c->h_out8) hipHostFree(c->h_out8);
    if (c->h_bmp) hipHostFree(c->h_bmp);
    for (int k = 0; k < kGraphSlots; k++) if (c->graph_exec[k]) hipGraphExecDestroy(c->graph_exec[k]);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static musica_ctx* create_impl(const musica_params* params, const musica_tunables* tunables);
void musica_tunables_default(musica_tunables* t) {
    if (!t) return;
    t->nr_high_cnr = 9.0f; t->nr_max_high_factor = 1.2f; t->nr_low_cnr = 3.0f; t->nr_min_low_factor = 0.6f;   // include/vk_processing.h:39-42
    t->high_contrast_max_reduction = 0.2f; t->low_contrast_max_enhancement = 3.0f;                             // :48-49
    t->linear_low_contrast = 0u; t->linear_high_contrast = 0u;                                                 // :16-17 (commented out)
}
int musica_get_tunables(const musica_ctx* c, musica_tunables* out) {
    if (!c || !out) return fail("musica_get_tunables: NULL argument");
    *out = c->tun;
    return 1;
}
musica_ctx* musica_create(const musica_params* params) { return musica_create_ex(params, nullptr); }
musica_ctx* musica_create_ex(const musica_params* params, const musica_tunables* tunables) {
    try {
        return create_impl(params, tunables);
    } catch (const std::exception& e) {
        fail("musica_create: %s", e.what());
    } catch (...) {
        fail("musica_create: unknown C++ exception");
    }
    return nullptr;
}

static musica_ctx* create_impl(const musica_params* params, const musica_tunables* tunables) {
    if (!params) { fail("musica_create: params is NULL"); return nullptr; }
    musica_tunables tun;
    musica_tunables_default(&tun);
    if (tunables) {
        tun = *tunables;
        const float v[6] = {tun.nr_high_cnr, tun.nr_max_high_factor, tun.nr_low_cnr, tun.nr_min_low_factor, tun.high_contrast_max_reduction, tun.low_contrast_max_enhancement};
        for (float x : v)
            if (!(x == x) || x > 3.0e38f || x < -3.0e38f) { fail("musica_create_ex: a tunable is not a finite number"); return nullptr; }
        if (tun.nr_high_cnr == tun.nr_low_cnr) { fail("musica_create_ex: nr_high_cnr == nr_low_cnr (the slope of linearFunction, noise_reduction.comp:28, divides by their difference)"); return nullptr; }
    }
    const uint32_t N = params->image_size;
    // 16384: a level-0 f32 plane is then 1 GiB — the kernels address planes through buffer descriptors with 32-bit
    // byte offsets and use bit 31 as the "nothing to load" marker, so a plane must stay below 2 GiB
    if (N < 16 || N > 16384) { fail("musica_create: image_size %u out of range [16, 16384]", N); return nullptr; }
    uint32_t Lref = 0;
    while ((1u << Lref) < N) Lref++;  // pyramidLevels = ceil(log2(imageSize)), src/vk_processing.cpp:1989
    const uint32_t L = params->levels ? params->levels : Lref;
    if (L < MUSICA_MIN_LEVELS || L > Lref || L > MUSICA_MAX_LEVELS) {
        fail("musica_create: levels %u out of range [%d, %u] for image_size %u", L, MUSICA_MIN_LEVELS, Lref, N);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail("musica_create: no HIP device available (this library has no CPU path)");
        return nullptr;
    }
    if (params->device < 0 || params->device >= ndev) { fail("musica_create: device %d out of range (%d devices)", params->device, ndev); return nullptr; }
    if (hipSetDevice(params->device) != hipSuccess) { fail("musica_create: hipSetDevice(%d) failed", params->device); return nullptr; }

    musica_ctx* c = new musica_ctx();
    c->p = *params;
    if (c->p.flags & MUSICA_FLAG_ONE_SHOT) c->p.flags |= MUSICA_FLAG_NO_AUTOTUNE | MUSICA_FLAG_NO_GRAPH;
    params = &c->p;
    c->tun = tun;
    c->N = (int)N; c->L = (int)L; c->B = params->batch ? (int)params->batch : 1;
    c->p.levels = L; c->p.batch = (uint32_t)c->B;
    c->ref_order = (params->flags & MUSICA_FLAG_REFERENCE_ORDER) ? 1 : 0;
    c->generic = (params->flags & MUSICA_FLAG_GENERIC_KERNELS) != 0 || c->ref_order;   // the literal order lives in the one-thread-per-texel kernels
    c->tuning = false;
    c->stream = nullptr; c->side = nullptr; c->ev_fork = nullptr; c->ev_join = nullptr; c->profiling = 0; c->spans_used = 0; c->cur_input = nullptr;
    c->d_out8 = nullptr; c->h_out8 = nullptr; c->h_bmp = nullptr; c->needs_reset = false;
    for (int k = 0; k < 3; k++) { c->lane_stream[k] = nullptr; c->lane_done[k] = nullptr; }
    c->lane_start = nullptr; c->lane_copy[0] = c->lane_copy[1] = nullptr;
    c->d_input2 = nullptr; c->copy_stream = nullptr; c->ev_copied[0] = c->ev_copied[1] = c->ev_consumed[0] = c->ev_consumed[1] = nullptr;
    memset(c->prof_total_us, 0, sizeof(c->prof_total_us));
    memset(c->prof_count, 0, sizeof(c->prof_count));
    int s = (int)N;
    for (int i = 0; i <= c->L; i++) {
        c->lv[i].S = s;
        c->lv[i].pitch = round_up4(s);
        c->lv[i].plane = (size_t)c->lv[i].pitch * s;
        s = (s + 1) / 2;  // ceil(currentImageSize / 2.0f), src/vk_processing.cpp:116,150
    }
    { uint32_t n = N; while (n % 8 == 0) n /= 8; c->min_chain_exact = (n == 1); }
    c->hist_cov = (int)(N / 512u) * 512;  // imageSize / histWorkgroupCoverage groups, src/vk_processing.cpp:2293-2295
    for (int i = 0; i < c->L; i++) c->h_cparams[i] = host_contrast_params((uint32_t)i, L, c->tun);
    for (int i = 0; i < 3; i++) c->h_nr[i] = host_nr_params((uint32_t)i, c->tun);
    c->min_waves = 2048;
    c->expand_rows = env_int("MUSICA_EXPAND_ROWS", 8);
    c->sdev_rows = env_int("MUSICA_SDEV_ROWS", 32) & ~15;
    if (c->sdev_rows < 16) c->sdev_rows = 16;
    g_min_waves = c->min_waves;
    c->grad_groups = 1;

    const size_t B = (size_t)c->B;
    // How a step is dispatched (DESIGN.md section 4 has the measurements behind every line; musica_get_dispatch reports the choice):
    //  * MUSICA_FLAG_LINEAR: ONE in-order stream in the reference's order, replayed as a graph — the form for contexts whose steps
    //    run beside other contexts' steps (musica_pipeline_*): such a context creates one stream, so that the runtime's round-robin
    //    puts consecutive contexts on different hardware queues (4 by default) and nothing of a step waits for another queue;
    //  * a context that runs alone with a small step — one image below 2048^2, or a batch of up to 3072^2 texels — is mostly
    //    launches smaller than their fixed cost: eager launches on one stream (a graph node costs more than a kernel launched behind
    //    its predecessor, a cross-stream join more than it hides: 512^2 0.081 ms against 0.116 for three streams + graph, 1024^2
    //    0.100 / 0.131, 8 x 1024^2 0.170 / 0.196, 2 x 2048^2 0.180 / 0.207);
    //  * everything larger: TWO streams (enqueue_fork: the analysis launches beside the reduce tail and the constant-gain expand
    //    slots, one fork and one join), replayed as a graph (2048^2 L6 0.138 against 0.142 on one stream, 4096^2 L8 + CLAHE 0.306 /
    //    0.319, 8192^2 L10 0.747 / 0.779, 4 x 2048^2 0.264 / 0.263 (three streams: 0.287), 2 x 4096^2 0.433 / 0.457 (0.451), 8 x 2048^2
    //    0.434 / 0.443 (0.431)); one image with a pyramid of 11 or more levels: eager (3072^2 L12 0.214 against 0.220 as a graph and
    //    0.240 on one stream, 4096^2 L12 0.277 / 0.292 / 0.308);
    //  * one-shot contexts (MUSICA_FLAG_ONE_SHOT: musica-standalone): one stream — creating a second one costs more than one step saves;
    //  (The three-stream script and the image groups of rounds 1 - 3 won nowhere by more than noise at the end of round 3 and left in round 4.)
    const bool lone = !(params->flags & MUSICA_FLAG_LINEAR);
    const bool one_shot = (params->flags & MUSICA_FLAG_ONE_SHOT) != 0;
    const bool small_step = c->B == 1 ? N < 2048 : (size_t)c->B * N * N <= (size_t)3072 * 3072;
    c->dag = !lone ? 0 : (env_int("MUSICA_STREAMS", (small_step || one_shot) ? 1 : 2) >= 2 ? 2 : 0);
    c->use_graph = !(params->flags & MUSICA_FLAG_NO_GRAPH) && env_int("MUSICA_GRAPH", (lone && (small_step || (c->B == 1 && L >= 11))) ? 0 : 1) != 0;
    Tick tick("create");
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    if (c->dag) {
        ok = ok && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    }
    for (int k = 0; k < kGraphSlots; k++) { c->graph_exec[k] = nullptr; c->graph_input[k] = nullptr; c->graph_used[k] = 0; }
    c->graph_clock = 0;
    c->fuse_u16 = (N % 8) == 0 && !c->generic;
    c->norm_valid = false;
    // fused gradation histogram: streaming level-0 kernels on raw pixels, cnr scale 8 (every N >= 57 with N % 8 == 0), no CLAHE
    // block (it wants the stored relevant image anyway)
    // (a CLAHE context fuses too since its relevant image comes from the raw pixels, k_relevant4<true>; MUSICA_CLAHE_FUSE=0: as before)
    c->clahe_raw = (params->flags & MUSICA_FLAG_CLAHE) && c->fuse_u16 && (N % 4) == 0 && env_int("MUSICA_CLAHE_FUSE", 1) != 0 &&
                   (cnr_scale(c->lv[0].S, c->lv[MUSICA_CNR_LEVEL].S) % 4) == 0;
    c->clahe_in_expand = env_int("MUSICA_CLAHE_IN_EXPAND", 1) != 0;
    c->clahe_one_apply = env_int("MUSICA_CLAHE_ONE_APPLY", 1) != 0;
    c->tiny_tail = env_int("MUSICA_TINY_TAIL", 1) != 0;
    c->grad_one_launch = env_int("MUSICA_GRAD_ONE_LAUNCH", 1) != 0;
    {
        // sdev computed inside the expand launches of levels 0 .. 2 (k_expand_fast<.., SD>) instead of stored by the sdev launch and read back:
        // 8 of a step's 48 bytes per input pixel against ~35 % more vector work in those expand launches. It pays where a step is bound by its
        // bytes — steps in flight (the contexts of a pipeline: MUSICA_FLAG_LINEAR) from 2 x 2048^2 texels per step, a lone context from
        // 8 x 2048^2 (same-box A/B, three steps in flight / one context: 8 x 2048^2 -7.8 % / -0.8 %, 8192^2 -12.4 % / -5.7 %, 4 x 2048^2
        // -2.6 %, 3072^2 L12 -5.0 % / +4.1 %, 4096^2 + CLAHE -1.5 % / +2.5 %, one 2048^2 image +2.5 % / +6.6 %). MUSICA_SDEV_IN_EXPAND=0 | 1 overrides.
        const size_t texels = (size_t)c->B * N * N;
        const bool pays = texels >= ((params->flags & MUSICA_FLAG_LINEAR) ? (size_t)2 : (size_t)8) * 2048 * 2048;
        c->sd_fused = env_int("MUSICA_SDEV_IN_EXPAND", pays ? 1 : 0) != 0 && !c->generic;
    }
    c->sd_active = false; c->sdev_stored = true;
    // every level's sdev pass in one launch for batches and for the contexts of a pipeline (same-box A/B: a lone 8 x 2048^2 context -3.5 %, 8192^2 -1.1 %;
    // three steps in flight 8 x 2048^2 -1 %, 8192^2 -1.5 %, 3072^2 L12 -3.5 %); a lone context with one image keeps one launch per marching level:
    // beside the reduce tail of its two-stream script the merged launch slows the tail's small launches (3072^2 L12 +4 %, 2048^2 +0.7 %)
    // the pairs of the one-stream script (k_rb_sdev) for the contexts of a pipeline from 2 x 2048^2 texels per step (same-box A/B, three steps in flight:
    // 8 x 2048^2 -3.1 %, 8192^2 -2.0 %, 4096^2 + CLAHE -2.1 %, 3072^2 L12 +0.6 %, one 2048^2 image +2.8 %); a lone one-stream context is slower with them
    // (512^2 +17 %, 1024^2 +3 %, 1536^2 +7 %, 2 / 4 x 1024^2 +5 / +4 %): alone on the chip a pair costs what its two launches cost one after the other
    c->pair_rb_sdev = env_int("MUSICA_PAIR_RB_SDEV", ((params->flags & MUSICA_FLAG_LINEAR) && (size_t)c->B * N * N >= (size_t)2 * 2048 * 2048) ? 1 : 0) != 0;
    c->sdev_one_launch = env_int("MUSICA_SDEV_ONE_LAUNCH", ((params->flags & MUSICA_FLAG_LINEAR) || c->B > 1) ? 1 : 0) != 0;
    c->fuse_gh = env_int("MUSICA_FUSE_GH", 1) != 0 && c->fuse_u16 && (!(params->flags & MUSICA_FLAG_CLAHE) || c->clahe_raw) &&
                 cnr_scale(c->lv[0].S, c->lv[MUSICA_CNR_LEVEL].S) == 8;
    tick.lap("streams + events");
    ok = ok && dalloc(c, &c->d_input, B * N * N);
    ok = ok && dalloc(c, &c->d_minmax, B * kMinMaxStride);
    ok = ok && dalloc(c, &c->d_mm_slots, B * kMinMaxSlots);
    ok = ok && dalloc(c, &c->d_mm_ticket, B * kMinMaxStride);   // one 128-byte line per image
    ok = ok && dalloc(c, &c->d_gr_ticket, B * kGradTicketStride);
    ok = ok && dalloc(c, &c->d_norm, B * c->lv[0].plane);
    for (int i = 0; i < c->L && ok; i++) {
        ok = ok && dalloc(c, &c->d_down[i], B * c->lv[i + 1].plane);
        ok = ok && dalloc(c, &c->d_band[i], B * c->lv[i].plane);
        ok = ok && dalloc(c, &c->d_recon[i], B * c->lv[i].plane);
        if (i <= MUSICA_CNR_LEVEL) ok = ok && dalloc(c, &c->d_sdev[i], B * c->lv[i].plane);
    }
    ok = ok && dalloc(c, &c->d_noise_hist, B * 4 * MUSICA_NOISE_BINS);
    ok = ok && dalloc(c, &c->d_noise_max, B * L);
    ok = ok && dalloc(c, &c->d_curves, B * L);
    ok = ok && dalloc(c, &c->d_luts, B * MUSICA_COARSER_LEVELS_START);
    ok = ok && dalloc(c, &c->d_cparams, (size_t)L);
    ok = ok && dalloc(c, &c->d_cnr, B * c->lv[MUSICA_CNR_LEVEL].plane);
    ok = ok && dalloc(c, &c->d_grad_hist, B * MUSICA_GRAD_BINS);
    ok = ok && dalloc(c, &c->d_grad_hist_b, B * MUSICA_GRAD_BINS);
    ok = ok && dalloc(c, &c->d_gzero, B);
    ok = ok && dalloc(c, &c->d_thr090, B);
    ok = ok && dalloc(c, &c->d_stats_partial, B * kStatsMaxBlocks);
    ok = ok && dalloc(c, &c->d_plot, (size_t)MUSICA_HIST_RENDER_WIDTH * MUSICA_HIST_RENDER_HEIGHT);
    c->d_le090 = nullptr;
    if (c->fuse_gh) ok = ok && dalloc(c, &c->d_le090, B * (size_t)c->lv[1].S * (c->lv[0].S / 8));
    ok = ok && dalloc(c, &c->d_grad_max, B);
    ok = ok && dalloc(c, &c->d_gcurve, B);
    ok = ok && dalloc(c, &c->d_graded, B * c->lv[0].plane);
    ok = ok && dalloc(c, &c->d_scratch, B * c->lv[0].plane);
    ok = ok && dalloc(c, &c->d_stats, B);
    c->d_clahe_hist = nullptr; c->d_clahe_pts = nullptr; c->d_clahe_graded = nullptr;
    if (ok && (params->flags & MUSICA_FLAG_CLAHE)) {
        const size_t tb = MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
        ok = ok && dalloc(c, &c->d_clahe_hist, B * tb);
        ok = ok && dalloc(c, &c->d_clahe_pts, B * tb);
        ok = ok && dalloc(c, &c->d_clahe_graded, B * c->lv[0].plane);
    }
    tick.lap("device buffers");
    ok = ok && hipMemcpy(c->d_cparams, c->h_cparams, sizeof(musica_contrast_params) * L, hipMemcpyHostToDevice) == hipSuccess;
    tick.lap("parameter upload");
    if (!ok) {
        fail("musica_create: device allocation failed (%s)", hipGetErrorString(hipGetLastError()));
        musica_destroy(c);
        return nullptr;
    }
    c->cur_input = c->d_input;
    c->cur = c->stream;
    for (int i = 0; i < c->L; i++) {
        c->rows_expand[i] = pick_rows(c->expand_rows, 1, c->lv[i].S, c->lv[i + 1].S, c->B);
        if (i <= MUSICA_CNR_LEVEL) c->rows_sdev[i] = sdev_rows_default(c, i, c->B);
        c->rows_rb[i] = pick_rows(env_int("MUSICA_RB_ROWS", 16), 1, c->lv[i].S, c->lv[i + 1].S, c->B);
    }
    const bool tune = !(params->flags & MUSICA_FLAG_NO_AUTOTUNE) && env_int("MUSICA_AUTOTUNE", 1) && !c->generic;
    if (tune) autotune(c);
    return c;
}

uint32_t musica_get_image_size(const musica_ctx* c) { return c ? (uint32_t)c->N : 0; }
uint32_t musica_get_levels(const musica_ctx* c) { return c ? (uint32_t)c->L : 0; }
uint32_t musica_get_batch(const musica_ctx* c) { return c ? (uint32_t)c->B : 0; }
int musica_get_dispatch(const musica_ctx* c, int* streams, int* graph) {
    if (!c) return 0;
    if (streams) *streams = c->dag == 0 ? 1 : 2;
    if (graph) *graph = c->use_graph ? 1 : 0;
    return 1;
}
int musica_fuses_gradation_histogram(const musica_ctx* c) { return (c && c->fuse_gh && !c->generic) ? 1 : 0; }
int musica_fuses_reduce_band(const musica_ctx* c) { return (c && rb_level(c, 0)) ? 1 : 0; }
int musica_fuses_sdev(const musica_ctx* c) { return (c && c->sd_fused && rb_level(c, 0)) ? 1 : 0; }
uint32_t musica_get_level_size(const musica_ctx* c, uint32_t level) { return (c && (int)level <= c->L) ? (uint32_t)c->lv[level].S : 0; }

}  // extern "C"

// ---- profiling spans --------------------------------------------------------------------
struct Span {
    musica_ctx* c;
    ProfSpan* s;
    hipStream_t st;
    Span(musica_ctx* ctx, int id) : c(ctx), s(nullptr), st(ctx->cur) {
        if (!((c->profiling >> id) & 1u)) return;
        if (c->spans_used == c->spans.size()) {
            ProfSpan n;
            n.id = id;
            hipEventCreate(&n.a);
            hipEventCreate(&n.b);
            c->spans.push_back(n);
        }
        s = &c->spans[c->spans_used++];
        s->id = id;
        hipEventRecord(s->a, st);
    }
    ~Span() {
        if (s) hipEventRecord(s->b, st);
    }
};

static void collect_spans(musica_ctx* c) {
    for (size_t i = 0; i < c->spans_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->spans[i].a, c->spans[i].b) == hipSuccess) {
            c->prof_total_us[c->spans[i].id] += (double)ms * 1000.0;
            c->prof_count[c->spans[i].id] += 1;
        }
    }
    c->spans_used = 0;
}

// ---- the dispatch script ----------------------------------------------------------------
static const float* level_input(musica_ctx* c, int i) { return i == 0 ? c->d_norm : c->d_down[i - 1]; }  // src/vk_processing.cpp:758-761

// stage "norm" (src/vk_processing.cpp:2182-2222)
// with_clears: the launch also zeroes the histograms (src/vk_processing.cpp:2153-2162)
static void enqueue_norm(musica_ctx* c, bool with_clears) {
    {
        Span sp(c, MUSICA_KERNEL_MINMAX);
        if (with_clears) launch_minmax(c->stream, c->cur_input, c->N, c->d_minmax, c->d_mm_slots, c->d_mm_ticket, c->B, c->d_noise_hist, c->d_grad_hist, c->d_grad_hist_b, c->d_gzero, c->d_clahe_hist);
        else launch_minmax(c->stream, c->cur_input, c->N, c->d_minmax, c->d_mm_slots, c->d_mm_ticket, c->B);
    }
    c->norm_valid = false;
    if (!c->fuse_u16 || (c->d_clahe_hist && !c->clahe_raw)) {  // a CLAHE block that reads the stored normalized image through k_relevant
        Span sp(c, MUSICA_KERNEL_NORMALIZE);
        launch_normalize(c->stream, c->cur_input, c->d_norm, c->lv[0], c->d_minmax, c->min_chain_exact, c->B);
        c->norm_valid = true;
    }
}
// The normalized image for getters / dumps when the hot path skipped it.
static void ensure_normalized(musica_ctx* c) {
    if (c->norm_valid) return;
    launch_normalize(c->stream, c->cur_input, c->d_norm, c->lv[0], c->d_minmax, c->min_chain_exact, c->B);
    c->norm_valid = true;
}

// reduce + band of a level in one launch: every level whose side is a multiple of 8 (level 0 then reads the raw pixels)
static bool rb_level(const musica_ctx* c, int i) { return !c->generic && c->lv[i].S >= 8 && (c->lv[i].S % 8) == 0; }
static void run_reduce_band(musica_ctx* c, int i, int rows) {
    if (i == 0)
        launch_reduce_band_u16(c->cur, c->cur_input, c->d_down[0], c->d_band[0], c->lv[0], c->lv[1], c->B, rows, c->d_minmax, c->min_chain_exact, c->d_le090);
    else
        launch_reduce_band(c->cur, level_input(c, i), c->d_down[i], c->d_band[i], c->lv[i], c->lv[i + 1], c->B, rows);
}
// reduce and band of level i, as one launch where that form applies, else the one-thread-per-texel kernels
static void run_reduce_and_band(musica_ctx* c, int i) {
    if (rb_level(c, i)) { Span sp(c, i == 0 ? MUSICA_KERNEL_REDUCE_L0 : MUSICA_KERNEL_REDUCE_REST); run_reduce_band(c, i, c->rows_rb[i]); return; }
    { Span sp(c, i == 0 ? MUSICA_KERNEL_REDUCE_L0 : MUSICA_KERNEL_REDUCE_REST); launch_reduce(c->cur, level_input(c, i), c->lv[i], c->d_down[i], c->lv[i + 1], c->B, true, i == 0 ? 0 : 1, c->ref_order); }
    { Span sp(c, i == 0 ? MUSICA_KERNEL_BAND_L0 : MUSICA_KERNEL_BAND_REST); launch_band(c->cur, level_input(c, i), c->d_down[i], c->d_band[i], c->lv[i], c->lv[i + 1], c->B, c->ref_order); }
}
// level i's sdev image is neither stored nor read by this step: its expand launch computes it (levels below the cnr level whose side
// takes the streaming kernels)
static bool sd_level(const musica_ctx* c, int i) { return c->sd_active && i < MUSICA_CNR_LEVEL && rb_level(c, i); }
static void run_sdev_level(musica_ctx* c, int i, int rows) {
    if (c->ref_order) {   // img_sdev.comp literally, then noise_hist.comp on the stored image
        launch_sdev_literal(c->cur, c->d_band[i], c->d_sdev[i], c->lv[i], c->B);
        launch_noise_hist_only(c->cur, c->d_sdev[i], c->lv[i], c->d_noise_hist + (size_t)i * MUSICA_NOISE_BINS, (size_t)4 * MUSICA_NOISE_BINS, c->hist_cov, c->B);
        return;
    }
    launch_sdev_hist(c->cur, c->d_band[i], sd_level(c, i) ? nullptr : c->d_sdev[i], c->lv[i], c->d_noise_hist + (size_t)i * MUSICA_NOISE_BINS,
                     (size_t)4 * MUSICA_NOISE_BINS, c->hist_cov, c->B, rows);
}
// The stored sdev images of levels whose hot path does not store them (getters, dumps, stage entry points, the generic kernels).
static void ensure_sdev(musica_ctx* c) {
    if (c->sdev_stored) return;
    for (int i = 0; i < MUSICA_CNR_LEVEL && i < c->L; i++)
        if (rb_level(c, i)) launch_sdev_only(c->stream, c->d_band[i], c->d_sdev[i], c->lv[i], c->B);
    c->sdev_stored = true;
}

// The first level of the tiny tail (k_tiny_tail: reduce + band of levels T .. L-1 and their expand slots in one launch), L if none:
// levels above the cnr level whose side is at most kTailSide, where there are at least two of them (a lone level is two launches
// either way). The whole-step scripts use it; the stage entry points (debug_stage) keep one launch per level and stage.
static int tail_first(const musica_ctx* c) {
    if (!c->tiny_tail) return c->L;
    int T = c->L;
    while (T - 1 > MUSICA_CNR_LEVEL && c->lv[T - 1].S <= kTailSide) T--;
    if (c->L - T > kTailMax) T = c->L - kTailMax;
    return c->L - T >= 2 ? T : c->L;
}
static void run_tiny_tail(musica_ctx* c, int T) {
    TailArgs a;
    a.n = c->L - T;
    for (int k = 0; k < a.n; k++) {
        const int i = T + k;
        a.l[k] = TailLevel{level_input(c, i), c->d_down[i], c->d_band[i], c->d_recon[i], c->lv[i].plane, c->lv[i].S, c->lv[i].pitch,
                           c->h_cparams[i].highContrastFactor};
    }
    for (int k = a.n; k < kTailMax; k++) a.l[k] = a.l[0];
    a.Sl = c->lv[c->L].S; a.lpitch = c->lv[c->L].pitch; a.lplane = c->lv[c->L].plane;
    a.ref = c->ref_order;
    Span sp(c, MUSICA_KERNEL_REDUCE_REST);
    launch_tiny_tail(c->cur, a, c->B);
}
// reduce + band of levels first .. L-1 and, with the tiny tail, the expand slots of its levels; returns the first level whose
// expand slot is still to run (L: none)
static int enqueue_reduce_from(musica_ctx* c, int first) {
    const int T = tail_first(c);
    for (int i = first; i < T; i++) run_reduce_and_band(c, i);
    if (T < c->L) run_tiny_tail(c, T);
    return T;
}

// stage "red" (src/vk_processing.cpp:2233-2273)
static void enqueue_reduce(musica_ctx* c) {
    for (int i = 0; i < c->L; i++) run_reduce_and_band(c, i);
}

// stage "anly" (src/vk_processing.cpp:2284-2357)
// levels first .. MUSICA_CNR_LEVEL of sdev + noise histogram in one launch (their launches depend on nothing but their own
// band image, and the 16-row-run form of a small level is a few dozen workgroups)
static void run_sdev_levels(musica_ctx* c, int first) {
    const float* band[kSdevRunLevelsMax];
    float* sdev[kSdevRunLevelsMax];
    uint32_t* hist[kSdevRunLevelsMax];
    LevelDesc lv[kSdevRunLevelsMax];
    int n = 0;
    for (int i = first; i <= MUSICA_CNR_LEVEL; i++, n++) {
        band[n] = c->d_band[i]; sdev[n] = sd_level(c, i) ? nullptr : c->d_sdev[i]; lv[n] = c->lv[i];
        hist[n] = c->d_noise_hist + (size_t)i * MUSICA_NOISE_BINS;
    }
    launch_sdev_hist_runs(c->cur, n, band, sdev, lv, hist, (size_t)4 * MUSICA_NOISE_BINS, c->hist_cov, c->B);
}
// the first level from which every sdev launch is the 16-row-run form (MUSICA_CNR_LEVEL + 1: none)
static int sdev_runs_from(const musica_ctx* c) {
    if (c->ref_order) return MUSICA_CNR_LEVEL + 1;
    int first = MUSICA_CNR_LEVEL + 1;
    while (first > 0 && c->rows_sdev[first - 1] <= 0) first--;
    return first;
}
// every level's sdev + noise-histogram pass in one launch, each level in its own form (k_sdev_hist_levels)
static void run_sdev_all_levels(musica_ctx* c, int first = 0) {
    const float* band[kSdevRunLevelsMax];
    float* sdev[kSdevRunLevelsMax];
    uint32_t* hist[kSdevRunLevelsMax];
    LevelDesc lv[kSdevRunLevelsMax];
    int rows[kSdevRunLevelsMax];
    int n = 0;
    for (int i = first; i <= MUSICA_CNR_LEVEL; i++, n++) {
        band[n] = c->d_band[i]; sdev[n] = sd_level(c, i) ? nullptr : c->d_sdev[i]; lv[n] = c->lv[i];
        hist[n] = c->d_noise_hist + (size_t)i * MUSICA_NOISE_BINS;
        rows[n] = c->rows_sdev[i] > 0 ? c->rows_sdev[i] : 0;
    }
    launch_sdev_hist_levels(c->cur, n, band, sdev, lv, hist, rows, (size_t)4 * MUSICA_NOISE_BINS, c->hist_cov, c->B);
}
static void enqueue_analysis(musica_ctx* c, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    if (c->sdev_one_launch && !c->ref_order && !c->tuning) {
        { Span sp(c, MUSICA_KERNEL_SDEV_HIST); run_sdev_all_levels(c); }
        Span sp2(c, MUSICA_KERNEL_CURVES);
        launch_curves_cnr(st, c->d_noise_hist, (size_t)4 * MUSICA_NOISE_BINS, c->d_noise_max, c->d_curves, c->d_cparams, c->L, c->B, c->d_luts,
                          c->d_sdev[MUSICA_CNR_LEVEL], c->d_cnr, c->lv[MUSICA_CNR_LEVEL], c->d_minmax, c->min_chain_exact, c->d_thr090);
        return;
    }
    int merged = sdev_runs_from(c);
    if (merged >= MUSICA_CNR_LEVEL) merged = MUSICA_CNR_LEVEL + 1;   // one level is its own launch
    for (int i = 0; i < merged; i++) {  // i < coarserLevelsStart || i <= cnrLevel, :2285
        Span sp(c, MUSICA_KERNEL_SDEV_HIST);
        run_sdev_level(c, i, c->rows_sdev[i]);
    }
    if (merged <= MUSICA_CNR_LEVEL) {
        Span sp(c, MUSICA_KERNEL_SDEV_HIST);
        run_sdev_levels(c, merged);
    }
    {   // curves of every level + cnr of level 3 in one launch (kernels_analysis.hip k_curves_cnr)
        Span sp(c, MUSICA_KERNEL_CURVES);
        launch_curves_cnr(st, c->d_noise_hist, (size_t)4 * MUSICA_NOISE_BINS, c->d_noise_max, c->d_curves, c->d_cparams, c->L, c->B, c->d_luts,
                          c->d_sdev[MUSICA_CNR_LEVEL], c->d_cnr, c->lv[MUSICA_CNR_LEVEL], c->d_minmax, c->min_chain_exact, c->d_thr090);
    }
}

static ExpandArgs expand_args(musica_ctx* c, int lvl, float* dst) {
    const LevelDesc& lf = c->lv[lvl];
    const LevelDesc& lc = c->lv[lvl + 1];
    const LevelDesc& l3 = c->lv[MUSICA_CNR_LEVEL];
    ExpandArgs a;
    a.prev = lvl == c->L - 1 ? c->d_down[c->L - 1] : c->d_recon[lvl + 1];  // src/vk_processing.cpp:930-934
    a.band = c->d_band[lvl];
    a.sdev = lvl <= MUSICA_CNR_LEVEL && !sd_level(c, lvl) ? c->d_sdev[lvl] : nullptr;
    a.cnr = c->d_cnr;
    a.recon = dst;
    a.curves = c->d_curves + lvl;
    a.luts = c->d_luts + (lvl < MUSICA_COARSER_LEVELS_START ? lvl : 0);
    a.curve_stride = (size_t)c->L;
    a.S = lf.S; a.pitch = lf.pitch; a.plane = lf.plane;
    a.Sc = lc.S; a.cpitch = lc.pitch; a.cplane = lc.plane;
    a.cnrS = l3.S; a.cnrPitch = l3.pitch; a.cnrPlane = l3.plane;
    a.cnrScale = (int)cnr_scale(lf.S, l3.S);
    a.high = c->h_cparams[lvl].highContrastFactor;
    const musica_nr_params& q = c->h_nr[lvl < 3 ? lvl : 0];
    a.lowCnr = q.lowCnr; a.lowFactor = q.lowFactor; a.highCnr = q.highCnr; a.highFactor = q.highFactor;
    a.rows_per_wave = c->rows_expand[lvl];
    a.raw = nullptr; a.ghist = nullptr; a.gzero = nullptr; a.thr090 = nullptr; a.le090 = nullptr; a.chist = nullptr;
    a.swz = xcd_swizzle_on();
    a.ref_order = c->ref_order;
    return a;
}
static int gain_mode(int lvl) { return lvl > MUSICA_CNR_LEVEL ? GAIN_CONST : (lvl == MUSICA_CNR_LEVEL ? GAIN_RANGE : GAIN_CURVE); }
static bool uses_nr(int lvl) { return lvl < MUSICA_CNR_LEVEL - 1; }  // currentLevel < cnrLevel - 1, src/vk_processing.cpp:1009-1016

// CLAHE contexts: the level-0 expand launch (rows_per_wave = rows) can also count the CLAHE histogram (k_expand_fast<.., CH>): it has
// the `normalized <= 0.9` bits, and a workgroup's 512 columns x 8 * rows rows touch at most 2 x 2 of the 4 x 4 tiles
static bool clahe_hist_in_expand(const musica_ctx* c, int rows) {
    const int G = c->lv[0].S / MUSICA_CLAHE_TILES;
    return c->d_clahe_hist && c->clahe_raw && c->fuse_gh && !c->generic && c->d_le090 && c->clahe_in_expand && (c->lv[0].S % MUSICA_CLAHE_TILES) == 0 &&
           G >= 512 && 8 * rows <= G;
}
// with_hist: the level-0 launch of a fusing context also accumulates the gradation histogram (enqueue_gradation(c, true) must follow)
static void run_expand_level_h(musica_ctx* c, int lvl, int rows, bool with_hist) {
    ExpandArgs a = expand_args(c, lvl, c->d_recon[lvl]);
    a.rows_per_wave = rows;
    if (with_hist && lvl == 0 && c->fuse_gh && !c->generic) {
        a.raw = c->cur_input; a.ghist = c->d_grad_hist; a.gzero = c->d_gzero; a.thr090 = c->d_thr090;
        a.le090 = c->d_le090;
        if (clahe_hist_in_expand(c, rows)) a.chist = c->d_clahe_hist;
    }
    launch_expand(c->cur, a, gain_mode(lvl), uses_nr(lvl), c->B, c->generic);
}
static void run_expand_level(musica_ctx* c, int lvl, int rows) { run_expand_level_h(c, lvl, rows, true); }

// stages "aply" + "exp" (src/vk_processing.cpp:2361-2431)
static void enqueue_expand(musica_ctx* c, bool with_hist, int top = -1 /* first slot to run, default L - 1 */) {
    for (int lvl = top < 0 ? c->L - 1 : top; lvl >= 0; lvl--) {
        Span sp(c, lvl == 0 ? MUSICA_KERNEL_EXPAND_L0 : MUSICA_KERNEL_EXPAND_REST);
        run_expand_level_h(c, lvl, c->rows_expand[lvl], with_hist);
    }
}

// stage "grad" (src/vk_processing.cpp:2456-2518)
// fused: the level-0 expand launch has already accumulated the histogram (run_expand_level_h with_hist); what is left of K18 + K19 is
// the literal recount of images that hold an exact zero (a launch that returns at once for every other image).
static void enqueue_gradation(musica_ctx* c, bool fused) {
    fused = fused && c->fuse_gh && !c->generic;
    const LevelDesc& l0 = c->lv[0];
    const LevelDesc& l3 = c->lv[MUSICA_CNR_LEVEL];
    const int scale = (int)cnr_scale(l0.S, l3.S);
    const bool ch_done = fused && clahe_hist_in_expand(c, c->rows_expand[0]);   // the level-0 expand launch counted the CLAHE histogram
    const bool one_apply = c->d_clahe_hist && c->clahe_raw && !c->generic && (l0.S % 4) == 0 && c->clahe_one_apply;   // both curves in one pass
    if (c->d_clahe_hist) {  // #ifdef ENABLE_CLAHE block, src/vk_processing.cpp:2471-2489
        hipStream_t cs = c->stream;
        if (c->clahe_raw) {   // relevance computed inside the histogram launch from the raw pixels: no relevant image on the hot path
            launch_clahe(cs, c->d_recon[0], nullptr, c->d_clahe_graded, l0, c->d_clahe_hist, c->d_clahe_pts, c->B, c->cur_input, c->d_thr090, c->d_cnr, &l3, scale,
                         ch_done, !one_apply);
        } else {
            launch_relevant(cs, c->d_norm, c->d_cnr, c->d_scratch, l0, l3, scale, c->B);
            launch_clahe(cs, c->d_recon[0], c->d_scratch, c->d_clahe_graded, l0, c->d_clahe_hist, c->d_clahe_pts, c->B);
        }
    }
    GradArgs g;
    g.img = c->d_recon[0]; g.normalized = c->d_norm; g.cnr = c->d_cnr; g.hist = fused ? c->d_grad_hist_b : c->d_grad_hist;
    g.only_if = fused ? c->d_gzero : nullptr;
    g.N = l0.S; g.pitch = l0.pitch; g.plane = l0.plane;
    g.cnrS = l3.S; g.cnrPitch = l3.pitch; g.cnrPlane = l3.plane; g.cnrScale = scale;
    g.groups_per_wave = c->grad_groups;
    g.raw = c->fuse_u16 ? c->cur_input : nullptr;
    g.minmax = c->d_minmax;
    g.min_chain_exact = c->min_chain_exact;
    if (fused && c->grad_one_launch) {   // the recount of images that hold an exact zero and the tone curve in one launch
        Span sp(c, MUSICA_KERNEL_GRAD_CURVE);
        launch_grad_recount_curve(c->stream, g, c->d_grad_hist, c->d_grad_max, c->d_gcurve, c->d_gr_ticket, c->B);
    } else {
        { Span sp(c, MUSICA_KERNEL_GRAD_HIST); launch_grad_hist(c->stream, g, c->B); }
        Span sp(c, MUSICA_KERNEL_GRAD_CURVE);
        launch_grad_curve(c->stream, c->d_grad_hist, c->d_grad_max, c->d_gcurve, c->B, fused ? c->d_grad_hist_b : nullptr, fused ? c->d_gzero : nullptr);
    }
    if (one_apply) {
        Span sp(c, MUSICA_KERNEL_GRAD_APPLY);
        launch_grad_clahe_apply(c->stream, c->d_recon[0], c->d_clahe_graded, c->d_graded, l0, c->d_clahe_pts, c->d_gcurve, c->B);
        return;
    }
    { Span sp(c, MUSICA_KERNEL_GRAD_APPLY); launch_grad_apply(c->stream, c->d_recon[0], c->d_graded, l0, c->d_gcurve, c->B); }
}

// One in-order stream, the order of the reference's command buffer (dag == 0).
// reduce + band of level i + 1 and the sdev + noise-histogram pass of level i as one launch (k_rb_sdev): both wait for reduce + band of level i only
static void run_rb_sdev_pair(musica_ctx* c, int i) {
    const int r = i + 1;
    RbSdevArgs a;
    a.fine = level_input(c, r); a.down = c->d_down[r]; a.band = c->d_band[r];
    a.S = c->lv[r].S; a.pitch = c->lv[r].pitch; a.plane = c->lv[r].plane;
    a.Sc = c->lv[r + 1].S; a.cpitch = c->lv[r + 1].pitch; a.cplane = c->lv[r + 1].plane;
    a.rows_rb = c->rows_rb[r];
    a.sl.band = c->d_band[i]; a.sl.sdev = sd_level(c, i) ? nullptr : c->d_sdev[i];
    a.sl.hist = c->d_noise_hist + (size_t)i * MUSICA_NOISE_BINS;
    a.sl.rows = c->rows_sdev[i] > 0 ? c->rows_sdev[i] : 0;
    a.hist_stride = (size_t)4 * MUSICA_NOISE_BINS; a.cov = c->hist_cov;
    launch_rb_sdev(c->cur, a, c->lv[i], c->B);
}
// One in-order stream with the pairs: minmax RB0 [RB1 | S0] [RB2 | S1] [RB3 | S2] [RB4 | S3] RB5 .. [tail] curves+cnr E .. gradation — the pyramid's
// dependent chain of ever smaller launches in the shadow of the sdev passes instead of in front of them.
static void enqueue_linear_paired(musica_ctx* c) {
    c->cur = c->stream;
    enqueue_norm(c, true);
    const int T = tail_first(c);
    run_reduce_and_band(c, 0);
    int i = 0;   // the first level whose sdev pass is still to run
    while (i <= MUSICA_CNR_LEVEL && i + 1 < T && rb_level(c, i) && rb_level(c, i + 1)) {
        Span sp(c, MUSICA_KERNEL_SDEV_HIST);
        run_rb_sdev_pair(c, i);
        i++;
    }
    for (int r = i + 1; r < T; r++) run_reduce_and_band(c, r);
    if (T < c->L) run_tiny_tail(c, T);
    if (i <= MUSICA_CNR_LEVEL) { Span sp(c, MUSICA_KERNEL_SDEV_HIST); run_sdev_all_levels(c, i); }
    {
        Span sp(c, MUSICA_KERNEL_CURVES);
        launch_curves_cnr(c->stream, c->d_noise_hist, (size_t)4 * MUSICA_NOISE_BINS, c->d_noise_max, c->d_curves, c->d_cparams, c->L, c->B, c->d_luts,
                          c->d_sdev[MUSICA_CNR_LEVEL], c->d_cnr, c->lv[MUSICA_CNR_LEVEL], c->d_minmax, c->min_chain_exact, c->d_thr090);
    }
    enqueue_expand(c, true, T - 1);
    enqueue_gradation(c, true);
}
static void enqueue_linear(musica_ctx* c) {
    if (c->pair_rb_sdev && !c->ref_order && !c->generic && !c->tuning) { enqueue_linear_paired(c); return; }
    c->cur = c->stream;
    enqueue_norm(c, true);   // with the clears of :2153-2162
    const int T = enqueue_reduce_from(c, 0);
    enqueue_analysis(c);
    enqueue_expand(c, true, T - 1);
    enqueue_gradation(c, true);
}
// Two streams, one fork and one join (dag == 2): the analysis (the sdev + noise-histogram launches and the curves: they read band 0 .. 3
// and nothing of the levels above) runs on the side stream BESIDE the tail of the reduce chain and the constant-gain expand slots
// (levels >= 4 down to expand 3: launches of a few microseconds each that read nothing the analysis writes).
//   stream : minmax RB0 RB1 RB2 RB3 | RB4 .. RB(L-1) E(L-1) .. E3 | (wait) E2 E1 E0 gradation
//   side   :                        | S(0..3) curves+cnr          |
static void enqueue_fork(musica_ctx* c) {
    c->cur = c->stream;
    enqueue_norm(c, true);
    for (int i = 0; i <= MUSICA_CNR_LEVEL; i++) run_reduce_and_band(c, i);
    hipEventRecord(c->ev_fork, c->stream);
    hipStreamWaitEvent(c->side, c->ev_fork, 0);
    c->cur = c->side;
    enqueue_analysis(c, c->side);
    hipEventRecord(c->ev_join, c->side);
    c->cur = c->stream;
    const int T = enqueue_reduce_from(c, MUSICA_CNR_LEVEL + 1);
    for (int lvl = T - 1; lvl >= MUSICA_CNR_LEVEL; lvl--) {
        Span sp(c, MUSICA_KERNEL_EXPAND_REST);
        run_expand_level(c, lvl, c->rows_expand[lvl]);
    }
    hipStreamWaitEvent(c->stream, c->ev_join, 0);
    for (int lvl = MUSICA_CNR_LEVEL - 1; lvl >= 0; lvl--) {
        Span sp(c, lvl == 0 ? MUSICA_KERNEL_EXPAND_L0 : MUSICA_KERNEL_EXPAND_REST);
        run_expand_level(c, lvl, c->rows_expand[lvl]);
    }
    enqueue_gradation(c, true);
}
static void enqueue_script(musica_ctx* c) {
    c->sd_active = c->sd_fused;
    if (c->dag) enqueue_fork(c);
    else enqueue_linear(c);
    if (c->sd_active) c->sdev_stored = false;
    c->sd_active = false;
}

// Captures the dispatch script (with two streams the side stream joins the capture through ev_fork and rejoins
// through ev_join) into an executable graph for the current input pointer.
static int capture_graph(musica_ctx* c) {
    int k = 0;   // an empty slot, else the least recently used one
    for (int j = 0; j < kGraphSlots; j++) {
        if (!c->graph_exec[j]) { k = j; break; }
        if (c->graph_used[j] < c->graph_used[k]) k = j;
    }
    if (c->graph_exec[k]) {
        // its last launch may still be running (musica_execute_device / musica_pipeline_step never wait): drain the stream before
        // the executable graph goes away. Only callers that rotate more than kGraphSlots input pointers ever get here.
        if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        hipGraphExecDestroy(c->graph_exec[k]);
        c->graph_exec[k] = nullptr; c->graph_input[k] = nullptr;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
    enqueue_script(c);
    if (hipStreamEndCapture(c->stream, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); return -1; }
    const hipError_t e = hipGraphInstantiate(&c->graph_exec[k], graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { c->graph_exec[k] = nullptr; (void)hipGetLastError(); return -1; }
    c->graph_input[k] = c->cur_input;
    return k;
}

static void reset_tickets_if_needed(musica_ctx* c) {
    if (!c->needs_reset) return;
    hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    hipMemsetAsync(c->d_mm_ticket, 0, (size_t)c->B * kMinMaxStride * sizeof(uint32_t), c->stream);
    hipMemsetAsync(c->d_gr_ticket, 0, (size_t)c->B * kGradTicketStride * sizeof(uint32_t), c->stream);
    c->needs_reset = false;
}
static int enqueue_all_impl(musica_ctx* c);
static int enqueue_all(musica_ctx* c) {
    reset_tickets_if_needed(c);
    const int ok = enqueue_all_impl(c);
    if (!ok) c->needs_reset = true;
    return ok;
}
static int enqueue_all_impl(musica_ctx* c) {
    if (!c->tuning && c->use_graph && c->profiling == 0) {
        int k = -1;
        for (int j = 0; j < kGraphSlots; j++)
            if (c->graph_exec[j] && c->graph_input[j] == c->cur_input) { k = j; break; }
        if (k < 0) {
            k = capture_graph(c);
            if (k < 0) c->use_graph = false;   // e.g. a runtime without capture support: stay eager
        }
        if (k >= 0) {
            c->graph_used[k] = ++c->graph_clock;
            c->norm_valid = (c->d_clahe_hist != nullptr && !c->clahe_raw) || !c->fuse_u16;
            if (c->sd_fused) c->sdev_stored = false;
            if (hipGraphLaunch(c->graph_exec[k], c->stream) != hipSuccess) return fail("hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
            return 1;
        }
    }
    if (c->tuning) enqueue_linear(c);
    else enqueue_script(c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(e));
    return 1;
}

// ---- init-time autotune ------------------------------------------------------------------
// How many rows a wavefront marches decides both the number of wavefronts and the halo re-reads;
// the best value depends on level size, batch and how the workgroups spread over the 8 XCDs, and
// no closed form predicted it on MI355X ("Launch geometry"). So musica_create runs the
// pipeline once on a synthetic input and then times each streaming kernel of every large level for
// a handful of candidates with HIP events, keeping the fastest. Results are independent of the
// choice (the kernels are bit-identical for every row count); only the speed changes.
static float time_launches(musica_ctx* c, hipEvent_t a, hipEvent_t b, int reps, void (*fn)(musica_ctx*, int, int), int level, int rows) {
    fn(c, level, rows);  // warm
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a, c->stream);
        fn(c, level, rows);
        fn(c, level, rows);
        hipEventRecord(b, c->stream);
        hipEventSynchronize(b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms < best) best = ms;
    }
    return best;
}

static void autotune(musica_ctx* c) {
    const size_t n = (size_t)c->B * c->N * c->N;
    std::vector<uint16_t> px(n);
    uint32_t h = 12345u;
    for (int b = 0; b < c->B; b++)
        for (int y = 0; y < c->N; y++)
            for (int x = 0; x < c->N; x++) {
                h = h * 1664525u + 1013904223u;
                const float base = 20000.0f + 12000.0f * sinf(0.011f * x + b) * cosf(0.007f * y) + ((x / 97 + y / 61) & 1) * 6000.0f;
                const float noise = (float)((h >> 9) & 0xFFFF) / 65536.0f - 0.5f;
                px[((size_t)b * c->N + y) * c->N + x] = (uint16_t)fminf(65535.0f, fmaxf(1.0f, base + 2.0f * sqrtf(base) * noise));
            }
    if (hipMemcpy(c->d_input, px.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return;
    c->cur_input = c->d_input;
    c->cur = c->stream;
    if (!enqueue_all(c)) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    c->tuning = true;
    c->sd_active = c->sd_fused;   // the forms the whole-step scripts run
    const int reps = 3;
    static const int cand_pair[] = {2, 4, 8, 16};   // band / expand count coarse rows (two fine rows each)
    static const int cand_sdev[] = {0, 16, 32, 64};   // 0: one run per workgroup
    for (int i = 0; i < c->L; i++) {
        if (c->lv[i].S < 512 || (c->lv[i].S % 8) != 0) continue;   // small levels are launch-bound: keep the heuristic
        static const int cand_rb[] = {4, 8, 16, 32, 64};
        struct { int* slot; const int* cand; int ncand; void (*fn)(musica_ctx*, int, int); bool use; } jobs[3] = {
            {&c->rows_rb[i], cand_rb, 5, run_reduce_band, rb_level(c, i)},
            {&c->rows_expand[i], cand_pair, 4, run_expand_level, true},
            {i <= MUSICA_CNR_LEVEL ? &c->rows_sdev[i] : nullptr, cand_sdev, 4, run_sdev_level, i <= MUSICA_CNR_LEVEL},
        };
        for (auto& j : jobs) {
            if (!j.use) continue;
            float best = time_launches(c, a, b, reps, j.fn, i, *j.slot);
            for (int k = 0; k < j.ncand; k++) {
                if (j.cand[k] == *j.slot) continue;
                const float t = time_launches(c, a, b, reps, j.fn, i, j.cand[k]);
                if (t < best * 0.97f) { best = t; *j.slot = j.cand[k]; }   // 3 % hysteresis against timer noise
            }
        }
    }
    c->tuning = false;
    if (c->sd_active) c->sdev_stored = false;
    c->sd_active = false;
    hipEventDestroy(a);
    hipEventDestroy(b);
    hipStreamSynchronize(c->stream);
}

// pitched device plane -> dense host image
static int download_plane(musica_ctx* c, const float* d_plane, const LevelDesc& l, float* dst) {
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy2D(dst, (size_t)l.S * sizeof(float), d_plane, (size_t)l.pitch * sizeof(float), (size_t)l.S * sizeof(float), (size_t)l.S,
                       hipMemcpyDeviceToHost));
    return 1;
}
static int upload_plane(musica_ctx* c, float* d_plane, const LevelDesc& l, const float* src) {
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy2D(d_plane, (size_t)l.pitch * sizeof(float), src, (size_t)l.S * sizeof(float), (size_t)l.S * sizeof(float), (size_t)l.S,
                       hipMemcpyHostToDevice));
    return 1;
}

template <typename T>
static int download_small(musica_ctx* c, const T* d_src, T* dst, size_t count) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(dst, d_src, count * sizeof(T), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("device read-back failed: %s", hipGetErrorString(e));
    return 1;
}

#define CHECK_CTX(c) do { if (!(c)) return fail("%s: ctx is NULL", __func__); if (hipSetDevice((c)->p.device) != hipSuccess) return fail("%s: hipSetDevice failed", __func__); } while (0)
#define CHECK_IMG(c, idx) do { if ((int)(idx) >= (c)->B) return fail("%s: image_index %u >= batch %d", __func__, (unsigned)(idx), (c)->B); } while (0)

extern "C" {

// ---- image lanes (see musica_ctx::lanes) ----------------------------------------------------------------------------------
constexpr int kLaneStreams = 1;   // one: a one-image chain (0.14 - 0.19 ms) is as long as an image's copy, so chains on several streams would barely
                                  // overlap, and which hardware queue a further stream lands on (4 queues, round-robin over every stream of the
                                  // process) decided whether three lanes were faster or slower than none
// Shallow copy of the context restricted to image i0: same buffers, every per-image pointer moved to that image, batch 1, one stream,
// its script replayed as a graph of its own.
static musica_ctx* make_lane(const musica_ctx* c, int i0, int nb) {
    musica_ctx* v = new musica_ctx(*c);
    v->allocations.clear();   // the parent owns the memory, the streams and the events
    v->spans.clear();
    v->spans_used = 0;
    v->lanes.clear();
    v->img_copied.clear();
    v->side = nullptr; v->ev_fork = nullptr; v->ev_join = nullptr; v->copy_stream = nullptr;
    for (int k = 0; k < kGraphSlots; k++) { v->graph_exec[k] = nullptr; v->graph_input[k] = nullptr; v->graph_used[k] = 0; }
    v->use_graph = !(c->p.flags & MUSICA_FLAG_NO_GRAPH) && env_int("MUSICA_GRAPH", 1) != 0;   // one graph per lane and input buffer: the host
                                                                                             // enqueues 8 replays per batch instead of ~136 launches
    v->dag = 0;
    v->profiling = 0;
    v->B = nb;
    v->p.batch = (uint32_t)nb;
    const size_t o = (size_t)i0, NN = (size_t)c->N * c->N;
    v->d_input += o * NN;
    v->cur_input = v->d_input;
    v->d_minmax += o * kMinMaxStride;
    v->d_mm_slots += o * kMinMaxSlots;
    v->d_mm_ticket += o * kMinMaxStride;
    v->d_gr_ticket += o * kGradTicketStride;
    v->d_norm += o * c->lv[0].plane;
    for (int i = 0; i < c->L; i++) {
        v->d_down[i] += o * c->lv[i + 1].plane;
        v->d_band[i] += o * c->lv[i].plane;
        v->d_recon[i] += o * c->lv[i].plane;
        if (i <= MUSICA_CNR_LEVEL) v->d_sdev[i] += o * c->lv[i].plane;
    }
    v->d_noise_hist += o * 4 * MUSICA_NOISE_BINS;
    v->d_noise_max += o * c->L;
    v->d_curves += o * c->L;
    v->d_luts += o * MUSICA_COARSER_LEVELS_START;
    v->d_cnr += o * c->lv[MUSICA_CNR_LEVEL].plane;
    v->d_grad_hist += o * MUSICA_GRAD_BINS;
    v->d_grad_hist_b += o * MUSICA_GRAD_BINS;
    v->d_gzero += o;
    v->d_thr090 += o;
    v->d_stats_partial += o * kStatsMaxBlocks;
    if (v->d_le090) v->d_le090 += o * (size_t)c->lv[1].S * (c->lv[0].S / 8);
    v->d_grad_max += o;
    v->d_gcurve += o;
    v->d_graded += o * c->lv[0].plane;
    v->d_scratch += o * c->lv[0].plane;
    v->d_stats += o;
    if (c->d_clahe_hist) {
        const size_t tb = (size_t)MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
        v->d_clahe_hist += o * tb;
        v->d_clahe_pts += o * tb;
        v->d_clahe_graded += o * c->lv[0].plane;
    }
    for (int i = 0; i < c->L; i++) {   // launch geometry of a one-image step (results never depend on it)
        v->rows_expand[i] = pick_rows(c->expand_rows, 1, c->lv[i].S, c->lv[i + 1].S, nb);
        if (i <= MUSICA_CNR_LEVEL) v->rows_sdev[i] = sdev_rows_default(c, i, nb);
        v->rows_rb[i] = pick_rows(16, 1, c->lv[i].S, c->lv[i + 1].S, nb);
    }
    return v;
}
// Only for page-locked host memory (musica_host_alloc, hipHostMalloc, hipHostRegister): a copy from pageable memory is staged by the
// runtime and blocks the host meanwhile, so eight of them in a row are slower than one (8 x 2048^2: 2.5 ms against 1.7 ms per batch).
static bool lanes_wanted(const musica_ctx* c, const void* host_pixels) {
    if (!(c->B > 1 && c->profiling == 0 && !c->tuning && env_int("MUSICA_HOST_LANES", 1) != 0)) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host_pixels) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
static bool ensure_lanes(musica_ctx* c) {
    if (!c->lanes.empty()) return true;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++)
        if (!c->lane_copy[k]) ok = hipStreamCreateWithFlags(&c->lane_copy[k], hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < kLaneStreams && ok; k++) {
        if (!c->lane_stream[k]) ok = ok && hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking) == hipSuccess;
        if (!c->lane_done[k]) ok = ok && hipEventCreateWithFlags(&c->lane_done[k], hipEventDisableTiming) == hipSuccess;
    }
    if (ok && !c->lane_start) ok = hipEventCreateWithFlags(&c->lane_start, hipEventDisableTiming) == hipSuccess;
    while (ok && (int)c->img_copied.size() < c->B) {
        hipEvent_t e = nullptr;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) c->img_copied.push_back(e);
    }
    if (!ok) return false;
    // images per lane: the chain of a lane should be shorter than its copy, or the lanes queue up behind each other instead of behind the
    // link — one 2048^2 image: 0.17 - 0.19 ms of kernels against 0.155 ms of copy; two: 0.18 ms against 0.31 ms
    const int per = c->B >= 4 ? 2 : 1;
    for (int k = 0; k < c->B; k += per) {   // last: c->lanes non-empty means everything above exists
        musica_ctx* v = make_lane(c, k, c->B - k < per ? c->B - k : per);
        v->stream = c->lane_stream[(k / per) % kLaneStreams];
        v->cur = v->stream;
        c->lanes.push_back(v);
    }
    return true;
}
// One batch from host memory into `d_dst` (c->d_input or the streaming path's second buffer), image by image: copy k on the copy
// stream, image k's one-stream script behind it on lane stream k % 3; the context's own stream continues when every lane has finished.
// `after`: an event the copies must wait for (the buffer's previous reader), or null.
static int enqueue_images_from_host(musica_ctx* c, uint16_t* d_dst, const uint16_t* pixels, bool copies_wait = true) {
    const size_t NN = (size_t)c->N * c->N;
    reset_tickets_if_needed(c);
    // whatever the context's stream still runs (the previous step: it reads and writes the planes the lanes are about to overwrite) comes
    // first; the copies too unless the caller knows `d_dst` is free (the streaming path's double buffer)
    hipEventRecord(c->lane_start, c->stream);
    if (copies_wait) { hipStreamWaitEvent(c->lane_copy[0], c->lane_start, 0); hipStreamWaitEvent(c->lane_copy[1], c->lane_start, 0); }
    for (int k = 0; k < kLaneStreams; k++) hipStreamWaitEvent(c->lane_stream[k], c->lane_start, 0);
    // every copy first (the host takes ~100 us to enqueue one lane's replay: with copy k + 1 enqueued behind lane k the copy engine
    // idled between the images — 1.68 ms for the eight copies of 8 x 2048^2 instead of 1.24), then the lanes
    int first = 0;
    for (size_t l = 0; l < c->lanes.size(); l++) {
        const int nb = c->lanes[l]->B;
        if (hipMemcpyAsync(d_dst + (size_t)first * NN, pixels + (size_t)first * NN, (size_t)nb * NN * sizeof(uint16_t), hipMemcpyHostToDevice, c->lane_copy[0]) != hipSuccess)
            return fail("host-to-device copy of images %d.. failed: %s", first, hipGetErrorString(hipGetLastError()));
        hipEventRecord(c->img_copied[l], c->lane_copy[0]);
        first += nb;
    }
    first = 0;
    for (size_t l = 0; l < c->lanes.size(); l++) {
        musica_ctx* v = c->lanes[l];
        const int k = first;
        first += v->B;
        hipStreamWaitEvent(v->stream, c->img_copied[l], 0);
        v->cur_input = d_dst + (size_t)k * NN;
        v->cur = v->stream;
        if (!enqueue_all_impl(v)) { c->needs_reset = true; return 0; }
    }
    for (int k = 0; k < kLaneStreams; k++) {
        hipEventRecord(c->lane_done[k], c->lane_stream[k]);
        hipStreamWaitEvent(c->stream, c->lane_done[k], 0);
    }
    c->cur_input = d_dst;
    c->norm_valid = c->lanes[0]->norm_valid;
    if (c->sd_fused) c->sdev_stored = false;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->needs_reset = true; return fail("per-image dispatch failed: %s", hipGetErrorString(e)); }
    return 1;
}

int musica_sync(musica_ctx* c) {
    CHECK_CTX(c);
    {
        const hipError_t e_ = hipStreamSynchronize(c->stream);
        if (e_ != hipSuccess) { c->needs_reset = true; return fail("hipStreamSynchronize failed: %s", hipGetErrorString(e_)); }
    }
    if (c->profiling) collect_spans(c);
    return 1;
}

int musica_upload(musica_ctx* c, const uint16_t* pixels) {
    CHECK_CTX(c);
    if (!pixels) return fail("musica_upload: pixels is NULL");
    HIP_OK(hipMemcpyAsync(c->d_input, pixels, (size_t)c->B * c->N * c->N * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 1;
}

const uint16_t* musica_input_device_ptr(musica_ctx* c) { return c ? c->d_input : nullptr; }

int musica_execute_device(musica_ctx* c, const uint16_t* d_pixels) {
    CHECK_CTX(c);
    if (!d_pixels) return fail("musica_execute_device: d_pixels is NULL");
    if (((uintptr_t)d_pixels & 15u) != 0) return fail("musica_execute_device: d_pixels must be 16-byte aligned");
    c->cur_input = d_pixels;
    return enqueue_all(c);
}

int musica_execute(musica_ctx* c, const uint16_t* pixels) {
    ABI_TRY
    CHECK_CTX(c);
    if (!pixels) return fail("musica_execute: pixels is NULL");
    if (lanes_wanted(c, pixels)) {   // a batch in pinned memory: image k's chain starts when image k has landed (the copy of a batch takes 2 - 3 x its kernels)
        if (!ensure_lanes(c)) return fail("musica_execute: stream / event creation for the image lanes failed");
        if (!enqueue_images_from_host(c, c->d_input, pixels, true)) return 0;
        return musica_sync(c);
    }
    HIP_OK(hipMemcpyAsync(c->d_input, pixels, (size_t)c->B * c->N * c->N * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));  // vk_state.cpp:313-342
    c->cur_input = c->d_input;
    if (!enqueue_all(c)) return 0;
    return musica_sync(c);  // vkWaitForFences, src/vk_processing.cpp:2535-2536
    ABI_CATCH("musica_execute")
}

// The reference uploads every image through a freshly allocated staging buffer and three queue-idle waits before a single
// dispatch starts (VulkanState::loadDataToImage, src/vk_state.cpp:313-342). A sequence of batches is pipelined instead: two
// device input buffers, host-to-device copies on a stream of their own, so the copy of batch j + 1 runs under the kernels of
// batch j and the host only blocks at the very end. Input in pinned memory (musica_host_alloc) moves at the PCIe rate;
// pageable memory works too (the runtime stages it, slower and partly synchronous).
int musica_execute_stream(musica_ctx* c, const uint16_t* const* pixels, uint32_t count, musica_stats* stats) {
    ABI_TRY
    CHECK_CTX(c);
    if (!pixels) return fail("musica_execute_stream: pixels is NULL");
    const size_t bytes = (size_t)c->B * c->N * c->N * sizeof(uint16_t);
    if (!c->ev_consumed[1]) {   // keyed on the LAST resource of the block: a call that failed half-way is retried, never half-initialised
        if (!c->d_input2 && !dalloc(c, &c->d_input2, (size_t)c->B * c->N * c->N)) return fail("musica_execute_stream: device allocation failed");
        if (!c->copy_stream) HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            if (!c->ev_copied[k]) HIP_OK(hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming));
            if (!c->ev_consumed[k]) HIP_OK(hipEventCreateWithFlags(&c->ev_consumed[k], hipEventDisableTiming));
        }
    }
    if (count == 0) return 1;
    // (Whole batches: the copy of batch j + 1 under the kernels of batch j already moves the pixels at the rate the link delivers — 53 - 54 GB/s
    // here. The image lanes of musica_execute were measured in this loop too: 20 - 22 GP/s against 26.5, the host needs ~0.8 ms per batch
    // to enqueue eight replays and the next batch's copies wait behind that.)
    musica_stats* d_rows = nullptr;
    musica_stats* h_rows = nullptr;
    struct RowsGuard {   // the two scratch buffers go away on every exit path
        musica_stats*& d; musica_stats*& h;
        ~RowsGuard() { if (d) hipFree(d); if (h) hipHostFree(h); }
    } rows_guard{d_rows, h_rows};
    if (stats) {
        HIP_OK(hipMalloc(&d_rows, (size_t)count * c->B * sizeof(musica_stats)));
        if (hipHostMalloc(&h_rows, (size_t)count * c->B * sizeof(musica_stats), hipHostMallocDefault) != hipSuccess) { h_rows = nullptr; return fail("musica_execute_stream: pinned allocation failed"); }
    }
    HIP_OK(hipStreamSynchronize(c->stream));   // nothing of an earlier call still reads the input buffers
    uint16_t* bufs[2] = {c->d_input, c->d_input2};
    int ok = 1;
    for (uint32_t j = 0; j < count && ok; j++) {
        const int k = (int)(j & 1u);
        if (!pixels[j]) { ok = fail("musica_execute_stream: pixels[%u] is NULL", j); break; }
        // batch j - 2 has been computed: its buffer is free. Waited for on the HOST: a copy-engine queue that waits for a compute
        // queue's event stalls for ~1 ms at a time on this part (one 2048^2 image per batch: 1.3 ms per batch with the wait on the copy
        // stream against 0.2 here; devtools/stream_probe.py), and the host has nothing else to do before it may enqueue this copy
        if (j >= 2 && hipEventSynchronize(c->ev_consumed[k]) != hipSuccess) { ok = fail("musica_execute_stream: waiting for batch %u failed", j - 2); break; }
        if (hipMemcpyAsync(bufs[k], pixels[j], bytes, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) { ok = fail("musica_execute_stream: H2D copy failed"); break; }
        hipEventRecord(c->ev_copied[k], c->copy_stream);
        hipStreamWaitEvent(c->stream, c->ev_copied[k], 0);
        c->cur_input = bufs[k];
        ok = enqueue_all(c);
        if (ok && stats) {
            launch_stats(c->stream, c->d_cnr, c->lv[MUSICA_CNR_LEVEL], c->d_minmax, c->min_chain_exact, c->d_noise_max, c->L, c->d_grad_max, c->d_gcurve,
                         d_rows + (size_t)j * c->B, j * (uint32_t)c->B, 1u, c->B, c->d_stats_partial);
        }
        hipEventRecord(c->ev_consumed[k], c->stream);
    }
    if (ok && stats) ok = hipMemcpyAsync(h_rows, d_rows, (size_t)count * c->B * sizeof(musica_stats), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    const hipError_t e = hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->copy_stream);
    if (ok && e == hipSuccess && stats) memcpy(stats, h_rows, (size_t)count * c->B * sizeof(musica_stats));
    if (e != hipSuccess) return fail("musica_execute_stream: %s", hipGetErrorString(e));
    if (c->profiling) collect_spans(c);
    return ok;
    ABI_CATCH("musica_execute_stream")
}

void* musica_host_alloc(musica_ctx* c, size_t bytes) {
    if (!c || hipSetDevice(c->p.device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail("musica_host_alloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void musica_host_free(musica_ctx* c, void* p) {
    if (!c || !p) return;
    hipSetDevice(c->p.device);
    hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    hipHostFree(p);
}

int musica_debug_run_stage(musica_ctx* c, musica_stage stage) {
    CHECK_CTX(c);
    c->cur = c->stream;
    switch (stage) {
        case MUSICA_STAGE_NORM:
            enqueue_norm(c, false);
            break;
        case MUSICA_STAGE_REDUCE: enqueue_reduce(c); break;
        case MUSICA_STAGE_ANALYSIS:
            launch_clear(c->stream, nullptr, c->d_noise_hist, nullptr, nullptr, c->B);
            enqueue_analysis(c);
            c->sdev_stored = true;   // the stage stores every level's sdev image (sd_active is a property of the whole-step scripts)
            break;
        case MUSICA_STAGE_EXPAND: ensure_sdev(c); enqueue_expand(c, false); break;
        case MUSICA_STAGE_GRADATION:
            launch_clear(c->stream, nullptr, nullptr, c->d_grad_hist, c->d_clahe_hist, c->B, c->d_grad_hist_b, c->d_gzero);
            enqueue_gradation(c, false);
            break;
        default: return fail("musica_debug_run_stage: unknown stage %d", (int)stage);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(e));
    return musica_sync(c);
}

uint32_t musica_image_side(const musica_ctx* c, musica_image_kind kind, uint32_t level) {
    if (!c) return 0;
    switch (kind) {
        case MUSICA_IMG_NORMALIZED: case MUSICA_IMG_GRADED: case MUSICA_IMG_RELEVANT: case MUSICA_IMG_SQRT: case MUSICA_IMG_CLAHE_GRADED:
            return level == 0 ? (uint32_t)c->N : 0;
        case MUSICA_IMG_CNR: return level == MUSICA_CNR_LEVEL ? (uint32_t)c->lv[MUSICA_CNR_LEVEL].S : 0;
        case MUSICA_IMG_DOWNSAMPLED: return (int)level < c->L ? (uint32_t)c->lv[level + 1].S : 0;
        case MUSICA_IMG_BANDPASS: case MUSICA_IMG_SDEV: case MUSICA_IMG_EXPAND: case MUSICA_IMG_LOWPASS: case MUSICA_IMG_EXP_BANDPASS: case MUSICA_IMG_CONTRAST_BAND:
            return (int)level < c->L ? (uint32_t)c->lv[level].S : 0;
        default: return 0;
    }
}

// Resolves (kind, level) to a device plane of image `idx`; on-demand kinds are computed into d_scratch.
static int resolve_image(musica_ctx* c, uint32_t idx, musica_image_kind kind, uint32_t level, const float** plane, const LevelDesc** desc) {
    if (musica_image_side(c, kind, level) == 0) return fail("musica_get_image: no image kind=%d level=%u", (int)kind, level);
    const LevelDesc& l0 = c->lv[0];
    const LevelDesc& l3 = c->lv[MUSICA_CNR_LEVEL];
    switch (kind) {
        case MUSICA_IMG_NORMALIZED: ensure_normalized(c); *plane = c->d_norm + idx * l0.plane; *desc = &c->lv[0]; return 1;
        case MUSICA_IMG_GRADED: *plane = c->d_graded + idx * l0.plane; *desc = &c->lv[0]; return 1;
        case MUSICA_IMG_CLAHE_GRADED:
            if (!c->d_clahe_graded) return fail("musica_get_image: ctx was created without MUSICA_FLAG_CLAHE");
            *plane = c->d_clahe_graded + idx * l0.plane; *desc = &c->lv[0]; return 1;
        case MUSICA_IMG_CNR: *plane = c->d_cnr + idx * l3.plane; *desc = &c->lv[MUSICA_CNR_LEVEL]; return 1;
        case MUSICA_IMG_DOWNSAMPLED: *plane = c->d_down[level] + idx * c->lv[level + 1].plane; *desc = &c->lv[level + 1]; return 1;
        case MUSICA_IMG_BANDPASS: *plane = c->d_band[level] + idx * c->lv[level].plane; *desc = &c->lv[level]; return 1;
        case MUSICA_IMG_EXPAND: *plane = c->d_recon[level] + idx * c->lv[level].plane; *desc = &c->lv[level]; return 1;
        case MUSICA_IMG_SDEV:
            *desc = &c->lv[level];
            ensure_sdev(c);
            if (level <= MUSICA_CNR_LEVEL) { *plane = c->d_sdev[level] + idx * c->lv[level].plane; return 1; }
            // levels >= 4: the reference never writes these images (src/vk_processing.cpp:2285) -> zeros (Q2)
            HIP_OK(hipMemsetAsync(c->d_scratch, 0, c->lv[level].plane * sizeof(float), c->stream));
            *plane = c->d_scratch; return 1;
        case MUSICA_IMG_RELEVANT:
            ensure_normalized(c);
            launch_relevant(c->stream, c->d_norm, c->d_cnr, c->d_scratch, l0, l3, (int)cnr_scale(l0.S, l3.S), c->B);
            *plane = c->d_scratch + idx * l0.plane; *desc = &c->lv[0]; return 1;
        case MUSICA_IMG_SQRT:
            launch_sqrt(c->stream, c->cur_input, c->d_scratch, l0, c->B);
            *plane = c->d_scratch + idx * l0.plane; *desc = &c->lv[0]; return 1;
        case MUSICA_IMG_LOWPASS:  // lowpassImageStates[level] = smooth_upsampled(upsample(downsampled[level]))
            launch_lowpass(c->stream, c->d_down[level], c->d_scratch, c->lv[level], c->lv[level + 1], c->B, c->ref_order);
            *plane = c->d_scratch + idx * c->lv[level].plane; *desc = &c->lv[level]; return 1;
        case MUSICA_IMG_EXP_BANDPASS: case MUSICA_IMG_CONTRAST_BAND: {
            ensure_sdev(c);
            const ExpandArgs a = expand_args(c, (int)level, c->d_scratch);
            launch_exp_band(c->stream, a, gain_mode((int)level), kind == MUSICA_IMG_EXP_BANDPASS && uses_nr((int)level), c->B);
            *plane = c->d_scratch + idx * c->lv[level].plane; *desc = &c->lv[level]; return 1;
        }
        default: return fail("musica_get_image: unsupported kind %d", (int)kind);
    }
}

int musica_get_image(musica_ctx* c, uint32_t idx, musica_image_kind kind, uint32_t level, float* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!dst) return fail("musica_get_image: dst is NULL");
    const float* plane = nullptr;
    const LevelDesc* d = nullptr;
    if (!resolve_image(c, idx, kind, level, &plane, &d)) return 0;
    return download_plane(c, plane, *d, dst);
}

int musica_debug_set_image(musica_ctx* c, uint32_t idx, musica_image_kind kind, uint32_t level, const float* src) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!src) return fail("musica_debug_set_image: src is NULL");
    if (musica_image_side(c, kind, level) == 0) return fail("musica_debug_set_image: no image kind=%d level=%u", (int)kind, level);
    switch (kind) {
        case MUSICA_IMG_NORMALIZED:
            if (c->fuse_u16) return fail("musica_debug_set_image: the normalized image is not stored on the hot path (level 0 reads the raw pixels)");
            return upload_plane(c, c->d_norm + idx * c->lv[0].plane, c->lv[0], src);
        case MUSICA_IMG_GRADED: return upload_plane(c, c->d_graded + idx * c->lv[0].plane, c->lv[0], src);
        case MUSICA_IMG_CNR: return upload_plane(c, c->d_cnr + idx * c->lv[MUSICA_CNR_LEVEL].plane, c->lv[MUSICA_CNR_LEVEL], src);
        case MUSICA_IMG_DOWNSAMPLED: return upload_plane(c, c->d_down[level] + idx * c->lv[level + 1].plane, c->lv[level + 1], src);
        case MUSICA_IMG_BANDPASS: return upload_plane(c, c->d_band[level] + idx * c->lv[level].plane, c->lv[level], src);
        case MUSICA_IMG_EXPAND: return upload_plane(c, c->d_recon[level] + idx * c->lv[level].plane, c->lv[level], src);
        case MUSICA_IMG_SDEV:
            if (level > MUSICA_CNR_LEVEL) return fail("musica_debug_set_image: sdev exists only for levels 0..3");
            ensure_sdev(c);   // the other levels' stored images first: a later getter must not overwrite what the caller sets here
            return upload_plane(c, c->d_sdev[level] + idx * c->lv[level].plane, c->lv[level], src);
        default: return fail("musica_debug_set_image: kind %d is not stored on the hot path", (int)kind);
    }
}

int musica_get_graded(musica_ctx* c, float* dst) {
    CHECK_CTX(c);
    if (!dst) return fail("musica_get_graded: dst is NULL");
    for (int b = 0; b < c->B; b++)
        if (!download_plane(c, c->d_graded + (size_t)b * c->lv[0].plane, c->lv[0], dst + (size_t)b * c->N * c->N)) return 0;
    return 1;
}

// saveOutImage: crop margin 10, (uint8_t)(255.0f * (v - 0) / (1 - 0)) (src/vk_processing.cpp:2624-2634) — on the device
// (k_out_pixels), then (N - 20)^2 bytes into pinned host memory: 1 B per output pixel instead of the reference's 4 B per input
// pixel through pageable memory (loadDataFromImage, src/vk_state.cpp:777-803). Returns the pinned buffer (valid until the next call).
static const uint8_t* out_pixels_pinned(musica_ctx* c, uint32_t idx) {
    const uint32_t N = (uint32_t)c->N, margin = MUSICA_OUT_MARGIN;
    if (N <= 2 * margin) { fail("saveOutImage: image too small for the %u-pixel margin", margin); return nullptr; }
    const size_t nw = N - 2 * margin, bytes = nw * nw;
    Tick tick("save");
    if (!c->h_out8) {   // keyed on the LAST resource of the block: a call that failed half-way is retried, never half-initialised
        if (!c->d_out8 && !dalloc(c, &c->d_out8, bytes)) { fail("saveOutImage: device allocation failed"); return nullptr; }
        if (hipHostMalloc((void**)&c->h_out8, bytes, hipHostMallocDefault) != hipSuccess) { c->h_out8 = nullptr; fail("saveOutImage: pinned allocation failed"); return nullptr; }
    }
    tick.lap("device + pinned buffers");
    launch_out_pixels(c->stream, c->d_graded + (size_t)idx * c->lv[0].plane, c->lv[0], (int)margin, c->d_out8);
    hipError_t e = hipMemcpyAsync(c->h_out8, c->d_out8, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { fail("saveOutImage: read-back failed: %s", hipGetErrorString(e)); return nullptr; }
    tick.lap("crop + quantise kernel, read-back");
    return c->h_out8;
}

int musica_get_out_pixels(musica_ctx* c, uint32_t idx, uint8_t* dst) {
    ABI_TRY
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!dst) return fail("musica_get_out_pixels: dst is NULL");
    const uint8_t* px = out_pixels_pinned(c, idx);
    if (!px) return 0;
    const size_t nw = (size_t)c->N - 2 * MUSICA_OUT_MARGIN;
    memcpy(dst, px, nw * nw);
    return 1;
    ABI_CATCH("musica_get_out_pixels")
}

int musica_save_out_image(musica_ctx* c, uint32_t idx, const char* path) {
    ABI_TRY
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!path) return fail("musica_save_out_image: path is NULL");
    const uint32_t N = (uint32_t)c->N, margin = MUSICA_OUT_MARGIN;
    if (N <= 2 * margin) return fail("saveOutImage: image too small for the %u-pixel margin", margin);
    const uint32_t nw = N - 2 * margin;
    // The file image is built where it is written from: header by the host, pixel array (24 bpp, bottom-up, padded rows: stbi_write_bmp's
    // layout) by the device straight into page-locked memory, then ONE write — instead of 1 byte per pixel read back, expanded to 3 bytes row
    // by row on the host and written through stdio (26 - 29 ms -> the sum below at 3072^2). MUSICA_SAVE_ON_DEVICE=0: the former path.
    Tick tick("save");
    if (env_int("MUSICA_SAVE_ON_DEVICE", 1) != 0) {
        uint8_t hdr[54];
        const size_t row_bytes = musica_bmp24_header(nw, nw, hdr), file_bytes = 54 + row_bytes * nw;
        if (!c->h_bmp && hipHostMalloc((void**)&c->h_bmp, 2 + file_bytes, hipHostMallocMapped) != hipSuccess) { c->h_bmp = nullptr; (void)hipGetLastError(); }
        void* dev = nullptr;
        if (c->h_bmp && hipHostGetDevicePointer(&dev, c->h_bmp, 0) == hipSuccess && dev) {
            tick.lap("page-locked file image");
            memcpy(c->h_bmp + 2, hdr, 54);
            launch_out_bmp24(c->stream, c->d_graded + (size_t)idx * c->lv[0].plane, c->lv[0], (int)margin, reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(dev) + 56));
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return fail("saveOutImage: pixel kernel failed: %s", hipGetErrorString(e));
            tick.lap("crop + quantise + 24-bpp rows over the link");
            if (!musica_write_file(path, c->h_bmp + 2, file_bytes)) return fail("failed to write out file");  // :2636-2642
            tick.lap("bmp file");
            return 1;
        }
        (void)hipGetLastError();
    }
    const uint8_t* px = out_pixels_pinned(c, idx);
    if (!px) return 0;
    if (!musica_write_bmp_gray(path, nw, nw, px)) return fail("failed to write out file");  // :2636-2642
    tick.lap("bmp file");
    return 1;
    ABI_CATCH("musica_save_out_image")
}

int musica_get_noise_hist(musica_ctx* c, uint32_t idx, uint32_t level, uint32_t* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if ((int)level >= c->L) return fail("musica_get_noise_hist: level %u >= %d", level, c->L);
    if (level > MUSICA_CNR_LEVEL) { memset(dst, 0, MUSICA_NOISE_BINS * sizeof(uint32_t)); return 1; }  // cleared, never filled
    return download_small(c, c->d_noise_hist + ((size_t)idx * 4 + level) * MUSICA_NOISE_BINS, dst, MUSICA_NOISE_BINS);
}
int musica_get_grad_hist(musica_ctx* c, uint32_t idx, uint32_t* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    return download_small(c, c->d_grad_hist + (size_t)idx * MUSICA_GRAD_BINS, dst, MUSICA_GRAD_BINS);
}
int musica_get_noise_hist_max(musica_ctx* c, uint32_t idx, uint32_t level, musica_hist_max_point* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if ((int)level >= c->L) return fail("musica_get_noise_hist_max: level %u >= %d", level, c->L);
    return download_small(c, c->d_noise_max + (size_t)idx * c->L + level, dst, 1);
}
int musica_get_grad_hist_max(musica_ctx* c, uint32_t idx, musica_hist_max_point* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    return download_small(c, c->d_grad_max + idx, dst, 1);
}
int musica_get_contrast_curve(musica_ctx* c, uint32_t idx, uint32_t level, musica_contrast_curve* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if ((int)level >= c->L) return fail("musica_get_contrast_curve: level %u >= %d", level, c->L);
    DevCurve dc;
    if (!download_small(c, c->d_curves + (size_t)idx * c->L + level, &dc, 1)) return 0;
    memset(dst, 0, sizeof(*dst));
    for (uint32_t i = 0; i < dc.count && i < MUSICA_MAX_POINTS; i++) { dst->points[i].x = dc.x[i]; dst->points[i].y = dc.y[i]; }
    dst->pointsCount = dc.count;
    return 1;
}
int musica_get_grad_curve(musica_ctx* c, uint32_t idx, musica_grad_curve* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    DevCurve dc;
    if (!download_small(c, c->d_gcurve + idx, &dc, 1)) return 0;
    memset(dst, 0, sizeof(*dst));
    for (uint32_t i = 0; i < dc.count && i < MUSICA_MAX_POINTS; i++) { dst->points[i].x = dc.x[i]; dst->points[i].y = dc.y[i]; }
    dst->pointsCount = dc.count;
    dst->t0 = dc.t0; dst->ta = dc.ta; dst->t1 = dc.t1;
    return 1;
}
// The RENDER_HISTS plots (kernels_gradation.hip k_render_*): rendered into d_plot on the ctx stream, copied out.
static int plot_out(musica_ctx* c, uint8_t* rgba) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rgba, c->d_plot, (size_t)MUSICA_HIST_RENDER_WIDTH * MUSICA_HIST_RENDER_HEIGHT * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("plot read-back failed: %s", hipGetErrorString(e));
    return 1;
}
int musica_render_noise_hist(musica_ctx* c, uint32_t idx, uint8_t* rgba) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!rgba) return fail("musica_render_noise_hist: rgba is NULL");
    launch_render_noise_hist(c->stream, c->d_noise_hist + ((size_t)idx * 4 + MUSICA_CNR_LEVEL) * MUSICA_NOISE_BINS,
                             c->d_noise_max + (size_t)idx * c->L + MUSICA_CNR_LEVEL, c->d_plot);   // src/vk_processing.cpp:1260-1266
    return plot_out(c, rgba);
}
int musica_render_grad_hist(musica_ctx* c, uint32_t idx, uint8_t* rgba) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!rgba) return fail("musica_render_grad_hist: rgba is NULL");
    launch_render_grad_hist(c->stream, c->d_grad_hist + (size_t)idx * MUSICA_GRAD_BINS, c->d_grad_max + idx, c->d_gcurve + idx, c->d_plot);   // :1668-1675
    return plot_out(c, rgba);
}
int musica_get_contrast_params(musica_ctx* c, uint32_t level, musica_contrast_params* dst) {
    if (!c || (int)level >= c->L) return fail("musica_get_contrast_params: bad ctx/level");
    *dst = c->h_cparams[level];
    return 1;
}
int musica_get_nr_params(musica_ctx* c, uint32_t level, musica_nr_params* dst) {
    if (!c || level >= 3) return fail("musica_get_nr_params: bad ctx/level");
    *dst = c->h_nr[level];
    return 1;
}
int musica_get_minmax(musica_ctx* c, uint32_t idx, float* min_sqrt, float* max_sqrt) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    uint32_t mm[2];
    if (!download_small(c, c->d_minmax + kMinMaxStride * (size_t)idx, &mm[0], 1)) return 0;
    if (!download_small(c, c->d_minmax + kMinMaxStride * (size_t)idx + kMaxWord, &mm[1], 1)) return 0;
    // same scalars the normalize kernel derives (kernels_analysis.hip chain_scalars)
    const float mx = sqrtf((float)mm[1]), mn = sqrtf((float)mm[0]);
    *max_sqrt = (float)(uint32_t)mx;
    *min_sqrt = c->min_chain_exact ? (float)(uint32_t)mn : 0.0f;
    return 1;
}
int musica_stats_device_strided(musica_ctx* c, void* d_dst, uint32_t image_id_base, uint32_t image_id_stride) {
    CHECK_CTX(c);
    if (!d_dst) return fail("musica_stats_device: d_dst is NULL");
    launch_stats(c->stream, c->d_cnr, c->lv[MUSICA_CNR_LEVEL], c->d_minmax, c->min_chain_exact, c->d_noise_max, c->L, c->d_grad_max,
                 c->d_gcurve, (musica_stats*)d_dst, image_id_base, image_id_stride, c->B, c->d_stats_partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(e));
    return 1;
}
int musica_stats_device(musica_ctx* c, void* d_dst, uint32_t image_id_base) { return musica_stats_device_strided(c, d_dst, image_id_base, 1u); }
int musica_get_stats(musica_ctx* c, uint32_t idx, musica_stats* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!musica_stats_device(c, c->d_stats, 0)) return 0;
    return download_small(c, c->d_stats + idx, dst, 1);
}
int musica_get_clahe_hist(musica_ctx* c, uint32_t idx, uint32_t* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!c->d_clahe_hist) return fail("musica_get_clahe_hist: ctx was created without MUSICA_FLAG_CLAHE");
    const size_t tb = MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
    return download_small(c, c->d_clahe_hist + idx * tb, dst, tb);
}
int musica_get_clahe_curves(musica_ctx* c, uint32_t idx, musica_point* dst) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    if (!c->d_clahe_pts) return fail("musica_get_clahe_curves: ctx was created without MUSICA_FLAG_CLAHE");
    const size_t tb = MUSICA_CLAHE_TILES * MUSICA_CLAHE_TILES * MUSICA_CLAHE_BINS;
    return download_small(c, c->d_clahe_pts + idx * tb, dst, tb);
}

// VulkanState::downloadAndSaveImage (src/vk_state.cpp:809-855): (uint8_t)(255 * (v - min) / (max - min)), full image.
static int dump_image(musica_ctx* c, uint32_t idx, musica_image_kind kind, uint32_t level, const std::string& path, float maxValue, float minValue) {
    const uint32_t side = musica_image_side(c, kind, level);
    std::vector<float> img((size_t)side * side);
    if (!musica_get_image(c, idx, kind, level, img.data())) return 0;
    std::vector<uint8_t> out((size_t)side * side);
    for (size_t i = 0; i < out.size(); i++) {
        const float q = 255.0f * (img[i] - minValue) / (maxValue - minValue);
        // the reference's C cast is undefined outside [0, 256); wrap like the common x86 lowering (cvttss2si + truncate)
        out[i] = (q == q && q > -2147483648.0f && q < 2147483648.0f) ? (uint8_t)(int32_t)q : 0;
    }
    if (!musica_write_bmp_gray(path.c_str(), side, side, out.data())) return fail("failed to write %s", path.c_str());
    return 1;
}

int musica_debug_process(musica_ctx* c, uint32_t idx, const char* dir) {
    CHECK_CTX(c); CHECK_IMG(c, idx);
    ABI_TRY
    const std::string d = std::string(dir && *dir ? dir : ".") + "/";
    if (!dump_image(c, idx, MUSICA_IMG_NORMALIZED, 0, d + "norm.bmp", 1.0f, 0.0f)) return 0;               // :2664-2671
    for (int i = 0; i < c->L; i++) {                                                                       // :2673-2690
        if (!dump_image(c, idx, MUSICA_IMG_BANDPASS, i, d + "red_bandpass_" + std::to_string(i) + ".bmp", 1.0f, -1.0f)) return 0;
        if (!dump_image(c, idx, MUSICA_IMG_LOWPASS, i, d + "red_lowpass_" + std::to_string(i) + ".bmp", 1.0f, 0.0f)) return 0;
    }
    if (!dump_image(c, idx, MUSICA_IMG_SDEV, MUSICA_CNR_LEVEL, d + "sdev.bmp", 1.0f, -1.0f)) return 0;      // :2692-2699
    if (!dump_image(c, idx, MUSICA_IMG_CNR, MUSICA_CNR_LEVEL, d + "cnr.bmp", 1.0f, 0.0f)) return 0;         // :2701-2708
    for (int i = 0; i < c->L; i++) {                                                                       // :2710-2727 (slot i <-> level L-1-i)
        const int lvl = c->L - 1 - i;
        // expandBandpassImageStates[i]: the contrast-curve output, before noise reduction (src/vk_processing.cpp:1100)
        if (!dump_image(c, idx, MUSICA_IMG_CONTRAST_BAND, lvl, d + "exp_bandpass_" + std::to_string(i) + ".bmp", 1.0f, -1.0f)) return 0;
        // expandLowpassImageStates[i] = smooth_upsampled(upsample(previous reconstruction))
        const LevelDesc& lf = c->lv[lvl];
        const float* prev = lvl == c->L - 1 ? c->d_down[c->L - 1] : c->d_recon[lvl + 1];
        launch_lowpass(c->stream, prev, c->d_scratch, lf, c->lv[lvl + 1], c->B, c->ref_order);
        std::vector<float> img((size_t)lf.S * lf.S);
        if (!download_plane(c, c->d_scratch + (size_t)idx * lf.plane, lf, img.data())) return 0;
        std::vector<uint8_t> out(img.size());
        for (size_t k = 0; k < out.size(); k++) {
            const float q = 255.0f * img[k];
            out[k] = (q == q && q > -2147483648.0f && q < 2147483648.0f) ? (uint8_t)(int32_t)q : 0;
        }
        if (!musica_write_bmp_gray((d + "exp_lowpass_" + std::to_string(i) + ".bmp").c_str(), lf.S, lf.S, out.data())) return fail("failed to write exp_lowpass");
    }
    if (!dump_image(c, idx, MUSICA_IMG_RELEVANT, 0, d + "relevant.bmp", 1.0f, 0.0f)) return 0;              // :2729-2736
    if (!dump_image(c, idx, MUSICA_IMG_GRADED, 0, d + "graded.bmp", 1.0f, 0.0f)) return 0;                  // :2749-2756
    {   // the two RGBA plots, :2758-2806
        std::vector<uint8_t> plot((size_t)MUSICA_HIST_RENDER_WIDTH * MUSICA_HIST_RENDER_HEIGHT * 4);
        if (!musica_render_noise_hist(c, idx, plot.data())) return 0;
        if (!musica_write_bmp_rgba((d + "noise_hist.bmp").c_str(), MUSICA_HIST_RENDER_WIDTH, MUSICA_HIST_RENDER_HEIGHT, plot.data())) return fail("failed to write noise_hist.bmp");
        if (!musica_render_grad_hist(c, idx, plot.data())) return 0;
        if (!musica_write_bmp_rgba((d + "grad_hist.bmp").c_str(), MUSICA_HIST_RENDER_WIDTH, MUSICA_HIST_RENDER_HEIGHT, plot.data())) return fail("failed to write grad_hist.bmp");
    }
    // and the numbers behind them as CSV (an addition)
    {
        std::vector<uint32_t> h(MUSICA_NOISE_BINS);
        FILE* f = fopen((d + "noise_hist.csv").c_str(), "w");
        if (!f) return fail("failed to write noise_hist.csv");
        fprintf(f, "bin,level0,level1,level2,level3\n");
        std::vector<uint32_t> all(4 * MUSICA_NOISE_BINS);
        for (uint32_t l = 0; l < 4; l++)
            if (!musica_get_noise_hist(c, idx, l, all.data() + l * MUSICA_NOISE_BINS)) { fclose(f); return 0; }
        for (int b = 0; b < MUSICA_NOISE_BINS; b++)
            fprintf(f, "%d,%u,%u,%u,%u\n", b, all[b], all[MUSICA_NOISE_BINS + b], all[2 * MUSICA_NOISE_BINS + b], all[3 * MUSICA_NOISE_BINS + b]);
        fclose(f);
    }
    {
        std::vector<uint32_t> h(MUSICA_GRAD_BINS);
        if (!musica_get_grad_hist(c, idx, h.data())) return 0;
        musica_grad_curve gc;
        if (!musica_get_grad_curve(c, idx, &gc)) return 0;
        FILE* f = fopen((d + "grad_hist.csv").c_str(), "w");
        if (!f) return fail("failed to write grad_hist.csv");
        fprintf(f, "bin,weight\n");
        for (int b = 0; b < MUSICA_GRAD_BINS; b++) fprintf(f, "%d,%u\n", b, h[b]);
        fclose(f);
        f = fopen((d + "grad_curve.csv").c_str(), "w");
        if (!f) return fail("failed to write grad_curve.csv");
        fprintf(f, "# t0=%.9g ta=%.9g t1=%.9g\nx,y\n", gc.t0, gc.ta, gc.t1);
        for (uint32_t i = 0; i < gc.pointsCount; i++) fprintf(f, "%.9g,%.9g\n", gc.points[i].x, gc.points[i].y);
        fclose(f);
    }
    return 1;
    ABI_CATCH("musica_debug_process")
}

// ---- profiling ------------------------------------------------------------------------
int musica_profile_enable(musica_ctx* c, int enabled) {
    CHECK_CTX(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    collect_spans(c);
    c->profiling = enabled < 0 ? 0xFFFFFFFFu : (uint32_t)enabled;  // < 0: every kernel family; else bit i = musica_kernel_id i
    return 1;
}
int musica_profile_reset(musica_ctx* c) {
    CHECK_CTX(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    c->spans_used = 0;
    memset(c->prof_total_us, 0, sizeof(c->prof_total_us));
    memset(c->prof_count, 0, sizeof(c->prof_count));
    return 1;
}
int musica_profile_get(musica_ctx* c, musica_kernel_id id, double* mean_us, uint64_t* launches) {
    CHECK_CTX(c);
    if ((int)id < 0 || id >= MUSICA_KERNEL_COUNT) return fail("musica_profile_get: bad kernel id %d", (int)id);
    if (mean_us) *mean_us = c->prof_count[id] ? c->prof_total_us[id] / (double)c->prof_count[id] : 0.0;
    if (launches) *launches = c->prof_count[id];
    return 1;
}

// ---- stand-alone metric kernel ----------------------------------------------------------
static int reduce_descs(uint32_t side, uint32_t in_pitch, uint32_t out_pitch, LevelDesc* li, LevelDesc* lo) {
    if (side < 1) return fail("musica_k_reduce: side must be >= 1");
    const uint32_t so = (side + 1) / 2;
    if (in_pitch < side || (in_pitch & 3) || out_pitch < so || (out_pitch & 3)) return fail("musica_k_reduce: pitches must be multiples of 4 floats and cover the rows");
    li->S = (int)side; li->pitch = (int)in_pitch; li->plane = (size_t)in_pitch * side;
    lo->S = (int)so; lo->pitch = (int)out_pitch; lo->plane = (size_t)out_pitch * so;
    return 1;
}

int musica_k_reduce(musica_ctx* c, const float* d_in, uint32_t side, uint32_t in_pitch, float* d_out, uint32_t out_pitch, uint32_t batch) {
    CHECK_CTX(c);
    LevelDesc li, lo;
    if (!reduce_descs(side, in_pitch, out_pitch, &li, &lo)) return 0;
    launch_reduce(c->stream, d_in, li, d_out, lo, (int)batch, c->generic, 2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("kernel launch failed: %s", hipGetErrorString(e));
    return 1;
}

int musica_selftest_exact_math(musica_ctx* c, uint64_t mismatches[4]) {
    CHECK_CTX(c);
    if (!mismatches) return fail("musica_selftest_exact_math: mismatches is NULL");
    unsigned long long* d = nullptr;
    HIP_OK(hipMalloc(&d, 4 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        launch_selftest_exact_math(c->stream, d);
        e = hipStreamSynchronize(c->stream);
    }
    unsigned long long h[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail("musica_selftest_exact_math: %s", hipGetErrorString(e));
    for (int i = 0; i < 4; i++) mismatches[i] = h[i];
    return 1;
}

int musica_k_reduce_timed(musica_ctx* c, const float* d_in, uint32_t side, uint32_t in_pitch, float* d_out, uint32_t out_pitch, uint32_t batch,
                          uint32_t iters, double* mean_us) {
    CHECK_CTX(c);
    LevelDesc li, lo;
    if (!reduce_descs(side, in_pitch, out_pitch, &li, &lo)) return 0;
    if (iters < 1) iters = 1;
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    HIP_OK(hipEventRecord(a, c->stream));
    for (uint32_t i = 0; i < iters; i++) launch_reduce(c->stream, d_in, li, d_out, lo, (int)batch, c->generic, 2);
    HIP_OK(hipEventRecord(b, c->stream));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (mean_us) *mean_us = (double)ms * 1000.0 / (double)iters;
    return 1;
}

int musica_k_reduce_timed_rot(musica_ctx* c, const float* d_in, uint32_t side, uint32_t in_pitch, float* d_out, uint32_t out_pitch, uint32_t nbuf,
                              uint32_t iters, double* mean_us) {
    CHECK_CTX(c);
    LevelDesc li, lo;
    if (!reduce_descs(side, in_pitch, out_pitch, &li, &lo)) return 0;
    if (iters < 1) iters = 1;
    if (nbuf < 1) nbuf = 1;
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    HIP_OK(hipEventRecord(a, c->stream));
    for (uint32_t i = 0; i < iters; i++)
        launch_reduce(c->stream, d_in + (size_t)(i % nbuf) * li.plane, li, d_out + (size_t)(i % nbuf) * lo.plane, lo, 1, c->generic, side <= 4096 ? 4 : 5);
    HIP_OK(hipEventRecord(b, c->stream));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (mean_us) *mean_us = (double)ms * 1000.0 / (double)iters;
    return 1;
}

int musica_k_copy41_timed_rot(musica_ctx* c, const float* d_in, uint32_t side, float* d_out, uint32_t nbuf, uint32_t iters, double* mean_us) {
    CHECK_CTX(c);
    if (side < 8 || (side & 7)) return fail("musica_k_copy41_timed_rot: side must be a multiple of 8");
    if (iters < 1) iters = 1;
    if (nbuf < 1) nbuf = 1;
    const size_t in_plane = (size_t)side * side, out_plane = in_plane / 4;
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    HIP_OK(hipEventRecord(a, c->stream));
    for (uint32_t i = 0; i < iters; i++) launch_copy41(c->stream, d_in + (size_t)(i % nbuf) * in_plane, d_out + (size_t)(i % nbuf) * out_plane, (int)side);
    HIP_OK(hipEventRecord(b, c->stream));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (mean_us) *mean_us = (double)ms * 1000.0 / (double)iters;
    return 1;
}

void* musica_device_alloc(musica_ctx* c, size_t bytes) {
    if (!c || hipSetDevice(c->p.device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail("musica_device_alloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void musica_device_free(musica_ctx* c, void* p) {
    if (!c || !p) return;
    hipSetDevice(c->p.device);
    hipStreamSynchronize(c->stream);
    hipFree(p);
}
int musica_memcpy_h2d(musica_ctx* c, void* d_dst, const void* src, size_t bytes) {
    CHECK_CTX(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return 1;
}
int musica_memcpy_d2h(musica_ctx* c, void* dst, const void* d_src, size_t bytes) {
    CHECK_CTX(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 1;
}

// ---- steps in flight -------------------------------------------------------------------
// `depth` contexts of one GPU whose steps alternate ("Steps in flight"): every context is created with
// MUSICA_FLAG_LINEAR (one in-order stream, its script a linear graph), step s is enqueued on context s mod depth and nothing waits.
// A stream's hardware queue is the runtime's round-robin over every stream the process has created, and two of the four queues
// of an MI355X do not run side by side, so one context per queue is created and musica_pipeline_prime() times every cyclic
// window of `depth` of them for a few steps, keeps the fastest and destroys the rest (three contexts: 0.36 ms per 8 x 2048^2
// step on a good window, 0.41 on a bad one). The reference has one VulkanProcessing and one frame in flight
// (src/vk_processing.cpp:2535-2536); this is the throughput form of it.
struct musica_pipeline {
    std::vector<musica_ctx*> ctx;   // before prime(): one per hardware queue; after: the `depth` that stay, in step order
    uint32_t depth;
    uint64_t steps;
    float window_ms[MUSICA_PIPELINE_QUEUES];
    uint32_t windows;               // how many windows prime() timed (0: nothing to choose)
};

static int pipeline_run(const std::vector<musica_ctx*>& use, uint32_t steps) {
    for (uint32_t s = 0; s < steps; s++) {
        musica_ctx* c = use[s % use.size()];
        if (!musica_execute_device(c, c->d_input)) return 0;
    }
    for (musica_ctx* c : use)
        if (!musica_sync(c)) return 0;
    return 1;
}

musica_pipeline* musica_pipeline_create(const musica_params* params, uint32_t depth) { return musica_pipeline_create_ex(params, depth, nullptr); }
musica_pipeline* musica_pipeline_create_ex(const musica_params* params, uint32_t depth, const musica_tunables* tunables) {
    if (!params) { fail("musica_pipeline_create: params is NULL"); return nullptr; }
    if (depth < 1 || depth > 16) { fail("musica_pipeline_create: depth %u out of range [1, 16]", depth); return nullptr; }
    musica_pipeline* pl = nullptr;
    try {
        pl = new musica_pipeline();
        pl->depth = depth; pl->steps = 0; pl->windows = 0;
        for (float& w : pl->window_ms) w = 0.0f;
        musica_params q = *params;
        if (depth > 1) q.flags |= MUSICA_FLAG_LINEAR;
        const uint32_t n = (depth > 1 && depth < MUSICA_PIPELINE_QUEUES) ? MUSICA_PIPELINE_QUEUES : depth;
        for (uint32_t k = 0; k < n; k++) {
            // the contexts are identical: the launch geometry the first one tuned is copied to the others instead of tuned again
            musica_params qk = q;
            if (k > 0) qk.flags |= MUSICA_FLAG_NO_AUTOTUNE;
            musica_ctx* c = musica_create_ex(&qk, tunables);
            if (c && k > 0) { copy_rows(c, pl->ctx[0]); c->p.flags = q.flags; }
            if (!c) {
                // the contexts beyond `depth` only exist for the queue calibration: without them (e.g. device memory is short at a
                // large N) the first `depth` stay and prime() has nothing to choose from
                if (k >= depth) break;
                musica_pipeline_destroy(pl); return nullptr;
            }
            pl->ctx.push_back(c);
        }
        return pl;
    } catch (const std::exception& e) {
        fail("musica_pipeline_create: %s", e.what());
    }
    if (pl) musica_pipeline_destroy(pl);
    return nullptr;
}

void musica_pipeline_destroy(musica_pipeline* pl) {
    if (!pl) return;
    for (musica_ctx* c : pl->ctx) musica_destroy(c);
    delete pl;
}

uint32_t musica_pipeline_depth(const musica_pipeline* pl) { return pl ? pl->depth : 0; }

musica_ctx* musica_pipeline_context(musica_pipeline* pl, uint32_t k) {
    if (!pl || k >= pl->ctx.size()) { fail("musica_pipeline_context: no context %u", k); return nullptr; }
    return pl->ctx[k];
}

int musica_pipeline_upload(musica_pipeline* pl, const uint16_t* pixels) {
    if (!pl) return fail("musica_pipeline_upload: pipeline is NULL");
    for (musica_ctx* c : pl->ctx)
        if (!musica_upload(c, pixels)) return 0;
    return 1;
}

int musica_pipeline_prime(musica_pipeline* pl, uint32_t calibration_steps) {
    if (!pl) return fail("musica_pipeline_prime: pipeline is NULL");
    ABI_TRY
    for (int rep = 0; rep < 2; rep++)   // the first step of a context captures its graph, the second replays it
        if (!pipeline_run(pl->ctx, (uint32_t)pl->ctx.size())) return 0;
    const uint32_t n = (uint32_t)pl->ctx.size();
    if (n > pl->depth) {
        if (calibration_steps < pl->depth) calibration_steps = 3 * pl->depth;
        uint32_t best = 0;
        for (uint32_t first = 0; first < n; first++) {
            std::vector<musica_ctx*> use;
            for (uint32_t k = 0; k < pl->depth; k++) use.push_back(pl->ctx[(first + k) % n]);
            if (!pipeline_run(use, pl->depth)) return 0;   // warm this combination
            float ms = 1e30f;
            for (int rep = 0; rep < 2; rep++) {
                const auto t0 = std::chrono::steady_clock::now();
                if (!pipeline_run(use, calibration_steps)) return 0;
                const float dt = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / (float)calibration_steps;
                if (dt < ms) ms = dt;
            }
            if (first < MUSICA_PIPELINE_QUEUES) pl->window_ms[first] = ms;
            if (ms < pl->window_ms[best]) best = first;
        }
        pl->windows = n < MUSICA_PIPELINE_QUEUES ? n : MUSICA_PIPELINE_QUEUES;
        std::vector<musica_ctx*> keep;
        for (uint32_t k = 0; k < pl->depth; k++) keep.push_back(pl->ctx[(best + k) % n]);
        for (musica_ctx* c : pl->ctx) {
            bool stays = false;
            for (musica_ctx* q : keep) stays = stays || q == c;
            if (!stays) musica_destroy(c);
        }
        pl->ctx = keep;
    }
    pl->steps = 0;
    return 1;
    ABI_CATCH("musica_pipeline_prime")
}

uint32_t musica_pipeline_calibration(const musica_pipeline* pl, float* window_ms) {
    if (!pl) return 0;
    if (window_ms)
        for (uint32_t k = 0; k < pl->windows; k++) window_ms[k] = pl->window_ms[k];
    return pl->windows;
}

int musica_pipeline_step(musica_pipeline* pl, const uint16_t* d_pixels) {
    if (!pl || pl->ctx.empty()) return fail("musica_pipeline_step: no pipeline");
    musica_ctx* c = pl->ctx[pl->steps % pl->ctx.size()];
    if (!musica_execute_device(c, d_pixels ? d_pixels : c->d_input)) return 0;
    pl->steps++;
    return 1;
}

musica_ctx* musica_pipeline_last(musica_pipeline* pl) {
    if (!pl || pl->ctx.empty() || pl->steps == 0) { fail("musica_pipeline_last: no step has been enqueued"); return nullptr; }
    return pl->ctx[(pl->steps - 1) % pl->ctx.size()];
}

int musica_pipeline_sync(musica_pipeline* pl) {
    if (!pl) return fail("musica_pipeline_sync: pipeline is NULL");
    for (musica_ctx* c : pl->ctx)
        if (!musica_sync(c)) return 0;
    return 1;
}

}  // extern "C"
