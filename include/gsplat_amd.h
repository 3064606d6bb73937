/* C ABI of libgsplat_amd.so: the MI355X (gfx950) Gaussian rasterizer hot path.
 *
 * Every pointer named "device" is a device address on the GPU that is current
 * when the call is made; `stream` is a hipStream_t passed as void*.  No entry
 * point allocates or frees device memory: the caller brings every buffer
 * (sizes from the gsr_*_bytes queries).  No torch types cross this boundary.
 * All entry points return GSR_OK (0) or a negative GsrStatus and never throw.
 *
 * Reference interface replaced (paths under
 * submodules/diff-gaussian-rasterization/ unless stated otherwise):
 *   gsr_forward_stage1 + gsr_forward_stage2
 *        CudaRasterizer::Rasterizer::forward   cuda_rasterizer/rasterizer.h:31-66,
 *                                              cuda_rasterizer/rasterizer_impl.cu:197-433
 *        (split in two because the binning buffer size is only known after the
 *         prefix sum; the reference resizes it through a callback at
 *         rasterizer_impl.cu:349-353)
 *   gsr_backward
 *        CudaRasterizer::Rasterizer::backward  cuda_rasterizer/rasterizer.h:68-102,
 *                                              cuda_rasterizer/rasterizer_impl.cu:437-577
 *   gsr_mark_visible
 *        CudaRasterizer::Rasterizer::markVisible  cuda_rasterizer/rasterizer.h:24-29
 *   gsr_geom_bytes / gsr_image_bytes / gsr_binning_bytes
 *        required<GeometryState|ImageState|BinningState>  cuda_rasterizer/rasterizer_impl.h:64-72
 *   gsr_sort_pairs_u64_u32
 *        cub::DeviceRadixSort::SortPairs call site  cuda_rasterizer/rasterizer_impl.cu:373-380
 *   gsk_dist2
 *        SimpleKNN::knn   submodules/simple-knn/simple_knn.h, simple_knn.cu:186-222
 *   gss_ssim_forward / gss_ssim_backward
 *        fusedssim / fusedssim_backward   submodules/fused-ssim/ssim.cu:368-444
 */
#ifndef GSPLAT_AMD_H
#define GSPLAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GSR_OK = 0,
    GSR_ERR_ARG = -1,        /* null / inconsistent argument */
    GSR_ERR_DIM = -2,        /* unsupported number of `reps` channels (supported: 0, 4, 8, 16) */
    GSR_ERR_OVERFLOW = -3,   /* more than 2^31 - 1 (Gaussian, tile) instances, the reference's limit (its count is an int) */
    GSR_ERR_BUFFER = -4,     /* a caller buffer is smaller than the matching *_bytes query */
    GSR_ERR_HIP = -5,        /* a HIP runtime call failed */
    GSR_ERR_SORT = -6,       /* radix sort look-back did not complete */
    GSR_ERR_LIMIT = -7,      /* input larger than the fixed scratch of gsr_sort_pairs_u64_u32_noalloc */
    GSR_ERR_PREFILTER = -8   /* scene->prefiltered was set and a point failed the near-plane test: where the reference
                                prints "Point is filtered although prefiltered is set" and traps on the device
                                (cuda_rasterizer/auxiliary.h:181-185), gsr_forward_stage1 returns this */
} GsrStatus;

/* Inputs of one view.  Nullable pointers select the options exactly as the
 * reference does: shs XOR colors_precomp, (scales AND rotations) XOR
 * cov3D_precomp; reps is NULL iff DIM == 0. */
typedef struct {
    int P;                 /* number of Gaussians */
    int D;                 /* active SH degree */
    int M;                 /* SH coefficients allocated per Gaussian */
    int DIM;               /* channels of `reps`: 0, 4, 8 or 16.  The reference fixes the number at compile time
                              (NUM_DIM 4, cuda_rasterizer/config.h:16, which a user edits to match the model's
                              --dim, arguments/__init__.py:98); here it is a run-time value served by one
                              instantiation of the blend kernels each */
    int W, H;              /* image size in pixels */
    const float* bg;              /* device [3] */
    const float* means3D;         /* device [P,3] */
    const float* shs;             /* device [P,M,3] or NULL */
    const float* colors_precomp;  /* device [P,3] or NULL */
    const float* opacities;       /* device [P] */
    const float* scales;          /* device [P,3] or NULL */
    const float* rotations;       /* device [P,4] or NULL */
    const float* cov3D_precomp;   /* device [P,6] or NULL */
    const float* reps;            /* device [P,DIM] or NULL */
    const float* viewmatrix;      /* device [16], as scene/cameras.py:86 stores it */
    const float* projmatrix;      /* device [16], full projection, scene/cameras.py:88 */
    const float* campos;          /* device [3] */
    float scale_modifier;
    float tanfovx, tanfovy;
    int prefiltered;       /* nonzero: the caller asserts that no point is culled; gsr_forward_stage1 checks it
                              (GSR_ERR_PREFILTER).  gsr_preprocess_batch does not look at it. */
    int antialiasing;
    int debug;             /* nonzero: synchronise and check for errors after every launch */
    void** profile_events; /* NULL, or GSR_NUM_PROFILE_EVENTS hipEvent_t handles that the calls record on
                              `stream` around their kernels (see GsrProfileEvent); the caller reads the
                              elapsed times after synchronising */
    /* Raw-parameter mode.  No counterpart in the reference's operator: it folds what the model does
     * before every render (scene/gaussian_model.py:41-53, 470-502) into the kernels.
     *   raw_params != 0: `opacities` holds logits (sigmoid is applied), `scales` log-scales (exp),
     *                    `rotations` unnormalised quaternions (x / max(|x|, 1e-12)); the backward then
     *                    returns gradients with respect to these raw tensors.
     *   shs_rest != NULL: the SH coefficients come as two tensors, `shs` = [P,1,3] (DC term) and
     *                    `shs_rest` = [P,M-1,3], instead of their concatenation (M still counts both). */
    int raw_params;
    const float* shs_rest;
} GsrScene;

/* Slots of GsrScene.profile_events.  Each is recorded AFTER the named step. */
typedef enum {
    GSR_EV_STAGE1_BEGIN = 0,
    GSR_EV_PREPROCESS = 1,     /* forward preprocess kernel */
    GSR_EV_SCAN = 2,           /* block-sum scan */
    GSR_EV_STAGE2_BEGIN = 3,
    GSR_EV_DEPTH_ORDER = 4,    /* front-to-back order of the Gaussians (32-bit depth sort) */
    GSR_EV_EMIT = 5,           /* key emission in depth order */
    GSR_EV_SORT = 6,           /* sort of the instances on the tile bits */
    GSR_EV_RANGES = 7,         /* tile ranges */
    GSR_EV_RENDER = 8,         /* forward blend */
    GSR_EV_FINISH = 9,         /* statistics unpack */
    GSR_EV_BWD_BEGIN = 10,
    GSR_EV_BWD_CLEAR = 11,     /* gradient accumulator clear */
    GSR_EV_BWD_RENDER = 12,    /* backward blend */
    GSR_EV_BWD_PREPROCESS = 13,/* per-Gaussian backward (gsr_backward) or screen-space gradient copy (gsr_backward_blend) */
    GSR_EV_BATCH_BEGIN = 14,
    GSR_EV_BATCH = 15,         /* per-Gaussian backward of a batch of views (gsr_backward_geometry_batch) */
    GSR_EV_FWD_BATCH_BEGIN = 16,
    GSR_EV_FWD_BATCH = 17,     /* preprocess of a batch of views (gsr_preprocess_batch) */
    GSR_EV_TILE_COUNT = 18,    /* tile ranges and sort histograms from the tile rectangles (after STAGE2_BEGIN, before the depth order) */
    GSR_EV_FWD_BATCH_SORT_BEGIN = 19,
    GSR_EV_FWD_BATCH_SORT = 20,/* front-to-back orders of the views of a batch (gsr_preprocess_batch, after its host block) */
    GSR_NUM_PROFILE_EVENTS = 21
} GsrProfileEvent;

/* Outputs of the forward pass.  None of them needs initialising by the caller. */
typedef struct {
    float* out_color;        /* device [3,H,W] */
    float* out_invdepth;     /* device [H,W] or NULL (then no inverse depth is rendered) */
    int* radii;              /* device [P] */
    float* accum_alpha;      /* device [H,W]: final transmittance per pixel */
    float* gauss_sum;        /* device [P] */
    int* gauss_count;        /* device [P] */
    int* last_contr_gauss;   /* device [H,W], -1 where nothing contributed */
    float* out_depths;       /* device [P]; written only where radii > 0 */
    float* out_reps;         /* device [DIM,H,W] or NULL */
    void* geom_buffer;       /* device, >= gsr_geom_bytes(P) */
    size_t geom_bytes;
    void* image_buffer;      /* device, >= gsr_image_bytes(W,H) */
    size_t image_bytes;
    int64_t num_visible;     /* written by gsr_forward_stage1, read by gsr_forward_stage2 (same struct): Gaussians
                                that produce at least one instance; stage 2 orders only those by depth.  A value
                                < 0 there means "not known" and all P are ordered. */
    int preprocessed;        /* nonzero: geom_buffer, radii and out_depths were filled by gsr_preprocess_batch for
                                this view; gsr_forward_stage1 then only runs the prefix sums.  2 (only where
                                gsr_preprocess_batch reported GsrForwardView.depth_ordered, and with that call's
                                num_visible): the geometry buffer also holds the view's front-to-back order of
                                the Gaussians, and gsr_forward_stage2 does not sort them again */
    uint32_t* sort_error_host; /* NULL, or a word of host memory that the GPU can write (hipHostMalloc / pinned):
                                the last kernel of gsr_forward_stage2 stores the radix sorts' error words there (the
                                instance sort's OR the depth order's, also when gsr_preprocess_batch made that order), in
                                stream order and without making the call wait.  Nonzero = a look-back did not
                                complete within its spin limit and the instance list is not valid (what
                                GSR_ERR_SORT reports when scene->debug makes the call wait; then 0 is stored).
                                A caller that sets the word to a value the library never stores (all ones) before
                                the call sees from the word alone when the GPU has passed the call.  The word must
                                stay valid until then. */
} GsrForwardOut;

/* Gradients.  Inputs are the three per-pixel gradients; every output is fully
 * written by the call (no zero-fill needed).  `work` is scratch. */
typedef struct {
    const float* dL_dout_color;     /* device [3,H,W] */
    const float* dL_dout_invdepth;  /* device [H,W] or NULL */
    const float* dL_dout_reps;      /* device [DIM,H,W] or NULL iff DIM == 0 */
    float* dL_dmeans2D;      /* device [P,3] (z stays 0) */
    float* dL_dcolors;       /* device [P,3]; may be NULL unless colors_precomp was given (then it is not stored) */
    float* dL_dopacity;      /* device [P] */
    float* dL_dmeans3D;      /* device [P,3] */
    float* dL_dcov3D;        /* device [P,6]; may be NULL unless cov3D_precomp was given */
    float* dL_dsh;           /* device [P,M,3] or NULL iff M == 0 */
    float* dL_dscales;       /* device [P,3] or NULL iff scales == NULL */
    float* dL_drotations;    /* device [P,4] or NULL iff rotations == NULL */
    float* dL_dreps;         /* device [P,DIM] or NULL iff DIM == 0 */
    void* work;              /* device, >= gsr_backward_work_bytes_dim(P, DIM) (= gsr_backward_work_bytes(P) for DIM <= 4) */
    size_t work_bytes;
    float* dL_dsh_rest;      /* device [P,M-1,3] iff scene.shs_rest != NULL (then dL_dsh is [P,1,3]) */
    int accumulate;          /* nonzero: ADD into dL_dmeans3D, dL_dopacity, dL_dsh(+rest), dL_dscales,
                                dL_drotations, dL_dreps instead of overwriting them (gradient accumulation
                                over the views of a batch; Gaussians that were not rendered are then not
                                touched).  dL_dmeans2D, dL_dcolors, dL_dcov3D are always overwritten. */
    uint8_t* row_blocks;     /* device [(P + 63) / 64] or NULL; gsr_backward_geometry_batch(_range) only.  One byte
                                per block of 64 consecutive Gaussians, kept by the caller next to PERSISTENT
                                gradient tensors: 0 = the block's rows of all the parameter gradients above are
                                zero.  The call does not rewrite such a block with zeros when no view reached one
                                of its Gaussians (in an occluded scene that is nearly all of its writes), and it
                                updates the bytes: 1 where it stored gradients, 0 where it stored zeros.  Start
                                from zeroed tensors and zeroed bytes (or bytes of 1 and any contents), and set the
                                bytes of rows anything else writes to (a collective that adds other GPUs' rows)
                                to 1.  NULL: every row is written, as documented above. */
} GsrBackwardIO;

size_t gsr_geom_bytes(int P);
size_t gsr_image_bytes(int W, int H);
size_t gsr_binning_bytes(int64_t num_rendered);
size_t gsr_backward_work_bytes(int P);
/* The same for a model with DIM feature channels: more than four take a second 64-byte record per Gaussian. */
size_t gsr_backward_work_bytes_dim(int P, int DIM);

/* Preprocess + prefix sum.  Blocks the host once (as the reference does,
 * rasterizer_impl.cu:338-346) to return the instance count in *num_rendered. */
int gsr_forward_stage1(const GsrScene* scene, GsrForwardOut* out, void* stream, int64_t* num_rendered);

/* Key emission, sort, tile ranges, blending.  binning_buffer must hold
 * gsr_binning_bytes(num_rendered) and be kept for the backward pass. */
int gsr_forward_stage2(const GsrScene* scene, const GsrForwardOut* out, int64_t num_rendered, void* binning_buffer,
                       size_t binning_bytes, void* stream);

int gsr_backward(const GsrScene* scene, const int* radii, const void* geom_buffer, const void* binning_buffer,
                 const void* image_buffer, int64_t num_rendered, const GsrBackwardIO* io, void* stream);

/* Deferred backward for a batch of views (no counterpart in the reference, which backpropagates one
 * view at a time; for callers that accumulate the gradients of several views before an optimiser
 * step).  gsr_backward_blend is the first half of gsr_backward: the blend backward of one view into
 * io->work (its per-Gaussian accumulators) and, if not NULL, io->dL_dmeans2D; only io->dL_dout_*,
 * io->work and io->dL_dmeans2D are used.  gsr_backward_geometry_batch is the second half for all the views
 * together: each Gaussian's parameters are read once and every parameter gradient (dL_dmeans3D,
 * dL_dopacity, dL_dsh + dL_dsh_rest, dL_dscales, dL_drotations, dL_dreps) is written or, with
 * io->accumulate, added to once, the sum over the views.  It needs the SH tensors split
 * (scene->shs_rest), M == 16 and scales + rotations; the camera fields of `scene` are ignored. */
typedef struct {
    const float* viewmatrix;   /* device [16] */
    const float* projmatrix;   /* device [16] */
    const float* campos;       /* device [3] */
    float tanfovx, tanfovy;
    int W, H;
    int antialiasing;
    int has_invdepth;          /* that view's backward was given dL_dout_invdepth */
    const int* radii;          /* device [P], from that view's forward (kept for symmetry with gsr_backward; the
                                  batch call does not read it: the hit counts in the geometry buffer say which
                                  Gaussians a view reached, and only rendered ones can have been reached) */
    const void* geom_buffer;   /* that view's geometry buffer */
    const void* work;          /* that view's accumulators, filled by gsr_backward_blend */
    float* dL_dmeans2D;        /* device [P,3] or NULL: that view's screen-space gradient, written by the batch call
                                  (then gsr_backward_blend can be given io->dL_dmeans2D = NULL and skips its copy) */
    uint8_t* means2D_blocks;   /* device [(P + 63) / 64] or NULL: the same bookkeeping as GsrBackwardIO.row_blocks for
                                  this view's dL_dmeans2D, for callers that keep that tensor from step to step */
} GsrBackwardView;

int gsr_backward_blend(const GsrScene* scene, const int* radii, const void* geom_buffer, const void* binning_buffer,
                       const void* image_buffer, int64_t num_rendered, const GsrBackwardIO* io, void* stream);
int gsr_backward_geometry_batch(const GsrScene* scene, int nviews, const GsrBackwardView* views,
                                const GsrBackwardIO* io, void* stream);
/* The same for the Gaussians [first, first + count) only (first a multiple of 256, first + count <= P,
 * count a multiple of 64 unless the range ends at P: the block states are kept per 64 rows; otherwise
 * GSR_ERR_ARG): rows of the gradient tensors outside the range are not touched.  A caller that exchanges the
 * gradients between GPUs runs the ranges one after the other and starts the exchange of a range's
 * rows as soon as its call is enqueued, so that the exchange overlaps the remaining ranges. */
int gsr_backward_geometry_batch_range(const GsrScene* scene, int nviews, const GsrBackwardView* views,
                                      const GsrBackwardIO* io, int first, int count, void* stream);

/* Forward preprocess of a batch of views in one pass over the Gaussians (no counterpart in the
 * reference, which renders one view per iteration; for callers that render several views from the
 * same parameters, e.g. the views of one optimiser step).  Each Gaussian's parameters and SH row are
 * read once for all the views; per view the call leaves in geom_buffer, radii and out_depths exactly
 * what the preprocess of gsr_forward_stage1 would (bit for bit), except the 3D covariance, which the
 * backward recomputes.  It also runs the prefix sums of every view and, blocking the host once for
 * all of them, returns each view's instance count in num_rendered / num_visible: the views can go
 * straight to gsr_forward_stage2 (GsrForwardOut.preprocessed = 1, or 2 where depth_ordered was reported,
 * num_visible copied over, the same three buffers), so a step of V views blocks the host once instead of
 * V times.
 * (gsr_forward_stage1 with preprocessed != 0 is also allowed: the sums are in place, so it only repeats the
 * copy of the view's totals and returns the same two counts.)
 * Needs the SH tensors split (scene->shs_rest), M == 16 and scales + rotations; the camera fields,
 * W, H and antialiasing of `scene` are ignored (each view brings its own).  Returns
 * GSR_ERR_OVERFLOW if a view has more than 2^31 - 1 instances. */
typedef struct {
    const float* viewmatrix;   /* device [16] */
    const float* projmatrix;   /* device [16] */
    const float* campos;       /* device [3] */
    float tanfovx, tanfovy;
    int W, H;
    int antialiasing;
    int* radii;                /* device [P], out */
    float* out_depths;         /* device [P], out (written where radii > 0) */
    void* geom_buffer;         /* device, >= gsr_geom_bytes(P), out */
    size_t geom_bytes;
    int64_t num_rendered;      /* out (host): (Gaussian, tile) instances of this view */
    int64_t num_visible;       /* out (host): Gaussians of this view with at least one instance */
    int depth_ordered;         /* out (host): nonzero = the call also ordered this view's Gaussians front to back (the
                                  depth sorts of up to eight views share their launches: a view's own sort is a
                                  single wave of workgroups waiting on its look-back chain); pass
                                  GsrForwardOut.preprocessed = 2 to gsr_forward_stage2 then.  The value says by
                                  which keys: 2 = the depth bits minus the near plane's on 27 bits (three radix
                                  passes; every depth of the batch below 13107.2), 1 = all 32 depth bits (four) */
} GsrForwardView;
int gsr_preprocess_batch(const GsrScene* scene, int nviews, GsrForwardView* views, void* stream);

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream);

/* Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [0, end_bit).
 * Both buffer pairs are clobbered; on return *result_buffer (0 or 1) says which
 * pair holds the sorted data. */
size_t gsr_sort_temp_bytes(int64_t n);
/* Byte offset, inside `temp`, of the sort's error word (u32): nonzero after the sort has run = a look-back
 * did not complete within its spin limit and the output is not valid. */
size_t gsr_sort_temp_error_offset(void);
int gsr_sort_pairs_u64_u32(uint64_t* keys0, uint32_t* vals0, uint64_t* keys1, uint32_t* vals1, int64_t n, int end_bit,
                           void* temp, size_t temp_bytes, void* stream, int* result_buffer);

/* The same sort for callers that cannot allocate device memory or bring a temp buffer (tight serving
 * loops, harnesses that forbid hipMalloc, code that will later run under HIP graph capture).  No
 * counterpart in the reference.  The sort's temp area (digit histograms, tickets, look-back status
 * words) is a device array inside the library, 4 MiB per GPU that the library runs on, held until the
 * process ends; the call allocates and frees nothing.  It is not free of other runtime calls: the
 * first call on a device looks the array's address up (hipGetSymbolAddress) and every call sets a
 * kernel attribute (hipFuncSetAttribute) before its launches and clears its temp area with
 * hipMemsetAsync; a caller that captures it into a graph makes one call outside the capture first
 * (capture itself is not covered by this library's tests).  n is limited to
 * gsr_sort_noalloc_max_pairs() (2^21 pairs); more, or a sort whose status words do not fit the array
 * (gsr_test_force_wide_sort_status doubles them), returns GSR_ERR_LIMIT.  Every call on a GPU uses
 * that GPU's one array: calls that overlap on one GPU (two streams, two threads) race and nothing
 * detects it; calls on one stream, or otherwise ordered, are fine.  Results are bit-identical to
 * gsr_sort_pairs_u64_u32.
 * gsr_sort_noalloc_error: the error word of the last such sort on the current device (nonzero = a
 * look-back did not complete, the output is not valid); waits for `stream`. */
int64_t gsr_sort_noalloc_max_pairs(void);
int gsr_sort_pairs_u64_u32_noalloc(uint64_t* keys0, uint32_t* vals0, uint64_t* keys1, uint32_t* vals1, int64_t n,
                                   int end_bit, void* stream, int* result_buffer);
int gsr_sort_noalloc_error(void* stream, uint32_t* err);

/* Debug / test access: byte offsets of the arrays inside the scratch buffers. */
typedef struct {
    size_t rec, cov3D, tiles, clamped, block_sums, total;           /* geometry buffer */
    size_t depth_sort;                                               /* geometry buffer: temp area of the depth order's
                                                                        sort (its error word: + gsr_sort_temp_error_offset()) */
    size_t final_T, n_contrib, ranges;                               /* image buffer */
    size_t keys0, keys1, vals0, vals1;                               /* binning buffer: u32 tile ids, u32 indices */
    int sorted_buffer;                                               /* which pair holds the sorted list; its vals
                                                                        array always does, its keys array (the sorted
                                                                        tile ids) only when the instances were sorted
                                                                        as pairs: see gsr_test_packed_instances */
} GsrLayout;
int gsr_layout(int P, int W, int H, int64_t num_rendered, GsrLayout* out);

/* sizeof of the structs above as the library was built, for bindings that mirror them field by field
 * (ctypes, cgo, JNA): 0 GsrScene, 1 GsrForwardOut, 2 GsrBackwardIO, 3 GsrBackwardView, 4 GsrForwardView,
 * 5 GsrLayout; 0 for any other number.  No counterpart in the reference, whose operator passes tensors. */
size_t gsr_struct_bytes(int which);

/* The sorted instance list as the reference holds it: point_list_keys[i] = (tile << 32 | depth bits)
 * (rasterizer_impl.cu:93-104, BinningState::point_list_keys).  This implementation sorts
 * (tile id, Gaussian index) pairs emitted in depth order and does not store the 64-bit keys; this
 * call rebuilds them on the device from the buffers a forward pass left: the tile of an instance is
 * the one whose range (image buffer) holds it, the depth bits are those of its Gaussian (geometry
 * buffer), the Gaussian is the sorted list's entry (binning buffer).  keys_out: device, R x u64. */
int gsr_sorted_keys_u64(int P, int W, int H, int64_t num_rendered, const void* geom_buffer,
                        const void* binning_buffer, const void* image_buffer, uint64_t* keys_out, void* stream);

/* simple-knn: mean squared distance to the 3 nearest neighbours. */
size_t gsk_dist2_temp_bytes(int P);
int gsk_dist2(int P, const float* points, float* mean_dists, void* temp, size_t temp_bytes, void* stream);

/* fused-ssim.  The derivative maps may be NULL in the forward (train = false). */
int gss_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);
int gss_ssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                      const float* dL_dmap, float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq,
                      const float* dm_dsigma12, void* stream);

/* The photometric training loss in two kernels (train.py:222-230 computes it with a dozen PyTorch
 * kernels around fused-ssim; gaussian_renderer/__init__.py:179 clamps the render before):
 *   loss[0] = (1 - lambda_dssim) * mean|clamp(img, 0, 1) - gt| + lambda_dssim * (1 - mean ssim(clamp(img, 0, 1), gt))
 *   loss[1] = the L1 mean, loss[2] = the ssim mean            (device [3])
 * img and gt are device [planes, H, W] (planes = batch x channels), img the UNCLAMPED render.
 * `partial` is scratch of gss_loss_partial_floats(planes, H, W) floats; the three derivative maps
 * (device [planes, H, W] each) are kept for the backward, which returns dL/dimg for the incoming
 * gradient *grad_loss of loss[0] (a device scalar, so the host never waits for it). */
size_t gss_loss_partial_floats(int planes, int H, int W);
int gss_l1_ssim_loss_forward(int planes, int H, int W, float C1, float C2, float lambda_dssim, const float* img,
                             const float* gt, float* loss, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                             float* partial, void* stream);
int gss_l1_ssim_loss_backward(int planes, int H, int W, float lambda_dssim, const float* grad_loss, const float* img,
                              const float* gt, float* dL_dimg, const float* dm_dmu1, const float* dm_dsigma1_sq,
                              const float* dm_dsigma12, void* stream);

/* mean|a - b| over n floats (utils/loss_utils.py:23-24, used at train.py:222 and :249 for the
 * rendered feature channels), b a device array or, when b is NULL, the number b_scalar: forward
 * in two kernels (loss: device [1]; `partial`: scratch of gss_l1_partial_floats(n) floats),
 * backward in one: dL_da[i] = *grad_loss / n * sign(a[i] - b[i]), *grad_loss a device scalar. */
size_t gss_l1_partial_floats(int64_t n);
int gss_l1_loss_forward(int64_t n, const float* a, const float* b, float b_scalar, float* loss, float* partial,
                        void* stream);
int gss_l1_loss_backward(int64_t n, const float* grad_loss, const float* a, const float* b, float b_scalar,
                         float* dL_da, void* stream);

/* Fused Adam step over one parameter tensor [N, M] (row = Gaussian), in place.
 * Replaces, for the model's parameter groups, torch.optim.Adam as configured at
 * scene/gaussian_model.py:112-113 (same operation order as its single-tensor path) and provides
 * the sparse step the reference looks for at train.py:38-42, 339-343
 * (SparseGaussianAdam.step(visibility, N) of graphdeco-inria/diff-gaussian-rasterization, branch
 * 3dgs_accel: not vendored by the fork; Adam without bias correction on the visible rows).
 *   visible: device [N] bytes, nonzero = update the row; NULL = every row.
 *   bias_correction1 = 1 - beta1^step, bias_correction2_sqrt = sqrt(1 - beta2^step); pass 1, 1 for
 *   the sparse variant.  The scalars are doubles because torch forms 1 - beta, lr / bias_correction1
 *   in double before rounding to float, and so does this call. */
int gso_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                    int64_t N, int M, double lr, double beta1, double beta2, double eps, double bias_correction1,
                    double bias_correction2_sqrt, void* stream);

/* Densification statistics after one view, in place (all device pointers): for every Gaussian with
 * radii > 0, max_radii2D = max(max_radii2D, radii) (train.py:298-309), xyz_gradient_accum +=
 * |dL_dmeans2D.xy| and denom += 1 (scene/gaussian_model.py:202-214).  radii [P] int, dL_dmeans2D [P,3],
 * xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P]. */
int gso_densify_stats(int P, const int* radii, const float* dL_dmeans2D, float* xyz_gradient_accum, float* denom,
                      float* max_radii2D, void* stream);

/* Densification decisions and the split, each in one pass (GaussianModel.densify_and_prune,
 * scene/gaussian_model.py:216-252, which the reference runs as dozens of elementwise PyTorch kernels
 * with boolean-mask temporaries).  All pointers are device pointers; scaling [N,3] holds log-scales,
 * opacity [N] logits, rotation [N,4] unnormalised quaternions, as the model stores them.
 *   gso_densify_select: selected[i] = 1 for the Gaussians to clone (split == 0: |g| >= grad_threshold and
 *     max(exp(scaling)) <= size_threshold, :263-270) or to split (split != 0: g >= grad_threshold and
 *     max(exp(scaling)) > size_threshold, :356-364), g = xyz_gradient_accum / denom with NaN -> 0
 *     (:224-225) for rows below n_grads and 0 for the rows beyond (the clones appended since, :356-357);
 *     size_threshold = percent_dense * scene_extent.
 *   gso_prune_mask: mask[i] = sigmoid(opacity) < min_opacity, OR-ed, when max_screen_size > 0, with
 *     max_radii2D > max_screen_size and max(exp(scaling)) > max_world_size (= 0.1 * extent) (:237-244),
 *     and with also[i] (nullable: the parents a split replaced, :388-392).
 *   gso_split_children: the K * n children of the K selected parents (`parents`: their row indices,
 *     ascending), child j of parent j mod K as `.repeat(n, 1)` orders them (:367-377): new_xyz =
 *     xyz[parent] + R(q / |q|) (normals[j] * exp(scaling[parent])) with normals [K*n,3] standard normal
 *     samples from the caller, new_scaling = log(exp(scaling[parent]) / (0.8 n)); parent_of_child [K*n] is
 *     written for gso_gather_rows to copy the other tensors' rows. */
int gso_densify_select(int64_t N, int64_t n_grads, const float* xyz_gradient_accum, const float* denom,
                       const float* scaling, float grad_threshold, float size_threshold, int split, uint8_t* selected,
                       void* stream);
int gso_prune_mask(int64_t N, const float* opacity, const float* max_radii2D, const float* scaling, float min_opacity,
                   float max_screen_size, float max_world_size, const uint8_t* also, uint8_t* mask, void* stream);
int gso_split_children(int64_t K, int n, const uint32_t* parents, const float* xyz, const float* scaling,
                       const float* rotation, const float* normals, float* new_xyz, float* new_scaling,
                       uint32_t* parent_of_child, void* stream);

/* mask[i] = 1 if counts[v][i] != 0 for one of the n arrays, else 0.  No counterpart in the reference: the row
 * mask of the Gaussians that some pixel of some view reached (the views' gauss_count outputs,
 * forward.cu:431-432), whose gradient rows are the only non-zero ones of a step; a data-parallel caller
 * exchanges those rows only.  counts: HOST array of n device pointers, each to [N] ints and 16-byte
 * aligned; mask: device [N] bytes, 4-byte aligned. */
int gso_mark_nonzero(int64_t N, int n, const int* const* counts, uint8_t* mask, void* stream);

/* Row compaction for pruning (scene/gaussian_model.py:402-441, which keeps the rows of a mask in
 * every parameter, Adam moment and statistics tensor by boolean indexing: one host-blocking
 * nonzero() and one gather per tensor).  gso_compact_map turns the mask (device [N] bytes, nonzero =
 * keep) into the ascending list of kept row indices src_of_dst (device [N] u32, the first *kept
 * entries are valid) and blocks the host once to return the count; gso_gather_rows then copies
 * dst[r, :] = src[src_of_dst[r], :] for rows of M 4-byte elements (any 4-byte type). */
size_t gso_compact_temp_bytes(int64_t N);
int gso_compact_map(int64_t N, const uint8_t* keep, uint32_t* src_of_dst, void* temp, size_t temp_bytes, int64_t* kept,
                    void* stream);
int gso_gather_rows(const void* src, void* dst, const uint32_t* src_of_dst, int64_t K, int M, void* stream);

/* The four test hooks below are per calling thread: they change the calls made from the thread that set
 * them and nothing else.
 * Test hook: while on, the radix sorts use the 64-bit look-back status words of sorts of 2^30 pairs and more
 * at any size (gsr_sort_temp_bytes and gsr_binning_bytes answer accordingly). */
void gsr_test_force_wide_sort_status(int on);

/* Test hook: which forward blend kernel gsr_forward_stage2 takes.  The kernel that leaves out, per 8x8 pixel
 * quadrant, the list entries whose alpha >= 1/255 ellipse cannot reach the quadrant is chosen when a view's
 * splats cover fewer than six tiles on average; 1 always takes it, 0 never, -1 restores the choice. */
void gsr_test_forward_reach_masks(int mode);

/* Test hook: how an instance travels between the key emission and the pass that orders each segment of
 * tiles.  On (the default): as one 32-bit word, (low tile digit << (32 - digit bits)) | Gaussian index,
 * whenever the tile ids have two digits and the number of Gaussians fits below the digit (2^25 Gaussians
 * at 1080p, 2^24 at 4K); the sorted tile ids are then not stored.  Off: always as a (tile id, index) pair. */
void gsr_test_packed_instances(int on);

/* Test hook: the keys gsr_preprocess_batch orders each view's Gaussians by.  Off (the default): when every
 * depth of the batch lies below 0.2 * 2^16 = 13107.2, the depth bits minus the bits of the near plane's 0.2,
 * on 27 bits in three radix passes; the raw 32 depth bits in four passes otherwise.  On: always the latter.
 * The order is the same either way. */
void gsr_test_wide_depth_keys(int on);

/* Test hook: while set, every gsr_forward_stage2 sets a sort's error word as a tripped look-back would:
 * 1 the instance sort's (binning buffer), 2 the depth order's (geometry buffer); 0 off. */
void gsr_test_inject_sort_error(int which);

const char* gsr_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif
