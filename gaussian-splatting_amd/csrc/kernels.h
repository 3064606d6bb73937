// Host-callable launchers of the HIP kernels (internal to libgsplat_amd.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsplat_amd.h"
#include "gs_device.h"

namespace gs {

// Geometry buffer with the extra per-Gaussian arrays this implementation keeps.
constexpr int SCAN_BATCH_MAX = 16;   // views whose prefix scans share a launch

struct GeomBuf {
    SplatRec* rec;        // [P]
    float* cov3D;         // [P,6]
    uint2* rect;          // [P] {x0 | y0<<16, w | h<<16} in tiles; zero for a Gaussian without instances
    uint2* dt;            // [P] {depth bits, tiles}, directly behind rect (the kernels that need one of the
                          //     two read 8 bytes per Gaussian instead of 16)
    GStat* gstat;         // [P] (sum of T, hit count), accumulated by the forward blend
    uint8_t* clamped;     // [P]
    uint32_t* block_sums; // [nblocks+1] tiles per block of 256 Gaussians in index order (gives the total)
    uint32_t* vis_sums;   // [nblocks+1] Gaussians with instances per block, then exclusive offsets
    uint64_t* total;      // [4] instances, overflow flag, Gaussians with instances, "a point was culled although
                          // the caller declared them prefiltered"; then [3 * SCAN_BATCH_MAX]: the first
                          // three of every view whose scans ran in one launch with this view as the first;
                          // then [1]: "a Gaussian of some view of the batch whose first view this is lies at
                          // or beyond DEPTH_KEY_FAR" (zeroed and read by gsr_preprocess_batch)
    uint32_t* dkeys[2];   // [P] depth bits (all ones for Gaussians that are not rendered), ping-pong
    uint32_t* dvals[2];   // [P] Gaussian index, ping-pong; the sorted one is the front-to-back order
    uint4* sbin;          // [P] {x0 | y0<<16, w | h<<16, Gaussian index, tiles} in front-to-back order
    uint32_t* order_sums; // [nblocks+1] tiles per block of 256 Gaussians in depth order, then offsets
    void* dsort_temp;
    size_t bytes;
};

inline GeomBuf geom_buf(const void* base, size_t P) {
    GeomBuf v;
    char* p = (char*)base;
    const size_t nb = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off = align_up(off + n, 256); return q; };
    v.rec = (SplatRec*)take(P * sizeof(SplatRec));
    v.cov3D = (float*)take(P * 6 * sizeof(float));
    v.rect = (uint2*)take(2 * P * sizeof(uint2));
    v.dt = v.rect ? v.rect + P : nullptr;
    v.gstat = (GStat*)take(P * sizeof(GStat));
    v.clamped = (uint8_t*)take(P);
    v.block_sums = (uint32_t*)take((nb + 1) * sizeof(uint32_t));
    v.vis_sums = (uint32_t*)take((nb + 1) * sizeof(uint32_t));
    v.total = (uint64_t*)take((4 + 3 * SCAN_BATCH_MAX + 1) * sizeof(uint64_t));
    v.dkeys[0] = (uint32_t*)take(P * sizeof(uint32_t));
    v.dkeys[1] = (uint32_t*)take(P * sizeof(uint32_t));
    v.dvals[0] = (uint32_t*)take(P * sizeof(uint32_t));
    v.dvals[1] = (uint32_t*)take(P * sizeof(uint32_t));
    v.sbin = (uint4*)take(P * sizeof(uint4));
    v.order_sums = (uint32_t*)take((nb + 1) * sizeof(uint32_t));
    v.dsort_temp = take(sort_temp_view(nullptr, P ? P : 1, SORT_BATCH_TEMP_PASSES).bytes);
    v.bytes = off;
    return v;
}

// Per-Gaussian gradient accumulator filled by the backward blend (atomics,
// one 64-byte line half per Gaussian) and consumed by the backward preprocess.
struct __attribute__((aligned(16))) GradRec {
    float mean2d_x, mean2d_y, conic_x, conic_y;
    float conic_w, opacity, col_r, col_g;
    float col_b, invdepth, rep0, rep1;
    float rep2, rep3, pad0, pad1;
};
static_assert(sizeof(GradRec) == 64, "GradRec must be 64 bytes");

// Feature ("reps") rows wider than four channels: the gradient sums of channels 4 .. DIM - 1 of every
// Gaussian live in a second array of 64-byte records behind the GradRecs of the backward work buffer.
constexpr int REPS_MAX = 16;
inline bool reps_dim_supported(int DIM) { return DIM == 0 || DIM == 4 || DIM == 8 || DIM == 16; }
inline size_t grad_rec_bytes(size_t P) { return align_up((P ? P : 1) * sizeof(GradRec), 256); }
inline float4* grad_rec_extra(const void* work, size_t P) {
    return work ? reinterpret_cast<float4*>((char*)work + grad_rec_bytes(P)) : nullptr;
}
inline size_t backward_work_bytes(size_t P, int DIM) { return grad_rec_bytes(P) * (DIM > 4 ? 2 : 1); }

int sort_passes_for(int begin_bit, int end_bit);
void render_fwd_set_reach_masks(int mode);   // test hook: 1 / 0 forces the forward blend with / without reach masks, -1 automatic
void sort_set_force_wide(bool on);   // test hook: 64-bit look-back status words at any size

hipError_t launch_preprocess_fwd(const GsrScene& s, const GsrForwardOut& o, const GeomBuf& g, hipStream_t st);
// The preprocess of several views in one pass over the Gaussians (everything launch_preprocess_fwd
// leaves in a view's geometry buffer except the 3D covariance).
// far_flag (device word, zeroed by the caller): set when a Gaussian in front of some view's near plane has a
// depth whose bits reach DEPTH_KEY_FAR.
hipError_t launch_preprocess_fwd_batch(const GsrScene& s, int nviews, const GsrForwardView* views, uint32_t* far_flag,
                                       hipStream_t st);
// Exclusive scans of the per-block tile sums and visible counts; totals into g.total.
hipError_t launch_scan_blocks(const GeomBuf& g, int nb, hipStream_t st);
// The same for `count` views (at most SCAN_BATCH_MAX) in one launch; *all: device array of 3 totals per view.
hipError_t launch_scan_blocks_batch(int count, const GeomBuf* g, int nb, hipStream_t st, uint64_t** all);
// Depth keys of the n Gaussians to order (n = number with instances: compacted; n = P: all of them),
// sorted front to back (stable): g.dvals[*order_buffer][0..n) is the order.
hipError_t launch_depth_order(const GeomBuf& g, int P, int n, hipStream_t st, int* order_buffer);
// The same for up to SORT_BATCH_MAX views of the same P Gaussians in shared launches (n[i] as above, each
// below 2^30); every view's order ends in its g.dvals[0].  narrow: every depth to order lies below
// DEPTH_KEY_FAR, so the keys are the depth bits minus DEPTH_KEY_BIAS on 27 bits (three passes instead of four;
// the same order).
hipError_t launch_depth_order_batch(int count, const GeomBuf* g, int P, const int* n, bool narrow, hipStream_t st);
// Key emission in depth order: (tile id, Gaussian index) pairs; only the tile remains to be sorted on.
hipError_t launch_emit_keys(const GsrScene& s, const GeomBuf& g, const uint32_t* order, int n, uint32_t* keys,
                            uint32_t* vals, hipStream_t st);
hipError_t launch_tile_ranges(const uint32_t* keys, int64_t R, uint2* ranges, int ntiles, hipStream_t st);
// (tile << 32 | depth bits) of every sorted instance, as the reference's sort keys.
hipError_t launch_keys64(const GeomBuf& g, const uint2* ranges, int ntiles, const uint32_t* point_list, int64_t R,
                         uint64_t* out, hipStream_t st);
// Unpacks the statistics; as the last kernel of the forward it also stores *sort_err | *depth_err (the error
// words of the instance sort and of the depth order; 0 for a null pointer) to *sort_err_host
// (device-accessible host memory, may be null).
hipError_t launch_finish_gstat(const GeomBuf& g, int P, float* gauss_sum, int* gauss_count, const uint32_t* sort_err,
                               const uint32_t* depth_err, uint32_t* sort_err_host, hipStream_t st);
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st);

// Returns the index (0/1) of the buffer pair that holds the sorted result.
hipError_t launch_radix_sort(uint64_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit, void* temp,
                             hipStream_t st, int* result_buffer);
// The same sort with its temp area in a device array of the library (no caller buffer, no allocation);
// n is limited to sort_static_max_pairs(); calls that overlap on one GPU race.
int64_t sort_static_max_pairs();
// Whether a sort of n pairs on these key bits fits the array (the 64-bit status words of the test hook need twice the room).
bool sort_static_fits(int64_t n, int begin_bit, int end_bit);
// Copies the error word the last such sort on the current device left (after everything enqueued on st) and waits.
hipError_t sort_static_read_error(hipStream_t st, uint32_t* err);
hipError_t launch_radix_sort_static(uint64_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit,
                                    hipStream_t st, int* result_buffer);
// Zeroes the sort's status area and returns where its digit histograms go (for a caller that fills them itself).
uint32_t* sort_prepare(void* temp, int64_t n, int passes, hipStream_t st, hipError_t* e);
// Tile ranges and the instance sort's digit histograms from the tile rectangles (image sizes whose
// tile grid fits the LDS of one workgroup: tile_counts_fit).
bool tile_counts_fit(int W, int H, int passes);
hipError_t launch_tile_ranges_from_rects(const GeomBuf& g, int P, int W, int H, const ImageView& im, uint32_t* ghist,
                                         int passes, int nbits, hipStream_t st, int lowbits = 0,
                                         uint32_t* seg_start = nullptr, uint32_t* seg_tile0 = nullptr);
// The tile rectangles in depth order (g.sbin) for the emission that follows.
hipError_t launch_order_gather(const GeomBuf& g, const uint32_t* order, int n, hipStream_t st);
// Key emission and instance sort in two kernels for two-digit tile ids (see radix_sort.hip): the emission
// partitions the instances by the high digit (into keys[1] / vals[1]), one segmented pass orders them by the low
// digit (into keys[0] / vals[0]).  fused_sort_fits: whether the sort's temp area is large enough for it;
// fused_sort_prepare zeroes what it needs there and returns the two segment tables for tile_finish.
bool fused_sort_fits(int64_t R, int n_visible, int nbits);
hipError_t fused_sort_prepare(void* temp, int64_t R, int n_visible, int nbits, hipStream_t st, uint32_t** ghist,
                              uint32_t** seg_start, uint32_t** seg_tile0);
// packed (fused_sort_packs: every Gaussian index fits below the low digit in 32 bits): between the two
// kernels an instance is ONE word, (low digit << (32 - digit bits)) | Gaussian index, in keys[1]; the
// segmented pass writes the Gaussian indices to vals[0] and nothing else (vals[1] and keys[0] stay unused).
bool fused_sort_packs(int P, int nbits);
// offsets_ready: launch_tile_finish_offsets has turned g.order_sums into offsets and written the chunk table
// (otherwise the call does both itself, in two one-workgroup launches).
hipError_t launch_emit_partition(const GeomBuf& g, int n_visible, int grid_x, int nbits, int64_t R, uint32_t* keys[2],
                                 uint32_t* vals[2], void* temp, hipStream_t st, bool packed, bool offsets_ready);
hipError_t launch_segmented_pass(int n_visible, int ntiles, int nbits, int64_t R, const uint2* ranges, uint32_t* keys[2],
                                 uint32_t* vals[2], void* temp, hipStream_t st, bool packed);
// The same with 32-bit keys.  hist_ready: the digit histograms in the temp area were filled by the caller.
hipError_t launch_radix_sort_u32_batch(int count, uint32_t* const (*keys)[2], uint32_t* const (*vals)[2], const int64_t* n,
                                       void* const* temp, hipStream_t st, int key_bits);
hipError_t launch_radix_sort_u32(uint32_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit,
                                 void* temp, hipStream_t st, int* result_buffer, bool hist_ready);

hipError_t launch_render_fwd(const GsrScene& s, const GsrForwardOut& o, const GeomBuf& g, const ImageView& im,
                             const uint32_t* point_list, int64_t num_rendered, int64_t num_visible, hipStream_t st);
hipError_t launch_render_bwd(const GsrScene& s, const GeomBuf& g, const ImageView& im, const uint32_t* point_list,
                             const GsrBackwardIO& io, GradRec* acc, hipStream_t st);
hipError_t launch_preprocess_bwd(const GsrScene& s, const int* radii, const GeomBuf& g, const GsrBackwardIO& io,
                                 const GradRec* acc, hipStream_t st);

hipError_t launch_means2d_grad(int P, const GeomBuf& g, const GradRec* acc, float* dL_dmeans2D, hipStream_t st);
// Zeroes the records of the Gaussians that the forward's hit count marks as reached; the rest stay undefined.
hipError_t launch_clear_reached(int P, const GeomBuf& g, GradRec* acc, int DIM, hipStream_t st);
// Gaussians [first, first + count) only; first is a multiple of 256.
hipError_t launch_preprocess_bwd_batch(const GsrScene& s, int nviews, const GsrBackwardView* views,
                                       const GsrBackwardIO& io, int first, int count, hipStream_t st);

hipError_t launch_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                       int64_t N, int M, const AdamConst& c, hipStream_t st);

hipError_t launch_densify_stats(int P, const int* radii, const float* dL_dmeans2D, float* xyz_gradient_accum,
                                float* denom, float* max_radii2D, hipStream_t st);

hipError_t launch_densify_select(int64_t N, int64_t n_grads, const float* xyz_gradient_accum, const float* denom,
                                 const float* scaling, float grad_threshold, float size_threshold, int split,
                                 uint8_t* selected, hipStream_t st);
hipError_t launch_prune_mask(int64_t N, const float* opacity, const float* max_radii2D, const float* scaling,
                             float min_opacity, float max_screen_size, float max_world_size, const uint8_t* also,
                             uint8_t* mask, hipStream_t st);
hipError_t launch_split_children(int64_t K, int n, const uint32_t* parents, const float* xyz, const float* scaling,
                                 const float* rotation, const float* normals, float* new_xyz, float* new_scaling,
                                 uint32_t* parent_of_child, hipStream_t st);
// counts: host array of n device pointers to [N] ints (4-byte aligned rows start 16-byte aligned); mask [N] bytes, 4-byte aligned
hipError_t launch_mark_nonzero(int64_t N, int n, const int* const* counts, uint8_t* mask, hipStream_t st);
hipError_t launch_scan_u32(uint32_t* sums, int nb, uint64_t* total, hipStream_t st);
// launch_tile_ranges_from_rects in two halves for the fused sort: the corner counts first; then, behind the
// ordered rectangle gather, one launch of two workgroups: the tile ranges / histograms / segment tables, and
// beside them the instance offsets of the depth-ordered blocks (in place in g.order_sums) with the emission's
// chunk table (fused_chunk_table).
hipError_t launch_tile_count(const GeomBuf& g, int P, int W, int H, const ImageView& im, hipStream_t st);
hipError_t launch_tile_finish_offsets(const GeomBuf& g, int n_visible, int W, int H, const ImageView& im, uint32_t* ghist,
                                      int passes, int nbits, int lowbits, uint32_t* seg_start, uint32_t* seg_tile0,
                                      uint2* chunk_first, uint32_t chunks, hipStream_t st);
// Where the emission's chunk table lives in the fused sort's temp area, and how many chunks there are.
void fused_chunk_table(void* temp, int64_t R, int n_visible, int nbits, uint2** chunk_first, uint32_t* chunks);
size_t compact_temp_bytes(int64_t N);
hipError_t launch_compact_map(int64_t N, const uint8_t* keep, uint32_t* src_of_dst, void* temp, hipStream_t st,
                              uint64_t** total_dev);
hipError_t launch_gather_rows(const void* src, void* dst, const uint32_t* src_of_dst, int64_t K, int M,
                              hipStream_t st);

hipError_t launch_knn(int P, const float* points, float* out, void* temp, hipStream_t st);
size_t knn_temp_bytes(int P);

hipError_t launch_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                           float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, hipStream_t st);
hipError_t launch_ssim_bwd(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                           float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                           hipStream_t st);

size_t loss_partial_floats(int planes, int H, int W);
size_t l1_partial_floats(int64_t n);
hipError_t launch_l1_fwd(int64_t n, const float* a, const float* b, float b_scalar, float* loss, float* partial,
                         hipStream_t st);
hipError_t launch_l1_bwd(int64_t n, const float* grad_loss, const float* a, const float* b, float b_scalar,
                         float* dL_da, hipStream_t st);
hipError_t launch_loss_fwd(int planes, int H, int W, float C1, float C2, float lambda, const float* img,
                           const float* gt, float* loss, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                           float* partial, hipStream_t st);
hipError_t launch_loss_bwd(int planes, int H, int W, float lambda, const float* grad_loss, const float* img,
                           const float* gt, float* dL_dimg, const float* dm_dmu1, const float* dm_dsigma1_sq,
                           const float* dm_dsigma12, hipStream_t st);

}  // namespace gs
