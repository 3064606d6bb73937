// extern "C" entry points of libgsplat_amd.so (declared in include/gsplat_amd.h).
// Host orchestration only: argument checks, scratch-buffer carving, kernel
// launches on the caller's stream.  Re-entrant, no allocation.  The only state the library keeps
// between calls: the test hooks below, which are per calling thread (a hook set in one thread
// changes nothing for calls made from another), and the per-device address of the array behind
// gsr_sort_pairs_u64_u32_noalloc (radix_sort.hip).
#include "kernels.h"

using namespace gs;

namespace {

thread_local int t_inject_sort_error = 0;  // test hook only (gsr_test_inject_sort_error): 1 instance sort, 2 depth sort
thread_local int t_packed_sort = 1;        // test hook only (gsr_test_packed_instances)
#ifndef GS_WIDE_DEPTH_KEYS
#define GS_WIDE_DEPTH_KEYS 0   // (1: builds a library for A/B timing that always orders by all 32 depth bits)
#endif
thread_local int t_wide_depth_keys = GS_WIDE_DEPTH_KEYS;    // test hook only (gsr_test_wide_depth_keys)

inline int hip_status(hipError_t e) { return e == hipSuccess ? GSR_OK : GSR_ERR_HIP; }

#define GSR_TRY(expr)                         \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return GSR_ERR_HIP; \
        if (debug) {                          \
            _e = hipStreamSynchronize(st);    \
            if (_e != hipSuccess) return GSR_ERR_HIP; \
        }                                     \
    } while (0)

inline void mark(const GsrScene* s, int slot, hipStream_t st) {
    if (s->profile_events) (void)hipEventRecord((hipEvent_t)s->profile_events[slot], st);
}

int check_scene(const GsrScene* s) {
    if (!s || s->P < 0 || s->W <= 0 || s->H <= 0) return GSR_ERR_ARG;
    if (!reps_dim_supported(s->DIM)) return GSR_ERR_DIM;
    if (s->P == 0) return GSR_OK;
    if (!s->means3D || !s->opacities || !s->viewmatrix || !s->projmatrix || !s->campos || !s->bg) return GSR_ERR_ARG;
    if ((s->shs == nullptr) == (s->colors_precomp == nullptr)) return GSR_ERR_ARG;
    const bool sr = s->scales && s->rotations;
    if (sr == (s->cov3D_precomp != nullptr)) return GSR_ERR_ARG;
    if (!sr && (s->scales || s->rotations)) return GSR_ERR_ARG;
    if (s->DIM > 0 && !s->reps) return GSR_ERR_ARG;
    if (s->shs && (s->M <= 0 || (s->D + 1) * (s->D + 1) > s->M || s->D < 0 || s->D > 3)) return GSR_ERR_ARG;
    if (s->shs_rest && (!s->shs || s->M < 2)) return GSR_ERR_ARG;
    if (s->raw_params && !sr) return GSR_ERR_ARG;  // raw mode activates scales and rotations
    if (s->W > 65535 * TILE_X || s->H > 65535 * TILE_Y) return GSR_ERR_ARG;
    return GSR_OK;
}

// Bits of a tile id (the reference sorts its 64-bit keys on 32 + this many bits, rasterizer_impl.cu:373).
int tile_bits(int W, int H) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    return (int)higher_msb((uint32_t)(gx * gy));
}

}  // namespace

extern "C" {

size_t gsr_geom_bytes(int P) { return geom_buf(nullptr, (size_t)(P > 0 ? P : 0)).bytes; }
size_t gsr_image_bytes(int W, int H) { return image_view(nullptr, (size_t)W, (size_t)H).bytes; }
size_t gsr_binning_bytes(int64_t R) { return bin_view(nullptr, (size_t)(R > 0 ? R : 0)).bytes; }
size_t gsr_struct_bytes(int which) {
    switch (which) {
        case 0: return sizeof(GsrScene);
        case 1: return sizeof(GsrForwardOut);
        case 2: return sizeof(GsrBackwardIO);
        case 3: return sizeof(GsrBackwardView);
        case 4: return sizeof(GsrForwardView);
        case 5: return sizeof(GsrLayout);
        default: return 0;
    }
}
size_t gsr_backward_work_bytes(int P) { return backward_work_bytes((size_t)(P > 0 ? P : 0), 4); }
size_t gsr_backward_work_bytes_dim(int P, int DIM) { return backward_work_bytes((size_t)(P > 0 ? P : 0), DIM); }
size_t gsr_sort_temp_error_offset(void) {
    char* const z = (char*)4096;
    return (size_t)((char*)sort_temp_view(z, 1, 0).err - z);
}
void gsr_test_force_wide_sort_status(int on) { sort_set_force_wide(on != 0); }
void gsr_test_forward_reach_masks(int mode) { render_fwd_set_reach_masks(mode); }
void gsr_test_inject_sort_error(int which) { t_inject_sort_error = which == 2 ? 2 : (which ? 1 : 0); }
void gsr_test_packed_instances(int on) { t_packed_sort = on ? 1 : 0; }
void gsr_test_wide_depth_keys(int on) { t_wide_depth_keys = on ? 1 : 0; }
size_t gsr_sort_temp_bytes(int64_t n) { return sort_temp_view(nullptr, (size_t)(n > 0 ? n : 1), SORT_MAX_PASSES).bytes; }

int gsr_forward_stage1(const GsrScene* s, GsrForwardOut* o, void* stream, int64_t* num_rendered) {
    int rc = check_scene(s);
    if (rc != GSR_OK) return rc;
    if (!o || !num_rendered) return GSR_ERR_ARG;
    *num_rendered = 0;
    o->num_visible = 0;
    if (s->P == 0) return GSR_OK;
    if (!o->radii || !o->out_depths || !o->geom_buffer) return GSR_ERR_ARG;
    if (o->geom_bytes < gsr_geom_bytes(s->P)) return GSR_ERR_BUFFER;
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    const GeomBuf g = geom_buf(o->geom_buffer, (size_t)s->P);
    mark(s, GSR_EV_STAGE1_BEGIN, st);
    // prefiltered: the view's flag word (g.total[3]) starts at zero; the preprocess raises it for a culled point
    const bool check_prefilter = s->prefiltered && !o->preprocessed;
    if (check_prefilter) GSR_TRY(hipMemsetAsync(g.total + 3, 0, sizeof(uint64_t), st));
    if (!o->preprocessed) GSR_TRY(launch_preprocess_fwd(*s, *o, g, st));
    mark(s, GSR_EV_PREPROCESS, st);
    // gsr_preprocess_batch has run this view's prefix sums already (they are in place: a second scan
    // would scan offsets) and left its three totals in g.total; only the copy is repeated then
    if (!o->preprocessed) GSR_TRY(launch_scan_blocks(g, (s->P + PRE_BLOCK - 1) / PRE_BLOCK, st));
    mark(s, GSR_EV_SCAN, st);
    uint64_t tot[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(tot, g.total, (check_prefilter ? 4 : 3) * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return GSR_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return GSR_ERR_HIP;
    if (check_prefilter && tot[3]) return GSR_ERR_PREFILTER;
    if (tot[1]) return GSR_ERR_OVERFLOW;
    *num_rendered = (int64_t)tot[0];
    o->num_visible = (int64_t)tot[2];
    return GSR_OK;
}

int gsr_forward_stage2(const GsrScene* s, const GsrForwardOut* o, int64_t R, void* binning_buffer, size_t binning_bytes,
                       void* stream) {
    int rc = check_scene(s);
    if (rc != GSR_OK) return rc;
    if (!o || R < 0) return GSR_ERR_ARG;
    if (R > (int64_t)INT32_MAX) return GSR_ERR_OVERFLOW;  // the reference holds this count in an int (rasterizer_impl.cu:338)
    if (!o->out_color || !o->accum_alpha || !o->last_contr_gauss || !o->image_buffer) return GSR_ERR_ARG;
    if (s->DIM > 0 && !o->out_reps) return GSR_ERR_ARG;
    if (o->image_bytes < gsr_image_bytes(s->W, s->H)) return GSR_ERR_BUFFER;
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    const ImageView im = image_view(o->image_buffer, (size_t)s->W, (size_t)s->H);
    const int ntiles = ((s->W + TILE_X - 1) / TILE_X) * ((s->H + TILE_Y - 1) / TILE_Y);
    if (s->P == 0) {
        // nothing to blend: background everywhere
        GSR_TRY(hipMemsetAsync(im.ranges, 0, (size_t)ntiles * sizeof(uint2), st));
    }
    GeomBuf g = geom_buf(s->P ? o->geom_buffer : nullptr, (size_t)s->P);
    const uint32_t* point_list = nullptr;
    if (s->P > 0) {
        if (!o->gauss_sum || !o->gauss_count || !o->geom_buffer) return GSR_ERR_ARG;
        if (R > 0) {
            if (!binning_buffer) return GSR_ERR_ARG;
            if (binning_bytes < gsr_binning_bytes(R)) return GSR_ERR_BUFFER;
        }
        BinView b = bin_view(binning_buffer, (size_t)R);
        int res = 0;
        mark(s, GSR_EV_STAGE2_BEGIN, st);
        const int nbits = tile_bits(s->W, s->H), passes = sort_passes_for(0, nbits);
        // the tile ranges and the sort's histograms from the tile rectangles, when the tile grid fits one
        // workgroup's LDS; otherwise from the keys (a histogram sweep before the sort, a range scan after it)
        const bool from_rects = R > 0 && tile_counts_fit(s->W, s->H, passes);
        // only the Gaussians with instances are ordered when stage 1 reported how many there are
        const int n = (o->num_visible > 0 && o->num_visible <= s->P) ? (int)o->num_visible : s->P;
        // two-digit tile ids: the emission partitions by the high digit, one segmented pass does the low one
        const bool fused = from_rects && fused_sort_fits(R, n, nbits);
        // ... and when every Gaussian index fits beside the low digit in 32 bits, an instance is one word
        // between the two kernels instead of a (tile id, index) pair
        const bool packed = fused && fused_sort_packs(s->P, nbits) && t_packed_sort;
        uint32_t *fused_ghist = nullptr, *fused_seg_start = nullptr, *fused_seg_tile0 = nullptr;
        int lowshift = 0, lowbits = 0;
        if (fused) {
            digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
            GSR_TRY(fused_sort_prepare(b.sort_temp, R, n, nbits, st, &fused_ghist, &fused_seg_start, &fused_seg_tile0));
            // the corner counts now; the ranges and segment tables in one launch with the instance offsets, behind
            // the ordered rectangle gather below (two one-workgroup steps side by side)
            GSR_TRY(launch_tile_count(g, s->P, s->W, s->H, im, st));
        } else if (from_rects) {
            hipError_t e = hipSuccess;
            uint32_t* ghist = sort_prepare(b.sort_temp, R, passes, st, &e);
            GSR_TRY(e);
            GSR_TRY(launch_tile_ranges_from_rects(g, s->P, s->W, s->H, im, ghist, passes, nbits, st));
        }
        mark(s, GSR_EV_TILE_COUNT, st);
        if (R > 0) {
            int ord = 0;   // gsr_preprocess_batch leaves the order it made in buffer 0
            const bool ordered = o->preprocessed == 2 && o->num_visible > 0 && o->num_visible <= s->P;
            if (!ordered) GSR_TRY(launch_depth_order(g, s->P, n, st, &ord));
            mark(s, GSR_EV_DEPTH_ORDER, st);
            if (fused) {
                GSR_TRY(launch_order_gather(g, g.dvals[ord], n, st));
                uint2* chunk_first = nullptr;
                uint32_t chunks = 0;
                fused_chunk_table(b.sort_temp, R, n, nbits, &chunk_first, &chunks);
                GSR_TRY(launch_tile_finish_offsets(g, n, s->W, s->H, im, fused_ghist, passes, nbits, lowbits, fused_seg_start,
                                                   fused_seg_tile0, chunk_first, chunks, st));
                GSR_TRY(launch_emit_partition(g, n, (s->W + TILE_X - 1) / TILE_X, nbits, R, b.keys, b.vals, b.sort_temp, st,
                                              packed, true));
                mark(s, GSR_EV_EMIT, st);
                GSR_TRY(launch_segmented_pass(n, ntiles, nbits, R, im.ranges, b.keys, b.vals, b.sort_temp, st, packed));
                res = 0;
            } else {
                GSR_TRY(launch_emit_keys(*s, g, g.dvals[ord], n, b.keys[0], b.vals[0], st));
                mark(s, GSR_EV_EMIT, st);
                // emission order is (depth, index): the stable sort only has to order the tile ids
                GSR_TRY(launch_radix_sort_u32(b.keys, b.vals, R, 0, nbits, b.sort_temp, st, &res, from_rects));
            }
            mark(s, GSR_EV_SORT, st);
        }
        if (!from_rects) GSR_TRY(launch_tile_ranges(b.keys[res], R, im.ranges, ntiles, st));
        mark(s, GSR_EV_RANGES, st);
        point_list = b.vals[res];
    }
    GSR_TRY(launch_render_fwd(*s, *o, g, im, point_list, R, o->num_visible, st));
    mark(s, GSR_EV_RENDER, st);
    // The sorts' look-back spins are bounded; a tripped limit sets an error word in the sort's temp area.
    // With debug the call waits and reports it; otherwise the last kernel stores the word to the caller's
    // host word in stream order (nothing here waits), or it stays unread if the caller gave none.
    // Two sorts ran for this view when it has instances: the instance sort (error word in the binning
    // buffer's sort area) and the front-to-back order of the Gaussians (error word in the geometry buffer's
    // sort area, written by this call or, for preprocessed == 2, by gsr_preprocess_batch).
    const uint32_t *sort_err = nullptr, *depth_err = nullptr;
    if (R > 0 && s->P > 0) {
        sort_err = sort_temp_view(bin_view(binning_buffer, (size_t)R).sort_temp, 1, 0).err;
        depth_err = sort_temp_view(g.dsort_temp, 1, 0).err;
        if (t_inject_sort_error &&
            hipMemsetAsync((void*)(t_inject_sort_error == 2 ? depth_err : sort_err), 1, 1, st) != hipSuccess)
            return GSR_ERR_HIP;
    }
    if (s->P > 0) {
        GSR_TRY(launch_finish_gstat(g, s->P, o->gauss_sum, o->gauss_count, sort_err, depth_err,
                                    debug ? nullptr : o->sort_error_host, st));
        mark(s, GSR_EV_FINISH, st);
    } else if (o->sort_error_host) {
        *o->sort_error_host = 0;  // nothing was sorted, nothing runs: the host word is set here
    }
    if (debug && sort_err) {
        uint32_t err[2] = {0, 0};
        if (hipMemcpyAsync(&err[0], sort_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return GSR_ERR_HIP;
        if (hipMemcpyAsync(&err[1], depth_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return GSR_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return GSR_ERR_HIP;
        if (err[0] | err[1]) return GSR_ERR_SORT;
    }
    if (debug && o->sort_error_host) *o->sort_error_host = 0;  // reported (or clean) through the return value
    return GSR_OK;
}

static int backward_blend(const GsrScene* s, const int* radii, const void* geom_buffer, const void* binning_buffer,
                          const void* image_buffer, int64_t R, const GsrBackwardIO* io, hipStream_t st) {
    if (!radii || !geom_buffer || !image_buffer || !io->dL_dout_color || !io->work) return GSR_ERR_ARG;
    if (s->DIM > 0 && !io->dL_dout_reps) return GSR_ERR_ARG;
    if (io->work_bytes < gsr_backward_work_bytes_dim(s->P, s->DIM)) return GSR_ERR_BUFFER;
    if (R > 0 && !binning_buffer) return GSR_ERR_ARG;
    const bool debug = s->debug != 0;
    const GeomBuf g = geom_buf(geom_buffer, (size_t)s->P);
    const ImageView im = image_view((void*)image_buffer, (size_t)s->W, (size_t)s->H);
    GradRec* acc = (GradRec*)io->work;
    mark(s, GSR_EV_BWD_BEGIN, st);
    GSR_TRY(launch_clear_reached(s->P, g, acc, s->DIM, st));
    mark(s, GSR_EV_BWD_CLEAR, st);
    if (R > 0) {
        const BinView b = bin_view((void*)binning_buffer, (size_t)R);
        const int res = sort_passes_for(0, tile_bits(s->W, s->H)) & 1;
        GSR_TRY(launch_render_bwd(*s, g, im, b.vals[res], *io, acc, st));
    }
    mark(s, GSR_EV_BWD_RENDER, st);
    return GSR_OK;
}

int gsr_backward(const GsrScene* s, const int* radii, const void* geom_buffer, const void* binning_buffer,
                 const void* image_buffer, int64_t R, const GsrBackwardIO* io, void* stream) {
    int rc = check_scene(s);
    if (rc != GSR_OK) return rc;
    if (!io) return GSR_ERR_ARG;
    if (s->P == 0) return GSR_OK;
    if (!io->dL_dmeans2D || !io->dL_dopacity || !io->dL_dmeans3D) return GSR_ERR_ARG;
    // the two intermediate gradients are only outputs when the matching precomputed input was given
    if (s->colors_precomp && !io->dL_dcolors) return GSR_ERR_ARG;
    if (s->cov3D_precomp && !io->dL_dcov3D) return GSR_ERR_ARG;
    if (s->M > 0 && s->shs && !io->dL_dsh) return GSR_ERR_ARG;
    if (s->shs_rest && !io->dL_dsh_rest) return GSR_ERR_ARG;
    if (s->scales && (!io->dL_dscales || !io->dL_drotations)) return GSR_ERR_ARG;
    if (s->DIM > 0 && !io->dL_dreps) return GSR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    rc = backward_blend(s, radii, geom_buffer, binning_buffer, image_buffer, R, io, st);
    if (rc != GSR_OK) return rc;
    const GeomBuf g = geom_buf(geom_buffer, (size_t)s->P);
    GSR_TRY(launch_preprocess_bwd(*s, radii, g, *io, (const GradRec*)io->work, st));
    mark(s, GSR_EV_BWD_PREPROCESS, st);
    return GSR_OK;
}

int gsr_backward_blend(const GsrScene* s, const int* radii, const void* geom_buffer, const void* binning_buffer,
                       const void* image_buffer, int64_t R, const GsrBackwardIO* io, void* stream) {
    int rc = check_scene(s);
    if (rc != GSR_OK) return rc;
    if (!io) return GSR_ERR_ARG;
    if (s->P == 0) return GSR_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    rc = backward_blend(s, radii, geom_buffer, binning_buffer, image_buffer, R, io, st);
    if (rc != GSR_OK) return rc;
    if (io->dL_dmeans2D) GSR_TRY(launch_means2d_grad(s->P, geom_buf(geom_buffer, (size_t)s->P), (const GradRec*)io->work, io->dL_dmeans2D, st));
    mark(s, GSR_EV_BWD_PREPROCESS, st);
    return GSR_OK;
}

int gsr_backward_geometry_batch(const GsrScene* s, int nviews, const GsrBackwardView* views, const GsrBackwardIO* io,
                                void* stream) {
    if (!s) return GSR_ERR_ARG;
    return gsr_backward_geometry_batch_range(s, nviews, views, io, 0, s->P, stream);
}

int gsr_backward_geometry_batch_range(const GsrScene* s, int nviews, const GsrBackwardView* views,
                                      const GsrBackwardIO* io, int first, int count, void* stream) {
    if (!s || !io || nviews < 0 || (nviews > 0 && !views)) return GSR_ERR_ARG;
    if (s->P < 0) return GSR_ERR_ARG;
    if (first < 0 || count < 0 || (first % PRE_BLOCK) != 0 || (int64_t)first + count > s->P) return GSR_ERR_ARG;
    // the block states (row_blocks, means2D_blocks) are kept per 64 rows: a range ends on a block boundary
    // or at the last Gaussian, so that no call updates the state of a block it handled only part of
    if ((count % 64) != 0 && (int64_t)first + count != s->P) return GSR_ERR_ARG;
    if (!reps_dim_supported(s->DIM)) return GSR_ERR_DIM;
    if (s->P == 0 || nviews == 0 || count == 0) return GSR_OK;
    if (!s->means3D || !s->opacities || !s->scales || !s->rotations || !s->shs || !s->shs_rest) return GSR_ERR_ARG;
    if (s->M != 16 || s->D < 0 || s->D > 3 || s->cov3D_precomp || s->colors_precomp) return GSR_ERR_ARG;
    if (!io->dL_dopacity || !io->dL_dmeans3D || !io->dL_dsh || !io->dL_dsh_rest || !io->dL_dscales ||
        !io->dL_drotations)
        return GSR_ERR_ARG;
    if (s->DIM > 0 && !io->dL_dreps) return GSR_ERR_ARG;
    for (int v = 0; v < nviews; ++v) {
        const GsrBackwardView& gv = views[v];
        if (!gv.viewmatrix || !gv.projmatrix || !gv.campos || !gv.radii || !gv.geom_buffer || !gv.work ||
            gv.W <= 0 || gv.H <= 0)
            return GSR_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    mark(s, GSR_EV_BATCH_BEGIN, st);
    GSR_TRY(launch_preprocess_bwd_batch(*s, nviews, views, *io, first, count, st));
    mark(s, GSR_EV_BATCH, st);
    return GSR_OK;
}

int gsr_preprocess_batch(const GsrScene* s, int nviews, GsrForwardView* views, void* stream) {
    if (!s || nviews < 0 || (nviews > 0 && !views)) return GSR_ERR_ARG;
    if (s->P < 0) return GSR_ERR_ARG;
    if (!reps_dim_supported(s->DIM)) return GSR_ERR_DIM;
    if (s->P == 0 || nviews == 0) return GSR_OK;
    if (!s->means3D || !s->opacities || !s->scales || !s->rotations || !s->shs || !s->shs_rest) return GSR_ERR_ARG;
    if (s->M != 16 || s->D < 0 || s->D > 3 || s->cov3D_precomp || s->colors_precomp) return GSR_ERR_ARG;
    if (s->DIM > 0 && !s->reps) return GSR_ERR_ARG;
    for (int v = 0; v < nviews; ++v) {
        GsrForwardView& fv = views[v];
        fv.num_rendered = fv.num_visible = 0;
        fv.depth_ordered = 0;
        if (!fv.viewmatrix || !fv.projmatrix || !fv.campos || !fv.radii || !fv.out_depths || !fv.geom_buffer ||
            fv.W <= 0 || fv.H <= 0 || fv.W > 65535 * TILE_X || fv.H > 65535 * TILE_Y)
            return GSR_ERR_ARG;
        if (fv.geom_bytes < gsr_geom_bytes(s->P)) return GSR_ERR_BUFFER;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool debug = s->debug != 0;
    mark(s, GSR_EV_FWD_BATCH_BEGIN, st);
    // one word for the whole call, in the first view's buffer: "some depth reaches DEPTH_KEY_FAR"
    uint64_t* const far_flag = geom_buf(views[0].geom_buffer, (size_t)s->P).total + 4 + 3 * SCAN_BATCH_MAX;
    GSR_TRY(hipMemsetAsync(far_flag, 0, sizeof(uint64_t), st));
    GSR_TRY(launch_preprocess_fwd_batch(*s, nviews, views, reinterpret_cast<uint32_t*>(far_flag), st));
    mark(s, GSR_EV_FWD_BATCH, st);
    // prefix sums of every view, then all the totals with one host block
    // (the scans of up to 16 views in one launch, their totals collected for one copy)
    constexpr int CHUNK = SCAN_BATCH_MAX;
    uint64_t far = 1;
    for (int v0 = 0; v0 < nviews; v0 += CHUNK) {
        const int n = nviews - v0 < CHUNK ? nviews - v0 : CHUNK;
        uint64_t tot[CHUNK][3];
        GeomBuf gb[CHUNK];
        if (v0 == 0 && hipMemcpyAsync(&far, far_flag, sizeof(far), hipMemcpyDeviceToHost, st) != hipSuccess)
            return GSR_ERR_HIP;   // (completes with the wait for the first chunk's totals)
        for (int k = 0; k < n; ++k) gb[k] = geom_buf(views[v0 + k].geom_buffer, (size_t)s->P);
        uint64_t* all = nullptr;
        GSR_TRY(launch_scan_blocks_batch(n, gb, (s->P + PRE_BLOCK - 1) / PRE_BLOCK, st, &all));
        if (hipMemcpyAsync(tot, all, (size_t)n * sizeof(tot[0]), hipMemcpyDeviceToHost, st) != hipSuccess)
            return GSR_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return GSR_ERR_HIP;
        for (int k = 0; k < n; ++k) {
            if (tot[k][1]) return GSR_ERR_OVERFLOW;
            views[v0 + k].num_rendered = (int64_t)tot[k][0];
            views[v0 + k].num_visible = (int64_t)tot[k][2];
        }
    }
#ifndef GS_NO_BATCH_DEPTH_ORDER
    // the front-to-back orders of the views that have instances, up to SORT_BATCH_MAX views per set of launches
    mark(s, GSR_EV_FWD_BATCH_SORT_BEGIN, st);
    for (int v0 = 0; v0 < nviews;) {
        GeomBuf gb[SORT_BATCH_MAX];
        int nvis[SORT_BATCH_MAX], which[SORT_BATCH_MAX], m = 0;
        for (; v0 < nviews && m < SORT_BATCH_MAX; ++v0) {
            const GsrForwardView& fv = views[v0];
            if (fv.num_rendered <= 0 || fv.num_visible <= 0 || fv.num_visible > s->P ||
                sort_status_wide((size_t)fv.num_visible))
                continue;
            gb[m] = geom_buf(fv.geom_buffer, (size_t)s->P);
            nvis[m] = (int)fv.num_visible;
            which[m++] = v0;
        }
        if (m == 0) continue;
        const bool narrow = far == 0 && !t_wide_depth_keys;
        GSR_TRY(launch_depth_order_batch(m, gb, s->P, nvis, narrow, st));
        for (int k = 0; k < m; ++k) views[which[k]].depth_ordered = narrow ? 2 : 1;
    }
    mark(s, GSR_EV_FWD_BATCH_SORT, st);
#endif
    return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream) {
    (void)projmatrix;
    if (P < 0) return GSR_ERR_ARG;
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !present) return GSR_ERR_ARG;
    return hip_status(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream));
}

int gsr_sort_pairs_u64_u32(uint64_t* keys0, uint32_t* vals0, uint64_t* keys1, uint32_t* vals1, int64_t n, int end_bit,
                           void* temp, size_t temp_bytes, void* stream, int* result_buffer) {
    if (n < 0 || end_bit < 0 || end_bit > 64 || !result_buffer) return GSR_ERR_ARG;
    *result_buffer = 0;
    if (n == 0 || end_bit == 0) return GSR_OK;
    if (n > (int64_t)INT32_MAX) return GSR_ERR_OVERFLOW;
    if (!keys0 || !vals0 || !keys1 || !vals1 || !temp) return GSR_ERR_ARG;
    if (temp_bytes < gsr_sort_temp_bytes(n)) return GSR_ERR_BUFFER;
    uint64_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    return hip_status(launch_radix_sort(k, v, n, 0, end_bit, temp, (hipStream_t)stream, result_buffer));
}

int64_t gsr_sort_noalloc_max_pairs(void) { return sort_static_max_pairs(); }

int gsr_sort_pairs_u64_u32_noalloc(uint64_t* keys0, uint32_t* vals0, uint64_t* keys1, uint32_t* vals1, int64_t n,
                                   int end_bit, void* stream, int* result_buffer) {
    if (n < 0 || end_bit < 0 || end_bit > 64 || !result_buffer) return GSR_ERR_ARG;
    *result_buffer = 0;
    if (n == 0 || end_bit == 0) return GSR_OK;
    if (!sort_static_fits(n, 0, end_bit)) return GSR_ERR_LIMIT;
    if (!keys0 || !vals0 || !keys1 || !vals1) return GSR_ERR_ARG;
    uint64_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    return hip_status(launch_radix_sort_static(k, v, n, 0, end_bit, (hipStream_t)stream, result_buffer));
}

int gsr_sort_noalloc_error(void* stream, uint32_t* err) {
    if (!err) return GSR_ERR_ARG;
    *err = 0;
    return hip_status(sort_static_read_error((hipStream_t)stream, err));
}

int gsr_layout(int P, int W, int H, int64_t R, GsrLayout* out) {
    if (!out || P < 0 || W <= 0 || H <= 0 || R < 0) return GSR_ERR_ARG;
    char* const z = (char*)4096;  // fake non-null base to turn pointers into offsets
    const GeomBuf g = geom_buf(z, (size_t)P);
    out->rec = (char*)g.rec - z;
    out->cov3D = (char*)g.cov3D - z;
    out->tiles = (char*)g.rect - z;
    out->clamped = (char*)g.clamped - z;
    out->block_sums = (char*)g.block_sums - z;
    out->total = (char*)g.total - z;
    out->depth_sort = (char*)g.dsort_temp - z;
    const ImageView im = image_view(z, (size_t)W, (size_t)H);
    out->final_T = (char*)im.final_T - z;
    out->n_contrib = (char*)im.n_contrib - z;
    out->ranges = (char*)im.ranges - z;
    const BinView b = bin_view(z, (size_t)R);
    out->keys0 = (char*)b.keys[0] - z;
    out->keys1 = (char*)b.keys[1] - z;
    out->vals0 = (char*)b.vals[0] - z;
    out->vals1 = (char*)b.vals[1] - z;
    out->sorted_buffer = sort_passes_for(0, tile_bits(W, H)) & 1;
    return GSR_OK;
}

int gsr_sorted_keys_u64(int P, int W, int H, int64_t R, const void* geom_buffer, const void* binning_buffer,
                        const void* image_buffer, uint64_t* keys_out, void* stream) {
    if (P < 0 || W <= 0 || H <= 0 || R < 0) return GSR_ERR_ARG;
    if (R == 0) return GSR_OK;
    if (!geom_buffer || !binning_buffer || !image_buffer || !keys_out) return GSR_ERR_ARG;
    const GeomBuf g = geom_buf(geom_buffer, (size_t)P);
    const BinView b = bin_view((void*)binning_buffer, (size_t)R);
    const ImageView im = image_view((void*)image_buffer, (size_t)W, (size_t)H);
    const int ntiles = ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    const int res = sort_passes_for(0, tile_bits(W, H)) & 1;
    return hip_status(launch_keys64(g, im.ranges, ntiles, b.vals[res], R, keys_out, (hipStream_t)stream));
}

size_t gsk_dist2_temp_bytes(int P) { return knn_temp_bytes(P); }

int gsk_dist2(int P, const float* points, float* mean_dists, void* temp, size_t temp_bytes, void* stream) {
    if (P < 0) return GSR_ERR_ARG;
    if (P == 0) return GSR_OK;
    if (!points || !mean_dists || !temp) return GSR_ERR_ARG;
    if (P >= (1 << 30)) return GSR_ERR_OVERFLOW;
    if (temp_bytes < knn_temp_bytes(P)) return GSR_ERR_BUFFER;
    return hip_status(launch_knn(P, points, mean_dists, temp, (hipStream_t)stream));
}

int gss_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (B < 0 || CH < 0 || H < 0 || W < 0) return GSR_ERR_ARG;
    if ((size_t)B * CH * H * W == 0) return GSR_OK;
    if (!img1 || !img2 || !ssim_map) return GSR_ERR_ARG;
    if (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) return GSR_ERR_ARG;
    return hip_status(launch_ssim_fwd(B, CH, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                      (hipStream_t)stream));
}

int gss_ssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                      const float* dL_dmap, float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq,
                      const float* dm_dsigma12, void* stream) {
    (void)C1;
    (void)C2;
    if (B < 0 || CH < 0 || H < 0 || W < 0) return GSR_ERR_ARG;
    if ((size_t)B * CH * H * W == 0) return GSR_OK;
    if (!img1 || !img2 || !dL_dmap || !dL_dimg1 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12) return GSR_ERR_ARG;
    return hip_status(launch_ssim_bwd(B, CH, H, W, img1, img2, dL_dmap, dL_dimg1, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                      (hipStream_t)stream));
}

size_t gss_loss_partial_floats(int planes, int H, int W) {
    return (planes > 0 && H > 0 && W > 0) ? loss_partial_floats(planes, H, W) : 0;
}

int gss_l1_ssim_loss_forward(int planes, int H, int W, float C1, float C2, float lambda_dssim, const float* img,
                             const float* gt, float* loss, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                             float* partial, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    if (!img || !gt || !loss || !partial) return GSR_ERR_ARG;
    if (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) return GSR_ERR_ARG;
    if (loss_partial_floats(planes, H, W) / 2 >= (1u << 30)) return GSR_ERR_OVERFLOW;
    return hip_status(launch_loss_fwd(planes, H, W, C1, C2, lambda_dssim, img, gt, loss, dm_dmu1, dm_dsigma1_sq,
                                      dm_dsigma12, partial, (hipStream_t)stream));
}

int gss_l1_ssim_loss_backward(int planes, int H, int W, float lambda_dssim, const float* grad_loss, const float* img,
                              const float* gt, float* dL_dimg, const float* dm_dmu1, const float* dm_dsigma1_sq,
                              const float* dm_dsigma12, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return GSR_ERR_ARG;
    if (!grad_loss || !img || !gt || !dL_dimg || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12) return GSR_ERR_ARG;
    return hip_status(launch_loss_bwd(planes, H, W, lambda_dssim, grad_loss, img, gt, dL_dimg, dm_dmu1, dm_dsigma1_sq,
                                      dm_dsigma12, (hipStream_t)stream));
}

size_t gss_l1_partial_floats(int64_t n) { return n > 0 ? l1_partial_floats(n) : 0; }

int gss_l1_loss_forward(int64_t n, const float* a, const float* b, float b_scalar, float* loss, float* partial,
                        void* stream) {
    if (n <= 0) return GSR_ERR_ARG;
    if (!a || !loss || !partial) return GSR_ERR_ARG;
    return hip_status(launch_l1_fwd(n, a, b, b_scalar, loss, partial, (hipStream_t)stream));
}

int gss_l1_loss_backward(int64_t n, const float* grad_loss, const float* a, const float* b, float b_scalar,
                         float* dL_da, void* stream) {
    if (n <= 0) return GSR_ERR_ARG;
    if (!grad_loss || !a || !dL_da) return GSR_ERR_ARG;
    if ((n + 255) / 256 >= (1ll << 31)) return GSR_ERR_OVERFLOW;
    return hip_status(launch_l1_bwd(n, grad_loss, a, b, b_scalar, dL_da, (hipStream_t)stream));
}

int gso_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                    int64_t N, int M, double lr, double beta1, double beta2, double eps, double bias_correction1,
                    double bias_correction2_sqrt, void* stream) {
    if (N < 0 || M <= 0) return GSR_ERR_ARG;
    if (N == 0) return GSR_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return GSR_ERR_ARG;
    if (!(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0)) return GSR_ERR_ARG;
    if (N * (int64_t)M >= (1ll << 40)) return GSR_ERR_OVERFLOW;
    return hip_status(launch_adam(param, grad, exp_avg, exp_avg_sq, visible, N, M,
                                  adam_const(lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt),
                                  (hipStream_t)stream));
}

int gso_densify_stats(int P, const int* radii, const float* dL_dmeans2D, float* xyz_gradient_accum, float* denom,
                      float* max_radii2D, void* stream) {
    if (P < 0) return GSR_ERR_ARG;
    if (P == 0) return GSR_OK;
    if (!radii || !dL_dmeans2D || !xyz_gradient_accum || !denom || !max_radii2D) return GSR_ERR_ARG;
    return hip_status(launch_densify_stats(P, radii, dL_dmeans2D, xyz_gradient_accum, denom, max_radii2D,
                                           (hipStream_t)stream));
}

int gso_densify_select(int64_t N, int64_t n_grads, const float* xyz_gradient_accum, const float* denom,
                       const float* scaling, float grad_threshold, float size_threshold, int split, uint8_t* selected,
                       void* stream) {
    if (N < 0 || n_grads < 0 || n_grads > N) return GSR_ERR_ARG;
    if (N == 0) return GSR_OK;
    if (!scaling || !selected || (n_grads > 0 && (!xyz_gradient_accum || !denom))) return GSR_ERR_ARG;
    return hip_status(launch_densify_select(N, n_grads, xyz_gradient_accum, denom, scaling, grad_threshold,
                                            size_threshold, split ? 1 : 0, selected, (hipStream_t)stream));
}

int gso_prune_mask(int64_t N, const float* opacity, const float* max_radii2D, const float* scaling, float min_opacity,
                   float max_screen_size, float max_world_size, const uint8_t* also, uint8_t* mask, void* stream) {
    if (N < 0) return GSR_ERR_ARG;
    if (N == 0) return GSR_OK;
    if (!opacity || !mask) return GSR_ERR_ARG;
    if (max_screen_size > 0.0f && (!max_radii2D || !scaling)) return GSR_ERR_ARG;
    return hip_status(launch_prune_mask(N, opacity, max_radii2D, scaling, min_opacity, max_screen_size, max_world_size,
                                        also, mask, (hipStream_t)stream));
}

int gso_split_children(int64_t K, int n, const uint32_t* parents, const float* xyz, const float* scaling,
                       const float* rotation, const float* normals, float* new_xyz, float* new_scaling,
                       uint32_t* parent_of_child, void* stream) {
    if (K < 0 || n <= 0) return GSR_ERR_ARG;
    if (K == 0) return GSR_OK;
    if (K * (int64_t)n >= (1ll << 31)) return GSR_ERR_OVERFLOW;
    if (!parents || !xyz || !scaling || !rotation || !normals || !new_xyz || !new_scaling || !parent_of_child)
        return GSR_ERR_ARG;
    return hip_status(launch_split_children(K, n, parents, xyz, scaling, rotation, normals, new_xyz, new_scaling,
                                            parent_of_child, (hipStream_t)stream));
}

int gso_mark_nonzero(int64_t N, int n, const int* const* counts, uint8_t* mask, void* stream) {
    if (N < 0 || n < 0) return GSR_ERR_ARG;
    if (N == 0) return GSR_OK;
    if (!mask || (n > 0 && !counts)) return GSR_ERR_ARG;
    if (((uintptr_t)mask & 3u) != 0) return GSR_ERR_ARG;
    for (int v = 0; v < n; ++v)
        if (!counts[v] || ((uintptr_t)counts[v] & 15u) != 0) return GSR_ERR_ARG;
    return hip_status(launch_mark_nonzero(N, n, counts, mask, (hipStream_t)stream));
}

size_t gso_compact_temp_bytes(int64_t N) { return compact_temp_bytes(N > 0 ? N : 1); }

int gso_compact_map(int64_t N, const uint8_t* keep, uint32_t* src_of_dst, void* temp, size_t temp_bytes, int64_t* kept,
                    void* stream) {
    if (N < 0 || !kept) return GSR_ERR_ARG;
    *kept = 0;
    if (N == 0) return GSR_OK;
    if (N >= (1ll << 31)) return GSR_ERR_OVERFLOW;
    if (!keep || !src_of_dst || !temp) return GSR_ERR_ARG;
    if (temp_bytes < compact_temp_bytes(N)) return GSR_ERR_BUFFER;
    hipStream_t st = (hipStream_t)stream;
    uint64_t* total_dev = nullptr;
    if (launch_compact_map(N, keep, src_of_dst, temp, st, &total_dev) != hipSuccess) return GSR_ERR_HIP;
    uint64_t tot[2] = {0, 0};
    if (hipMemcpyAsync(tot, total_dev, sizeof(tot), hipMemcpyDeviceToHost, st) != hipSuccess) return GSR_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return GSR_ERR_HIP;
    *kept = (int64_t)tot[0];
    return GSR_OK;
}

int gso_gather_rows(const void* src, void* dst, const uint32_t* src_of_dst, int64_t K, int M, void* stream) {
    if (K < 0 || M <= 0) return GSR_ERR_ARG;
    if (K == 0) return GSR_OK;
    if (!src || !dst || !src_of_dst) return GSR_ERR_ARG;
    if (K * (int64_t)M >= (1ll << 40)) return GSR_ERR_OVERFLOW;
    return hip_status(launch_gather_rows(src, dst, src_of_dst, K, M, (hipStream_t)stream));
}

const char* gsr_status_string(int status) {
    switch (status) {
        case GSR_OK: return "ok";
        case GSR_ERR_ARG: return "invalid argument";
        case GSR_ERR_DIM: return "unsupported number of reps channels (supported: 0, 4, 8, 16)";
        case GSR_ERR_OVERFLOW: return "too many Gaussian/tile instances (more than 2^31 - 1)";
        case GSR_ERR_BUFFER: return "caller buffer too small";
        case GSR_ERR_HIP: return "HIP runtime error";
        case GSR_ERR_SORT: return "radix sort look-back did not complete";
        case GSR_ERR_PREFILTER: return "a point was culled although the caller declared the points prefiltered";
        case GSR_ERR_LIMIT: return "input larger than the fixed scratch of the entry point that does not allocate";
        default: return "unknown status";
    }
}

}  // extern "C"
