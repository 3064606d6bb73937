// Per-Gaussian backward: conic -> cov2D -> cov3D / mean (EWA), 2D mean -> 3D
// mean, colour -> SH, cov3D -> scale / rotation.  Reference:
// cuda_rasterizer/backward.cu:147-329 (computeCov2DCUDA) and :401-471
// (preprocessCUDA), which run as two kernels over the same data; here they are
// one pass, one lane per Gaussian.  HBM-bound: reads the 64-byte GradRec the
// blend backward filled plus the Gaussian's inputs and writes every gradient
// tensor completely (zeros for Gaussians that were not rendered), so the caller
// never zero-fills; in accumulate mode it adds into the parameter gradients
// and leaves the rows of unrendered Gaussians alone.
//
// SH rows (192 bytes in, 192 bytes out per Gaussian with 16 coefficients) are
// the bulk of the traffic.  Three ways to move them:
//   SH_CAT16    one [P,16,3] tensor: each wave moves its 64 rows as 16-byte
//               chunks in address order through a padded LDS image
//               (sh_rows_load / sh_rows_store);
//   SH_SPLIT16  [P,1,3] + [P,15,3] tensors: the wave's 64 x 180 bytes of the
//               second are contiguous and move as 16-byte chunks through a
//               linear LDS image (rest_rows_copy); a lane reads and writes its
//               row dword by dword (row stride 45 dwords is odd: conflict-free);
//   SH_DIRECT   any other size: each lane walks its own row in global memory.
#include "kernels.h"

namespace gs {

struct PreBwdArgs {
    int P, D, M, DIM, W, H;
    float mod, tanx, tany;
    int aa;
    int raw;         // opacities / scales / rotations are pre-activation values
    int accumulate;  // add into the parameter gradients
    const float* means3D;
    const float* shs;
    const float* shs_rest;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D;  // the precomputed 3D covariance, if that is what the caller gave
    CamPtrs cam;
    const int* radii;
    const GStat* gstat;   // the forward's per-Gaussian (sum of T, pixels that reached it)
    const uint8_t* clamped;
    const GradRec* acc;
    const float4* accx;   // [P][4] float4: sums of the feature channels 4 .. DIM - 1 (DIM > 4), else unused
    int has_invdepth;
    float* dL_dmeans2D;
    float* dL_dcolors;
    float* dL_dopacity;
    float* dL_dmeans3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dsh_rest;
    float* dL_dscales;
    float* dL_drotations;
    float* dL_dreps;
    uint8_t* row_blocks;  // batch kernel only, may be null: see GsrBackwardIO.row_blocks
};


// dL/dreps row of Gaussian i (DIM = 4, 8 or 16 floats): channels 0 .. 3 from the GradRec (`lo`), the others
// from the second record (x[0 .. DIM / 4 - 2]).
// (a fixed trip count with guards: `x` may be an array in registers, which must not be indexed dynamically)
__device__ __forceinline__ void reps_row_out(float* dL_dreps, size_t i, int DIM, bool accum, float4 lo, const float4* x) {
    float4* p = reinterpret_cast<float4*>(dL_dreps + i * (size_t)DIM);
#pragma unroll
    for (int e = 0; e < REPS_MAX / 4; ++e) {
        if (4 * e >= DIM) break;
        const float4 v = e == 0 ? lo : x[e - 1];
        if (accum) {
            const float4 o = p[e];
            p[e] = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        } else {
            p[e] = v;
        }
    }
}

// Everything for one rendered Gaussian.
template <int MODE>
__device__ __forceinline__ void pre_bwd_rendered(const PreBwdArgs& a, size_t i, float* lds, int lane) {
    const Cam c = load_cam(a.cam, a.W, a.H, a.tanx, a.tany, a.mod, a.D, a.M, a.aa);
    const float4* gq = reinterpret_cast<const float4*>(a.acc + i);
    const float4 g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
    const V3 mean = v3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);
    const bool accum = a.accumulate != 0;

    // SH first: its live registers are released before the covariance chain starts.
    V3 dmean = v3(0, 0, 0);
    if (MODE != SH_NONE) {
        const uint8_t cl = a.clamped[i];
        V3 gcol = v3(g1.z, g1.w, g2.x);
        gcol.x *= (cl & 1) ? 0 : 1;
        gcol.y *= (cl & 2) ? 0 : 1;
        gcol.z *= (cl & 4) ? 0 : 1;
        const V3 campos = v3(c.campos[0], c.campos[1], c.campos[2]);
        const int ncoef = (c.sh_degree + 1) * (c.sh_degree + 1);
        if (MODE == SH_CAT16) {
            // The row goes to registers; dL/dSH overwrites it in LDS coefficient by
            // coefficient (only this lane touches its row between the wave-wide copies).
            float4* lds4 = reinterpret_cast<float4*>(lds);
            float shv[SH_ROW_F];
            sh_row_get(lds4, lane, shv);
            float* row = reinterpret_cast<float*>(lds4 + lane * SH_LDS_STRIDE);
#pragma unroll
            for (int ch = 0; ch < SH_ROW_CHUNKS; ++ch)
                if (4 * ch + 3 >= 3 * ncoef) lds4[lane * SH_LDS_STRIDE + ch] = make_float4(0, 0, 0, 0);
            auto out = [row](int k, V3 v) {
                row[3 * k] = v.x;
                row[3 * k + 1] = v.y;
                row[3 * k + 2] = v.z;
            };
            ShRegs sh = {shv};
            dmean = sh_to_rgb_bwd(c.sh_degree, mean, campos, gcol, sh, out);
        } else if (MODE == SH_SPLIT16) {
            float shv[SH_ROW_F];
            float* row = lds + lane * REST_ROW_F;
            shv[0] = a.shs[3 * i];
            shv[1] = a.shs[3 * i + 1];
            shv[2] = a.shs[3 * i + 2];
#pragma unroll
            for (int k = 0; k < REST_ROW_F; ++k) shv[3 + k] = row[k];
#pragma unroll
            for (int k = 0; k < REST_ROW_F; ++k)
                if (3 + k >= 3 * ncoef) row[k] = 0.0f;
            float* ddc = a.dL_dsh + 3 * i;
            auto out = [row, ddc, accum](int k, V3 v) {
                if (k == 0) {
                    if (accum) { ddc[0] += v.x; ddc[1] += v.y; ddc[2] += v.z; }
                    else { ddc[0] = v.x; ddc[1] = v.y; ddc[2] = v.z; }
                } else {
                    row[3 * k - 3] = v.x;
                    row[3 * k - 2] = v.y;
                    row[3 * k - 1] = v.z;
                }
            };
            ShRegs sh = {shv};
            dmean = sh_to_rgb_bwd(c.sh_degree, mean, campos, gcol, sh, out);
        } else {
            ShPtr sh;
            float *ddc, *drest;
            if (a.shs_rest) {
                sh.dc = a.shs + 3 * i;
                sh.rest = a.shs_rest + i * (size_t)(a.M - 1) * 3;
                ddc = a.dL_dsh + 3 * i;
                drest = a.dL_dsh_rest + i * (size_t)(a.M - 1) * 3;
            } else {
                sh.dc = a.shs + i * (size_t)a.M * 3;
                sh.rest = sh.dc + 3;
                ddc = a.dL_dsh + i * (size_t)a.M * 3;
                drest = ddc + 3;
            }
            auto out = [ddc, drest, accum](int k, V3 v) {
                float* p = k == 0 ? ddc : drest + 3 * (k - 1);
                if (accum) { p[0] += v.x; p[1] += v.y; p[2] += v.z; }
                else { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
            };
            dmean = sh_to_rgb_bwd(c.sh_degree, mean, campos, gcol, sh, out);
            if (!accum)
                for (int k = ncoef; k < a.M; ++k) drest[3 * k - 3] = drest[3 * k - 2] = drest[3 * k - 1] = 0.0f;
        }
    }
    if (MODE == SH_CAT16 || MODE == SH_SPLIT16) wave_lds_sync();

    // The 3D covariance: given, or recomputed from scale and rotation (the same arithmetic as the
    // forward, identical values; a forward that went through gsr_preprocess_batch does not store it).
    float cov6[6];
    V3 sc = v3(1, 1, 1);
    V4 q = {1, 0, 0, 0};
    float dn = 1.0f;
    if (a.scales) {
        sc = v3(a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]);
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[i];
        q = {q4.x, q4.y, q4.z, q4.w};
        if (a.raw) {
            sc = act_exp3(sc);
            q = act_normalize4(q, &dn);
        }
        cov3d_from_scale_rot(sc, c.scale_modifier, q, cov6);
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov6[k] = a.cov3D[6 * i + k];
    }
    float opac = a.opacities[i];
    if (a.raw) opac = act_sigmoid(opac);
    float dop = g1.y;
    const float dinv = g2.y;
    float dcov6[6];
    const V3 dconic = v3(g0.z, g0.w, g1.x);
    dmean = add(dmean, cov2d_bwd(c, mean, cov6, opac, dconic, &dop, a.has_invdepth ? &dinv : nullptr, dcov6));
    dmean = add(dmean, mean2d_bwd(mean, c.proj, g0.x, g0.y));
    if (a.raw) dop = act_sigmoid_bwd(opac, dop);

    if (a.scales) {
        V3 ds;
        V4 dq;
        cov3d_from_scale_rot_bwd(sc, c.scale_modifier, q, dcov6, &ds, &dq);
        if (a.raw) {
            ds = act_exp3_bwd(sc, ds);
            dq = act_normalize4_bwd(q, dn, dq);
        }
        float4* dr = reinterpret_cast<float4*>(a.dL_drotations) + i;
        if (accum) {
            a.dL_dscales[3 * i] += ds.x;
            a.dL_dscales[3 * i + 1] += ds.y;
            a.dL_dscales[3 * i + 2] += ds.z;
            const float4 o = *dr;
            *dr = make_float4(o.x + dq.x, o.y + dq.y, o.z + dq.z, o.w + dq.w);
        } else {
            a.dL_dscales[3 * i] = ds.x;
            a.dL_dscales[3 * i + 1] = ds.y;
            a.dL_dscales[3 * i + 2] = ds.z;
            *dr = make_float4(dq.x, dq.y, dq.z, dq.w);
        }
    }
    a.dL_dmeans2D[3 * i] = g0.x;
    a.dL_dmeans2D[3 * i + 1] = g0.y;
    a.dL_dmeans2D[3 * i + 2] = 0.f;
    if (a.dL_dcolors) {
        a.dL_dcolors[3 * i] = g1.z;
        a.dL_dcolors[3 * i + 1] = g1.w;
        a.dL_dcolors[3 * i + 2] = g2.x;
    }
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = dcov6[k];
    }
    if (accum) {
        a.dL_dopacity[i] += dop;
        a.dL_dmeans3D[3 * i] += dmean.x;
        a.dL_dmeans3D[3 * i + 1] += dmean.y;
        a.dL_dmeans3D[3 * i + 2] += dmean.z;
    } else {
        a.dL_dopacity[i] = dop;
        a.dL_dmeans3D[3 * i] = dmean.x;
        a.dL_dmeans3D[3 * i + 1] = dmean.y;
        a.dL_dmeans3D[3 * i + 2] = dmean.z;
    }
    if (a.DIM >= 4)
        reps_row_out(a.dL_dreps, i, a.DIM, accum, make_float4(g2.z, g2.w, g3.x, g3.y), a.accx + 4 * i);
}

// A Gaussian the forward did not render: every gradient is zero.
template <int MODE>
__device__ __forceinline__ void pre_bwd_unrendered(const PreBwdArgs& a, size_t i, float* lds, int lane) {
    a.dL_dmeans2D[3 * i] = a.dL_dmeans2D[3 * i + 1] = a.dL_dmeans2D[3 * i + 2] = 0.f;
    if (a.dL_dcolors) a.dL_dcolors[3 * i] = a.dL_dcolors[3 * i + 1] = a.dL_dcolors[3 * i + 2] = 0.f;
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = 0.f;
    }
    if (MODE == SH_SPLIT16) {  // also when accumulating: a 16-byte chunk can straddle a rendered neighbour's row
#pragma unroll
        for (int k = 0; k < REST_ROW_F; ++k) lds[lane * REST_ROW_F + k] = 0.0f;
    }
    if (a.accumulate) return;  // adding zero: the parameter gradients stay as they are
    if (MODE == SH_CAT16) {
        float4* lds4 = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int ch = 0; ch < SH_ROW_CHUNKS; ++ch) lds4[lane * SH_LDS_STRIDE + ch] = make_float4(0, 0, 0, 0);
    } else if (MODE == SH_SPLIT16) {
        a.dL_dsh[3 * i] = a.dL_dsh[3 * i + 1] = a.dL_dsh[3 * i + 2] = 0.f;
    } else if (MODE == SH_DIRECT) {
        if (a.shs_rest) {
            a.dL_dsh[3 * i] = a.dL_dsh[3 * i + 1] = a.dL_dsh[3 * i + 2] = 0.f;
            float* r = a.dL_dsh_rest + i * (size_t)(a.M - 1) * 3;
            for (int k = 0; k < 3 * (a.M - 1); ++k) r[k] = 0.f;
        } else {
            float* r = a.dL_dsh + i * (size_t)a.M * 3;
            for (int k = 0; k < 3 * a.M; ++k) r[k] = 0.f;
        }
    }
    a.dL_dopacity[i] = 0.f;
    a.dL_dmeans3D[3 * i] = a.dL_dmeans3D[3 * i + 1] = a.dL_dmeans3D[3 * i + 2] = 0.f;
    if (a.dL_dscales) {
        a.dL_dscales[3 * i] = a.dL_dscales[3 * i + 1] = a.dL_dscales[3 * i + 2] = 0.f;
        reinterpret_cast<float4*>(a.dL_drotations)[i] = make_float4(0, 0, 0, 0);
    }
    for (int e = 0; e < a.DIM / 4; ++e)
        reinterpret_cast<float4*>(a.dL_dreps + i * (size_t)a.DIM)[e] = make_float4(0, 0, 0, 0);
}

#ifndef PRE_BWD_WAVES
#define PRE_BWD_WAVES 2
#endif

template <int MODE>
__global__ void __launch_bounds__(256, PRE_BWD_WAVES) preprocess_bwd_kernel(const PreBwdArgs a) {
    constexpr int WAVE_LDS = ShLds<MODE>::bytes_per_wave;
    __shared__ __align__(16) char s_sh[WAVE_LDS ? 4 * WAVE_LDS : 16];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave_g0 = blockIdx.x * 256 + w * 64;  // first Gaussian of this wave
    const int nrows = min(64, a.P - wave_g0);        // <= 0 for waves past the end
    if (nrows <= 0) return;
    const bool live = idx < a.P;
    // A Gaussian takes part in the backward blend only through pixels that reached it in the
    // forward (backward.cu:592-611 walks the pairs the forward blended).  One that no pixel
    // reached, hidden behind saturated tiles, say, has an all-zero GradRec and with it zero
    // gradients: it is treated like one that was culled, and its GradRec is not read.
    const bool rendered = live && a.radii[live ? idx : 0] > 0 && a.gstat[live ? idx : 0].count != 0u;
    float* lds = reinterpret_cast<float*>(s_sh + w * WAVE_LDS);
    const uint64_t rows = __ballot(rendered);
    if (rows == 0) {
        // No Gaussian of this wave received a gradient (in an occluded scene: nearly every wave).  Each tensor's
        // block of 64 rows is contiguous: zeroed by whole-wave 16-byte stores in address order, where a lane
        // storing its own three or six floats makes every store instruction touch 64 lines a fraction full.
        const size_t g0 = (size_t)wave_g0;
        wave_zero_bytes(a.dL_dmeans2D + 3 * g0, nrows * 12, lane);
        if (a.dL_dcolors) wave_zero_bytes(a.dL_dcolors + 3 * g0, nrows * 12, lane);
        if (a.dL_dcov3D) wave_zero_bytes(a.dL_dcov3D + 6 * g0, nrows * 24, lane);
        if (a.accumulate) return;  // adding zero: the parameter gradients stay as they are
        if (MODE == SH_CAT16) {
            wave_zero_bytes(a.dL_dsh + g0 * SH_ROW_F, nrows * SH_ROW_F * 4, lane);
        } else if (MODE == SH_SPLIT16) {
            wave_zero_bytes(a.dL_dsh + 3 * g0, nrows * 12, lane);
            wave_zero_bytes(a.dL_dsh_rest + g0 * REST_ROW_F, nrows * REST_ROW_F * 4, lane);
        } else if (MODE == SH_DIRECT) {
            if (a.shs_rest) {
                wave_zero_bytes(a.dL_dsh + 3 * g0, nrows * 12, lane);
                wave_zero_bytes(a.dL_dsh_rest + g0 * (size_t)(a.M - 1) * 3, nrows * (a.M - 1) * 12, lane);
            } else {
                wave_zero_bytes(a.dL_dsh + g0 * (size_t)a.M * 3, nrows * a.M * 12, lane);
            }
        }
        wave_zero_bytes(a.dL_dopacity + g0, nrows * 4, lane);
        wave_zero_bytes(a.dL_dmeans3D + 3 * g0, nrows * 12, lane);
        if (a.dL_dscales) {
            wave_zero_bytes(a.dL_dscales + 3 * g0, nrows * 12, lane);
            wave_zero_bytes(a.dL_drotations + 4 * g0, nrows * 16, lane);
        }
        if (a.DIM >= 4) wave_zero_bytes(a.dL_dreps + (size_t)a.DIM * g0, nrows * a.DIM * 4, lane);
        return;
    }
    if (MODE == SH_CAT16) {
        if (rows) sh_rows_load(a.shs + (size_t)wave_g0 * SH_ROW_F, reinterpret_cast<float4*>(lds), rows, nrows, lane);
        wave_lds_sync();
    } else if (MODE == SH_SPLIT16) {
        if (rows) rest_rows_copy<0>(const_cast<float*>(a.shs_rest) + (size_t)wave_g0 * REST_ROW_F, lds, rows, nrows, lane);
        wave_lds_sync();
    }
    if (rendered) pre_bwd_rendered<MODE>(a, (size_t)idx, lds, lane);
    else if (live) pre_bwd_unrendered<MODE>(a, (size_t)idx, lds, lane);
    if (MODE == SH_CAT16) {
        wave_lds_sync();
        float* g = a.dL_dsh + (size_t)wave_g0 * SH_ROW_F;
        if (a.accumulate) sh_rows_add(g, reinterpret_cast<const float4*>(lds), rows, nrows, lane);
        else sh_rows_store(g, reinterpret_cast<const float4*>(lds), nrows, lane);
    } else if (MODE == SH_SPLIT16) {
        wave_lds_sync();
        float* g = a.dL_dsh_rest + (size_t)wave_g0 * REST_ROW_F;
        if (a.accumulate) {
            if (rows) rest_rows_copy<2>(g, lds, rows, nrows, lane);
        } else {
            rest_rows_copy<1>(g, lds, ~0ull, nrows, lane);
        }
    }
}

// ------------------------------------------------------------------ several views at once
// The per-Gaussian backward of a batch of views in one pass: the Gaussian's
// parameters and SH row are read once, every view adds its share (its own
// camera, radii, clamp flags and GradRec), and each gradient is written (or
// added to the caller's buffer) once.  Per view this replaces 575 B of
// traffic per visible Gaussian by the 64-byte GradRec plus 5 bytes of flags.
// What is linear in the incoming gradients (3D covariance -> scale/rotation,
// the activations' chain rule) is applied once to the sums.
// SH layout: [P,1,3] + [P,15,3]; scales and rotations given (no precomputed
// covariance); the other cases take the one-view path.
constexpr int MAX_BATCH_VIEWS = 8;

struct BatchView {
    CamPtrs cam;
    float tanx, tany;
    int W, H, aa, has_invdepth;
    const int* radii;
    const GStat* gstat;
    const uint8_t* clamped;
    const GradRec* acc;
    const float4* accx;  // that view's second records (feature channels 4 .. DIM - 1), DIM > 4
    float* dL_dmeans2D;  // [P,3] of this view, or null
    uint8_t* m2_blocks;  // may be null: see GsrBackwardView.means2D_blocks
};

struct PreBwdBatchArgs {
    PreBwdArgs a;  // parameters and outputs; its per-view members are unused; a.P = end of the Gaussian range
    int first;     // first Gaussian of the range this launch processes (a multiple of 256)
    int nviews;
    BatchView v[MAX_BATCH_VIEWS];
};

#ifndef PRE_BWD_BATCH_WAVES
#define PRE_BWD_BATCH_WAVES 2
#endif
// WIDE: more than four feature channels (their sums come from every view's second record).
template <bool WIDE>
__global__ void __launch_bounds__(256, PRE_BWD_BATCH_WAVES) preprocess_bwd_batch_kernel(const PreBwdBatchArgs b) {
    constexpr int WAVE_LDS = ShLds<SH_SPLIT16>::bytes_per_wave;
    __shared__ __align__(16) char s_sh[4 * WAVE_LDS];
    const PreBwdArgs& a = b.a;
    const int idx = b.first + blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave_g0 = b.first + blockIdx.x * 256 + w * 64;
    const int nrows = min(64, a.P - wave_g0);
    if (nrows <= 0) return;
    const bool live = idx < a.P;
    const size_t i = (size_t)(live ? idx : b.first);
    float* lds = reinterpret_cast<float*>(s_sh + w * WAVE_LDS);
    float* row = lds + lane * REST_ROW_F;

    // All the loads that decide what this wave does are issued before the first of them is waited for
    // (eight views' hit counts, up to nine block states: one memory latency, not seventeen in a row; the
    // launcher points the unused views at view 0's counts so that no load depends on the view count).
    const size_t g0 = (size_t)wave_g0;
    const int blk = wave_g0 >> 6;
    uint32_t cnt[MAX_BATCH_VIEWS];
    uint32_t m2_state[MAX_BATCH_VIEWS];
    const uint8_t* valid = reinterpret_cast<const uint8_t*>(b.v[0].gstat);
#pragma unroll
    for (int v = 0; v < MAX_BATCH_VIEWS; ++v) {
        // a hit count can only be non-zero for a Gaussian that was rendered (radius > 0): one array to read
        cnt[v] = b.v[v].gstat[i].count;
        m2_state[v] = *(b.v[v].m2_blocks ? b.v[v].m2_blocks + blk : valid);   // an absent state array: any valid byte
    }
    uint32_t row_state = *(a.row_blocks ? a.row_blocks + blk : valid);
    // (the values pass through an empty asm statement here: left alone, the compiler sinks the state loads
    // below the first use of the counts, and the wave waits for memory twice)
    static_assert(MAX_BATCH_VIEWS == 8, "the statement below names eight states");
    asm volatile("" : "+v"(m2_state[0]), "+v"(m2_state[1]), "+v"(m2_state[2]), "+v"(m2_state[3]), "+v"(m2_state[4]),
                      "+v"(m2_state[5]), "+v"(m2_state[6]), "+v"(m2_state[7]), "+v"(row_state));
#pragma unroll
    for (int v = 0; v < MAX_BATCH_VIEWS; ++v)
        if (!b.v[v].m2_blocks) m2_state[v] = 1;
    if (!a.row_blocks) row_state = 1;
    uint32_t seen = 0;  // bit v: rendered in view v and reached by a pixel there (else its GradRec is zero)
#pragma unroll
    for (int v = 0; v < MAX_BATCH_VIEWS; ++v)
        if (live && v < b.nviews && cnt[v] != 0u) seen |= 1u << v;
    const uint64_t rows = __ballot(seen != 0);
    // Block states (one byte per 64 Gaussians = per wave, optional): a zero byte says that the block's
    // rows of a gradient tensor are all zero already; such a block is not rewritten with zeros, and the
    // byte is kept up to date (lane 0 stores it; nobody else reads or writes this block's byte).
#pragma unroll
    for (int v = 0; v < MAX_BATCH_VIEWS; ++v) {   // the screen-space gradient of a view: zero where it did not reach
        if (v >= b.nviews) break;
        const BatchView& bv = b.v[v];
        if (!bv.dL_dmeans2D) continue;
        if (__ballot((seen >> v) & 1u) == 0) {
            // whole-wave 16-byte stores in address order (a lane storing its own three floats would make
            // every store instruction touch 64 lines a third full)
            if (m2_state[v] == 0) continue;
            wave_zero_bytes(bv.dL_dmeans2D + 3 * g0, nrows * 12, lane);
            if (bv.m2_blocks && lane == 0) bv.m2_blocks[blk] = 0;
        } else {
            if (live && !((seen >> v) & 1u)) {
                float* m2 = bv.dL_dmeans2D + 3 * i;
                m2[0] = m2[1] = m2[2] = 0.f;
            }
            if (bv.m2_blocks && lane == 0) bv.m2_blocks[blk] = 1;
        }
    }
    if (rows == 0) {
        // No Gaussian of this wave was reached in any view (in a heavily occluded scene: nearly every
        // wave): when overwriting, the wave's rows of every parameter gradient are zero.  Each tensor's
        // block of 64 rows is contiguous and is zeroed like the screen-space gradients above.
        if (a.accumulate) return;
        if (row_state == 0) return;
        wave_zero_bytes(a.dL_dsh + 3 * g0, nrows * 12, lane);
        wave_zero_bytes(a.dL_dopacity + g0, nrows * 4, lane);
        wave_zero_bytes(a.dL_dmeans3D + 3 * g0, nrows * 12, lane);
        wave_zero_bytes(a.dL_dscales + 3 * g0, nrows * 12, lane);
        wave_zero_bytes(a.dL_drotations + 4 * g0, nrows * 16, lane);
        if (a.DIM >= 4) wave_zero_bytes(a.dL_dreps + (size_t)a.DIM * g0, nrows * a.DIM * 4, lane);
        wave_zero_bytes(a.dL_dsh_rest + g0 * REST_ROW_F, nrows * REST_ROW_F * 4, lane);
        if (a.row_blocks && lane == 0) a.row_blocks[blk] = 0;
        return;
    }
    if (a.row_blocks && lane == 0) a.row_blocks[blk] = 1;
    rest_rows_copy<0>(const_cast<float*>(a.shs_rest) + (size_t)wave_g0 * REST_ROW_F, lds, rows, nrows, lane);
    wave_lds_sync();

    V3 dmean = v3(0, 0, 0);
    float dop = 0.f, ddc[3] = {0.f, 0.f, 0.f}, drep[4] = {0.f, 0.f, 0.f, 0.f};
    float4 drepx[WIDE ? 3 : 1];
#pragma unroll
    for (int e = 0; e < (WIDE ? 3 : 1); ++e) drepx[e] = make_float4(0, 0, 0, 0);
    float dcov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    V3 sc = v3(1, 1, 1);
    V4 q = {1, 0, 0, 0};
    float dn = 1.0f, opac = 0.f;
    if (seen) {
        float shv[SH_ROW_F];
        shv[0] = a.shs[3 * i];
        shv[1] = a.shs[3 * i + 1];
        shv[2] = a.shs[3 * i + 2];
#pragma unroll
        for (int k = 0; k < REST_ROW_F; ++k) shv[3 + k] = row[k];
#pragma unroll
        for (int k = 0; k < REST_ROW_F; ++k) row[k] = 0.0f;  // the row becomes the dL/dSH accumulator
        const V3 mean = v3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);
        sc = v3(a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]);
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[i];
        q = {q4.x, q4.y, q4.z, q4.w};
        opac = a.opacities[i];
        if (a.raw) {
            sc = act_exp3(sc);
            q = act_normalize4(q, &dn);
            opac = act_sigmoid(opac);
        }
        float cov6[6];
        cov3d_from_scale_rot(sc, a.mod, q, cov6);  // the same arithmetic as the forward: identical values
        // Two sweeps over the views, so that the SH row (48 registers) is not live during the
        // covariance chain: colour -> SH first, everything else second (the GradRec of a view is
        // read in both; the second read comes from L2).
        {
            ShRegs sh = {shv};
#pragma unroll 1
            for (int v = 0; v < b.nviews; ++v) {
                if (!((seen >> v) & 1u)) continue;
                const BatchView& bv = b.v[v];
                const float4* gq = reinterpret_cast<const float4*>(bv.acc + i);
                const float4 g1 = gq[1];
                const float g2x = gq[2].x;
                const uint8_t cl = bv.clamped[i];
                V3 gcol = v3(g1.z, g1.w, g2x);
                gcol.x *= (cl & 1) ? 0 : 1;
                gcol.y *= (cl & 2) ? 0 : 1;
                gcol.z *= (cl & 4) ? 0 : 1;
                auto out = [row, &ddc](int k, V3 g) {
                    if (k == 0) {
                        ddc[0] += g.x;
                        ddc[1] += g.y;
                        ddc[2] += g.z;
                    } else {
                        row[3 * k - 3] += g.x;
                        row[3 * k - 2] += g.y;
                        row[3 * k - 1] += g.z;
                    }
                };
                const V3 campos = v3(bv.cam.campos[0], bv.cam.campos[1], bv.cam.campos[2]);
                dmean = add(dmean, sh_to_rgb_bwd(a.D, mean, campos, gcol, sh, out));
            }
        }
#pragma unroll 1
        for (int v = 0; v < b.nviews; ++v) {
            if (!((seen >> v) & 1u)) continue;
            const BatchView& bv = b.v[v];
            const Cam c = load_cam(bv.cam, bv.W, bv.H, bv.tanx, bv.tany, a.mod, a.D, a.M, bv.aa);
            const float4* gq = reinterpret_cast<const float4*>(bv.acc + i);
            const float4 g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
            float dop_v = g1.y;
            const float dinv = g2.y;
            float dcov_v[6];
            dmean = add(dmean, cov2d_bwd(c, mean, cov6, opac, v3(g0.z, g0.w, g1.x), &dop_v,
                                         bv.has_invdepth ? &dinv : nullptr, dcov_v));
            dmean = add(dmean, mean2d_bwd(mean, c.proj, g0.x, g0.y));
            if (bv.dL_dmeans2D) {
                float* m2 = bv.dL_dmeans2D + 3 * i;
                m2[0] = g0.x;
                m2[1] = g0.y;
                m2[2] = 0.f;
            }
            dop += dop_v;
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov6[k] += dcov_v[k];
            drep[0] += g2.z;
            drep[1] += g2.w;
            drep[2] += g3.x;
            drep[3] += g3.y;
            if (WIDE) {
                const float4* x = bv.accx + 4 * i;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    if (4 * e + 4 >= a.DIM) break;
                    const float4 xv = x[e];
                    drepx[e] = make_float4(drepx[e].x + xv.x, drepx[e].y + xv.y, drepx[e].z + xv.z, drepx[e].w + xv.w);
                }
            }
        }
    } else if (live) {
#pragma unroll
        for (int k = 0; k < REST_ROW_F; ++k) row[k] = 0.0f;  // a 16-byte chunk can straddle a neighbour's row
    }
    if (live && (seen || !a.accumulate)) {
        V3 ds = v3(0, 0, 0);
        V4 dq = {0, 0, 0, 0};
        if (seen) {
            cov3d_from_scale_rot_bwd(sc, a.mod, q, dcov6, &ds, &dq);
            if (a.raw) {
                dop = act_sigmoid_bwd(opac, dop);
                ds = act_exp3_bwd(sc, ds);
                dq = act_normalize4_bwd(q, dn, dq);
            }
        }
        float4* dr = reinterpret_cast<float4*>(a.dL_drotations) + i;
        if (a.DIM >= 4)
            reps_row_out(a.dL_dreps, i, WIDE ? a.DIM : 4, a.accumulate != 0, make_float4(drep[0], drep[1], drep[2], drep[3]), drepx);
        if (a.accumulate) {
            a.dL_dsh[3 * i] += ddc[0];
            a.dL_dsh[3 * i + 1] += ddc[1];
            a.dL_dsh[3 * i + 2] += ddc[2];
            a.dL_dopacity[i] += dop;
            a.dL_dmeans3D[3 * i] += dmean.x;
            a.dL_dmeans3D[3 * i + 1] += dmean.y;
            a.dL_dmeans3D[3 * i + 2] += dmean.z;
            a.dL_dscales[3 * i] += ds.x;
            a.dL_dscales[3 * i + 1] += ds.y;
            a.dL_dscales[3 * i + 2] += ds.z;
            const float4 o = *dr;
            *dr = make_float4(o.x + dq.x, o.y + dq.y, o.z + dq.z, o.w + dq.w);
        } else {
            a.dL_dsh[3 * i] = ddc[0];
            a.dL_dsh[3 * i + 1] = ddc[1];
            a.dL_dsh[3 * i + 2] = ddc[2];
            a.dL_dopacity[i] = dop;
            a.dL_dmeans3D[3 * i] = dmean.x;
            a.dL_dmeans3D[3 * i + 1] = dmean.y;
            a.dL_dmeans3D[3 * i + 2] = dmean.z;
            a.dL_dscales[3 * i] = ds.x;
            a.dL_dscales[3 * i + 1] = ds.y;
            a.dL_dscales[3 * i + 2] = ds.z;
            *dr = make_float4(dq.x, dq.y, dq.z, dq.w);
        }
    }
    wave_lds_sync();
    float* g = a.dL_dsh_rest + (size_t)wave_g0 * REST_ROW_F;
    if (a.accumulate) {
        if (rows) rest_rows_copy<2>(g, lds, rows, nrows, lane);
    } else {
        rest_rows_copy<1>(g, lds, ~0ull, nrows, lane);
    }
}

// Zeroes the GradRecs the backward blend can add to: those of the Gaussians some pixel reached
// in the forward.  The others are never read (see `rendered` / `seen` above), so a view costs
// 8 bytes read per Gaussian and 64 written per reached one instead of a 64-byte fill of all.
// One lane per Gaussian: the hit counts are read coalesced, and only about one record in ten is written.
// (accx: the second records of a model with more than four feature channels, or null)
__global__ void __launch_bounds__(256) clear_reached_kernel(int P, const GStat* gstat, GradRec* acc, float4* accx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (gstat[i].count != 0u) {
        float4* r = reinterpret_cast<float4*>(acc + i);
        r[0] = r[1] = r[2] = r[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accx) {
            float4* x = accx + 4 * (size_t)i;
            x[0] = x[1] = x[2] = x[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

hipError_t launch_clear_reached(int P, const GeomBuf& g, GradRec* acc, int DIM, hipStream_t st) {
    hipLaunchKernelGGL(clear_reached_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, g.gstat, acc,
                       DIM > 4 ? grad_rec_extra(acc, (size_t)P) : (float4*)nullptr);
    return hipGetLastError();
}

// dL/dmeans2D of one view straight from its GradRec (the only per-Gaussian
// gradient a caller needs per view: the densification statistics read it).
__global__ void __launch_bounds__(256) means2d_grad_kernel(int P, const GStat* gstat, const GradRec* acc,
                                                           float* dL_dmeans2D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float2 g = make_float2(0.f, 0.f);
    if (gstat[i].count != 0u) g = *reinterpret_cast<const float2*>(acc + i);  // records nobody reached hold no data
    dL_dmeans2D[3 * (size_t)i] = g.x;
    dL_dmeans2D[3 * (size_t)i + 1] = g.y;
    dL_dmeans2D[3 * (size_t)i + 2] = 0.f;
}

hipError_t launch_means2d_grad(int P, const GeomBuf& g, const GradRec* acc, float* dL_dmeans2D, hipStream_t st) {
    hipLaunchKernelGGL(means2d_grad_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, g.gstat, acc, dL_dmeans2D);
    return hipGetLastError();
}

hipError_t launch_preprocess_bwd_batch(const GsrScene& s, int nviews, const GsrBackwardView* views,
                                       const GsrBackwardIO& io, int first, int count, hipStream_t st) {
    PreBwdBatchArgs b;
    PreBwdArgs& a = b.a;
    a = PreBwdArgs();
    b.first = first;
    a.P = first + count; a.D = s.D; a.M = s.M; a.DIM = s.DIM;
    a.mod = s.scale_modifier; a.raw = s.raw_params;
    a.means3D = s.means3D; a.shs = s.shs; a.shs_rest = s.shs_rest; a.opacities = s.opacities;
    a.scales = s.scales; a.rotations = s.rotations;
    a.dL_dopacity = io.dL_dopacity; a.dL_dmeans3D = io.dL_dmeans3D; a.dL_dsh = io.dL_dsh;
    a.dL_dsh_rest = io.dL_dsh_rest; a.dL_dscales = io.dL_dscales; a.dL_drotations = io.dL_drotations;
    a.dL_dreps = io.dL_dreps;
    a.row_blocks = io.row_blocks;
    const dim3 grid((count + 255) / 256), block(256);
    for (int v0 = 0; v0 < nviews; v0 += MAX_BATCH_VIEWS) {
        b.nviews = nviews - v0 < MAX_BATCH_VIEWS ? nviews - v0 : MAX_BATCH_VIEWS;
        a.accumulate = (io.accumulate || v0 > 0) ? 1 : 0;
        for (int k = 0; k < b.nviews; ++k) {
            const GsrBackwardView& gv = views[v0 + k];
            BatchView& bv = b.v[k];
            bv.cam.view = gv.viewmatrix; bv.cam.proj = gv.projmatrix; bv.cam.campos = gv.campos;
            bv.tanx = gv.tanfovx; bv.tany = gv.tanfovy; bv.W = gv.W; bv.H = gv.H; bv.aa = gv.antialiasing;
            bv.has_invdepth = gv.has_invdepth;
            bv.radii = gv.radii;
            const GeomBuf gb = geom_buf(gv.geom_buffer, (size_t)s.P);
            bv.clamped = gb.clamped;
            bv.gstat = gb.gstat;
            bv.acc = (const GradRec*)gv.work;
            bv.accx = grad_rec_extra(gv.work, (size_t)s.P);
            bv.dL_dmeans2D = gv.dL_dmeans2D;
            bv.m2_blocks = gv.dL_dmeans2D ? gv.means2D_blocks : nullptr;
        }
        for (int k = b.nviews; k < MAX_BATCH_VIEWS; ++k) {   // every view's count pointer is dereferenced
            b.v[k] = BatchView();
            b.v[k].gstat = b.v[0].gstat;
        }
        if (s.DIM > 4) hipLaunchKernelGGL(preprocess_bwd_batch_kernel<true>, grid, block, 0, st, b);
        else hipLaunchKernelGGL(preprocess_bwd_batch_kernel<false>, grid, block, 0, st, b);
    }
    return hipGetLastError();
}

hipError_t launch_preprocess_bwd(const GsrScene& s, const int* radii, const GeomBuf& g, const GsrBackwardIO& io,
                                 const GradRec* acc, hipStream_t st) {
    PreBwdArgs a;
    a.P = s.P; a.D = s.D; a.M = s.M; a.DIM = s.DIM; a.W = s.W; a.H = s.H;
    a.mod = s.scale_modifier; a.tanx = s.tanfovx; a.tany = s.tanfovy; a.aa = s.antialiasing;
    a.raw = s.raw_params; a.accumulate = io.accumulate;
    a.means3D = s.means3D; a.shs = s.shs; a.shs_rest = s.shs_rest; a.colors_precomp = s.colors_precomp;
    a.opacities = s.opacities; a.scales = s.scales; a.rotations = s.rotations;
    a.cov3D = s.cov3D_precomp;  // null: recomputed from scales and rotations
    a.cam.view = s.viewmatrix; a.cam.proj = s.projmatrix; a.cam.campos = s.campos;
    a.radii = radii; a.gstat = g.gstat; a.clamped = g.clamped; a.acc = acc;
    a.accx = grad_rec_extra(acc, (size_t)s.P);
    a.has_invdepth = io.dL_dout_invdepth != nullptr;
    a.dL_dmeans2D = io.dL_dmeans2D; a.dL_dcolors = io.dL_dcolors; a.dL_dopacity = io.dL_dopacity;
    a.dL_dmeans3D = io.dL_dmeans3D; a.dL_dcov3D = io.dL_dcov3D; a.dL_dsh = io.dL_dsh; a.dL_dsh_rest = io.dL_dsh_rest;
    a.dL_dscales = s.scales ? io.dL_dscales : nullptr;
    a.dL_drotations = s.scales ? io.dL_drotations : nullptr;
    a.dL_dreps = io.dL_dreps;
    const dim3 grid((s.P + 255) / 256), block(256);
    if (!s.shs) hipLaunchKernelGGL(preprocess_bwd_kernel<SH_NONE>, grid, block, 0, st, a);
    else if (s.M == 16 && s.shs_rest) hipLaunchKernelGGL(preprocess_bwd_kernel<SH_SPLIT16>, grid, block, 0, st, a);
    else if (s.M == 16) hipLaunchKernelGGL(preprocess_bwd_kernel<SH_CAT16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(preprocess_bwd_kernel<SH_DIRECT>, grid, block, 0, st, a);
    return hipGetLastError();
}

}  // namespace gs
