// TEST INFRASTRUCTURE -- CPU checker for the Gaussian rasterizer hot path.
//
// This file is the OpenMP C++ restatement of the reference's CUDA path.  It is
// what the GPU results are compared against and what bench.py times as the CPU
// baseline.  Nothing in the product package may include, link or call it.
//
// Pinning: the reference holds no golden vectors for the rasterizer or for
// simple-knn (its only test, submodules/fused-ssim/tests/test.py, pins
// fused-ssim against a conv2d SSIM).  The restatement is pinned in
// tests/test_oracle_*.py against the reference's own importable Python
// (utils/sh_utils.eval_sh, utils/loss_utils.ssim, utils/graphics_utils) through
// committed fixtures, and against independent float64 / autograd restatements.
//
// Function -> reference map (all under submodules/):
//   preprocess_one            diff-gaussian-rasterization/cuda_rasterizer/forward.cu:151-308
//   gso_forward_geom          .../rasterizer_impl.cu:243-346 (preprocess + InclusiveSum)
//   emit_keys                 .../rasterizer_impl.cu:70-110  (duplicateWithKeys)
//   radix_sort_pairs          .../rasterizer_impl.cu:373-380 (cub::DeviceRadixSort::SortPairs, stable LSD)
//   tile_ranges               .../rasterizer_impl.cu:115-137, 388-400
//   render_tile_fwd           .../forward.cu:313-490
//   render_tile_bwd           .../backward.cu:474-686
//   gso_backward              .../rasterizer_impl.cu:437-577, backward.cu:147-329, 401-471
//   gso_mark_visible          .../rasterizer_impl.cu:54-66
//   gso_dist2                 simple-knn/simple_knn.cu:186-222
//   gso_ssim_fwd / _bwd       fused-ssim/ssim.cu:187-366
//
// Floating-point sums that the reference forms with atomicAdd (order undefined
// there) are formed here per tile-instance in pixel order and then added per
// Gaussian in sorted-list order, so the checker is deterministic.

#include <omp.h>
#include <stdlib.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <vector>

#include "../gaussian-splatting_amd/csrc/gs_math.h"

using namespace gs;

extern "C" {

struct GsoScene {
    int P, D, M, DIM, W, H;
    const float* bg;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    const float* reps;
    float scale_modifier, tanfovx, tanfovy;
    int prefiltered, antialiasing;
};

struct GsoGeom {
    int* radii;
    float* means2D;        // [P,2]
    float* depths;         // [P]
    float* cov3D;          // [P,6]
    float* conic_opacity;  // [P,4]
    float* rgb;            // [P,3]
    uint8_t* clamped;      // [P,3]
    uint32_t* tiles_touched;
    uint32_t* point_offsets;
    float* out_depths;
};

struct GsoBin {
    uint64_t* keys_unsorted;
    uint32_t* vals_unsorted;
    uint64_t* keys;
    uint32_t* point_list;
};

struct GsoImage {
    uint32_t* ranges;  // [tiles,2]
    float* final_T;
    uint32_t* n_contrib;
    float* out_color;   // [3,H,W]
    float* invdepth;    // [H,W] or null
    float* accum_alpha;
    float* out_reps;    // [DIM,H,W]
    int* last_contr;    // [H,W], caller presets -1
    float* gauss_sum;   // [P], caller presets 0
    int* gauss_count;   // [P], caller presets 0
};

struct GsoGrads {
    const float* dL_dpix;       // [3,H,W]
    const float* dL_dinvdepth;  // [H,W] or null
    const float* dL_dreps_pix;  // [DIM,H,W]
    float* dL_dmean2D;          // [P,3]
    float* dL_dconic;           // [P,4]
    float* dL_dopacity;         // [P]
    float* dL_dcolor;           // [P,3]
    float* dL_dinvdepths;       // [P] or null
    float* dL_dmean3D;          // [P,3]
    float* dL_dcov3D;           // [P,6]
    float* dL_dsh;              // [P,M,3]
    float* dL_dscale;           // [P,3]
    float* dL_drot;             // [P,4]
    float* dL_dreps;            // [P,DIM]
};

}  // extern "C"

static Cam make_cam(const GsoScene& s) {
    Cam c;
    memcpy(c.view, s.view, sizeof(c.view));
    memcpy(c.proj, s.proj, sizeof(c.proj));
    memcpy(c.campos, s.campos, sizeof(c.campos));
    c.W = s.W;
    c.H = s.H;
    c.tanfovx = s.tanfovx;
    c.tanfovy = s.tanfovy;
    c.scale_modifier = s.scale_modifier;
    c.sh_degree = s.D;
    c.sh_coeffs = s.M;
    c.antialiasing = s.antialiasing;
    cam_finish(&c);
    return c;
}

struct ShRow {
    const float* p;
    V3 operator()(int k) const { return v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]); }
};

static void preprocess_one(const GsoScene& s, const Cam& c, int i, GsoGeom& g) {
    g.radii[i] = 0;
    g.tiles_touched[i] = 0;
    const V3 p = v3(s.means3D[3 * i], s.means3D[3 * i + 1], s.means3D[3 * i + 2]);
    float depth;
    if (near_cull(p, c.view, &depth)) return;

    const float* cov6;
    if (s.cov3D_precomp) {
        cov6 = s.cov3D_precomp + 6 * i;
    } else {
        const V3 sc = v3(s.scales[3 * i], s.scales[3 * i + 1], s.scales[3 * i + 2]);
        const V4 q = {s.rotations[4 * i], s.rotations[4 * i + 1], s.rotations[4 * i + 2], s.rotations[4 * i + 3]};
        cov3d_from_scale_rot(sc, c.scale_modifier, q, g.cov3D + 6 * i);
        cov6 = g.cov3D + 6 * i;
    }
    Splat sp;
    if (!project_splat(c, p, cov6, s.opacities[i], depth, &sp)) return;

    if (!s.colors_precomp) {
        ShRow sh = {s.shs + (size_t)i * s.M * 3};
        V3 col = sh_to_rgb(c.sh_degree, p, v3(c.campos[0], c.campos[1], c.campos[2]), sh);
        g.clamped[3 * i + 0] = col.x < 0;
        g.clamped[3 * i + 1] = col.y < 0;
        g.clamped[3 * i + 2] = col.z < 0;
        g.rgb[3 * i + 0] = fmax_(col.x, 0.0f);
        g.rgb[3 * i + 1] = fmax_(col.y, 0.0f);
        g.rgb[3 * i + 2] = fmax_(col.z, 0.0f);
    }
    g.depths[i] = sp.depth;
    g.out_depths[i] = sp.depth;
    g.radii[i] = sp.radius;
    g.means2D[2 * i] = sp.xy.x;
    g.means2D[2 * i + 1] = sp.xy.y;
    g.conic_opacity[4 * i + 0] = sp.conic_opacity.x;
    g.conic_opacity[4 * i + 1] = sp.conic_opacity.y;
    g.conic_opacity[4 * i + 2] = sp.conic_opacity.z;
    g.conic_opacity[4 * i + 3] = sp.conic_opacity.w;
    g.tiles_touched[i] = sp.tiles;
}

// Stable LSD radix sort of (u64 key, u32 value) on key bits [0, end_bit).
static void radix_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, int64_t n,
                             int end_bit) {
    if (n <= 0) return;
    std::vector<uint64_t> ka(kin, kin + n), kb(n);
    std::vector<uint32_t> va(vin, vin + n), vb(n);
    uint64_t* ks = ka.data();
    uint64_t* kd = kb.data();
    uint32_t* vs = va.data();
    uint32_t* vd = vb.data();
    const int nt = omp_get_max_threads();
    std::vector<int64_t> hist((size_t)nt * 256);
    for (int shift = 0; shift < end_bit; shift += 8) {
        const int bits = std::min(8, end_bit - shift);
        const uint64_t mask = (1ull << bits) - 1;
        std::fill(hist.begin(), hist.end(), 0);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            int64_t* h = &hist[(size_t)t * 256];
            for (int64_t i = lo; i < hi; ++i) h[(ks[i] >> shift) & mask]++;
#pragma omp barrier
#pragma omp single
            {
                int64_t run = 0;
                for (int d = 0; d < 256; ++d)
                    for (int tt = 0; tt < nt; ++tt) {
                        const int64_t cnt = hist[(size_t)tt * 256 + d];
                        hist[(size_t)tt * 256 + d] = run;
                        run += cnt;
                    }
            }
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t dst = h[(ks[i] >> shift) & mask]++;
                kd[dst] = ks[i];
                vd[dst] = vs[i];
            }
        }
        std::swap(ks, kd);
        std::swap(vs, vd);
    }
    memcpy(kout, ks, n * sizeof(uint64_t));
    memcpy(vout, vs, n * sizeof(uint32_t));
}

static void emit_keys(const GsoScene& s, const Cam& c, const GsoGeom& g, GsoBin& b) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < s.P; ++i) {
        if (g.radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : g.point_offsets[i - 1];
        const V2 xy = {g.means2D[2 * i], g.means2D[2 * i + 1]};
        const Rect r = tile_rect(xy, g.radii[i], c.grid_x, c.grid_y);
        const uint32_t dbits = f2bits(g.depths[i]);
        for (int y = r.y0; y < r.y1; ++y)
            for (int x = r.x0; x < r.x1; ++x) {
                uint64_t key = (uint64_t)(y * c.grid_x + x);
                key <<= 32;
                key |= dbits;
                b.keys_unsorted[off] = key;
                b.vals_unsorted[off] = (uint32_t)i;
                off++;
            }
    }
}

static void tile_ranges(const uint64_t* keys, int64_t R, uint32_t* ranges, int ntiles) {
    memset(ranges, 0, sizeof(uint32_t) * 2 * ntiles);
    if (R <= 0) return;
#pragma omp parallel for
    for (int64_t i = 0; i < R; ++i) {
        const uint32_t cur = (uint32_t)(keys[i] >> 32);
        if (i == 0) {
            ranges[2 * cur] = 0;
        } else {
            const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (cur != prev) {
                ranges[2 * prev + 1] = (uint32_t)i;
                ranges[2 * cur] = (uint32_t)i;
            }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

static const int MAXDIM = 32;

static void render_tile_fwd(const GsoScene& s, const Cam& c, const GsoGeom& g, const GsoBin& b, GsoImage& im, int tx,
                            int ty, const float* feat, float* gs_part, int* cnt_part) {
    const int W = s.W, H = s.H, DIM = s.DIM;
    const int tile = ty * c.grid_x + tx;
    const uint32_t r0 = im.ranges[2 * tile], r1 = im.ranges[2 * tile + 1];
    float T[TILE_PIX], C[3][TILE_PIX], invd[TILE_PIX];
    std::vector<float> Re((size_t)DIM * TILE_PIX, 0.0f);
    uint32_t last[TILE_PIX];
    bool done[TILE_PIX], inside[TILE_PIX];
    int ndone = 0;
    for (int t = 0; t < TILE_PIX; ++t) {
        const int px = tx * TILE_X + (t % TILE_X), py = ty * TILE_Y + (t / TILE_X);
        inside[t] = px < W && py < H;
        done[t] = !inside[t];
        ndone += done[t];
        T[t] = 1.0f;
        C[0][t] = C[1][t] = C[2][t] = 0.0f;
        invd[t] = 0.0f;
        last[t] = 0;
    }
    for (uint32_t i = r0; i < r1 && ndone < TILE_PIX; ++i) {
        const uint32_t gid = b.point_list[i];
        const uint32_t contributor = i - r0 + 1;
        const V2 xy = {g.means2D[2 * gid], g.means2D[2 * gid + 1]};
        const V4 co = {g.conic_opacity[4 * gid], g.conic_opacity[4 * gid + 1], g.conic_opacity[4 * gid + 2],
                       g.conic_opacity[4 * gid + 3]};
        float gsum = 0.0f;
        int gcnt = 0;
        for (int t = 0; t < TILE_PIX; ++t) {
            if (done[t]) continue;
            const int px = tx * TILE_X + (t % TILE_X), py = ty * TILE_Y + (t / TILE_X);
            float dx, dy;
            const float power = splat_power(xy, (float)px, (float)py, co, &dx, &dy);
            if (power > 0.0f) continue;
            gsum += T[t];
            gcnt += 1;
            const float alpha = fmin_(GS_ALPHA_MAX, co.w * gs_expf(power));
            if (alpha < GS_ALPHA_MIN) continue;
            const float test_T = T[t] * (1 - alpha);
            if (test_T < GS_T_MIN) {
                done[t] = true;
                ndone++;
                continue;
            }
            for (int ch = 0; ch < 3; ++ch) C[ch][t] += feat[gid * 3 + ch] * alpha * T[t];
            for (int ch = 0; ch < DIM; ++ch) Re[(size_t)ch * TILE_PIX + t] += s.reps[(size_t)gid * DIM + ch] * alpha * T[t];
            if (im.invdepth) invd[t] += (1 / g.depths[gid]) * alpha * T[t];
            T[t] = test_T;
            last[t] = contributor;
            im.last_contr[py * W + px] = (int)gid;
        }
        gs_part[i] = gsum;
        cnt_part[i] = gcnt;
    }
    for (int t = 0; t < TILE_PIX; ++t) {
        if (!inside[t]) continue;
        const int px = tx * TILE_X + (t % TILE_X), py = ty * TILE_Y + (t / TILE_X);
        const size_t pid = (size_t)py * W + px;
        im.final_T[pid] = T[t];
        im.accum_alpha[pid] = T[t];
        im.n_contrib[pid] = last[t];
        for (int ch = 0; ch < 3; ++ch) im.out_color[(size_t)ch * H * W + pid] = C[ch][t] + T[t] * s.bg[ch];
        for (int ch = 0; ch < DIM; ++ch) im.out_reps[(size_t)ch * H * W + pid] = Re[(size_t)ch * TILE_PIX + t];
        if (im.invdepth) im.invdepth[pid] = invd[t];
    }
}

struct PixBwd {
    float T, T_final, last_alpha, accum_invd, last_invd, dL_invd;
    float dpix[3], accum_rec[3], last_color[3];
    float drep[MAXDIM], accum_rep[MAXDIM], last_rep[MAXDIM];
    uint32_t last_contributor;
    bool inside;
};

// part layout per instance: [0..1] mean2D, [2..4] conic x,y,w, [5] opacity, [6..8] colour, [9] invdepth, [10..] reps
static void render_tile_bwd(const GsoScene& s, const Cam& c, const GsoGeom& g, const GsoBin& b, const GsoImage& im,
                            const GsoGrads& gr, int tx, int ty, const float* feat, double* part, int stride) {
    const int W = s.W, H = s.H, DIM = s.DIM;
    const int tile = ty * c.grid_x + tx;
    const uint32_t r0 = im.ranges[2 * tile], r1 = im.ranges[2 * tile + 1];
    if (r1 <= r0) return;
    std::vector<PixBwd> st(TILE_PIX);
    uint32_t max_last = 0;
    for (int t = 0; t < TILE_PIX; ++t) {
        PixBwd& p = st[t];
        memset(&p, 0, sizeof(p));
        const int px = tx * TILE_X + (t % TILE_X), py = ty * TILE_Y + (t / TILE_X);
        p.inside = px < W && py < H;
        if (!p.inside) continue;
        const size_t pid = (size_t)py * W + px;
        p.T_final = im.final_T[pid];
        p.T = p.T_final;
        p.last_contributor = im.n_contrib[pid];
        max_last = std::max(max_last, p.last_contributor);
        for (int ch = 0; ch < 3; ++ch) p.dpix[ch] = gr.dL_dpix[(size_t)ch * H * W + pid];
        for (int ch = 0; ch < DIM; ++ch) p.drep[ch] = gr.dL_dreps_pix[(size_t)ch * H * W + pid];
        if (gr.dL_dinvdepth) p.dL_invd = gr.dL_dinvdepth[pid];
    }
    const float ddelx_dx = 0.5f * W;
    const float ddely_dy = 0.5f * H;
    for (int64_t i = (int64_t)r0 + max_last - 1; i >= (int64_t)r0; --i) {
        const uint32_t contributor = (uint32_t)(i - r0);  // 0-based position, == reference's decremented counter
        const uint32_t gid = b.point_list[i];
        const V2 xy = {g.means2D[2 * gid], g.means2D[2 * gid + 1]};
        const V4 co = {g.conic_opacity[4 * gid], g.conic_opacity[4 * gid + 1], g.conic_opacity[4 * gid + 2],
                       g.conic_opacity[4 * gid + 3]};
        double* acc = part + (size_t)i * stride;
        for (int t = 0; t < TILE_PIX; ++t) {
            PixBwd& p = st[t];
            if (!p.inside || contributor >= p.last_contributor) continue;
            const int px = tx * TILE_X + (t % TILE_X), py = ty * TILE_Y + (t / TILE_X);
            float dx, dy;
            const float power = splat_power(xy, (float)px, (float)py, co, &dx, &dy);
            if (power > 0.0f) continue;
            const float G = gs_expf(power);
            const float alpha = fmin_(GS_ALPHA_MAX, co.w * G);
            if (alpha < GS_ALPHA_MIN) continue;

            p.T = p.T / (1.f - alpha);
            const float dchannel_dcolor = alpha * p.T;
            float dL_dalpha = 0.0f;
            for (int ch = 0; ch < 3; ++ch) {
                const float cc = feat[gid * 3 + ch];
                p.accum_rec[ch] = p.last_alpha * p.last_color[ch] + (1.f - p.last_alpha) * p.accum_rec[ch];
                p.last_color[ch] = cc;
                const float dL_dchannel = p.dpix[ch];
                dL_dalpha += (cc - p.accum_rec[ch]) * dL_dchannel;
                acc[6 + ch] += dchannel_dcolor * dL_dchannel;
            }
            for (int ch = 0; ch < DIM; ++ch) {
                const float cc = s.reps[(size_t)gid * DIM + ch];
                p.accum_rep[ch] = p.last_alpha * p.last_rep[ch] + (1.f - p.last_alpha) * p.accum_rep[ch];
                p.last_rep[ch] = cc;
                const float dL_dchannel = p.drep[ch];
                dL_dalpha += (cc - p.accum_rep[ch]) * dL_dchannel;
                acc[10 + ch] += dchannel_dcolor * dL_dchannel;
            }
            if (gr.dL_dinvdepths) {
                const float invd = 1.f / g.depths[gid];
                p.accum_invd = p.last_alpha * p.last_invd + (1.f - p.last_alpha) * p.accum_invd;
                p.last_invd = invd;
                dL_dalpha += (invd - p.accum_invd) * p.dL_invd;
                acc[9] += dchannel_dcolor * p.dL_invd;
            }
            dL_dalpha *= p.T;
            p.last_alpha = alpha;
            float bg_dot_dpixel = 0;
            for (int ch = 0; ch < 3; ++ch) bg_dot_dpixel += s.bg[ch] * p.dpix[ch];
            dL_dalpha += (-p.T_final / (1.f - alpha)) * bg_dot_dpixel;

            acc[5] += G * dL_dalpha;
            const float dL_dG = co.w * dL_dalpha;
            const float gdx = G * dx;
            const float gdy = G * dy;
            const float dG_ddelx = -gdx * co.x - gdy * co.y;
            const float dG_ddely = -gdy * co.z - gdx * co.y;
            acc[0] += dL_dG * dG_ddelx * ddelx_dx;
            acc[1] += dL_dG * dG_ddely * ddely_dy;
            acc[2] += -0.5f * gdx * dx * dL_dG;
            acc[3] += -0.5f * gdx * dy * dL_dG;
            acc[4] += -0.5f * gdy * dy * dL_dG;
        }
    }
}

extern "C" {

int64_t gso_forward_geom(const GsoScene* sp, GsoGeom* g) {
    const GsoScene& s = *sp;
    const Cam c = make_cam(s);
#pragma omp parallel for schedule(static, 4096)
    for (int i = 0; i < s.P; ++i) preprocess_one(s, c, i, *g);
    uint64_t run = 0;
    for (int i = 0; i < s.P; ++i) {
        run += g->tiles_touched[i];
        g->point_offsets[i] = (uint32_t)run;
    }
    return (int64_t)run;
}

int gso_forward_render(const GsoScene* sp, const GsoGeom* g, GsoBin* b, GsoImage* im, int64_t R) {
    const GsoScene& s = *sp;
    if (s.DIM > MAXDIM) return -1;
    const Cam c = make_cam(s);
    const int ntiles = c.grid_x * c.grid_y;
    emit_keys(s, c, *g, *b);
    const int bit = (int)higher_msb((uint32_t)ntiles);
    radix_sort_pairs(b->keys_unsorted, b->vals_unsorted, b->keys, b->point_list, R, 32 + bit);
    tile_ranges(b->keys, R, im->ranges, ntiles);
    const float* feat = s.colors_precomp ? s.colors_precomp : g->rgb;
    std::vector<float> gs_part((size_t)std::max<int64_t>(R, 1), 0.0f);
    std::vector<int> cnt_part((size_t)std::max<int64_t>(R, 1), 0);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int ty = 0; ty < c.grid_y; ++ty)
        for (int tx = 0; tx < c.grid_x; ++tx)
            render_tile_fwd(s, c, *g, *b, *im, tx, ty, feat, gs_part.data(), cnt_part.data());
    for (int64_t i = 0; i < R; ++i) {
        const uint32_t gid = b->point_list[i];
        im->gauss_sum[gid] += gs_part[i];
        im->gauss_count[gid] += cnt_part[i];
    }
    return 0;
}

// All gradient outputs must be zero-filled by the caller (the reference does
// this with torch::zeros, rasterize_points.cu:246-256).
int gso_backward(const GsoScene* sp, const GsoGeom* g, const GsoBin* b, const GsoImage* im, int64_t R, GsoGrads* gr) {
    const GsoScene& s = *sp;
    if (s.DIM > MAXDIM) return -1;
    const Cam c = make_cam(s);
    const float* feat = s.colors_precomp ? s.colors_precomp : g->rgb;
    const int stride = 10 + s.DIM;
    {
        // The reference adds every pair's terms to the per-Gaussian gradients with float atomics, in no
        // defined order.  The checker takes the order out of the result: each term is formed in float as
        // the reference forms it, the SUMS are carried in double (per instance, then per Gaussian in list
        // order) and rounded to float once.  A float implementation is compared with this at 1e-4.
        std::vector<double> part((size_t)std::max<int64_t>(R, 1) * stride, 0.0);
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int ty = 0; ty < c.grid_y; ++ty)
            for (int tx = 0; tx < c.grid_x; ++tx) render_tile_bwd(s, c, *g, *b, *im, *gr, tx, ty, feat, part.data(), stride);
        std::vector<double> sum((size_t)std::max(s.P, 1) * stride, 0.0);
        for (int64_t i = 0; i < R; ++i) {
            const uint32_t gid = b->point_list[i];
            const double* a = &part[(size_t)i * stride];
            double* d = &sum[(size_t)gid * stride];
            for (int k = 0; k < stride; ++k) d[k] += a[k];
        }
#pragma omp parallel for schedule(static, 4096)
        for (int gid = 0; gid < s.P; ++gid) {
            const double* a = &sum[(size_t)gid * stride];
            gr->dL_dmean2D[3 * gid + 0] += (float)a[0];
            gr->dL_dmean2D[3 * gid + 1] += (float)a[1];
            gr->dL_dconic[4 * gid + 0] += (float)a[2];
            gr->dL_dconic[4 * gid + 1] += (float)a[3];
            gr->dL_dconic[4 * gid + 3] += (float)a[4];
            gr->dL_dopacity[gid] += (float)a[5];
            for (int ch = 0; ch < 3; ++ch) gr->dL_dcolor[3 * gid + ch] += (float)a[6 + ch];
            if (gr->dL_dinvdepths) gr->dL_dinvdepths[gid] += (float)a[9];
            for (int ch = 0; ch < s.DIM; ++ch) gr->dL_dreps[(size_t)gid * s.DIM + ch] += (float)a[10 + ch];
        }
    }
    const float* cov3D = s.cov3D_precomp ? s.cov3D_precomp : g->cov3D;
    const V3 campos = v3(c.campos[0], c.campos[1], c.campos[2]);
#pragma omp parallel for schedule(static, 4096)
    for (int i = 0; i < s.P; ++i) {
        if (!(g->radii[i] > 0)) continue;
        const V3 mean = v3(s.means3D[3 * i], s.means3D[3 * i + 1], s.means3D[3 * i + 2]);
        // computeCov2DCUDA: assigns dL_dmean3D
        const V3 dconic = v3(gr->dL_dconic[4 * i], gr->dL_dconic[4 * i + 1], gr->dL_dconic[4 * i + 3]);
        V3 dmean = cov2d_bwd(c, mean, cov3D + 6 * i, s.opacities[i], dconic, &gr->dL_dopacity[i],
                             gr->dL_dinvdepths ? &gr->dL_dinvdepths[i] : nullptr, gr->dL_dcov3D + 6 * i);
        // preprocessCUDA (backward): += projection term, += SH term
        dmean = add(dmean, mean2d_bwd(mean, c.proj, gr->dL_dmean2D[3 * i], gr->dL_dmean2D[3 * i + 1]));
        if (s.shs) {
            ShRow sh = {s.shs + (size_t)i * s.M * 3};
            V3 gcol = v3(gr->dL_dcolor[3 * i], gr->dL_dcolor[3 * i + 1], gr->dL_dcolor[3 * i + 2]);
            gcol.x *= g->clamped[3 * i + 0] ? 0 : 1;
            gcol.y *= g->clamped[3 * i + 1] ? 0 : 1;
            gcol.z *= g->clamped[3 * i + 2] ? 0 : 1;
            float* dsh = gr->dL_dsh + (size_t)i * s.M * 3;
            auto out = [dsh](int k, V3 v) {
                dsh[3 * k] = v.x;
                dsh[3 * k + 1] = v.y;
                dsh[3 * k + 2] = v.z;
            };
            dmean = add(dmean, sh_to_rgb_bwd(c.sh_degree, mean, campos, gcol, sh, out));
        }
        gr->dL_dmean3D[3 * i + 0] = dmean.x;
        gr->dL_dmean3D[3 * i + 1] = dmean.y;
        gr->dL_dmean3D[3 * i + 2] = dmean.z;
        if (s.scales) {
            const V3 sc = v3(s.scales[3 * i], s.scales[3 * i + 1], s.scales[3 * i + 2]);
            const V4 q = {s.rotations[4 * i], s.rotations[4 * i + 1], s.rotations[4 * i + 2], s.rotations[4 * i + 3]};
            V3 ds;
            V4 dq;
            cov3d_from_scale_rot_bwd(sc, c.scale_modifier, q, gr->dL_dcov3D + 6 * i, &ds, &dq);
            gr->dL_dscale[3 * i + 0] = ds.x;
            gr->dL_dscale[3 * i + 1] = ds.y;
            gr->dL_dscale[3 * i + 2] = ds.z;
            gr->dL_drot[4 * i + 0] = dq.x;
            gr->dL_drot[4 * i + 1] = dq.y;
            gr->dL_drot[4 * i + 2] = dq.z;
            gr->dL_drot[4 * i + 3] = dq.w;
        }
    }
    return 0;
}

void gso_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present) {
    (void)proj;
    for (int i = 0; i < P; ++i) {
        float d;
        present[i] = !near_cull(v3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]), view, &d);
    }
}

void gso_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, int64_t n, int end_bit) {
    radix_sort_pairs(kin, vin, kout, vout, n, end_bit);
}

// SH -> RGB (+0.5, clamp at 0) for N points, as preprocess does for a visible Gaussian.
void gso_sh_eval(int N, int deg, int M, const float* pos, const float* campos, const float* shs, float* rgb,
                 uint8_t* clamped) {
    for (int i = 0; i < N; ++i) {
        ShRow sh = {shs + (size_t)i * M * 3};
        const V3 c = sh_to_rgb(deg, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                               v3(campos[0], campos[1], campos[2]), sh);
        const float v[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; ++k) {
            clamped[3 * i + k] = v[k] < 0;
            rgb[3 * i + k] = fmax_(v[k], 0.0f);
        }
    }
}

void gso_expf(const float* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = gs_expf(x[i]);
}

// Activations the model applies before every render (scene/gaussian_model.py:41-53:
// scaling = exp, opacity = sigmoid, rotation = F.normalize) and their backward,
// for checking the kernels' raw-parameter mode.
void gso_activate(int P, const float* logit, const float* logscale, const float* rot, float* opacity, float* scales,
                  float* rotations) {
#pragma omp parallel for
    for (int i = 0; i < P; ++i) {
        opacity[i] = act_sigmoid(logit[i]);
        const V3 s = act_exp3(v3(logscale[3 * i], logscale[3 * i + 1], logscale[3 * i + 2]));
        scales[3 * i] = s.x; scales[3 * i + 1] = s.y; scales[3 * i + 2] = s.z;
        float dn;
        const V4 q = {rot[4 * i], rot[4 * i + 1], rot[4 * i + 2], rot[4 * i + 3]};
        const V4 y = act_normalize4(q, &dn);
        rotations[4 * i] = y.x; rotations[4 * i + 1] = y.y; rotations[4 * i + 2] = y.z; rotations[4 * i + 3] = y.w;
    }
}

// g_* are gradients w.r.t. the activated values on entry and w.r.t. the raw values on return.
void gso_activate_bwd(int P, const float* logit, const float* logscale, const float* rot, float* g_opacity,
                      float* g_scales, float* g_rotations) {
#pragma omp parallel for
    for (int i = 0; i < P; ++i) {
        g_opacity[i] = act_sigmoid_bwd(act_sigmoid(logit[i]), g_opacity[i]);
        const V3 s = act_exp3(v3(logscale[3 * i], logscale[3 * i + 1], logscale[3 * i + 2]));
        const V3 gs = act_exp3_bwd(s, v3(g_scales[3 * i], g_scales[3 * i + 1], g_scales[3 * i + 2]));
        g_scales[3 * i] = gs.x; g_scales[3 * i + 1] = gs.y; g_scales[3 * i + 2] = gs.z;
        float dn;
        const V4 q = {rot[4 * i], rot[4 * i + 1], rot[4 * i + 2], rot[4 * i + 3]};
        const V4 y = act_normalize4(q, &dn);
        const V4 g = {g_rotations[4 * i], g_rotations[4 * i + 1], g_rotations[4 * i + 2], g_rotations[4 * i + 3]};
        const V4 r = act_normalize4_bwd(y, dn, g);
        g_rotations[4 * i] = r.x; g_rotations[4 * i + 1] = r.y; g_rotations[4 * i + 2] = r.z; g_rotations[4 * i + 3] = r.w;
    }
}

// Adam step over one [N, M] tensor (torch.optim.Adam's single-tensor operation
// order; visible == NULL updates every row), for checking the fused kernel.
void gso_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                     int64_t N, int M, double lr, double beta1, double beta2, double eps, double bc1, double bc2_sqrt) {
    const AdamConst c = adam_const(lr, beta1, beta2, eps, bc1, bc2_sqrt);
#pragma omp parallel for
    for (int64_t r = 0; r < N; ++r) {
        if (visible && !visible[r]) continue;
        for (int64_t i = r * M; i < (r + 1) * M; ++i)
            param[i] = adam_update(param[i], grad[i], &exp_avg[i], &exp_avg_sq[i], c);
    }
}

// ------------------------------------------------------------------ simple-knn

// stride > 1: only every stride-th point of the Morton order is searched (the
// same search, for checking large inputs in bounded time); the others get -1.
void gso_dist2_strided(int P, const float* pts, float* mean_dists, int stride) {
    if (P <= 0) return;
    if (stride > 1)
        for (int i = 0; i < P; ++i) mean_dists[i] = -1.0f;
    auto pt = [&](uint32_t i) { return v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]); };
    V3 mn = v3(0, 0, 0), mx = v3(0, 0, 0);  // reduce seeded with the origin (simple_knn.cu:192-200)
    for (int i = 0; i < P; ++i) {
        const V3 p = pt(i);
        mn = v3(fmin_(mn.x, p.x), fmin_(mn.y, p.y), fmin_(mn.z, p.z));
        mx = v3(fmax_(mx.x, p.x), fmax_(mx.y, p.y), fmax_(mx.z, p.z));
    }
    std::vector<uint64_t> codes(P), codes_sorted(P);
    std::vector<uint32_t> idx(P), order(P);
    for (int i = 0; i < P; ++i) {
        codes[i] = morton_code(pt(i), mn, mx);
        idx[i] = i;
    }
    radix_sort_pairs(codes.data(), idx.data(), codes_sorted.data(), order.data(), P, 32);
    const int BOX = 1024;
    const int nb = (P + BOX - 1) / BOX;
    std::vector<Box> boxes(nb);
    for (int b = 0; b < nb; ++b) {
        Box bx = {v3(FLT_MAX, FLT_MAX, FLT_MAX), v3(-FLT_MAX, -FLT_MAX, -FLT_MAX)};
        for (int i = b * BOX; i < std::min(P, (b + 1) * BOX); ++i) {
            const V3 p = pt(order[i]);
            bx.mn = v3(fmin_(bx.mn.x, p.x), fmin_(bx.mn.y, p.y), fmin_(bx.mn.z, p.z));
            bx.mx = v3(fmax_(bx.mx.x, p.x), fmax_(bx.mx.y, p.y), fmax_(bx.mx.z, p.z));
        }
        boxes[b] = bx;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < P; ++i) {
        if (stride > 1 && i % stride) continue;
        const V3 p = pt(order[i]);
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = std::max(0, i - 3); j <= std::min(P - 1, i + 3); ++j) {
            if (j == i) continue;
            knn3_update(p, pt(order[j]), best);
        }
        const float reject = best[2];
        best[0] = best[1] = best[2] = FLT_MAX;
        for (int b = 0; b < nb; ++b) {
            const float d = box_point_dist2(boxes[b], p);
            if (d > reject || d > best[2]) continue;
            for (int j = b * BOX; j < std::min(P, (b + 1) * BOX); ++j) {
                if (j == i) continue;
                knn3_update(p, pt(order[j]), best);
            }
        }
        mean_dists[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

void gso_dist2(int P, const float* pts, float* mean_dists) { gso_dist2_strided(P, pts, mean_dists, 1); }

// ------------------------------------------------------------------ fused-ssim

// One 11-tap zero-padded separable pass pair over a [H,W] plane.
static void sep_conv(const float* in, float* out, int H, int W, std::vector<float>& tmp) {
    tmp.assign((size_t)H * W, 0.0f);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float v = 0.0f;
            for (int k = 0; k < GS_SSIM_TAPS; ++k) {
                const int xx = x + k - 5;
                const float p = (xx >= 0 && xx < W) ? in[(size_t)y * W + xx] : 0.0f;
                v = fmaf(ssim_tap(k), p, v);   // one fused multiply-add per tap, as the kernels
            }
            tmp[(size_t)y * W + x] = v;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float v = 0.0f;
            for (int k = 0; k < GS_SSIM_TAPS; ++k) {
                const int yy = y + k - 5;
                const float p = (yy >= 0 && yy < H) ? tmp[(size_t)yy * W + x] : 0.0f;
                v = fmaf(ssim_tap(k), p, v);   // one fused multiply-add per tap, as the kernels
            }
            out[(size_t)y * W + x] = v;
        }
}

void gso_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, float* ssim_map,
                  float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12) {
    const size_t N = (size_t)H * W;
#pragma omp parallel for
    for (int pl = 0; pl < B * CH; ++pl) {
        const float* a = img1 + pl * N;
        const float* b = img2 + pl * N;
        std::vector<float> tmp, sq(N), mu1(N), ex2(N), mu2(N), ey2(N), exy(N);
        sep_conv(a, mu1.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) sq[i] = a[i] * a[i];
        sep_conv(sq.data(), ex2.data(), H, W, tmp);
        sep_conv(b, mu2.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) sq[i] = b[i] * b[i];
        sep_conv(sq.data(), ey2.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) sq[i] = a[i] * b[i];
        sep_conv(sq.data(), exy.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) {
            const SsimOut o = ssim_point(mu1[i], ex2[i], mu2[i], ey2[i], exy[i], C1, C2);
            ssim_map[pl * N + i] = o.m;
            if (dm_dmu1) {
                dm_dmu1[pl * N + i] = o.dm_dmu1;
                dm_dsigma1_sq[pl * N + i] = o.dm_dsigma1_sq;
                dm_dsigma12[pl * N + i] = o.dm_dsigma12;
            }
        }
    }
}

void gso_ssim_bwd(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                  const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1) {
    const size_t N = (size_t)H * W;
#pragma omp parallel for
    for (int pl = 0; pl < B * CH; ++pl) {
        std::vector<float> tmp, prod(N), c1(N), c2(N), c3(N);
        const float* dl = dL_dmap + pl * N;
        for (size_t i = 0; i < N; ++i) prod[i] = dm_dmu1[pl * N + i] * dl[i];
        sep_conv(prod.data(), c1.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) prod[i] = dm_dsigma1_sq[pl * N + i] * dl[i];
        sep_conv(prod.data(), c2.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) prod[i] = dm_dsigma12[pl * N + i] * dl[i];
        sep_conv(prod.data(), c3.data(), H, W, tmp);
        for (size_t i = 0; i < N; ++i) {
            float d = 0.0f;
            d += c1[i];
            d += img1[pl * N + i] * 2.0f * c2[i];
            d += img2[pl * N + i] * c3[i];
            dL_dimg1[pl * N + i] = d;
        }
    }
}

int gso_num_threads() { return omp_get_max_threads(); }

}  // extern "C"
