// Per-Gaussian forward stages: preprocess (cull, EWA projection, conic, radius,
// tile rectangle, SH -> RGB), the prefix sums over touched tiles and visible
// Gaussians, the front-to-back ordering of the visible Gaussians, key emission
// in that order and tile-range detection.  Reference: cuda_rasterizer/forward.cu:151-308,
// rasterizer_impl.cu:54-137, 326-400.
//
// HBM-bound stages.  One thread per Gaussian; the per-Gaussian results that the
// blend kernels gather later are packed into one 64-byte SplatRec so that a
// gather touches a single line half instead of five arrays.
#include "kernels.h"

namespace gs {

struct PreArgs {
    int P, D, M, DIM, W, H;
    float mod, tanx, tany;
    int aa;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* reps;
    int raw;                // opacities / scales / rotations are pre-activation values
    const float* shs_rest;  // SH split into shs [P,1,3] and shs_rest [P,M-1,3]
    CamPtrs cam;
    int* radii;
    float* out_depths;
    SplatRec* rec;
    float* cov3D;
    uint2* rect;
    uint2* dt;
    GStat* gstat;
    uint8_t* clamped;
    uint32_t* block_sums;
    uint32_t* vis_sums;
    uint32_t* prefilter_flag;   // not null: the caller declared the points prefiltered; set to 1 if one is culled
};

// The 64-byte records of a wave's Gaussians (has: this lane has one, q0 .. q3 its four quarters; wave_rec:
// the record of lane 0).  A lane that stores its own record issues four 16-byte stores 64 bytes apart, so
// every store instruction of the wave touches 64 line halves a quarter full.  Sixteen records at a time go
// through LDS instead (sr: 80 float4 of this wave's own, pitch five per record: conflict-free both ways)
// and leave as ONE store instruction of 64 consecutive 16-byte pieces.
constexpr int REC_STAGE_F4 = 16 * 5;
__device__ __forceinline__ void wave_store_records(SplatRec* wave_rec, float4* sr, bool has, float4 q0, float4 q1,
                                                   float4 q2, float4 q3, int lane) {
    const uint64_t recs = __ballot(has);
    float4* const out4 = reinterpret_cast<float4*>(wave_rec);
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        if (((recs >> (16 * round)) & 0xffffull) == 0) continue;   // wave-uniform
        if (has && (lane >> 4) == round) {
            float4* mine = sr + (lane & 15) * 5;
            mine[0] = q0;
            mine[1] = q1;
            mine[2] = q2;
            mine[3] = q3;
        }
        wave_lds_sync();
        const int r = 16 * round + (lane >> 2);   // the record this lane stores a quarter of
        if ((recs >> r) & 1ull) out4[4 * r + (lane & 3)] = sr[(lane >> 2) * 5 + (lane & 3)];
        wave_lds_sync();
    }
}

// MODE: how the SH rows are read (ShMode).  With 16 coefficients a wave first projects its 64 Gaussians,
// then moves the rows of those that are on screen from global memory into LDS as 16-byte chunks in
// address order (a lane walking its own 192-byte row issues 48 strided dword loads and drags whole
// lines of culled neighbours along), and every lane evaluates its colour from its row in LDS.
template <int MODE>
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_fwd_kernel(const PreArgs a) {
    constexpr int WAVE_LDS = ShLds<MODE>::bytes_per_wave;
    __shared__ __align__(16) char s_sh[WAVE_LDS ? 4 * WAVE_LDS : 16];
    __shared__ uint32_t s_tmp[8];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave_g0 = blockIdx.x * PRE_BLOCK + w * 64;
    const int nrows = min(64, a.P - wave_g0);  // <= 0 for waves past the end
    float* lds = reinterpret_cast<float*>(s_sh + w * WAVE_LDS);
    uint32_t tiles = 0;
    const bool live = idx < a.P;
    const Cam c = load_cam(a.cam, a.W, a.H, a.tanx, a.tany, a.mod, a.D, a.M, a.aa);
    bool vis = false;
    Splat sp;
    V3 p = v3(0, 0, 0);
    float cov6[6];
    if (live) {
        p = v3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        float depth;
        const bool culled = near_cull(p, c.view, &depth);
        // auxiliary.h:181-185: a point culled although the caller said the points were prefiltered.  The
        // reference prints and traps on the device; here the view's flag is raised and the host call reports it.
        if (culled && a.prefilter_flag) atomicOr(a.prefilter_flag, 1u);
        if (!culled) {
            if (a.cov3D_precomp) {
#pragma unroll
                for (int i = 0; i < 6; ++i) cov6[i] = a.cov3D_precomp[6 * (size_t)idx + i];
            } else {
                V3 sc = v3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
                const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
                V4 q = {q4.x, q4.y, q4.z, q4.w};
                if (a.raw) {
                    float dn;
                    sc = act_exp3(sc);
                    q = act_normalize4(q, &dn);
                }
                cov3d_from_scale_rot(sc, c.scale_modifier, q, cov6);
            }
            float opac = a.opacities[idx];
            if (a.raw) opac = act_sigmoid(opac);
            vis = project_splat(c, p, cov6, opac, depth, &sp);
        }
    }
    if (MODE == SH_CAT16 || MODE == SH_SPLIT16) {
        const uint64_t rows = __ballot(vis);
        if (rows && nrows > 0) {
            if (MODE == SH_CAT16)
                sh_rows_load(a.shs + (size_t)wave_g0 * SH_ROW_F, reinterpret_cast<float4*>(lds), rows, nrows, lane);
            else
                rest_rows_copy<0>(const_cast<float*>(a.shs_rest) + (size_t)wave_g0 * REST_ROW_F, lds, rows, nrows, lane);
        }
        wave_lds_sync();
    }
    if (live) {
        int radius = 0;
        uint4 bin = make_uint4(0, 0, 0, 0);
        if (vis) {
            const V3 campos = v3(c.campos[0], c.campos[1], c.campos[2]);
            V3 col;
            if (MODE == SH_NONE) {
                col = v3(a.colors_precomp[3 * idx], a.colors_precomp[3 * idx + 1], a.colors_precomp[3 * idx + 2]);
            } else {
                if (MODE == SH_CAT16) {
                    float shv[SH_ROW_F];
                    sh_row_get(reinterpret_cast<const float4*>(lds), lane, shv);
                    const ShRegs sh = {shv};
                    col = sh_to_rgb(c.sh_degree, p, campos, sh);
                } else if (MODE == SH_SPLIT16) {
                    // coefficient 0 from global memory, the others from the lane's row in LDS (row stride 45
                    // dwords: odd, no bank conflicts)
                    struct ShRow {
                        float dc0, dc1, dc2;
                        const float* row;
                        __device__ __forceinline__ V3 operator()(int k) const {
                            return k == 0 ? v3(dc0, dc1, dc2) : v3(row[3 * k - 3], row[3 * k - 2], row[3 * k - 1]);
                        }
                    };
                    const ShRow sh = {a.shs[3 * (size_t)idx], a.shs[3 * (size_t)idx + 1], a.shs[3 * (size_t)idx + 2],
                                      lds + lane * REST_ROW_F};
                    col = sh_to_rgb(c.sh_degree, p, campos, sh);
                } else {
                    ShPtr sh;
                    if (a.shs_rest) {
                        sh.dc = a.shs + 3 * (size_t)idx;
                        sh.rest = a.shs_rest + (size_t)idx * (a.M - 1) * 3;
                    } else {
                        sh.dc = a.shs + (size_t)idx * a.M * 3;
                        sh.rest = sh.dc + 3;
                    }
                    col = sh_to_rgb(c.sh_degree, p, campos, sh);
                }
                a.clamped[idx] = (uint8_t)((col.x < 0 ? 1 : 0) | (col.y < 0 ? 2 : 0) | (col.z < 0 ? 4 : 0));
                col = v3(fmax_(col.x, 0.0f), fmax_(col.y, 0.0f), fmax_(col.z, 0.0f));
            }
            float4 rp = make_float4(0, 0, 0, 0);
            if (a.DIM >= 4) rp = *reinterpret_cast<const float4*>(a.reps + (size_t)idx * a.DIM);   // rows of 4, 8 or 16 floats
            // (each lane stores its own record here: staging the records through LDS as the batch kernel does
            // costs this kernel, whose SH rows already take 53 KiB, a workgroup per CU: 0.51 -> 0.63 ms)
            float4* r4 = reinterpret_cast<float4*>(a.rec + idx);
            r4[0] = make_float4(sp.xy.x, sp.xy.y, sp.conic_opacity.x, sp.conic_opacity.y);
            r4[1] = make_float4(sp.conic_opacity.z, sp.conic_opacity.w, col.x, col.y);
            r4[2] = make_float4(col.z, 1 / sp.depth, rp.x, rp.y);
            r4[3] = make_float4(rp.z, rp.w, sp.depth, 0.0f);
            if (!a.cov3D_precomp) {
#pragma unroll
                for (int i = 0; i < 6; ++i) a.cov3D[6 * (size_t)idx + i] = cov6[i];
            }
            a.out_depths[idx] = sp.depth;
            radius = sp.radius;
            tiles = sp.tiles;
            bin.x = (uint32_t)sp.rect.x0 | ((uint32_t)sp.rect.y0 << 16);
            bin.y = (uint32_t)(sp.rect.x1 - sp.rect.x0) | ((uint32_t)(sp.rect.y1 - sp.rect.y0) << 16);
            bin.z = f2bits(sp.depth);
            bin.w = tiles;
        }
        a.radii[idx] = radius;
        a.rect[idx] = make_uint2(bin.x, bin.y);
        a.dt[idx] = make_uint2(bin.z, bin.w);
        a.gstat[idx] = GStat{0.0f, 0u};
    }
    // tiles touched by this block of 256 Gaussians
    const uint32_t ws = wave_sum_u32(tiles);
    const uint32_t wv = (uint32_t)__popcll(__ballot(tiles > 0));
    if ((threadIdx.x & 63) == 0) {
        s_tmp[threadIdx.x >> 6] = ws;
        s_tmp[4 + (threadIdx.x >> 6)] = wv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.block_sums[blockIdx.x] = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
        a.vis_sums[blockIdx.x] = s_tmp[4] + s_tmp[5] + s_tmp[6] + s_tmp[7];
    }
}

// ------------------------------------------------------------------ several views at once
// The forward preprocess of a batch of views in one pass (no counterpart in the
// reference, which renders one view per iteration): each Gaussian's parameters
// and SH row are read once for all the views of an optimiser step instead of
// once per view; per view only the results are written.  The same functions in
// the same order as preprocess_fwd_kernel: every per-view array comes out bit
// for bit as from the one-view kernel.  The kernel is bound by its arithmetic
// (about 650 instructions per Gaussian and view, most of them the projection),
// not by memory: everything is computed once, and each 64-byte SplatRec is
// written whole.
// Needs [P,1,3] + [P,15,3] SH tensors and scales + rotations; the 3D covariance
// is not stored (the backward recomputes it from scale and rotation).
constexpr int MAX_FWD_BATCH_VIEWS = 8;

struct FwdBatchView {
    CamPtrs cam;
    float tanx, tany;
    int W, H, aa;
    int* radii;
    float* out_depths;
    SplatRec* rec;
    uint2* rect;
    uint2* dt;
    GStat* gstat;
    uint8_t* clamped;
    uint32_t* block_sums;
    uint32_t* vis_sums;
};

struct PreBatchArgs {
    int P, D, M, DIM;
    float mod;
    int raw;
    const float* means3D;
    const float* shs;
    const float* shs_rest;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* reps;
    int nviews;
    uint32_t* far_flag;   // set when a depth in front of a near plane reaches DEPTH_KEY_FAR
    FwdBatchView v[MAX_FWD_BATCH_VIEWS];
};

// The SH row of a Gaussian goes straight into registers; the colours of all the views are evaluated
// first and wait in LDS (6 KiB per wave), then the views are projected one after the other: the row's 48
// registers and the projection's are never live together (82 VGPRs, 24 KiB of LDS per workgroup: 5 waves
// per SIMD).  A first version kept the wave's rows as an LDS image (11.25 KiB per wave) next to the
// projection's registers: 129 VGPRs, 3 waves per SIMD, 1.92 ms instead of 1.53 for 8 views.
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_fwd_batch_kernel(const PreBatchArgs b) {
    __shared__ float s_col[4][MAX_FWD_BATCH_VIEWS][3][64];
    __shared__ float4 s_rec[4][REC_STAGE_F4];   // wave_store_records
    __shared__ uint32_t s_tiles[MAX_FWD_BATCH_VIEWS][4];
    __shared__ uint32_t s_vis[MAX_FWD_BATCH_VIEWS][4];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool live = idx < b.P;
    const size_t i = (size_t)(live ? idx : 0);

    const V3 p = v3(b.means3D[3 * i], b.means3D[3 * i + 1], b.means3D[3 * i + 2]);
    float cov6[6];
    float opac = b.opacities[i];
    {
        V3 sc = v3(b.scales[3 * i], b.scales[3 * i + 1], b.scales[3 * i + 2]);
        const float4 q4 = reinterpret_cast<const float4*>(b.rotations)[i];
        V4 q = {q4.x, q4.y, q4.z, q4.w};
        if (b.raw) {
            float dn;
            sc = act_exp3(sc);
            q = act_normalize4(q, &dn);
            opac = act_sigmoid(opac);
        }
        cov3d_from_scale_rot(sc, b.mod, q, cov6);
    }
    uint32_t front = 0;   // bit v: in front of camera v's near plane
    if (live) {
#pragma unroll 1
        for (int v = 0; v < b.nviews; ++v) {
            float vm[16], depth;
#pragma unroll
            for (int k = 0; k < 16; ++k) vm[k] = b.v[v].cam.view[k];
            if (!near_cull(p, vm, &depth)) front |= 1u << v;
        }
    }
    if (front) {
        float shv[SH_ROW_F];
        shv[0] = b.shs[3 * i];
        shv[1] = b.shs[3 * i + 1];
        shv[2] = b.shs[3 * i + 2];
        const float* rest = b.shs_rest + i * REST_ROW_F;
#pragma unroll
        for (int k = 0; k < REST_ROW_F; ++k) shv[3 + k] = rest[k];
        const ShRegs sh = {shv};
#pragma unroll 1
        for (int v = 0; v < b.nviews; ++v) {
            if (!((front >> v) & 1u)) continue;
            const float* cp = b.v[v].cam.campos;
            const V3 col = sh_to_rgb(b.D, p, v3(cp[0], cp[1], cp[2]), sh);
            s_col[w][v][0][lane] = col.x;
            s_col[w][v][1][lane] = col.y;
            s_col[w][v][2][lane] = col.z;
        }
    }
    float4 rp = make_float4(0, 0, 0, 0);
    if (b.DIM >= 4) rp = *reinterpret_cast<const float4*>(b.reps + i * b.DIM);   // rows of 4, 8 or 16 floats
    bool far = false;

#pragma unroll 1
    for (int v = 0; v < b.nviews; ++v) {
        const FwdBatchView& bv = b.v[v];
        uint32_t tiles = 0;
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;   // this lane's record, if it has one
        bool has_rec = false;
        if (live) {
            const Cam c = load_cam(bv.cam, bv.W, bv.H, bv.tanx, bv.tany, b.mod, b.D, b.M, bv.aa);
            int radius = 0;
            uint4 bin = make_uint4(0, 0, 0, 0);
            float depth;
            if (!near_cull(p, c.view, &depth)) {
                Splat sp;
                if (project_splat(c, p, cov6, opac, depth, &sp)) {
                    V3 col = v3(s_col[w][v][0][lane], s_col[w][v][1][lane], s_col[w][v][2][lane]);
                    bv.clamped[i] = (uint8_t)((col.x < 0 ? 1 : 0) | (col.y < 0 ? 2 : 0) | (col.z < 0 ? 4 : 0));
                    col = v3(fmax_(col.x, 0.0f), fmax_(col.y, 0.0f), fmax_(col.z, 0.0f));
                    q0 = make_float4(sp.xy.x, sp.xy.y, sp.conic_opacity.x, sp.conic_opacity.y);
                    q1 = make_float4(sp.conic_opacity.z, sp.conic_opacity.w, col.x, col.y);
                    q2 = make_float4(col.z, 1 / sp.depth, rp.x, rp.y);
                    q3 = make_float4(rp.z, rp.w, sp.depth, 0.0f);
                    has_rec = true;
                    bv.out_depths[i] = sp.depth;
                    radius = sp.radius;
                    tiles = sp.tiles;
                    bin.x = (uint32_t)sp.rect.x0 | ((uint32_t)sp.rect.y0 << 16);
                    bin.y = (uint32_t)(sp.rect.x1 - sp.rect.x0) | ((uint32_t)(sp.rect.y1 - sp.rect.y0) << 16);
                    bin.z = f2bits(sp.depth);
                    bin.w = tiles;
                    far = far || bin.z >= DEPTH_KEY_FAR;
                }
            }
            bv.radii[i] = radius;
            bv.rect[i] = make_uint2(bin.x, bin.y);
            bv.dt[i] = make_uint2(bin.z, bin.w);
            bv.gstat[i] = GStat{0.0f, 0u};
        }
        wave_store_records(bv.rec + ((size_t)blockIdx.x * PRE_BLOCK + w * 64), s_rec[w], has_rec, q0, q1, q2, q3, lane);
        const uint32_t ws = wave_sum_u32(tiles);
        const uint32_t wv = (uint32_t)__popcll(__ballot(tiles > 0));
        if (lane == 0) {
            s_tiles[v][w] = ws;
            s_vis[v][w] = wv;
        }
    }
    if (__ballot(far) != 0 && lane == 0) atomicOr(b.far_flag, 1u);   // (practically never: depths beyond 13107)
    __syncthreads();
    if ((int)threadIdx.x < b.nviews) {
        const int v = threadIdx.x;
        b.v[v].block_sums[blockIdx.x] = s_tiles[v][0] + s_tiles[v][1] + s_tiles[v][2] + s_tiles[v][3];
        b.v[v].vis_sums[blockIdx.x] = s_vis[v][0] + s_vis[v][1] + s_vis[v][2] + s_vis[v][3];
    }
}

// Exclusive scan of the per-block sums, total instance count into total[0],
// overflow flag into total[1].  One workgroup of 1024 threads, 32 consecutive
// sums per thread and step (16-byte loads and stores), so 23 k sums (6 M Gaussians) take one step; the
// running values are 64-bit so that an overflowing total is detected, not wrapped.
constexpr int SCAN_THREADS = 1024, SCAN_ITEMS = 32;
// Workgroup 1, if launched, scans a second array the same way (the per-block
// counts of Gaussians with instances) and writes its total into total[2].
// `all`, if given: the three totals once more, where the totals of several views are collected.
// (primary: the array of tile sums, whose total goes to total[0..1]; otherwise total[2])
__device__ __forceinline__ void scan_blocks_one(uint32_t* block_sums, bool primary, int nb, uint64_t* total,
                                                uint64_t* all) {
    __shared__ uint64_t s_wave[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint64_t carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS * SCAN_ITEMS) {
        const int i0 = base + tid * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        uint64_t mine = 0;
        if (i0 + SCAN_ITEMS <= nb) {  // whole 16-byte groups (the arrays start on 256-byte boundaries)
            const uint4* src = reinterpret_cast<const uint4*>(block_sums + i0);
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS / 4; ++k) {
                const uint4 q = src[k];
                v[4 * k] = q.x;
                v[4 * k + 1] = q.y;
                v[4 * k + 2] = q.z;
                v[4 * k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = i0 + k < nb ? block_sums[i0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) mine += v[k];
        uint64_t incl = mine;  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < SCAN_THREADS / 64; ++k) {
            const uint64_t t = s_wave[k];
            if (k < w) before += t;
            all += t;
        }
        __syncthreads();
        uint64_t run = carry + before + incl - mine;
        if (i0 + SCAN_ITEMS <= nb) {
            uint4* dst = reinterpret_cast<uint4*>(block_sums + i0);
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS / 4; ++k) {
                uint4 q;
                q.x = (uint32_t)run;
                run += v[4 * k];
                q.y = (uint32_t)run;
                run += v[4 * k + 1];
                q.z = (uint32_t)run;
                run += v[4 * k + 2];
                q.w = (uint32_t)run;
                run += v[4 * k + 3];
                dst[k] = q;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) {
                if (i0 + k < nb) block_sums[i0 + k] = (uint32_t)run;
                run += v[k];
            }
        }
        carry += all;
    }
    if (tid == 0) {
        block_sums[nb] = (uint32_t)carry;
        if (primary) {
            total[0] = carry;
            total[1] = carry > 0x7fffffffull ? 1 : 0;
            if (all) {
                all[0] = total[0];
                all[1] = total[1];
            }
        } else {
            total[2] = carry;
            if (all) all[2] = carry;
        }
    }
}

__device__ __forceinline__ void scan_blocks_body(uint32_t* sums0, uint32_t* sums1, int nb, uint64_t* total,
                                                 uint64_t* all) {
    scan_blocks_one(blockIdx.x == 0 ? sums0 : sums1, blockIdx.x == 0, nb, total, all);
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_blocks_kernel(uint32_t* sums0, uint32_t* sums1, int nb,
                                                                   uint64_t* total) {
    scan_blocks_body(sums0, sums1, nb, total, nullptr);
}

// The scans of several views in one launch (blockIdx.y = view); every view's totals also go to
// all[3 * view ..], so that the host fetches them with one copy.
struct ScanBatchArgs {
    uint32_t* sums0[SCAN_BATCH_MAX];
    uint32_t* sums1[SCAN_BATCH_MAX];
    uint64_t* total[SCAN_BATCH_MAX];
    uint64_t* all;
};
__global__ void __launch_bounds__(SCAN_THREADS) scan_blocks_batch_kernel(const ScanBatchArgs b, int nb) {
    const int v = blockIdx.y;
    scan_blocks_body(b.sums0[v], b.sums1[v], nb, b.total[v], b.all + 3 * v);
}

// Depth keys for ordering the Gaussians front to back: the f32 bits of the
// view-space depth (positive floats order like their bit patterns), all ones
// for Gaussians that produce no instance so that they sort to the end.
// bias / none: subtracted from the depth bits / the key of a Gaussian without instances (0 and all ones for
// keys on 32 bits, DEPTH_KEY_BIAS and 2^27 - 1 for the narrow keys).
__device__ __forceinline__ void depth_keys_body(int P, const uint2* dt, const uint32_t* vis_off, uint32_t* keys,
                                                uint32_t* vals, uint32_t bias = 0u, uint32_t none = 0xFFFFFFFFu) {
    // vis_off == null: one pair per Gaussian (those without instances get the largest key).
    // Otherwise only Gaussians with instances get a pair, packed in index order: a block's pairs
    // start at vis_off[block] (blocks = the preprocess blocks that counted them).
    __shared__ uint32_t s_w[4];
    const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const uint2 b = i < P ? dt[i] : make_uint2(0, 0);   // depth bits, tiles
    if (!vis_off) {
        if (i < P) {
            keys[i] = b.y ? b.x - bias : none;
            vals[i] = (uint32_t)i;
        }
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t base = vis_off[blockIdx.x];   // (in flight during the ballot and the barrier)
    const uint64_t m = __ballot(b.y > 0);
    if (lane == 0) s_w[w] = (uint32_t)__popcll(m);
    __syncthreads();
    for (int k = 0; k < w; ++k) base += s_w[k];
    if (b.y > 0) {
        const uint32_t dst = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        keys[dst] = b.x - bias;
        vals[dst] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(PRE_BLOCK) depth_keys_kernel(int P, const uint2* dt, const uint32_t* vis_off,
                                                               uint32_t* keys, uint32_t* vals) {
    depth_keys_body(P, dt, vis_off, keys, vals);
}

struct DepthKeysBatch {   // blockIdx.y = view
    uint32_t bias, none;
    const uint2* dt[SORT_BATCH_MAX];
    const uint32_t* vis_off[SORT_BATCH_MAX];
    uint32_t* keys[SORT_BATCH_MAX];
    uint32_t* vals[SORT_BATCH_MAX];
};

__global__ void __launch_bounds__(PRE_BLOCK) depth_keys_batch_kernel(int P, const DepthKeysBatch b) {
    const int v = blockIdx.y;
    depth_keys_body(P, b.dt[v], b.vis_off[v], b.keys[v], b.vals[v], b.bias, b.none);
}

// The tile rectangles in depth order: sbin[i] = {rect origin, rect size, Gaussian
// index, tiles} of the i-th Gaussian front to back (the one random gather of the
// emission), and the tiles touched per block of 256 of them.
__global__ void __launch_bounds__(PRE_BLOCK) order_gather_kernel(int P, const uint32_t* order, const uint2* rect,
                                                                 uint4* sbin, uint32_t* sums) {
    __shared__ uint32_t s_tmp[4];
    const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
    uint32_t t = 0;
    if (i < P) {
        const uint32_t gid = order[i];
        const uint2 r = rect[gid];                      // 8 bytes per random access
        t = (r.y & 0xffffu) * (r.y >> 16);              // tiles = width * height
        sbin[i] = make_uint4(r.x, r.y, gid, t);
    }
    const uint32_t ws = wave_sum_u32(t);
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

// duplicateWithKeys (rasterizer_impl.cu:70-110), restated for the hardware:
//  * Gaussians are visited in depth order (ties in index order), so the emitted
//    list is already sorted by (depth, index).  The reference's 64-bit key is
//    (tile << 32 | depth bits); with the depth part already in order the stable
//    sort that follows only has to look at the tile, so the key that is written
//    and sorted is the 32-bit tile id: 2 passes over 8-byte pairs instead of 6
//    over 12-byte pairs, with the same final list;
//  * the instances of a block of 256 Gaussians are numbered consecutively and
//    dealt to the threads round-robin, so work and stores are uniform however
//    large individual splats are.
__global__ void __launch_bounds__(PRE_BLOCK) emit_keys_kernel(int P, int grid_x, const uint4* sbin,
                                                              const uint32_t* block_offsets, uint32_t* keys,
                                                              uint32_t* vals) {
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_off[PRE_BLOCK];
    __shared__ uint4 s_bin[PRE_BLOCK];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const uint4 b = idx < P ? sbin[idx] : make_uint4(0, 0, 0, 0);
    uint32_t total;
    const uint32_t ex = block_scan_excl_256(b.w, s_tmp, &total);
    s_off[threadIdx.x] = ex;
    s_bin[threadIdx.x] = b;
    __syncthreads();
    const size_t base = block_offsets[blockIdx.x];
    for (uint32_t s = threadIdx.x; s < total; s += PRE_BLOCK) {
        // largest g with s_off[g] <= s
        int lo = 0;
#pragma unroll
        for (int step = PRE_BLOCK / 2; step >= 1; step >>= 1)
            if (s_off[lo + step] <= s) lo += step;
        const uint4 gb = s_bin[lo];
        const uint32_t q = s - s_off[lo];
        const uint32_t w = gb.y & 0xffffu;
        const uint32_t ty = q / w;
        const uint32_t tx = q - ty * w;
        keys[base + s] = ((gb.x >> 16) + ty) * (uint32_t)grid_x + (gb.x & 0xffffu) + tx;
        vals[base + s] = gb.z;
    }
}

// Tile ranges and the sort's digit histograms without reading the instances: how many instances
// a tile gets is how many tile rectangles cover it, and that follows from the rectangles alone.
// Every rectangle marks its four corners in a (grid_y + 1) x (grid_x + 1) array (+1 at two, -1 at
// the other two); the two-dimensional prefix sum of the marks is the cover count per tile, the
// running sum of the counts in tile order gives the ranges that identifyTileRanges
// (rasterizer_impl.cu:115-137) finds by scanning the sorted keys, and the counts summed per
// digit are the histograms the radix passes start from.  Per view this reads 8 bytes per
// Gaussian instead of two sweeps over the instances (histogram and range scan).
constexpr int TC_THREADS = 1024;
constexpr int TC_BLOCKS = 256;
constexpr int TF_THREADS = 1024;
constexpr size_t TILE_LDS_LIMIT = 150 * 1024;

size_t tile_finish_lds(int gx, int gy, int passes) {
    return ((size_t)(gx + 1) * (gy + 1) + (size_t)passes * 256 + TF_THREADS / 64 + 1 + 2 * 260) * sizeof(int);
}

bool tile_counts_fit(int W, int H, int passes) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    return tile_finish_lds(gx, gy, passes) <= TILE_LDS_LIMIT;
}

__global__ void __launch_bounds__(TC_THREADS) tile_count_kernel(int P, const uint2* __restrict__ rect, int gx, int gy,
                                                               int* __restrict__ diff) {
    extern __shared__ int s_diff[];
    const int nd = (gx + 1) * (gy + 1), pitch = gx + 1;
    for (int i = threadIdx.x; i < nd; i += TC_THREADS) s_diff[i] = 0;
    __syncthreads();
    // four independent loads in flight per lane: the loop is bound by their latency
    const int stride = gridDim.x * TC_THREADS;
    for (int i0 = blockIdx.x * TC_THREADS + threadIdx.x; i0 < P; i0 += 4 * stride) {
        uint2 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * stride;
            b[u] = i < P ? rect[i] : make_uint2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (b[u].y == 0) continue;   // no instances: a rendered Gaussian's rectangle has width and height > 0
            const int x0 = (int)(b[u].x & 0xffffu), y0 = (int)(b[u].x >> 16);
            const int x1 = x0 + (int)(b[u].y & 0xffffu), y1 = y0 + (int)(b[u].y >> 16);
            atomicAdd(&s_diff[y0 * pitch + x0], 1);
            atomicAdd(&s_diff[y0 * pitch + x1], -1);
            atomicAdd(&s_diff[y1 * pitch + x0], -1);
            atomicAdd(&s_diff[y1 * pitch + x1], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += TC_THREADS) {
        const int v = s_diff[i];
        if (v) atomicAdd(&diff[i], v);
    }
}

// One workgroup: prefix sums of the corner marks along x and y, then the running sum over the
// tiles in key order.  A tile without instances keeps the range (0, 0), as in the reference,
// whose ranges are zero-filled and written only where the sorted keys change.
// With seg_start: also where the instances of each value of the high digit (tile id >> lowbits, a
// "segment") start, and the first sort-pass tile of each segment when no tile straddles two.
__device__ __forceinline__ void tile_finish_body(int gx, int gy, const int* __restrict__ diff,
                                                 uint2* __restrict__ ranges, uint32_t* __restrict__ ghist,
                                                 int passes, int nbits, int lowbits, int nseg,
                                                 uint32_t* __restrict__ seg_start,
                                                 uint32_t* __restrict__ seg_tile0) {
    extern __shared__ int s_all[];
    const int pitch = gx + 1, nd = pitch * (gy + 1), ntiles = gx * gy;
    int* s_c = s_all;                                                  // [nd] marks, then cover counts
    uint32_t* s_h = reinterpret_cast<uint32_t*>(s_all + nd);           // [passes][256]
    uint32_t* s_w = s_h + passes * 256;                                // [waves + 1]
    uint32_t* s_seg = s_w + TF_THREADS / 64 + 1;                       // [nseg + 1]
    uint32_t* s_t0 = s_seg + 260;                                      // [nseg + 1]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < nd; i += TF_THREADS) s_c[i] = diff[i];
    for (int i = tid; i < passes * 256; i += TF_THREADS) s_h[i] = 0;
    __syncthreads();
    // running sums along x, then along y: one wave per line, 64 elements per step
    constexpr int NWAVES = TF_THREADS / 64;
    for (int r = w; r <= gy; r += NWAVES) {
        uint32_t carry = 0;
        for (int x0 = 0; x0 <= gx; x0 += 64) {
            const int x = x0 + lane;
            const uint32_t v = x <= gx ? (uint32_t)s_c[r * pitch + x] : 0u;
            const uint32_t incl = wave_scan_incl_u32(v) + carry;
            if (x <= gx) s_c[r * pitch + x] = (int)incl;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __syncthreads();
    for (int c = w; c <= gx; c += NWAVES) {
        uint32_t carry = 0;
        for (int y0 = 0; y0 <= gy; y0 += 64) {
            const int y = y0 + lane;
            const uint32_t v = y <= gy ? (uint32_t)s_c[y * pitch + c] : 0u;
            const uint32_t incl = wave_scan_incl_u32(v) + carry;
            if (y <= gy) s_c[y * pitch + c] = (int)incl;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __syncthreads();
    // thread t owns `chunk` consecutive tiles
    const int chunk = (ntiles + TF_THREADS - 1) / TF_THREADS;
    const int t0 = tid * chunk, t1 = min(ntiles, t0 + chunk);
    uint32_t mine = 0;
    for (int t = t0; t < t1; ++t) mine += (uint32_t)s_c[(t / gx) * pitch + (t % gx)];
    const uint32_t incl = wave_scan_incl_u32(mine);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t start = incl - mine;
    for (int k = 0; k < w; ++k) start += s_w[k];
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = (uint32_t)s_c[(t / gx) * pitch + (t % gx)];
        ranges[t] = c ? make_uint2(start, start + c) : make_uint2(0u, 0u);
        if (seg_start && (t & ((1 << lowbits) - 1)) == 0) s_seg[t >> lowbits] = start;
        if (c) {
            for (int p = 0; p < passes; ++p) {
                int shift, bits;
                digit_of_pass(nbits, passes, p, &shift, &bits);
                atomicAdd(&s_h[p * 256 + (((uint32_t)t >> shift) & ((1u << bits) - 1))], c);
            }
        }
        start += c;
    }
    __syncthreads();
    for (int i = tid; i < passes * 256; i += TF_THREADS) ghist[i] = s_h[i];
    if (!seg_start) return;
    uint32_t all = 0;
    for (int k = 0; k < TF_THREADS / 64; ++k) all += s_w[k];
    for (int h = tid; h <= nseg; h += TF_THREADS)
        if ((h << lowbits) >= ntiles) s_seg[h] = all;  // segments past the last tile are empty
    __syncthreads();
    if (w == 0) {
        const int per = (nseg + 63) / 64;
        uint32_t cnt = 0;
        for (int k = 0; k < per; ++k) {
            const int h = lane * per + k;
            if (h < nseg) cnt += (s_seg[h + 1] - s_seg[h] + SORT_TILE - 1) / SORT_TILE;
        }
        uint32_t run = wave_scan_incl_u32(cnt) - cnt;
        for (int k = 0; k < per; ++k) {
            const int h = lane * per + k;
            if (h < nseg) {
                s_t0[h] = run;
                run += (s_seg[h + 1] - s_seg[h] + SORT_TILE - 1) / SORT_TILE;
            }
            if (h == nseg - 1) s_t0[nseg] = run;
        }
    }
    __syncthreads();
    for (int h = tid; h <= nseg; h += TF_THREADS) {
        seg_start[h] = s_seg[h];
        seg_tile0[h] = s_t0[h];
    }
}

__global__ void __launch_bounds__(TF_THREADS) tile_finish_kernel(int gx, int gy, const int* __restrict__ diff,
                                                                 uint2* __restrict__ ranges, uint32_t* __restrict__ ghist,
                                                                 int passes, int nbits, int lowbits, int nseg,
                                                                 uint32_t* __restrict__ seg_start,
                                                                 uint32_t* __restrict__ seg_tile0) {
    tile_finish_body(gx, gy, diff, ranges, ghist, passes, nbits, lowbits, nseg, seg_start, seg_tile0);
}

// The two one-workgroup steps in front of the key emission that partitions, side by side in one launch instead
// of one after the other with the rest of the GPU idle: workgroup 0 is tile_finish_kernel; workgroup 1 scans
// the tile sums of the depth-ordered blocks of 256 Gaussians into instance offsets (scan_blocks_kernel) and
// from them writes, for every chunk of EMIT_CHUNK instances, the block its first instance falls in and that
// block's offset (the emission's chunk table).
static_assert(TF_THREADS == SCAN_THREADS, "one launch, one workgroup size");
__global__ void __launch_bounds__(TF_THREADS) tile_finish_offsets_kernel(
    int gx, int gy, const int* __restrict__ diff, uint2* __restrict__ ranges, uint32_t* __restrict__ ghist, int passes,
    int nbits, int lowbits, int nseg, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_tile0,
    uint32_t* order_sums, int nb256, uint64_t* total, uint2* __restrict__ chunk_first, uint32_t chunks) {
    if (blockIdx.x == 0) {
        tile_finish_body(gx, gy, diff, ranges, ghist, passes, nbits, lowbits, nseg, seg_start, seg_tile0);
        return;
    }
    scan_blocks_one(order_sums, true, nb256, total, nullptr);
    __syncthreads();   // the offsets are this workgroup's own stores
    for (int b = threadIdx.x; b < nb256; b += TF_THREADS) {
        const uint32_t s0 = order_sums[b], s1 = order_sums[b + 1];
        for (uint32_t c = (s0 + EMIT_CHUNK - 1) / EMIT_CHUNK; (uint64_t)c * EMIT_CHUNK < s1; ++c)
            chunk_first[c] = make_uint2((uint32_t)b, s0);
        if (b == nb256 - 1) chunk_first[chunks] = make_uint2((uint32_t)b, s0);  // "the chunk after the last" starts in the last block
    }
}

hipError_t launch_tile_count(const GeomBuf& g, int P, int W, int H, const ImageView& im, hipStream_t st) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const size_t nd = (size_t)(gx + 1) * (gy + 1);
    hipError_t e = hipMemsetAsync(im.tile_diff, 0, nd * sizeof(int), st);
    if (e != hipSuccess) return e;
    const size_t lds_c = nd * sizeof(int);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_c);
    if (e != hipSuccess) return e;
    const int blocks = min(TC_BLOCKS, (P + TC_THREADS - 1) / TC_THREADS);
    hipLaunchKernelGGL(tile_count_kernel, dim3(blocks), dim3(TC_THREADS), lds_c, st, P, g.rect, gx, gy, im.tile_diff);
    return hipGetLastError();
}

hipError_t launch_tile_finish_offsets(const GeomBuf& g, int n_visible, int W, int H, const ImageView& im, uint32_t* ghist,
                                      int passes, int nbits, int lowbits, uint32_t* seg_start, uint32_t* seg_tile0,
                                      uint2* chunk_first, uint32_t chunks, hipStream_t st) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const size_t lds_f = tile_finish_lds(gx, gy, passes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_finish_offsets_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    if (e != hipSuccess) return e;
    const int nb256 = (n_visible + PRE_BLOCK - 1) / PRE_BLOCK;
    hipLaunchKernelGGL(tile_finish_offsets_kernel, dim3(2), dim3(TF_THREADS), lds_f, st, gx, gy, im.tile_diff, im.ranges,
                       ghist, passes, nbits, lowbits, 1 << (nbits - lowbits), seg_start, seg_tile0, g.order_sums, nb256,
                       g.total, chunk_first, chunks);
    return hipGetLastError();
}

hipError_t launch_tile_ranges_from_rects(const GeomBuf& g, int P, int W, int H, const ImageView& im, uint32_t* ghist,
                                         int passes, int nbits, hipStream_t st, int lowbits, uint32_t* seg_start,
                                         uint32_t* seg_tile0) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const size_t nd = (size_t)(gx + 1) * (gy + 1);
    hipError_t e = hipMemsetAsync(im.tile_diff, 0, nd * sizeof(int), st);
    if (e != hipSuccess) return e;
    const size_t lds_c = nd * sizeof(int), lds_f = tile_finish_lds(gx, gy, passes);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_f);
    if (e != hipSuccess) return e;
    const int blocks = min(TC_BLOCKS, (P + TC_THREADS - 1) / TC_THREADS);
    hipLaunchKernelGGL(tile_count_kernel, dim3(blocks), dim3(TC_THREADS), lds_c, st, P, g.rect, gx, gy, im.tile_diff);
    hipLaunchKernelGGL(tile_finish_kernel, dim3(1), dim3(TF_THREADS), lds_f, st, gx, gy, im.tile_diff, im.ranges, ghist,
                       passes, nbits, lowbits, seg_start ? 1 << (nbits - lowbits) : 0, seg_start, seg_tile0);
    return hipGetLastError();
}

// rasterizer_impl.cu:115-137, on the sorted tile ids.
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* keys, int64_t R, uint2* ranges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = keys[i];
    if (i == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = keys[i - 1];
        if (cur != prev) {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
    }
    if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

// The reference's 64-bit sort keys, rebuilt from the tile ranges (the tile an instance of the sorted
// list belongs to) and the depth of each instance's Gaussian (for inspection and tests).  One workgroup
// per tile walks that tile's part of the list.
__global__ void __launch_bounds__(256) keys64_kernel(const uint2* ranges, const uint32_t* point_list,
                                                     const uint2* dt, int64_t R, uint64_t* out) {
    const uint32_t tile = blockIdx.x;
    const uint2 r = ranges[tile];
    for (int64_t i = (int64_t)r.x + threadIdx.x; i < (int64_t)r.y && i < R; i += 256)
        out[i] = ((uint64_t)tile << 32) | dt[point_list[i]].x;
}

// Also the last kernel of a forward pass: it hands the sorts' error words (instance sort, depth order) to the
// caller's host word (device-accessible pinned memory; a system-scope store, no copy engine, nothing for the
// stream to wait on).
__global__ void __launch_bounds__(256) finish_gstat_kernel(int P, const GStat* gstat, float* gauss_sum,
                                                           int* gauss_count, const uint32_t* sort_err,
                                                           const uint32_t* depth_err, uint32_t* sort_err_host) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && sort_err_host)
        __hip_atomic_store(sort_err_host, (sort_err ? *sort_err : 0u) | (depth_err ? *depth_err : 0u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (i >= P) return;
    const GStat v = gstat[i];
    gauss_sum[i] = v.sum;
    gauss_count[i] = (int)v.count;
}

// rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* means3D, const float* view,
                                                           uint8_t* present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float vm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) vm[k] = view[k];
    float d;
    present[i] = near_cull(v3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]), vm, &d) ? 0 : 1;
}

// ------------------------------------------------------------------ launchers

hipError_t launch_preprocess_fwd(const GsrScene& s, const GsrForwardOut& o, const GeomBuf& g, hipStream_t st) {
    PreArgs a;
    a.P = s.P; a.D = s.D; a.M = s.M; a.DIM = s.DIM; a.W = s.W; a.H = s.H;
    a.mod = s.scale_modifier; a.tanx = s.tanfovx; a.tany = s.tanfovy; a.aa = s.antialiasing;
    a.means3D = s.means3D; a.shs = s.shs; a.colors_precomp = s.colors_precomp; a.opacities = s.opacities;
    a.scales = s.scales; a.rotations = s.rotations; a.cov3D_precomp = s.cov3D_precomp; a.reps = s.reps;
    a.raw = s.raw_params; a.shs_rest = s.shs_rest;
    a.cam.view = s.viewmatrix; a.cam.proj = s.projmatrix; a.cam.campos = s.campos;
    a.radii = o.radii; a.out_depths = o.out_depths;
    a.rec = g.rec; a.cov3D = g.cov3D; a.rect = g.rect; a.dt = g.dt; a.gstat = g.gstat; a.clamped = g.clamped;
    a.block_sums = g.block_sums;
    a.vis_sums = g.vis_sums;
    a.prefilter_flag = s.prefiltered ? reinterpret_cast<uint32_t*>(g.total + 3) : nullptr;   // zeroed by the caller
    const int nb = (s.P + PRE_BLOCK - 1) / PRE_BLOCK;
    const dim3 grid(nb), block(PRE_BLOCK);
    if (!s.shs) hipLaunchKernelGGL(preprocess_fwd_kernel<SH_NONE>, grid, block, 0, st, a);
    else if (s.M == 16 && s.shs_rest) hipLaunchKernelGGL(preprocess_fwd_kernel<SH_SPLIT16>, grid, block, 0, st, a);
    else if (s.M == 16) hipLaunchKernelGGL(preprocess_fwd_kernel<SH_CAT16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(preprocess_fwd_kernel<SH_DIRECT>, grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_preprocess_fwd_batch(const GsrScene& s, int nviews, const GsrForwardView* views, uint32_t* far_flag,
                                       hipStream_t st) {
    PreBatchArgs b;
    b.far_flag = far_flag;
    b.P = s.P; b.D = s.D; b.M = s.M; b.DIM = s.DIM;
    b.mod = s.scale_modifier; b.raw = s.raw_params;
    b.means3D = s.means3D; b.shs = s.shs; b.shs_rest = s.shs_rest; b.opacities = s.opacities;
    b.scales = s.scales; b.rotations = s.rotations; b.reps = s.reps;
    const int nb = (s.P + PRE_BLOCK - 1) / PRE_BLOCK;
    for (int v0 = 0; v0 < nviews; v0 += MAX_FWD_BATCH_VIEWS) {
        b.nviews = nviews - v0 < MAX_FWD_BATCH_VIEWS ? nviews - v0 : MAX_FWD_BATCH_VIEWS;
        for (int k = 0; k < b.nviews; ++k) {
            const GsrForwardView& fv = views[v0 + k];
            const GeomBuf g = geom_buf(fv.geom_buffer, (size_t)s.P);
            FwdBatchView& bv = b.v[k];
            bv.cam.view = fv.viewmatrix; bv.cam.proj = fv.projmatrix; bv.cam.campos = fv.campos;
            bv.tanx = fv.tanfovx; bv.tany = fv.tanfovy; bv.W = fv.W; bv.H = fv.H; bv.aa = fv.antialiasing;
            bv.radii = fv.radii; bv.out_depths = fv.out_depths;
            bv.rec = g.rec; bv.rect = g.rect; bv.dt = g.dt; bv.gstat = g.gstat; bv.clamped = g.clamped;
            bv.block_sums = g.block_sums; bv.vis_sums = g.vis_sums;
        }
        hipLaunchKernelGGL(preprocess_fwd_batch_kernel, dim3(nb), dim3(PRE_BLOCK), 0, st, b);
    }
    return hipGetLastError();
}

hipError_t launch_scan_blocks(const GeomBuf& g, int nb, hipStream_t st) {
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(2), dim3(SCAN_THREADS), 0, st, g.block_sums, g.vis_sums, nb, g.total);
    return hipGetLastError();
}

hipError_t launch_scan_blocks_batch(int count, const GeomBuf* g, int nb, hipStream_t st, uint64_t** all) {
    if (count <= 0 || count > SCAN_BATCH_MAX) return hipErrorInvalidValue;
    ScanBatchArgs b;
    for (int i = 0; i < count; ++i) {
        b.sums0[i] = g[i].block_sums;
        b.sums1[i] = g[i].vis_sums;
        b.total[i] = g[i].total;
    }
    b.all = g[0].total + 4;   // behind the first view's own totals
    *all = b.all;
    hipLaunchKernelGGL(scan_blocks_batch_kernel, dim3(2, count), dim3(SCAN_THREADS), 0, st, b, nb);
    return hipGetLastError();
}

// In-place exclusive scan of sums[0..nb) (sums[nb] = total), total[0] = total, total[1] = overflow flag.
hipError_t launch_scan_u32(uint32_t* sums, int nb, uint64_t* total, hipStream_t st) {
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, sums, (uint32_t*)nullptr, nb, total);
    return hipGetLastError();
}

hipError_t launch_depth_order(const GeomBuf& g, int P, int n, hipStream_t st, int* order_buffer) {
    hipLaunchKernelGGL(depth_keys_kernel, dim3((P + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, st, P, g.dt,
                       n < P ? g.vis_sums : (const uint32_t*)nullptr, g.dkeys[0], g.dvals[0]);
    uint32_t* k[2] = {g.dkeys[0], g.dkeys[1]};
    uint32_t* v[2] = {g.dvals[0], g.dvals[1]};
    return launch_radix_sort_u32(k, v, n, 0, 32, g.dsort_temp, st, order_buffer, false);
}

hipError_t launch_depth_order_batch(int count, const GeomBuf* g, int P, const int* n, bool narrow, hipStream_t st) {
    if (count <= 0 || P <= 0) return hipSuccess;
    if (count > SORT_BATCH_MAX) return hipErrorInvalidValue;
    DepthKeysBatch b;
    b.bias = narrow ? DEPTH_KEY_BIAS : 0u;
    b.none = narrow ? (1u << SORT_NARROW_KEY_BITS) - 1u : 0xFFFFFFFFu;
    const int first = narrow ? 1 : 0;   // three passes from buffer 1, four from buffer 0: the order ends in buffer 0
    uint32_t* keys[SORT_BATCH_MAX][2];
    uint32_t* vals[SORT_BATCH_MAX][2];
    int64_t n64[SORT_BATCH_MAX];
    void* temp[SORT_BATCH_MAX];
    for (int i = 0; i < count; ++i) {
        b.dt[i] = g[i].dt;
        b.vis_off[i] = n[i] < P ? g[i].vis_sums : (const uint32_t*)nullptr;
        b.keys[i] = g[i].dkeys[first];
        b.vals[i] = g[i].dvals[first];
        keys[i][0] = g[i].dkeys[0]; keys[i][1] = g[i].dkeys[1];
        vals[i][0] = g[i].dvals[0]; vals[i][1] = g[i].dvals[1];
        n64[i] = n[i];
        temp[i] = g[i].dsort_temp;
    }
    hipLaunchKernelGGL(depth_keys_batch_kernel, dim3((P + PRE_BLOCK - 1) / PRE_BLOCK, count), dim3(PRE_BLOCK), 0, st, P, b);
    return launch_radix_sort_u32_batch(count, keys, vals, n64, temp, st, narrow ? SORT_NARROW_KEY_BITS : 32);
}

hipError_t launch_order_gather(const GeomBuf& g, const uint32_t* order, int n, hipStream_t st) {
    const int nb = (n + PRE_BLOCK - 1) / PRE_BLOCK;
    hipLaunchKernelGGL(order_gather_kernel, dim3(nb), dim3(PRE_BLOCK), 0, st, n, order, g.rect, g.sbin, g.order_sums);
    return hipGetLastError();
}

hipError_t launch_emit_keys(const GsrScene& s, const GeomBuf& g, const uint32_t* order, int n, uint32_t* keys,
                            uint32_t* vals, hipStream_t st) {
    const int nb = (n + PRE_BLOCK - 1) / PRE_BLOCK;
    const int grid_x = (s.W + TILE_X - 1) / TILE_X;
    hipLaunchKernelGGL(order_gather_kernel, dim3(nb), dim3(PRE_BLOCK), 0, st, n, order, g.rect, g.sbin, g.order_sums);
    // total[0..1] were read by the host already; the scan rewrites the same values
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, g.order_sums, (uint32_t*)nullptr, nb,
                       g.total);
    hipLaunchKernelGGL(emit_keys_kernel, dim3(nb), dim3(PRE_BLOCK), 0, st, n, grid_x, g.sbin, g.order_sums, keys,
                       vals);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(const uint32_t* keys, int64_t R, uint2* ranges, int ntiles, hipStream_t st) {
    hipError_t e = hipMemsetAsync(ranges, 0, (size_t)ntiles * sizeof(uint2), st);
    if (e != hipSuccess || R <= 0) return e;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, keys, R, ranges);
    return hipGetLastError();
}

hipError_t launch_keys64(const GeomBuf& g, const uint2* ranges, int ntiles, const uint32_t* point_list, int64_t R,
                         uint64_t* out, hipStream_t st) {
    if (R <= 0 || ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(keys64_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, ranges, point_list, g.dt, R, out);
    return hipGetLastError();
}

hipError_t launch_finish_gstat(const GeomBuf& g, int P, float* gauss_sum, int* gauss_count, const uint32_t* sort_err,
                               const uint32_t* depth_err, uint32_t* sort_err_host, hipStream_t st) {
    hipLaunchKernelGGL(finish_gstat_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, g.gstat, gauss_sum,
                       gauss_count, sort_err, depth_err, sort_err_host);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, view, present);
    return hipGetLastError();
}

}  // namespace gs
