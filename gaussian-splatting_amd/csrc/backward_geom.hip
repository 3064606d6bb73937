// Per-Gaussian backward: conic -> cov2D -> cov3D / mean (EWA), 2D mean -> 3D
// mean, colour -> SH, cov3D -> scale / rotation.  Reference:
// cuda_rasterizer/backward.cu:147-329 (computeCov2DCUDA) and :401-471
// (preprocessCUDA), which run as two kernels over the same data; here they are
// one pass, one lane per Gaussian.  HBM-bound: reads the 64-byte GradRec the
// blend backward filled plus the Gaussian's inputs and writes every gradient
// tensor completely (zeros for Gaussians that were not rendered), so the caller
// never zero-fills.  With the usual 16 SH coefficients the 192-byte SH rows come
// in and the 192-byte dL/dSH rows go out through an LDS image, as whole
// 16-byte chunks in address order (sh_rows_load / sh_rows_store); other SH
// sizes use sh_bwd_kernel (16 lanes per Gaussian, one coefficient each).
#include "kernels.h"

namespace gs {

struct PreBwdArgs {
    int P, D, M, DIM, W, H;
    float mod, tanx, tany;
    int aa;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D;  // supplied or the one the forward computed
    CamPtrs cam;
    const int* radii;
    const uint8_t* clamped;
    const GradRec* acc;
    int has_invdepth;
    float* dL_dmeans2D;
    float* dL_dcolors;
    float* dL_dopacity;
    float* dL_dmeans3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscales;
    float* dL_drotations;
    float* dL_dreps;
};

// backward.cu:23-142 with one coefficient per lane.  Writes dL/dSH and leaves
// the direction term of dL/dmean in dL_dmeans3D (the per-Gaussian kernel that
// runs next adds the other two terms to it).
__global__ void __launch_bounds__(256) sh_bwd_kernel(const PreBwdArgs a) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i = t >> 4;
    const int k = (int)(t & 15);
    if (i >= (size_t)a.P) return;
    float* dsh = a.dL_dsh + i * a.M * 3;
    if (!(a.radii[i] > 0)) {
        for (int kk = k; kk < a.M; kk += 16) dsh[3 * kk] = dsh[3 * kk + 1] = dsh[3 * kk + 2] = 0.f;
        return;
    }
    const float* acc = reinterpret_cast<const float*>(a.acc + i);
    const uint8_t cl = a.clamped[i];
    V3 g = v3(acc[6], acc[7], acc[8]);
    g.x *= (cl & 1) ? 0 : 1;
    g.y *= (cl & 2) ? 0 : 1;
    g.z *= (cl & 4) ? 0 : 1;
    const V3 mean = v3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);
    const V3 dir_orig = sub(mean, v3(a.cam.campos[0], a.cam.campos[1], a.cam.campos[2]));
    const float len = sqrtf(dot(dir_orig, dir_orig));
    const V3 dir = v3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
    const int ncoef = (a.D + 1) * (a.D + 1);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (k < ncoef) {
        float B, Bx, By, Bz;
        sh_basis_grad(k, dir.x, dir.y, dir.z, &B, &Bx, &By, &Bz);
        const float* sh = a.shs + i * a.M * 3 + 3 * k;
        const float s = sh[0] * g.x + sh[1] * g.y + sh[2] * g.z;
        dsh[3 * k] = B * g.x;
        dsh[3 * k + 1] = B * g.y;
        dsh[3 * k + 2] = B * g.z;
        sx = Bx * s;
        sy = By * s;
        sz = Bz * s;
    }
    for (int kk = k < ncoef ? k + 16 : k; kk < a.M; kk += 16) dsh[3 * kk] = dsh[3 * kk + 1] = dsh[3 * kk + 2] = 0.f;
    // sum over the 16 lanes of this Gaussian
    sx += lane_xor1(sx); sy += lane_xor1(sy); sz += lane_xor1(sz);
    sx += lane_xor2(sx); sy += lane_xor2(sy); sz += lane_xor2(sz);
    sx += lane_xor4(sx); sy += lane_xor4(sy); sz += lane_xor4(sz);
    sx += lane_xor8(sx); sy += lane_xor8(sy); sz += lane_xor8(sz);
    if (k == 0) {
        const V3 d = dnormvdv(dir_orig, v3(sx, sy, sz));
        a.dL_dmeans3D[3 * i] = d.x;
        a.dL_dmeans3D[3 * i + 1] = d.y;
        a.dL_dmeans3D[3 * i + 2] = d.z;
    }
}

// Everything for one rendered Gaussian.  LDS_SH: its SH row is in `lds` and its
// dL/dSH row is left there; otherwise sh_bwd_kernel did the SH part.
template <bool LDS_SH>
__device__ __forceinline__ void pre_bwd_rendered(const PreBwdArgs& a, size_t i, float4* lds, int lane) {
    const Cam c = load_cam(a.cam, a.W, a.H, a.tanx, a.tany, a.mod, a.D, a.M, a.aa);
    const float4* gq = reinterpret_cast<const float4*>(a.acc + i);
    const float4 g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
    const V3 mean = v3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);

    // SH first: its 96 live registers are released before the covariance chain starts.
    V3 dmean = v3(0, 0, 0);
    if (LDS_SH) {
        float shv[SH_ROW_F];
        sh_row_get(lds, lane, shv);
        const uint8_t cl = a.clamped[i];
        V3 gcol = v3(g1.z, g1.w, g2.x);
        gcol.x *= (cl & 1) ? 0 : 1;
        gcol.y *= (cl & 2) ? 0 : 1;
        gcol.z *= (cl & 4) ? 0 : 1;
        // The row is in registers now; dL/dSH overwrites it in LDS coefficient by
        // coefficient (only this lane touches its row between the two wave-wide copies).
        float* row = reinterpret_cast<float*>(lds + lane * SH_LDS_STRIDE);
        const int ncoef = (c.sh_degree + 1) * (c.sh_degree + 1);
#pragma unroll
        for (int ch = 0; ch < SH_ROW_CHUNKS; ++ch)
            if (4 * ch + 3 >= 3 * ncoef) lds[lane * SH_LDS_STRIDE + ch] = make_float4(0, 0, 0, 0);
        auto out = [row](int k, V3 v) {
            row[3 * k] = v.x;
            row[3 * k + 1] = v.y;
            row[3 * k + 2] = v.z;
        };
        ShRegs sh = {shv};
        dmean = sh_to_rgb_bwd(c.sh_degree, mean, v3(c.campos[0], c.campos[1], c.campos[2]), gcol, sh, out);
    } else if (a.shs) {  // direction term, left there by sh_bwd_kernel
        dmean = v3(a.dL_dmeans3D[3 * i], a.dL_dmeans3D[3 * i + 1], a.dL_dmeans3D[3 * i + 2]);
    }
    if (LDS_SH) wave_lds_sync();
    float cov6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov6[k] = a.cov3D[6 * i + k];
    float dop = g1.y;
    const float dinv = g2.y;
    float dcov6[6];
    const V3 dconic = v3(g0.z, g0.w, g1.x);
    dmean = add(dmean, cov2d_bwd(c, mean, cov6, a.opacities[i], dconic, &dop, a.has_invdepth ? &dinv : nullptr, dcov6));
    dmean = add(dmean, mean2d_bwd(mean, c.proj, g0.x, g0.y));

    if (a.scales) {
        const V3 sc = v3(a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]);
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[i];
        const V4 q = {q4.x, q4.y, q4.z, q4.w};
        V3 ds;
        V4 dq;
        cov3d_from_scale_rot_bwd(sc, c.scale_modifier, q, dcov6, &ds, &dq);
        a.dL_dscales[3 * i] = ds.x;
        a.dL_dscales[3 * i + 1] = ds.y;
        a.dL_dscales[3 * i + 2] = ds.z;
        reinterpret_cast<float4*>(a.dL_drotations)[i] = make_float4(dq.x, dq.y, dq.z, dq.w);
    }
    a.dL_dmeans2D[3 * i] = g0.x;
    a.dL_dmeans2D[3 * i + 1] = g0.y;
    a.dL_dmeans2D[3 * i + 2] = 0.f;
    a.dL_dcolors[3 * i] = g1.z;
    a.dL_dcolors[3 * i + 1] = g1.w;
    a.dL_dcolors[3 * i + 2] = g2.x;
    a.dL_dopacity[i] = dop;
    a.dL_dmeans3D[3 * i] = dmean.x;
    a.dL_dmeans3D[3 * i + 1] = dmean.y;
    a.dL_dmeans3D[3 * i + 2] = dmean.z;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = dcov6[k];
    if (a.DIM == 4) reinterpret_cast<float4*>(a.dL_dreps)[i] = make_float4(g2.z, g2.w, g3.x, g3.y);
}

// A Gaussian the forward did not render: every gradient is zero.
template <bool LDS_SH>
__device__ __forceinline__ void pre_bwd_unrendered(const PreBwdArgs& a, size_t i, float4* lds, int lane) {
    if (LDS_SH) {
#pragma unroll
        for (int ch = 0; ch < SH_ROW_CHUNKS; ++ch) lds[lane * SH_LDS_STRIDE + ch] = make_float4(0, 0, 0, 0);
    }
    a.dL_dmeans2D[3 * i] = a.dL_dmeans2D[3 * i + 1] = a.dL_dmeans2D[3 * i + 2] = 0.f;
    a.dL_dcolors[3 * i] = a.dL_dcolors[3 * i + 1] = a.dL_dcolors[3 * i + 2] = 0.f;
    a.dL_dopacity[i] = 0.f;
    a.dL_dmeans3D[3 * i] = a.dL_dmeans3D[3 * i + 1] = a.dL_dmeans3D[3 * i + 2] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = 0.f;
    if (a.dL_dscales) {
        a.dL_dscales[3 * i] = a.dL_dscales[3 * i + 1] = a.dL_dscales[3 * i + 2] = 0.f;
        reinterpret_cast<float4*>(a.dL_drotations)[i] = make_float4(0, 0, 0, 0);
    }
    if (a.DIM == 4) reinterpret_cast<float4*>(a.dL_dreps)[i] = make_float4(0, 0, 0, 0);
}

#ifndef PRE_BWD_WAVES
#define PRE_BWD_WAVES 2
#endif
// LDS_SH: shs is [P,16,3] and is handled inside this kernel through LDS;
// otherwise the SH part was done by sh_bwd_kernel (or there are no SHs).
template <bool LDS_SH>
__global__ void __launch_bounds__(256, PRE_BWD_WAVES) preprocess_bwd_kernel(const PreBwdArgs a) {
    __shared__ float4 s_sh[LDS_SH ? 4 * SH_WAVE_LDS_F4 : 1];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave_g0 = blockIdx.x * 256 + w * 64;  // first Gaussian of this wave
    const int nrows = min(64, a.P - wave_g0);        // <= 0 for waves past the end
    if (nrows <= 0) return;
    const bool live = idx < a.P;
    const bool rendered = live && a.radii[live ? idx : 0] > 0;
    float4* lds = s_sh + (LDS_SH ? w * SH_WAVE_LDS_F4 : 0);
    if (LDS_SH) {
        const uint64_t rows = __ballot(rendered);
        if (rows) sh_rows_load(a.shs + (size_t)wave_g0 * SH_ROW_F, lds, rows, nrows, lane);
        wave_lds_sync();
    }
    if (rendered) pre_bwd_rendered<LDS_SH>(a, (size_t)idx, lds, lane);
    else if (live) pre_bwd_unrendered<LDS_SH>(a, (size_t)idx, lds, lane);
    if (LDS_SH) {
        wave_lds_sync();
        sh_rows_store(a.dL_dsh + (size_t)wave_g0 * SH_ROW_F, lds, nrows, lane);
    }
}

hipError_t launch_preprocess_bwd(const GsrScene& s, const int* radii, const GeomBuf& g, const GsrBackwardIO& io,
                                 const GradRec* acc, hipStream_t st) {
    PreBwdArgs a;
    a.P = s.P; a.D = s.D; a.M = s.M; a.DIM = s.DIM; a.W = s.W; a.H = s.H;
    a.mod = s.scale_modifier; a.tanx = s.tanfovx; a.tany = s.tanfovy; a.aa = s.antialiasing;
    a.means3D = s.means3D; a.shs = s.shs; a.colors_precomp = s.colors_precomp; a.opacities = s.opacities;
    a.scales = s.scales; a.rotations = s.rotations;
    a.cov3D = s.cov3D_precomp ? s.cov3D_precomp : g.cov3D;
    a.cam.view = s.viewmatrix; a.cam.proj = s.projmatrix; a.cam.campos = s.campos;
    a.radii = radii; a.clamped = g.clamped; a.acc = acc;
    a.has_invdepth = io.dL_dout_invdepth != nullptr;
    a.dL_dmeans2D = io.dL_dmeans2D; a.dL_dcolors = io.dL_dcolors; a.dL_dopacity = io.dL_dopacity;
    a.dL_dmeans3D = io.dL_dmeans3D; a.dL_dcov3D = io.dL_dcov3D; a.dL_dsh = io.dL_dsh;
    a.dL_dscales = s.scales ? io.dL_dscales : nullptr;
    a.dL_drotations = s.scales ? io.dL_drotations : nullptr;
    a.dL_dreps = io.dL_dreps;
    const bool lds_sh = s.shs && s.M == 16;
    if (s.shs && !lds_sh) {
        const size_t threads = (size_t)s.P * 16;
        hipLaunchKernelGGL(sh_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
    }
    if (lds_sh) hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3((s.P + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((s.P + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace gs
