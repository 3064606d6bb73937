// Per-Gaussian backward: conic -> cov2D -> cov3D / mean (EWA), 2D mean -> 3D
// mean, colour -> SH, cov3D -> scale / rotation.  Reference:
// cuda_rasterizer/backward.cu:147-329 (computeCov2DCUDA) and :401-471
// (preprocessCUDA).  HBM-bound.  Two kernels:
//   sh_bwd_kernel         16 lanes per Gaussian, one SH coefficient per lane, so
//                         the 192-byte SH rows are read and the 192-byte dL/dSH
//                         rows written as contiguous lines; the direction
//                         gradient is summed over the 16 lanes with DPP adds.
//   preprocess_bwd_kernel one lane per Gaussian for everything else.
// Together they read the 64-byte GradRec the blend backward filled plus the
// Gaussian's inputs and write every gradient tensor completely (zeros for
// Gaussians that were not rendered), so the caller never zero-fills.
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

__global__ void __launch_bounds__(256) preprocess_bwd_kernel(const PreBwdArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.P) return;
    const size_t i = (size_t)idx;
    if (!(a.radii[idx] > 0)) {
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
        return;
    }
    const Cam c = load_cam(a.cam, a.W, a.H, a.tanx, a.tany, a.mod, a.D, a.M, a.aa);
    const float4* gq = reinterpret_cast<const float4*>(a.acc + i);
    const float4 g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
    const V3 mean = v3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);

    float cov6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov6[k] = a.cov3D[6 * i + k];
    float dop = g1.y;
    const float dinv = g2.y;
    float dcov6[6];
    const V3 dconic = v3(g0.z, g0.w, g1.x);
    V3 dmean = cov2d_bwd(c, mean, cov6, a.opacities[i], dconic, &dop, a.has_invdepth ? &dinv : nullptr, dcov6);
    dmean = add(dmean, mean2d_bwd(mean, c.proj, g0.x, g0.y));

    if (a.shs) {  // direction term, left there by sh_bwd_kernel
        dmean = add(dmean, v3(a.dL_dmeans3D[3 * i], a.dL_dmeans3D[3 * i + 1], a.dL_dmeans3D[3 * i + 2]));
    }
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
    if (s.shs) {
        const size_t threads = (size_t)s.P * 16;
        hipLaunchKernelGGL(sh_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((s.P + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace gs
