// Scalar math of the Gaussian rasterizer, shared by the HIP kernels (hipcc,
// gfx950) and by the CPU checker (g++).  Everything in here is plain f32
// arithmetic written out in one fixed operation order; both compilers are run
// with -ffp-contract=off so that the only fused operations are the explicit
// fmaf() calls of gs_expf().  That makes every integer that is derived from
// floats (radii, tile rectangles, depth key bits, n_contrib) come out bit for
// bit the same on the CPU and on the GPU.
//
// The formulas follow the reference rasterizer:
//   submodules/diff-gaussian-rasterization/cuda_rasterizer/auxiliary.h
//   submodules/diff-gaussian-rasterization/cuda_rasterizer/forward.cu
//   submodules/diff-gaussian-rasterization/cuda_rasterizer/backward.cu
// and GLM's column-major 3x3 algebra
//   third_party/glm/glm/detail/type_mat3x3.inl:486-519.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD inline
#endif

namespace gs {

constexpr int TILE_X = 16;   // config.h:17
constexpr int TILE_Y = 16;   // config.h:18
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr int NUM_CH = 3;    // config.h:15

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

// 3x3 matrix stored as GLM does: c[col][row].
struct M3 { float c[3][3]; };

// ---------------------------------------------------------------- scalars

GS_HD float fmin_(float a, float b) { return a < b ? a : b; }
GS_HD float fmax_(float a, float b) { return a > b ? a : b; }
GS_HD int imin_(int a, int b) { return a < b ? a : b; }
GS_HD int imax_(int a, int b) { return a > b ? a : b; }

GS_HD uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
GS_HD float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// float -> int32 with the saturating behaviour of the GPU conversion
// instruction (NaN -> 0), so that out-of-range projections give the same
// tile rectangle on the host.
GS_HD int f2i_sat(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the conversion instruction itself (NaN -> 0, out of range -> the nearest int): what the three
    // comparisons below spell out for the host
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}

// exp() for the blending weight.  Range-reduced minimax polynomial with
// explicit fused multiply-adds; max error 1 ulp on [-87, 0].  Both the kernels
// and the checker call this one function, so alpha, T and hence the
// termination decisions (alpha < 1/255, T < 1e-4) agree exactly.
GS_HD float gs_expf(float x) {
    x = x < -87.0f ? -87.0f : (x > 88.0f ? 88.0f : x);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045e-06f, r);
    float p = 0.000198527617612853646278381f;
    p = fmaf(p, r, 0.00139304355252534151077271f);
    p = fmaf(p, r, 0.00833336077630519866943359f);
    p = fmaf(p, r, 0.0416664853692054748535156f);
    p = fmaf(p, r, 0.166666671633720397949219f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(r * r, p, r) + 1.0f;
    const int e = (int)n + 127;
    return p * bits2f((uint32_t)e << 23);
}

// ---------------------------------------------------------------- vectors

GS_HD V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
GS_HD V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
GS_HD V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
GS_HD V3 scale(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
GS_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ---------------------------------------------------------------- mat3

GS_HD M3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2,
                 float c0, float c1, float c2) {
    M3 m;
    m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
    m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
    m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
    return m;
}

GS_HD M3 m3_T(const M3& a) {
    M3 t;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) t.c[c][r] = a.c[r][c];
    return t;
}

// type_mat3x3.inl:486-519 -- R[c][r] = A[0][r]B[c][0] + A[1][r]B[c][1] + A[2][r]B[c][2]
GS_HD M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
    for (int c = 0; c < 3; ++c)
        for (int w = 0; w < 3; ++w)
            r.c[c][w] = a.c[0][w] * b.c[c][0] + a.c[1][w] * b.c[c][1] + a.c[2][w] * b.c[c][2];
    return r;
}

GS_HD M3 m3_scale(float s, const M3& a) {
    M3 r;
    for (int c = 0; c < 3; ++c)
        for (int w = 0; w < 3; ++w) r.c[c][w] = a.c[c][w] * s;
    return r;
}

// ---------------------------------------------------------------- transforms
// Matrices are 16 floats, element (row r, col c) of the mathematical matrix at
// m[4*c + r] (auxiliary.h:70-109).

GS_HD V3 xform_pt_4x3(V3 p, const float* m) {
    return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
              m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}

GS_HD V4 xform_pt_4x4(V3 p, const float* m) {
    V4 r;
    r.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    r.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    r.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    r.w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    return r;
}

GS_HD V3 xform_vec_4x3_T(V3 p, const float* m) {
    return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z,
              m[4] * p.x + m[5] * p.y + m[6] * p.z,
              m[8] * p.x + m[9] * p.y + m[10] * p.z);
}

// Backward of d = v / |v|: the Jacobian is (I - d d^T) / |v|, so the gradient with respect to v is the part
// of the incoming gradient that is orthogonal to d, over |v|.  (The reference's dnormvdv,
// auxiliary.h:119-129, expands the same product per component.)
GS_HD V3 unit_vector_bwd(V3 v, V3 g_unit) {
    const float len = sqrtf(dot(v, v));
    const V3 d = v3(v.x / len, v.y / len, v.z / len);
    const float along = dot(d, g_unit);
    return v3((g_unit.x - along * d.x) / len, (g_unit.y - along * d.y) / len, (g_unit.z - along * d.z) / len);
}

// auxiliary.h:40-43 -- evaluated in double, rounded once.
GS_HD float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

struct Rect { int x0, y0, x1, y1; };

// auxiliary.h:45-55
GS_HD Rect tile_rect(V2 p, int max_radius, int grid_x, int grid_y) {
    Rect r;
    r.x0 = imin_(grid_x, imax_(0, f2i_sat((p.x - max_radius) / TILE_X)));
    r.y0 = imin_(grid_y, imax_(0, f2i_sat((p.y - max_radius) / TILE_Y)));
    r.x1 = imin_(grid_x, imax_(0, f2i_sat((p.x + max_radius + TILE_X - 1) / TILE_X)));
    r.y1 = imin_(grid_y, imax_(0, f2i_sat((p.y + max_radius + TILE_Y - 1) / TILE_Y)));
    return r;
}

// rasterizer_impl.cu:35-50
inline uint32_t higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// ---------------------------------------------------------------- SH
// auxiliary.h:21-38
#define GS_SH_C0 0.28209479177387814f
#define GS_SH_C1 0.4886025119029199f
#define GS_SH_C2_0 1.0925484305920792f
#define GS_SH_C2_1 -1.0925484305920792f
#define GS_SH_C2_2 0.31539156525252005f
#define GS_SH_C2_3 -1.0925484305920792f
#define GS_SH_C2_4 0.5462742152960396f
#define GS_SH_C3_0 -0.5900435899266435f
#define GS_SH_C3_1 2.890611442640554f
#define GS_SH_C3_2 -0.4570457994644658f
#define GS_SH_C3_3 0.3731763325901154f
#define GS_SH_C3_4 -0.4570457994644658f
#define GS_SH_C3_5 1.445305721320277f
#define GS_SH_C3_6 -0.5900435899266435f

// acc + k * v per channel as ONE fused multiply-add (and acc - k * v as fma(-k, v, acc), the same value).
// The library is built without floating-point contraction because the integer outputs of the projection
// have to come out bit for bit on every compiler; the colour feeds none of them, the reference's own
// compiler contracts these chains (forward.cu:20-71 is plain "result + C * sh" code under nvcc's default
// fmad), and the checker calls this same function.  Half the instructions of the SH evaluation.
GS_HD V3 axpy(V3 acc, float k, V3 v) { return v3(fmaf(k, v.x, acc.x), fmaf(k, v.y, acc.y), fmaf(k, v.z, acc.z)); }
GS_HD V3 axmy(V3 acc, float k, V3 v) { return axpy(acc, -k, v); }

// forward.cu:20-71.  SH is any callable k -> V3 giving coefficient k of this
// Gaussian.  Returns the unclamped colour + 0.5; the caller clamps at zero and
// records which channels were clamped.
template <class SH>
GS_HD V3 sh_to_rgb(int deg, V3 pos, V3 campos, SH sh) {
    V3 dir = sub(pos, campos);
    const float len = sqrtf(dot(dir, dir));
    dir = v3(dir.x / len, dir.y / len, dir.z / len);

    V3 res = scale(GS_SH_C0, sh(0));
    if (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        res = axmy(res, GS_SH_C1 * y, sh(1));
        res = axpy(res, GS_SH_C1 * z, sh(2));
        res = axmy(res, GS_SH_C1 * x, sh(3));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            res = axpy(res, GS_SH_C2_0 * xy, sh(4));
            res = axpy(res, GS_SH_C2_1 * yz, sh(5));
            res = axpy(res, GS_SH_C2_2 * (2.0f * zz - xx - yy), sh(6));
            res = axpy(res, GS_SH_C2_3 * xz, sh(7));
            res = axpy(res, GS_SH_C2_4 * (xx - yy), sh(8));
            if (deg > 2) {
                res = axpy(res, GS_SH_C3_0 * y * (3.0f * xx - yy), sh(9));
                res = axpy(res, GS_SH_C3_1 * xy * z, sh(10));
                res = axpy(res, GS_SH_C3_2 * y * (4.0f * zz - xx - yy), sh(11));
                res = axpy(res, GS_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), sh(12));
                res = axpy(res, GS_SH_C3_4 * x * (4.0f * zz - xx - yy), sh(13));
                res = axpy(res, GS_SH_C3_5 * z * (xx - yy), sh(14));
                res = axpy(res, GS_SH_C3_6 * x * (xx - 3.0f * yy), sh(15));
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    return res;
}

// Backward of sh_to_rgb (what backward.cu:23-142 computes), written from the structure of the forward:
// colour = 0.5 + sum_k B_k(d) sh_k with d the unit view direction, so
//     dL/dsh_k = B_k(d) * g                       (g = dL/dRGB, clamped channels already zeroed)
//     dL/dd    = sum_k (sh_k . g) * grad B_k(d)
// and dL/dposition follows through d = (pos - campos) / |pos - campos|.  The coefficients are
// projected on g first (one scalar per coefficient), so the direction gradient is a sum of scalars
// times the polynomial gradients below instead of three-vectors.  OUT is any callable (k, V3) that
// stores dL/dsh[k].  Returns the part of dL/dmean that flows through the view direction.
template <class SH, class OUT>
GS_HD V3 sh_to_rgb_bwd(int deg, V3 pos, V3 campos, V3 g, SH sh, OUT out) {
    const V3 to_point = sub(pos, campos);
    const float len = sqrtf(dot(to_point, to_point));
    const float x = to_point.x / len, y = to_point.y / len, z = to_point.z / len;
    V3 gd = v3(0, 0, 0);   // dL/dd
    out(0, scale(GS_SH_C0, g));
    if (deg > 0) {
        out(1, scale(-GS_SH_C1 * y, g));
        out(2, scale(GS_SH_C1 * z, g));
        out(3, scale(-GS_SH_C1 * x, g));
        const float s1 = dot(sh(1), g), s2 = dot(sh(2), g), s3 = dot(sh(3), g);
        gd = v3(-GS_SH_C1 * s3, -GS_SH_C1 * s1, GS_SH_C1 * s2);
    }
    if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        out(4, scale(GS_SH_C2_0 * xy, g));
        out(5, scale(GS_SH_C2_1 * yz, g));
        out(6, scale(GS_SH_C2_2 * (2.f * zz - xx - yy), g));
        out(7, scale(GS_SH_C2_3 * xz, g));
        out(8, scale(GS_SH_C2_4 * (xx - yy), g));
        const float s4 = GS_SH_C2_0 * dot(sh(4), g), s5 = GS_SH_C2_1 * dot(sh(5), g), s6 = GS_SH_C2_2 * dot(sh(6), g),
                    s7 = GS_SH_C2_3 * dot(sh(7), g), s8 = GS_SH_C2_4 * dot(sh(8), g);
        // B4 = xy, B5 = yz, B6 = 2zz - xx - yy, B7 = xz, B8 = xx - yy (times their constants)
        gd.x += s4 * y - 2.f * s6 * x + s7 * z + 2.f * s8 * x;
        gd.y += s4 * x + s5 * z - 2.f * s6 * y - 2.f * s8 * y;
        gd.z += s5 * y + 4.f * s6 * z + s7 * x;
        if (deg > 2) {
            out(9, scale(GS_SH_C3_0 * y * (3.f * xx - yy), g));
            out(10, scale(GS_SH_C3_1 * xy * z, g));
            out(11, scale(GS_SH_C3_2 * y * (4.f * zz - xx - yy), g));
            out(12, scale(GS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy), g));
            out(13, scale(GS_SH_C3_4 * x * (4.f * zz - xx - yy), g));
            out(14, scale(GS_SH_C3_5 * z * (xx - yy), g));
            out(15, scale(GS_SH_C3_6 * x * (xx - 3.f * yy), g));
            const float s9 = GS_SH_C3_0 * dot(sh(9), g), s10 = GS_SH_C3_1 * dot(sh(10), g),
                        s11 = GS_SH_C3_2 * dot(sh(11), g), s12 = GS_SH_C3_3 * dot(sh(12), g),
                        s13 = GS_SH_C3_4 * dot(sh(13), g), s14 = GS_SH_C3_5 * dot(sh(14), g),
                        s15 = GS_SH_C3_6 * dot(sh(15), g);
            // B9 = y(3xx - yy), B10 = xyz, B11 = y(4zz - xx - yy), B12 = z(2zz - 3xx - 3yy),
            // B13 = x(4zz - xx - yy), B14 = z(xx - yy), B15 = x(xx - 3yy)
            gd.x += 6.f * s9 * xy + s10 * yz - 2.f * s11 * xy - 6.f * s12 * xz + s13 * (4.f * zz - 3.f * xx - yy) +
                    2.f * s14 * xz + 3.f * s15 * (xx - yy);
            gd.y += 3.f * s9 * (xx - yy) + s10 * xz + s11 * (4.f * zz - xx - 3.f * yy) - 6.f * s12 * yz -
                    2.f * s13 * xy - 2.f * s14 * yz - 6.f * s15 * xy;
            gd.z += s10 * xy + 8.f * s11 * yz + 3.f * s12 * (2.f * zz - xx - yy) + 8.f * s13 * xz + s14 * (xx - yy);
        }
    }
    return unit_vector_bwd(to_point, gd);
}

// SH coefficient k of one Gaussian when the DC term and the other
// coefficients may live in two tensors (scene/gaussian_model.py:487-498 keeps
// _features_dc [P,1,3] and _features_rest [P,M-1,3]); for one [P,M,3] tensor
// rest = dc + 3.
struct ShPtr {
    const float* dc;
    const float* rest;
    GS_HD V3 operator()(int k) const {
        const float* p = k == 0 ? dc : rest + 3 * (k - 1);
        return v3(p[0], p[1], p[2]);
    }
};

// ---------------------------------------------------------------- activations
// Raw-parameter mode folds the activations the model applies before every
// render (scene/gaussian_model.py:41-53, 470-502) into the kernels:
// scaling = exp, opacity = sigmoid, rotation = x / max(|x|, 1e-12).

GS_HD float act_sigmoid(float x) { return 1.0f / (1.0f + gs_expf(-x)); }
GS_HD float act_sigmoid_bwd(float y, float g) { return g * ((1.0f - y) * y); }
GS_HD V3 act_exp3(V3 x) { return v3(gs_expf(x.x), gs_expf(x.y), gs_expf(x.z)); }
GS_HD V3 act_exp3_bwd(V3 y, V3 g) { return v3(g.x * y.x, g.y * y.y, g.z * y.z); }

GS_HD V4 act_normalize4(V4 q, float* denom) {
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    const float d = fmax_(n, 1e-12f);
    *denom = d;
    V4 r = {q.x / d, q.y / d, q.z / d, q.w / d};
    return r;
}
// y = act_normalize4(x), g = dL/dy.  Below the clamp the map is x / 1e-12 (linear).
GS_HD V4 act_normalize4_bwd(V4 y, float denom, V4 g) {
    float t = ((y.x * g.x + y.y * g.y) + y.z * g.z) + y.w * g.w;
    if (!(denom > 1e-12f)) t = 0.0f;
    V4 r = {(g.x - y.x * t) / denom, (g.y - y.y * t) / denom, (g.z - y.z * t) / denom, (g.w - y.w * t) / denom};
    return r;
}

// ---------------------------------------------------------------- optimizer
// One Adam update of one element, in the operation order of torch.optim.Adam's
// single-tensor path (exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2)
// .addcmul_(grad, grad, value = 1 - beta2); denom = sqrt(exp_avg_sq) /
// sqrt(bias_correction2) + eps; param.addcdiv_(exp_avg, denom, value = -lr /
// bias_correction1)).  The constants are formed in double on the host, as
// Python does there, and rounded once.  With step_size = lr and bc2_sqrt = 1
// it is the update of the sparse Gaussian Adam (no bias correction).
struct AdamConst {
    float one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bc2_sqrt;
};
inline AdamConst adam_const(double lr, double beta1, double beta2, double eps, double bc1, double bc2_sqrt) {
    AdamConst c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1),
                   (float)bc2_sqrt};
    return c;
}
GS_HD float adam_update(float p, float g, float* m, float* v, const AdamConst& c) {
    const float m1 = *m + c.one_minus_beta1 * (g - *m);
    const float v1 = *v * c.beta2 + c.one_minus_beta2 * (g * g);
    *m = m1;
    *v = v1;
    const float denom = sqrtf(v1) / c.bc2_sqrt + c.eps;
    return p + (-c.step_size) * (m1 / denom);
}

// ---------------------------------------------------------------- covariance

// forward.cu:114-148.  q = (r, x, y, z), used as given (not normalised).
GS_HD M3 quat_to_R(V4 q) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    return m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

GS_HD void cov3d_from_scale_rot(V3 s, float mod, V4 q, float* cov6) {
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * s.x;
    S.c[1][1] = mod * s.y;
    S.c[2][2] = mod * s.z;
    const M3 R = quat_to_R(q);
    const M3 M = m3_mul(S, R);
    const M3 Sigma = m3_mul(m3_T(M), M);
    cov6[0] = Sigma.c[0][0]; cov6[1] = Sigma.c[0][1]; cov6[2] = Sigma.c[0][2];
    cov6[3] = Sigma.c[1][1]; cov6[4] = Sigma.c[1][2]; cov6[5] = Sigma.c[2][2];
}

// Backward of cov3d_from_scale_rot (what backward.cu:333-396 computes).  Read as mathematical matrices
// (element (row i, column k) is stored at c[k][i]), M = S R with S = diag(mod * scale) scales ROW i of R by
// s_i, and Sigma = M^T M.  The six stored entries g6 are gradients of the UNIQUE entries, so the full
// symmetric gradient G has the off-diagonals halved, and
//     dL/dM = 2 M G,      dL/ds_i = sum_k dL/dM(i,k) R(i,k),      dL/dR(i,k) = s_i dL/dM(i,k).
// The quaternion gradient is the Frobenius product of dL/dR with the partial derivatives of quat_to_R with
// respect to (r, x, y, z) -- of the quaternion as given: it is not normalised here.
GS_HD float m3_frobenius(const M3& a, const M3& b) {
    float acc = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int w = 0; w < 3; ++w) acc += a.c[c][w] * b.c[c][w];
    return acc;
}

GS_HD void cov3d_from_scale_rot_bwd(V3 sc, float mod, V4 q, const float* g6, V3* dscale, V4* drot) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const M3 R = quat_to_R(q);
    const float s[3] = {mod * sc.x, mod * sc.y, mod * sc.z};
    M3 S = m3_cols(s[0], 0, 0, 0, s[1], 0, 0, 0, s[2]);
    const M3 M = m3_mul(S, R);
    const M3 G = m3_cols(g6[0], 0.5f * g6[1], 0.5f * g6[2],
                         0.5f * g6[1], g6[3], 0.5f * g6[4],
                         0.5f * g6[2], 0.5f * g6[4], g6[5]);
    const M3 dM = m3_mul(m3_scale(2.0f, M), G);   // stored like R: dM.c[k][i] pairs with R.c[k][i]
    float ds[3];
    M3 dR;
    for (int i = 0; i < 3; ++i) {
        ds[i] = R.c[0][i] * dM.c[0][i] + R.c[1][i] * dM.c[1][i] + R.c[2][i] * dM.c[2][i];
        for (int k = 0; k < 3; ++k) dR.c[k][i] = s[i] * dM.c[k][i];
    }
    *dscale = v3(ds[0], ds[1], ds[2]);   // with respect to mod * scale, as the reference returns it (backward.cu:377-380)
    // partial derivatives of quat_to_R, entry by entry in m3_cols order
    const M3 dR_dr = m3_cols(0, -2 * z, 2 * y, 2 * z, 0, -2 * x, -2 * y, 2 * x, 0);
    const M3 dR_dx = m3_cols(0, 2 * y, 2 * z, 2 * y, -4 * x, -2 * r, 2 * z, 2 * r, -4 * x);
    const M3 dR_dy = m3_cols(-4 * y, 2 * x, 2 * r, 2 * x, 0, 2 * z, -2 * r, 2 * z, -4 * y);
    const M3 dR_dz = m3_cols(-4 * z, -2 * r, 2 * x, 2 * r, -4 * z, 2 * y, 2 * x, 2 * y, 0);
    drot->x = m3_frobenius(dR, dR_dr);
    drot->y = m3_frobenius(dR, dR_dx);
    drot->z = m3_frobenius(dR, dR_dy);
    drot->w = m3_frobenius(dR, dR_dz);
}

// The view-dependent pieces of the EWA projection that forward and backward
// both need (forward.cu:74-109, backward.cu:176-204).
struct Ewa {
    V3 t;             // camera-space mean, x/y clamped to the 1.3*tanfov frustum
    float txtz, tytz; // unclamped ratios (for the gradient masks)
    float limx, limy;
    M3 W, T, Vrk;
    float cxx, cxy, cyy;  // cov2D before dilation
};

GS_HD Ewa ewa_project(V3 mean, float fx, float fy, float tanx, float tany, const float* cov6, const float* view) {
    Ewa e;
    e.t = xform_pt_4x3(mean, view);
    e.limx = 1.3f * tanx;
    e.limy = 1.3f * tany;
    e.txtz = e.t.x / e.t.z;
    e.tytz = e.t.y / e.t.z;
    e.t.x = fmin_(e.limx, fmax_(-e.limx, e.txtz)) * e.t.z;
    e.t.y = fmin_(e.limy, fmax_(-e.limy, e.tytz)) * e.t.z;
    const M3 J = m3_cols(fx / e.t.z, 0.0f, -(fx * e.t.x) / (e.t.z * e.t.z),
                         0.0f, fy / e.t.z, -(fy * e.t.y) / (e.t.z * e.t.z),
                         0, 0, 0);
    e.W = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    e.T = m3_mul(e.W, J);
    e.Vrk = m3_cols(cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]);
    const M3 cov = m3_mul(m3_mul(m3_T(e.T), m3_T(e.Vrk)), e.T);
    e.cxx = cov.c[0][0];
    e.cxy = cov.c[0][1];
    e.cyy = cov.c[1][1];
    return e;
}

// ---------------------------------------------------------------- forward preprocess

struct Cam {
    float view[16];
    float proj[16];
    float campos[3];
    int W, H;
    float tanfovx, tanfovy;
    float focal_x, focal_y;   // rasterizer_impl.cu:238-239
    float scale_modifier;
    int grid_x, grid_y;
    int sh_degree;            // active degree D
    int sh_coeffs;            // allocated coefficients per Gaussian M
    int antialiasing;
};

GS_HD void cam_finish(Cam* c) {
    c->focal_y = c->H / (2.0f * c->tanfovy);
    c->focal_x = c->W / (2.0f * c->tanfovx);
    c->grid_x = (c->W + TILE_X - 1) / TILE_X;
    c->grid_y = (c->H + TILE_Y - 1) / TILE_Y;
}

// auxiliary.h:151-189: only the near-plane test is active.
GS_HD bool near_cull(V3 p, const float* view, float* depth) {
    const V3 pv = xform_pt_4x3(p, view);
    *depth = pv.z;
    return pv.z <= 0.2f;
}

struct Splat {
    int radius;        // 0 => not rendered
    uint32_t tiles;
    Rect rect;
    V2 xy;
    float depth;
    V4 conic_opacity;
};

// forward.cu:192-271 and 294-307, everything except covariance construction
// and colour.  cov6 is the world-space covariance (computed or supplied).
// Returns false when the Gaussian is dropped; *out then has radius = tiles = 0.
GS_HD bool project_splat(const Cam& c, V3 p, const float* cov6, float opacity, float depth, Splat* out) {
    out->radius = 0;
    out->tiles = 0;
    const V4 ph = xform_pt_4x4(p, c.proj);
    const float pw = 1.0f / (ph.w + 0.0000001f);
    const V3 pp = v3(ph.x * pw, ph.y * pw, ph.z * pw);

    const Ewa e = ewa_project(p, c.focal_x, c.focal_y, c.tanfovx, c.tanfovy, cov6, c.view);
    float cx = e.cxx, cy = e.cxy, cz = e.cyy;
    const float h_var = 0.3f;
    const float det_cov = cx * cz - cy * cy;
    cx += h_var;
    cz += h_var;
    const float det_h = cx * cz - cy * cy;
    float aa = 1.0f;
    if (c.antialiasing) aa = sqrtf(fmax_(0.000025f, det_cov / det_h));
    const float det = det_h;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    const V3 conic = v3(cz * det_inv, -cy * det_inv, cx * det_inv);

    const float mid = 0.5f * (cx + cz);
    const float lambda1 = mid + sqrtf(fmax_(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmax_(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmax_(lambda1, lambda2)));
    const V2 pix = {ndc2pix(pp.x, c.W), ndc2pix(pp.y, c.H)};
    const int rad_i = f2i_sat(my_radius);
    const Rect rc = tile_rect(pix, rad_i, c.grid_x, c.grid_y);
    const uint32_t nt = (uint32_t)(rc.y1 - rc.y0) * (uint32_t)(rc.x1 - rc.x0);
    if (nt == 0) return false;

    out->radius = rad_i;
    out->tiles = nt;
    out->rect = rc;
    out->xy = pix;
    out->depth = depth;
    out->conic_opacity.x = conic.x;
    out->conic_opacity.y = conic.y;
    out->conic_opacity.z = conic.z;
    out->conic_opacity.w = opacity * aa;
    return true;
}

// ---------------------------------------------------------------- blending

// forward.cu:419-422 / backward.cu:598-602
GS_HD float splat_power(V2 xy, float px, float py, V4 co, float* dx, float* dy) {
    *dx = xy.x - px;
    *dy = xy.y - py;
    return -0.5f * (co.x * *dx * *dx + co.z * *dy * *dy) - co.y * *dx * *dy;
}

#define GS_ALPHA_MAX 0.99f
#define GS_ALPHA_MIN (1.0f / 255.0f)
#define GS_T_MIN 0.0001f

// ---------------------------------------------------------------- backward preprocess

// Partial derivatives of the antialiasing ratio f = (x y - z^2) / ((x + w)(y + w) - z^2) with respect to the
// covariance entries x = c_xx, y = c_yy, z = c_xy, times `scale`: with N the numerator and D the denominator,
// df/dx = (y D - N (y + w)) / D^2 = w (w y + y^2 + z^2) / D^2, symmetrically for y, and
// df/dz = -2 z (D - N) / D^2 = -2 w z (w + x + y) / D^2.
GS_HD V3 aa_ratio_grad(float x, float y, float z, float w, float scale_by) {
    const float D = w * w + w * (x + y) + x * y - z * z;
    const float k = scale_by / (D * D);
    return v3(w * (w * y + y * y + z * z) * k, -2.f * w * z * (w + x + y) * k, w * (w * x + x * x + z * z) * k);
}

// Backward of the EWA projection for one Gaussian (what backward.cu:147-329 computes).
// g_conic = gradients of the inverse 2D covariance K = (conic.x, conic.y; conic.y, conic.w), the off-diagonal
// one PER ENTRY (the blend backward accumulates it with the factor 0.5, backward.cu:682).  dopacity is read
// and, under antialiasing, rewritten.  Returns dL/dmean through the covariance (and the inverse depth) and
// fills dcov6, the gradients of the six stored entries of the 3D covariance.
//
// Derivation.  The 2D covariance is C = (a'Va, a'Vb; a'Vb, b'Vb) + 0.3 I, with V the 3D covariance and a, b the
// first two columns of T = W J (GLM product; a = J00 w0 + J02 w2, b = J11 w1 + J12 w2 with w_k the columns of W
// as stored).  K = C^-1 gives dL/dC = -K G K per entry, G = (gx, gy; gy, gw); the pair parameter c_xy takes twice
// the off-diagonal entry.  The determinant is regularised as the reference does: 1 / (det^2 + 1e-7).
GS_HD V3 cov2d_bwd(const Cam& c, V3 mean, const float* cov6, float opacity, V3 g_conic,
                   float* dopacity, const float* dinvdepth, float* dcov6) {
    const float fx = c.focal_x, fy = c.focal_y;
    const Ewa e = ewa_project(mean, fx, fy, c.tanfovx, c.tanfovy, cov6, c.view);
    const float blur = 0.3f;
    const float A = e.cxx + blur, B = e.cxy, Cc = e.cyy + blur;   // the dilated covariance
    // gradient with respect to (c_xx, c_xy as a pair, c_yy)
    float g_xx = 0.f, g_xy = 0.f, g_yy = 0.f;
    if (c.antialiasing) {
        // opacity' = opacity * sqrt(max(2.5e-5, ratio)): the factor scales dL/dopacity, and its own gradient goes
        // into the covariance.  The reference evaluates the ratio's partials at the DILATED entries
        // (backward.cu:234-249); so does this.
        const float ratio = (e.cxx * e.cyy - B * B) / (A * Cc - B * B);
        const float factor = sqrtf(fmax_(0.000025f, ratio));
        const float g_factor = *dopacity * opacity;
        *dopacity = *dopacity * factor;
        const float g_ratio = ratio <= 0.000025f ? 0.f : g_factor / (2 * factor);
        const V3 d = aa_ratio_grad(A, Cc, B, blur, g_ratio);
        g_xx = d.x;
        g_xy = d.y;
        g_yy = d.z;
    }
    const float det = A * Cc - B * B;
    const float inv2 = 1.0f / ((det * det) + 0.0000001f);
    if (inv2 != 0) {
        // -adj G adj / det^2 with adj = (Cc, -B; -B, A)
        const float gx = g_conic.x, gy = g_conic.y, gw = g_conic.z;
        const float u0 = Cc * gx - B * gy, u1 = Cc * gy - B * gw;    // first row of adj G
        const float v0 = A * gy - B * gx, v1 = A * gw - B * gy;      // second row of adj G
        g_xx += -inv2 * (u0 * Cc - u1 * B);
        g_yy += -inv2 * (v1 * A - v0 * B);
        g_xy += -2.f * inv2 * (u1 * A - u0 * B);
    }
    const V3 a = v3(e.T.c[0][0], e.T.c[0][1], e.T.c[0][2]);
    const V3 b = v3(e.T.c[1][0], e.T.c[1][1], e.T.c[1][2]);
    if (inv2 != 0) {
        // dL/dV_kl = g_xx a_k a_l + g_xy a_k b_l + g_yy b_k b_l; a stored off-diagonal entry stands for both (k,l) and (l,k)
        const float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
        const int slot[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        for (int i = 0; i < 6; ++i) dcov6[i] = 0.f;
        for (int k = 0; k < 3; ++k)
            for (int l = 0; l < 3; ++l)
                dcov6[slot[k][l]] += g_xx * av[k] * av[l] + g_xy * av[k] * bv[l] + g_yy * bv[k] * bv[l];
    } else {
        for (int i = 0; i < 6; ++i) dcov6[i] = 0.f;
    }
    // dL/da = 2 g_xx V a + g_xy V b,  dL/db = 2 g_yy V b + g_xy V a
    const float (*V)[3] = e.Vrk.c;
    const V3 Va = v3(V[0][0] * a.x + V[0][1] * a.y + V[0][2] * a.z, V[1][0] * a.x + V[1][1] * a.y + V[1][2] * a.z,
                     V[2][0] * a.x + V[2][1] * a.y + V[2][2] * a.z);
    const V3 Vb = v3(V[0][0] * b.x + V[0][1] * b.y + V[0][2] * b.z, V[1][0] * b.x + V[1][1] * b.y + V[1][2] * b.z,
                     V[2][0] * b.x + V[2][1] * b.y + V[2][2] * b.z);
    const V3 ga = add(scale(2.f * g_xx, Va), scale(g_xy, Vb));
    const V3 gb = add(scale(2.f * g_yy, Vb), scale(g_xy, Va));
    // a = J00 w0 + J02 w2, b = J11 w1 + J12 w2
    const V3 w0 = v3(e.W.c[0][0], e.W.c[0][1], e.W.c[0][2]);
    const V3 w1 = v3(e.W.c[1][0], e.W.c[1][1], e.W.c[1][2]);
    const V3 w2 = v3(e.W.c[2][0], e.W.c[2][1], e.W.c[2][2]);
    const float gJ00 = dot(w0, ga), gJ02 = dot(w2, ga), gJ11 = dot(w1, gb), gJ12 = dot(w2, gb);
    // J00 = fx / tz, J02 = -fx tx / tz^2, J11 = fy / tz, J12 = -fy ty / tz^2, with tx, ty clamped to the
    // 1.3 x frustum: a clamped coordinate passes no gradient to itself (the reference also drops its
    // dependence on tz, backward.cu:185-186, 313-315; so does this)
    const V3 t = e.t;
    const float inside_x = (e.txtz < -e.limx || e.txtz > e.limx) ? 0.f : 1.f;
    const float inside_y = (e.tytz < -e.limy || e.tytz > e.limy) ? 0.f : 1.f;
    const float iz = 1.f / t.z, iz2 = iz * iz, iz3 = iz2 * iz;
    const float g_tx = inside_x * (-fx * iz2) * gJ02;
    const float g_ty = inside_y * (-fy * iz2) * gJ12;
    float g_tz = -fx * iz2 * gJ00 - fy * iz2 * gJ11 + (2 * fx * t.x) * iz3 * gJ02 + (2 * fy * t.y) * iz3 * gJ12;
    if (dinvdepth) g_tz -= *dinvdepth / (t.z * t.z);   // the rendered inverse depth is 1 / t.z per Gaussian
    return xform_vec_4x3_T(v3(g_tx, g_ty, g_tz), c.view);
}

// backward.cu:429-443: gradient of the 3D mean through the 2D mean.
GS_HD V3 mean2d_bwd(V3 m, const float* proj, float gx, float gy) {
    const V4 mh = xform_pt_4x4(m, proj);
    const float m_w = 1.0f / (mh.w + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    V3 d;
    d.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    d.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    d.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    return d;
}

// ---------------------------------------------------------------- simple-knn helpers
// submodules/simple-knn/simple_knn.cu:46-62

GS_HD uint32_t morton_spread(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

// float -> uint32 with GPU saturation (negative and NaN -> 0).
GS_HD uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

GS_HD uint32_t morton_code(V3 p, V3 mn, V3 mx) {
    const uint32_t x = morton_spread(f2u_sat(((p.x - mn.x) / (mx.x - mn.x)) * ((1 << 10) - 1)));
    const uint32_t y = morton_spread(f2u_sat(((p.y - mn.y) / (mx.y - mn.y)) * ((1 << 10) - 1)));
    const uint32_t z = morton_spread(f2u_sat(((p.z - mn.z) / (mx.z - mn.z)) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}

struct Box { V3 mn, mx; };

// simple_knn.cu:120-130
GS_HD float box_point_dist2(const Box& b, V3 p) {
    V3 d = v3(0, 0, 0);
    if (p.x < b.mn.x || p.x > b.mx.x) d.x = fmin_(fabsf(p.x - b.mn.x), fabsf(p.x - b.mx.x));
    if (p.y < b.mn.y || p.y > b.mx.y) d.y = fmin_(fabsf(p.y - b.mn.y), fabsf(p.y - b.mx.y));
    if (p.z < b.mn.z || p.z > b.mx.z) d.z = fmin_(fabsf(p.z - b.mn.z), fabsf(p.z - b.mx.z));
    return d.x * d.x + d.y * d.y + d.z * d.z;
}

// simple_knn.cu:132-146 with K = 3
// (the squared distance with fused multiply-adds, as the reference's compiler contracts the expression)
GS_HD float knn_dist2(V3 ref, V3 pt) {
    const V3 d = sub(pt, ref);
    return fmaf(d.z, d.z, fmaf(d.y, d.y, d.x * d.x));
}
GS_HD void knn3_update(V3 ref, V3 pt, float* best) {
    float dist = knn_dist2(ref, pt);
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

// ---------------------------------------------------------------- fused-ssim helpers
// submodules/fused-ssim/ssim.cu:9-19

#define GS_SSIM_TAPS 11
GS_HD float ssim_tap(int k) {
    const float g[GS_SSIM_TAPS] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                   0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                   0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                   0.0075987582094967365f, 0.001028380123898387f};
    return g[k];
}

struct SsimOut { float m, dm_dmu1, dm_dsigma1_sq, dm_dsigma12; };

// ssim.cu:262-283 from the five filtered moments.
GS_HD SsimOut ssim_point(float mu1, float ex2, float mu2, float ey2, float exy, float C1, float C2) {
    const float sigma1_sq = ex2 - mu1 * mu1;
    const float sigma2_sq = ey2 - mu2 * mu2;
    const float sigma12 = exy - mu1 * mu2;
    const float mu1_sq = mu1 * mu1;
    const float mu2_sq = mu2 * mu2;
    const float mu1_mu2 = mu1 * mu2;
    const float C = (2.0f * mu1_mu2 + C1);
    const float D = (2.0f * sigma12 + C2);
    const float A = (mu1_sq + mu2_sq + C1);
    const float B = (sigma1_sq + sigma2_sq + C2);
    // The reference divides seven times by products of A and B; here the two reciprocals are formed once
    // (the results agree to a few ulp; the map is compared with the reference's at 1e-5 relative).
    const float rA = 1.0f / A, rB = 1.0f / B;
    const float rAB = rA * rB;
    const float m = (C * D) * rAB;
    SsimOut o;
    o.m = m;
    o.dm_dmu1 = 2.0f * (mu2 * (D - C) * rAB + mu1 * m * (rB - rA));
    o.dm_dsigma1_sq = -m * rB;
    o.dm_dsigma12 = (2.0f * C) * rAB;
    return o;
}

}  // namespace gs
