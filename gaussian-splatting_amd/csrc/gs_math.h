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

// auxiliary.h:119-129
GS_HD V3 dnormvdv(V3 v, V3 dv) {
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    V3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
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

// acc + k * v, as GLM evaluates "acc + (scalar) * vec3".
GS_HD V3 axpy(V3 acc, float k, V3 v) { return v3(acc.x + k * v.x, acc.y + k * v.y, acc.z + k * v.z); }
GS_HD V3 axmy(V3 acc, float k, V3 v) { return v3(acc.x - k * v.x, acc.y - k * v.y, acc.z - k * v.z); }

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

// backward.cu:23-142.  g = dL/dRGB with clamped channels already zeroed.
// OUT is any callable (k, V3) that stores dL/dsh[k].  Returns the part of
// dL/dmean that flows through the view direction.
template <class SH, class OUT>
GS_HD V3 sh_to_rgb_bwd(int deg, V3 pos, V3 campos, V3 g, SH sh, OUT out) {
    const V3 dir_orig = sub(pos, campos);
    const float len = sqrtf(dot(dir_orig, dir_orig));
    const V3 dir = v3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
    const float x = dir.x, y = dir.y, z = dir.z;

    V3 dx = v3(0, 0, 0), dy = v3(0, 0, 0), dz = v3(0, 0, 0);
    out(0, scale(GS_SH_C0, g));
    if (deg > 0) {
        out(1, scale(-GS_SH_C1 * y, g));
        out(2, scale(GS_SH_C1 * z, g));
        out(3, scale(-GS_SH_C1 * x, g));
        dx = scale(-GS_SH_C1, sh(3));
        dy = scale(-GS_SH_C1, sh(1));
        dz = scale(GS_SH_C1, sh(2));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            out(4, scale(GS_SH_C2_0 * xy, g));
            out(5, scale(GS_SH_C2_1 * yz, g));
            out(6, scale(GS_SH_C2_2 * (2.f * zz - xx - yy), g));
            out(7, scale(GS_SH_C2_3 * xz, g));
            out(8, scale(GS_SH_C2_4 * (xx - yy), g));
            {
                V3 t = scale(GS_SH_C2_0 * y, sh(4));
                t = axpy(t, GS_SH_C2_2 * 2.f * -x, sh(6));
                t = axpy(t, GS_SH_C2_3 * z, sh(7));
                t = axpy(t, GS_SH_C2_4 * 2.f * x, sh(8));
                dx = add(dx, t);
                t = scale(GS_SH_C2_0 * x, sh(4));
                t = axpy(t, GS_SH_C2_1 * z, sh(5));
                t = axpy(t, GS_SH_C2_2 * 2.f * -y, sh(6));
                t = axpy(t, GS_SH_C2_4 * 2.f * -y, sh(8));
                dy = add(dy, t);
                t = scale(GS_SH_C2_1 * y, sh(5));
                t = axpy(t, GS_SH_C2_2 * 2.f * 2.f * z, sh(6));
                t = axpy(t, GS_SH_C2_3 * x, sh(7));
                dz = add(dz, t);
            }
            if (deg > 2) {
                out(9, scale(GS_SH_C3_0 * y * (3.f * xx - yy), g));
                out(10, scale(GS_SH_C3_1 * xy * z, g));
                out(11, scale(GS_SH_C3_2 * y * (4.f * zz - xx - yy), g));
                out(12, scale(GS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy), g));
                out(13, scale(GS_SH_C3_4 * x * (4.f * zz - xx - yy), g));
                out(14, scale(GS_SH_C3_5 * z * (xx - yy), g));
                out(15, scale(GS_SH_C3_6 * x * (xx - 3.f * yy), g));
                // "C * sh[k] * a * b" is evaluated left to right: ((C*sh)*a)*b.
                auto s2 = [](float c, V3 s, float a) { return v3(c * s.x * a, c * s.y * a, c * s.z * a); };
                auto s3 = [](float c, V3 s, float a, float b) {
                    return v3(c * s.x * a * b, c * s.y * a * b, c * s.z * a * b);
                };
                V3 t = s3(GS_SH_C3_0, sh(9), 3.f * 2.f, xy);
                t = add(t, s2(GS_SH_C3_1, sh(10), yz));
                t = add(t, s3(GS_SH_C3_2, sh(11), -2.f, xy));
                t = add(t, s3(GS_SH_C3_3, sh(12), -3.f * 2.f, xz));
                t = add(t, s2(GS_SH_C3_4, sh(13), (-3.f * xx + 4.f * zz - yy)));
                t = add(t, s3(GS_SH_C3_5, sh(14), 2.f, xz));
                t = add(t, s3(GS_SH_C3_6, sh(15), 3.f, (xx - yy)));
                dx = add(dx, t);
                t = s3(GS_SH_C3_0, sh(9), 3.f, (xx - yy));
                t = add(t, s2(GS_SH_C3_1, sh(10), xz));
                t = add(t, s2(GS_SH_C3_2, sh(11), (-3.f * yy + 4.f * zz - xx)));
                t = add(t, s3(GS_SH_C3_3, sh(12), -3.f * 2.f, yz));
                t = add(t, s3(GS_SH_C3_4, sh(13), -2.f, xy));
                t = add(t, s3(GS_SH_C3_5, sh(14), -2.f, yz));
                t = add(t, s3(GS_SH_C3_6, sh(15), -3.f * 2.f, xy));
                dy = add(dy, t);
                t = s2(GS_SH_C3_1, sh(10), xy);
                t = add(t, s3(GS_SH_C3_2, sh(11), 4.f * 2.f, yz));
                t = add(t, s3(GS_SH_C3_3, sh(12), 3.f, (2.f * zz - xx - yy)));
                t = add(t, s3(GS_SH_C3_4, sh(13), 4.f * 2.f, xz));
                t = add(t, s2(GS_SH_C3_5, sh(14), (xx - yy)));
                dz = add(dz, t);
            }
        }
    }
    const V3 ddir = v3(dot(dx, g), dot(dy, g), dot(dz, g));
    return dnormvdv(dir_orig, ddir);
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

// backward.cu:333-396
GS_HD void cov3d_from_scale_rot_bwd(V3 sc, float mod, V4 q, const float* g6, V3* dscale, V4* drot) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const M3 R = quat_to_R(q);
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    const V3 s = v3(mod * sc.x, mod * sc.y, mod * sc.z);
    S.c[0][0] = s.x; S.c[1][1] = s.y; S.c[2][2] = s.z;
    const M3 M = m3_mul(S, R);
    const M3 dSigma = m3_cols(g6[0], 0.5f * g6[1], 0.5f * g6[2],
                              0.5f * g6[1], g6[3], 0.5f * g6[4],
                              0.5f * g6[2], 0.5f * g6[4], g6[5]);
    const M3 dM = m3_mul(m3_scale(2.0f, M), dSigma);
    const M3 Rt = m3_T(R);
    M3 dMt = m3_T(dM);
    dscale->x = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
    dscale->y = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
    dscale->z = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
    for (int k = 0; k < 3; ++k) { dMt.c[0][k] *= s.x; dMt.c[1][k] *= s.y; dMt.c[2][k] *= s.z; }
    const float (*d)[3] = dMt.c;
    drot->x = 2 * z * (d[0][1] - d[1][0]) + 2 * y * (d[2][0] - d[0][2]) + 2 * x * (d[1][2] - d[2][1]);
    drot->y = 2 * y * (d[1][0] + d[0][1]) + 2 * z * (d[2][0] + d[0][2]) + 2 * r * (d[1][2] - d[2][1]) - 4 * x * (d[2][2] + d[1][1]);
    drot->z = 2 * x * (d[1][0] + d[0][1]) + 2 * r * (d[2][0] - d[0][2]) + 2 * z * (d[1][2] + d[2][1]) - 4 * y * (d[2][2] + d[0][0]);
    drot->w = 2 * r * (d[0][1] - d[1][0]) + 2 * x * (d[2][0] + d[0][2]) + 2 * y * (d[1][2] + d[2][1]) - 4 * z * (d[1][1] + d[0][0]);
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

// backward.cu:147-329, one Gaussian.  dconic = (dL/dconic.x, .y, .w).
// dopacity is read and, under antialiasing, rewritten.  Returns dL/dmean (the
// covariance path) and fills dcov6.
GS_HD V3 cov2d_bwd(const Cam& c, V3 mean, const float* cov6, float opacity, V3 dconic,
                   float* dopacity, const float* dinvdepth, float* dcov6) {
    const float h_x = c.focal_x, h_y = c.focal_y;
    const Ewa e = ewa_project(mean, h_x, h_y, c.tanfovx, c.tanfovy, cov6, c.view);
    const V3 t = e.t;
    const float x_grad_mul = (e.txtz < -e.limx || e.txtz > e.limx) ? 0.f : 1.f;
    const float y_grad_mul = (e.tytz < -e.limy || e.tytz > e.limy) ? 0.f : 1.f;
    const float (*T)[3] = e.T.c;
    const float (*W)[3] = e.W.c;
    const float (*Vrk)[3] = e.Vrk.c;

    float c_xx = e.cxx, c_xy = e.cxy, c_yy = e.cyy;
    const float h_var = 0.3f;
    float d_inside_root = 0.f;
    if (c.antialiasing) {
        const float det_cov = c_xx * c_yy - c_xy * c_xy;
        c_xx += h_var;
        c_yy += h_var;
        const float det_h = c_xx * c_yy - c_xy * c_xy;
        const float hcs = sqrtf(fmax_(0.000025f, det_cov / det_h));
        const float dop = *dopacity;
        const float d_hcs = dop * opacity;
        *dopacity = dop * hcs;
        d_inside_root = (det_cov / det_h) <= 0.000025f ? 0.f : d_hcs / (2 * hcs);
    } else {
        c_xx += h_var;
        c_yy += h_var;
    }

    float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
    if (c.antialiasing) {
        const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
        const float q = w * w + w * (x + y) + x * y - z * z;
        const float denom_f = d_inside_root / (q * q);
        dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
        dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
        dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
    }

    const float denom = c_xx * c_yy - c_xy * c_xy;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_dc_xx += denom2inv * (-c_yy * c_yy * dconic.x + 2 * c_xy * c_yy * dconic.y + (denom - c_xx * c_yy) * dconic.z);
        dL_dc_yy += denom2inv * (-c_xx * c_xx * dconic.z + 2 * c_xx * c_xy * dconic.y + (denom - c_xx * c_yy) * dconic.x);
        dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dconic.x - (denom + 2 * c_xy * c_xy) * dconic.y + c_xx * c_xy * dconic.z);

        dcov6[0] = (T[0][0] * T[0][0] * dL_dc_xx + T[0][0] * T[1][0] * dL_dc_xy + T[1][0] * T[1][0] * dL_dc_yy);
        dcov6[3] = (T[0][1] * T[0][1] * dL_dc_xx + T[0][1] * T[1][1] * dL_dc_xy + T[1][1] * T[1][1] * dL_dc_yy);
        dcov6[5] = (T[0][2] * T[0][2] * dL_dc_xx + T[0][2] * T[1][2] * dL_dc_xy + T[1][2] * T[1][2] * dL_dc_yy);
        dcov6[1] = 2 * T[0][0] * T[0][1] * dL_dc_xx + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][1] * dL_dc_yy;
        dcov6[2] = 2 * T[0][0] * T[0][2] * dL_dc_xx + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][2] * dL_dc_yy;
        dcov6[4] = 2 * T[0][2] * T[0][1] * dL_dc_xx + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_dc_xy + 2 * T[1][1] * T[1][2] * dL_dc_yy;
    } else {
        for (int i = 0; i < 6; i++) dcov6[i] = 0;
    }

    const float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_dc_xx +
                          (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc_xy;
    const float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_dc_xx +
                          (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc_xy;
    const float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_dc_xx +
                          (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc_xy;
    const float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc_yy +
                          (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_dc_xy;
    const float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc_yy +
                          (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_dc_xy;
    const float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc_yy +
                          (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_dc_xy;

    const float dL_dJ00 = W[0][0] * dL_dT00 + W[0][1] * dL_dT01 + W[0][2] * dL_dT02;
    const float dL_dJ02 = W[2][0] * dL_dT00 + W[2][1] * dL_dT01 + W[2][2] * dL_dT02;
    const float dL_dJ11 = W[1][0] * dL_dT10 + W[1][1] * dL_dT11 + W[1][2] * dL_dT12;
    const float dL_dJ12 = W[2][0] * dL_dT10 + W[2][1] * dL_dT11 + W[2][2] * dL_dT12;

    const float tz = 1.f / t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;

    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    if (dinvdepth) dL_dtz -= *dinvdepth / (t.z * t.z);

    return xform_vec_4x3_T(v3(dL_dtx, dL_dty, dL_dtz), c.view);
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
GS_HD void knn3_update(V3 ref, V3 pt, float* best) {
    const V3 d = sub(pt, ref);
    float dist = d.x * d.x + d.y * d.y + d.z * d.z;
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
    SsimOut o;
    o.m = (C * D) / (A * B);
    o.dm_dmu1 = ((mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * C) / (A * B) - (mu1 * 2.0f * C * D) / (A * A * B) +
                 (mu1 * 2.0f * C * D) / (A * B * B));
    o.dm_dsigma1_sq = ((-C * D) / (A * B * B));
    o.dm_dsigma12 = ((2 * C) / (A * B));
    return o;
}

}  // namespace gs
