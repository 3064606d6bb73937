// Fused SSIM map (forward, with the three derivative maps) and its backward.
// Reference: submodules/fused-ssim/ssim.cu:187-366 (11-tap sigma-1.5 separable
// Gaussian, zero padding, taps accumulated in order G_00..G_10).
//
// One 256-thread workgroup produces a 32x32 output tile of one image plane:
// the 42x42 halo of the inputs goes to LDS once, the horizontal pass writes
// all filtered moments of the 42 halo rows back to LDS, the vertical pass
// reads them column-wise (consecutive lanes -> consecutive banks).  Both passes
// compute four neighbouring outputs per thread from a sliding window of 14
// inputs, which cuts the LDS reads per output from 11 to 3.5 (the kernels were
// bound by LDS reads and barriers, not by the 6 / 7 planes of HBM traffic);
// every output still accumulates its taps in the order G_00..G_10.
#include "kernels.h"

namespace gs {

constexpr int SS_T = 32;          // output tile edge
constexpr int SS_H = SS_T + 10;   // halo edge

__device__ __forceinline__ float ssim_ld(const float* img, int y, int x, int H, int W) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

__global__ void __launch_bounds__(256) ssim_fwd_kernel(int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                       float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                       float* __restrict__ dm_dsigma12) {
    __shared__ float s_a[SS_H][SS_H + 1];
    __shared__ float s_b[SS_H][SS_H + 1];
    __shared__ __align__(16) float s_x[5][SS_H][SS_T];  // mu1, E[x^2], mu2, E[y^2], E[xy] after the horizontal pass
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* a = img1 + plane;
    const float* b = img2 + plane;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int tid = threadIdx.x;

    for (int i = tid; i < SS_H * SS_H; i += 256) {
        const int ly = i / SS_H, lx = i - ly * SS_H;
        s_a[ly][lx] = ssim_ld(a, y0 + ly - 5, x0 + lx - 5, H, W);
        s_b[ly][lx] = ssim_ld(b, y0 + ly - 5, x0 + lx - 5, H, W);
    }
    __syncthreads();
    for (int i = tid; i < SS_H * (SS_T / 4); i += 256) {
        const int ly = i / (SS_T / 4), lx = (i - ly * (SS_T / 4)) * 4;
        float v[5][4];
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < GS_SSIM_TAPS + 3; ++j) {
            const float pa = s_a[ly][lx + j], pb = s_b[ly][lx + j];
            const float aa = pa * pa, bb = pb * pb, ab = pa * pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = j - r;  // output lx + r sees this input as its tap k
                if (k >= 0 && k < GS_SSIM_TAPS) {
                    const float g = ssim_tap(k);
                    v[0][r] += g * pa;
                    v[1][r] += g * aa;
                    v[2][r] += g * pb;
                    v[3][r] += g * bb;
                    v[4][r] += g * ab;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 5; ++c)
            *reinterpret_cast<float4*>(&s_x[c][ly][lx]) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    }
    __syncthreads();
    const int lx = tid & 31, ly0 = (tid >> 5) * 4;
    float v[5][4];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[c][r] = 0.f;
#pragma unroll
    for (int j = 0; j < GS_SSIM_TAPS + 3; ++j) {
        float in[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) in[c] = s_x[c][ly0 + j][lx];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = j - r;
            if (k >= 0 && k < GS_SSIM_TAPS) {
                const float g = ssim_tap(k);
#pragma unroll
                for (int c = 0; c < 5; ++c) v[c][r] += g * in[c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int px = x0 + lx, py = y0 + ly0 + r;
        if (px < W && py < H) {
            const SsimOut o = ssim_point(v[0][r], v[1][r], v[2][r], v[3][r], v[4][r], C1, C2);
            const size_t gi = plane + (size_t)py * W + px;
            ssim_map[gi] = o.m;
            if (dm_dmu1) {
                dm_dmu1[gi] = o.dm_dmu1;
                dm_dsigma1_sq[gi] = o.dm_dsigma1_sq;
                dm_dsigma12[gi] = o.dm_dsigma12;
            }
        }
    }
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2,
                                                       const float* __restrict__ dL_dmap, float* __restrict__ dL_dimg1,
                                                       const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsigma1_sq,
                                                       const float* __restrict__ dm_dsigma12) {
    __shared__ float s_p[3][SS_H][SS_H + 1];  // dL * (dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
    __shared__ __align__(16) float s_x[3][SS_H][SS_T];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int tid = threadIdx.x;
    for (int i = tid; i < SS_H * SS_H; i += 256) {
        const int ly = i / SS_H, lx = i - ly * SS_H;
        const int y = y0 + ly - 5, x = x0 + lx - 5;
        const float dl = ssim_ld(dL_dmap + plane, y, x, H, W);
        s_p[0][ly][lx] = ssim_ld(dm_dmu1 + plane, y, x, H, W) * dl;
        s_p[1][ly][lx] = ssim_ld(dm_dsigma1_sq + plane, y, x, H, W) * dl;
        s_p[2][ly][lx] = ssim_ld(dm_dsigma12 + plane, y, x, H, W) * dl;
    }
    __syncthreads();
    for (int i = tid; i < SS_H * (SS_T / 4); i += 256) {
        const int ly = i / (SS_T / 4), lx = (i - ly * (SS_T / 4)) * 4;
        float v[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < GS_SSIM_TAPS + 3; ++j) {
            float in[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) in[c] = s_p[c][ly][lx + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = j - r;
                if (k >= 0 && k < GS_SSIM_TAPS) {
                    const float g = ssim_tap(k);
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c][r] += g * in[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4*>(&s_x[c][ly][lx]) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    }
    __syncthreads();
    const int lx = tid & 31, ly0 = (tid >> 5) * 4;
    float v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[c][r] = 0.f;
#pragma unroll
    for (int j = 0; j < GS_SSIM_TAPS + 3; ++j) {
        float in[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) in[c] = s_x[c][ly0 + j][lx];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = j - r;
            if (k >= 0 && k < GS_SSIM_TAPS) {
                const float g = ssim_tap(k);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c][r] += g * in[c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int px = x0 + lx, py = y0 + ly0 + r;
        if (px < W && py < H) {
            const size_t gi = plane + (size_t)py * W + px;
            float d = 0.0f;
            d += v[0][r];
            d += img1[gi] * 2.0f * v[1][r];
            d += img2[gi] * v[2][r];
            dL_dimg1[gi] = d;
        }
    }
}

hipError_t launch_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                           float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B * CH);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, st, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12);
    return hipGetLastError();
}

hipError_t launch_ssim_bwd(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                           float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                           hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B * CH);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, st, H, W, img1, img2, dL_dmap, dL_dimg1, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12);
    return hipGetLastError();
}

}  // namespace gs
