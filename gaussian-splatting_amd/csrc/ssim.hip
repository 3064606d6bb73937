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
// every output still accumulates its taps in the order G_00..G_10, one fused multiply-add per tap (the
// library is built without floating-point contraction, so the fusion is written out; the reference's
// taps are contracted to FMAs by its compiler as well).
#include "kernels.h"

namespace gs {

constexpr int SS_T = 32;          // output tile edge
constexpr int SS_H = SS_T + 10;   // halo edge

__device__ __forceinline__ float ssim_ld(const float* img, int y, int x, int H, int W) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

// The halo loads of a workgroup, all in flight at once: each thread fetches its SS_LD elements of the
// 42 x 42 halo (element tid + 256 k) with unconditional loads from clamped coordinates and zeroes what lies
// outside the image afterwards (zero padding, ssim.cu:36-42).  Written as a loop over conditional loads, the
// compiler waits for every round's loads before it issues the next: seven memory round trips in a row per
// workgroup, which is what these kernels spent most of their time on.
constexpr int SS_LD = (SS_H * SS_H + 255) / 256;
template <int N>
__device__ __forceinline__ void ssim_halo_load(const float* const (&img)[N], int y0, int x0, int H, int W, int tid,
                                               float (&v)[N][SS_LD]) {
#pragma unroll
    for (int k = 0; k < SS_LD; ++k) {
        const int i = min(tid + 256 * k, SS_H * SS_H - 1);
        const int ly = i / SS_H, lx = i - ly * SS_H;
        const int y = y0 + ly - 5, x = x0 + lx - 5;
        const size_t at = (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
#pragma unroll
        for (int c = 0; c < N; ++c) v[c][k] = img[c][at];
    }
#pragma unroll
    for (int k = 0; k < SS_LD; ++k) {
        const int i = min(tid + 256 * k, SS_H * SS_H - 1);
        const int ly = i / SS_H, lx = i - ly * SS_H;
        const int y = y0 + ly - 5, x = x0 + lx - 5;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
#pragma unroll
        for (int c = 0; c < N; ++c) v[c][k] = in ? v[c][k] : 0.0f;
    }
}

// LOSS: the training loss of train.py:222-230 around the map.  img1 is the unclamped render: it is
// clamped to [0, 1] on the way into LDS (gaussian_renderer/__init__.py:179); no map is stored, instead
// the workgroup's sums of |img1 - img2| and of the map go to partial[2 * workgroup] (+1), in a fixed
// order, for the finishing kernel.
__device__ __forceinline__ float clamp01(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

// Horizontal pass: one work item = SS_HG neighbouring outputs of one halo row from a sliding window of
// SS_HG + 10 inputs; 42 rows x 4 items = 168 items, one round of the 256 threads.  The results wait in
// registers across a barrier and then go to LDS OVER the halo they were computed from: the filtered rows
// and the halo are never live together, so the workgroup needs 26.9 KB instead of 41.3 (forward) and five
// or six workgroups fit a CU instead of three.
constexpr int SS_HG = 8;
constexpr int SS_HITEMS = SS_H * (SS_T / SS_HG);
static_assert(SS_HITEMS <= 256, "the horizontal pass is one round of the workgroup");

template <bool LOSS>
__global__ void __launch_bounds__(256) ssim_fwd_kernel(int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                       float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                       float* __restrict__ dm_dsigma12, float* __restrict__ partial) {
    constexpr int HALO_F = 2 * SS_H * (SS_H + 1), ROWS_F = 5 * SS_H * SS_T;
    __shared__ __align__(16) float s_mem[HALO_F > ROWS_F ? HALO_F : ROWS_F];
    float (*s_a)[SS_H + 1] = reinterpret_cast<float (*)[SS_H + 1]>(s_mem);
    float (*s_b)[SS_H + 1] = reinterpret_cast<float (*)[SS_H + 1]>(s_mem + SS_H * (SS_H + 1));
    // mu1, E[x^2], mu2, E[y^2], E[xy] after the horizontal pass (same memory as the halo)
    float (*s_x)[SS_H][SS_T] = reinterpret_cast<float (*)[SS_H][SS_T]>(s_mem);
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* a = img1 + plane;
    const float* b = img2 + plane;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int tid = threadIdx.x;

    float sum_l1 = 0.f, sum_m = 0.f;
    {
        const float* const src[2] = {a, b};
        float v[2][SS_LD];
        ssim_halo_load<2>(src, y0, x0, H, W, tid, v);
#pragma unroll
        for (int k = 0; k < SS_LD; ++k) {
            const int i = tid + 256 * k;
            if (i < SS_H * SS_H) {
                const int ly = i / SS_H, lx = i - ly * SS_H;
                const float pa = LOSS ? clamp01(v[0][k]) : v[0][k];
                s_a[ly][lx] = pa;
                s_b[ly][lx] = v[1][k];
                // the L1 part of the loss, over the pixels of the tile proper that lie inside the image
                if (LOSS && ly >= 5 && ly < 5 + SS_T && lx >= 5 && lx < 5 + SS_T && y0 + ly - 5 < H && x0 + lx - 5 < W)
                    sum_l1 += fabsf(pa - v[1][k]);
            }
        }
    }
    __syncthreads();
    float hv[5][SS_HG];
    const int hy = tid / (SS_T / SS_HG), hx = (tid - hy * (SS_T / SS_HG)) * SS_HG;
    if (tid < SS_HITEMS) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int r = 0; r < SS_HG; ++r) hv[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < GS_SSIM_TAPS + SS_HG - 1; ++j) {
            const float pa = s_a[hy][hx + j], pb = s_b[hy][hx + j];
            const float aa = pa * pa, bb = pb * pb, ab = pa * pb;
#pragma unroll
            for (int r = 0; r < SS_HG; ++r) {
                const int k = j - r;  // output hx + r sees this input as its tap k
                if (k >= 0 && k < GS_SSIM_TAPS) {
                    const float g = ssim_tap(k);
                    hv[0][r] = fmaf(g, pa, hv[0][r]);
                    hv[1][r] = fmaf(g, aa, hv[1][r]);
                    hv[2][r] = fmaf(g, pb, hv[2][r]);
                    hv[3][r] = fmaf(g, bb, hv[3][r]);
                    hv[4][r] = fmaf(g, ab, hv[4][r]);
                }
            }
        }
    }
    __syncthreads();   // every read of the halo is done: its memory takes the filtered rows
    if (tid < SS_HITEMS) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int q = 0; q < SS_HG / 4; ++q)
                *reinterpret_cast<float4*>(&s_x[c][hy][hx + 4 * q]) =
                    make_float4(hv[c][4 * q], hv[c][4 * q + 1], hv[c][4 * q + 2], hv[c][4 * q + 3]);
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
                for (int c = 0; c < 5; ++c) v[c][r] = fmaf(g, in[c], v[c][r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int px = x0 + lx, py = y0 + ly0 + r;
        if (px < W && py < H) {
            const SsimOut o = ssim_point(v[0][r], v[1][r], v[2][r], v[3][r], v[4][r], C1, C2);
            const size_t gi = plane + (size_t)py * W + px;
            if (LOSS) sum_m += o.m;
            else ssim_map[gi] = o.m;
            if (dm_dmu1) {
                dm_dmu1[gi] = o.dm_dmu1;
                dm_dsigma1_sq[gi] = o.dm_dsigma1_sq;
                dm_dsigma12[gi] = o.dm_dsigma12;
            }
        }
    }
    if (LOSS) {
        __shared__ float s_red[2][4];
        const float w1 = wave_sum(sum_l1), wm = wave_sum(sum_m);
        if ((tid & 63) == 0) {
            s_red[0][tid >> 6] = w1;
            s_red[1][tid >> 6] = wm;
        }
        __syncthreads();
        if (tid == 0) {
            const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partial[2 * wg] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
            partial[2 * wg + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        }
    }
}

// loss[0] = (1 - lambda) * mean|img1 - img2| + lambda * (1 - mean(ssim map)), loss[1] = the L1 mean,
// loss[2] = the ssim mean; one workgroup sums the partial sums in a fixed order (double accumulators).
constexpr int LF_THREADS = 1024;
__global__ void __launch_bounds__(LF_THREADS) loss_finish_kernel(const float* __restrict__ partial, int n,
                                                                 double inv_count, float lambda,
                                                                 float* __restrict__ loss) {
    __shared__ double s_r[2][LF_THREADS];
    double a = 0.0, b = 0.0;
    // two pairs per sixteen-byte load where the array's address allows it
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    const int n2 = ((uintptr_t)partial & 15) == 0 ? n >> 1 : 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n2; i += LF_THREADS) {
        const float4 v = p4[i];
        a += (double)v.x + (double)v.z;
        b += (double)v.y + (double)v.w;
    }
    for (int i = 2 * n2 + threadIdx.x; i < n; i += LF_THREADS) {
        a += (double)partial[2 * i];
        b += (double)partial[2 * i + 1];
    }
    s_r[0][threadIdx.x] = a;
    s_r[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = LF_THREADS / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_r[0][threadIdx.x] += s_r[0][threadIdx.x + off];
            s_r[1][threadIdx.x] += s_r[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(s_r[0][0] * inv_count), m = (float)(s_r[1][0] * inv_count);
        loss[0] = (1.0f - lambda) * l1 + lambda * (1.0f - m);
        loss[1] = l1;
        loss[2] = m;
    }
}

// LOSS: backward of the loss above with respect to the unclamped img1.  dL/dmap is the same number
// everywhere, -lambda / count times the incoming gradient of the loss (a device scalar: no host
// block to read it), the L1 term adds (1 - lambda) / count * sign(clamp(img1) - img2), and the clamp
// passes the gradient where 0 <= img1 <= 1.
template <bool LOSS>
__global__ void __launch_bounds__(256) ssim_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2,
                                                       const float* __restrict__ dL_dmap, float* __restrict__ dL_dimg1,
                                                       const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsigma1_sq,
                                                       const float* __restrict__ dm_dsigma12, float lambda,
                                                       float inv_count) {
    constexpr int HALO_F = 3 * SS_H * (SS_H + 1), ROWS_F = 3 * SS_H * SS_T;
    __shared__ __align__(16) float s_mem[HALO_F > ROWS_F ? HALO_F : ROWS_F];
    // dL * (dm_dmu1, dm_dsigma1_sq, dm_dsigma12), and over it, later, the rows after the horizontal pass
    float (*s_p)[SS_H][SS_H + 1] = reinterpret_cast<float (*)[SS_H][SS_H + 1]>(s_mem);
    float (*s_x)[SS_H][SS_T] = reinterpret_cast<float (*)[SS_H][SS_T]>(s_mem);
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int tid = threadIdx.x;
    {
        // LOSS: dL/dmap is one number; the fourth array then is a stand-in that is loaded and not used
        const float* const src[4] = {dm_dmu1 + plane, dm_dsigma1_sq + plane, dm_dsigma12 + plane,
                                     LOSS ? dm_dmu1 + plane : dL_dmap + plane};
        float v[4][SS_LD];
        ssim_halo_load<4>(src, y0, x0, H, W, tid, v);
        const float dl_all = LOSS ? (-lambda * inv_count) * dL_dmap[0] : 0.0f;
#pragma unroll
        for (int k = 0; k < SS_LD; ++k) {
            const int i = tid + 256 * k;
            if (i < SS_H * SS_H) {
                const int ly = i / SS_H, lx = i - ly * SS_H;
                const float dl = LOSS ? dl_all : v[3][k];
                s_p[0][ly][lx] = v[0][k] * dl;
                s_p[1][ly][lx] = v[1][k] * dl;
                s_p[2][ly][lx] = v[2][k] * dl;
            }
        }
    }
    __syncthreads();
    float hv[3][SS_HG];
    const int hy = tid / (SS_T / SS_HG), hx = (tid - hy * (SS_T / SS_HG)) * SS_HG;
    if (tid < SS_HITEMS) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < SS_HG; ++r) hv[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < GS_SSIM_TAPS + SS_HG - 1; ++j) {
            float in[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) in[c] = s_p[c][hy][hx + j];
#pragma unroll
            for (int r = 0; r < SS_HG; ++r) {
                const int k = j - r;
                if (k >= 0 && k < GS_SSIM_TAPS) {
                    const float g = ssim_tap(k);
#pragma unroll
                    for (int c = 0; c < 3; ++c) hv[c][r] = fmaf(g, in[c], hv[c][r]);
                }
            }
        }
    }
    __syncthreads();   // every read of the halo is done: its memory takes the filtered rows
    if (tid < SS_HITEMS) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < SS_HG / 4; ++q)
                *reinterpret_cast<float4*>(&s_x[c][hy][hx + 4 * q]) =
                    make_float4(hv[c][4 * q], hv[c][4 * q + 1], hv[c][4 * q + 2], hv[c][4 * q + 3]);
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
                for (int c = 0; c < 3; ++c) v[c][r] = fmaf(g, in[c], v[c][r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int px = x0 + lx, py = y0 + ly0 + r;
        if (px < W && py < H) {
            const size_t gi = plane + (size_t)py * W + px;
            const float raw = img1[gi], pb = img2[gi];
            const float pa = LOSS ? clamp01(raw) : raw;
            float d = 0.0f;
            d += v[0][r];
            d += pa * 2.0f * v[1][r];
            d += pb * v[2][r];
            if (LOSS) {
                const float df = pa - pb;
                d += ((1.0f - lambda) * inv_count) * dL_dmap[0] * (df > 0.f ? 1.0f : (df < 0.f ? -1.0f : 0.0f));
                if (!(raw >= 0.0f && raw <= 1.0f)) d = 0.0f;
            }
            dL_dimg1[gi] = d;
        }
    }
}

hipError_t launch_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                           float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B * CH);
    hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, st, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, (float*)nullptr);
    return hipGetLastError();
}

hipError_t launch_ssim_bwd(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                           float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                           hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B * CH);
    hipLaunchKernelGGL(ssim_bwd_kernel<false>, grid, dim3(256), 0, st, H, W, img1, img2, dL_dmap, dL_dimg1, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, 0.0f, 0.0f);
    return hipGetLastError();
}

size_t loss_partial_floats(int planes, int H, int W) {
    return 2 * (size_t)((W + SS_T - 1) / SS_T) * ((H + SS_T - 1) / SS_T) * planes;
}

hipError_t launch_loss_fwd(int planes, int H, int W, float C1, float C2, float lambda, const float* img,
                           const float* gt, float* loss, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                           float* partial, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, planes);
    hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(256), 0, st, H, W, C1, C2, img, gt, (float*)nullptr, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, partial);
    const int n = (int)(loss_partial_floats(planes, H, W) / 2);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LF_THREADS), 0, st, partial, n, 1.0 / ((double)planes * H * W), lambda,
                       loss);
    return hipGetLastError();
}

hipError_t launch_loss_bwd(int planes, int H, int W, float lambda, const float* grad_loss, const float* img,
                           const float* gt, float* dL_dimg, const float* dm_dmu1, const float* dm_dsigma1_sq,
                           const float* dm_dsigma12, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, planes);
    hipLaunchKernelGGL(ssim_bwd_kernel<true>, grid, dim3(256), 0, st, H, W, img, gt, grad_loss, dL_dimg, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, lambda, (float)(1.0 / ((double)planes * H * W)));
    return hipGetLastError();
}

// ------------------------------------------------------------------ plain L1 loss
// mean|a - b| (utils/loss_utils.py:23-24) with b a tensor or one number, in two kernels, and its
// backward in one: what train.py:249-251 adds for the rendered feature channels costs seven
// elementwise PyTorch kernels otherwise.  A fixed number of workgroups, each summing a strided
// share in a fixed order; the second kernel sums their partial sums in double.
constexpr int L1_MAX_BLOCKS = 1024;

size_t l1_partial_floats(int64_t n) {
    const int64_t nb = (n + 1023) / 1024;
    return (size_t)(nb < 1 ? 1 : (nb > L1_MAX_BLOCKS ? L1_MAX_BLOCKS : nb));
}

__global__ void __launch_bounds__(256) l1_partial_kernel(int64_t n, const float* __restrict__ a,
                                                         const float* __restrict__ b, float b_scalar,
                                                         float* __restrict__ partial) {
    __shared__ float s_w[4];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // sixteen-byte loads, four of them in flight per lane, when the rows allow it (one load per
    // iteration leaves the kernel waiting for memory: 1.8 TB/s on 33 MB)
    const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        auto term = [&](int64_t k) {
            const float4 x = a4[k];
            const float4 y = b ? b4[k] : make_float4(b_scalar, b_scalar, b_scalar, b_scalar);
            return (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
        };
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float t0 = term(i), t1 = term(i + stride), t2 = term(i + 2 * stride), t3 = term(i + 3 * stride);
            p0 += t0; p1 += t1; p2 += t2; p3 += t3;
        }
        for (; i < n4; i += stride) p0 += term(i);
        acc = (p0 + p1) + (p2 + p3);
        i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;   // the last n % 4 elements
    }
    for (; i < n; i += stride) acc += fabsf(a[i] - (b ? b[i] : b_scalar));
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(256) l1_finish_kernel(const float* __restrict__ partial, int nb, double inv_count,
                                                        float* __restrict__ loss) {
    __shared__ double s_r[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) a += (double)partial[i];
    s_r[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) s_r[threadIdx.x] += s_r[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(s_r[0] * inv_count);
}

// dL/da = grad / n * sign(a - b); sign(0) = 0 as torch.sign.  grad is a device scalar.
__global__ void __launch_bounds__(256) l1_bwd_kernel(int64_t n, const float* __restrict__ grad_loss,
                                                     const float* __restrict__ a, const float* __restrict__ b,
                                                     float b_scalar, float inv_count, float* __restrict__ dL_da) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad_loss[0] * inv_count;
    const float d = a[i] - (b ? b[i] : b_scalar);
    dL_da[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
}

hipError_t launch_l1_fwd(int64_t n, const float* a, const float* b, float b_scalar, float* loss, float* partial,
                         hipStream_t st) {
    const int nb = (int)l1_partial_floats(n);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, st, n, a, b, b_scalar, partial);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, st, partial, nb, 1.0 / (double)n, loss);
    return hipGetLastError();
}

hipError_t launch_l1_bwd(int64_t n, const float* grad_loss, const float* a, const float* b, float b_scalar,
                         float* dL_da, hipStream_t st) {
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, grad_loss, a, b,
                       b_scalar, (float)(1.0 / (double)n), dL_da);
    return hipGetLastError();
}

}  // namespace gs
