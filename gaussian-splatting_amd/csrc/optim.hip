// Fused Adam step over one parameter tensor of the Gaussians ([N, M] floats,
// row = Gaussian): parameter, gradient and both moments are read once and the
// parameter and moments written once (28 bytes per element), rows of Gaussians
// that are not flagged visible are skipped without being read.
//
// The reference optimises with torch.optim.Adam (scene/gaussian_model.py:112-113)
// and tries to import a sparse variant, SparseGaussianAdam(params, lr, eps)
// .step(visibility, N), from the rasterizer package (train.py:38-42, 339-343);
// that class lives in graphdeco-inria/diff-gaussian-rasterization (branch
// 3dgs_accel), which the fork does not vendor.  Its published update is Adam
// without bias correction on the visible rows only.
#include "kernels.h"

namespace gs {

template <int VEC>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                   const uint8_t* __restrict__ visible, int64_t total, int M,
                                                   const AdamConst c) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= total) return;
    if (visible && !visible[i / M]) return;  // VEC divides M, so the VEC elements share a row
    if (VEC == 4) {
        float4 p = *reinterpret_cast<float4*>(param + i);
        const float4 g = *reinterpret_cast<const float4*>(grad + i);
        float4 m = *reinterpret_cast<float4*>(exp_avg + i);
        float4 v = *reinterpret_cast<float4*>(exp_avg_sq + i);
        p.x = adam_update(p.x, g.x, &m.x, &v.x, c);
        p.y = adam_update(p.y, g.y, &m.y, &v.y, c);
        p.z = adam_update(p.z, g.z, &m.z, &v.z, c);
        p.w = adam_update(p.w, g.w, &m.w, &v.w, c);
        *reinterpret_cast<float4*>(param + i) = p;
        *reinterpret_cast<float4*>(exp_avg + i) = m;
        *reinterpret_cast<float4*>(exp_avg_sq + i) = v;
    } else {
        float m = exp_avg[i], v = exp_avg_sq[i];
        param[i] = adam_update(param[i], grad[i], &m, &v, c);
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
    }
}

hipError_t launch_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                       int64_t N, int M, const AdamConst& c, hipStream_t st) {
    const int64_t total = N * M;
    if (total <= 0) return hipSuccess;
    // 16-byte accesses when they cannot straddle a row (sparse) or always (dense); tensors are 16-byte aligned
    const bool vec = visible ? (M % 4 == 0) : (total % 4 == 0);
    const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
    if (vec && aligned) {
        const int64_t threads = total / 4;
        hipLaunchKernelGGL(adam_kernel<4>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, param, grad,
                           exp_avg, exp_avg_sq, visible, total, M, c);
    } else {
        hipLaunchKernelGGL(adam_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, param, grad, exp_avg,
                           exp_avg_sq, visible, total, M, c);
    }
    return hipGetLastError();
}

// Densification statistics of one rendered view in one pass over the Gaussians
// (train.py:298-312 and scene/gaussian_model.py:202-214 do it with a nonzero()
// of the visibility mask, which blocks the host, and five indexed updates):
// for every Gaussian with radii > 0
//   max_radii2D       = max(max_radii2D, radii)
//   xyz_gradient_accum += |dL/dmeans2D.xy|
//   denom             += 1
__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int* __restrict__ radii,
                                                            const float* __restrict__ dL_dmeans2D,
                                                            float* __restrict__ xyz_gradient_accum,
                                                            float* __restrict__ denom, float* __restrict__ max_radii2D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (!(r > 0)) return;
    const float gx = dL_dmeans2D[3 * (size_t)i], gy = dL_dmeans2D[3 * (size_t)i + 1];
    max_radii2D[i] = fmax_(max_radii2D[i], (float)r);
    xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

hipError_t launch_densify_stats(int P, const int* radii, const float* dL_dmeans2D, float* xyz_gradient_accum,
                                float* denom, float* max_radii2D, hipStream_t st) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, radii, dL_dmeans2D,
                       xyz_gradient_accum, denom, max_radii2D);
    return hipGetLastError();
}

// ------------------------------------------------------------------ row compaction
// Pruning (scene/gaussian_model.py:402-441) keeps the rows of a mask in seven
// parameter tensors, their fourteen Adam moment tensors and four statistics
// tensors; the reference does it with boolean indexing, i.e. one nonzero()
// (a host block) and one gather per tensor.  Here the mask is turned once into
// the list of kept row indices (ascending, as boolean indexing orders them),
// the host learns the count with one copy, and every tensor is gathered with
// that list.
__global__ void __launch_bounds__(256) compact_count_kernel(int64_t N, const uint8_t* __restrict__ keep,
                                                            uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t m = __ballot(i < N && keep[i] != 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256) compact_map_kernel(int64_t N, const uint8_t* __restrict__ keep,
                                                          const uint32_t* __restrict__ block_offsets,
                                                          uint32_t* __restrict__ src_of_dst) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool k = i < N && keep[i] != 0;
    const uint64_t m = __ballot(k);
    if (lane == 0) s_w[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = block_offsets[blockIdx.x];
    for (int j = 0; j < w; ++j) base += s_w[j];
    if (k) src_of_dst[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)i;
}

// dst[r, :] = src[src_of_dst[r], :], rows of M floats; one thread per 4 bytes, consecutive threads
// walk a destination row.
__global__ void __launch_bounds__(256) gather_rows_kernel(const uint32_t* __restrict__ src,
                                                          uint32_t* __restrict__ dst,
                                                          const uint32_t* __restrict__ src_of_dst, int64_t total,
                                                          int M) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / M;
    const int c = (int)(i - r * M);
    dst[i] = src[(int64_t)src_of_dst[r] * M + c];
}

// ------------------------------------------------------------------ densification: who is cloned, split, pruned
// GaussianModel.densify_and_prune (scene/gaussian_model.py:216-252) decides per Gaussian from the
// accumulated screen-space gradient norm and from its size:
//   grads = xyz_gradient_accum / denom, NaN -> 0                                    (:224-225)
//   clone:  |grads| >= grad_threshold  and  max(exp(scaling)) <= size_threshold      (:263-270)
//   split:  grads  >= grad_threshold  and  max(exp(scaling)) >  size_threshold       (:356-364)
// with size_threshold = percent_dense * scene_extent.  The split is decided after the clones were
// appended, with the gradients padded by zeros for them: rows at and beyond n_grads have gradient 0.
// The reference does it with a dozen elementwise kernels and temporaries; this is one pass.
__global__ void __launch_bounds__(256) densify_select_kernel(int64_t N, int64_t n_grads,
                                                             const float* __restrict__ xyz_gradient_accum,
                                                             const float* __restrict__ denom,
                                                             const float* __restrict__ scaling, float grad_threshold,
                                                             float size_threshold, int split,
                                                             uint8_t* __restrict__ selected) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float g = 0.0f;
    if (i < n_grads) {
        g = xyz_gradient_accum[i] / denom[i];
        if (g != g) g = 0.0f;
    }
    const float smax = fmaxf(fmaxf(expf(scaling[3 * i]), expf(scaling[3 * i + 1])), expf(scaling[3 * i + 2]));
    bool sel;
    if (split) sel = g >= grad_threshold && smax > size_threshold;
    else sel = sqrtf(g * g) >= grad_threshold && smax <= size_threshold;
    selected[i] = sel ? 1 : 0;
}

// The prune decision of densify_and_prune (scene/gaussian_model.py:237-244): opacity below the
// minimum, or (when a screen-size limit is given) too large on screen or in the world; `also`
// (nullable) is OR-ed in: the parents a split has just replaced (:388-392).
__global__ void __launch_bounds__(256) prune_mask_kernel(int64_t N, const float* __restrict__ opacity,
                                                         const float* __restrict__ max_radii2D,
                                                         const float* __restrict__ scaling, float min_opacity,
                                                         float max_screen_size, float max_world_size,
                                                         const uint8_t* __restrict__ also, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    bool m = (1.0f / (1.0f + expf(-opacity[i]))) < min_opacity;
    if (max_screen_size > 0.0f) {
        const float smax = fmaxf(fmaxf(expf(scaling[3 * i]), expf(scaling[3 * i + 1])), expf(scaling[3 * i + 2]));
        m = m || max_radii2D[i] > max_screen_size || smax > max_world_size;
    }
    if (also) m = m || also[i] != 0;
    mask[i] = m ? 1 : 0;
}

// mask[i] |= (counts[0][i] != 0) | ... | (counts[n-1][i] != 0) (accumulate), or = (overwrite): the Gaussians
// that a pixel of some view reached, from the views' gauss_count outputs (forward.cu:431-432).  Four
// Gaussians per thread: 16-byte loads of the counts, one 4-byte access of the mask.
constexpr int MARK_MAX_ARRAYS = 8;
struct MarkArgs {
    const int* counts[MARK_MAX_ARRAYS];
    int n;
    int accumulate;
    int64_t N;
    uint8_t* mask;
};

__global__ void __launch_bounds__(256) mark_nonzero_kernel(const MarkArgs a) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= a.N) return;
    if (i0 + 4 <= a.N) {
        uint32_t m = a.accumulate ? *reinterpret_cast<const uint32_t*>(a.mask + i0) : 0u;
        // all the arrays' loads first (the launcher repeats the first array in the unused slots): a loop over
        // a run-time count would wait for each load before issuing the next
        int4 c[MARK_MAX_ARRAYS];
#pragma unroll
        for (int v = 0; v < MARK_MAX_ARRAYS; ++v) c[v] = *reinterpret_cast<const int4*>(a.counts[v] + i0);
#pragma unroll
        for (int v = 0; v < MARK_MAX_ARRAYS; ++v)
            m |= (c[v].x != 0 ? 1u : 0u) | (c[v].y != 0 ? 1u << 8 : 0u) | (c[v].z != 0 ? 1u << 16 : 0u) |
                 (c[v].w != 0 ? 1u << 24 : 0u);
        *reinterpret_cast<uint32_t*>(a.mask + i0) = m;
    } else {
        for (int64_t i = i0; i < a.N; ++i) {
            uint8_t m = a.accumulate ? a.mask[i] : (uint8_t)0;
            for (int v = 0; v < a.n; ++v) m |= a.counts[v][i] != 0 ? 1 : 0;
            a.mask[i] = m;
        }
    }
}

// densify_and_split (scene/gaussian_model.py:367-377): every selected Gaussian is replaced by n
// children.  Child j (of K * n, laid out as `.repeat(n, 1)` lays them out: child j belongs to parent
// j mod K) sits at parent position + R(q / |q|) (z_j * exp(scaling)), z_j a standard normal sample
// the caller brings (torch.normal(0, std) is z * std), and has scale exp(scaling) / (0.8 n), stored
// as its logarithm.  parent_of_child[j] is written for the plain row copies of the other tensors.
__global__ void __launch_bounds__(256) split_children_kernel(int64_t K, int n, const uint32_t* __restrict__ parents,
                                                             const float* __restrict__ xyz,
                                                             const float* __restrict__ scaling,
                                                             const float* __restrict__ rotation,
                                                             const float* __restrict__ normals,
                                                             float* __restrict__ new_xyz, float* __restrict__ new_scaling,
                                                             uint32_t* __restrict__ parent_of_child) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K * n) return;
    const uint32_t p = parents[j % K];
    parent_of_child[j] = p;
    const float sx = expf(scaling[3 * (size_t)p]), sy = expf(scaling[3 * (size_t)p + 1]), sz = expf(scaling[3 * (size_t)p + 2]);
    const float ox = normals[3 * j] * sx, oy = normals[3 * j + 1] * sy, oz = normals[3 * j + 2] * sz;
    // utils/general_utils.py:87-108 (build_rotation): normalised quaternion (r, x, y, z) to a matrix
    float r = rotation[4 * (size_t)p], x = rotation[4 * (size_t)p + 1], y = rotation[4 * (size_t)p + 2],
          z = rotation[4 * (size_t)p + 3];
    const float norm = sqrtf(r * r + x * x + y * y + z * z);
    r /= norm; x /= norm; y /= norm; z /= norm;
    const float R00 = 1 - 2 * (y * y + z * z), R01 = 2 * (x * y - r * z), R02 = 2 * (x * z + r * y);
    const float R10 = 2 * (x * y + r * z), R11 = 1 - 2 * (x * x + z * z), R12 = 2 * (y * z - r * x);
    const float R20 = 2 * (x * z - r * y), R21 = 2 * (y * z + r * x), R22 = 1 - 2 * (x * x + y * y);
    new_xyz[3 * j] = (R00 * ox + R01 * oy + R02 * oz) + xyz[3 * (size_t)p];
    new_xyz[3 * j + 1] = (R10 * ox + R11 * oy + R12 * oz) + xyz[3 * (size_t)p + 1];
    new_xyz[3 * j + 2] = (R20 * ox + R21 * oy + R22 * oz) + xyz[3 * (size_t)p + 2];
    const float shrink = 0.8f * (float)n;
    new_scaling[3 * j] = logf(sx / shrink);
    new_scaling[3 * j + 1] = logf(sy / shrink);
    new_scaling[3 * j + 2] = logf(sz / shrink);
}

hipError_t launch_densify_select(int64_t N, int64_t n_grads, const float* xyz_gradient_accum, const float* denom,
                                 const float* scaling, float grad_threshold, float size_threshold, int split,
                                 uint8_t* selected, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(densify_select_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, n_grads,
                       xyz_gradient_accum, denom, scaling, grad_threshold, size_threshold, split, selected);
    return hipGetLastError();
}

hipError_t launch_prune_mask(int64_t N, const float* opacity, const float* max_radii2D, const float* scaling,
                             float min_opacity, float max_screen_size, float max_world_size, const uint8_t* also,
                             uint8_t* mask, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(prune_mask_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, opacity, max_radii2D,
                       scaling, min_opacity, max_screen_size, max_world_size, also, mask);
    return hipGetLastError();
}

hipError_t launch_mark_nonzero(int64_t N, int n, const int* const* counts, uint8_t* mask, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    if (n <= 0) return hipMemsetAsync(mask, 0, (size_t)N, st);
    for (int first = 0; first < n; first += MARK_MAX_ARRAYS) {
        MarkArgs a;
        a.n = n - first < MARK_MAX_ARRAYS ? n - first : MARK_MAX_ARRAYS;
        for (int v = 0; v < MARK_MAX_ARRAYS; ++v) a.counts[v] = counts[first + (v < a.n ? v : 0)];   // every slot is read
        a.accumulate = first > 0;
        a.N = N;
        a.mask = mask;
        hipLaunchKernelGGL(mark_nonzero_kernel, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_split_children(int64_t K, int n, const uint32_t* parents, const float* xyz, const float* scaling,
                                 const float* rotation, const float* normals, float* new_xyz, float* new_scaling,
                                 uint32_t* parent_of_child, hipStream_t st) {
    if (K <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_children_kernel, dim3((unsigned)((K * n + 255) / 256)), dim3(256), 0, st, K, n, parents, xyz,
                       scaling, rotation, normals, new_xyz, new_scaling, parent_of_child);
    return hipGetLastError();
}

size_t compact_temp_bytes(int64_t N) {
    const size_t nb = (size_t)((N + 255) / 256);
    return align_up((nb + 1) * sizeof(uint32_t), 256) + 4 * sizeof(uint64_t);
}

hipError_t launch_compact_map(int64_t N, const uint8_t* keep, uint32_t* src_of_dst, void* temp, hipStream_t st,
                              uint64_t** total_dev) {
    const int nb = (int)((N + 255) / 256);
    uint32_t* block_sums = (uint32_t*)temp;
    uint64_t* total = (uint64_t*)((char*)temp + align_up(((size_t)nb + 1) * sizeof(uint32_t), 256));
    *total_dev = total;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, st, N, keep, block_sums);
    hipError_t e = launch_scan_u32(block_sums, nb, total, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(compact_map_kernel, dim3(nb), dim3(256), 0, st, N, keep, block_sums, src_of_dst);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const void* src, void* dst, const uint32_t* src_of_dst, int64_t K, int M,
                              hipStream_t st) {
    const int64_t total = K * M;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t*)src, (uint32_t*)dst, src_of_dst, total, M);
    return hipGetLastError();
}

}  // namespace gs
