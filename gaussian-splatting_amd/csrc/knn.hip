// Mean squared distance to the 3 nearest neighbours (simple-knn's distCUDA2).
// Reference: submodules/simple-knn/simple_knn.cu:46-222.
//
// Same scheme as the reference -- Morton order, boxes of 1024 consecutive
// points, exact search with box pruning -- with these changes for the hardware:
//   * the bounding box stays on the device (no host round trips);
//   * points are gathered once into Morton order (float4: xyz + original index)
//     so the search streams them instead of chasing indices;
//   * a second level of boxes (32 boxes each) lets a whole wave skip 32768
//     points with one test;
//   * candidate points and boxes are wave-uniform, so they arrive through
//     scalar loads and the vector units only do distance + insertion.
// The result per point is the exact 3-NN mean, which does not depend on the
// visiting order, so it equals the reference's bit for bit.
#include <cfloat>

#include "kernels.h"

namespace gs {

#ifndef GS_KNN_BOX
#define GS_KNN_BOX 1024
#endif
constexpr int KNN_BOX = GS_KNN_BOX;   // (a multiple of the 256 points of a workgroup: a wave lies inside one box)
constexpr int KNN_SUPER = 32;  // boxes per super box

struct KnnTemp {
    float* bbox_part;    // [nred][6]
    float* bbox;         // [6] min xyz, max xyz
    uint64_t* keys[2];   // [P]
    uint32_t* vals[2];   // [P]
    float4* sorted;      // [P] xyz + original index bits
    Box* boxes;          // [nb]
    Box* supers;         // [nsb]
    void* sort_temp;
    size_t bytes;
};

constexpr int KNN_RED_BLOCKS = 1024;

static KnnTemp knn_temp(void* base, size_t P) {
    KnnTemp t;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off = align_up(off + n, 256); return q; };
    const size_t nb = (P + KNN_BOX - 1) / KNN_BOX, nsb = (nb + KNN_SUPER - 1) / KNN_SUPER;
    t.bbox_part = (float*)take(KNN_RED_BLOCKS * 6 * sizeof(float));
    t.bbox = (float*)take(6 * sizeof(float));
    t.keys[0] = (uint64_t*)take(P * sizeof(uint64_t));
    t.keys[1] = (uint64_t*)take(P * sizeof(uint64_t));
    t.vals[0] = (uint32_t*)take(P * sizeof(uint32_t));
    t.vals[1] = (uint32_t*)take(P * sizeof(uint32_t));
    t.sorted = (float4*)take(P * sizeof(float4));
    t.boxes = (Box*)take(nb * sizeof(Box));
    t.supers = (Box*)take(nsb * sizeof(Box));
    t.sort_temp = take(sort_temp_view(nullptr, P, SORT_MAX_PASSES).bytes);
    t.bytes = off;
    return t;
}

size_t knn_temp_bytes(int P) { return knn_temp(nullptr, (size_t)(P > 0 ? P : 1)).bytes; }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin_(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax_(v, __shfl_xor(v, off));
    return v;
}

// Block-wide min/max of 6 values (min xyz, max xyz); result in thread 0.
__device__ __forceinline__ void block_minmax(float* mn, float* mx, float (*s)[6]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min(mn[c]);
        mx[c] = wave_max(mx[c]);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[w][c] = mn[c];
            s[w][3 + c] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mn[c] = fmin_(mn[c], s[i][c]);
                mx[c] = fmax_(mx[c], s[i][3 + c]);
            }
    }
}

// simple_knn.cu:192-200: the reduction is seeded with the origin.
__global__ void __launch_bounds__(256) knn_bbox_partial_kernel(int P, const float* pts, float* part) {
    __shared__ float s[4][6];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * (size_t)i + c];
            mn[c] = fmin_(mn[c], v);
            mx[c] = fmax_(mx[c], v);
        }
    }
    block_minmax(mn, mx, s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            part[blockIdx.x * 6 + c] = mn[c];
            part[blockIdx.x * 6 + 3 + c] = mx[c];
        }
    }
}

__global__ void __launch_bounds__(256) knn_bbox_final_kernel(int n, const float* part, float* bbox) {
    __shared__ float s[4][6];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mn[c] = fmin_(mn[c], part[i * 6 + c]);
            mx[c] = fmax_(mx[c], part[i * 6 + 3 + c]);
        }
    }
    block_minmax(mn, mx, s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            bbox[c] = mn[c];
            bbox[3 + c] = mx[c];
        }
    }
}

// simple_knn.cu:55-71
__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* pts, const float* bbox, uint64_t* keys,
                                                         uint32_t* vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const V3 mn = v3(bbox[0], bbox[1], bbox[2]), mx = v3(bbox[3], bbox[4], bbox[5]);
    const V3 p = v3(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
    keys[i] = morton_code(p, mn, mx);
    vals[i] = (uint32_t)i;
}

// Gather into Morton order and reduce each run of 1024 points to its box
// (simple_knn.cu:79-118).  256 threads x 4 points.
__global__ void __launch_bounds__(256) knn_gather_boxes_kernel(int P, const float* pts, const uint32_t* order,
                                                               float4* sorted, Box* boxes) {
    __shared__ float s[4][6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = blockIdx.x * KNN_BOX + r * 256 + threadIdx.x;
        if (i < P) {
            const uint32_t o = order[i];
            const float x = pts[3 * (size_t)o], y = pts[3 * (size_t)o + 1], z = pts[3 * (size_t)o + 2];
            sorted[i] = make_float4(x, y, z, __uint_as_float(o));
            mn[0] = fmin_(mn[0], x); mn[1] = fmin_(mn[1], y); mn[2] = fmin_(mn[2], z);
            mx[0] = fmax_(mx[0], x); mx[1] = fmax_(mx[1], y); mx[2] = fmax_(mx[2], z);
        }
    }
    block_minmax(mn, mx, s);
    if (threadIdx.x == 0) {
        Box b = {v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2])};
        boxes[blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(256) knn_super_kernel(int nb, const Box* boxes, Box* supers) {
    const int sb = blockIdx.x * 256 + threadIdx.x;
    if (sb * KNN_SUPER >= nb) return;
    Box u = boxes[sb * KNN_SUPER];
    for (int b = sb * KNN_SUPER + 1; b < min(nb, (sb + 1) * KNN_SUPER); ++b) {
        const Box o = boxes[b];
        u.mn = v3(fmin_(u.mn.x, o.mn.x), fmin_(u.mn.y, o.mn.y), fmin_(u.mn.z, o.mn.z));
        u.mx = v3(fmax_(u.mx.x, o.mx.x), fmax_(u.mx.y, o.mx.y), fmax_(u.mx.z, o.mx.z));
    }
    supers[sb] = u;
}

// knn3_update (gs_math.h) as a network of minima and maxima: the three smallest of best[0] <= best[1] <=
// best[2] and dist, five full-rate instructions instead of three compare-and-swap steps with selects.
__device__ __forceinline__ void knn3_insert(float dist, float* best) {
    const float r0 = __builtin_fmaxf(best[0], dist);
    best[0] = __builtin_fminf(best[0], dist);
    const float r1 = __builtin_fmaxf(best[1], r0);
    best[1] = __builtin_fminf(best[1], r0);
    best[2] = __builtin_fminf(best[2], r1);
}
__device__ __forceinline__ float knn_dist2(V3 ref, float4 o) { return knn_dist2(ref, v3(o.x, o.y, o.z)); }

#ifdef GS_KNN_STATS
static __device__ unsigned long long g_knn_stats[8];
extern "C" void gsk_debug_stats(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_stats), sizeof(g_knn_stats)); }
#endif
// simple_knn.cu:148-184
__global__ void __launch_bounds__(256) knn_search_kernel(int P, const float4* __restrict__ sorted,
                                                         const Box* __restrict__ boxes,
                                                         const Box* __restrict__ supers, float* __restrict__ dists) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < P;
    const float4 me = sorted[live ? idx : P - 1];
    const V3 p = v3(me.x, me.y, me.z);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    if (live) {
        for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
            if (i == idx) continue;
            const float4 o = sorted[i];
            knn3_update(p, v3(o.x, o.y, o.z), best);
        }
    }
    const float reject = live ? best[2] : -1.0f;  // dead lanes reject every box
    best[0] = best[1] = best[2] = FLT_MAX;

    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    const int nsb = (nb + KNN_SUPER - 1) / KNN_SUPER;
    // The wave's own box first (its 64 points are consecutive in Morton order and lie in one box): after it
    // the third-nearest distance so far is close to the final one, and the box tests below reject with that
    // instead of the distance among the six Morton neighbours.  The result does not depend on the order.
    const int own = (blockIdx.x * 256 + (threadIdx.x & ~63)) / KNN_BOX;
    // (No lane is masked in these loops: a candidate a lane did not ask for can only be one of its true
    // neighbours or lose against them, and the lanes past P throw their result away.  Only the point itself
    // has to be kept out, and it lies in the own box.)
    {
        const int jend = min(P, (own + 1) * KNN_BOX);
#pragma unroll 4
        for (int j = own * KNN_BOX; j < jend; ++j) {
            const float d2 = knn_dist2(p, sorted[j]);
            knn3_insert(j == idx ? FLT_MAX : d2, best);
        }
    }
    for (int sb = 0; sb < nsb; ++sb) {
        const float ds = box_point_dist2(supers[sb], p);
        if (__ballot(!(ds > reject || ds > best[2])) == 0) continue;
        const int bend = min(nb, (sb + 1) * KNN_SUPER);
        for (int b = sb * KNN_SUPER; b < bend; ++b) {
            if (b == own) continue;
            const float d = box_point_dist2(boxes[b], p);
            const bool want = !(d > reject || d > best[2]);
            if (__ballot(want) == 0) continue;
#ifdef GS_KNN_STATS
            {
                const int nwant = __popcll(__ballot(want));
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&g_knn_stats[0], 1ull);                       // box visits
                    atomicAdd(&g_knn_stats[1], (unsigned long long)nwant);  // lanes that wanted
                    if (nwant <= 4) atomicAdd(&g_knn_stats[2], 1ull);       // visits for <= 4 lanes
                    if (nwant <= 16) atomicAdd(&g_knn_stats[3], 1ull);
                }
            }
#endif
            const int jend = min(P, (b + 1) * KNN_BOX);
            // four candidates at a time; the insertions are skipped when none of them comes closer than any
            // lane's third-nearest so far, which after the own box is the usual case
            int j = b * KNN_BOX;
            for (; j + 4 <= jend; j += 4) {
                const float d0 = knn_dist2(p, sorted[j]), d1 = knn_dist2(p, sorted[j + 1]);
                const float d2 = knn_dist2(p, sorted[j + 2]), d3 = knn_dist2(p, sorted[j + 3]);
                const float m = __builtin_fminf(__builtin_fminf(d0, d1), __builtin_fminf(d2, d3));
                if (__ballot(m < best[2]) == 0) continue;
                knn3_insert(d0, best);
                knn3_insert(d1, best);
                knn3_insert(d2, best);
                knn3_insert(d3, best);
            }
            for (; j < jend; ++j) knn3_insert(knn_dist2(p, sorted[j]), best);
        }
    }
    if (live) dists[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

hipError_t launch_knn(int P, const float* points, float* out, void* temp, hipStream_t st) {
    const KnnTemp t = knn_temp(temp, (size_t)P);
    const int nred = min(KNN_RED_BLOCKS, (P + 255) / 256);
    hipLaunchKernelGGL(knn_bbox_partial_kernel, dim3(nred), dim3(256), 0, st, P, points, t.bbox_part);
    hipLaunchKernelGGL(knn_bbox_final_kernel, dim3(1), dim3(256), 0, st, nred, t.bbox_part, t.bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, points, t.bbox, t.keys[0],
                       t.vals[0]);
    uint64_t* k[2] = {t.keys[0], t.keys[1]};
    uint32_t* v[2] = {t.vals[0], t.vals[1]};
    int res = 0;
    hipError_t e = launch_radix_sort(k, v, P, 0, 30, t.sort_temp, st, &res);
    if (e != hipSuccess) return e;
    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    const int nsb = (nb + KNN_SUPER - 1) / KNN_SUPER;
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(nb), dim3(256), 0, st, P, points, t.vals[res], t.sorted, t.boxes);
    hipLaunchKernelGGL(knn_super_kernel, dim3((nsb + 255) / 256), dim3(256), 0, st, nb, t.boxes, t.supers);
    hipLaunchKernelGGL(knn_search_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, t.sorted, t.boxes, t.supers,
                       out);
    return hipGetLastError();
}

}  // namespace gs
