// Device-side helpers for gfx950 (wave64): DPP reductions, scans, and the
// layout of the three scratch buffers that travel from forward to backward.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gs_math.h"

namespace gs {

constexpr int WAVE = 64;

// ------------------------------------------------------------------ wave ops

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}

// Sum over the 64 lanes; result valid in lane 63 only.
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);  // row_mirror
    v += dpp_f<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum over the 64 lanes, returned wave-uniform (SGPR).
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_sum_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_u<0xB1, 0xf>(v);
    v += dpp_u<0x4E, 0xf>(v);
    v += dpp_u<0x141, 0xf>(v);
    v += dpp_u<0x140, 0xf>(v);
    v += dpp_u<0x142, 0xa>(v);
    v += dpp_u<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive prefix sum over the 64 lanes (row_shr scan, then row broadcasts).
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    v += dpp_u<0x111, 0xf>(v);  // row_shr:1
    v += dpp_u<0x112, 0xf>(v);  // row_shr:2
    v += dpp_u<0x114, 0xf>(v);  // row_shr:4
    v += dpp_u<0x118, 0xf>(v);  // row_shr:8
    v += dpp_u<0x142, 0xa>(v);  // row_bcast:15
    v += dpp_u<0x143, 0xc>(v);  // row_bcast:31
    return v;
}

// Inclusive running maximum over the 64 lanes (the same steps; lanes a step does not reach read 0).
__device__ __forceinline__ uint32_t wave_scan_incl_max_u32(uint32_t v) {
    v = max(v, dpp_u<0x111, 0xf>(v));  // row_shr:1
    v = max(v, dpp_u<0x112, 0xf>(v));  // row_shr:2
    v = max(v, dpp_u<0x114, 0xf>(v));  // row_shr:4
    v = max(v, dpp_u<0x118, 0xf>(v));  // row_shr:8
    v = max(v, dpp_u<0x142, 0xa>(v));  // row_bcast:15
    v = max(v, dpp_u<0x143, 0xc>(v));  // row_bcast:31
    return v;
}

// Lane exchanges inside a quad (DPP, no LDS): value of lane ^ 1, ^ 2.
__device__ __forceinline__ float lane_xor1(float v) { return dpp_f<0xB1, 0xf>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_f<0x4E, 0xf>(v); }   // quad_perm [2,3,0,1]

// Transposing reduction of 16 per-lane values over the wave: on return every lane holds the sum over
// all 64 lanes of v[wave_transpose_slot(lane)].  Each fold halves the number of live registers: of a
// pair of values a lane keeps the one that one of its lane bits selects and adds the partner lane's
// copy of it.  The two big folds (8 and 4 pairs) use lane bits 3 and 2, which are *bank* bits of the
// DPP control (a bank is four lanes): a v_add_f32_dpp restricted by bank_mask writes only the lanes
// that keep that value, so a pair folds in two instructions with no select.  The two small folds go
// across the rows of 16 lanes (lane bits 5 and 4): v_permlane32_swap / v_permlane16_swap put the
// halves (the odd and even rows) of two registers side by side, so one swap and one add fold a pair.
// What is left is one register whose four lanes of a quad hold partial sums of the same value: two
// plain DPP adds.  26 DPP adds, 3 swaps and 3 adds in all, where a value-by-value reduction of 16
// values needs 96 DPP adds.  The asm blocks start with the wait states a DPP read of a just-written
// VGPR needs (the compiler does not see inside them).
__device__ __forceinline__ int wave_transpose_slot(int lane) {
    return ((lane >> 4) & 1) * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
}

// N: 16, or 14 when v[14] and v[15] are zero (their fold is skipped).
template <int N = 16>
__device__ __forceinline__ float wave_transpose_sum16(const float (&v)[16]) {
    static_assert(N == 16 || N == 14, "16 values, or 14 and two zeros");
    float a[8], b[4];
    // a[k] = lane bit 3 ? v[2k + 1] : v[2k], summed over lane bit 3
#define GS_FOLD3(i, lo, hi) \
    "v_add_f32_dpp %" #i ", %" #lo ", %" #lo " row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
    "v_add_f32_dpp %" #i ", %" #hi ", %" #hi " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
    if (N == 16) {
        asm("s_nop 1\n\t"
            GS_FOLD3(0, 8, 9) GS_FOLD3(1, 10, 11) GS_FOLD3(2, 12, 13) GS_FOLD3(3, 14, 15)
            GS_FOLD3(4, 16, 17) GS_FOLD3(5, 18, 19) GS_FOLD3(6, 20, 21) GS_FOLD3(7, 22, 23)
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
              "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    } else {
        asm("s_nop 1\n\t"
            GS_FOLD3(0, 7, 8) GS_FOLD3(1, 9, 10) GS_FOLD3(2, 11, 12) GS_FOLD3(3, 13, 14)
            GS_FOLD3(4, 15, 16) GS_FOLD3(5, 17, 18) GS_FOLD3(6, 19, 20)
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6])
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
              "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]));
        a[7] = 0.f;
    }
#undef GS_FOLD3
    // b[i] = lane bit 2 ? a[2i + 1] : a[2i], summed over lane bit 2
#define GS_FOLD2(i, lo, hi) \
    "v_add_f32_dpp %" #i ", %" #lo ", %" #lo " row_shl:4 row_mask:0xf bank_mask:0x5\n\t" \
    "v_add_f32_dpp %" #i ", %" #hi ", %" #hi " row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
    asm("s_nop 1\n\t"
        GS_FOLD2(0, 4, 5) GS_FOLD2(1, 6, 7) GS_FOLD2(2, 8, 9) GS_FOLD2(3, 10, 11)
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
#undef GS_FOLD2
    // lanes 0-31 keep b[2i], lanes 32-63 b[2i + 1], summed over lane bit 5
    const auto h0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(b[0]), __float_as_uint(b[1]), false, false);
    const auto h1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(b[2]), __float_as_uint(b[3]), false, false);
    const float c0 = __uint_as_float(h0[0]) + __uint_as_float(h0[1]);
    const float c1 = __uint_as_float(h1[0]) + __uint_as_float(h1[1]);
    // even rows keep c0, odd rows c1, summed over lane bit 4
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0), __float_as_uint(c1), false, false);
    float d = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    d += lane_xor1(d);
    d += lane_xor2(d);
    return d;
}

// Lane masks.  A ballot of a direct comparison is one v_cmp into an SGPR pair, and a
// wave-uniform mask turns back into a per-lane predicate for free (the hardware predicate
// *is* such a mask).  Keeping per-lane flags as 64-bit masks and combining them with scalar
// and / andn2 avoids the v_cndmask + v_cmp pair the compiler emits for a ballot of a
// boolean expression.
__device__ __forceinline__ uint64_t lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Which of the four 8x8 quadrants of the 16x16 tile at (x0, y0) can hold a pixel with
// power >= tau for the splat at (x, y) with conic (a, b, c) — bit q for the quadrant at
// (x0 + 8 (q & 1), y0 + 8 (q >> 1)) — from the bounding box of the ellipse
// a dx^2 + 2 b dx dy + c dy^2 <= -2 tau, whose half extents are sqrt(-2 tau c / det) and
// sqrt(-2 tau a / det).  Conservative: the box is enlarged by 0.1 % + 0.01 pixel, which is far more
// than the rounding of the power as the blend kernels compute it can move the ellipse as long as
// the conic is not nearly degenerate (b^2 < 0.99 a c: then a dx^2 + c dy^2 <= 200 times the
// quadratic form, so a few ulps of its terms are < 1e-4 of its value); nearly degenerate or
// indefinite conics get all four bits.  tau > 0 (alpha >= 1/255 out of reach): no bit.
// A quadrant without its bit has, besides no pixel with power >= tau, no pixel with power > 0
// either, also as computed (b^2 < 0.99 a c puts the power at least 0.5 % of its largest term
// below zero; only at dx = dy = 0 it is 0): the forward blend's statistics rely on that.
__device__ __forceinline__ uint32_t quadrant_reach_bits(float x, float y, float a, float b, float c, float tau,
                                                        float x0, float y0) {
    const float det = a * c - b * b;
    if (!(a > 0.0f && c > 0.0f && det > 0.01f * a * c)) return 0xFu;
    if (tau > 0.0f) return 0u;
    const float s = -2.0f * tau * __builtin_amdgcn_rcpf(det);
    const float hx = __builtin_amdgcn_sqrtf(s * c) * 1.001f + 0.01f;
    const float hy = __builtin_amdgcn_sqrtf(s * a) * 1.001f + 0.01f;
    // pixel centres of the quadrant columns: x0 .. x0 + 7 and x0 + 8 .. x0 + 15 (rows likewise)
    const uint32_t cx = (x - hx <= x0 + 7.0f && x + hx >= x0 ? 1u : 0u) | (x - hx <= x0 + 15.0f && x + hx >= x0 + 8.0f ? 2u : 0u);
    const uint32_t cy = (y - hy <= y0 + 7.0f && y + hy >= y0 ? 1u : 0u) | (y - hy <= y0 + 15.0f && y + hy >= y0 + 8.0f ? 2u : 0u);
    // bit q = column bit (q & 1) and row bit (q >> 1)
    return ((cy & 1u) ? cx : 0u) | ((cy & 2u) ? cx << 2 : 0u);
}
__device__ __forceinline__ bool lane_in(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

typedef float f2 __attribute__((ext_vector_type(2)));

// m ? v : 0 per lane for a wave-uniform lane mask m.  Written out because the compiler puts
// such a mask into VCC (s_and_b64 vcc, ...) and selects with the VOP2 form, and a VOP2
// v_cndmask whose VCC was last written by the scalar unit costs ~20 cycles on gfx950
// (measured; 4 with the mask in any other SGPR pair).
__device__ __forceinline__ float select_or_zero(uint64_t m, float v) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

// splat_power() with its two squared terms as one packed multiply each; same operation
// order per term, hence the same roundings.  xy: centre, pix: pixel, cxz: (conic.x, conic.z).
__device__ __forceinline__ float splat_power_pk(f2 xy, f2 pix, f2 cxz, float cy, f2* d) {
    *d = xy - pix;
    const f2 t = cxz * *d * *d;
    return -0.5f * (t.x + t.y) - cy * d->x * d->y;
}

// The same with the two squared terms' coefficients given as -0.5 * conic.x and -0.5 * conic.z: scaling
// by a power of two commutes with every rounding, so the result has the same bits as splat_power_pk
// and the multiplication by -0.5 per (pixel, instance) pair is paid once per instance instead.
__device__ __forceinline__ float splat_power_prescaled(f2 xy, f2 pix, f2 mh_cxz, float cy, f2* d) {
    *d = xy - pix;
    const f2 t = mh_cxz * *d * *d;
    return (t.x + t.y) - cy * d->x * d->y;
}

// gs_expf for the blend kernels: no clamps (callers only use the result on lanes whose
// argument is in [-87, 0]) and the scaling by 2^n as one v_ldexp.  Same roundings as gs_expf
// on that interval.
__device__ __forceinline__ float gs_expf_blend(float x) {
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
    return __builtin_amdgcn_ldexpf(p, (int)n);
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

// Block-wide exclusive scan of one u32 per thread, 256 threads.  `tmp` holds
// at least 5 words of LDS.  Returns the exclusive prefix; *total gets the sum.
__device__ __forceinline__ uint32_t block_scan_excl_256(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl_u32(v);
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = tmp[i];
        if (i < w) base += t;
        sum += t;
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}

// ------------------------------------------------------------------ SH rows through LDS
// The SH tensors are [P, 16, 3] floats: 192 bytes per Gaussian.  A lane that
// walks its own row issues 48 strided dword accesses; instead each wave moves
// the 12 KiB of its 64 consecutive rows as whole 16-byte chunks (coalesced)
// through an LDS image and every lane then reads / writes its own row with
// ds_read_b128 / ds_write_b128.  Rows are padded to 13 chunks: with an odd
// chunk stride the 16 lanes that one ds_*_b128 group serves fall on 16
// different 16-byte slots.
constexpr int SH_ROW_F = 48;          // floats per row (M = 16)
constexpr int SH_ROW_CHUNKS = 12;     // float4 per row
constexpr int SH_LDS_STRIDE = 13;     // float4 per row in LDS
constexpr int SH_WAVE_LDS_F4 = 64 * SH_LDS_STRIDE;

// Orders one wave's LDS writes before its later LDS reads (other lanes' data).
// The LDS unit serves a wave's instructions in order, so no hardware wait is
// needed; the wavefront-scope fences keep the compiler from moving accesses
// across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global -> LDS for the rows whose bit is set in `rows`; `g` points at row 0 of the wave.
__device__ __forceinline__ void sh_rows_load(const float* __restrict__ g, float4* lds, uint64_t rows, int nrows, int lane) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int it = 0; it < SH_ROW_CHUNKS; ++it) {
        const int i = it * 64 + lane;          // chunk index inside the wave's 768 chunks
        const int r = i / SH_ROW_CHUNKS, c = i - r * SH_ROW_CHUNKS;
        if (r < nrows && ((rows >> r) & 1ull)) lds[r * SH_LDS_STRIDE + c] = g4[i];
    }
}

// LDS -> global for rows [0, nrows).
__device__ __forceinline__ void sh_rows_store(float* __restrict__ g, const float4* lds, int nrows, int lane) {
    float4* g4 = reinterpret_cast<float4*>(g);
#pragma unroll
    for (int it = 0; it < SH_ROW_CHUNKS; ++it) {
        const int i = it * 64 + lane;
        const int r = i / SH_ROW_CHUNKS, c = i - r * SH_ROW_CHUNKS;
        if (r < nrows) g4[i] = lds[r * SH_LDS_STRIDE + c];
    }
}

// global += LDS for the rows whose bit is set in `rows`.
__device__ __forceinline__ void sh_rows_add(float* __restrict__ g, const float4* lds, uint64_t rows, int nrows, int lane) {
    float4* g4 = reinterpret_cast<float4*>(g);
#pragma unroll
    for (int it = 0; it < SH_ROW_CHUNKS; ++it) {
        const int i = it * 64 + lane;
        const int r = i / SH_ROW_CHUNKS, c = i - r * SH_ROW_CHUNKS;
        if (r < nrows && ((rows >> r) & 1ull)) {
            const float4 o = g4[i], v = lds[r * SH_LDS_STRIDE + c];
            g4[i] = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        }
    }
}

constexpr int REST_ROW_F = 45;              // dwords per row of the [P,15,3] tensor
constexpr int REST_WAVE_F = 64 * REST_ROW_F;  // 2880 dwords = 11520 bytes per wave

// The wave's block of the [P,15,3] tensor (nrows x 45 dwords, contiguous, 16-byte
// aligned because a wave starts at a multiple of 64 rows) between global memory
// and its linear LDS image, 16 bytes per lane and step.  A chunk can straddle
// two rows; it moves if either row's bit is set in `rows`.
// OP 0: LDS = global, 1: global = LDS, 2: global += LDS.
template <int OP>
__device__ __forceinline__ void rest_rows_copy(float* __restrict__ g, float* lds, uint64_t rows, int nrows, int lane) {
    const int total = nrows * REST_ROW_F;
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* l4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int it = 0; it < (REST_WAVE_F / 4 + 63) / 64; ++it) {
        const int i = it * 64 + lane;
        if (4 * i + 3 < total) {
            const int r0 = (4 * i) / REST_ROW_F, r1 = (4 * i + 3) / REST_ROW_F;
            if (((rows >> r0) | (rows >> r1)) & 1ull) {
                if (OP == 0) {
                    l4[i] = g4[i];
                } else if (OP == 1) {
                    g4[i] = l4[i];
                } else {
                    const float4 o = g4[i], v = l4[i];
                    g4[i] = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                }
            }
        }
    }
    // up to three dwords left when the last wave is partial
    const int f = (total & ~3) + lane;
    if (lane < 3 && f < total && ((rows >> (f / REST_ROW_F)) & 1ull)) {
        if (OP == 0) lds[f] = g[f];
        else if (OP == 1) g[f] = lds[f];
        else g[f] += lds[f];
    }
}

// Zeroes `bytes` bytes (a multiple of 4) at `p` (16-byte aligned) with the whole wave: 16 bytes per lane
// and step in address order, dwords for what is left.
__device__ __forceinline__ void wave_zero_bytes(void* p, int bytes, int lane) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const int n4 = bytes >> 4;
    for (int i = lane; i < n4; i += 64) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int f = (n4 << 2) + lane;
    if (f < (bytes >> 2)) reinterpret_cast<float*>(p)[f] = 0.f;
}

// How a per-Gaussian kernel gets at the SH rows: none (precomputed colours), each lane walking its own
// row in global memory (any row size), or the wave's 64 rows staged through LDS in 16-byte chunks
// (16 coefficients; one [P,16,3] tensor, or [P,1,3] + [P,15,3]).
enum ShMode { SH_NONE = 0, SH_DIRECT = 1, SH_CAT16 = 2, SH_SPLIT16 = 3 };

template <int MODE>
struct ShLds {
    static constexpr int bytes_per_wave =
        MODE == SH_CAT16 ? SH_WAVE_LDS_F4 * 16 : (MODE == SH_SPLIT16 ? REST_WAVE_F * 4 : 0);
};

// One lane's row between LDS and registers.
__device__ __forceinline__ void sh_row_get(const float4* lds, int lane, float (&v)[SH_ROW_F]) {
#pragma unroll
    for (int c = 0; c < SH_ROW_CHUNKS; ++c) {
        const float4 q = lds[lane * SH_LDS_STRIDE + c];
        v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
    }
}
__device__ __forceinline__ void sh_row_put(float4* lds, int lane, const float (&v)[SH_ROW_F]) {
#pragma unroll
    for (int c = 0; c < SH_ROW_CHUNKS; ++c)
        lds[lane * SH_LDS_STRIDE + c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// Coefficient accessor over a row held in registers (for sh_to_rgb / sh_to_rgb_bwd).
struct ShRegs {
    const float* v;
    __device__ __forceinline__ V3 operator()(int k) const { return v3(v[3 * k], v[3 * k + 1], v[3 * k + 2]); }
};

// ------------------------------------------------------------------ buffers
// One Gaussian's screen-space record, written by the forward preprocess and
// gathered (one 64-byte line half) by both render kernels.
struct __attribute__((aligned(16))) SplatRec {
    float x, y, cx, cy;        // pixel centre, conic.x, conic.y
    float cz, opacity, r, g;   // conic.z, opacity (x antialiasing factor), colour
    float b, invdepth, rep0, rep1;
    float rep2, rep3, depth, pad;
};
static_assert(sizeof(SplatRec) == 64, "SplatRec must be 64 bytes");

// The fork's per-Gaussian statistics (forward.cu:431-432) while the forward blend accumulates them:
// the sum of the transmittances as a float, the number of tested pixels as an integer (the reference
// counts with an int atomicAdd; a float count would stop being exact at 2^24 pixels per Gaussian).
// One 8-byte slot: a workgroup updates both with two adjacent atomics.
struct __attribute__((aligned(8))) GStat {
    float sum;
    uint32_t count;
};
static_assert(sizeof(GStat) == 8, "GStat must be 8 bytes");

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int PRE_BLOCK = 256;  // Gaussians per preprocess / key-emit workgroup

struct ImageView {
    float* final_T;       // [H*W]
    uint32_t* n_contrib;  // [H*W]
    uint2* ranges;        // [tiles]
    int* tile_diff;       // [(grid_y + 1) * (grid_x + 1)] corner marks of the tile rectangles (scratch of the forward)
    size_t bytes;
};

inline ImageView image_view(void* base, size_t W, size_t H) {
    ImageView v;
    char* p = (char*)base;
    const size_t N = W * H;
    const size_t ntiles = ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off = align_up(off + n, 256); return q; };
    v.final_T = (float*)take(N * sizeof(float));
    v.n_contrib = (uint32_t*)take(N * sizeof(uint32_t));
    v.ranges = (uint2*)take(ntiles * sizeof(uint2));
    v.tile_diff = (int*)take(((W + TILE_X - 1) / TILE_X + 1) * ((H + TILE_Y - 1) / TILE_Y + 1) * sizeof(int));
    v.bytes = off;
    return v;
}

#ifndef GS_SORT_TILE
#define GS_SORT_TILE 4096
#endif
constexpr int SORT_TILE = GS_SORT_TILE;   // keys per onesweep workgroup
constexpr int SORT_MAX_PASSES = 8;
constexpr int EMIT_CHUNK = 4096;          // instances per workgroup of the key emission that partitions
constexpr int SORT_BATCH_MAX = 8;         // independent sorts that share the launches of launch_radix_sort_u32_batch
// Depth keys: the f32 bits of a view-space depth above the near plane (0.2, auxiliary.h:154-168: a point
// with depth <= 0.2 is culled) minus the bits of 0.2 stay below 2^27 while the depth is below 0.2 * 2^16
// = 13107.2 (sixteen octaves of 2^23 codes each): three 9-bit radix passes order them instead of four 8-bit
// ones over the raw bits.  A view with a Gaussian beyond that takes the four passes.
constexpr int SORT_NARROW_KEY_BITS = 27;
constexpr uint32_t DEPTH_KEY_BIAS = 0x3E4CCCCDu;                        // bits of 0.2f
constexpr uint32_t DEPTH_KEY_FAR = DEPTH_KEY_BIAS + (1u << SORT_NARROW_KEY_BITS);   // bits of 13107.2f
constexpr int SORT_BATCH_TEMP_PASSES = 6;  // status words of a batched sort: 4 passes x 256 or 3 x 512 per tile

// The key bits are dealt evenly to the passes (13 bits: 7 + 6, not 8 + 5): the fewer buckets a
// pass has, the longer the contiguous run a tile writes into each of them.
__host__ __device__ inline void digit_of_pass(int nbits, int passes, int p, int* shift, int* bits) {
    const int base = nbits / passes, extra = nbits % passes;
    *bits = base + (p < extra ? 1 : 0);
    *shift = p * base + (p < extra ? p : extra);
}

struct SortTemp {
    uint32_t* ghist;    // [SORT_MAX_PASSES][256] digit histograms, then exclusive bases
    uint32_t* ticket;   // [SORT_MAX_PASSES]
    uint32_t* err;      // [1]
    void* status;       // [passes][tiles][256] look-back words: u32 (2 flag bits + 30-bit count), or u64 when wide
    bool wide;
    size_t zero_bytes;  // bytes from ghist that must be zero before a sort
    size_t bytes;
};

// A look-back status word carries a prefix count of up to n: 30 bits beside the two flag bits of a
// 32-bit word, so sorts of 2^30 pairs and more use 64-bit words (twice the status traffic, which is
// 1/16 of a pass's data).  sort_force_wide(): test hook that selects the 64-bit words at any size.
bool sort_force_wide();
inline bool sort_status_wide(size_t n) { return n >= ((size_t)1 << 30) || sort_force_wide(); }

inline SortTemp sort_temp_view(void* base, size_t n, int passes) {
    SortTemp v;
    char* p = (char*)base;
    const size_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    size_t off = 0;
    auto take = [&](size_t nb) { char* q = p ? p + off : nullptr; off = align_up(off + nb, 256); return q; };
    v.ghist = (uint32_t*)take(SORT_MAX_PASSES * 256 * sizeof(uint32_t));
    v.ticket = (uint32_t*)take(SORT_MAX_PASSES * sizeof(uint32_t));
    v.err = (uint32_t*)take(sizeof(uint32_t));
    v.wide = sort_status_wide(n);
    v.status = take((size_t)passes * tiles * 256 * (v.wide ? sizeof(uint64_t) : sizeof(uint32_t)));
    v.zero_bytes = off;
    v.bytes = off;
    return v;
}

// Temp area of the emission that partitions and the segmented pass after it, behind the header of the plain
// sort's temp area: up to 256 status words per chunk of 2048 instances, 256 per sort tile (up to 256 segments
// each end in a partly filled tile), two words per chunk for the chunk -> Gaussian block table.
inline size_t fused_sort_temp_bytes(size_t R) {
    const size_t header = sort_temp_view(nullptr, 1, 0).bytes;
    return header + ((R / 2048 + 2) + (R / SORT_TILE + 258)) * 256 * sizeof(uint32_t) + (R / 2048 + 3) * 2 * sizeof(uint32_t) + 512;
}

struct BinView {
    uint32_t* keys[2];  // tile id of each instance, ping-pong
    uint32_t* vals[2];  // Gaussian index of each instance, ping-pong
    void* sort_temp;
    size_t bytes;
};

inline BinView bin_view(void* base, size_t R) {
    BinView v;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off = align_up(off + n, 256); return q; };
    const size_t n = R ? R : 1;
    v.keys[0] = (uint32_t*)take(n * sizeof(uint32_t));
    v.keys[1] = (uint32_t*)take(n * sizeof(uint32_t));
    v.vals[0] = (uint32_t*)take(n * sizeof(uint32_t));
    v.vals[1] = (uint32_t*)take(n * sizeof(uint32_t));
    const size_t plain = sort_temp_view(nullptr, n, SORT_MAX_PASSES).bytes, fused = fused_sort_temp_bytes(n);
    v.sort_temp = take(plain > fused ? plain : fused);
    v.bytes = off;
    return v;
}

// Pointers to camera constants that live in device memory (the caller's
// tensors); kernels read them through uniform (scalar) loads.
struct CamPtrs {
    const float* view;
    const float* proj;
    const float* campos;
};

__device__ __forceinline__ Cam load_cam(const CamPtrs& cp, int W, int H, float tanx, float tany, float mod, int D, int M,
                                        int aa) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        c.view[i] = cp.view[i];
        c.proj[i] = cp.proj[i];
    }
    c.campos[0] = cp.campos[0];
    c.campos[1] = cp.campos[1];
    c.campos[2] = cp.campos[2];
    c.W = W;
    c.H = H;
    c.tanfovx = tanx;
    c.tanfovy = tany;
    c.scale_modifier = mod;
    c.sh_degree = D;
    c.sh_coeffs = M;
    c.antialiasing = aa;
    cam_finish(&c);
    return c;
}

}  // namespace gs
