// Forward alpha compositing, one 16x16 tile per 256-thread workgroup (4
// wave64 wavefronts, each owning an 8x8 pixel quadrant).  Reference:
// cuda_rasterizer/forward.cu:313-490.
//
// Per round the workgroup gathers the 64-byte SplatRec of 256 instances into
// LDS; every wave then walks the batch with wave-uniform (broadcast) LDS
// reads.  A wave whose 64 pixels are all saturated stops computing (ballot);
// the workgroup stops when its four waves have.  The per-Gaussian statistics
// of the fork (sum of transmittance and hit count, forward.cu:431-432) are
// reduced over the wave 16 instances at a time by a transposing reduction,
// over the four waves through LDS, and leave the workgroup as one pair of
// adjacent float atomics per instance.
#include "kernels.h"


namespace gs {

struct RenderFwdArgs {
    const uint2* ranges;
    const uint32_t* point_list;
    const SplatRec* rec;
    GStat* gstat;
    const float* bg;
    int W, H, grid_x, ntiles;
    float* final_T;
    uint32_t* n_contrib;
    float* out_color;
    float* out_invdepth;
    float* accum_alpha;
    int* last_contr;
    float* out_reps;
    const float* reps;   // the caller's [P, DIM] tensor: channels 4 .. DIM - 1 come from there (DIM > 4)
};

// Consecutive tile ids land on one XCD (workgroups are dealt round-robin over
// the 8 XCDs), so neighbouring tiles share that XCD's L2 for SplatRec gathers.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// REACH: entries whose alpha >= 1/255 ellipse does not reach a wave's quadrant are not tested by that
// wave (quadrant_reach_bits).  Pays when the splats are small against a tile; costs a scalar test and
// two scalar copies per visit, which large splats (that reach every quadrant) do not earn back.
// DIM: feature ("reps") channels, 0, 4, 8 or 16 (the reference fixes them at compile time,
// cuda_rasterizer/config.h:16).  The first four travel in the SplatRec; wider rows add (DIM - 4) / 4
// staged 16-byte pieces per entry, fetched from the caller's tensor, and as many accumulators per pixel
// (those instantiations are compiled for 5 waves per SIMD: 92-96 registers with 8 channels; with 16 the
// allocator cannot get there, takes 116-128 and runs 4 waves, but schedules better than when asked for 4:
// 0.416 -> 0.400 ms.  Hence the diagnostic about the missed target is turned off for this kernel.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int DIM, bool INVD, bool REACH>
__global__ void __launch_bounds__(TILE_PIX) __attribute__((amdgpu_waves_per_eu(DIM > 4 ? 5 : 6, DIM > 4 ? 5 : 6)))
render_fwd_kernel(const RenderFwdArgs a) {
    constexpr int XQ = DIM > 4 ? (DIM - 4) / 4 : 0;   // extra 16-byte pieces of a feature row
    // One SplatRec per instance, regrouped: the test of an instance needs the first two
    // quarters only, the other two feed packed (two-float) multiply-adds as they are.
    __shared__ float4 s_q[4][TILE_PIX];
    float4* const s_q0 = s_q[0];  // x, y, -conic.x / 2, -conic.z / 2 (see splat_power_prescaled)
    float4* const s_q1 = s_q[1];  // conic.y, tau, opacity, Gaussian index (bits)
    float4* const s_q2 = s_q[2];  // r, g, b, 1/depth
    float4* const s_q3 = s_q[3];  // rep0 .. rep3
    __shared__ float4 s_qx[XQ ? XQ : 1][XQ ? TILE_PIX : 1];   // rep4 .. rep(DIM - 1)
    __shared__ float s_gsum[4][TILE_PIX];
    __shared__ uint32_t s_gcnt[4][TILE_PIX];
    __shared__ int s_wdone[4];
    // s_reach[q][g]: bit u = entry 16 g + u of the round reaches quadrant q (quadrant_reach_bits); without
    // the bit no pixel of q can pass the alpha test and every pixel of q has a non-positive power
    __shared__ uint32_t s_reach[4][TILE_PIX / 16];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const f2 pix = {(float)px, (float)py};
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    // Per-lane flags live in wave-uniform 64-bit masks (see lanes_where / lane_in).
    uint64_t live = lanes_where(inside);  // pixels that still accept contributions
    float T = 1.0f;
    f2 Crg = {0.f, 0.f}, Cbd = {0.f, 0.f}, R01 = {0.f, 0.f}, R23 = {0.f, 0.f};  // (r, g), (b, 1/depth), reps
    f2 RX[XQ ? 2 * XQ : 1];                                                      // reps 4 .. DIM - 1, in pairs
#pragma unroll
    for (int e = 0; e < (XQ ? 2 * XQ : 1); ++e) RX[e] = (f2){0.f, 0.f};
    uint32_t last = 0;
    int last_g = -1;

    if (tid < 4) s_wdone[tid] = 0;
    const int slot = wave_transpose_slot(lane);   // which of a group's 16 sums this lane ends up with

    for (int base = 0; base < total; base += TILE_PIX) {
        __syncthreads();
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int nb = min(TILE_PIX, total - base);
        uint32_t reach = 0xFu;   // the padding entries below are visited (their power is positive everywhere)
        if (tid < nb) {
            const uint32_t id = a.point_list[range.x + base + tid];
            const float4* r4 = reinterpret_cast<const float4*>(a.rec + id);
            const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
            s_q0[tid] = make_float4(r0.x, r0.y, -0.5f * r0.z, -0.5f * r1.x);
            // tau: the power below which alpha = opacity * exp(power) < 1/255 for certain, i.e.
            // -log(255 * opacity) less a margin that covers the rounding of the fast logarithm and of
            // the exponential by orders of magnitude.  Never below -87, where gs_expf stops falling:
            // the exponential of a pair that passes this test needs no clamp.
            const float tau = fmaxf(-__logf(255.0f * r1.y) - 0.01f, -87.0f);
            s_q1[tid] = make_float4(r0.w, tau, r1.y, __uint_as_float(id));
            s_q2[tid] = make_float4(r1.z, r1.w, r2.x, r2.y);
            if (DIM > 2) {
                const float4 r3 = r4[3];
                s_q3[tid] = make_float4(r2.z, r2.w, r3.x, r3.y);
            }
            if (XQ) {
                const float4* rx = reinterpret_cast<const float4*>(a.reps + (size_t)id * DIM) + 1;
#pragma unroll
                for (int e = 0; e < XQ; ++e) s_qx[e][tid] = rx[e];
            }
            if (REACH)
                reach = quadrant_reach_bits(r0.x, r0.y, r0.z, r0.w, r1.x, tau, (float)(tx * TILE_X), (float)(ty * TILE_Y));
        } else if (tid < ((nb + 15) & ~15)) {
            // the last group of 16 is filled up with entries whose power is positive everywhere
            s_q0[tid] = make_float4(-1e4f, -1e4f, 0.5f, 0.5f);
            s_q1[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (REACH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // this wave staged the entries 64 w .. 64 w + 63: four groups of 16
                const uint64_t bq = lanes_where((reach >> q) & 1u);
                if (lane < 4) s_reach[q][4 * w + lane] = (uint32_t)(bq >> (16 * lane)) & 0xffffu;
            }
        }
        __syncthreads();

        // Instances in groups of 16: the transmittance a lane contributes to each instance's
        // statistic waits in a register and one transposing reduction per group sums all 16 over
        // the wave (about 2 lane exchanges per instance instead of 6).
#pragma unroll 1
        for (int j0 = 0; j0 < nb; j0 += 16) {
            if (live == 0) {  // nothing left for this wave: its share of the remaining statistics is zero
                if (lane < 16) {
                    s_gsum[w][j0 + lane] = 0.f;
                    s_gcnt[w][j0 + lane] = 0u;
                }
                continue;
            }
            // LDS address of the group's first entry, pinned in a VGPR: the broadcast reads of the 16
            // visits use it with immediate offsets (left to itself the compiler keeps the address in an
            // SGPR and copies it into a VGPR, twice, in front of every visit's reads)
            typedef __attribute__((address_space(3))) float4 lds_f4;
            uint32_t gbase = (uint32_t)(uintptr_t)(lds_f4*)&s_q[0][j0];
            asm volatile("" : "+v"(gbase));
            float tv[16];
            uint64_t cnt_lo = 0, cnt_hi = 0;  // byte u: number of pixels of the wave that tested instance j0 + u
#pragma unroll
            for (int u = 0; u < 16; ++u) tv[u] = 0.f;
            const uint32_t reached = REACH ? __builtin_amdgcn_readfirstlane(s_reach[w][j0 >> 4]) : 0xffffu;
            uint64_t none = 0;   // "no pixel can pass", opaque to the compiler (it would split the shared part again)
            asm volatile("" : "+s"(none));
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + u;
                // An entry out of this quadrant's reach is not read: every live pixel tests it (power <= 0),
                // none can blend it.  Only the test is branched around; what follows is shared, so that
                // the pixels' state does not meet itself at a join (the register allocator then copies it).
                uint64_t hm = live, pm = none;
                float4 q0, q1;
                float power = 0.f;
                if (!REACH || ((reached >> u) & 1u)) {
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q0), "=&v"(q1)
                                 : "v"(gbase), "i"(16 * u), "i"((int)(TILE_PIX * sizeof(float4)) + 16 * u)
                                 : "memory");
                    f2 d;
                    power = splat_power_prescaled((f2){q0.x, q0.y}, pix, (f2){q0.z, q0.w}, q1.x, &d);
                    // forward.cu:423-432: the statistics count every pixel that reaches this instance with a
                    // non-positive power, whatever its alpha
                    hm = lanes_where(!(power > 0.0f)) & live;
                    pm = lanes_where(power >= q1.y) & hm;
                }
                tv[u] = select_or_zero(hm, T);
                if (u < 8) cnt_lo |= (uint64_t)__popcll(hm) << (8 * u);
                else cnt_hi |= (uint64_t)__popcll(hm) << (8 * (u - 8));
                if (pm == 0) continue;  // no pixel of the wave can pass the alpha test

                // lanes outside pm compute values nobody reads
                const float alpha = __builtin_fminf(GS_ALPHA_MAX, q1.z * gs_expf_blend(power));
                const uint64_t am = lanes_where(!(alpha < GS_ALPHA_MIN)) & pm;
                if (am == 0) continue;
                const float test_T = T * (1 - alpha);
                const uint64_t tm = lanes_where(test_T < GS_T_MIN) & am;  // these pixels are finished
                live &= ~tm;
                const uint64_t bm = am & ~tm;
                if (bm == 0) continue;
                float4 q2, q3 = make_float4(0, 0, 0, 0);
                if (DIM >= 4)
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q2), "=&v"(q3)
                                 : "v"(gbase), "i"(2 * (int)(TILE_PIX * sizeof(float4)) + 16 * u),
                                   "i"(3 * (int)(TILE_PIX * sizeof(float4)) + 16 * u)
                                 : "memory");
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q2)
                                 : "v"(gbase), "i"(2 * (int)(TILE_PIX * sizeof(float4)) + 16 * u)
                                 : "memory");
                if (lane_in(bm)) {
                    // forward.cu:454-464: C += c * alpha * T per channel.  The weight alpha * T is formed
                    // once and each pair of channels is one packed fused multiply-add (the sums are
                    // compared at 1e-5 absolute, no integer output depends on them): 5 instructions where
                    // the reference's order of operations takes 12.
                    const float wgt = alpha * T;
                    const f2 w2 = {wgt, wgt};
                    Crg = __builtin_elementwise_fma((f2){q2.x, q2.y}, w2, Crg);
                    Cbd = __builtin_elementwise_fma((f2){q2.z, q2.w}, w2, Cbd);
                    if (DIM >= 4) {
                        R01 = __builtin_elementwise_fma((f2){q3.x, q3.y}, w2, R01);
                        R23 = __builtin_elementwise_fma((f2){q3.z, q3.w}, w2, R23);
                    }
#pragma unroll
                    for (int e = 0; e < XQ; ++e) {
                        const float4 x = s_qx[e][j];
                        RX[2 * e] = __builtin_elementwise_fma((f2){x.x, x.y}, w2, RX[2 * e]);
                        RX[2 * e + 1] = __builtin_elementwise_fma((f2){x.z, x.w}, w2, RX[2 * e + 1]);
                    }
                    T = test_T;
                    last = (uint32_t)(base + j + 1);
                    last_g = (int)__float_as_uint(q1.w);
                }
            }
            const float tot = wave_transpose_sum16(tv);
            if ((lane & 3) == 0) {   // one lane of every quad: the sum of instance j0 + slot
                s_gsum[w][j0 + slot] = tot;
                s_gcnt[w][j0 + slot] = (uint32_t)((((slot & 8) ? cnt_hi : cnt_lo) >> ((slot & 7) * 8)) & 0xffu);
            }
        }
        if (lane == 0) s_wdone[w] = live == 0 ? 1 : 0;
        __syncthreads();
        // thread t: component (t & 1) of instance (t >> 1) + 128 h
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = (tid >> 1) + 128 * h;
            if (j < nb) {
                const uint32_t c = s_gcnt[0][j] + s_gcnt[1][j] + s_gcnt[2][j] + s_gcnt[3][j];
                if (c != 0u) {
                    GStat* dst = a.gstat + __float_as_uint(s_q1[j].w);
                    if (tid & 1) atomicAdd(&dst->count, c);   // integer, as the reference's int atomicAdd
                    else atomicAdd(&dst->sum, s_gsum[0][j] + s_gsum[1][j] + s_gsum[2][j] + s_gsum[3][j]);
                }
            }
        }
    }

    if (inside) {
        const size_t pid = (size_t)py * a.W + px;
        const size_t HW = (size_t)a.H * a.W;
        a.final_T[pid] = T;
        a.accum_alpha[pid] = T;
        a.n_contrib[pid] = last;
        a.last_contr[pid] = last_g;
        a.out_color[pid] = Crg.x + T * a.bg[0];
        a.out_color[HW + pid] = Crg.y + T * a.bg[1];
        a.out_color[2 * HW + pid] = Cbd.x + T * a.bg[2];
        if (DIM >= 4) {
            a.out_reps[pid] = R01.x;
            a.out_reps[HW + pid] = R01.y;
            a.out_reps[2 * HW + pid] = R23.x;
            a.out_reps[3 * HW + pid] = R23.y;
        }
#pragma unroll
        for (int e = 0; e < 2 * XQ; ++e) {
            a.out_reps[(size_t)(4 + 2 * e) * HW + pid] = RX[e].x;
            a.out_reps[(size_t)(5 + 2 * e) * HW + pid] = RX[e].y;
        }
        if (INVD) a.out_invdepth[pid] = Cbd.y;
    }
}
#pragma clang diagnostic pop

static thread_local int t_reach_masks = -1;   // test hook, per calling thread
void render_fwd_set_reach_masks(int mode) { t_reach_masks = mode < 0 ? -1 : (mode ? 1 : 0); }

hipError_t launch_render_fwd(const GsrScene& s, const GsrForwardOut& o, const GeomBuf& g, const ImageView& im,
                             const uint32_t* point_list, int64_t num_rendered, int64_t num_visible, hipStream_t st) {
    RenderFwdArgs a;
    a.ranges = im.ranges;
    a.point_list = point_list;
    a.rec = g.rec;
    a.gstat = g.gstat;
    a.bg = s.bg;
    a.W = s.W;
    a.H = s.H;
    a.grid_x = (s.W + TILE_X - 1) / TILE_X;
    a.ntiles = a.grid_x * ((s.H + TILE_Y - 1) / TILE_Y);
    a.final_T = im.final_T;
    a.n_contrib = im.n_contrib;
    a.out_color = o.out_color;
    a.out_invdepth = o.out_invdepth;
    a.accum_alpha = o.accum_alpha;
    a.last_contr = o.last_contr_gauss;
    a.out_reps = o.out_reps;
    a.reps = s.reps;
    const dim3 grid(a.ntiles), block(TILE_PIX);
    const bool invd = o.out_invdepth != nullptr;
    // small splats (fewer than six tiles per rendered Gaussian on average): the quadrant reach masks pay
    const int64_t nvis = num_visible > 0 ? num_visible : s.P;
    const int forced = t_reach_masks;
    const bool reach = forced < 0 ? num_rendered < 6 * nvis : forced != 0;
#define GS_LAUNCH_FWD(D, I, R) hipLaunchKernelGGL((render_fwd_kernel<D, I, R>), grid, block, 0, st, a)
    if (s.DIM == 4) {
        if (invd) { if (reach) GS_LAUNCH_FWD(4, true, true); else GS_LAUNCH_FWD(4, true, false); }
        else { if (reach) GS_LAUNCH_FWD(4, false, true); else GS_LAUNCH_FWD(4, false, false); }
    } else if (s.DIM == 8) {
        if (invd) { if (reach) GS_LAUNCH_FWD(8, true, true); else GS_LAUNCH_FWD(8, true, false); }
        else { if (reach) GS_LAUNCH_FWD(8, false, true); else GS_LAUNCH_FWD(8, false, false); }
    } else if (s.DIM == 16) {
        if (invd) { if (reach) GS_LAUNCH_FWD(16, true, true); else GS_LAUNCH_FWD(16, true, false); }
        else { if (reach) GS_LAUNCH_FWD(16, false, true); else GS_LAUNCH_FWD(16, false, false); }
    } else {
        if (invd) { if (reach) GS_LAUNCH_FWD(0, true, true); else GS_LAUNCH_FWD(0, true, false); }
        else { if (reach) GS_LAUNCH_FWD(0, false, true); else GS_LAUNCH_FWD(0, false, false); }
    }
#undef GS_LAUNCH_FWD
    return hipGetLastError();
}

}  // namespace gs
