// Backward of the alpha compositing.  Reference: cuda_rasterizer/backward.cu:474-686.
//
// Same tiling as the forward (16x16 tile, 4 wave64 quadrants); the tile's list
// is walked back to front starting at the last instance any pixel of the tile
// actually blended.  The reference issues up to 14 global float atomics per
// (pixel, Gaussian) pair; here the 14 gradient components of an instance are
// summed over the wave's 64 pixels by a transposing reduction (each fold halves
// the live registers: 31 lane-exchanging instructions for all 14 sums), over
// the four waves by LDS atomics (16 lanes, 16 different words: an LDS atomic
// whose lanes share a word serialises), and
// leave the workgroup once per (tile, instance) as a 64-byte GradRec update:
// 16 lanes add to 16 consecutive floats, so one wave instruction carries four
// complete line-half atomics instead of 64 scattered ones.
#include "kernels.h"


namespace gs {

struct RenderBwdArgs {
    const uint2* ranges;
    const uint32_t* point_list;
    const SplatRec* rec;
    const float* bg;
    int W, H, grid_x, ntiles;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    const float* dL_dinvdepth;
    const float* dL_dreps;
    float* acc;  // GradRec[P] viewed as floats
    const float* reps;  // the caller's [P, DIM] tensor: channels 4 .. DIM - 1 come from there (DIM > 4)
    float* accx;        // [P][16] floats: gradient sums of the feature channels 4 .. DIM - 1 (DIM > 4)
};

__device__ __forceinline__ int xcd_remap_b(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Instances staged per round.  128 rather than 256 keeps the workgroup at 16 KiB of LDS, so the
// occupancy is set by the registers (8 waves per SIMD) and not by LDS (4).
constexpr int BWD_BATCH = 128;

// DIM: feature ("reps") channels, 0, 4, 8 or 16.  The first four are components 10 .. 13 of the GradRec;
// channels 4 .. DIM - 1 are a second group of up to 12 components with its own staged data, reduction,
// LDS sums and 64-byte record per Gaussian (accx).
template <int DIM> constexpr int bwd_waves() { return DIM > 8 ? 5 : DIM > 4 ? 6 : 8; }

template <int DIM, bool INVD>
// 8 waves per SIMD: the register allocator gets 64 VGPRs (it takes 66 when left alone, which costs a wave).
// The wide-row instantiations are bound by the latency of a visit's chain of dependent instructions, so they
// are compiled for the most waves their registers allow without spilling: 6 with 8 channels (80 VGPRs), 5 with
// 16 (88 VGPRs; 30.8 KB of LDS allow no more).  Left alone the allocator takes 84 and 102 (4 waves): 16
// channels 0.770 -> 0.736 ms, 8 channels 0.732 -> 0.69 ms on the street scene.
__global__ void __launch_bounds__(TILE_PIX) __attribute__((amdgpu_waves_per_eu(bwd_waves<DIM>(), bwd_waves<DIM>())))
render_bwd_kernel(const RenderBwdArgs a) {
    constexpr int XQ = DIM > 4 ? (DIM - 4) / 4 : 0;   // extra 16-byte pieces of a feature row
    // same regrouping of the SplatRec as in the forward kernel
    __shared__ float4 s_q[4][BWD_BATCH];
    float4* const s_q0 = s_q[0];  // x, y, -conic.x / 2, -conic.z / 2 (see splat_power_prescaled)
    float4* const s_q1 = s_q[1];  // conic.y, tau, opacity, Gaussian index (bits)
    float4* const s_q2 = s_q[2];  // r, g, b, 1/depth
    float4* const s_q3 = s_q[3];  // rep0 .. rep3
    __shared__ float s_acc[BWD_BATCH][16];
    __shared__ float4 s_qx[XQ ? XQ : 1][XQ ? BWD_BATCH : 1];   // rep4 .. rep(DIM - 1)
    __shared__ float s_accx[XQ ? BWD_BATCH : 1][16];
    __shared__ uint32_t s_max[4];
    // s_reach[q][h]: bit i = entry 64 h + i of the round can give a gradient in quadrant q: one of q's pixels
    // still had it in its list, and its alpha >= 1/255 ellipse reaches the quadrant (quadrant_reach_bits)
    __shared__ uint64_t s_reach[4][BWD_BATCH / 64];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = xcd_remap_b(blockIdx.x, a.ntiles);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const f2 pix = {(float)px, (float)py};
    const uint2 range = a.ranges[tile];
    const size_t pid = (size_t)py * a.W + px;
    const size_t HW = (size_t)a.H * a.W;

    const float T_final = inside ? a.final_T[pid] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? a.n_contrib[pid] : 0u;

    // dL/d(pixel) in the pairing of s_q2 / s_q3: (r, g), (b, 1/depth), reps
    f2 D01 = {0.f, 0.f}, D2i = {0.f, 0.f}, E01 = {0.f, 0.f}, E23 = {0.f, 0.f};
    f2 EX[XQ ? 2 * XQ : 1];   // dL/d(pixel) of the feature channels 4 .. DIM - 1, in pairs
#pragma unroll
    for (int e = 0; e < (XQ ? 2 * XQ : 1); ++e) EX[e] = (f2){0.f, 0.f};
    if (inside) {
        D01.x = a.dL_dpix[pid];
        D01.y = a.dL_dpix[HW + pid];
        D2i.x = a.dL_dpix[2 * HW + pid];
        if (INVD) D2i.y = a.dL_dinvdepth[pid];
        if (DIM >= 4) {
            E01.x = a.dL_dreps[pid];
            E01.y = a.dL_dreps[HW + pid];
            E23.x = a.dL_dreps[2 * HW + pid];
            E23.y = a.dL_dreps[3 * HW + pid];
        }
#pragma unroll
        for (int e = 0; e < 2 * XQ; ++e) {
            EX[e].x = a.dL_dreps[(size_t)(4 + 2 * e) * HW + pid];
            EX[e].y = a.dL_dreps[(size_t)(5 + 2 * e) * HW + pid];
        }
    }
    float bg_dot_dpixel = 0;
    bg_dot_dpixel += a.bg[0] * D01.x;
    bg_dot_dpixel += a.bg[1] * D01.y;
    bg_dot_dpixel += a.bg[2] * D2i.x;
    // backward.cu:664-667: the background's share of dL/dalpha is -T_final / (1 - alpha) * (bg . dL/dpixel);
    // the part that does not depend on the Gaussian is formed once per pixel
    const float bg_term = -T_final * bg_dot_dpixel;

    float accum_S = 0.f, last_S = 0.f, last_alpha = 0.f;

    // constant factor of gradient component (tid & 15), applied once per (tile, instance) sum:
    // ddelx_dx = W/2 and ddely_dy = H/2 (backward.cu:568-569) and the -0.5 of the conic terms
    const int acc_slot = wave_transpose_slot(lane);   // the component whose wave sum this lane ends up with
    const int comp_id = tid & 15;
    const float scale = comp_id == 0 ? -0.5f * a.W : comp_id == 1 ? -0.5f * a.H : comp_id <= 4 ? -0.5f : 1.0f;

    // number of list entries any pixel of this tile needs
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane == 0) s_max[w] = m;
    }
    __syncthreads();
    const int max_last = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

    for (int base = 0; base < max_last; base += BWD_BATCH) {
        __syncthreads();
        const int nb = min(BWD_BATCH, max_last - base);
        uint32_t reach = 0;
        if (tid < nb) {
            const uint32_t contributor = (uint32_t)(max_last - 1 - base - tid);
            const uint32_t id = a.point_list[range.x + contributor];
            const float4* r4 = reinterpret_cast<const float4*>(a.rec + id);
            const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
            s_q0[tid] = make_float4(r0.x, r0.y, -0.5f * r0.z, -0.5f * r1.x);
            // tau as in the forward kernel: power below it cannot reach alpha >= 1/255; never below -87
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
            reach = quadrant_reach_bits(r0.x, r0.y, r0.z, r0.w, r1.x, tau, (float)(tx * TILE_X), (float)(ty * TILE_Y));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (contributor >= s_max[q]) reach &= ~(1u << q);   // behind the last entry any pixel of q blended
        }
        if (w < BWD_BATCH / 64) {   // the staging waves: whole waves take part in the ballots
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t bq = lanes_where((reach >> q) & 1u);
                if (lane == 0) s_reach[q][w] = bq;
            }
        }
        {
            float4* z = reinterpret_cast<float4*>(&s_acc[0][0]) + tid;  // 128 x 16 floats = 512 float4
            z[0] = z[TILE_PIX] = make_float4(0, 0, 0, 0);
            if (XQ) {
                float4* zx = reinterpret_cast<float4*>(&s_accx[0][0]) + tid;
                zx[0] = zx[TILE_PIX] = make_float4(0, 0, 0, 0);
            }
        }
        __syncthreads();

        typedef __attribute__((address_space(3))) float4 lds_f4;
        const uint32_t qbase = (uint32_t)(uintptr_t)(lds_f4*)&s_q[0][0];
        // The wave visits the entries that can give its quadrant a gradient, in list order (the others
        // would leave every pixel's state as it is and add exact zeros).
#pragma unroll 1
        for (int half = 0; half < BWD_BATCH / 64; ++half) {
          const uint64_t mine = s_reach[w][half];
          uint64_t todo = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(mine >> 32)) << 32) |
                          __builtin_amdgcn_readfirstlane((uint32_t)mine);
#pragma unroll 1
          while (todo) {
            const int j = half * 64 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t contributor = (uint32_t)(max_last - 1 - base - j);
            // the LDS address of entry j in a VGPR; all four reads of a visit use it with immediate offsets
            // (the compiler would copy it from an SGPR before each)
            uint32_t qa = qbase + 16u * (uint32_t)j;
            asm volatile("" : "+v"(qa));
            float4 q0, q1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1)
                         : "v"(qa), "i"((int)(BWD_BATCH * sizeof(float4)))
                         : "memory");
            f2 d;
            const float power = splat_power_prescaled((f2){q0.x, q0.y}, pix, (f2){q0.z, q0.w}, q1.x, &d);
            const float dx = d.x, dy = d.y;
            const V4 co = {q0.z, q1.x, q0.w, q1.z};  // -conic.x / 2, conic.y, -conic.z / 2, opacity (.y and .w are used)
            // backward.cu:592-611 as lane masks; pixels outside the image have last_contributor 0
            const uint64_t hm = lanes_where(contributor < last_contributor) & lanes_where(!(power > 0.0f));
            const uint64_t pm = lanes_where(power >= q1.y) & hm;
            if (pm == 0) continue;  // nobody can pass the alpha test
            // The hardware exponential (v_exp_f32) instead of the polynomial, which takes three times as
            // long.  With the scaling of its argument it is within 4e-7 relative of gs_expf where a
            // passing pixel of a Gaussian with opacity <= 1 can be (power >= -log(255)), and within 6e-6
            // anywhere on [-87, 0].  Which pixels pass the alpha test must not depend on that: the forward
            // decided it with gs_expf, and a pixel taken here but not there would shift its transmittance
            // for the rest of its list.  A wave in which some pixel's alpha lies within 1e-5 relative of
            // the threshold repeats the visit's exponential exactly.
#ifdef GS_BWD_EXACT_EXP   // (diagnostic build: the forward's polynomial everywhere; see also GS_BWD_EXACT_DIV)
            float G = gs_expf_blend(power);
#else
            float G = __builtin_amdgcn_exp2f(power * 1.44269504088896341f);  // lanes outside pm compute values nobody reads
#endif
            float alpha = __builtin_fminf(GS_ALPHA_MAX, co.w * G);
            uint64_t vm = lanes_where(alpha >= GS_ALPHA_MIN * (1.0f + 1e-5f)) & pm;
            if ((lanes_where(alpha >= GS_ALPHA_MIN * (1.0f - 1e-5f)) & pm) != vm) {
                G = gs_expf_blend(power);
                alpha = __builtin_fminf(GS_ALPHA_MAX, co.w * G);
                vm = lanes_where(!(alpha < GS_ALPHA_MIN)) & pm;
            }
            if (vm == 0) continue;

            float4 q2, q3 = make_float4(0, 0, 0, 0);
            if (DIM >= 4)
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(q2), "=&v"(q3)
                             : "v"(qa), "i"(2 * (int)(BWD_BATCH * sizeof(float4))), "i"(3 * (int)(BWD_BATCH * sizeof(float4)))
                             : "memory");
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(q2)
                             : "v"(qa), "i"(2 * (int)(BWD_BATCH * sizeof(float4)))
                             : "memory");

            // No branch around the arithmetic: a pixel outside vm takes part as a contributor with
            // alpha = 0 and G = 0.  Its transmittance stays as it is (T / (1 - 0), exactly), all 14
            // gradient terms come out as exact zeros (each is a product with alpha or G), and the
            // recurrence below only moves the update it would have made at its next real contributor
            // forward: accum_S takes the value it would take there, and with last_alpha = 0 every later
            // step until then is accum_S + 0 * (...) = accum_S.  Same results bit for bit as skipping
            // the pixel, without the 14 registers that would have to be zeroed for the skipped lanes
            // before the reduction (a wave instruction costs the same with 1 or 64 lanes enabled).
            const float alpha_e = select_or_zero(vm, alpha);
            const float G_e = select_or_zero(vm, G);
            // 1 / (1 - alpha): v_rcp_f32 plus one Newton step (two fused multiply-adds).  The bare
            // v_rcp_f32 (1 ulp, not unbiased) is not enough: T is recovered by one such division
            // per blended Gaussian, its error grows along the list, and in gradients that are small
            // sums of large terms it showed up as 3e-4 relative against a divide; with the step
            // the results match an IEEE divide (ten instructions) within the 1e-4 bar element-wise.
            const float oma = 1.f - alpha_e;
#ifdef GS_BWD_EXACT_DIV
            float inv_1ma = 1.0f / oma;
            T = T / oma;
#else
            float inv_1ma = __builtin_amdgcn_rcpf(oma);
            inv_1ma = fmaf(fmaf(-oma, inv_1ma, 1.0f), inv_1ma, inv_1ma);
            T = T * inv_1ma;
#endif
            const float dchannel_dcolor = alpha_e * T;
            const f2 dc2 = {dchannel_dcolor, dchannel_dcolor};
            // The reference keeps, per channel, accum = last_alpha * last_c + (1 - last_alpha) * accum
            // and adds (c - accum) * dL_dpixel over the channels (backward.cu:618-660).  Only the
            // sum is used, and it obeys the same recurrence with S = sum_ch c * dL_dpixel in the
            // place of c: two scalars of state per pixel instead of two per channel.  S is summed
            // two channels at a time.
            f2 S2 = (f2){q2.x, q2.y} * D01;
            S2 = __builtin_elementwise_fma((f2){q2.z, q2.w}, D2i, S2);
            const f2 g_rg = dc2 * D01;
            const f2 g_bd = dc2 * D2i;
            f2 g_r01 = {0.f, 0.f}, g_r23 = {0.f, 0.f};
            if (DIM >= 4) {
                S2 = __builtin_elementwise_fma((f2){q3.x, q3.y}, E01, S2);
                S2 = __builtin_elementwise_fma((f2){q3.z, q3.w}, E23, S2);
                g_r01 = dc2 * E01;
                g_r23 = dc2 * E23;
            }
            float compx[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) compx[e] = 0.f;
#pragma unroll
            for (int e = 0; e < XQ; ++e) {
                const float4 x = s_qx[e][j];
                S2 = __builtin_elementwise_fma((f2){x.x, x.y}, EX[2 * e], S2);
                S2 = __builtin_elementwise_fma((f2){x.z, x.w}, EX[2 * e + 1], S2);
                const f2 ga = dc2 * EX[2 * e], gb = dc2 * EX[2 * e + 1];
                compx[4 * e] = ga.x;
                compx[4 * e + 1] = ga.y;
                compx[4 * e + 2] = gb.x;
                compx[4 * e + 3] = gb.y;
            }
            const float S = S2.x + S2.y;
            accum_S = fmaf(last_alpha, last_S - accum_S, accum_S);
            last_S = S;
            last_alpha = alpha_e;
            const float dL_dalpha = fmaf(bg_term, inv_1ma, (S - accum_S) * T);

            // backward.cu:669-684 with the constant factors taken out of the per-pixel terms: with
            // GL = G * dL/dG (= opacity * G * dL/dalpha), h = GL * dx, k = GL * dy the reference's
            //   dL/dmean2D = -(h * conic.x + k * conic.y) * W/2,  -(k * conic.z + h * conic.y) * H/2
            //   dL/dconic  = -0.5 * (h * dx, h * dy, k * dy)
            // The conic is the same for every pixel, so h and k are what is summed over the pixels;
            // the combination with the conic and the factors -W/2, -H/2, -0.5 are applied to the sums
            // when they leave LDS.
            const float g_op = G_e * dL_dalpha;
            const float GL = co.w * g_op;
            const float h = GL * dx, k = GL * dy;
            const float g_mx = h;
            const float g_my = k;
            const float g_cx = h * dx;
            const float g_cy = h * dy;
            const float g_cw = k * dy;

            // 16 gradient components x 64 pixels -> 16 sums by the transposing reduction over the
            // wave; one lane of every quad then adds its sum to LDS, 16 lanes on 16 different words
            // (the four waves of the tile meet there).
            const float comp[16] = {g_mx, g_my, g_cx, g_cy, g_cw, g_op, g_rg.x, g_rg.y, g_bd.x, g_bd.y,
                                    g_r01.x, g_r01.y, g_r23.x, g_r23.y, 0.f, 0.f};
            const float tot = wave_transpose_sum16<14>(comp);
            if ((lane & 3) == 0) atomicAdd(&s_acc[j][acc_slot], tot);
            if (XQ) {
                const float totx = wave_transpose_sum16<16>(compx);
                if ((lane & 3) == 0) atomicAdd(&s_accx[j][acc_slot], totx);
            }
          }
        }
        __syncthreads();
        // 16 lanes per instance: one 64-byte GradRec update
#pragma unroll 1
        for (int h = 0; h < BWD_BATCH / 16; ++h) {
            const int j = (tid >> 4) + 16 * h;
            if (j >= nb) break;
            float v = s_acc[j][comp_id];
            if (comp_id < 2) {   // sums of h and k -> dL/dmean2D
                const float other = s_acc[j][comp_id ^ 1];
                const float4 q0 = s_q0[j];
                v = -2.0f * (comp_id == 0 ? q0.z : q0.w) * v + s_q1[j].x * other;   // q0 holds -conic / 2
            }
            v *= scale;
            if (v != 0.0f) atomicAdd(a.acc + (size_t)__float_as_uint(s_q1[j].w) * 16 + (tid & 15), v);
            if (XQ) {
                const float vx = s_accx[j][comp_id];
                if (vx != 0.0f) atomicAdd(a.accx + (size_t)__float_as_uint(s_q1[j].w) * 16 + (tid & 15), vx);
            }
        }
    }
}

hipError_t launch_render_bwd(const GsrScene& s, const GeomBuf& g, const ImageView& im, const uint32_t* point_list,
                             const GsrBackwardIO& io, GradRec* acc, hipStream_t st) {
    RenderBwdArgs a;
    a.ranges = im.ranges;
    a.point_list = point_list;
    a.rec = g.rec;
    a.bg = s.bg;
    a.W = s.W;
    a.H = s.H;
    a.grid_x = (s.W + TILE_X - 1) / TILE_X;
    a.ntiles = a.grid_x * ((s.H + TILE_Y - 1) / TILE_Y);
    a.final_T = im.final_T;
    a.n_contrib = im.n_contrib;
    a.dL_dpix = io.dL_dout_color;
    a.dL_dinvdepth = io.dL_dout_invdepth;
    a.dL_dreps = io.dL_dout_reps;
    a.acc = reinterpret_cast<float*>(acc);
    a.reps = s.reps;
    a.accx = reinterpret_cast<float*>(grad_rec_extra(acc, (size_t)s.P));
    const dim3 grid(a.ntiles), block(TILE_PIX);
    const bool invd = io.dL_dout_invdepth != nullptr;
    if (s.DIM == 4) {
        if (invd) hipLaunchKernelGGL((render_bwd_kernel<4, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((render_bwd_kernel<4, false>), grid, block, 0, st, a);
    } else if (s.DIM == 8) {
        if (invd) hipLaunchKernelGGL((render_bwd_kernel<8, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((render_bwd_kernel<8, false>), grid, block, 0, st, a);
    } else if (s.DIM == 16) {
        if (invd) hipLaunchKernelGGL((render_bwd_kernel<16, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((render_bwd_kernel<16, false>), grid, block, 0, st, a);
    } else {
        if (invd) hipLaunchKernelGGL((render_bwd_kernel<0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((render_bwd_kernel<0, false>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace gs
