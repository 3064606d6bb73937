// What would a splat-parallel backward blend cost per 64 (pixel, entry) pairs?  Development tool.
//
// One lane = one list entry of a bucket of 64 consecutive entries, the wave steps through the pixels of its
// 8x8 quadrant, every lane keeps private sums of the 14 gradient components of its entry.  Per pixel a lane
// needs the transmittance in front of its entry (checkpoint * product of (1 - alpha) over the earlier lanes:
// an exclusive prefix product over the wave) and the recurrence behind its entry
// (accum = alpha * S + (1 - alpha) * accum, back to front: a suffix scan of affine maps).  This kernel is that
// inner loop on synthetic data, checked against a sequential evaluation; the instruction count of the loop
// body is read off the compiled code (llvm-objdump) and the time per step is measured.
//
// build: make -C tools/probe splat_parallel_probe     run on the GPU box: tools/probe/splat_parallel_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int NPIX = 64;       // pixels of a quadrant
constexpr int NCH = 8;         // colour (3) + inverse depth (1) + 4 feature channels

struct Entry {   // one list entry, one per lane
    float x, y, ca, cb, cc, op;
    float c[NCH];
};
struct Pixel {   // per-pixel data, wave-uniform in a step
    float px, py, T0, bgterm;
    float d[NCH];   // dL/d(pixel)
};

template <int CTRL>
__device__ __forceinline__ float dpp_or(float old, float v) {   // lanes the pattern does not reach get `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, 0xf, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_rows(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROWS, 0xf, false));
}

// inclusive scans over the 64 lanes: product, and composition of affine maps x -> m x + b (later lane after earlier)
__device__ __forceinline__ float scan_mul(float v) {
    v *= dpp_or<0x111>(1.f, v);
    v *= dpp_or<0x112>(1.f, v);
    v *= dpp_or<0x114>(1.f, v);
    v *= dpp_or<0x118>(1.f, v);
    v *= dpp_rows<0x142, 0xa>(1.f, v);   // row_bcast:15 into rows 1 and 3
    v *= dpp_rows<0x143, 0xc>(1.f, v);   // row_bcast:31 into rows 2 and 3
    return v;
}
// (m, b) of lane l becomes the composition f_l o f_(l-1) o ... o f_0
__device__ __forceinline__ void scan_affine(float& m, float& b) {
#define STEP(CTRL)                                      \
    {                                                   \
        const float pm = dpp_or<CTRL>(1.f, m);          \
        const float pb = dpp_or<CTRL>(0.f, b);          \
        b = fmaf(m, pb, b);                             \
        m = m * pm;                                     \
    }
    STEP(0x111) STEP(0x112) STEP(0x114) STEP(0x118)
#undef STEP
#define STEPR(CTRL, ROWS)                               \
    {                                                   \
        const float pm = dpp_rows<CTRL, ROWS>(1.f, m);  \
        const float pb = dpp_rows<CTRL, ROWS>(0.f, b);  \
        b = fmaf(m, pb, b);                             \
        m = m * pm;                                     \
    }
    STEPR(0x142, 0xa) STEPR(0x143, 0xc)
#undef STEPR
}
// lane l <-> lane 63 - l
__device__ __forceinline__ float reverse64(float v) {
    v = dpp_or<0x140>(v, v);                                                  // row_mirror: inside rows of 16
    return __shfl(v, (threadIdx.x & 63) ^ 48);                                 // rows 0<->3, 1<->2 (ds_bpermute)
}

__global__ void __launch_bounds__(256) splat_parallel_kernel(const Entry* entries, const Pixel* pixels, int reps, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const Entry e = entries[(size_t)wave * 64 + lane];
    float acc[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int p = 0; p < NPIX; ++p) {
            const Pixel& P = pixels[p];   // wave-uniform: scalar loads
            const float dx = e.x - P.px, dy = e.y - P.py;
            const float power = fmaf(e.ca * dx, dx, fmaf(e.cc * dy, dy, e.cb * dx * dy));
            const float G = __builtin_amdgcn_exp2f(power * 1.44269504f);
            float alpha = fminf(0.99f, e.op * G);
            const bool ok = !(power > 0.f) && alpha >= (1.f / 255.f);
            alpha = ok ? alpha : 0.f;
            const float Ge = ok ? G : 0.f;
            const float om = 1.f - alpha;
            // transmittance in front of this entry: exclusive prefix product
            const float incl = scan_mul(om);
            float Tpre = P.T0 * dpp_or<0x138>(1.f, incl);        // wave_shr:1
            // entries behind the point where the pixel is done (T < 1e-4) do not take part
            const unsigned long long dead = __ballot(Tpre * om < 1e-4f);
            const int first_dead = dead ? __builtin_ctzll(dead) : 64;
            const bool live = lane <= first_dead && ok;   // (the entry that crosses the threshold still blends)
            const float a_e = live ? alpha : 0.f;
            const float w = a_e * Tpre;
            float S = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) S = fmaf(e.c[c], P.d[c], S);
            // accum behind this entry: suffix scan of x -> a x' S' + (1 - a') x over the later entries
            const float om_e = 1.f - a_e;
            float m = reverse64(om_e), b = reverse64(a_e * S);
            scan_affine(m, b);
            // what the later entries alone leave behind: the inclusive result of the neighbour (in reversed order)
            const float accum = reverse64(dpp_or<0x138>(0.f, b));   // wave_shr:1
            float inv = __builtin_amdgcn_rcpf(om_e);
            inv = fmaf(fmaf(-om_e, inv, 1.0f), inv, inv);           // Newton step, as in the kernel in use
            const float dL_dalpha = fmaf(P.bgterm, inv, (S - accum) * Tpre);
            const float g_op = (live ? Ge : 0.f) * dL_dalpha;
            const float GL = e.op * g_op;
            const float h = GL * dx, k2 = GL * dy;
            acc[0] += h;
            acc[1] += k2;
            acc[2] = fmaf(h, dx, acc[2]);
            acc[3] = fmaf(h, dy, acc[3]);
            acc[4] = fmaf(k2, dy, acc[4]);
            acc[5] += g_op;
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[6 + c] = fmaf(w, P.d[c], acc[6 + c]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 14; ++k) s += acc[k];
    out[(size_t)wave * 64 + lane] = s;
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 14; ++k) out[(size_t)gridDim.x * blockDim.x + lane * 14 + k] = acc[k];
    }
}

int main() {
    const int blocks = 256 * 8, threads = 256, waves = blocks * threads / 64;
    std::vector<Entry> he((size_t)waves * 64);
    std::vector<Pixel> hp(NPIX);
    srand(1);
    auto rnd = [] { return rand() / (float)RAND_MAX; };
    for (auto& e : he) {
        e.x = 8 * rnd(); e.y = 8 * rnd(); e.ca = -0.5f * (0.02f + 0.2f * rnd()); e.cc = -0.5f * (0.02f + 0.2f * rnd());
        e.cb = -0.02f * (rnd() - 0.5f); e.op = 0.05f + 0.3f * rnd();
        for (float& c : e.c) c = rnd();
    }
    for (int p = 0; p < NPIX; ++p) {
        hp[p].px = p & 7; hp[p].py = p >> 3; hp[p].T0 = 0.9f; hp[p].bgterm = 0.01f;
        for (float& d : hp[p].d) d = rnd() - 0.5f;
    }
    Entry* de; Pixel* dp; float* dout;
    CK(hipMalloc(&de, he.size() * sizeof(Entry)));
    CK(hipMalloc(&dp, hp.size() * sizeof(Pixel)));
    CK(hipMalloc(&dout, ((size_t)waves * 64 + 64 * 14) * sizeof(float)));
    CK(hipMemcpy(de, he.data(), he.size() * sizeof(Entry), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, hp.data(), hp.size() * sizeof(Pixel), hipMemcpyHostToDevice));
    // correctness of the scans: wave 0, one repetition, against the sequential recurrences
    hipLaunchKernelGGL(splat_parallel_kernel, dim3(blocks), dim3(threads), 0, 0, de, dp, 1, dout);
    CK(hipDeviceSynchronize());
    std::vector<float> got(64 * 14);
    CK(hipMemcpy(got.data(), dout + (size_t)waves * 64, got.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<double> ref(64 * 14, 0.0);
    for (int p = 0; p < NPIX; ++p) {
        double alpha[64], G[64], S[64], T[64], dxs[64], dys[64];
        bool live[64];
        double Trun = hp[p].T0;
        bool done = false;
        for (int l = 0; l < 64; ++l) {
            const Entry& e = he[l];
            const double dx = e.x - hp[p].px, dy = e.y - hp[p].py;
            const double power = e.ca * dx * dx + e.cc * dy * dy + e.cb * dx * dy;
            G[l] = std::exp(power);
            double a = std::fmin(0.99, e.op * G[l]);
            const bool ok = !(power > 0) && a >= 1.0 / 255.0;
            live[l] = ok && !done;
            alpha[l] = live[l] ? a : 0.0;
            T[l] = Trun;
            if (live[l]) {
                if (Trun * (1 - a) < 1e-4) done = true;
                Trun *= (1 - a);
            }
            S[l] = 0;
            for (int c = 0; c < NCH; ++c) S[l] += (double)e.c[c] * hp[p].d[c];
            dxs[l] = dx; dys[l] = dy;
        }
        double accum = 0, last_a = 0, last_S = 0;
        for (int l = 63; l >= 0; --l) {
            accum = last_a * last_S + (1 - last_a) * accum;
            const double a = alpha[l];
            if (a > 0) { last_a = a; last_S = S[l]; } else { last_a = 0; }
            const double dLda = hp[p].bgterm / (1 - a) + (S[l] - accum) * T[l];
            const double g_op = (live[l] ? G[l] : 0.0) * dLda, GL = he[l].op * g_op, h = GL * dxs[l], k2 = GL * dys[l];
            double* r = &ref[l * 14];
            r[0] += h; r[1] += k2; r[2] += h * dxs[l]; r[3] += h * dys[l]; r[4] += k2 * dys[l]; r[5] += g_op;
            for (int c = 0; c < NCH; ++c) r[6 + c] += a * T[l] * hp[p].d[c];
        }
    }
    double worst = 0, top = 0;
    for (size_t i = 0; i < ref.size(); ++i) top = std::fmax(top, std::fabs(ref[i]));
    for (size_t i = 0; i < ref.size(); ++i) worst = std::fmax(worst, std::fabs(got[i] - ref[i]) / (std::fabs(ref[i]) + 1e-2 * top));
    printf("scan check (wave 0, 64 entries x 64 pixels x 14 sums): largest deviation %.2e\n", worst);

    const int reps = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(splat_parallel_kernel, dim3(blocks), dim3(threads), 0, 0, de, dp, reps, dout);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(splat_parallel_kernel, dim3(blocks), dim3(threads), 0, 0, de, dp, reps, dout);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double steps = (double)waves * reps * NPIX;   // wave steps of 64 (pixel, entry) pairs
    printf("%d waves (8 per SIMD), %.0f steps of 64 pairs in %.3f ms: %.1f G pairs/s, %.1f ns per step and SIMD\n", waves, steps,
           ms, steps * 64 / (ms * 1e-3) / 1e9, ms * 1e6 / (steps / 1024.0));
    return 0;
}
