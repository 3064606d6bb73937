// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, one
// sweep over the data per pass ("onesweep"): digit histograms of all passes
// are taken in one upfront read; each pass then ranks a 4096-key tile in LDS,
// obtains its global digit offsets by decoupled look-back over the preceding
// tiles' published counts, and writes the tile out through an LDS staging
// buffer so that stores to one digit's bucket are contiguous.
// Replaces cub::DeviceRadixSort::SortPairs at rasterizer_impl.cu:373-380.
//
// Traffic per pass: 12 B read + 12 B written per pair, plus 3 KiB of status
// words per 4096-pair tile.
#include "kernels.h"

namespace gs {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = SORT_TILE / SORT_THREADS;  // 16
constexpr uint32_t ST_AGG = 1u << 30;   // tile count published
constexpr uint32_t ST_INCL = 2u << 30;  // inclusive prefix published
constexpr uint32_t ST_MASK = (1u << 30) - 1;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

int sort_passes_for(int end_bit) { return (end_bit + 7) / 8; }

__global__ void __launch_bounds__(256) sort_hist_kernel(const uint64_t* keys, int64_t n, int passes, int end_bit,
                                                        uint32_t* ghist) {
    __shared__ uint32_t h[SORT_MAX_PASSES * 256];
    for (int i = threadIdx.x; i < passes * 256; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t k = keys[i];
        for (int p = 0; p < passes; ++p) {
            const int bits = min(8, end_bit - 8 * p);
            const uint32_t d = (uint32_t)(k >> (8 * p)) & ((1u << bits) - 1);
            atomicAdd(&h[p * 256 + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * 256; i += 256) {
        const uint32_t v = h[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

// Exclusive scan of each pass's 256-bin histogram, in place.
__global__ void __launch_bounds__(256) sort_scan_hist_kernel(uint32_t* ghist) {
    __shared__ uint32_t s_tmp[8];
    uint32_t* h = ghist + blockIdx.x * 256;
    uint32_t total;
    const uint32_t ex = block_scan_excl_256(h[threadIdx.x], s_tmp, &total);
    h[threadIdx.x] = ex;
}

__global__ void __launch_bounds__(SORT_THREADS) onesweep_kernel(const uint64_t* __restrict__ kin,
                                                                 const uint32_t* __restrict__ vin,
                                                                 uint64_t* __restrict__ kout,
                                                                 uint32_t* __restrict__ vout, int64_t n, int shift,
                                                                 int bits, const uint32_t* __restrict__ gbase,
                                                                 uint32_t* status, uint32_t* ticket, uint32_t* err) {
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_whist[4][256];   // per-wave digit counts, then exclusive offsets across waves
    __shared__ uint32_t s_dstart[256];     // first slot of each digit inside the tile
    __shared__ uint32_t s_gofs[256];       // global slot of the digit's first key minus s_dstart
    __shared__ uint64_t s_keys[SORT_TILE];
    __shared__ uint32_t s_vals[SORT_TILE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < 4 * 256; i += SORT_THREADS) (&s_whist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t tile_base = (int64_t)tile * SORT_TILE;
    const int64_t wave_base = tile_base + (int64_t)w * (64 * SORT_ITEMS);
    const uint32_t dmask = (1u << bits) - 1;

    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        const bool ok = i < n;
        key[k] = ok ? kin[i] : ~0ull;
        val[k] = ok ? vin[i] : 0u;
    }

    // Stable ranking inside the wave: lanes holding the same digit find each
    // other with 8 ballots; the lowest such lane advances the wave's counter.
    const uint64_t lt_mask = (1ull << lane) - 1;
    volatile uint32_t* wh = s_whist[w];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const bool ok = (wave_base + k * 64 + lane) < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t lower = __popcll(peers & lt_mask);
        const uint32_t cnt = __popcll(peers);
        const uint32_t prev = wh[d];
        if (ok && lower == 0) wh[d] = prev + cnt;
        rank[k] = prev + lower;
    }
    __syncthreads();

    // Thread d: totals of digit d, offsets of each wave inside the digit.
    uint32_t dcount;
    {
        const uint32_t c0 = s_whist[0][tid], c1 = s_whist[1][tid], c2 = s_whist[2][tid], c3 = s_whist[3][tid];
        s_whist[0][tid] = 0;
        s_whist[1][tid] = c0;
        s_whist[2][tid] = c0 + c1;
        s_whist[3][tid] = c0 + c1 + c2;
        dcount = c0 + c1 + c2 + c3;
    }
    uint32_t tile_total;
    const uint32_t dstart = block_scan_excl_256(dcount, s_tmp, &tile_total);
    s_dstart[tid] = dstart;

    // Decoupled look-back for digit `tid`.
    {
        uint32_t* my = status + (size_t)tile * 256 + tid;
        uint32_t excl = 0;
        if (tile == 0) {
            __hip_atomic_store(my, dcount | ST_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(my, dcount | ST_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t look = (int64_t)tile - 1;
            uint32_t spins = 0;
            while (true) {
                const uint32_t v = __hip_atomic_load(status + (size_t)look * 256 + tid, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = v & ~ST_MASK;
                if (flag == 0) {
                    if (++spins > SPIN_LIMIT) {  // predecessor never published: give up, flag the sort
                        atomicOr(err, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += v & ST_MASK;
                if (flag == ST_INCL) break;
                --look;
            }
            __hip_atomic_store(my, ((excl + dcount) & ST_MASK) | ST_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gofs[tid] = gbase[tid] + excl - dstart;
    }
    __syncthreads();

    // Place the tile in LDS in sorted order.
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const bool ok = (wave_base + k * 64 + lane) < n;
        if (ok) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
            const uint32_t pos = s_dstart[d] + s_whist[w][d] + rank[k];
            s_keys[pos] = key[k];
            s_vals[pos] = val[k];
        }
    }
    __syncthreads();

    const int64_t left = n - tile_base;
    const uint32_t tile_n = left < SORT_TILE ? (uint32_t)left : (uint32_t)SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const uint32_t pos = tid + k * SORT_THREADS;
        if (pos < tile_n) {
            const uint64_t kk = s_keys[pos];
            const uint32_t d = (uint32_t)(kk >> shift) & dmask;
            const size_t dst = (size_t)s_gofs[d] + pos;
            kout[dst] = kk;
            vout[dst] = s_vals[pos];
        }
    }
}

hipError_t launch_radix_sort(uint64_t* keys[2], uint32_t* vals[2], int64_t n, int end_bit, void* temp, hipStream_t st,
                             int* result_buffer) {
    *result_buffer = 0;
    if (n <= 0 || end_bit <= 0) return hipSuccess;
    const int passes = sort_passes_for(end_bit);
    const SortTemp t = sort_temp_view(temp, (size_t)n, passes);
    hipError_t e = hipMemsetAsync(t.ghist, 0, t.zero_bytes, st);
    if (e != hipSuccess) return e;
    const int64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const int hist_blocks = (int)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL(sort_hist_kernel, dim3(hist_blocks), dim3(256), 0, st, keys[0], n, passes, end_bit, t.ghist);
    hipLaunchKernelGGL(sort_scan_hist_kernel, dim3(passes), dim3(256), 0, st, t.ghist);
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        const int bits = end_bit - 8 * p < 8 ? end_bit - 8 * p : 8;
        hipLaunchKernelGGL(onesweep_kernel, dim3((unsigned)tiles), dim3(SORT_THREADS), 0, st, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], n, 8 * p, bits, t.ghist + p * 256,
                           t.status + (size_t)p * tiles * 256, t.ticket + p, t.err);
        cur ^= 1;
    }
    *result_buffer = cur;
    return hipGetLastError();
}

}  // namespace gs
