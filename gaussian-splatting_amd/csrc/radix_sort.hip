// Stable LSD radix sort of (key, u32 value) pairs with u64 or u32 keys, at most
// 8 bits per pass (the key bits are dealt evenly to the passes), one sweep over
// the data per pass ("onesweep"): digit histograms of
// all passes are taken in one upfront read (or handed in by the producer of the
// keys); each pass then ranks a 4096-key tile in LDS, obtains its global digit
// offsets by decoupled look-back over the preceding tiles' published counts,
// and writes the tile out through an LDS staging buffer so that stores to one
// digit's bucket are contiguous.
// Replaces cub::DeviceRadixSort::SortPairs at rasterizer_impl.cu:373-380.
//
// Traffic per pass: key + 4 B read and written per pair, plus 1 KiB of status
// words per 4096-pair tile.
#include "kernels.h"

namespace gs {

constexpr uint32_t ST_AGG = 1u << 30;   // tile count published
constexpr uint32_t ST_INCL = 2u << 30;  // inclusive prefix published
constexpr uint32_t ST_MASK = (1u << 30) - 1;
constexpr uint32_t SPIN_LIMIT = 1u << 22;
#ifndef GS_LOOK_BATCH
#define GS_LOOK_BATCH 4
#endif
constexpr int LOOK_BATCH = GS_LOOK_BATCH;

int sort_passes_for(int begin_bit, int end_bit) { return (end_bit - begin_bit + 7) / 8; }

// Digit histograms of all passes in one read of the keys, 16 bytes per lane and load (a lane
// that reads one key per trip spends its time waiting for that load).
template <class K>
__global__ void __launch_bounds__(256) sort_hist_kernel(const K* keys, int64_t n, int passes, int begin_bit,
                                                        int end_bit, uint32_t* ghist) {
    constexpr int V = 16 / sizeof(K);
    __shared__ uint32_t h[SORT_MAX_PASSES * 256];
    for (int i = threadIdx.x; i < passes * 256; i += 256) h[i] = 0;
    __syncthreads();
    int shift[SORT_MAX_PASSES], bits[SORT_MAX_PASSES];
    for (int p = 0; p < passes; ++p) digit_of_pass(end_bit - begin_bit, passes, p, &shift[p], &bits[p]);
    auto count = [&](K k) {
        for (int p = 0; p < passes; ++p)
            atomicAdd(&h[p * 256 + ((uint32_t)(k >> (begin_bit + shift[p])) & ((1u << bits[p]) - 1))], 1u);
    };
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // whole 16-byte groups when the array starts on a 16-byte boundary, single keys otherwise and for the rest
    const int64_t nvec = (reinterpret_cast<uintptr_t>(keys) & 15) == 0 ? n / V : 0;
    struct alignas(16) Group { K k[V]; };
    const Group* groups = reinterpret_cast<const Group*>(keys);
    for (int64_t i = gtid; i < nvec; i += stride) {
        const Group g = groups[i];
        // neighbouring keys often share a digit (the tile ids of one splat row): one add per run
        for (int p = 0; p < passes; ++p) {
            const uint32_t mask = (1u << bits[p]) - 1;
            uint32_t cur = (uint32_t)(g.k[0] >> (begin_bit + shift[p])) & mask, c = 1;
#pragma unroll
            for (int e = 1; e < V; ++e) {
                const uint32_t d = (uint32_t)(g.k[e] >> (begin_bit + shift[p])) & mask;
                if (d != cur) {
                    atomicAdd(&h[p * 256 + cur], c);
                    cur = d;
                    c = 0;
                }
                ++c;
            }
            atomicAdd(&h[p * 256 + cur], c);
        }
    }
    for (int64_t i = nvec * V + gtid; i < n; i += stride) count(keys[i]);
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

// Exclusive scan over the first 256 threads' values in a block of NW waves.
template <int NW>
__device__ __forceinline__ uint32_t block_scan_excl_256_nw(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl_u32(v);
    if (lane == 63 && w < 4) tmp[w] = incl;
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

// One pass over one tile of SORT_TILE = THREADS * ITEMS pairs.  LDS: an
// 8-byte-per-pair staging area (u64 keys first and then, once they have left,
// the values; u32 keys and values side by side in one phase), per-wave digit
// counters and three 256-entry tables; about 41 KiB with 8 waves of 8 keys per lane (78 VGPRs, 6 waves/SIMD).
#ifndef GS_OS_BLOCKS
#define GS_OS_BLOCKS 2
#endif
template <class K, int THREADS, int ITEMS, int BITS>
__global__ void __launch_bounds__(THREADS, (GS_OS_BLOCKS * THREADS) / 256)
onesweep_kernel(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
                uint32_t* __restrict__ vout, int64_t n, int shift, int bits, const uint32_t* __restrict__ gbase,
                uint32_t* status, uint32_t* ticket, uint32_t* err) {
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * ITEMS;
    constexpr bool ONE_PHASE = sizeof(K) == 4;  // keys and values fit the staging area together
    static_assert(TILE == SORT_TILE, "tile size must match the status array layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* s_keys = reinterpret_cast<K*>(smem);                               // [TILE]
    uint32_t* s_vals = reinterpret_cast<uint32_t*>(smem) + (ONE_PHASE ? TILE : 0);  // [TILE]; u64 keys: second phase
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * 8);  // [NW][256]
    uint32_t* s_hist = s_whist + NW * 256;                                // [256]
    uint32_t* s_dstart = s_hist + 256;                                    // [256]
    uint32_t* s_gofs = s_dstart + 256;                                    // [256]
    uint32_t* s_tmp = s_gofs + 256;                                       // [8]
    uint32_t* s_tile = s_tmp + 8;                                         // [1]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) *s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < (NW + 1) * 256; i += THREADS) s_whist[i] = 0;  // counters and s_hist
    __syncthreads();
    const uint32_t tile = *s_tile;
    const int64_t tile_base = (int64_t)tile * TILE;
    const int64_t wave_base = tile_base + (int64_t)w * (64 * ITEMS);
    const uint32_t dmask = (1u << bits) - 1;

    K key[ITEMS];
    uint32_t val[ITEMS];
    uint32_t rank[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        const bool ok = i < n;
        key[k] = ok ? kin[i] : (K)~(K)0;
        val[k] = ok ? vin[i] : 0u;
    }

    // 1. Tile histogram, published at once so that later tiles can look back
    //    over this one while it is still ranking.
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if ((wave_base + k * 64 + lane) < n) atomicAdd(&s_hist[(uint32_t)(key[k] >> shift) & dmask], 1u);
    }
    __syncthreads();
    const bool dthread = tid < 256;  // threads that own a digit
    const uint32_t dcount = dthread ? s_hist[tid] : 0u;
    uint32_t* const my = status + (size_t)tile * 256 + tid;
    if (dthread)
        __hip_atomic_store(my, dcount | (tile == 0 ? ST_INCL : ST_AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tile_total;
    const uint32_t dstart = block_scan_excl_256_nw<NW>(dcount, s_tmp, &tile_total);
    if (dthread) s_dstart[tid] = dstart;

    // 2. Stable ranking inside the wave: lanes holding the same digit find each
    //    other with one ballot per digit bit; the lowest such lane advances the
    //    wave's counter.  The peer mask is kept as two 32-bit halves: with the
    //    digit bit sign-extended to m = 0 / ~0, "keep the lanes whose bit equals
    //    mine" is peers &= ~(ballot ^ m), one three-input bit operation per half.
    //    Bits above `bits` are zero in every digit and change nothing, so the
    //    loop runs over the BITS the kernel was instantiated for without a test.
    typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
    lds_u32* const wh = (lds_u32*)(s_whist + w * 256);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const bool ok = (wave_base + k * 64 + lane) < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
        const uint64_t okm = __builtin_amdgcn_ballot_w64(ok);
        uint32_t plo = (uint32_t)okm, phi = (uint32_t)(okm >> 32);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            uint32_t m = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
            asm("" : "+v"(m));  // the ballot compares m itself, not a second extraction of the bit
            const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
            plo &= ~((uint32_t)bal ^ m);
            phi &= ~((uint32_t)(bal >> 32) ^ m);
        }
        const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t cnt = __popc(plo) + __popc(phi);
        const uint32_t prev = wh[d];
        if (ok && lower == 0) wh[d] = prev + cnt;
        rank[k] = prev + lower;
    }
    __syncthreads();
    if (dthread) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t c = s_whist[i * 256 + tid];
            s_whist[i * 256 + tid] = run;
            run += c;
        }
    }

    // 3. Decoupled look-back for digit `tid`, LOOK_BATCH predecessors per round trip.
    if (dthread) {
        uint32_t excl = 0;
        if (tile > 0) {
            int64_t look = (int64_t)tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                uint32_t v[LOOK_BATCH];
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    const int64_t t = look - i;
                    v[i] = t >= 0 ? __hip_atomic_load(status + (size_t)t * 256 + tid, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : ST_INCL;
                }
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    if (done) break;
                    const uint32_t flag = v[i] & ~ST_MASK;
                    if (flag == 0) {  // not published yet: poll again from this tile
                        look -= i;
                        if (++spins > SPIN_LIMIT) {
                            atomicOr(err, 1u);
                            done = true;
                        } else {
                            __builtin_amdgcn_s_sleep(1);
                        }
                        goto next_round;
                    }
                    excl += v[i] & ST_MASK;
                    if (flag == ST_INCL) done = true;
                }
                look -= LOOK_BATCH;
            next_round:;
            }
            __hip_atomic_store(my, ((excl + dcount) & ST_MASK) | ST_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gofs[tid] = gbase[tid] + excl - dstart;
    }
    __syncthreads();

    // 4. Place the pairs in LDS in sorted order and stream them out.
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const bool ok = (wave_base + k * 64 + lane) < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
        rank[k] = s_dstart[d] + s_whist[w * 256 + d] + rank[k];  // slot inside the tile
        if (ok) {
            s_keys[rank[k]] = key[k];
            if (ONE_PHASE) s_vals[rank[k]] = val[k];
        }
    }
    __syncthreads();
    const int64_t left = n - tile_base;
    const uint32_t tile_n = left < TILE ? (uint32_t)left : (uint32_t)TILE;
    uint32_t dst[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t pos = tid + k * THREADS;
        dst[k] = 0;
        if (pos < tile_n) {
            const K kk = s_keys[pos];
            const uint32_t d = (uint32_t)(kk >> shift) & dmask;
            dst[k] = s_gofs[d] + pos;
            kout[dst[k]] = kk;
            if (ONE_PHASE) vout[dst[k]] = s_vals[pos];
        }
    }
    if (ONE_PHASE) return;
    __syncthreads();
    // 5. u64 keys: the values follow through the same staging area (the output
    //    slots were kept in registers).
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if ((wave_base + k * 64 + lane) < n) s_vals[rank[k]] = val[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t pos = tid + k * THREADS;
        if (pos < tile_n) vout[dst[k]] = s_vals[pos];
    }
}

#ifndef GS_OS_THREADS
#define GS_OS_THREADS 512
#endif
constexpr int OS_THREADS = GS_OS_THREADS;
constexpr int OS_ITEMS = SORT_TILE / OS_THREADS;
constexpr size_t OS_LDS = (size_t)SORT_TILE * 8 + (OS_THREADS / 64 + 4) * 256 * 4 + 64;

uint32_t* sort_prepare(void* temp, int64_t n, int passes, hipStream_t st, hipError_t* e) {
    const SortTemp t = sort_temp_view(temp, (size_t)(n > 0 ? n : 1), passes);
    *e = hipMemsetAsync(t.ghist, 0, t.zero_bytes, st);
    return t.ghist;
}

template <class K>
hipError_t radix_sort_impl(K* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit, void* temp,
                           hipStream_t st, int* result_buffer, bool hist_ready) {
    *result_buffer = 0;
    if (n <= 0 || end_bit <= begin_bit) return hipSuccess;
    const int passes = sort_passes_for(begin_bit, end_bit);
    const SortTemp t = sort_temp_view(temp, (size_t)n, passes);
    const int64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    if (!hist_ready) {
        hipError_t e = hipMemsetAsync(t.ghist, 0, t.zero_bytes, st);
        if (e != hipSuccess) return e;
        const int hist_blocks = (int)(tiles < 2048 ? tiles : 2048);
        hipLaunchKernelGGL(sort_hist_kernel<K>, dim3(hist_blocks), dim3(256), 0, st, keys[0], n, passes, begin_bit,
                           end_bit, t.ghist);
    }
    hipLaunchKernelGGL(sort_scan_hist_kernel, dim3(passes), dim3(256), 0, st, t.ghist);
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        int shift, bits;
        digit_of_pass(end_bit - begin_bit, passes, p, &shift, &bits);
        // u32 keys (the rasterizer's two sorts) have kernels for 6-, 7- and 8-bit digits
        auto kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, 8>;
        if (sizeof(K) == 4 && bits <= 6) kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, sizeof(K) == 4 ? 6 : 8>;
        else if (sizeof(K) == 4 && bits == 7) kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, sizeof(K) == 4 ? 7 : 8>;
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)OS_LDS);
        if (ea != hipSuccess) return ea;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(OS_THREADS), OS_LDS, st, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], n, begin_bit + shift, bits, t.ghist + p * 256,
                           t.status + (size_t)p * tiles * 256, t.ticket + p, t.err);
        cur ^= 1;
    }
    *result_buffer = cur;
    return hipGetLastError();
}

hipError_t launch_radix_sort(uint64_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit, void* temp,
                             hipStream_t st, int* result_buffer) {
    return radix_sort_impl<uint64_t>(keys, vals, n, begin_bit, end_bit, temp, st, result_buffer, false);
}

hipError_t launch_radix_sort_u32(uint32_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit,
                                 void* temp, hipStream_t st, int* result_buffer, bool hist_ready) {
    return radix_sort_impl<uint32_t>(keys, vals, n, begin_bit, end_bit, temp, st, result_buffer, hist_ready);
}

}  // namespace gs
