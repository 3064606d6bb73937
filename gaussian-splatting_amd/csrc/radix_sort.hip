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
#include <atomic>
#include <mutex>

#include "kernels.h"

namespace gs {

constexpr uint32_t ST_AGG = 1u << 30;   // tile count published
constexpr uint32_t ST_INCL = 2u << 30;  // inclusive prefix published
constexpr uint32_t ST_MASK = (1u << 30) - 1;
// The same three for a status word of type S (u32: counts below 2^30; u64: any count a u32 position can reach).
// peers & ~(ballot ^ mine) as ONE three-input bit operation (truth table 0x90 = a & ~(b ^ c)); written
// as an expression the compiler splits it into an xor per bit and an or over three of them.
__device__ __forceinline__ uint32_t keep_equal_bits(uint32_t peers, uint32_t ballot, uint32_t mine) {
    return __builtin_amdgcn_bitop3_b32(peers, ballot, mine, 0x90);
}

template <class S> struct St {
    static constexpr S AGG = (S)1 << (sizeof(S) * 8 - 2);
    static constexpr S INCL = (S)2 << (sizeof(S) * 8 - 2);
    static constexpr S MASK = AGG - 1;
};

namespace {
thread_local int t_force_wide = 0;   // test hook, per calling thread
}
bool sort_force_wide() { return t_force_wide != 0; }
void sort_set_force_wide(bool on) { t_force_wide = on ? 1 : 0; }
constexpr uint32_t SPIN_LIMIT = 1u << 22;
#ifndef GS_LOOK_BATCH
#define GS_LOOK_BATCH 4
#endif
constexpr int LOOK_BATCH = GS_LOOK_BATCH;

int sort_passes_for(int begin_bit, int end_bit) { return (end_bit - begin_bit + 7) / 8; }

// Digit histograms of all passes in one read of the keys, 16 bytes per lane and load (a lane
// that reads one key per trip spends its time waiting for that load).
// (nblocks workgroups, numbered blockIdx.x = 0 .. nblocks - 1, share the keys)
// NBINS: bins per pass (256 for digits of up to 8 bits, 512 for 9-bit digits); passes * NBINS <= 2048.
template <class K, int NBINS = 256>
__device__ __forceinline__ void sort_hist_body(const K* keys, int64_t n, int passes, int begin_bit, int end_bit,
                                               uint32_t* ghist, int nblocks) {
    constexpr int V = 16 / sizeof(K);
    __shared__ uint32_t h[SORT_MAX_PASSES * 256];
    for (int i = threadIdx.x; i < passes * NBINS; i += 256) h[i] = 0;
    __syncthreads();
    int shift[SORT_MAX_PASSES], bits[SORT_MAX_PASSES];
    for (int p = 0; p < passes; ++p) digit_of_pass(end_bit - begin_bit, passes, p, &shift[p], &bits[p]);
    auto count = [&](K k) {
        for (int p = 0; p < passes; ++p)
            atomicAdd(&h[p * NBINS + ((uint32_t)(k >> (begin_bit + shift[p])) & ((1u << bits[p]) - 1))], 1u);
    };
    const int64_t stride = (int64_t)nblocks * 256;
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
                    atomicAdd(&h[p * NBINS + cur], c);
                    cur = d;
                    c = 0;
                }
                ++c;
            }
            atomicAdd(&h[p * NBINS + cur], c);
        }
    }
    for (int64_t i = nvec * V + gtid; i < n; i += stride) count(keys[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < passes * NBINS; i += 256) {
        const uint32_t v = h[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

template <class K>
__global__ void __launch_bounds__(256) sort_hist_kernel(const K* keys, int64_t n, int passes, int begin_bit,
                                                        int end_bit, uint32_t* ghist) {
    sort_hist_body<K>(keys, n, passes, begin_bit, end_bit, ghist, (int)gridDim.x);
}

// Several independent sorts of u32 pairs in one grid (blockIdx.y = which): the depth orders of the views
// of a batch, each of which alone is a single wave of workgroups that waits on its look-back chain.
struct SortBatchArgs {
    const uint32_t* kin[SORT_BATCH_MAX];
    const uint32_t* vin[SORT_BATCH_MAX];
    uint32_t* kout[SORT_BATCH_MAX];
    uint32_t* vout[SORT_BATCH_MAX];
    int64_t n[SORT_BATCH_MAX];
    uint32_t* ghist[SORT_BATCH_MAX];    // this pass's 256 bins (all passes' bins for the histogram kernels)
    uint32_t* status[SORT_BATCH_MAX];   // this pass's look-back words
    uint32_t* ticket[SORT_BATCH_MAX];
    uint32_t* err[SORT_BATCH_MAX];
};

template <int NBINS>
__global__ void __launch_bounds__(256) sort_hist_batch_kernel(const SortBatchArgs b, int passes, int begin_bit,
                                                              int end_bit) {
    const int v = blockIdx.y;
    // the grid is sized for the longest array: an array is shared by as many workgroups as it has tiles
    const int64_t tiles = (b.n[v] + SORT_TILE - 1) / SORT_TILE;
    const int nblocks = (int)(tiles < (int64_t)gridDim.x ? tiles : (int64_t)gridDim.x);
    if ((int)blockIdx.x >= nblocks) return;
    sort_hist_body<uint32_t, NBINS>(b.kin[v], b.n[v], passes, begin_bit, end_bit, b.ghist[v], nblocks);
}

// Exclusive scan of each pass's 256-bin histogram, in place.
__global__ void __launch_bounds__(256) sort_scan_hist_kernel(uint32_t* ghist) {
    __shared__ uint32_t s_tmp[8];
    uint32_t* h = ghist + blockIdx.x * 256;
    uint32_t total;
    const uint32_t ex = block_scan_excl_256(h[threadIdx.x], s_tmp, &total);
    h[threadIdx.x] = ex;
}

// Exclusive scan over all threads of a block of NW waves.
template <int NW>
__device__ __forceinline__ uint32_t block_scan_excl_nw(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl_u32(v);
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t t = tmp[i];
        if (i < w) base += t;
        sum += t;
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}

// NBINS threads: one per bin of the pass blockIdx.x
template <int NBINS>
__global__ void __launch_bounds__(NBINS) sort_scan_hist_batch_kernel(const SortBatchArgs b) {
    __shared__ uint32_t s_tmp[NBINS / 64];
    if (b.n[blockIdx.y] <= 0) return;   // left out: no histogram
    uint32_t* h = b.ghist[blockIdx.y] + blockIdx.x * NBINS;
    uint32_t total;
    const uint32_t ex = block_scan_excl_nw<NBINS / 64>(h[threadIdx.x], s_tmp, &total);
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
// SEG: the input is already partitioned into segments by the key bits above `bits` (the key
// emission does that), and the pass orders each segment on its own: workgroup tiles do not
// straddle segments (seg.tile0[s] is the first tile of segment s, seg.start[s] its first
// pair), the look-back ends at the first tile of the segment, and the place of digit d of
// segment s in the output is where the tile range of tile id (s << bits) + d starts.
struct SegArgs {
    const uint32_t* start;  // [nseg + 1] first pair of each segment; start[nseg] = n
    const uint32_t* tile0;  // [nseg + 1] first workgroup tile of each segment; tile0[nseg] = tiles in all
    const uint2* ranges;    // [ntiles] tile ranges (x = first instance of the tile)
    int nseg, ntiles;
};

template <class K, int THREADS, int ITEMS, int BITS, bool SEG, class S = uint32_t, bool PACKED = false>
__device__ __forceinline__ void onesweep_tile(const K* __restrict__ kin, const uint32_t* __restrict__ vin,
                                              K* __restrict__ kout, uint32_t* __restrict__ vout, int64_t n, int shift,
                                              int bits, const uint32_t* __restrict__ gbase, S* status, uint32_t* ticket,
                                              uint32_t* err, const SegArgs& seg) {
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * ITEMS;
    constexpr bool ONE_PHASE = sizeof(K) == 4;  // keys and values fit the staging area together
    constexpr int NB = BITS > 8 ? 512 : 256;    // bins, and status words per tile
    static_assert(TILE == SORT_TILE, "tile size must match the status array layout");
    static_assert(NB <= THREADS, "one thread per bin");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* s_keys = reinterpret_cast<K*>(smem);                               // [TILE]
    uint32_t* s_vals = reinterpret_cast<uint32_t*>(smem) + (ONE_PHASE ? TILE : 0);  // [TILE]; u64 keys: second phase
    // (packed instances are staged as keys only: the tables follow 4 bytes per pair, not 8)
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * (PACKED ? 4 : 8));  // [NW][NB]
    uint32_t* s_hist = s_whist + NW * NB;                                 // [NB]
    uint32_t* s_dstart = s_hist + NB;                                     // [NB]
    uint32_t* s_gofs = s_dstart + NB;                                     // [NB]
    uint32_t* s_tmp = s_gofs + NB;                                        // [8]
    uint32_t* s_tile = s_tmp + 8;                                         // [1]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) *s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < (NW + 1) * NB; i += THREADS) s_whist[i] = 0;  // counters and s_hist
    __syncthreads();
    const uint32_t tile = *s_tile;
    int64_t tile_base = (int64_t)tile * TILE;
    bool first_tile = tile == 0;
    uint32_t my_seg = 0;
    if (SEG) {
        if (tile >= seg.tile0[seg.nseg]) return;  // the grid is an upper bound of the tiles there are
        // the thread that finds the tile's segment fetches what the others need of it in the same round trip
        for (int i = tid; i < seg.nseg; i += THREADS) {
            const uint32_t a0 = seg.tile0[i], a1 = seg.tile0[i + 1];
            const uint32_t p0 = seg.start[i], p1 = seg.start[i + 1];   // (with the two above: one round trip, not two)
            if (a0 <= tile && tile < a1) {
                s_tile[1] = (uint32_t)i;
                s_tile[2] = a0;
                s_tile[3] = p0;
                s_tile[4] = p1;
            }
        }
        __syncthreads();
        my_seg = s_tile[1];
        const uint32_t t0 = s_tile[2];
        first_tile = tile == t0;
        tile_base = (int64_t)s_tile[3] + (int64_t)(tile - t0) * TILE;
        n = s_tile[4];
    }
    const int64_t wave_base = tile_base + (int64_t)w * (64 * ITEMS);
    const uint32_t dmask = (1u << bits) - 1;
    // where digit `tid` of this pass (of this segment) starts in the output: loaded here, long before its use
    // behind the look-back, so that it is not one more round trip at the end of the tile's dependent accesses
    // (segmented pass 0.160 -> 0.148 ms)
    uint32_t out_base = 0u;
    if (tid < NB) {
        if (SEG) {
            const uint32_t t = (my_seg << bits) + (uint32_t)tid;
            out_base = (tid < (1 << bits) && t < (uint32_t)seg.ntiles) ? seg.ranges[t].x : 0u;
        } else {
            out_base = gbase[tid];
        }
    }

    K key[ITEMS];
    uint32_t val[ITEMS];
    uint32_t rank[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        const bool ok = i < n;
        key[k] = ok ? kin[i] : (K)~(K)0;
        val[k] = (ok && !PACKED) ? vin[i] : 0u;
    }

    // 1. Tile histogram, published at once so that later tiles can look back
    //    over this one while it is still ranking.
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if ((wave_base + k * 64 + lane) < n) atomicAdd(&s_hist[(uint32_t)(key[k] >> shift) & dmask], 1u);
    }
    __syncthreads();
    const bool dthread = tid < NB;  // threads that own a digit
    const uint32_t dcount = dthread ? s_hist[tid] : 0u;
    S* const my = status + (size_t)tile * NB + tid;
    if (dthread)
        __hip_atomic_store(my, (S)dcount | (first_tile ? St<S>::INCL : St<S>::AGG), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tile_total;
    uint32_t dstart;
    if constexpr (NB == 256) dstart = block_scan_excl_256_nw<NW>(dcount, s_tmp, &tile_total);
    else dstart = block_scan_excl_nw<NW>(dcount, s_tmp, &tile_total);
    if (dthread) s_dstart[tid] = dstart;

    // 2. Stable ranking inside the wave: lanes holding the same digit find each
    //    other with one ballot per digit bit; the lowest such lane advances the
    //    wave's counter.  The peer mask is kept as two 32-bit halves: with the
    //    digit bit sign-extended to m = 0 / ~0, "keep the lanes whose bit equals
    //    mine" is peers &= ~(ballot ^ m), one three-input bit operation per half.
    //    Bits above `bits` are zero in every digit and change nothing, so the
    //    loop runs over the BITS the kernel was instantiated for without a test.
    typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
    lds_u32* const wh = (lds_u32*)(s_whist + w * NB);
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
            plo = keep_equal_bits(plo, (uint32_t)bal, m);
            phi = keep_equal_bits(phi, (uint32_t)(bal >> 32), m);
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
            const uint32_t c = s_whist[i * NB + tid];
            s_whist[i * NB + tid] = run;
            run += c;
        }
    }

    // 3. Decoupled look-back for digit `tid`, LOOK_BATCH predecessors per round trip.
    if (dthread) {
        uint32_t excl = 0;
        if (!first_tile) {
            int64_t look = (int64_t)tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                S v[LOOK_BATCH];
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    const int64_t t = look - i;
                    v[i] = t >= 0 ? __hip_atomic_load(status + (size_t)t * NB + tid, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : St<S>::INCL;
                }
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    if (done) break;
                    const S flag = v[i] & ~St<S>::MASK;
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
                    excl += (uint32_t)(v[i] & St<S>::MASK);
                    if (flag == St<S>::INCL) done = true;
                }
                look -= LOOK_BATCH;
            next_round:;
            }
            __hip_atomic_store(my, ((S)(excl + dcount) & St<S>::MASK) | St<S>::INCL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gofs[tid] = out_base + excl - dstart;
    }
    __syncthreads();

    // 4. Place the pairs in LDS in sorted order and stream them out.
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const bool ok = (wave_base + k * 64 + lane) < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
        rank[k] = s_dstart[d] + s_whist[w * NB + d] + rank[k];  // slot inside the tile
        if (ok) {
            s_keys[rank[k]] = key[k];
            if (ONE_PHASE && !PACKED) s_vals[rank[k]] = val[k];
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
            if (PACKED) {
                vout[dst[k]] = (uint32_t)kk & ((1u << shift) - 1u);  // the value sits below the digit
            } else {
                kout[dst[k]] = kk;
                if (ONE_PHASE) vout[dst[k]] = s_vals[pos];
            }
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

// (the packed pass holds no values in registers or LDS and runs four workgroups per CU; for the others a
// fourth one costs spills: the batched depth sort 1.22 -> 1.32 ms)
template <class K, int THREADS, int ITEMS, int BITS, bool SEG, class S = uint32_t, bool PACKED = false>
__global__ void __launch_bounds__(THREADS, ((PACKED ? 4 : GS_OS_BLOCKS) * THREADS) / 256)
onesweep_kernel(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
                uint32_t* __restrict__ vout, int64_t n, int shift, int bits, const uint32_t* __restrict__ gbase,
                S* status, uint32_t* ticket, uint32_t* err, const SegArgs seg) {
    static_assert(!PACKED || (SEG && sizeof(K) == 4), "packed pairs only exist between the emission and the segmented pass");
    onesweep_tile<K, THREADS, ITEMS, BITS, SEG, S, PACKED>(kin, vin, kout, vout, n, shift, bits, gbase, status, ticket, err, seg);
}

// One pass of several independent sorts (see SortBatchArgs).  Exactly ceil(n / tile) workgroups of a sort
// go on and take its tickets, so that its tile numbers stay inside its status words.
template <int THREADS, int ITEMS, int BITS>
__global__ void __launch_bounds__(THREADS, (GS_OS_BLOCKS * THREADS) / 256)
onesweep_batch_kernel(const SortBatchArgs b, int shift, int bits) {
    const int v = blockIdx.y;
    const int64_t n = b.n[v];
    if ((int64_t)blockIdx.x * (THREADS * ITEMS) >= n) return;
    onesweep_tile<uint32_t, THREADS, ITEMS, BITS, false, uint32_t>(b.kin[v], b.vin[v], b.kout[v], b.vout[v], n, shift, bits,
                                                                 b.ghist[v], b.status[v], b.ticket[v], b.err[v], SegArgs());
}

// ------------------------------------------------------------------ key emission that partitions
// duplicateWithKeys (rasterizer_impl.cu:70-110) and the first half of the sort in one kernel.
// The Gaussians come in depth order, so their instances only have to be brought into tile
// order stably.  Instead of storing the instances in emission order for a radix pass to read
// back, every 4096 consecutive ones are generated straight into the registers of a workgroup
// and treated as one tile of a radix pass on the HIGH digit of the tile id: chunk histogram
// published at once, stable ranks from ballots, decoupled look-back over the chunks before it,
// coalesced stores through LDS into the segment of that digit, whose place is known
// beforehand from the tile counts.  What is left is one pass on the low digit inside each
// segment (onesweep_kernel<SEG>).
//
// Work is dealt by instances, not by Gaussians: in depth order the first Gaussians are the
// nearest and largest, and a workgroup that owned a fixed number of them would emit for a
// millisecond while the rest of the GPU idles.  chunk_first[c] is the block of 256 Gaussians
// in which instance c * 4096 falls; the chunk's Gaussians (at most 4096 + 1, i.e. seventeen
// blocks; typically two or three) get their running instance offsets into LDS, and each
// instance finds its Gaussian there: every Gaussian marks the slot of its first instance, and a running
// maximum over the marks, in instance order, tells each instance the last Gaussian that started at or
// before it.
#ifndef GS_EP_THREADS
#define GS_EP_THREADS 512
#endif
constexpr int EP_THREADS = GS_EP_THREADS;
#ifndef GS_EP_ITEMS
#define GS_EP_ITEMS 8
#endif
constexpr int EP_ITEMS = GS_EP_ITEMS;
constexpr int EP_CHUNK = EP_THREADS * EP_ITEMS;  // instances per chunk
static_assert(EP_CHUNK == EMIT_CHUNK, "the chunk table is also written by tile_finish_offsets_kernel");
constexpr int EP_MAX_BLOCKS = EP_CHUNK / 256 + 1;
constexpr int EP_MAX_PER = (EP_MAX_BLOCKS * 256 + EP_THREADS - 1) / EP_THREADS;
static_assert(EP_CHUNK >= 2048, "fused_sort_temp_bytes sizes the chunk tables for chunks of at least 2048");

struct EmitPartArgs {
    const uint4* sbin;            // [n] {x0 | y0<<16, w | h<<16, Gaussian index, tiles} in depth order
    const uint32_t* offs;         // [nb256 + 1] exclusive instance offsets per block of 256 of them
    const uint2* chunk_first;     // [chunks + 1] block in which the chunk's first instance falls, that block's offset
    int n, nb256, grid_x, lowbits;
    int vbits;                    // packed output: (low digit << vbits) | Gaussian index in kout, vout unused
    uint32_t R;
    const uint32_t* seg_start;    // [nseg + 1]
    uint32_t* status;             // [chunks][nseg]
    uint32_t* ticket;
    uint32_t* err;
    uint32_t* kout;
    uint32_t* vout;
};

__global__ void __launch_bounds__(256) chunk_map_kernel(int nb256, const uint32_t* __restrict__ offs,
                                                        uint2* __restrict__ chunk_first, uint32_t chunks) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb256) return;
    const uint32_t s0 = offs[b], s1 = offs[b + 1];
    for (uint32_t c = (s0 + EP_CHUNK - 1) / EP_CHUNK; (uint64_t)c * EP_CHUNK < s1; ++c)
        chunk_first[c] = make_uint2((uint32_t)b, s0);
    if (b == nb256 - 1) chunk_first[chunks] = make_uint2((uint32_t)b, s0);  // "the chunk after the last" starts in the last block
}

template <int HB, bool PACKED>
__global__ void __launch_bounds__(EP_THREADS) emit_partition_kernel(const EmitPartArgs a) {
    constexpr int NB = 1 << HB;  // segments = values of the high digit
    constexpr int NW = EP_THREADS / 64;
    // The running offsets are needed until the keys exist, the staging area from then on: one piece of LDS
    // (barriers lie between the last read of the one and the first write of the other).
    __shared__ uint32_t s_stage[2 * EP_CHUNK];
    static_assert(EP_MAX_BLOCKS * 256 + 1 <= 2 * EP_CHUNK, "the offsets must fit the staging area");
    uint32_t* const s_off = s_stage;
    uint32_t* const s_keys = s_stage;
    uint32_t* const s_vals = s_stage + EP_CHUNK;
    __shared__ uint32_t s_whist[NW * NB + NB];  // per-wave counters, then the chunk histogram
    __shared__ uint32_t s_dstart[NB];
    __shared__ uint32_t s_gofs[NB];
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_chunk;
    uint32_t* const s_hist = s_whist + NW * NB;
    // s_mark[q] = 1 + the entry whose first instance is number q of the chunk (0: none starts there), behind
    // the offsets in the same piece of LDS; s_first = 1 + the entry the chunk's first instance falls in
    constexpr int MARK_OFF = (EP_MAX_BLOCKS * 256 + 2 + 1) & ~1;
    static_assert(MARK_OFF + EP_CHUNK / 2 <= 2 * EP_CHUNK, "offsets and marks must fit the staging area");
    static_assert(EP_MAX_BLOCKS * 256 < 65535, "marks are 16-bit");
    uint16_t* const s_mark = reinterpret_cast<uint16_t*>(s_stage + MARK_OFF);
    __shared__ uint32_t s_wmax[NW];
    __shared__ uint32_t s_first;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (loaded here, long before its use behind the look-back: one round trip less at the end of the chain of
    // dependent accesses that bounds a chunk's time)
    const uint32_t seg_base = tid < NB ? a.seg_start[tid] : 0u;
    if (tid == 0) {
        s_chunk = atomicAdd(a.ticket, 1u);
        s_first = 1u;   // entry 0, should the tables not name the entry of the first instance (every index stays in range)
    }
    for (int i = tid; i < (NW + 1) * NB; i += EP_THREADS) s_whist[i] = 0;
    for (int i = tid; i < EP_CHUNK / 2; i += EP_THREADS) reinterpret_cast<uint32_t*>(s_mark)[i] = 0;
    __syncthreads();
    const uint32_t chunk = s_chunk;
    const uint32_t i0 = chunk * EP_CHUNK;             // first instance of the chunk
    const uint32_t chunk_n = min((uint32_t)EP_CHUNK, a.R - i0);
    // The chunk's Gaussians: those of the block its first instance falls in up to the block in which
    // the next chunk starts (one more than needed when that one starts exactly at a block boundary;
    // its entries are then never chosen).  Their running instance offsets go to LDS: every thread
    // sums `per` consecutive tile counts, one scan over the threads' sums gives the rest.
    const uint2 cf = a.chunk_first[chunk], cf_next = a.chunk_first[chunk + 1];
    const int b0 = (int)cf.x;
    const int nblk = min((int)cf_next.x - b0 + 1, EP_MAX_BLOCKS);
    const int cnt = nblk * 256;  // entries of s_off
    {
        const int per = (cnt + EP_THREADS - 1) / EP_THREADS;
        uint32_t tl[EP_MAX_PER];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < EP_MAX_PER; ++j) {
            const int e = tid * per + j;
            const int g = b0 * 256 + e;
            tl[j] = (j < per && e < cnt && g < a.n) ? a.sbin[g].w : 0u;
            mine += tl[j];
        }
        uint32_t total;
        uint32_t ex = cf.y + block_scan_excl_nw<NW>(mine, s_tmp, &total);
#pragma unroll
        for (int j = 0; j < EP_MAX_PER; ++j) {
            const int e = tid * per + j;
            if (j < per && e < cnt) {
                s_off[e] = ex;
                if (ex <= i0) {
                    if (i0 - ex < tl[j]) s_first = (uint32_t)e + 1u;   // exactly one entry holds instance i0
                } else if (ex - i0 < (uint32_t)EP_CHUNK) {
                    s_mark[ex - i0] = (uint16_t)(e + 1);   // (the empty entries behind the last Gaussian all
                }                                          //  start at R, beyond the chunk's instances)
            }
            ex += tl[j];
        }
        if (tid == 0) s_off[cnt] = 0xFFFFFFFFu;
    }
    __syncthreads();

    uint32_t key[EP_ITEMS], val[EP_ITEMS], rank[EP_ITEMS];
    const uint32_t hmask = NB - 1;
    // a. the instances of the chunk, wave-striped: item k of lane l in wave w is number w * 512 + k * 64 + l.
    int lo[EP_ITEMS];  // largest entry with s_off[entry] <= the instance's number
    //    The entry of an instance is the last one that starts at or before it: a running maximum over the
    //    marks in instance order (inside each item across the lanes, from item to item through lane 63,
    //    from wave to wave through LDS), about a fifth of the instructions of eight binary searches.
    {
        uint32_t m[EP_ITEMS];
#pragma unroll
        for (int k = 0; k < EP_ITEMS; ++k) m[k] = s_mark[w * (64 * EP_ITEMS) + k * 64 + lane];
#pragma unroll
        for (int k = 0; k < EP_ITEMS; ++k) m[k] = wave_scan_incl_max_u32(m[k]);
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < EP_ITEMS; ++k) {
            m[k] = max(m[k], carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)m[k], 63);
        }
        if (lane == 0) s_wmax[w] = carry;
        __syncthreads();
        uint32_t before = s_first;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (i < w) before = max(before, s_wmax[i]);
#pragma unroll
        for (int k = 0; k < EP_ITEMS; ++k) lo[k] = (int)max(m[k], before) - 1;
    }
#pragma unroll
    for (int k = 0; k < EP_ITEMS; ++k) {
        const uint32_t q0 = (uint32_t)(w * (64 * EP_ITEMS) + k * 64 + lane);
        key[k] = 0xFFFFFFFFu;
        val[k] = 0u;
        if (q0 < chunk_n) {
            const uint32_t sidx = i0 + q0;
            const uint4 gb = a.sbin[b0 * 256 + lo[k]];
            const uint32_t q = sidx - s_off[lo[k]];
            const uint32_t rw = gb.y & 0xffffu;
            // q / rw without the integer division (about 25 instructions): q < the rectangle's tile count
            // <= 2^16 here (two-digit tile ids), so the float quotient is exact or one off, and the two
            // comparisons repair it
            uint32_t ty = (uint32_t)((float)q * __builtin_amdgcn_rcpf((float)rw));
            ty -= (ty * rw > q) ? 1u : 0u;
            ty += ((ty + 1u) * rw <= q) ? 1u : 0u;
            const uint32_t tx = q - ty * rw;
            key[k] = ((gb.x >> 16) + ty) * (uint32_t)a.grid_x + (gb.x & 0xffffu) + tx;
            val[k] = gb.z;
        }
    }
    // b. chunk histogram of the high digit, published at once
#pragma unroll
    for (int k = 0; k < EP_ITEMS; ++k) {
        const uint32_t q0 = (uint32_t)(w * (64 * EP_ITEMS) + k * 64 + lane);
        if (q0 < chunk_n) atomicAdd(&s_hist[(key[k] >> a.lowbits) & hmask], 1u);
    }
    __syncthreads();
    const bool dthread = tid < NB;
    const uint32_t dcount = dthread ? s_hist[tid] : 0u;
    uint32_t* const my = a.status + (size_t)chunk * NB + tid;
    if (dthread)
        __hip_atomic_store(my, dcount | (chunk == 0 ? ST_INCL : ST_AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t chunk_total;
    const uint32_t dstart = block_scan_excl_nw<NW>(dcount, s_tmp, &chunk_total);
    if (dthread) s_dstart[tid] = dstart;
    // c. stable ranks inside the wave
    typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
    lds_u32* const wh = (lds_u32*)(s_whist + w * NB);
#pragma unroll
    for (int k = 0; k < EP_ITEMS; ++k) {
        const uint32_t q0 = (uint32_t)(w * (64 * EP_ITEMS) + k * 64 + lane);
        const bool ok = q0 < chunk_n;
        const uint32_t d = (key[k] >> a.lowbits) & hmask;
        const uint64_t okm = __builtin_amdgcn_ballot_w64(ok);
        uint32_t plo = (uint32_t)okm, phi = (uint32_t)(okm >> 32);
#pragma unroll
        for (int bit = 0; bit < HB; ++bit) {
            uint32_t m = (uint32_t)(((int32_t)(d << (31 - bit))) >> 31);
            asm("" : "+v"(m));
            const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
            plo = keep_equal_bits(plo, (uint32_t)bal, m);
            phi = keep_equal_bits(phi, (uint32_t)(bal >> 32), m);
        }
        const uint32_t lower = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t cnt_same = __popc(plo) + __popc(phi);
        const uint32_t prev = wh[d];
        if (ok && lower == 0) wh[d] = prev + cnt_same;
        rank[k] = prev + lower;
    }
    __syncthreads();
    if (dthread) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t cw = s_whist[i * NB + tid];
            s_whist[i * NB + tid] = run;
            run += cw;
        }
        // d. decoupled look-back over the chunks before this one
        uint32_t excl = 0;
        if (chunk > 0) {
            int64_t look = (int64_t)chunk - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                uint32_t v[LOOK_BATCH];
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    const int64_t t = look - i;
                    v[i] = t >= 0 ? __hip_atomic_load(a.status + (size_t)t * NB + tid, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : ST_INCL;
                }
#pragma unroll
                for (int i = 0; i < LOOK_BATCH; ++i) {
                    if (done) break;
                    const uint32_t flag = v[i] & ~ST_MASK;
                    if (flag == 0) {  // not published yet: poll again from this chunk
                        look -= i;
                        if (++spins > SPIN_LIMIT) {
                            atomicOr(a.err, 1u);
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
        s_gofs[tid] = seg_base + excl - dstart;
    }
    __syncthreads();
    // e. through LDS in segment order, then out
#pragma unroll
    for (int k = 0; k < EP_ITEMS; ++k) {
        const uint32_t q0 = (uint32_t)(w * (64 * EP_ITEMS) + k * 64 + lane);
        if (q0 < chunk_n) {
            const uint32_t d = (key[k] >> a.lowbits) & hmask;
            const uint32_t slot = s_dstart[d] + s_whist[w * NB + d] + rank[k];
            s_keys[slot] = key[k];
            s_vals[slot] = val[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EP_ITEMS; ++k) {
        const uint32_t pos = (uint32_t)(tid + k * EP_THREADS);
        if (pos < chunk_n) {
            const uint32_t kk = s_keys[pos];
            const uint32_t dst = s_gofs[(kk >> a.lowbits) & hmask] + pos;
            if (PACKED) {
                a.kout[dst] = ((kk & ((1u << a.lowbits) - 1u)) << a.vbits) | s_vals[pos];
            } else {
                a.kout[dst] = kk;
                a.vout[dst] = s_vals[pos];
            }
        }
    }
}

#ifndef GS_OS_THREADS
#define GS_OS_THREADS 512
#endif
constexpr int OS_THREADS = GS_OS_THREADS;
constexpr int OS_ITEMS = SORT_TILE / OS_THREADS;
constexpr size_t OS_LDS = (size_t)SORT_TILE * 8 + (OS_THREADS / 64 + 4) * 256 * 4 + 64;
constexpr size_t OS_LDS_PACKED = (size_t)SORT_TILE * 4 + (OS_THREADS / 64 + 4) * 256 * 4 + 64;

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
        if (t.wide) {
            // 2^30 pairs and more: 64-bit look-back words, one kernel per key type (8-bit digit loop)
            auto kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, 8, false, uint64_t>;
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)OS_LDS);
            if (ea != hipSuccess) return ea;
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(OS_THREADS), OS_LDS, st, keys[cur], vals[cur],
                               keys[cur ^ 1], vals[cur ^ 1], n, begin_bit + shift, bits, t.ghist + p * 256,
                               (uint64_t*)t.status + (size_t)p * tiles * 256, t.ticket + p, t.err, SegArgs());
            cur ^= 1;
            continue;
        }
        // u32 keys (the rasterizer's two sorts) have kernels for 6-, 7- and 8-bit digits
        auto kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, 8, false>;
        if (sizeof(K) == 4 && bits <= 6) kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, sizeof(K) == 4 ? 6 : 8, false>;
        else if (sizeof(K) == 4 && bits == 7) kern = onesweep_kernel<K, OS_THREADS, OS_ITEMS, sizeof(K) == 4 ? 7 : 8, false>;
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)OS_LDS);
        if (ea != hipSuccess) return ea;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(OS_THREADS), OS_LDS, st, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], n, begin_bit + shift, bits, t.ghist + p * 256,
                           (uint32_t*)t.status + (size_t)p * tiles * 256, t.ticket + p, t.err, SegArgs());
        cur ^= 1;
    }
    *result_buffer = cur;
    return hipGetLastError();
}

// Up to SORT_BATCH_MAX stable sorts of u32 pairs, pass by pass in shared launches.  key_bits = 32: four passes
// of 8-bit digits, the input in keys[i][0] / vals[i][0]; key_bits = 27 (SORT_NARROW_KEY_BITS; the caller
// guarantees keys below 2^27): three passes of 9-bit digits with 512 bins and status words per tile, the input
// in keys[i][1] / vals[i][1].  Either way every result ends in keys[i][0] / vals[i][0].  temp[i]: room for
// sort_temp_view(n, SORT_BATCH_TEMP_PASSES).  Arrays of at most 2^30 pairs; n[i] <= 0 leaves sort i out.
template <int NBINS, int BITS>
static hipError_t radix_sort_u32_batch_impl(int count, uint32_t* const (*keys)[2], uint32_t* const (*vals)[2],
                                            const int64_t* n, void* const* temp, hipStream_t st, int key_bits,
                                            int passes) {
    SortBatchArgs b = SortBatchArgs();
    SortTemp t[SORT_BATCH_MAX];
    int64_t tiles[SORT_BATCH_MAX], max_tiles = 0;
    int cur = passes & 1;   // an odd number of passes starts in buffer 1
    for (int i = 0; i < count; ++i) {
        b.n[i] = n[i] > 0 ? n[i] : 0;
        tiles[i] = (b.n[i] + SORT_TILE - 1) / SORT_TILE;
        if (tiles[i] == 0) continue;
        if (sort_status_wide((size_t)b.n[i])) return hipErrorInvalidValue;
        t[i] = sort_temp_view(temp[i], (size_t)b.n[i], SORT_BATCH_TEMP_PASSES);
        hipError_t e = hipMemsetAsync(t[i].ghist, 0, t[i].zero_bytes, st);
        if (e != hipSuccess) return e;
        if (tiles[i] > max_tiles) max_tiles = tiles[i];
        b.kin[i] = keys[i][cur];
        b.ghist[i] = t[i].ghist;
    }
    if (max_tiles == 0) return hipSuccess;
    const unsigned hist_blocks = (unsigned)(max_tiles < 2048 ? max_tiles : 2048);
    hipLaunchKernelGGL(sort_hist_batch_kernel<NBINS>, dim3(hist_blocks, count), dim3(256), 0, st, b, passes, 0, key_bits);
    hipLaunchKernelGGL(sort_scan_hist_batch_kernel<NBINS>, dim3(passes, count), dim3(NBINS), 0, st, b);
    auto kern = onesweep_batch_kernel<OS_THREADS, OS_ITEMS, BITS>;
    constexpr size_t lds = (size_t)SORT_TILE * 8 + (OS_THREADS / 64 + 4) * NBINS * 4 + 64;
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds);
    if (ea != hipSuccess) return ea;
    for (int p = 0; p < passes; ++p) {
        int shift, bits;
        digit_of_pass(key_bits, passes, p, &shift, &bits);
        for (int i = 0; i < count; ++i) {
            if (tiles[i] == 0) continue;
            b.kin[i] = keys[i][cur]; b.vin[i] = vals[i][cur];
            b.kout[i] = keys[i][cur ^ 1]; b.vout[i] = vals[i][cur ^ 1];
            b.ghist[i] = t[i].ghist + p * NBINS;
            b.status[i] = (uint32_t*)t[i].status + (size_t)p * tiles[i] * NBINS;
            b.ticket[i] = t[i].ticket + p;
            b.err[i] = t[i].err;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)max_tiles, count), dim3(OS_THREADS), lds, st, b, shift, bits);
        cur ^= 1;
    }
    return hipGetLastError();
}

hipError_t launch_radix_sort_u32_batch(int count, uint32_t* const (*keys)[2], uint32_t* const (*vals)[2], const int64_t* n,
                                       void* const* temp, hipStream_t st, int key_bits) {
    if (count <= 0) return hipSuccess;
    if (count > SORT_BATCH_MAX) return hipErrorInvalidValue;
    static_assert(3 * 512 <= SORT_BATCH_TEMP_PASSES * 256 && 4 * 256 <= SORT_BATCH_TEMP_PASSES * 256,
                  "status words of either digit width must fit the temp area");
    if (key_bits == SORT_NARROW_KEY_BITS) return radix_sort_u32_batch_impl<512, 9>(count, keys, vals, n, temp, st, 27, 3);
    if (key_bits == 32) return radix_sort_u32_batch_impl<256, 8>(count, keys, vals, n, temp, st, 32, 4);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------ emission + segmented pass
namespace {
struct FusedTemp {
    uint32_t *ghist, *seg_start, *seg_tile0, *ticket, *err, *status_a, *status_b, *chunk_first;
    size_t zero_bytes;
};

FusedTemp fused_temp_view(void* temp, int64_t R, int n_visible, int nbits) {
    // the header of the plain sort's temp area (histograms, tickets, error word), then the status words
    const SortTemp t = sort_temp_view(temp, 1, 0);
    const size_t header = sort_temp_view(nullptr, 1, 0).bytes;
    int lowshift, lowbits;
    digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
    const size_t nseg = (size_t)1 << (nbits - lowbits);
    const size_t chunks = ((size_t)R + EP_CHUNK - 1) / EP_CHUNK;
    const size_t words_a = align_up(chunks * nseg + 1, 64);
    const size_t words_b = ((size_t)R / SORT_TILE + nseg + 2) * 256;
    (void)n_visible;
    FusedTemp f;
    f.ghist = t.ghist;
    f.seg_start = temp ? t.ghist + 1024 : nullptr;  // [nseg + 1] each, in the part of the histogram area two passes leave free
    f.seg_tile0 = temp ? t.ghist + 1536 : nullptr;
    f.ticket = t.ticket;
    f.err = t.err;
    f.status_a = temp ? (uint32_t*)((char*)temp + header) : nullptr;
    f.status_b = temp ? f.status_a + words_a : nullptr;
    f.chunk_first = temp ? f.status_b + words_b : nullptr;  // [chunks + 1] pairs, written before they are read: outside the zeroed part
    f.zero_bytes = header + (words_a + words_b) * sizeof(uint32_t);
    return f;
}
}  // namespace

bool fused_sort_fits(int64_t R, int n_visible, int nbits) {
    if (R <= 0 || n_visible <= 0 || sort_passes_for(0, nbits) != 2) return false;
    if (sort_status_wide((size_t)R)) return false;  // its look-back words are 32-bit: the plain passes take over
    int lowshift, lowbits;
    digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
    const int hb = nbits - lowbits;
    if (hb < 1 || hb > 8 || (int64_t)n_visible > R) return false;
    return fused_temp_view(nullptr, R, n_visible, nbits).zero_bytes + (R / EP_CHUNK + 3) * sizeof(uint2) <=
           fused_sort_temp_bytes((size_t)R);
}

hipError_t fused_sort_prepare(void* temp, int64_t R, int n_visible, int nbits, hipStream_t st, uint32_t** ghist,
                              uint32_t** seg_start, uint32_t** seg_tile0) {
    const FusedTemp f = fused_temp_view(temp, R, n_visible, nbits);
    *ghist = f.ghist;
    *seg_start = f.seg_start;
    *seg_tile0 = f.seg_tile0;
    return hipMemsetAsync(f.ghist, 0, f.zero_bytes, st);
}

bool fused_sort_packs(int P, int nbits) {
    int lowshift, lowbits;
    digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
    return (int64_t)P <= ((int64_t)1 << (32 - lowbits));
}

void fused_chunk_table(void* temp, int64_t R, int n_visible, int nbits, uint2** chunk_first, uint32_t* chunks) {
    *chunk_first = reinterpret_cast<uint2*>(fused_temp_view(temp, R, n_visible, nbits).chunk_first);
    *chunks = (uint32_t)((R + EP_CHUNK - 1) / EP_CHUNK);
}

hipError_t launch_emit_partition(const GeomBuf& g, int n_visible, int grid_x, int nbits, int64_t R, uint32_t* keys[2],
                                 uint32_t* vals[2], void* temp, hipStream_t st, bool packed, bool offsets_ready) {
    const FusedTemp f = fused_temp_view(temp, R, n_visible, nbits);
    int lowshift, lowbits;
    digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
    const int hb = nbits - lowbits;
    const int nb256 = (n_visible + PRE_BLOCK - 1) / PRE_BLOCK;
    const unsigned chunks = (unsigned)((R + EP_CHUNK - 1) / EP_CHUNK);
    if (!offsets_ready) {
        // instance offsets per block of 256 Gaussians (g.total was read by the host already; the scan rewrites the same values)
        hipError_t e0 = launch_scan_u32(g.order_sums, nb256, g.total, st);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL(chunk_map_kernel, dim3((nb256 + 255) / 256), dim3(256), 0, st, nb256, g.order_sums,
                           reinterpret_cast<uint2*>(f.chunk_first), chunks);
    }
    EmitPartArgs a;
    a.sbin = g.sbin;
    a.offs = g.order_sums;
    a.chunk_first = reinterpret_cast<const uint2*>(f.chunk_first);
    a.n = n_visible;
    a.nb256 = nb256;
    a.grid_x = grid_x;
    a.lowbits = lowbits;
    a.vbits = 32 - lowbits;
    a.R = (uint32_t)R;
    a.seg_start = f.seg_start;
    a.status = f.status_a;
    a.ticket = f.ticket;
    a.err = f.err;
    a.kout = keys[1];
    a.vout = vals[1];
    const dim3 grid(chunks), block(EP_THREADS);
#define GS_EMIT_CASE(HB_)                                                                              \
    case HB_:                                                                                          \
        if (packed) hipLaunchKernelGGL((emit_partition_kernel<HB_, true>), grid, block, 0, st, a);     \
        else hipLaunchKernelGGL((emit_partition_kernel<HB_, false>), grid, block, 0, st, a);           \
        break;
    switch (hb < 8 ? hb : 8) {
        GS_EMIT_CASE(1) GS_EMIT_CASE(2) GS_EMIT_CASE(3) GS_EMIT_CASE(4)
        GS_EMIT_CASE(5) GS_EMIT_CASE(6) GS_EMIT_CASE(7) GS_EMIT_CASE(8)
    }
#undef GS_EMIT_CASE
    return hipGetLastError();
}

// The low digit inside each segment: keys[1] -> keys[0].
hipError_t launch_segmented_pass(int n_visible, int ntiles, int nbits, int64_t R, const uint2* ranges, uint32_t* keys[2],
                                 uint32_t* vals[2], void* temp, hipStream_t st, bool packed) {
    const FusedTemp f = fused_temp_view(temp, R, n_visible, nbits);
    int lowshift, lowbits;
    digit_of_pass(nbits, 2, 0, &lowshift, &lowbits);
    const int hb = nbits - lowbits;
    SegArgs seg;
    seg.start = f.seg_start;
    seg.tile0 = f.seg_tile0;
    seg.ranges = ranges;
    seg.nseg = 1 << hb;
    seg.ntiles = ntiles;
    auto kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 8, true>;
    if (lowbits <= 6) kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 6, true>;
    else if (lowbits == 7) kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 7, true>;
    if (packed) {
        kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 8, true, uint32_t, true>;
        if (lowbits <= 6) kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 6, true, uint32_t, true>;
        else if (lowbits == 7) kern = onesweep_kernel<uint32_t, OS_THREADS, OS_ITEMS, 7, true, uint32_t, true>;
    }
    const size_t lds = packed ? OS_LDS_PACKED : OS_LDS;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    const unsigned tiles = (unsigned)(R / SORT_TILE + seg.nseg + 1);  // an upper bound; the kernel knows the number
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(OS_THREADS), lds, st, (const uint32_t*)keys[1],
                       (const uint32_t*)vals[1], keys[0], vals[0], R, packed ? 32 - lowbits : 0, lowbits,
                       (const uint32_t*)nullptr, f.status_b,
                       f.ticket + 1, f.err, seg);
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

// ------------------------------------------------------------------ sort without a caller-provided temp area
// For callers that cannot allocate device memory (HIP graph capture, serving loops): the sort's temp
// area (digit histograms, tickets, error word, look-back status words) lives in a file-scope device
// array instead of a buffer the caller brings.  Each GPU holds its own copy of the array at its own
// address; the address is looked up once per device and kept.  All calls on one GPU share that array:
// calls that overlap on one GPU (two streams, two threads) race.
constexpr int64_t SORT_STATIC_MAX_PAIRS = 1ll << 21;
// header (8 KiB of histograms + ticket and error lines) + 8 passes x 512 tiles x 256 status words
constexpr size_t SORT_STATIC_BYTES = 8192 + 256 + 256 + (size_t)SORT_MAX_PASSES * (SORT_STATIC_MAX_PAIRS / SORT_TILE) * 256 * 4;
static __device__ __align__(256) unsigned char scratch[SORT_STATIC_BYTES];

namespace {
constexpr int STATIC_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[STATIC_MAX_DEVICES] = {};

// Device address of `scratch` on the current device (hipGetSymbolAddress once per device).
hipError_t static_scratch(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= STATIC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return hipSuccess;
}
}  // namespace

int64_t sort_static_max_pairs() { return SORT_STATIC_MAX_PAIRS; }

bool sort_static_fits(int64_t n, int begin_bit, int end_bit) {
    if (n <= 0 || end_bit <= begin_bit) return true;
    return n <= SORT_STATIC_MAX_PAIRS &&
           sort_temp_view(nullptr, (size_t)n, sort_passes_for(begin_bit, end_bit)).bytes <= SORT_STATIC_BYTES;
}

hipError_t sort_static_read_error(hipStream_t st, uint32_t* err) {
    void* temp = nullptr;
    hipError_t e = static_scratch(&temp);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(err, sort_temp_view(temp, 1, 0).err, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

hipError_t launch_radix_sort_static(uint64_t* keys[2], uint32_t* vals[2], int64_t n, int begin_bit, int end_bit,
                                    hipStream_t st, int* result_buffer) {
    *result_buffer = 0;
    if (n <= 0 || end_bit <= begin_bit) return hipSuccess;
    if (!sort_static_fits(n, begin_bit, end_bit)) return hipErrorInvalidValue;
    void* temp = nullptr;
    hipError_t e = static_scratch(&temp);
    if (e != hipSuccess) return e;
    // radix_sort_impl zeroes the histograms, tickets, error word and status words it is going to use
    // (hipMemsetAsync on `temp`), so what the previous call left there does not matter
    return radix_sort_impl<uint64_t>(keys, vals, n, begin_bit, end_bit, temp, st, result_buffer, false);
}

}  // namespace gs
