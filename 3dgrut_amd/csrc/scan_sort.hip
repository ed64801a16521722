// scan_sort.hip — device-wide prefix sum and stable LSD radix sort for gfx950 (wave64).
//
// These replace the two CUB primitives on the reference's 3DGUT path
// (cub::DeviceScan::InclusiveSum gutRenderer.cu:302-310, cub::DeviceRadixSort::SortPairs :356-365).
// Both are three-kernel, fully deterministic formulations (no inter-workgroup hand-off inside a
// launch, so no dependence on dispatch order — cdna_hip_programming.md §6 G16):
//   scan : per-block reduce -> single-block scan of block sums -> per-block scan + offset
//   sort : per pass { per-block digit histogram -> scan of the digit-major histogram matrix ->
//                     per-block stable ranking (wave64 ballot match) + scatter }
// HBM-bound integer work: 16 B/lane vector loads where the layout allows, 8-bit digits.
#include "common.hpp"

namespace grut {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems   = 8;
constexpr int kScanTile    = kScanThreads * kScanItems;  // 2048

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total, uint32_t* s_wave /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = s_wave[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ void load_items(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather,
                                           uint32_t base, uint32_t n, uint32_t (&x)[kScanItems]) {
    const uint32_t i0 = base + threadIdx.x * kScanItems;
    if (!gather && i0 + kScanItems <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(in + i0);
        const uint4 b = *reinterpret_cast<const uint4*>(in + i0 + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            const uint32_t i = i0 + k;
            x[k] = i < n ? (gather ? in[gather[i]] : in[i]) : 0u;
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                                   const uint32_t* __restrict__ gather, uint32_t n,
                                                                   uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_wave[4];
    uint32_t x[kScanItems];
    load_items(in, gather, blockIdx.x * kScanTile, n, x);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += x[k];
    uint32_t total;
    (void)block_excl_scan_256(s, &total, s_wave);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// exclusive scan of nb block sums in place, one workgroup
__global__ __launch_bounds__(kScanThreads) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t s_wave[4];
    const uint32_t per = (nb + kScanThreads - 1) / kScanThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = b0 + k;
        if (i < nb) s += block_sums[i];
    }
    uint32_t total;
    uint32_t run = block_excl_scan_256(s, &total, s_wave);
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = b0 + k;
        if (i < nb) {
            const uint32_t v = block_sums[i];
            block_sums[i] = run;
            run += v;
        }
    }
}

// FUSED: block_sums holds the raw per-block totals and every block adds up the ones in front of it itself (a few loads per thread for
// the array sizes of this path) — the single-block scan of the totals and its launch are skipped
template <bool EXCLUSIVE, bool FUSED>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                                  const uint32_t* __restrict__ gather, uint32_t n,
                                                                  const uint32_t* __restrict__ block_sums,
                                                                  uint32_t* __restrict__ out) {
    __shared__ uint32_t s_wave[4];
    uint32_t x[kScanItems];
    const uint32_t base = blockIdx.x * kScanTile;
    load_items(in, gather, base, n, x);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += x[k];
    uint32_t total;
    uint32_t block_base;
    if (FUSED) {
        uint32_t pre = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanThreads) pre += block_sums[i];
        (void)block_excl_scan_256(pre, &block_base, s_wave);
    } else {
        block_base = block_sums[blockIdx.x];
    }
    uint32_t run = block_excl_scan_256(s, &total, s_wave) + block_base;
    const uint32_t i0 = base + threadIdx.x * kScanItems;
    uint32_t y[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (EXCLUSIVE) { y[k] = run; run += x[k]; }
        else { run += x[k]; y[k] = run; }
    }
    if (i0 + kScanItems <= n) {
        *reinterpret_cast<uint4*>(out + i0)     = make_uint4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<uint4*>(out + i0 + 4) = make_uint4(y[4], y[5], y[6], y[7]);
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (i0 + k < n) out[i0 + k] = y[k];
    }
}

int scan_impl(hipStream_t s, uint32_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out, bool exclusive,
              void* scratch, size_t scratch_bytes) {
    if (n == 0) return GRUT_OK;
    const uint32_t nb = div_up(n, kScanTile);
    GRUT_REQUIRE(scratch_bytes >= (size_t)nb * sizeof(uint32_t), "scan scratch too small");
    uint32_t* sums = reinterpret_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kScanThreads), 0, s, in, gather, n, sums);
    const bool fused = nb <= 8192u;   // (<= 32 loads per thread for the prefix of the block totals)
    if (!fused) hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(kScanThreads), 0, s, sums, nb);
    if (exclusive) {
        if (fused) hipLaunchKernelGGL((scan_apply_kernel<true, true>), dim3(nb), dim3(kScanThreads), 0, s, in, gather, n, sums, out);
        else hipLaunchKernelGGL((scan_apply_kernel<true, false>), dim3(nb), dim3(kScanThreads), 0, s, in, gather, n, sums, out);
    } else {
        if (fused) hipLaunchKernelGGL((scan_apply_kernel<false, true>), dim3(nb), dim3(kScanThreads), 0, s, in, gather, n, sums, out);
        else hipLaunchKernelGGL((scan_apply_kernel<false, false>), dim3(nb), dim3(kScanThreads), 0, s, in, gather, n, sums, out);
    }
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

// ---------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------
constexpr int kSortThreads = 256;
// keys per lane (ROUNDS): 16 (4096 keys per workgroup) for large arrays; 8 for small ones (<= 2 M keys: the depth sort of 1 M particles
// is 245 workgroups of 4096 keys — fewer than the chip has CUs; with 2048 keys it is 0.082 instead of 0.095 ms, while the 11.5 M-entry tile
// sort loses with them, 0.210 vs 0.164 ms)
#ifndef GRUT_SORT_SMALL
#define GRUT_SORT_SMALL 8
#endif
constexpr int kSortRoundsLarge = 16, kSortRoundsSmall = GRUT_SORT_SMALL;
constexpr uint32_t kSortSmallLimit = 2u << 20;
constexpr int kRadix       = 256;

__device__ __forceinline__ uint32_t eff_count(uint32_t n, const uint32_t* n_dev) {
    if (!n_dev) return n;
    const uint32_t m = *n_dev;
    return m < n ? m : n;
}

// histogram matrix is digit-major: hist[d * nb + b]
template <int kSortRounds>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                  const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
                                                                  uint32_t nb, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_hist[4][kRadix];
    const int wave = threadIdx.x >> 6;
    const uint32_t ne = eff_count(n, n_dev);
#pragma unroll
    for (int w = 0; w < 4; ++w) s_hist[w][threadIdx.x] = 0;
    __syncthreads();
    constexpr int kSortTile = kSortThreads * kSortRounds;
    const uint32_t base = blockIdx.x * kSortTile;
    if (base < ne) {
        // 16 B per lane, 4 loads per thread
#pragma unroll
        for (int k = 0; k < kSortRounds / 4; ++k) {
            const uint32_t i = base + (k * kSortThreads + threadIdx.x) * 4;
            if (i + 4 <= ne) {
                const uint4 v = *reinterpret_cast<const uint4*>(keys + i);
                atomicAdd(&s_hist[wave][(v.x >> shift) & mask], 1u);
                atomicAdd(&s_hist[wave][(v.y >> shift) & mask], 1u);
                atomicAdd(&s_hist[wave][(v.z >> shift) & mask], 1u);
                atomicAdd(&s_hist[wave][(v.w >> shift) & mask], 1u);
            } else {
                for (uint32_t j = i; j < ne && j < i + 4; ++j) atomicAdd(&s_hist[wave][(keys[j] >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
    const uint32_t d = threadIdx.x;
    hist[(size_t)d * nb + blockIdx.x] = s_hist[0][d] + s_hist[1][d] + s_hist[2][d] + s_hist[3][d];
}

// Exclusive scan of each digit's row of per-block counts (in place) + the digit totals: with the 256-entry scan of the
// totals done inside the scatter kernel this replaces a generic three-kernel scan of the whole matrix per pass.
// One workgroup per digit that can occur (2^nbits of them: the rows of the other digits are all zero and stay untouched; their totals
// are zeroed by workgroup 0).
__global__ __launch_bounds__(kSortThreads) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nb, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[4];
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x) totals[threadIdx.x] = 0u;
    uint32_t* row = hist + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += kSortThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_excl_scan_256(v, &total, s_wave);
        if (i < nb) row[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int kSortRounds>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in, uint32_t n,
                                                                     const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
                                                                     uint32_t nb, const uint32_t* __restrict__ hist_scanned,
                                                                     const uint32_t* __restrict__ digit_totals,
                                                                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    constexpr int kSortTile = kSortThreads * kSortRounds, kSortWaveKeys = 64 * kSortRounds;
    // per-wave digit counters; after the ranking phase they are turned into global scatter bases
    __shared__ uint32_t s_cnt[4][kRadix];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ne = eff_count(n, n_dev);
    const uint32_t base = blockIdx.x * kSortTile;
    if (base >= ne) return;
#pragma unroll
    for (int w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();

    uint32_t key[kSortRounds], val[kSortRounds], rank[kSortRounds];
    const uint32_t wbase = base + wave * kSortWaveKeys;
    volatile uint32_t* cnt = s_cnt[wave];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < ne;
        key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[r] = valid ? (vals_in ? vals_in[i] : i) : 0u;   // vals_in == nullptr: the values are the positions themselves (first pass of an iota payload)
    }
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < ne;
        const uint32_t d = (key[r] >> shift) & mask;
        // lanes holding the same digit (wave64 match via 8 ballots)
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t count  = (uint32_t)__popcll(m);
        uint32_t old = 0;
        if (valid) old = cnt[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) cnt[d] = old + count;
        __builtin_amdgcn_wave_barrier();
        rank[r] = old + before;
    }
    __syncthreads();
    // Reorder the block's pairs by digit in LDS first, then write them out in that order: neighbouring lanes then store to
    // neighbouring addresses (a run per digit) instead of 64 unrelated lines per store instruction.
    __shared__ uint32_t s_keys[kSortTile], s_vals[kSortTile];
    __shared__ uint32_t s_dstart[kRadix];  // first slot of digit d in the block-sorted order
    __shared__ uint32_t s_gdelta[kRadix];  // (global base of (digit, block)) - s_dstart[d]
    __shared__ uint32_t s_wave_tot[4];
    {
        const uint32_t d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // exclusive prefix of digit d over the 4 waves
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan_256(run, &total, s_wave_tot);
        const uint32_t digit_base = block_excl_scan_256(digit_totals[d], &total, s_wave_tot);  // keys with a smaller digit, all blocks
        s_dstart[d] = excl;
        s_gdelta[d] = digit_base + hist_scanned[(size_t)d * nb + blockIdx.x] - excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < ne) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = s_dstart[d] + s_cnt[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }
    __syncthreads();
    const uint32_t nvalid = min((uint32_t)kSortTile, ne - base);
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t idx = r * kSortThreads + threadIdx.x;
        if (idx < nvalid) {
            const uint32_t k = s_keys[idx];
            const uint32_t dst = s_gdelta[(k >> shift) & mask] + idx;
            keys_out[dst] = k;
            vals_out[dst] = s_vals[idx];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// One-sweep passes (round 5): ONE kernel per digit instead of three.  The digit totals of ALL passes come from one read of the keys
// (radix_global_hist_kernel); a pass's workgroups then take their tiles in ticket order, rank them as above, publish the tile's digit
// counts and obtain the counts of the tiles in front of them by a decoupled look-back over per-(tile, digit) status words
// (flag in the top two bits: 1 = this tile's count, 2 = count of this tile and every tile before it).  A tile publishes its own count
// BEFORE it looks back, and tiles are handed out by an atomic ticket, so every tile a workgroup waits for is resident and never waits
// for anything itself: no dependence on dispatch order.  The status words carry their own payload (relaxed agent-scope atomics; the
// eight XCDs' L2s are not coherent with each other, so plain loads could spin on a stale line).
// Per pass this removes the histogram kernel (a second read of the keys), the row scan and two dependent launches.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFlagAgg = 1u << 30, kFlagPrefix = 2u << 30, kCountMask = (1u << 30) - 1u;
constexpr int kMaxPasses = 4;
#ifndef GRUT_SORT_LOOKBACK
#define GRUT_SORT_LOOKBACK 8
#endif
constexpr int kLookback = GRUT_SORT_LOOKBACK;

__global__ __launch_bounds__(kSortThreads) void radix_global_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                         int begin_bit, int width, int passes, int end_bit,
                                                                         uint32_t* __restrict__ hist /* [kMaxPasses][kRadix], zeroed */) {
    __shared__ uint32_t s_hist[kMaxPasses][kRadix];
    const uint32_t ne = eff_count(n, n_dev);
    for (int p = 0; p < kMaxPasses; ++p) s_hist[p][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * kSortThreads * 4;
    for (uint32_t i = (blockIdx.x * kSortThreads + threadIdx.x) * 4; i < ne; i += stride) {
        uint32_t k[4];
        if (i + 4 <= ne) {
            const uint4 v = *reinterpret_cast<const uint4*>(keys + i);
            k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
        } else {
            for (int j = 0; j < 4; ++j) k[j] = i + j < ne ? keys[i + j] : 0xFFFFFFFFu;
        }
        const int cnt = (int)min(4u, ne - i);
        for (int j = 0; j < cnt; ++j)
            for (int p = 0; p < passes; ++p) {
                const int bit = begin_bit + p * width;
                const int nbits = (end_bit - bit) < width ? (end_bit - bit) : width;
                atomicAdd(&s_hist[p][(k[j] >> bit) & ((1u << nbits) - 1u)], 1u);
            }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const uint32_t c = s_hist[p][threadIdx.x];
        if (c) atomicAdd(&hist[p * kRadix + threadIdx.x], c);
    }
}

template <int kSortRounds>
__global__ __launch_bounds__(kSortThreads) void radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n,
                                                                      const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
                                                                      const uint32_t* __restrict__ digit_totals /* [kRadix] of this pass */,
                                                                      uint32_t* __restrict__ status /* [tiles][kRadix], zeroed */, uint32_t* __restrict__ ticket,
                                                                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    constexpr int kSortTile = kSortThreads * kSortRounds, kSortWaveKeys = 64 * kSortRounds;
    __shared__ uint32_t s_cnt[4][kRadix];
    __shared__ uint32_t s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ne = eff_count(n, n_dev);
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * kSortTile;
    if (base >= ne) return;

    uint32_t key[kSortRounds], val[kSortRounds], rank[kSortRounds];
    const uint32_t wbase = base + wave * kSortWaveKeys;
    volatile uint32_t* cnt = s_cnt[wave];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < ne;
        key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[r] = valid ? (vals_in ? vals_in[i] : i) : 0u;
    }
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < ne;
        const uint32_t d = (key[r] >> shift) & mask;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t count  = (uint32_t)__popcll(m);
        uint32_t old = 0;
        if (valid) old = cnt[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) cnt[d] = old + count;
        __builtin_amdgcn_wave_barrier();
        rank[r] = old + before;
    }
    __syncthreads();
    __shared__ uint32_t s_keys[kSortTile], s_vals[kSortTile];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_gdelta[kRadix];
    __shared__ uint32_t s_wave_tot[4];
    {
        const uint32_t d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
        // publish this tile's count of digit d, then collect the counts of the tiles in front
        uint32_t before_tiles = 0;
        if (d <= mask) {
            uint32_t* mine = status + (size_t)tile * kRadix + d;
            __hip_atomic_store(mine, (tile == 0 ? kFlagPrefix : kFlagAgg) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tile > 0) {
                // a window of kLookback predecessors per round trip (independent loads), consumed nearest first: the walk's dependent
                // chain is what a pass waits for when a thousand tiles finish ranking together (one status word at a time: 92 us per
                // 11.5 M-pair pass, no faster than the three-kernel pass it replaces)
                int t = (int)tile - 1;
                bool done = false;
                while (!done) {
                    uint32_t w[kLookback];
#pragma unroll
                    for (int k = 0; k < kLookback; ++k)
                        w[k] = t - k >= 0 ? __hip_atomic_load(status + (size_t)(t - k) * kRadix + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagPrefix;
                    int used = 0;
#pragma unroll
                    for (int k = 0; k < kLookback; ++k) {
                        if (!done && used == k) {
                            if ((w[k] >> 30) != 0u) {
                                before_tiles += w[k] & kCountMask;
                                ++used;
                                done = (w[k] & kFlagPrefix) != 0u;   // (tile 0 publishes a prefix: the walk ends there at the latest)
                            }
                        }
                    }
                    t -= used;
                    if (!done && used == 0) __builtin_amdgcn_s_sleep(1);
                }
                __hip_atomic_store(mine, kFlagPrefix | (before_tiles + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan_256(run, &total, s_wave_tot);
        const uint32_t digit_base = block_excl_scan_256(digit_totals[d], &total, s_wave_tot);  // keys with a smaller digit, all tiles
        s_dstart[d] = excl;
        s_gdelta[d] = digit_base + before_tiles - excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < ne) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = s_dstart[d] + s_cnt[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }
    __syncthreads();
    const uint32_t nvalid = min((uint32_t)kSortTile, ne - base);
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t idx = r * kSortThreads + threadIdx.x;
        if (idx < nvalid) {
            const uint32_t k = s_keys[idx];
            const uint32_t dst = s_gdelta[(k >> shift) & mask] + idx;
            keys_out[dst] = k;
            vals_out[dst] = s_vals[idx];
        }
    }
}

}  // namespace

size_t scan_scratch_bytes(uint32_t n) { return (size_t)(div_up(n, kScanTile) + 1) * sizeof(uint32_t); }

int inclusive_scan_u32(hipStream_t s, uint32_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out,
                       void* scratch, size_t scratch_bytes) {
    return scan_impl(s, n, in, gather, out, false, scratch, scratch_bytes);
}

// one-sweep layout of the scratch: [kMaxPasses][kRadix] digit totals | kMaxPasses tickets (+ padding to 64 words) | [passes][tiles][kRadix] status words
constexpr size_t kOnesweepHead = (size_t)kMaxPasses * kRadix + 64;
static bool sort_legacy();
size_t sort_scratch_bytes(uint32_t n) {
    const uint32_t nb = div_up(n, (uint32_t)(kSortThreads * kSortRoundsSmall));   // (the smaller tile: an upper bound for either layout)
    const size_t hist = (size_t)kRadix * nb * sizeof(uint32_t);
    const size_t legacy = hist + kRadix * sizeof(uint32_t) + 256;  // per-block digit counts + digit totals
    const size_t onesweep = (kOnesweepHead + (size_t)kMaxPasses * nb * kRadix) * sizeof(uint32_t);
    // (the one-sweep layout is ~4x the three-kernel one: only handles that opted into it pay for it)
    return (sort_legacy() || legacy > onesweep) ? legacy : onesweep;
}

// Measured (round 5, one box, 1 M Gaussians @ 1080p, profiles/r05_sort_variants.txt): the one-sweep pass of the 11.5 M tile entries takes 92 us -
// what histogram + row scan + scatter of the three-kernel pass take together - and the 1 M-key depth passes 25 us each against 22; with the
// shared histogram kernel (22 us) and the status memset in front the frame is 0.06-0.08 ms SLOWER (2.07-2.09 against 2.00 ms), whatever the
// look-back window (1 / 4 / 8 / 16 status words per round trip: 2.090 / 2.065 / 2.065-2.088 / 2.084 ms): the pass is not waiting for its
// look-back.  The three-kernel passes stay the default; GRUT_SORT_ONESWEEP=1 selects these.
static bool sort_legacy() {
    static const bool v = [] { const char* e = getenv("GRUT_SORT_ONESWEEP"); return !(e && e[0] == '1'); }();
    return v;
}

int sort_pairs_u32(hipStream_t s, uint32_t n, const uint32_t* n_dev, int begin_bit, int end_bit,
                   uint32_t* keys, uint32_t* vals, uint32_t* keys_tmp, uint32_t* vals_tmp,
                   void* scratch, size_t scratch_bytes, uint32_t** out_keys, uint32_t** out_vals, bool vals_iota) {
    *out_keys = keys;
    *out_vals = vals;
    if (n == 0 || end_bit <= begin_bit) return GRUT_OK;
    GRUT_REQUIRE(scratch_bytes >= sort_scratch_bytes(n), "sort scratch too small");
    const bool small = n <= kSortSmallLimit;
    const uint32_t nb = div_up(n, (uint32_t)(kSortThreads * (small ? kSortRoundsSmall : kSortRoundsLarge)));
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* totals = hist + (size_t)kRadix * nb;
    uint32_t *ki = keys, *vi = vals, *ko = keys_tmp, *vo = vals_tmp;
    {
        const int total_bits = end_bit - begin_bit, passes = (total_bits + 7) / 8, width = (total_bits + passes - 1) / passes;
        if (!sort_legacy() && passes <= kMaxPasses) {
            uint32_t* head = reinterpret_cast<uint32_t*>(scratch);
            uint32_t* tickets = head + (size_t)kMaxPasses * kRadix;
            uint32_t* status = head + kOnesweepHead;
            GRUT_HIP(hipMemsetAsync(head, 0, (kOnesweepHead + (size_t)passes * nb * kRadix) * sizeof(uint32_t), s));
            const uint32_t hist_blocks = nb < 1024u ? nb : 1024u;
            hipLaunchKernelGGL(radix_global_hist_kernel, dim3(hist_blocks), dim3(kSortThreads), 0, s, ki, n, n_dev, begin_bit, width, passes, end_bit, head);
            int p = 0;
            for (int bit = begin_bit; bit < end_bit; bit += width, ++p) {
                const int nbits = (end_bit - bit) < width ? (end_bit - bit) : width;
                const uint32_t mask = (1u << nbits) - 1u;
                const uint32_t* vin = (vals_iota && bit == begin_bit) ? nullptr : vi;
                uint32_t* st = status + (size_t)p * nb * kRadix;
                if (small) hipLaunchKernelGGL(radix_onesweep_kernel<kSortRoundsSmall>, dim3(nb), dim3(kSortThreads), 0, s, ki, vin, n, n_dev, bit, mask, head + p * kRadix, st, tickets + p, ko, vo);
                else hipLaunchKernelGGL(radix_onesweep_kernel<kSortRoundsLarge>, dim3(nb), dim3(kSortThreads), 0, s, ki, vin, n, n_dev, bit, mask, head + p * kRadix, st, tickets + p, ko, vo);
                uint32_t* t;
                t = ki; ki = ko; ko = t;
                t = vi; vi = vo; vo = t;
            }
            GRUT_HIP(hipGetLastError());
            *out_keys = ki;
            *out_vals = vi;
            return GRUT_OK;
        }
    }
    // digits of equal width: 13 tile bits are sorted as 7 + 6, not 8 + 5 — the row scan works on 128 + 64 rows instead of 256 + 256
    const int total_bits = end_bit - begin_bit, passes = (total_bits + 7) / 8, width = (total_bits + passes - 1) / passes;
    for (int bit = begin_bit; bit < end_bit; bit += width) {
        const int nbits = (end_bit - bit) < width ? (end_bit - bit) : width;
        const uint32_t mask = (1u << nbits) - 1u;
        if (small) hipLaunchKernelGGL(radix_hist_kernel<kSortRoundsSmall>, dim3(nb), dim3(kSortThreads), 0, s, ki, n, n_dev, bit, mask, nb, hist);
        else hipLaunchKernelGGL(radix_hist_kernel<kSortRoundsLarge>, dim3(nb), dim3(kSortThreads), 0, s, ki, n, n_dev, bit, mask, nb, hist);
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1u << nbits), dim3(kSortThreads), 0, s, hist, nb, totals);
        const uint32_t* vin = (vals_iota && bit == begin_bit) ? nullptr : vi;   // iota payload: generated by the first pass, never read
        if (small) hipLaunchKernelGGL(radix_scatter_kernel<kSortRoundsSmall>, dim3(nb), dim3(kSortThreads), 0, s, ki, vin, n, n_dev, bit, mask, nb, hist, totals, ko, vo);
        else hipLaunchKernelGGL(radix_scatter_kernel<kSortRoundsLarge>, dim3(nb), dim3(kSortThreads), 0, s, ki, vin, n, n_dev, bit, mask, nb, hist, totals, ko, vo);
        uint32_t* t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    GRUT_HIP(hipGetLastError());
    *out_keys = ki;
    *out_vals = vi;
    return GRUT_OK;
}

}  // namespace grut

// ---- C-ABI stage wrappers -------------------------------------------------------------------
extern "C" {

uint64_t grut_sort_scratch_bytes(uint32_t n) { return grut::sort_scratch_bytes(n); }
uint64_t grut_scan_scratch_bytes(uint32_t n) { return grut::scan_scratch_bytes(n); }

int grut_sort_pairs_u32(void* stream, uint32_t n, int begin_bit, int end_bit, uint32_t* keys, uint32_t* values,
                        uint32_t* keys_tmp, uint32_t* values_tmp, void* scratch, uint64_t scratch_bytes,
                        uint32_t** sorted_keys, uint32_t** sorted_values) {
    return grut::sort_pairs_u32(reinterpret_cast<hipStream_t>(stream), n, nullptr, begin_bit, end_bit, keys, values,
                                keys_tmp, values_tmp, scratch, scratch_bytes, sorted_keys, sorted_values);
}

int grut_inclusive_scan_u32(void* stream, uint32_t n, const uint32_t* in, uint32_t* out, void* scratch, uint64_t scratch_bytes) {
    return grut::inclusive_scan_u32(reinterpret_cast<hipStream_t>(stream), n, in, nullptr, out, scratch, scratch_bytes);
}

}  // extern "C"
