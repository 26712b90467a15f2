// device_lists.h -- per-wave candidate lists shared by the scan and select kernels.
//
// A candidate is (ordered key << 32) | row; unsigned order is (key, row) order.
// WaveList keeps the kp best candidates a wave has seen, sorted ascending:
//  * kp <= 64: in registers, entry i in lane i.  Insert = one ballot, a popcount
//    and a one-lane wave shift (DPP wave_shr:1), about a dozen VALU ops;
//  * kp  > 64: in LDS (64 entries per chunk), shifted chunk by chunk.
// Only the owning wave touches its list.  DS operations of one wave complete in
// issue order, so no s_barrier is needed; the wave barriers stop the compiler
// from moving one lane's store across another lane's load (it reasons per
// thread).  Plain shared pointers keep the accesses ds_* instructions -- a
// volatile generic pointer turns them into flat_* ops that drain the whole
// global-load queue.
#pragma once

#include "kernels.h"

namespace szg {

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return ((uint64_t)hi << 32) | lo;
}

// value of lane `src`, src wave-uniform: v_readlane_b32 through a scalar register -- a few cycles, where
// __shfl (ds_bpermute_b32 through the LDS crossbar) costs ~100.  The list code broadcasts once per candidate that
// passes the pre-filter and once per insert; while a list is still filling (every wave's first kp rows; a wave of
// a 125 K-row shard never sees more than that) these broadcasts were most of the selection's cost.
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// lane i receives lane i-1's value (lane 0 keeps its own)
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138,
                                                              0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32),
                                                              (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF,
                                                              false);
    return ((uint64_t)hi << 32) | lo;
}

struct WaveList {
    uint64_t *lds;    // this wave's kp entries in LDS (always allocated: the block merge reads it)
    uint64_t mine;    // register form: entry `lane` (kp <= 64)
    uint64_t worst;   // current kp-th best (wave-uniform)
    float worst_key;  // its key as a float (+inf while the list is not full): the scan's cheap pre-filter --
                      // a row whose key is above it cannot enter the list; anything else (NaN included)
                      // takes the exact 64-bit path
    int kp;
    bool in_regs;

    __device__ __forceinline__ void init(uint64_t *lds_, int kp_, int lane)
    {
        lds = lds_;
        kp = kp_;
        in_regs = kp_ <= 64;
        mine = kInvalidCand;
        worst = kInvalidCand;
        worst_key = __builtin_inff();
        if (!in_regs)
            for (int i = lane; i < kp; i += 64) lds[i] = kInvalidCand;
    }

    // c is wave-uniform and c < worst
    __device__ __forceinline__ void insert(uint64_t c, int lane)
    {
        if (in_regs) {
            const int pos = __popcll(__ballot(mine < c));  // entries are unique, sorted ascending
            const uint64_t up = wave_shr1_u64(mine);
            mine = lane < pos ? mine : (lane == pos ? c : up);
            worst = readlane_u64(mine, kp - 1);
            return;
        }
        int pos = 0;
        for (int base = 0; base < kp; base += 64) {
            const int e = base + lane;
            const bool lt = e < kp && lds[e] < c;
            pos += __popcll(__ballot(lt));
        }
        // shift [pos, kp-2] one slot up, highest chunk first
        for (int base = ((kp - 1) / 64) * 64; base >= 0; base -= 64) {
            if (base + 63 <= pos) break;
            const int e = base + lane;
            const bool mv = e > pos && e < kp;
            uint64_t v = 0;
            if (mv) v = lds[e - 1];
            __builtin_amdgcn_wave_barrier();
            if (mv) lds[e] = v;
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) lds[pos] = c;
        __builtin_amdgcn_wave_barrier();
        worst = lds[kp - 1];
    }

    // offer the candidates of the lanes with `have` set (at most one per lane)
    __device__ __forceinline__ void offer(bool have, uint64_t c, int lane)
    {
        uint64_t m = __ballot(have && c < worst);
        if (!m) return;
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint64_t cc = readlane_u64(c, src);
            if (cc < worst) insert(cc, lane);
        }
        worst_key = worst == kInvalidCand ? __builtin_inff() : key_from_ordered((uint32_t)(worst >> 32));
    }

    // make the LDS copy current (before the block-wide merge)
    __device__ __forceinline__ void flush(int lane)
    {
        if (in_regs && lane < kp) lds[lane] = mine;
    }
};

// number of entries of sorted list[0..n) that are < c
__device__ __forceinline__ int lower_count(const uint64_t *list, int n, uint64_t c)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// block-wide k-select: rank-merge the waves' sorted lists (entries are unique)
// into one sorted list of kp entries at `out`.  Call with all threads of the block.
__device__ __forceinline__ void block_merge_lists(const uint64_t *lists, int nwaves, int kp,
                                                  uint64_t *out, int tid, int nthreads)
{
    for (int i = tid; i < kp; i += nthreads) out[i] = kInvalidCand;
    __syncthreads();
    const int total = nwaves * kp;
    for (int it = tid; it < total; it += nthreads) {
        const int w = it / kp;
        const int i = it - w * kp;
        const uint64_t c = lists[it];
        if (c == kInvalidCand) continue;
        int rank = i;
        for (int w2 = 0; w2 < nwaves; w2++) {
            if (w2 == w) continue;
            rank += lower_count(lists + (size_t)w2 * kp, kp, c);
            if (rank >= kp) break;
        }
        if (rank < kp) out[rank] = c;
    }
}

}  // namespace szg
