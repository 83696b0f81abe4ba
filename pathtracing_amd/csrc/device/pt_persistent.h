// The persistent-wave work loop shared by the traversal kernels.
//
// Work items 0 .. n_items-1 are split into PT_QUEUE_REGIONS contiguous regions,
// each with its own head word (PT_HEAD_STRIDE words apart, so in different
// cache lines and memory channels).  A wave draws chunks of `chunk` items with
// one atomicAdd per chunk from the region of its group (blockIdx % 8: blocks
// of one group share an XCD and its L2) and moves on to the next region when
// that one is drained, so the grid always empties every region.  Why:
//   * a single head word takes only ~90 atomics per microsecond chip-wide;
//     at 64 items per chunk that is a floor of 2.9 ms for 16.7 M items;
//   * small chunks keep the window of items in flight short -- the items are
//     sorted by origin tile and direction, so waves that run at the same time
//     then read the same nodes and triangles -- and the waves finish together;
//   * a region per XCD gives each L2 one contiguous part of the sorted queue.
// Idle lanes are refilled by ballot + prefix popcount as soon as at least
// `refill_min` of them are idle.
//
// Policy::start(item) fetches an item and begins its first unit of work; it
// returns false when the item needed none and is already complete.
// Policy::step(active) advances the lane by one unit when `active` and returns
// whether the lane still has work.  It is called by every lane, active or
// not, on every turn: the loop has a single back edge, which keeps the lane
// state in one set of registers (a `continue` around the step costs a copy of
// the whole state per turn).
// A wave leaves when every region is drained and all of its lanes are done, so
// the grid always drains; surplus waves of a launch find the queue empty and exit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_QUEUE_REGIONS 8u
#define PT_HEAD_STRIDE 64u                    // words between region heads
#define PT_HEAD_WORDS (PT_QUEUE_REGIONS * PT_HEAD_STRIDE)

// The end of a launch: every wave has to find every region drained, and 8 192 waves each adding to each of the eight
// head words queue up for 90 us per word.  The first wave of a group that has seen all eight drained therefore says so in a
// word of its group, and the other waves of the group look there before they try another region: two or three updates
// per wave at the end instead of eight.  The word only ever says what is true for good (heads only grow); a wave that
// looks too early checks the regions itself, as before.
__device__ __forceinline__ bool group_done(uint32_t *heads, uint32_t home) {
    uint32_t v = 0;
    // (an update that changes the word -- it counts the lookers in the bits above the flag -- so that it is not turned into a
    // load that a cache may answer with an old value)
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(heads + home * PT_HEAD_STRIDE + PT_HEAD_STRIDE / 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (__builtin_amdgcn_readfirstlane(v) & 1u) != 0u;
}
__device__ __forceinline__ void group_is_done(uint32_t *heads, uint32_t home) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(heads + home * PT_HEAD_STRIDE + PT_HEAD_STRIDE / 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class Policy>
__device__ __forceinline__ void persistent_loop(Policy &P, uint32_t n_items, uint32_t *heads, uint32_t chunk, int refill_min) {
    const uint32_t lane = threadIdx.x & 63;
    // region length: a multiple of 64 so that a wave's 64 consecutive items stay in one region
    const uint32_t region_len = (((n_items + PT_QUEUE_REGIONS - 1u) / PT_QUEUE_REGIONS) + 63u) & ~63u;
    const uint32_t home = blockIdx.x & (PT_QUEUE_REGIONS - 1u);
    bool active = false, more = true;
    uint32_t chunk_next = 0, chunk_end = 0, tries = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (more && __popcll(idle) >= refill_min) {
            if (chunk_next == chunk_end) {
                while (tries < PT_QUEUE_REGIONS) {
                    if (tries && group_done(heads, home)) { tries = PT_QUEUE_REGIONS; break; }
                    const uint32_t h = (home + tries) & (PT_QUEUE_REGIONS - 1u);
                    const uint32_t lo = h * region_len < n_items ? h * region_len : n_items;
                    const uint32_t hi = lo + region_len < n_items ? lo + region_len : n_items;
                    uint32_t base = 0;
                    if (lane == 0 && lo < hi) base = atomicAdd(heads + h * PT_HEAD_STRIDE, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (lo < hi && base < hi - lo) {               // n_items < 2^31 and at most waves * chunk of overshoot: no wrap
                        chunk_next = lo + base;
                        chunk_end = lo + base + chunk < hi ? lo + base + chunk : hi;
                        break;
                    }
                    tries++;
                }
                if (tries == PT_QUEUE_REGIONS) { more = false; group_is_done(heads, home); }
            }
            const uint32_t want = (uint32_t)__popcll(idle), have = chunk_end - chunk_next;
            const uint32_t take = want < have ? want : have, base = chunk_next;
            chunk_next += take;
            if (!active) {
                const uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (k < take) active = P.start(base + k);
            }
        }
        if (!more && !__any(active)) break;
        active = P.step(active);
    }
}
