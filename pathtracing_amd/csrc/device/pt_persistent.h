// The persistent-wave work loop shared by the traversal kernels.
//
// Work items 0 .. n_items-1 are split into PT_QUEUE_REGIONS contiguous regions,
// each with its own head word (PT_HEAD_STRIDE words apart, so in different
// cache lines and memory channels).  A wave draws chunks of `chunk` items with
// one atomicAdd per chunk from the region of its group (blockIdx % 8: blocks
// of one group share an XCD and its L2) and moves on to the next region when
// that one is drained (see below for how far it goes), so the grid always empties
// every region.  Why:
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

// The end of a launch is a stampede: thousands of waves find their region drained within microseconds of one another, and
// if each then tries every other region, every head word takes one update per wave of the launch, each queued behind a
// thousand others at ~90 updates per microsecond.  A wave therefore tries PT_STEAL_TRIES other regions after its own (which
// ones depends on the block, so that every region gets help from some of every other group) and then leaves; every region
// is emptied by its own group in any case.  Same-box A/B, ms per 16-spp -rb view, 1 / 3 / 7 other regions: atrium 2.15 /
// 2.13 / 2.14, car 0.97 / 0.98 / 1.03.  Launches of fewer than 64 blocks, where a region may have no wave of its own, try all.
#ifndef PT_STEAL_TRIES
#define PT_STEAL_TRIES 3u
#endif

template <class Policy>
__device__ __forceinline__ void persistent_loop(Policy &P, uint32_t n_items, uint32_t *heads, uint32_t chunk, int refill_min) {
    const uint32_t lane = threadIdx.x & 63;
    // region length: a multiple of 64 so that a wave's 64 consecutive items stay in one region
    const uint32_t region_len = (((n_items + PT_QUEUE_REGIONS - 1u) / PT_QUEUE_REGIONS) + 63u) & ~63u;
    const uint32_t home = blockIdx.x & (PT_QUEUE_REGIONS - 1u);
    const uint32_t max_tries = gridDim.x >= 64u ? 1u + PT_STEAL_TRIES : PT_QUEUE_REGIONS;
    bool active = false, more = true;
    uint32_t chunk_next = 0, chunk_end = 0, tries = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (more && __popcll(idle) >= refill_min) {
            if (chunk_next == chunk_end) {
                while (tries < max_tries) {
                    // the region tried after the wave's own: own + 1 + (block / 8) % 7, the one after that the next, ...
                    const uint32_t h = tries == 0 ? home : (home + 1u + ((blockIdx.x >> 3) + tries - 1u) % (PT_QUEUE_REGIONS - 1u)) & (PT_QUEUE_REGIONS - 1u);
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
                if (tries == max_tries) more = false;
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
