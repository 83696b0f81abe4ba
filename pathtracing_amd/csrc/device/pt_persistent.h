// The persistent-wave work loop shared by the traversal kernels.
//
// Work items 0 .. n_items-1 are drawn from *head in chunks: one atomicAdd per
// chunk per wave, so the head word is touched once per `chunk` items and not
// once per refill (a single word takes only ~90 atomics per microsecond
// chip-wide).  Idle lanes are refilled by ballot + prefix popcount as soon as
// at least `refill_min` of them are idle.
//
// Policy::start(item) fetches an item and begins its first unit of work; it
// returns false when the item needed none and is already complete.
// Policy::turn() advances the lane by one unit and returns whether the lane
// still has work (it may move on to further rays of the same item).
// A wave leaves when the queue is drained and all of its lanes are done, so the
// grid always drains; surplus waves of a launch find the queue empty and exit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <class Policy>
__device__ __forceinline__ void persistent_loop(Policy &P, uint32_t n_items, uint32_t *head, uint32_t chunk, int refill_min) {
    const uint32_t lane = threadIdx.x & 63;
    bool active = false, more = true;
    uint32_t chunk_next = 0, chunk_end = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (more && __popcll(idle) >= refill_min) {
            if (chunk_next == chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_next = base < n_items ? base : n_items;
                chunk_end = base + chunk < n_items ? base + chunk : n_items;
                if (chunk_next == chunk_end) more = false;
            }
            const uint32_t want = (uint32_t)__popcll(idle), have = chunk_end - chunk_next;
            const uint32_t take = want < have ? want : have, base = chunk_next;
            chunk_next += take;
            if (!active) {
                const uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (k < take) active = P.start(base + k);
            }
        }
        // a single back edge: when no lane is active (only while the last chunk is being handed out) the step
        // below runs with every lane predicated off rather than being jumped over, which keeps the lane state in
        // one set of registers around the loop
        if (!more && !__any(active)) break;
        active = P.step(active);
    }
}
