// Kernels of the fp64 reference-semantics path: one thread per sub-ray for the
// primary pass (all of castRay at depth 0), one thread per (pixel, sample) for
// the GI pass, an ordered accumulation and the final quantisation.  The kernels
// that search the tree run as at most PT_EXACT_WAVES one-wave blocks that stride
// over their work: each launched block owns a region of the stack overflow buffer.
// Build with -ffp-contract=off.
#include "pt_exact.h"
#include "pt_frame.h"
#include "pt_packet.h"

// 8x8 pixel tiles: one wave = one tile, so that the rays of a wave stay close.
PT_DEV bool tile_pixel(int linear, int size, int &i, int &j) {
    int tiles_x = (size + 7) >> 3;
    int tile = linear >> 6, lane = linear & 63;
    int ti = tile / tiles_x, tj = tile - ti * tiles_x;
    i = ti * 8 + (lane >> 3);
    j = tj * 8 + (lane & 7);
    return i < size && j < size;
}

PT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The word a block adds its count of ray class k to: one set of class words per group of blocks (see RC_HEADS), summed by the host.
PT_DEV unsigned long long *count_word(unsigned long long *counters, int k) {
    return counters + RC_HEADS + (blockIdx.x & (RC_HEAD_REGIONS - 1u)) * RC_HEAD_STRIDE + 1u + (unsigned)k;
}
#ifndef PT_BAND_INTERLEAVE
#define PT_BAND_INTERLEAVE 32u                   // 0: a band is a contiguous eighth of the blocks; n: runs of n blocks dealt to the bands in turn
#endif
// The next work block of a persistent kernel whose nblocks blocks are handed out from the eight per-band counters in word
// `word` of the bands' lines (RC_HEADS): from the band of the block's group first, then from up to `steal` other bands
// (which ones depends on the block; a launch of fewer than 64 blocks tries all).  false: nothing left for this wave.
// Every lane of the wave must call it; `tries` is the wave's progress through its bands, 0 at the start.
PT_DEV bool draw_block(unsigned long long *counters, uint32_t word, uint32_t nblocks, uint32_t steal, uint32_t &tries, uint32_t &blk) {
    const uint32_t band = (nblocks + RC_HEAD_REGIONS - 1u) / RC_HEAD_REGIONS, home = blockIdx.x & (RC_HEAD_REGIONS - 1u);
    const uint32_t max_tries = gridDim.x >= 64u ? 1u + steal : RC_HEAD_REGIONS;
    while (tries < max_tries) {
        const uint32_t h = tries == 0 ? home : (home + 1u + ((blockIdx.x >> 3) + tries - 1u) % (RC_HEAD_REGIONS - 1u)) & (RC_HEAD_REGIONS - 1u);
#if PT_BAND_INTERLEAVE
        // band h: the runs of PT_BAND_INTERLEAVE consecutive blocks number h, h + 8, h + 16, ...
        const uint32_t runs = (nblocks + PT_BAND_INTERLEAVE - 1u) / PT_BAND_INTERLEAVE, mine = runs > h ? (runs - h + RC_HEAD_REGIONS - 1u) / RC_HEAD_REGIONS : 0u;
        const uint32_t lo = 0, hi = mine * PT_BAND_INTERLEAVE;
#else
        const uint32_t lo = h * band < nblocks ? h * band : nblocks, hi = lo + band < nblocks ? lo + band : nblocks;
#endif
        uint32_t at = 0;
        if ((threadIdx.x & 63) == 0 && lo < hi) at = (uint32_t)atomicAdd(&counters[RC_HEADS + h * RC_HEAD_STRIDE + word], 1ull);
        at = __builtin_amdgcn_readfirstlane(at);
        if (lo < hi && at < hi - lo) {
#if PT_BAND_INTERLEAVE
            blk = ((at / PT_BAND_INTERLEAVE) * RC_HEAD_REGIONS + h) * PT_BAND_INTERLEAVE + at % PT_BAND_INTERLEAVE;
            if (blk >= nblocks) continue;                // the last run is short: draw again
#else
            blk = lo + at;
#endif
            return true;
        }
        tries++;
    }
    (void)band;
    return false;
}
PT_DEV void flush_counts(const RayCounts &rc, unsigned long long *counters) {
#pragma unroll
    for (int k = 0; k < RC_WORK_PRIMARY; k++) {          // the ray classes; the words after them are not ray counts
        uint32_t s = wave_sum(rc.n[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(count_word(counters, k), (unsigned long long)s);
    }
}

// Which pixel and sub-ray lane `lane` of work block `blk` of the primary search handles.  A work block is a packet of
// camera rays that the wave searches together (pt_packet.h): with -rb the nine sub-rays of each of 7 neighbouring pixels
// of one image row (63 lanes, one idle; the last packet of a row may hold fewer pixels), otherwise an 8x8 pixel tile.
PT_DEV bool packet_pixel(const ViewParams &V, uint32_t blk, uint32_t lane, int &i, int &j, int &sub) {
    if (V.rb) {
        const uint32_t segs = ((uint32_t)V.size + 6u) / 7u;
        const uint32_t row = blk / segs, seg = blk - row * segs, px = lane / 9u;
        sub = (int)(lane - px * 9u);
        i = (int)row; j = (int)(seg * 7u + px);
        return lane < 63u && j < V.size;
    }
    sub = 4;
    return tile_pixel((int)(blk * 64u + lane), V.size, i, j);
}
static inline unsigned packet_blocks(int size, int rb) {
    return rb ? (unsigned)size * (((unsigned)size + 6u) / 7u) : (unsigned)(tile_threads(size) / 64);
}

// The camera ray of sub-ray `sub` of pixel (i, j): the offsets as the reference's loops produce them,
// -o, -o + o, (-o + o) + o  (main.cpp:116-117).
PT_DEV RayD sub_ray(const ViewParams &V, int i, int j, int sub) {
    const double o0 = -V.off, o1 = o0 + V.off, o2 = o1 + V.off;
    const int xi = sub % 3, yi = sub / 3;
    const double x = xi == 0 ? o0 : (xi == 1 ? o1 : o2), y = yi == 0 ? o0 : (yi == 1 ? o1 : o2);
    return camera_ray(d3(V.eye[0], V.eye[1], V.eye[2]), V.zoom, V.sa, V.ca, V.size, i, j, x, y);
}

// Before the primary search of a view: what the packet test needs of every entry of the reference lists for this
// camera (pt_packet.h), four arrays of n_ref float4.
__global__ void k_packet_prep(DevScene S, ViewParams V, float4 *forms) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nr = (size_t)S.n_ref;
    if (r >= nr) return;
    const double *tri = S.refTri + r * 9;
    const PacketForms F = packet_forms(d3(V.ca, 0, V.sa), d3(V.sa, 0, -V.ca), d3(0, 1, 0), d3(V.eye[0], V.eye[1], V.eye[2]),
                                       dload(tri), dload(tri + 3), dload(tri + 6));
    forms[r] = F.det; forms[nr + r] = F.un; forms[2 * nr + r] = F.vn; forms[3 * nr + r] = F.w;
}

// Primary pass, first kernel: the search of castRay at depth 0 (main.cpp:43 / mainOMP.cpp:39) for every sub-ray, one
// packet per wave.  Writes the hit record of each sub-ray and, for the mainOMP model, whether the second search of
// mainOMP.cpp:60 has to be run (SearchNote); k_primary_shade does the rest of castRay from that record.
// The kernel's arguments are one struct, read where they are needed through the kernel-argument pointer made opaque to
// the compiler: what the walk does not need (the view, the output arrays) then does not sit in scalar registers while it runs.
#ifndef PT_PRIMARY_STEAL
#define PT_PRIMARY_STEAL 3u                       // other bands a wave of the search kernel tries after its own
#endif
struct PrimaryArgs { DevScene S; ViewParams V; FrameBuffers F; SubBuffers B; uint32_t nblocks; };
typedef const PrimaryArgs __attribute__((address_space(4))) *PrimaryArgsU;
PT_DEV PrimaryArgsU primary_args() {
    PrimaryArgsU p = (PrimaryArgsU)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
PT_DEV ViewParams primary_view(PrimaryArgsU a) {
    ViewParams V;
    V.eye[0] = a->V.eye[0]; V.eye[1] = a->V.eye[1]; V.eye[2] = a->V.eye[2];
    V.zoom = a->V.zoom; V.sa = a->V.sa; V.ca = a->V.ca; V.size = a->V.size; V.rb = a->V.rb; V.off = a->V.off;
    V.seed = a->V.seed; V.view = a->V.view; V.model = a->V.model; V.part = a->V.part; V.n_parts = a->V.n_parts;
    return V;
}
template <int MODEL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_PKT_WAVES, PT_PKT_WAVES))) k_primary_search(PrimaryArgs A) {
    extern __shared__ double stack_lds[];
    const uint32_t lane = threadIdx.x & 63;
    uint32_t nprimary = 0;
    PktClk C;
    C.reset();
    // persistent waves: a launched block (it owns a region of the stack overflow buffer) draws packets from a counter:
    // from the counter of the image band of its group first (blockIdx % 8: the blocks of a group share an XCD and its L2, and
    // with it the band's part of the packet test's table), then from the other bands' until every band is drained
    uint32_t tries = 0;
    for (;;) {
        const PrimaryArgsU a0 = primary_args();
        uint32_t blk = 0;
        if (!draw_block((unsigned long long *)a0->F.counters, 0u, a0->nblocks, PT_PRIMARY_STEAL, tries, blk)) break;
        int i = 0, j = 0, sub = 4;
        bool live;
        RayD ray;
        PacketBoundsF PB;
        {
            const ViewParams V = primary_view(a0);
            live = packet_pixel(V, blk, lane, i, j, sub) && own_row(V.part, V.n_parts, i);
            if (!__ballot(live)) continue;
            ray = sub_ray(V, i, j, sub);             // lanes that are not live get a ray too: same origin, never searched
            PB = packet_bounds_f(packet_bounds(V.sa, V.ca, ray.dir, live));
        }
        HitD h;
        h.tmin = (double)FLT_MAX; h.tri = -1; h.u = 0; h.v = 0;
        SearchNote note;
        const bool ok = MODEL == PT_MODEL_OMP ? packet_search<true>(A.S, ray, live, PB, A.B.forms, h, &note, stack_lds, C)
                                              : packet_search<false>(A.S, ray, live, PB, A.B.forms, h, nullptr, stack_lds, C);
        const PrimaryArgsU a1 = primary_args();
        const int size = a1->V.size, rb = a1->V.rb;
        const size_t p = (size_t)i * size + j, slot = rb ? p * 9 + sub : p;
        const bool again = MODEL == PT_MODEL_OMP && live && ok && note.may_find_closer(h.tmin);
        if (MODEL == PT_MODEL_OMP) {                 // the few sub-rays whose second search has to be run go on a list (k_primary_again)
            const unsigned long long am = __ballot(again);
            if (am) {
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(am)) base = (uint32_t)atomicAdd(&a1->F.counters[RC_AGAIN], (unsigned long long)__popcll(am));
                base = __builtin_amdgcn_readlane(base, __builtin_ctzll(am));
                if (again) a1->B.again_list[base + __popcll(am & ((1ull << lane) - 1ull))] = (uint32_t)slot;
            }
        }
        if (!live) continue;
        nprimary++;
        a1->B.found[slot] = ok ? 1 : 0; a1->B.tri[slot] = ok ? h.tri : -1;
        double *tuv = a1->B.tuv + 3 * slot;
        tuv[0] = ok ? h.tmin : -1.0; tuv[1] = h.u; tuv[2] = h.v;
        a1->B.again[slot] = again ? 1 : 0;
        C.lap(5);
        C.packet_done();
    }
    C.flush();
    const uint32_t total = wave_sum(nprimary);
    // (into the word of its group: 4 096 waves ending on one word would queue up for 45 microseconds)
    if (lane == 0 && total) atomicAdd(count_word(primary_args()->F.counters, RC_PRIMARY), (unsigned long long)total);
}

// castRay at depth 0 after its search (main.cpp:44-78 / mainOMP.cpp:40-78) up to its mirror bounce, for the sub-ray in
// `slot`, whose hit record k_primary_search wrote.  A reflective hit leaves a = 0.6 txt albedo direct / pi in `paint`
// and its mirror ray's direction in `mdir`; k_mirror_exact adds the bounce.  The sub-ray results are folded into the
// frame buffers by k_primary_combine in the reference's order.
// SEARCH = false: mainOMP model, for sub-rays whose second search is known to find nothing.
template <int MODEL, bool SEARCH>
PT_DEV void shade_sub_ray(const DevScene &S, const ViewParams &V, const SubBuffers &B, StackD &st, RayCounts &rc, int i, int j, int sub, size_t slot) {
    D3 paint;
    int mirror = 0;
    // whose G-buffer candidate k_primary_combine reads: every sub-ray's for main.cpp (the last one wins), the centre ray's
    // for mainOMP.cpp; of the others only the hit point is kept, and only where the mirror pass starts its ray from it
    const bool g_read = MODEL == PT_MODEL_MAIN || !V.rb || sub == 4;
    if (!B.found[slot]) {
        if (g_read) dstore(B.g + 9 * slot + 6, d3(-99999, -99999, -99999));      // (colour and normal of a miss are not read)
        paint = d3(S.bg[0], S.bg[1], S.bg[2]);
    } else {
        const RayD ray = sub_ray(V, i, j, sub);
        CastHit c;
        c.h.tri = B.tri[slot]; c.h.tmin = B.tuv[3 * slot]; c.h.u = B.tuv[3 * slot + 1]; c.h.v = B.tuv[3 * slot + 2];
        cast_lights<MODEL, SEARCH>(S, ray, st, rc, c, MODEL == PT_MODEL_OMP && !B.again[slot]);
        // the G-buffer candidate goes to memory as soon as it is known
        if (g_read) { dstore(B.g + 9 * slot, dmul(c.txt, c.alb)); dstore(B.g + 9 * slot + 3, c.nrm); }
        if (g_read || c.reft) dstore(B.g + 9 * slot + 6, c.hit);
        if (MODEL == PT_MODEL_OMP && SEARCH) {       // a second search that did find something closer moved the record (mainOMP.cpp:60)
            B.tri[slot] = c.h.tri;
            B.tuv[3 * slot] = c.h.tmin; B.tuv[3 * slot + 1] = c.h.u; B.tuv[3 * slot + 2] = c.h.v;
        }
        if (c.reft) {
            mirror = 1;
            paint = ddivs(dmul(dmul(dscale(0.6, c.txt), c.alb), c.direct), 3.14159265358979323846);
            dstore(B.mdir + 3 * slot, dreflect(ray.dir, c.nrm));
            rc.n[RC_MIRROR]++;
        } else {
            paint = cast_value<MODEL>(c, 0, d3(0, 0, 0));
        }
    }
    B.mirror[slot] = mirror;
    dstore(B.paint + 3 * slot, paint);
}

// Primary pass, second kernel of the main.cpp model, one thread per sub-ray: the shadow rays are searches, so the
// kernel runs as resident one-wave blocks with a stack each.  Without -rb a wave is one 8x8 pixel tile; with -rb the
// nine sub-rays of a pixel sit in adjacent lanes (7 pixels x 9 = 63 lanes of a wave, one lane idle).
template <int MODEL>
__global__ void __launch_bounds__(64) k_primary_shade(DevScene S, ViewParams V, FrameBuffers F, SubBuffers B, uint32_t nblocks) {
    extern __shared__ double stack_lds[];
    StackD st = make_stack(S, stack_lds);
    RayCounts rc;
#pragma unroll
    for (int k = 0; k < RC_N; k++) rc.n[k] = 0;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t blk = 0, tries = 0;;) {
        if (!draw_block(F.counters, 8u, nblocks, PT_PRIMARY_STEAL, tries, blk)) break;      // (word 8 of the bands' lines)
        uint32_t lin;
        int sub = 4;
        bool live = true;
        if (V.rb) {
            const uint32_t item = blk * 63u + lane;
            lin = item / 9u;
            sub = (int)(item - lin * 9u);
            live = lane < 63u;
        } else {
            lin = blk * 64u + lane;
        }
        int i = 0, j = 0;
        live = live && tile_pixel((int)lin, V.size, i, j) && own_row(V.part, V.n_parts, i);
        if (!live) continue;
        const size_t p = (size_t)i * V.size + j;
        shade_sub_ray<MODEL, true>(S, V, B, st, rc, i, j, sub, V.rb ? p * 9 + sub : p);
    }
    flush_counts(rc, F.counters);
}

// The same for the mainOMP model, whose light loop searches nothing new: the second search of mainOMP.cpp:60 is known
// to find nothing for all but the few sub-rays on the list (SearchNote), so this kernel has no search, no stack, and
// runs at full occupancy, one thread per sub-ray in memory order; k_primary_again does the listed ones.
__global__ void __launch_bounds__(256) k_primary_shade_omp(DevScene S, ViewParams V, FrameBuffers F, SubBuffers B) {
    RayCounts rc;
#pragma unroll
    for (int k = 0; k < RC_N; k++) rc.n[k] = 0;
    const size_t nsub = V.rb ? 9 : 1, total = (size_t)V.size * V.size * nsub;
    for (size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += (size_t)gridDim.x * blockDim.x) {
        const size_t p = slot / nsub;
        const int i = (int)(p / (size_t)V.size), j = (int)(p - (size_t)i * V.size);
        if (own_row(V.part, V.n_parts, i) && !B.again[slot]) {
            StackD none;
            shade_sub_ray<PT_MODEL_OMP, false>(S, V, B, none, rc, i, j, V.rb ? (int)(slot - p * nsub) : 4, slot);
        }
    }
    // one update of the view's counters per block: updates of one word run at some 90 per microsecond on the whole chip,
    // so one per wave of a launch of 37 000 waves would take longer than the shading
    __shared__ uint32_t sums[RC_WORK_PRIMARY];
    if (threadIdx.x < RC_WORK_PRIMARY) sums[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RC_WORK_PRIMARY; k++) {
        const uint32_t w = wave_sum(rc.n[k]);
        if ((threadIdx.x & 63) == 0 && w) atomicAdd(&sums[k], w);
    }
    __syncthreads();
    if (threadIdx.x < RC_WORK_PRIMARY && sums[threadIdx.x]) atomicAdd(count_word(F.counters, (int)threadIdx.x), (unsigned long long)sums[threadIdx.x]);
}
__global__ void __launch_bounds__(64) k_primary_again(DevScene S, ViewParams V, FrameBuffers F, SubBuffers B) {
    extern __shared__ double stack_lds[];
    StackD st = make_stack(S, stack_lds);
    RayCounts rc;
#pragma unroll
    for (int k = 0; k < RC_N; k++) rc.n[k] = 0;
    const size_t n = (size_t)F.counters[RC_AGAIN], nsub = V.rb ? 9 : 1;
    for (size_t k = (size_t)blockIdx.x * 64 + threadIdx.x; k < n; k += (size_t)gridDim.x * 64) {
        const size_t slot = B.again_list[k], p = slot / nsub;
        const int i = (int)(p / (size_t)V.size), j = (int)(p - (size_t)i * V.size);
        shade_sub_ray<PT_MODEL_OMP, true>(S, V, B, st, rc, i, j, V.rb ? (int)(slot - p * nsub) : 4, slot);
    }
    flush_counts(rc, F.counters);
}

// The mirror bounce of the reflective primary hits (main.cpp:71-77 / mainOMP.cpp:68-76): same mapping of sub-rays to
// lanes as k_primary_exact; paint = a + 0.4 * (0 + castRay(reflected ray)).
template <int MODEL>
__global__ void __launch_bounds__(64) k_mirror_exact(DevScene S, ViewParams V, FrameBuffers F, SubBuffers B, uint32_t nblocks) {
    extern __shared__ double stack_lds[];
    StackD st = make_stack(S, stack_lds);
    RayCounts rc;
#pragma unroll
    for (int k = 0; k < RC_N; k++) rc.n[k] = 0;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t blk = 0, tries = 0;;) {
        if (!draw_block(F.counters, 9u, nblocks, PT_PRIMARY_STEAL, tries, blk)) break;      // (word 9 of the bands' lines)
        uint32_t lin;
        int sub = 0;
        bool live = true;
        if (V.rb) {
            const uint32_t item = blk * 63u + lane;
            lin = item / 9u;
            sub = (int)(item - lin * 9u);
            live = lane < 63u;
        } else {
            lin = blk * 64u + lane;
        }
        int i = 0, j = 0;
        live = live && tile_pixel((int)lin, V.size, i, j) && own_row(V.part, V.n_parts, i);
        if (!live) continue;
        const size_t p = (size_t)i * V.size + j, slot = V.rb ? p * 9 + sub : p;
        if (!B.mirror[slot]) continue;
        const RayD rray = make_ray(dload(B.g + 9 * slot + 6), dload(B.mdir + 3 * slot));
        D3 ref = d3(0, 0, 0);
        ref = dadd(ref, cast_secondary<MODEL>(S, rray, MODEL == PT_MODEL_MAIN ? 1 : -1, st, rc));
        dstore(B.paint + 3 * slot, dadd(dload(B.paint + 3 * slot), dscale(0.4, ref)));
    }
    flush_counts(rc, F.counters);
}

// Sum of the sub-rays in the reference's order, /9 for a bundle (main.cpp:122-125).  G-buffer owner:
// MAIN -- every sub-ray writes, the last wins, a miss only writes mhit (main.cpp:44-47,68-70);
// OMP -- the centre ray only (mainOMP.cpp:124-127).
// A pixel of a row that another worker renders gets the record of a ray that was never shot:
// black, and the G-buffer mark that keeps the GI pass away from it.
template <int MODEL>
__global__ void k_primary_combine(FrameBuffers F, SubBuffers B, int size, int rb, uint32_t part, uint32_t n_parts) {
    size_t P = (size_t)size * size, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (!own_row(part, n_parts, (int)(p / size))) {
        const D3 zero = d3(0, 0, 0);
        dstore(F.mpnt + 3 * p, zero);
        dstore(F.gclr + 3 * p, zero); dstore(F.gnrm + 3 * p, zero); dstore(F.ghit + 3 * p, d3(-99999, -99999, -99999));
        F.rec_found[p] = 0; F.rec_tri[p] = -1;
        F.rec_t[p] = 0; F.rec_u[p] = 0; F.rec_v[p] = 0;
        return;
    }
    const int nsub = rb ? 9 : 1;
    D3 paint = d3(0, 0, 0), clr = d3(0, 0, 0), nrm = d3(0, 0, 0), hit = d3(0, 0, 0);
    size_t owner = p * nsub;
    for (int s = 0; s < nsub; s++) {
        size_t k = p * nsub + s;
        paint = dadd(paint, dload(B.paint + 3 * k));
        bool writes = MODEL == PT_MODEL_MAIN || !rb || s == 4;
        if (writes) {
            owner = k;
            if (B.found[k]) { clr = dload(B.g + 9 * k); nrm = dload(B.g + 9 * k + 3); }
            hit = dload(B.g + 9 * k + 6);
        }
    }
    if (rb) paint = ddivs(paint, 9.0);
    dstore(F.mpnt + 3 * p, paint);
    dstore(F.gclr + 3 * p, clr); dstore(F.gnrm + 3 * p, nrm); dstore(F.ghit + 3 * p, hit);
    F.rec_found[p] = B.found[owner]; F.rec_tri[p] = B.tri[owner];
    F.rec_t[p] = B.tuv[3 * owner]; F.rec_u[p] = B.tuv[3 * owner + 1]; F.rec_v[p] = B.tuv[3 * owner + 2];
}

// globalIllumination, main.cpp:85-101 / mainOMP.cpp:85-101, one sample per thread.
// Thread t of the launch handles sample s0 + t / tile_threads and the pixel of
// t % tile_threads; contrib holds r1 * castRay(iray, -1) in sample order.
template <int MODEL>
__global__ void __launch_bounds__(64) k_gi_exact(DevScene S, ViewParams V, FrameBuffers F, int s0, int tthreads, uint32_t nblocks) {
    extern __shared__ double stack_lds[];
    StackD st = make_stack(S, stack_lds);
    RayCounts rc;
#pragma unroll
    for (int k = 0; k < RC_N; k++) rc.n[k] = 0;
    for (uint32_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        int lin = (int)(blk * 64u + threadIdx.x);
        int sl = lin / tthreads;
        int i, j;
        if (!tile_pixel(lin - sl * tthreads, V.size, i, j)) continue;
        size_t P = (size_t)V.size * V.size, p = (size_t)i * V.size + j;
        D3 hit = dload(F.ghit + 3 * p);
        D3 out = d3(0, 0, 0);
        if (!(hit.x == -99999)) {
            D3 nrm = dload(F.gnrm + 3 * p), Nt, Nb;
            coord_system(nrm, Nt, Nb);
            double r1;
            D3 sm = sample_hemisphere(V.seed, V.view, (uint32_t)p, (uint32_t)(s0 + sl), r1);
            D3 world = d3(sm.x * Nb.x + sm.y * nrm.x + sm.z * Nt.x,
                          sm.x * Nb.y + sm.y * nrm.y + sm.z * Nt.y,
                          sm.x * Nb.z + sm.y * nrm.z + sm.z * Nt.z);
            RayD iray = make_ray(hit, world);
            rc.n[RC_GI]++;
            out = dscale(r1, cast_secondary<MODEL>(S, iray, -1, st, rc));
        }
        dstore(F.contrib + ((size_t)sl * P + p) * 3, out);
    }
    flush_counts(rc, F.counters);
}

// indirect = indirect + r1 * L, in sample order (main.cpp:98).
__global__ void k_gi_accumulate(FrameBuffers F, int size, int nb, int first_batch) {
    size_t P = (size_t)size * size, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    D3 acc = first_batch ? d3(0, 0, 0) : dload(F.gi_acc + 3 * p);
    for (int s = 0; s < nb; s++) acc = dadd(acc, dload(F.contrib + ((size_t)s * P + p) * 3));
    dstore(F.gi_acc + 3 * p, acc);
}

// mpnt + 2*mclr*GI/N (main.cpp:134, mainOMP.cpp:134), then the 8-bit BGR
// quantisation of main.cpp:142-144 / mainOMP.cpp:194-196.
__global__ void k_resolve(FrameBuffers F, int size, int gi, int nsamples) {
    size_t P = (size_t)size * size, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    D3 paint = dload(F.mpnt + 3 * p);
    if (gi) {
        D3 G = dload(F.gi_acc + 3 * p), clr = dload(F.gclr + 3 * p);
        paint = dadd(paint, ddivs(dmul(dscale(2.0, clr), G), (double)nsamples));
    }
    dstore(F.rgb + 3 * p, paint);
    int b = (int)(255.0 * paint.z), g = (int)(255.0 * paint.y), r = (int)(255.0 * paint.x);
    F.bgr[3 * p] = (uint8_t)(b < 255 ? b : 255);
    F.bgr[3 * p + 1] = (uint8_t)(g < 255 ? g : 255);
    F.bgr[3 * p + 2] = (uint8_t)(r < 255 ? r : 255);
}

// search() on caller-supplied rays.  With `research` the outputs are those of a second search of the same ray bounded
// by the first one's hit (mainOMP.cpp:60), done the way cast_ray does it: only when the first search's note asks for it.
__global__ void __launch_bounds__(64) k_trace_exact(DevScene S, size_t n, const double *src, const double *dir, const double *tmin_in,
                                                    int32_t *found, int32_t *tri, double *t, double *u, double *v, int research) {
    extern __shared__ double stack_lds[];
    StackD st = make_stack(S, stack_lds);
    for (size_t k = (size_t)blockIdx.x * 64 + threadIdx.x; k < n; k += (size_t)gridDim.x * 64) {
        RayD ray = make_ray(dload(src + 3 * k), dload(dir + 3 * k));
        HitD h; h.tmin = tmin_in[k]; h.tri = -1; h.u = 0; h.v = 0;
        bool ok;
        if (research) {
            SearchNote note;
            ok = search_ref_t<true>(S, ray, h, &note, st);
            if (ok) {
                h.tri = -1; h.u = 0; h.v = 0;                 // the bound stays: the first hit's distance
                ok = note.may_find_closer(h.tmin) && search_ref(S, ray, h, st);
            }
        } else {
            ok = search_ref(S, ray, h, st);
        }
        found[k] = ok; tri[k] = h.tri; t[k] = h.tmin; u[k] = h.u; v[k] = h.v;
    }
}

__global__ void k_camera_rays(ViewParams V, double x, double y, double *dir) {
    size_t P = (size_t)V.size * V.size, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    int i = (int)(p / V.size), j = (int)(p % V.size);
    RayD r = camera_ray(d3(V.eye[0], V.eye[1], V.eye[2]), V.zoom, V.sa, V.ca, V.size, i, j, x, y);
    dstore(dir + 3 * p, r.dir);
}

__global__ void k_hemisphere(uint32_t seed, uint32_t view, uint32_t pixel, int n, double *out) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    double r1;
    D3 v = sample_hemisphere(seed, view, pixel, (uint32_t)s, r1);
    dstore(out + 3 * s, v);
}

// ---- host-side launchers ----------------------------------------------------------
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? PT_OK : PT_ERR_HIP)

// Blocks to launch a striding search kernel with: as many one-wave blocks as the device holds at once (so that no
// block waits for another to finish: a second round of blocks would run at a fraction of the occupancy), at most
// PT_EXACT_WAVES (the stack overflow regions) and at most the work.
template <typename K>
static unsigned resident_waves(K kernel, size_t work_blocks, size_t lds_bytes = PT_RING64_LDS_BYTES) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 8;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    size_t w = (size_t)per_cu * (size_t)cus;
    if (w > PT_EXACT_WAVES) w = PT_EXACT_WAVES;
    if (w > work_blocks) w = work_blocks;
    return (unsigned)(w < 1 ? 1 : w);
}

int exact_primary(const DevScene &S, const ViewParams &V, const FrameBuffers &F, const SubBuffers &B, hipStream_t s, hipStream_t s2, hipEvent_t ev_a, hipEvent_t ev_b) {
    const size_t P = (size_t)V.size * V.size;
    const unsigned cblocks = (unsigned)((P + 255) / 256);
    const unsigned blocks = V.rb ? (unsigned)(((size_t)tile_threads(V.size) * 9 + 62) / 63) : (unsigned)(tile_threads(V.size) / 64);
    const unsigned pblocks = packet_blocks(V.size, V.rb);
    if (S.n_ref > 0) hipLaunchKernelGGL(k_packet_prep, dim3(((unsigned)S.n_ref + 255u) / 256u), dim3(256), 0, s, S, V, B.forms);
    PrimaryArgs PA;
    PA.S = S; PA.V = V; PA.F = F; PA.B = B; PA.nblocks = pblocks;
    if (V.model == PT_MODEL_MAIN) {
        static const unsigned ws = resident_waves(k_primary_search<PT_MODEL_MAIN>, PT_EXACT_WAVES, PT_PKT_LDS_BYTES), wp = resident_waves(k_primary_shade<PT_MODEL_MAIN>, PT_EXACT_WAVES),
                              wm = resident_waves(k_mirror_exact<PT_MODEL_MAIN>, PT_EXACT_WAVES);
        hipLaunchKernelGGL(k_primary_search<PT_MODEL_MAIN>, dim3(pblocks < ws ? pblocks : ws), dim3(64), PT_PKT_LDS_BYTES, s, PA);
        hipLaunchKernelGGL(k_primary_shade<PT_MODEL_MAIN>, dim3(blocks < wp ? blocks : wp), dim3(64), PT_RING64_LDS_BYTES, s, S, V, F, B, blocks);
        if (S.any_mirror) hipLaunchKernelGGL(k_mirror_exact<PT_MODEL_MAIN>, dim3(blocks < wm ? blocks : wm), dim3(64), PT_RING64_LDS_BYTES, s, S, V, F, B, blocks);
        hipLaunchKernelGGL(k_primary_combine<PT_MODEL_MAIN>, dim3(cblocks), dim3(256), 0, s, F, B, V.size, V.rb, V.part, V.n_parts);
    } else {
        static const unsigned ws = resident_waves(k_primary_search<PT_MODEL_OMP>, PT_EXACT_WAVES, PT_PKT_LDS_BYTES), wm = resident_waves(k_mirror_exact<PT_MODEL_OMP>, PT_EXACT_WAVES);
        const size_t slots = P * (V.rb ? 9 : 1);
        hipLaunchKernelGGL(k_primary_search<PT_MODEL_OMP>, dim3(pblocks < ws ? pblocks : ws), dim3(64), PT_PKT_LDS_BYTES, s, PA);
#ifdef PT_PKT_CLOCKS
        {   // instrumented build: wave clocks per part of the walk, summed over the launch
            unsigned long long c[16];
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(c, HIP_SYMBOL(g_pkt_clk), sizeof c) == hipSuccess) {
                fprintf(stderr, "pkt_clk");
                for (int k = 0; k < 8; k++) fprintf(stderr, " %llu/%llu", c[k], c[8 + k]);
                fprintf(stderr, "\n");
                for (int k = 0; k < 16; k++) c[k] = 0;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pkt_clk), c, sizeof c);
            }
        }
#endif
        const size_t sblocks = (slots + 255) / 256;
        // The listed sub-rays (few; the list's length is on the device) run on the second stream beside the shading of the others:
        // a handful of waves that each make a whole search alone are latency, not load, and the two kernels write different sub-rays.
        const bool side = s2 && ev_a && ev_b && hipEventRecord(ev_a, s) == hipSuccess && hipStreamWaitEvent(s2, ev_a, 0) == hipSuccess;
        hipLaunchKernelGGL(k_primary_again, dim3(128), dim3(64), PT_RING64_LDS_BYTES, side ? s2 : s, S, V, F, B);
        if (side && hipEventRecord(ev_b, s2) != hipSuccess) return PT_ERR_HIP;
        hipLaunchKernelGGL(k_primary_shade_omp, dim3((unsigned)(sblocks < 2048 ? sblocks : 2048)), dim3(256), 0, s, S, V, F, B);
        if (side && hipStreamWaitEvent(s, ev_b, 0) != hipSuccess) return PT_ERR_HIP;
        if (S.any_mirror) hipLaunchKernelGGL(k_mirror_exact<PT_MODEL_OMP>, dim3(blocks < wm ? blocks : wm), dim3(64), PT_RING64_LDS_BYTES, s, S, V, F, B, blocks);
        hipLaunchKernelGGL(k_primary_combine<PT_MODEL_OMP>, dim3(cblocks), dim3(256), 0, s, F, B, V.size, V.rb, V.part, V.n_parts);
    }
    return LAUNCH_OK();
}

int exact_gi(const DevScene &S, const ViewParams &V, const FrameBuffers &F, int s0, int nb, hipStream_t s) {
    int tthreads = tile_threads(V.size);
    unsigned blocks = (unsigned)(((size_t)tthreads * nb) / 64);
    static const unsigned w0 = resident_waves(k_gi_exact<PT_MODEL_MAIN>, PT_EXACT_WAVES), w1 = resident_waves(k_gi_exact<PT_MODEL_OMP>, PT_EXACT_WAVES);
    const unsigned wr = V.model == PT_MODEL_MAIN ? w0 : w1, waves = blocks < wr ? blocks : wr;
    if (V.model == PT_MODEL_MAIN)
        hipLaunchKernelGGL(k_gi_exact<PT_MODEL_MAIN>, dim3(waves), dim3(64), PT_RING64_LDS_BYTES, s, S, V, F, s0, tthreads, blocks);
    else
        hipLaunchKernelGGL(k_gi_exact<PT_MODEL_OMP>, dim3(waves), dim3(64), PT_RING64_LDS_BYTES, s, S, V, F, s0, tthreads, blocks);
    return LAUNCH_OK();
}

int gi_accumulate(const FrameBuffers &F, int size, int nb, int first_batch, hipStream_t s) {
    size_t P = (size_t)size * size;
    hipLaunchKernelGGL(k_gi_accumulate, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, F, size, nb, first_batch);
    return LAUNCH_OK();
}

int resolve_frame(const FrameBuffers &F, int size, int gi, int nsamples, hipStream_t s) {
    size_t P = (size_t)size * size;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, F, size, gi, nsamples);
    return LAUNCH_OK();
}

int exact_trace(const DevScene &S, size_t n, const double *src, const double *dir, const double *tmin_in,
                int32_t *found, int32_t *tri, double *t, double *u, double *v, int research, hipStream_t s) {
    const size_t blocks = (n + 63) / 64;
    static const unsigned wr = resident_waves(k_trace_exact, PT_EXACT_WAVES);
    hipLaunchKernelGGL(k_trace_exact, dim3((unsigned)(blocks < wr ? blocks : wr)), dim3(64), PT_RING64_LDS_BYTES, s, S, n,
                       src, dir, tmin_in, found, tri, t, u, v, research);
    return LAUNCH_OK();
}

int camera_rays(const ViewParams &V, double x, double y, double *dir, hipStream_t s) {
    size_t P = (size_t)V.size * V.size;
    hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, V, x, y, dir);
    return LAUNCH_OK();
}

int hemisphere_samples(uint32_t seed, uint32_t view, uint32_t pixel, int n, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_hemisphere, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, view, pixel, n, out);
    return LAUNCH_OK();
}
