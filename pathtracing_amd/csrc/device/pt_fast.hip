// fp32 wavefront path for the GI sample rays (the bulk of the rays at any
// useful sample count):
//
//   k_tile_valid/scan  once per view: where the rays of every 8x8 pixel tile start in the queue
//   k_gi_gen        G-buffer + counter RNG -> ray queue (SoA), sorted by tile (Z curve) and
//                   direction cell; decides in fp64 whether a ray hits the triangle it starts on
//   k_trace_closest kd traversal, nearest hit: persistent waves, a triangle test and/or a
//                   node step per lane per turn, per-lane ring stack in LDS (pt_trace_f32.h)
//   k_shade_gi      surface properties + lights; OMP model: writes the sample's
//                   contribution; MAIN model: compacts hits into a shading-point queue and
//                   queues a shadow ray per light that can reach the point
//   k_shadow_gi     MAIN model: any-hit query per shadow ray (same machinery)
//   k_light_sum     MAIN model: the lights of a point added in order, each if its ray found nothing
//   k_gi_sum        ordered per-pixel sum of the pass's contributions (fp64)
//   k_shade_sum_omp mainOMP model: the two stages before in one kernel (the contributions stay in LDS)
//
// A contribution is stored at its ray's place in the queue (streamed, not scattered) and carries the
// ray's (sample, pixel) code; the summing block of a tile sorts them by that code in LDS and adds each
// pixel's samples in sample order, so the image does not depend on the order in which waves draw from
// the queues or on how the generator sorted the rays.
#include <float.h>
#include <stdlib.h>

#include "pt_frame.h"
#include "pt_persistent.h"
#include "pt_trace_f32.h"
#include "pt_trace_quad.h"
#include "pt_vec.h"

#pragma clang fp contract(fast)

#define FS_SLOTS 8                  // ring entries per lane in LDS (one of them is kept free)
#define PT_REFILL_MIN 16             // idle lanes that trigger a refill
#define PT_PASS_COUNT_WORDS 64       // per-pass counters, padded to a 256-byte line
// Dynamic LDS asked for by a block of the shading kernel, which uses none of it: a way to keep half of a CU's wave
// slots free for another kernel.  It paid (+2.5 %) while the ray generator could share a CU with a shading block; the
// generator now sorts its rays in 84 KB of LDS and fills a CU by itself, and the shading kernel is faster at full
// occupancy (car +3 %).  PT_TUNE_SHADE_LDS sets it for experiments.
#define PT_SHADE_LDS_BYTES 0
#define PT_QUEUE_CHUNK 64          // work items a wave takes from a queue head per atomicAdd

// Queue append for a whole workgroup: ballot + prefix popcount inside each wave, a
// scan over the block's waves in LDS, ONE atomicAdd per block.  (One atomicAdd per
// wave on a single word saturates at ~90 per microsecond chip-wide, which is 3 ms
// for the 262 144 waves of a 16 M-ray pass.)  Every thread of the block must call it.
#define PT_APPEND_BLOCK 1024
__device__ __forceinline__ uint32_t block_append(bool valid, uint32_t *counter) {
    __shared__ uint32_t wave_n[PT_APPEND_BLOCK / 64];
    __shared__ uint32_t block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long mask = __ballot(valid);
    if (lane == 0) wave_n[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < nw; w++) { uint32_t c = wave_n[w]; wave_n[w] = tot; tot += c; }
        block_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    return block_base + wave_n[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// The same for n entries per thread (n may be 0): the thread's entries are consecutive, threads in order.
__device__ __forceinline__ uint32_t block_append_n(uint32_t n, uint32_t *counter) {
    __shared__ uint32_t wave_n[PT_APPEND_BLOCK / 64];
    __shared__ uint32_t block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t incl = n;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t nbr = __shfl_up(incl, o, 64);
        if (lane >= o) incl += nbr;
    }
    if (lane == 63) wave_n[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < nw; w++) { uint32_t c = wave_n[w]; wave_n[w] = tot; tot += c; }
        block_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    return block_base + wave_n[wave] + incl - n;
}

// ---- stage 1: ray generation from the G-buffer --------------------------------
__device__ __forceinline__ bool tile_pixel_f(int linear, int size, int &i, int &j) {
    int tiles_x = (size + 7) >> 3;
    int tile = linear >> 6, lane = linear & 63;
    int ti = tile / tiles_x, tj = tile - ti * tiles_x;
    i = ti * 8 + (lane >> 3);
    j = tj * 8 + (lane & 7);
    return i < size && j < size;
}

// createCoordSystem (util/lighting.cpp:42-48) and the sample direction in that frame (util/lighting.cpp:53-59,
// main.cpp:94-96), in fp64 so that the fast path shoots the same sample directions as the fp64 path.  The frame
// belongs to the pixel: it is made once per thread, not once per sample.
__device__ __forceinline__ void gi_frame(D3 nrm, D3 &Nt, D3 &Nb) {
#pragma clang fp contract(off)
    if (fabs(nrm.x) > fabs(nrm.y))
        Nt = ddivs(d3(nrm.z, 0, -nrm.x), sqrt(nrm.x * nrm.x + nrm.z * nrm.z));
    else
        Nt = ddivs(d3(0, -nrm.z, nrm.y), sqrt(nrm.y * nrm.y + nrm.z * nrm.z));
    Nb = dcross(nrm, Nt);
}
__device__ __forceinline__ D3 gi_direction(D3 nrm, D3 Nt, D3 Nb, uint32_t seed, uint32_t view, uint32_t pixel, uint32_t sample, double &r1) {
#pragma clang fp contract(off)
    D3 sm = sample_hemisphere(seed, view, pixel, sample, r1);
    return d3(sm.x * Nb.x + sm.y * nrm.x + sm.z * Nt.x,
              sm.x * Nb.y + sm.y * nrm.y + sm.z * Nt.y,
              sm.x * Nb.z + sm.y * nrm.z + sm.z * Nt.z);
}

// The reference starts a GI ray 1e-4 along the SHADING normal from the hit point and tests it against
// every triangle, the one it starts on included (main.cpp:94-99): where the interpolated normal leans
// away from the geometric one, sample directions below the triangle's own plane hit that triangle at
// t ~ 1e-4.  fp32 cannot decide that test (the offset is below one ulp of a coordinate of a few
// thousand), so the closest-hit kernel skips the origin triangle and this kernel decides it in fp64:
// Moeller-Trumbore of util/polygonMesh.cpp:54-76 on the one triangle.  Returns t > 0 on a hit.
__device__ __forceinline__ bool origin_triangle_hit(const double *T9, D3 o, D3 w, float &t, float &u, float &v) {
    D3 v0 = dload(T9), ba = dload(T9 + 3), ca = dload(T9 + 6);
    D3 P = dcross(w, ca);
    double inv = 1.0 / ddot(P, ba);
    D3 T = dsub(o, v0);
    double uu = ddot(P, T) * inv;
    D3 Qv = dcross(T, ba);
    double vv = ddot(Qv, w) * inv, tt = ddot(Qv, ca) * inv;
    t = (float)tt; u = (float)uu; v = (float)vv;
    return uu >= 0.0 && vv >= 0.0 && uu + vv <= 1.0 && tt > 0.0;
}

// Queue order of the 8x8 pixel tiles: along a Z curve when the image is a power of two of tiles wide, so that a
// run of consecutive tiles -- what the waves of an XCD work on at the same time -- is a compact patch of the image.
__device__ __forceinline__ uint32_t compact_bits(uint32_t v) {
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}
__device__ __forceinline__ int ordered_tile(int q, int size) {
    const int tiles_x = (size + 7) >> 3;
    if (tiles_x & (tiles_x - 1)) return q;
    return (int)(compact_bits((uint32_t)q >> 1) * (uint32_t)tiles_x + compact_bits((uint32_t)q));
}

// Pixels with a primary hit per tile, in queue order (once per view) ...
__global__ void __launch_bounds__(256) k_tile_valid(FrameBuffers F, FastBuffers Q, int size, int tiles) {
    const int q = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);          // a wave per tile, a lane per pixel
    if (q >= tiles) return;
    int i, j;
    const bool hit = tile_pixel_f(ordered_tile(q, size) * 64 + (int)(threadIdx.x & 63), size, i, j) && !(F.ghit[3 * ((size_t)i * size + j)] == -99999);
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0) Q.tile_base[q] = (uint32_t)__popcll(m);
}
// ... and their exclusive prefix sum in place, total in tile_base[tiles]: where each tile's rays start in the queue.
__global__ void __launch_bounds__(1024) k_tile_scan(FastBuffers Q, int tiles) {
    __shared__ uint32_t wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int run = (tiles + 1023) / 1024, first = tid * run;
    uint32_t local = 0;
    for (int k = 0; k < run; k++) local += first + k < tiles ? Q.tile_base[first + k] : 0u;
    uint32_t incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t nbr = __shfl_up(incl, o, 64);
        if (lane >= o) incl += nbr;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 16; w++) { before += w < wave ? wave_sum[w] : 0u; total += wave_sum[w]; }
    uint32_t acc = before + incl - local;
    for (int k = 0; k < run; k++)
        if (first + k < tiles) { const uint32_t n = Q.tile_base[first + k]; Q.tile_base[first + k] = acc; acc += n; }
    if (tid == 0) Q.tile_base[tiles] = total;
}

// One block = one 8x8 pixel tile x up to 64 samples of the pass (thread = pixel x sample lane,
// four samples each): 4096 rays with nearby origins.  They are written to the queue sorted by
// direction cell (counting sort in LDS), so that the 64 rays a wave of the closest-hit kernel
// draws together start close to each other and point the same way.  Queue order inside the block is
// free: the ray's code, (sample of the pass) * 64 + (pixel of the tile), travels with it.
#define PT_GEN_THREADS 1024                         // threads of a generator block (PT_TUNE_GEN_THREADS: 512 or 1024)
#define PT_GEN_CELL_BITS 6                          // direction cells per axis of the octahedral map = 1 << bits
#define PT_GEN_BINS (1 << (2 * PT_GEN_CELL_BITS))
__device__ __forceinline__ uint32_t dir_bin(float x, float y, float z) {
    // The direction's place on the octahedral map of the sphere ([-1, 1]^2, every cell meets the sphere: a grid over the
    // cube around the sphere wastes five cells in six), cells numbered along a Z curve so that the 64 rays a wave draws
    // -- a run of 30-odd cells at the ~2 rays a cell holds -- point into one small patch of directions.
    // 64 x 64 cells: closest-hit kernel +2.4 % over 8 x 8 x 8 cube cells in x-major order; 8 x 8 octahedral cells -4 %.
    const float k = 1.0f / (fabsf(x) + fabsf(y) + fabsf(z));
    float u = x * k, v = y * k;
    if (z < 0.0f) {
        const float uu = (1.0f - fabsf(v)) * (u >= 0.0f ? 1.0f : -1.0f), vv = (1.0f - fabsf(u)) * (v >= 0.0f ? 1.0f : -1.0f);
        u = uu; v = vv;
    }
    const float n = (float)(1 << PT_GEN_CELL_BITS), hi = n - 1.0f;
    const uint32_t qu = (uint32_t)fminf(fmaxf((u * 0.5f + 0.5f) * n, 0.0f), hi), qv = (uint32_t)fminf(fmaxf((v * 0.5f + 0.5f) * n, 0.0f), hi);
    // interleave the bits of the two cell coordinates (at most 8 bits each)
    auto spread = [](uint32_t x) { x = (x | (x << 4)) & 0x0F0Fu; x = (x | (x << 2)) & 0x3333u; return (x | (x << 1)) & 0x5555u; };
    return (spread(qu) << 1) | spread(qv);
}
static_assert(PT_GEN_CELL_BITS <= 8, "the bit interleave of dir_bin handles 8 bits per coordinate");

// T threads per block (1 024 or 512), G tiles per block (1 or 4): a block takes G neighbouring tiles of 64 pixels (in queue
// order) and 4 * T / (64 G) samples of each -- four rays per thread -- and sorts them as one run.  G = 4 is for passes of at most
// 16 samples (the reference's defaults): the run is as long as with 64 samples of one tile, so the closest-hit kernel
// meets rays as well sorted by direction as at high sample counts, and every thread of the block has rays to make.
template <int T, int G>
__global__ void __launch_bounds__(T) k_gi_gen(DevScene S, ViewParams V, FrameBuffers F, FastBuffers Q, int s0, int nb, int groups, int tiles) {
    constexpr int WAVES = T / 64, SPB = 4 * WAVES / G, PX = 64 * G, RAYS = PX * SPB;      // samples per pixel, pixels and rays of a block
    static_assert(WAVES % G == 0, "waves of a block per tile");
    __shared__ uint32_t hist[PT_GEN_BINS];                   // rays per direction cell, then (in place) the cell's first place in the block's run
    __shared__ uint32_t block_base, wave_sum[WAVES];
    const int tid = threadIdx.x, wv = tid >> 6;
    const int qgroup = blockIdx.x / groups, group = blockIdx.x - qgroup * groups;
    const int qtile = qgroup * G + wv % G, ssub = wv / G;    // this thread's tile and its place among the samples of a round
    int i = 0, j = 0;
    const bool live = qtile < tiles && tile_pixel_f(ordered_tile(qtile, V.size) * 64 + (tid & 63), V.size, i, j);
    const size_t p = live ? (size_t)i * V.size + j : 0;
    for (int k = tid; k < PT_GEN_BINS; k += T) hist[k] = 0;
    __syncthreads();
    bool pix_valid = false;
    float4 ro = make_float4(0.f, 0.f, 0.f, 0.f);
    D3 nrm = d3(0, 0, 0), hit = d3(0, 0, 0), ng = d3(0, 0, 0);
    double side = 0;
    int otri = -1;
    if (live) {
        hit = dload(F.ghit + 3 * p);
        pix_valid = !(hit.x == -99999);
        if (pix_valid) {
            nrm = dload(F.gnrm + 3 * p);
            otri = F.rec_tri[p];
            ro = make_float4((float)hit.x, (float)hit.y, (float)hit.z, __int_as_float(otri));
            if (otri >= 0) {
                // geometric normal of the origin triangle and the side of it the ray origin is on
                const double *T9 = S.triGeo + (size_t)otri * 9;
                ng = dcross(dload(T9 + 3), dload(T9 + 6));
                side = ddot(dsub(hit, dload(T9)), ng);
            }
        }
    }
    D3 Nt = d3(0, 0, 0), Nb = d3(0, 0, 0);
    if (pix_valid) gi_frame(nrm, Nt, Nb);
    float4 rd[4];
    float st[4], su[4], sv[4];          // origin-triangle hit of the sample, when there is one
    uint32_t key[4], rank[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int sl = group * SPB + r * (WAVES / G) + ssub;
        key[r] = 0xFFFFFFFFu;
        if (live && sl < nb) {
            if (pix_valid) {
                double r1;
                D3 w = gi_direction(nrm, Nt, Nb, V.seed, V.view, (uint32_t)p, (uint32_t)(s0 + sl), r1);
                rd[r] = make_float4((float)w.x, (float)w.y, (float)w.z, (float)r1);
                key[r] = dir_bin(rd[r].x, rd[r].y, rd[r].z);
                // a ray that hits the triangle it starts on is resolved here; the sign of r1 (>= 0) tells
                // the closest-hit kernel to leave it alone
                // (only a direction that points through the triangle's own plane can hit it: one dot product decides
                // for the 95 % of the samples that do not)
                if (otri >= 0 && (w.x * ng.x + w.y * ng.y + w.z * ng.z) * side <= 0.0 &&
                    origin_triangle_hit(S.triGeo + (size_t)otri * 9, hit, w, st[r], su[r], sv[r]))
                    rd[r].w = __uint_as_float(__float_as_uint(rd[r].w) | 0x80000000u);
                rank[r] = atomicAdd(&hist[key[r]], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t in_block;                                       // rays of the block
    {
        // exclusive prefix sum of the histogram, in place: a run of bins per thread, a scan per wave, a scan over the waves
        constexpr int RUN = (PT_GEN_BINS + T - 1) / T;
        const int lane = tid & 63, wave = tid >> 6, first = tid * RUN;
        uint32_t mine[RUN], local = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) { mine[k] = first + k < PT_GEN_BINS ? hist[first + k] : 0u; local += mine[k]; }
        uint32_t incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t nbr = __shfl_up(incl, o, 64);
            if (lane >= o) incl += nbr;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < WAVES; w++) { before += w < wave ? wave_sum[w] : 0u; total += wave_sum[w]; }
        uint32_t run = before + incl - local;
#pragma unroll
        for (int k = 0; k < RUN; k++)
            if (first + k < PT_GEN_BINS) { hist[first + k] = run; run += mine[k]; }
        // the block's place in the queue is fixed: tiles in queue order, the sample groups of a tile one after another
        if (tid == 0) {
            const int q0 = qgroup * G, q1 = q0 + G < tiles ? q0 + G : tiles;
            const uint32_t t0 = Q.tile_base[q0], valid = Q.tile_base[q1] - t0;
            const uint32_t done_samples = (uint32_t)(group * SPB < nb ? group * SPB : nb);
            block_base = t0 * (uint32_t)nb + valid * done_samples;
            if (blockIdx.x == 0) Q.counts[0] = Q.tile_base[tiles] * (uint32_t)nb;
        }
        in_block = total;
    }
    __syncthreads();
    // The sorted order is made in LDS -- every ray's direction and code dropped at its place in the block's run -- and the
    // run then leaves for the queue in whole lines: written straight from the threads, 16 bytes each at a random place
    // of the run, the three arrays cost 0.3 ms more per 67 M rays.  The origin is the pixel's, found through the code.
    extern __shared__ float4 staged[];                       // RAYS directions, PX origins, RAYS codes
    float4 *stage_d = staged, *stage_o = staged + RAYS;
    uint32_t *stage_c = (uint32_t *)(staged + RAYS + PX);
    if (tid < PX) stage_o[tid] = ro;                         // (the first G waves: ssub == 0, one thread per pixel of the block)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (key[r] != 0xFFFFFFFFu) {
            const int sl = group * SPB + r * (WAVES / G) + ssub;
            const uint32_t at = hist[key[r]] + rank[r];
            stage_d[at] = rd[r];
            stage_c[at] = (uint32_t)sl * (uint32_t)PX + (uint32_t)(tid & (PX - 1));
            if (__float_as_uint(rd[r].w) >> 31) Q.hits[block_base + at] = make_float4(st[r], su[r], sv[r], __int_as_float(otri));
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < in_block; k += T) {
        const uint32_t code = stage_c[k], q = block_base + k;
        Q.rays_o[q] = stage_o[code & (uint32_t)(PX - 1)];
        Q.rays_d[q] = stage_d[k];
        Q.ray_slot[q] = code;
    }
}
#define PT_GEN_LDS_BYTES(T, G) ((size_t)((T) * 4 + 64 * (G)) * sizeof(float4) + (size_t)(T) * 4 * sizeof(uint32_t))

// ---- stage 2: closest hit -------------------------------------------------------
// LS: LaneState (binary node records, pt_trace_f32.h) or LaneStateQ (quad records, pt_trace_quad.h); the same search.
template <bool COUNT, class LS>
struct ClosestPolicy {
    const DevScene &S;
    const FastBuffers &Q;
    const RootBox box;
    uint2 *stk;
    uint32_t slot_mask;
    LS L;
    uint32_t r, work[2];
    __device__ __forceinline__ void census() { if (COUNT) { work[0]++; work[1] += tris_here(L); } }
    __device__ __forceinline__ bool step(bool active) {
        const bool done = lane_turn<false, COUNT>(L, S, stk, slot_mask, work, active);
        if (done) Q.hits[r] = make_float4(L.tmin, L.hu, L.hv, __int_as_float(L.htri));
        return active & !done;
    }
    __device__ __forceinline__ bool start(uint32_t item) {
        r = item;
        const float4 o = Q.rays_o[r], d = Q.rays_d[r];
        if (__float_as_uint(d.w) >> 31) return false;      // resolved by the generator (hit on its own triangle)
        if (lane_begin(L, S, box, o.x, o.y, o.z, d.x, d.y, d.z, FLT_MAX, __float_as_int(o.w))) { census(); return true; }
        Q.hits[r] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        return false;
    }
};

// PT_TRACE_CLOCKS (an instrumented build for tools/trace_exit.sh, never the product): when every wave of the closest-hit
// launch began and ended, on the 100 MHz clock.
#ifdef PT_TRACE_CLOCKS
__device__ unsigned long long g_trace_clk[2][PT_TRACE_WAVES];
#endif
template <bool COUNT, class LS>
__global__ void __launch_bounds__(64) k_trace_closest(DevScene S, FastBuffers Q, uint32_t slot_mask, int refill_min, uint32_t chunk) {
    extern __shared__ uint2 stack[];            // (slot_mask + 1) * 64 entries
#ifdef PT_TRACE_CLOCKS
    if ((threadIdx.x & 63) == 0) g_trace_clk[0][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    ClosestPolicy<COUNT, LS> P{S, Q, root_box(S), stack + (threadIdx.x & 63), slot_mask};
    P.work[0] = P.work[1] = 0;
    persistent_loop(P, Q.counts[0], Q.heads, chunk, refill_min);
#ifdef PT_TRACE_CLOCKS
    if ((threadIdx.x & 63) == 0) g_trace_clk[1][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    if (COUNT) {
        for (int k = 0; k < 2; k++) {
            uint32_t v = P.work[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd(&Q.work[k], (unsigned long long)v);
        }
    }
}

// ---- stage 3: shading of the GI hits ---------------------------------------------
struct SurfF { float nx, ny, nz, tr, tg, tb; };

__device__ __forceinline__ int clampi_f(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// texel (px, py) of the texture with table entry tm = (rows, cols, first texel), clamped as the reference's Mat::at is used
__device__ __forceinline__ uint32_t texel_f(const DevScene &S, const int4 tm, int px, int py) {
    const int row = clampi_f(py, 0, tm.x - 1), col = clampi_f(px, 0, tm.y - 1);
    return S.texels4[(size_t)tm.z + (size_t)row * tm.y + col];
}

// surfaceProperties (util/polygonMesh.cpp:81-118) in fp32, from the packed per-triangle record.
// Also returns the albedo of the triangle's mesh.
__device__ __forceinline__ SurfF surface_f(const DevScene &S, int tri, float bu, float bv, float &ar, float &ag, float &ab) {
    SurfF o;
    const float4 *R = S.triShadeF + (size_t)tri * 4;
    const float4 r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3];
    const uint32_t code = __float_as_uint(r3.w), mode = code & ((1u << S.tex_bits) - 1u);
    const float4 al = S.meshAlbF[code >> S.tex_bits];
    ar = al.x; ag = al.y; ab = al.z;
    const float w = 1.0f - bu - bv;
    float nx = w * r0.x + bu * r0.w + bv * r1.z;
    float ny = w * r0.y + bu * r1.x + bv * r1.w;
    float nz = w * r0.z + bu * r1.y + bv * r2.x;
    float il = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    o.nx = nx * il; o.ny = ny * il; o.nz = nz * il;
    if (mode == 0u) { o.tr = o.tg = o.tb = 1.0f; return o; }
    const float u0 = r2.y, v0 = r2.z, u1 = r2.w, v1 = r3.x, u2 = r3.y, v2 = r3.z;
    const bool at_hit = mode > (uint32_t)S.n_tex;
    const int4 tm = S.texF[mode - 1u - (at_hit ? (uint32_t)S.n_tex : 0u)];
    if (at_hit) {
        const float rows = (float)tm.x;
        int r = (int)((u0 * w + u1 * bu + u2 * bv) * rows);
        int cc = (int)((v0 * w + v1 * bu + v2 * bv) * rows);
        const uint32_t p = texel_f(S, tm, r, cc);
        o.tr = (float)((p >> 16) & 255u) * (1.0f / 256.0f); o.tg = (float)((p >> 8) & 255u) * (1.0f / 256.0f); o.tb = (float)(p & 255u) * (1.0f / 256.0f);
    } else {
        const float rows = (float)tm.x, cols = (float)tm.y;
        float cr = 0, cg = 0, cb = 0;
        const float wt[3] = {w, bu, bv}, uu[3] = {u0, u1, u2}, vv[3] = {v0, v1, v2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint32_t p = texel_f(S, tm, (int)(uu[i] * rows), (int)(vv[i] * cols));
            cr += wt[i] * (float)((p >> 16) & 255u); cg += wt[i] * (float)((p >> 8) & 255u); cb += wt[i] * (float)(p & 255u);
        }
        o.tr = cr * (1.0f / 256.0f); o.tg = cg * (1.0f / 256.0f); o.tb = cb * (1.0f / 256.0f);
    }
    return o;
}

// Light::illuminate (util/lighting.cpp:24-36) in fp32: direction, distance, radiance.
// Lights come as two float4 each: (position, type) and (mag * colour, 0).
__device__ __forceinline__ void light_f(const float4 *lights, int k, float hx, float hy, float hz, float &lx, float &ly, float &lz,
                                        float &dis, float &sr, float &sg, float &sb) {
    const float4 a = lights[2 * k], b = lights[2 * k + 1];
    if (__float_as_int(a.w) == 'd') {
        float il = __builtin_amdgcn_rsqf(a.x * a.x + a.y * a.y + a.z * a.z);
        lx = a.x * il; ly = a.y * il; lz = a.z * il;
        dis = FLT_MAX;
        sr = b.x; sg = b.y; sb = b.z;
    } else {
        lx = a.x - hx; ly = a.y - hy; lz = a.z - hz;
        float d2 = lx * lx + ly * ly + lz * lz;
        float il = __builtin_amdgcn_rsqf(d2);
        dis = d2 * il;
        lx *= il; ly *= il; lz *= il;
        float k4 = il * (1.0f / (4.0f * 3.14159265358979f));          // 1 / (4 * dis * pi): linear falloff
        sr = k4 * b.x; sg = k4 * b.y; sb = k4 * b.z;
    }
}

// r1 * castRay(sample ray, depth -1) of the mainOMP model (mainOMP.cpp:55-62,74-75: every light counts as visible;
// txt * albedo * direct / pi) for a ray from o along d (d.w: its weight r1) with hit record h; 0 on a miss.
__device__ __forceinline__ float4 omp_contribution(const DevScene &S, float ox, float oy, float oz, float4 d, float4 h) {
    const int tri = __float_as_int(h.w);
    if (tri < 0) return make_float4(0.f, 0.f, 0.f, 0.f);           // castRay returns 0 on a miss at depth -1
    const float3 o = make_float3(ox, oy, oz);
    float ar, ag, ab;
    const SurfF sf = surface_f(S, tri, h.y, h.z, ar, ag, ab);
    ar *= sf.tr; ag *= sf.tg; ab *= sf.tb;
    const float hx = o.x + h.x * d.x + 0.0001f * sf.nx, hy = o.y + h.x * d.y + 0.0001f * sf.ny, hz = o.z + h.x * d.z + 0.0001f * sf.nz;
    float dr = 0, dg = 0, db = 0;
    for (int k = 0; k < S.n_light; k++) {
        float lx, ly, lz, dis, sr, sg, sb;
        light_f(S.lightsF, k, hx, hy, hz, lx, ly, lz, dis, sr, sg, sb);
        const float c = fmaxf(0.0f, lx * sf.nx + ly * sf.ny + lz * sf.nz);
        dr += c * sr; dg += c * sg; db += c * sb;
    }
    const float k = fabsf(d.w) * (1.0f / 3.14159265358979f);
    return make_float4(k * ar * dr, k * ag * dg, k * ab * db, 0.f);
}

// main.cpp:60-66 for light k at a shading point (position xyz | triangle, normal xyz): direction and length of the
// shadow ray; false when no ray is traced because the light adds nothing whatever the ray says -- it is behind the
// surface, or the triangle the ray starts on shadows it.  The reference starts the ray
// 1e-4 along the shading normal and tests it against that triangle as well (main.cpp:63-66): the triangle shadows the
// point when the light is on the other side of its plane than the shading normal.  Decided here by signs; the search
// skips the triangle.
__device__ __forceinline__ bool shadow_ray_wanted(const DevScene &S, int k, float4 pos, float4 nrm, float &lx, float &ly, float &lz, float &dis) {
#pragma clang fp contract(off)
    float sr, sg, sb;
    light_f(S.lightsF, k, pos.x, pos.y, pos.z, lx, ly, lz, dis, sr, sg, sb);
    const float c = lx * nrm.x + ly * nrm.y + lz * nrm.z;
    if (!(c > 0.0f)) return false;
    const float4 ng = S.triNgF[__float_as_int(pos.w)];
    return !((lx * ng.x + ly * ng.y + lz * ng.z) * (nrm.x * ng.x + nrm.y * ng.y + nrm.z * ng.z) < 0.0f);
}

template <int MODEL>
__global__ void __launch_bounds__(PT_APPEND_BLOCK) k_shade_gi(DevScene S, FastBuffers Q) {
    uint32_t r = blockIdx.x * PT_APPEND_BLOCK + threadIdx.x;
    bool in_range = r < Q.counts[0];
    bool hit = false;
    float4 sp_pos, sp_nrm, sp_base;
    if (in_range && MODEL == PT_MODEL_OMP) {
        const float4 o = Q.rays_o[r];
        Q.contrib[r] = omp_contribution(S, o.x, o.y, o.z, Q.rays_d[r], Q.hits[r]);
    } else if (in_range) {
        float4 h = Q.hits[r];
        int tri = __float_as_int(h.w);
        if (tri < 0) {
            Q.contrib[r] = make_float4(0.f, 0.f, 0.f, 0.f);        // castRay returns 0 on a miss at depth -1
        } else {
            float4 d = Q.rays_d[r];
            float ar, ag, ab;
            SurfF sf = surface_f(S, tri, h.y, h.z, ar, ag, ab);
            ar *= sf.tr; ag *= sf.tg; ab *= sf.tb;
            // Shadow rays start 1e-4 off the surface (main.cpp:50), and that step has to survive: o + t d in fp32
            // is off by t * 2^-24, more than 1e-4 for a ray a few thousand units long, and puts the point behind a
            // second triangle lying in the same plane (two-sided leaves are modelled that way), which then shadows it.
            // The point is taken from the triangle instead, v0 + u e1 + v e2 in fp64, and rounded once.
            const double *T9 = S.triGeo + (size_t)tri * 9;
            const double bu = (double)h.y, bv = (double)h.z;
            const float hx = (float)(T9[0] + bu * T9[3] + bv * T9[6] + 1e-4 * (double)sf.nx);
            const float hy = (float)(T9[1] + bu * T9[4] + bv * T9[7] + 1e-4 * (double)sf.ny);
            const float hz = (float)(T9[2] + bu * T9[5] + bv * T9[8] + 1e-4 * (double)sf.nz);
            hit = true;
            float k = fabsf(d.w) * 0.6f * (1.0f / 3.14159265358979f);            // main.cpp:77 with ref = 0
            sp_pos = make_float4(hx, hy, hz, __int_as_float(tri));
            sp_nrm = make_float4(sf.nx, sf.ny, sf.nz, __uint_as_float(r));
            sp_base = make_float4(k * ar, k * ag, k * ab, 0.f);
        }
    }
    if (MODEL == PT_MODEL_MAIN) {
        // The shadow rays of the point, one per light that can add something (shadow_ray_wanted), are made here, where every
        // lane has a point, and queued one after the other in light order; the any-hit kernel is then a plain ray query.
        // Which lights get a ray: a bit per light while there are at most 32 of them; with more, a count, and every stage that
        // needs to know which ones asks shadow_ray_wanted again (the same arithmetic on the same numbers gives the same answer).
        const bool masked = S.n_light <= 32;
        uint32_t want = 0, nwant = 0;                          // bit k: light k gets a ray | how many do
        float4 one = make_float4(0.f, 0.f, 0.f, 0.f);          // direction and length of the newest ray
        if (hit)
            for (int k = 0; k < S.n_light; k++) {
                float lx, ly, lz, dis;
                if (shadow_ray_wanted(S, k, sp_pos, sp_nrm, lx, ly, lz, dis)) { if (masked) want |= 1u << k; nwant++; one = make_float4(lx, ly, lz, dis); }
            }
        const uint32_t q = block_append(hit, &Q.counts[1]);
        uint32_t sq = block_append_n(nwant, &Q.counts[2]);
        if (hit) {
            // the point's record: position | lights with a ray (more than 32 lights: the point's triangle), normal | queue index
            // of the GI ray, base colour | first shadow ray
            Q.sp_pos[q] = make_float4(sp_pos.x, sp_pos.y, sp_pos.z, masked ? __uint_as_float(want) : sp_pos.w);
            Q.sp_nrm[q] = sp_nrm;
            Q.sp_base[q] = make_float4(sp_base.x, sp_base.y, sp_base.z, __uint_as_float(sq));
            if (nwant == 1) {                                   // the usual case: the one ray is still in registers
                Q.sh_o[sq] = sp_pos; Q.sh_d[sq] = one;
            } else if (nwant) {
                for (int k = 0; k < S.n_light; k++) {
                    float lx, ly, lz, dis;
                    if (masked ? !((want >> k) & 1u) : !shadow_ray_wanted(S, k, sp_pos, sp_nrm, lx, ly, lz, dis)) continue;
                    if (masked) shadow_ray_wanted(S, k, sp_pos, sp_nrm, lx, ly, lz, dis);
                    Q.sh_o[sq] = sp_pos;
                    Q.sh_d[sq] = make_float4(lx, ly, lz, dis);
                    sq++;
                }
            }
        }
    }
}

// ---- stage 4 (MAIN model): shadow rays of the GI hits ------------------------------
// One work item = one shadow ray; the answer (1.0f: nothing between the point and the light) replaces the ray's length.
template <class LS>
struct AnyHitPolicy {
    const DevScene &S;
    const FastBuffers &Q;
    const RootBox box;
    uint2 *stk;
    uint32_t slot_mask;
    LS L;
    uint32_t q;
    __device__ __forceinline__ void answer(bool visible) { ((float *)(Q.sh_d + q))[3] = visible ? 1.0f : 0.0f; }
    __device__ __forceinline__ bool step(bool active) {
        const bool done = lane_turn<true, false>(L, S, stk, slot_mask, nullptr, active);
        if (done) answer(L.htri < 0);
        return active & !done;
    }
    __device__ __forceinline__ bool start(uint32_t item) {
        q = item;
        const float4 o = Q.sh_o[q], d = Q.sh_d[q];
        if (lane_begin(L, S, box, o.x, o.y, o.z, d.x, d.y, d.z, d.w, __float_as_int(o.w))) return true;
        answer(true);                                              // the ray misses the scene's box
        return false;
    }
};

template <class LS>
__global__ void __launch_bounds__(64) k_shadow_gi(DevScene S, FastBuffers Q, uint32_t slot_mask, int refill_min, uint32_t chunk) {
    extern __shared__ uint2 stack[];
    AnyHitPolicy<LS> P{S, Q, root_box(S), stack + (threadIdx.x & 63), slot_mask};
    persistent_loop(P, Q.counts[2], Q.heads + PT_HEAD_WORDS, chunk, refill_min);
}

// The light loop of a shading point once its shadow rays are answered (main.cpp:60-66,77): the lights in order, each
// adding cos * radiance when its ray found nothing.
__global__ void __launch_bounds__(256) k_light_sum(DevScene S, FastBuffers Q) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= Q.counts[1]) return;
    const float4 pos = Q.sp_pos[p], nrm = Q.sp_nrm[p], base = Q.sp_base[p];
    uint32_t sq = __float_as_uint(base.w);
    float dr = 0.f, dg = 0.f, db = 0.f;
    auto add_light = [&](int k) {
        float lx, ly, lz, dis, sr, sg, sb;
        light_f(S.lightsF, k, pos.x, pos.y, pos.z, lx, ly, lz, dis, sr, sg, sb);
        const float c = lx * nrm.x + ly * nrm.y + lz * nrm.z;
        dr += __fmul_rn(c, sr); dg += __fmul_rn(c, sg); db += __fmul_rn(c, sb);
    };
    if (S.n_light <= 32) {
        for (uint32_t want = __float_as_uint(pos.w); want; want &= want - 1u, sq++)
            if (Q.sh_d[sq].w != 0.0f) add_light(__ffs((int)want) - 1);      // 0: something lies between the point and the light
    } else {
        // more lights than a mask holds: which of them got a ray is asked again, as k_shade_gi asked (pos.w is the point's triangle)
        for (int k = 0; k < S.n_light; k++) {
            float lx, ly, lz, dis;
            if (!shadow_ray_wanted(S, k, pos, nrm, lx, ly, lz, dis)) continue;
            if (Q.sh_d[sq].w != 0.0f) add_light(k);
            sq++;
        }
    }
    Q.contrib[__float_as_uint(nrm.w)] = make_float4(base.x * dr, base.y * dg, base.z * db, 0.f);
}

// ---- stage 5: indirect += r1 * L over the pass's samples, in sample order ------------
// One block per 8x8 pixel tile.  The tile's contributions lie in the queue where its rays lay: per group of up to 64
// samples one run of (valid pixels x samples) entries in the generator's sort order.  The block streams a run in,
// drops every entry into LDS at the place its code names -- the scatter that sorting the rays made necessary happens
// in LDS, not in device memory, where 16-byte writes at random places cost the shading kernel 0.7 ms per 67 M rays --
// and then every pixel adds its samples in sample order in fp64, as the reference's loop does (main.cpp:129-133).
// G: the tiles a block of the generator sorted as one run (1 or 4); the block then takes those G tiles, 64 / G samples at a time.
#define PT_SUM_THREADS 256
template <int G>
__global__ void __launch_bounds__(PT_SUM_THREADS) k_gi_sum(FrameBuffers F, FastBuffers Q, int size, int nb, int first_batch, int tiles) {
    constexpr int PX = 64 * G, CH = 64 / G;
    __shared__ float cr[64 * 64], cg[64 * 64], cb[64 * 64];
    const int tid = threadIdx.x, qgroup = blockIdx.x;
    if (qgroup == 0 && tid == 0) {                  // the pass's ray counts join the totals of the view (read once, at its end)
        Q.totals[0] += Q.counts[0];
        Q.totals[1] += Q.counts[2];
    }
    const int q0 = qgroup * G, q1 = q0 + G < tiles ? q0 + G : tiles, qtile = q0 + (tid >> 6);
    const uint32_t t0 = Q.tile_base[q0], valid = Q.tile_base[q1] - t0;
    int i = 0, j = 0;
    const bool mine = tid < PX && qtile < tiles && tile_pixel_f(ordered_tile(qtile, size) * 64 + (tid & 63), size, i, j);
    const size_t p = mine ? (size_t)i * size + j : 0;
    double ax = 0, ay = 0, az = 0;
    if (mine && !first_batch) { ax = F.gi_acc[3 * p]; ay = F.gi_acc[3 * p + 1]; az = F.gi_acc[3 * p + 2]; }
    // a pixel without a primary hit shot no rays (main.cpp:86 returns 0 for it)
    const bool shot = mine && !(F.ghit[3 * p] == -99999);
    for (int g0 = 0; g0 < nb && valid; g0 += CH) {
        const int gs = nb - g0 < CH ? nb - g0 : CH;
        const uint32_t base = t0 * (uint32_t)nb + valid * (uint32_t)g0, n = valid * (uint32_t)gs;
        for (uint32_t k = tid; k < n; k += PT_SUM_THREADS) {
            const float4 c = Q.contrib[base + k];
            const uint32_t at = Q.ray_slot[base + k] - (uint32_t)g0 * (uint32_t)PX;
            cr[at] = c.x; cg[at] = c.y; cb[at] = c.z;
        }
        __syncthreads();
        if (shot)
            for (int s = 0; s < gs; s++) { ax += (double)cr[s * PX + tid]; ay += (double)cg[s * PX + tid]; az += (double)cb[s * PX + tid]; }
        __syncthreads();
    }
    if (mine) { F.gi_acc[3 * p] = ax; F.gi_acc[3 * p + 1] = ay; F.gi_acc[3 * p + 2] = az; }
}

// The two stages above in one kernel for the mainOMP model, whose contribution needs no shadow ray: the block of a tile
// shades the tile's rays itself and drops the results straight into LDS, so the contributions never travel to device
// memory and back (32 bytes per ray) and one launch per pass goes.
#define PT_FUSED_THREADS 1024
template <int G>
__global__ void __launch_bounds__(PT_FUSED_THREADS) k_shade_sum_omp(DevScene S, FrameBuffers F, FastBuffers Q, int size, int nb, int first_batch, int tiles) {
    constexpr int PX = 64 * G, CH = 64 / G;
    __shared__ float cr[64 * 64], cg[64 * 64], cb[64 * 64];
    __shared__ float org[3][PX];                    // the rays' origins: the primary hits of the block's pixels, as k_gi_gen rounds them
    const int tid = threadIdx.x, qgroup = blockIdx.x;
    if (qgroup == 0 && tid == 0) {
        Q.totals[0] += Q.counts[0];
        Q.totals[1] += Q.counts[2];
    }
    const int q0 = qgroup * G, q1 = q0 + G < tiles ? q0 + G : tiles, qtile = q0 + (tid >> 6);
    const uint32_t t0 = Q.tile_base[q0], valid = Q.tile_base[q1] - t0;
    int i = 0, j = 0;
    const bool mine = tid < PX && qtile < tiles && tile_pixel_f(ordered_tile(qtile, size) * 64 + (tid & 63), size, i, j);
    const size_t p = mine ? (size_t)i * size + j : 0;
    double ax = 0, ay = 0, az = 0;
    if (mine && !first_batch) { ax = F.gi_acc[3 * p]; ay = F.gi_acc[3 * p + 1]; az = F.gi_acc[3 * p + 2]; }
    bool shot = false;
    if (mine) {
        const double hx = F.ghit[3 * p], hy = F.ghit[3 * p + 1], hz = F.ghit[3 * p + 2];
        shot = !(hx == -99999);
        org[0][tid] = (float)hx; org[1][tid] = (float)hy; org[2][tid] = (float)hz;
    }
    __syncthreads();
    for (int g0 = 0; g0 < nb && valid; g0 += CH) {
        const int gs = nb - g0 < CH ? nb - g0 : CH;
        const uint32_t base = t0 * (uint32_t)nb + valid * (uint32_t)g0, n = valid * (uint32_t)gs;
        for (uint32_t k = tid; k < n; k += PT_FUSED_THREADS) {
            const uint32_t r = base + k;
            const uint32_t code = Q.ray_slot[r], px = code & (uint32_t)(PX - 1), at = code - (uint32_t)g0 * (uint32_t)PX;
            const float4 c = omp_contribution(S, org[0][px], org[1][px], org[2][px], Q.rays_d[r], Q.hits[r]);
            cr[at] = c.x; cg[at] = c.y; cb[at] = c.z;
        }
        __syncthreads();
        if (shot)
            for (int s = 0; s < gs; s++) { ax += (double)cr[s * PX + tid]; ay += (double)cg[s * PX + tid]; az += (double)cb[s * PX + tid]; }
        __syncthreads();
    }
    if (mine) { F.gi_acc[3 * p] = ax; F.gi_acc[3 * p + 1] = ay; F.gi_acc[3 * p + 2] = az; }
}

// ---- raw queries for tests: fp64 in/out around the fp32 traversal -------------------
template <class LS>
struct RawPolicy {
    const DevScene &S;
    const RootBox box;
    uint2 *stk;
    uint32_t slot_mask;
    LS L;
    const double *src, *dir, *tmin_in;
    int32_t *found, *tri;
    double *t, *u, *v;
    size_t k;
    __device__ __forceinline__ void miss() { found[k] = 0; tri[k] = -1; t[k] = tmin_in[k]; u[k] = 0; v[k] = 0; }
    __device__ __forceinline__ bool start(uint32_t item) {
        k = item;
        double tm = tmin_in[k];
        float tmax = tm >= (double)FLT_MAX ? FLT_MAX : (float)tm;
        if (lane_begin(L, S, box, (float)src[3 * k], (float)src[3 * k + 1], (float)src[3 * k + 2], (float)dir[3 * k],
                       (float)dir[3 * k + 1], (float)dir[3 * k + 2], tmax, -1))
            return true;
        miss();
        return false;
    }
    __device__ __forceinline__ bool step(bool active) { return active ? turn() : false; }
    __device__ __forceinline__ bool turn() {
        if (!lane_turn<false, false>(L, S, stk, slot_mask, nullptr)) return true;
        if (L.htri < 0) { miss(); return false; }
        found[k] = 1; tri[k] = L.htri; t[k] = (double)L.tmin; u[k] = (double)L.hu; v[k] = (double)L.hv;
        return false;
    }
};

template <class LS>
__global__ void __launch_bounds__(64) k_trace_fast_f64io(DevScene S, uint32_t n, uint32_t *head, const double *src, const double *dir,
                                                         const double *tmin_in, int32_t *found, int32_t *tri, double *t,
                                                         double *u, double *v, uint32_t slot_mask) {
    extern __shared__ uint2 stack[];
    RawPolicy<LS> P{S, root_box(S), stack + (threadIdx.x & 63), slot_mask, LS(), src, dir, tmin_in, found, tri, t, u, v, 0};
    persistent_loop(P, n, head, 64u, 1);
}

// ---- host side ------------------------------------------------------------------------
#define FQ(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return PT_ERR_HIP; } while (0)

// Two sets of ray arrays, pass counters and queue heads (Q[0], Q[1]: the generator fills one while the other is being
// traced and shaded); the shading-point queue, the contribution slots and the per-view data are shared.
int fast_alloc(FastBuffers Q[2], size_t pixels, int batch, int n_light, PtAllocFn alloc, void *ctx) {
    size_t n = pixels * (size_t)batch;
    const size_t n_shadow = n * (size_t)(n_light < 0 ? 0 : n_light);       // a shadow ray per point and light (0: no shadow-ray queue)
    if (n > 0xFFFFFF00u || n_shadow >= 0x80000000u) return PT_ERR_LIMIT;
    int rc;
#define FA(q, field, type) if ((rc = alloc(ctx, n * sizeof(type), (void **)&q.field))) return rc
    for (int k = 0; k < 2; k++) {
        FA(Q[k], rays_o, float4); FA(Q[k], rays_d, float4); FA(Q[k], ray_slot, uint32_t); FA(Q[k], hits, float4);
        if ((rc = alloc(ctx, (PT_PASS_COUNT_WORDS + 2 * PT_HEAD_WORDS) * sizeof(uint32_t), (void **)&Q[k].counts))) return rc;
        Q[k].heads = Q[k].counts + PT_PASS_COUNT_WORDS;
        Q[k].max_rays = n;
    }
    FA(Q[0], sp_pos, float4); FA(Q[0], sp_nrm, float4); FA(Q[0], sp_base, float4); FA(Q[0], contrib, float4);
#undef FA
    if (n_shadow) {
        if ((rc = alloc(ctx, n_shadow * sizeof(float4), (void **)&Q[0].sh_o))) return rc;
        if ((rc = alloc(ctx, n_shadow * sizeof(float4), (void **)&Q[0].sh_d))) return rc;
    }
    if ((rc = alloc(ctx, 2 * sizeof(unsigned long long), (void **)&Q[0].totals))) return rc;
    if ((rc = alloc(ctx, 2 * sizeof(unsigned long long), (void **)&Q[0].work))) return rc;
    if ((rc = alloc(ctx, ((pixels + 63) / 64 * 4 + 2) * sizeof(uint32_t), (void **)&Q[0].tile_base))) return rc;
    Q[1].sp_pos = Q[0].sp_pos; Q[1].sp_nrm = Q[0].sp_nrm; Q[1].sp_base = Q[0].sp_base; Q[1].contrib = Q[0].contrib;
    Q[1].sh_o = Q[0].sh_o; Q[1].sh_d = Q[0].sh_d;
    Q[1].totals = Q[0].totals; Q[1].work = Q[0].work; Q[1].tile_base = Q[0].tile_base;
    return PT_OK;
}
// Launch parameters of the closest-hit kernel.  The defaults are what the
// measurements in profiles/ were taken with; the PT_TUNE_* environment
// variables exist for sweeps.
struct TraceTune { int slots, refill, waves, chunk, gen_lds, shade_lds, gen_threads, sort_tiles; };
// Tiles the generator sorts as one run in a pass of nb samples per pixel: 4 when the pass is short (see k_gi_gen).
static int sort_tiles(int nb, const TraceTune &t) { return nb <= 16 ? t.sort_tiles : 1; }
static TraceTune trace_tune() {
    TraceTune t = [] {
        TraceTune v{FS_SLOTS, PT_REFILL_MIN, PT_TRACE_WAVES, PT_QUEUE_CHUNK, 0, PT_SHADE_LDS_BYTES, PT_GEN_THREADS, 4};
        if (const char *e = getenv("PT_TUNE_SORT_TILES")) { int k = atoi(e); if (k == 1 || k == 4) v.sort_tiles = k; }
        if (const char *e = getenv("PT_TUNE_GEN_THREADS")) { int k = atoi(e); if (k == 512 || k == 1024) v.gen_threads = k; }
        if (const char *e = getenv("PT_TUNE_GEN_LDS")) { int k = atoi(e); if (k >= 0 && k <= 150000) v.gen_lds = k; }
        if (const char *e = getenv("PT_TUNE_SHADE_LDS")) { int k = atoi(e); if (k >= 0 && k <= 150000) v.shade_lds = k; }
        if (const char *e = getenv("PT_TUNE_SLOTS")) { int k = atoi(e); if (k == 2 || k == 4 || k == 8 || k == 16 || k == 32) v.slots = k; }
        if (const char *e = getenv("PT_TUNE_REFILL")) { int k = atoi(e); if (k >= 1 && k <= 64) v.refill = k; }
        if (const char *e = getenv("PT_TUNE_CHUNK")) { int k = atoi(e); if (k >= 16 && k <= 65536) v.chunk = k; }
        if (const char *e = getenv("PT_TUNE_WAVES")) { int k = atoi(e); if (k >= 64 && k <= PT_TRACE_WAVES) v.waves = k; }
        return v;
    }();
    return t;
}

static bool use_quads() {
    static const bool on = !(getenv("PT_TUNE_QUAD") && atoi(getenv("PT_TUNE_QUAD")) == 0);
    return on;
}

int fast_gi(const DevScene &S, const ViewParams &V, const FrameBuffers &F, const FastBuffers QQ[2], int nsamples, int batch,
            hipStream_t s, hipStream_t s_gen, std::vector<hipEvent_t> &events, float *trace_ms, uint64_t *launches, uint64_t *rays,
            uint64_t *shadow, uint64_t *work /* null, or [2]: nodes visited, triangles tested by the closest-hit launches */) {
    const int tthreads = tile_threads(V.size);
    const size_t P = (size_t)V.size * V.size;
    const TraceTune tune = trace_tune();
    *trace_ms = 0; *launches = 0; *rays = 0; *shadow = 0;
    // The passes of a view are queued back to back; the host waits once, at the end.  Pass i traces, shades and sums on
    // stream `s` out of ray set i & 1 while the generator fills the other set for pass i + 1 on stream `s_gen`.
    // Events per pass: [4i] generation done, [4i + 1] / [4i + 2] around the closest-hit launch, [4i + 3] pass done.
    const int npass = (nsamples + batch - 1) / batch;
    while (events.size() < 4 * (size_t)npass + 1) {
        hipEvent_t e;
        FQ(hipEventCreate(&e));
        events.push_back(e);
    }
    const FastBuffers &Q0 = QQ[0];
    if (work) FQ(hipMemsetAsync(Q0.work, 0, 2 * sizeof(unsigned long long), s));
    FQ(hipMemsetAsync(Q0.totals, 0, 2 * sizeof(unsigned long long), s));
    const int tiles = tthreads / 64;
    hipLaunchKernelGGL(k_tile_valid, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, F, Q0, V.size, tiles);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, Q0, tiles);
    hipEvent_t view_ready = events[4 * (size_t)npass];             // G-buffer and tile offsets are in place
    FQ(hipEventRecord(view_ready, s));
    auto generate = [&](int pass) -> int {
        const FastBuffers &Q = QQ[pass & 1];
        const int s0 = pass * batch, nb = nsamples - s0 < batch ? nsamples - s0 : batch;
        const int G = sort_tiles(nb, tune), spb = tune.gen_threads / (16 * G), groups = (nb + spb - 1) / spb;
        const unsigned blocks = (unsigned)(((tiles + G - 1) / G) * groups);
        // the ray set is free once the pass before the previous one is done (pass 0 and 1: once the view is ready);
        // the generator then runs beside the start of the previous pass's traversal.  (Holding it back until that
        // traversal has ended, so that it runs beside the shading instead, was slower.)
        FQ(hipStreamWaitEvent(s_gen, pass >= 2 ? events[4 * (pass - 2) + 3] : view_ready, 0));
        FQ(hipMemsetAsync(Q.counts, 0, (PT_PASS_COUNT_WORDS + 2 * PT_HEAD_WORDS) * sizeof(uint32_t), s_gen));
#define PT_GEN_LAUNCH(T, G) hipLaunchKernelGGL((k_gi_gen<T, G>), dim3(blocks), dim3(T), PT_GEN_LDS_BYTES(T, G) + (size_t)tune.gen_lds, s_gen, S, V, F, Q, s0, nb, groups, tiles)
        if (tune.gen_threads == 512) { if (G == 4) PT_GEN_LAUNCH(512, 4); else PT_GEN_LAUNCH(512, 1); }
        else { if (G == 4) PT_GEN_LAUNCH(1024, 4); else PT_GEN_LAUNCH(1024, 1); }
#undef PT_GEN_LAUNCH
        FQ(hipEventRecord(events[4 * pass], s_gen));
        return PT_OK;
    };
    if (int rc = generate(0)) return rc;
    for (int pass = 0; pass < npass; pass++) {
        const FastBuffers &Q = QQ[pass & 1];
        const int s0 = pass * batch, nb = nsamples - s0 < batch ? nsamples - s0 : batch, G = sort_tiles(nb, tune);
        if ((size_t)nb * P > Q.max_rays) return PT_ERR_LIMIT;
        if (V.model == PT_MODEL_MAIN && !Q.sh_o) return PT_ERR_LIMIT;     // the shadow-ray queue was not asked for
        if (pass + 1 < npass)
            if (int rc = generate(pass + 1)) return rc;
        unsigned sblocks = (unsigned)((P * nb + PT_APPEND_BLOCK - 1) / PT_APPEND_BLOCK);
        unsigned qblocks = (unsigned)((P * nb + 63) / 64);
        FQ(hipStreamWaitEvent(s, events[4 * pass], 0));
        FQ(hipEventRecord(events[4 * pass + 1], s));
        unsigned tblocks = qblocks < (unsigned)tune.waves ? qblocks : (unsigned)tune.waves;
        size_t lds = (size_t)tune.slots * 64 * sizeof(uint2);
        const uint32_t chunk = (uint32_t)tune.chunk;
        // quad records (two tree levels per request) whenever the tree has them; PT_TUNE_QUAD=0 walks the binary records
        const bool quads = S.has_quads && use_quads();
        const uint32_t smask = (uint32_t)tune.slots - 1u;
        if (quads && work)
            hipLaunchKernelGGL((k_trace_closest<true, LaneStateQ>), dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
        else if (quads)
            hipLaunchKernelGGL((k_trace_closest<false, LaneStateQ>), dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
        else if (work)
            hipLaunchKernelGGL((k_trace_closest<true, LaneState>), dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
        else
            hipLaunchKernelGGL((k_trace_closest<false, LaneState>), dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
        FQ(hipEventRecord(events[4 * pass + 2], s));
#ifdef PT_TRACE_CLOCKS
        {   // instrumented build: when the waves of this launch began and ended, as percentiles after the first wave's start (us)
            static std::vector<unsigned long long> c(2 * (size_t)PT_TRACE_WAVES);
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_trace_clk), c.size() * 8) == hipSuccess) {
                std::vector<double> b(c.begin(), c.begin() + tblocks), e(c.begin() + PT_TRACE_WAVES, c.begin() + PT_TRACE_WAVES + tblocks);
                std::sort(b.begin(), b.end()); std::sort(e.begin(), e.end());
                const double t0 = b[0];
                auto pc = [&](const std::vector<double> &v, double q) { return (v[(size_t)(q * (v.size() - 1))] - t0) / 100.0; };
                fprintf(stderr, "trace_clk rays %zu waves %u begin p50 %.1f p99 %.1f max %.1f  end p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n", (size_t)(P * nb), tblocks,
                        pc(b, .5), pc(b, .99), pc(b, 1), pc(e, .01), pc(e, .1), pc(e, .5), pc(e, .9), pc(e, .99), pc(e, 1));
            }
        }
#endif
        if (V.model == PT_MODEL_MAIN) {
            hipLaunchKernelGGL(k_shade_gi<PT_MODEL_MAIN>, dim3(sblocks), dim3(PT_APPEND_BLOCK), (size_t)tune.shade_lds, s, S, Q);
            if (quads)
                hipLaunchKernelGGL(k_shadow_gi<LaneStateQ>, dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
            else
                hipLaunchKernelGGL(k_shadow_gi<LaneState>, dim3(tblocks), dim3(64), lds, s, S, Q, smask, tune.refill, chunk);
            hipLaunchKernelGGL(k_light_sum, dim3((unsigned)((P * nb + 255) / 256)), dim3(256), 0, s, S, Q);
        } else {
            static const bool fused = !(getenv("PT_TUNE_FUSED_SUM") && atoi(getenv("PT_TUNE_FUSED_SUM")) == 0);
            if (fused) {
                if (G == 4) hipLaunchKernelGGL(k_shade_sum_omp<4>, dim3((unsigned)((tiles + 3) / 4)), dim3(PT_FUSED_THREADS), 0, s, S, F, Q, V.size, nb, s0 == 0, tiles);
                else hipLaunchKernelGGL(k_shade_sum_omp<1>, dim3((unsigned)tiles), dim3(PT_FUSED_THREADS), 0, s, S, F, Q, V.size, nb, s0 == 0, tiles);
                FQ(hipEventRecord(events[4 * pass + 3], s));
                FQ(hipGetLastError());
                continue;
            }
            hipLaunchKernelGGL(k_shade_gi<PT_MODEL_OMP>, dim3(sblocks), dim3(PT_APPEND_BLOCK), (size_t)tune.shade_lds, s, S, Q);
        }
        if (G == 4) hipLaunchKernelGGL(k_gi_sum<4>, dim3((unsigned)((tiles + 3) / 4)), dim3(PT_SUM_THREADS), 0, s, F, Q, V.size, nb, s0 == 0, tiles);
        else hipLaunchKernelGGL(k_gi_sum<1>, dim3((unsigned)tiles), dim3(PT_SUM_THREADS), 0, s, F, Q, V.size, nb, s0 == 0, tiles);
        FQ(hipEventRecord(events[4 * pass + 3], s));
        FQ(hipGetLastError());
    }
    unsigned long long tot[2];
    FQ(hipMemcpyAsync(tot, Q0.totals, sizeof tot, hipMemcpyDeviceToHost, s));
    FQ(hipStreamSynchronize(s));
    *rays = tot[0]; *shadow = tot[1];
    for (int k = 0; k < npass; k++) {
        float ms = 0;
        FQ(hipEventElapsedTime(&ms, events[4 * k + 1], events[4 * k + 2]));
        *trace_ms += ms; (*launches)++;
    }
    if (work) {
        unsigned long long w[2];
        FQ(hipMemcpy(w, Q0.work, sizeof w, hipMemcpyDeviceToHost));
        work[0] = w[0]; work[1] = w[1];
    }
    return PT_OK;
}

// ---- the raw query without any allocation ------------------------------------------------
// Callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops) get the same search with
// its two scratch buffers -- the queue heads and the traversal overflow stack -- in a fixed array of the code object.
// Every GPU that loads the library holds its own copy of the array, at its own address, until the process ends.
// Layout: [0, PT_STATIC_HEAD_BYTES) queue heads; then the overflow stack [wave][entry][lane].
#define PT_STATIC_HEAD_BYTES (PT_HEAD_WORDS * sizeof(uint32_t))
#define PT_STATIC_BYTES (PT_STATIC_HEAD_BYTES + (size_t)PT_STATIC_WAVES * PT_STATIC_DEPTH * 64 * sizeof(uint2))
static __device__ __align__(256) unsigned char scratch[PT_STATIC_BYTES];
// Device address of `scratch` per device ordinal.  In host code the array's name is only a handle for
// hipGetSymbolAddress; kernels, memsets and copies take these addresses.
static unsigned char *g_scratch_ptr[PT_STATIC_MAX_DEVICES];

int static_scratch_setup(int device) {
    if (device < 0 || device >= PT_STATIC_MAX_DEVICES) return PT_ERR_LIMIT;
    if (g_scratch_ptr[device]) return PT_OK;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return PT_ERR_HIP;
    g_scratch_ptr[device] = (unsigned char *)p;
    return PT_OK;
}

// (mesh, triangle of the mesh) of a scene-wide triangle index, the reference's `pii tind` (util/kdtree.cpp:192).
__global__ void k_tri_to_mesh_local(DevScene S, uint32_t n, int32_t *tri, int32_t *mesh) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t g = tri[k];
    mesh[k] = g < 0 ? -1 : S.tri_mesh[g];
    tri[k] = g < 0 ? -1 : S.tri_local[g];
}

int fast_trace_static(const DevScene &S0, int device, size_t n, const double *src, const double *dir, const double *tmin_in,
                      int32_t *found, int32_t *mesh, int32_t *tri, double *t, double *u, double *v, hipStream_t s) {
    if (device < 0 || device >= PT_STATIC_MAX_DEVICES || !g_scratch_ptr[device]) return PT_ERR_NO_DEVICE;
    if (n > PT_STATIC_MAX_RAYS) return PT_ERR_LIMIT;              // the caller has checked the tree's depth against PT_STATIC_DEPTH
    unsigned char *base = g_scratch_ptr[device];
    DevScene S = S0;
    S.ovf = (uint2 *)(base + PT_STATIC_HEAD_BYTES);
    S.ovf_depth = PT_STATIC_DEPTH;
    S.ovf_waves = PT_STATIC_WAVES;
    FQ(hipMemsetAsync(base, 0, PT_STATIC_HEAD_BYTES, s));          // the heads start at zero; the stack is written before it is read
    const TraceTune tune = trace_tune();
    unsigned blocks = (unsigned)((n + 63) / 64);
    if (blocks > PT_STATIC_WAVES) blocks = PT_STATIC_WAVES;
    // binary records here: one stack entry per tree level, which is what the fixed scratch is sized for
    hipLaunchKernelGGL(k_trace_fast_f64io<LaneState>, dim3(blocks), dim3(64), (size_t)tune.slots * 64 * sizeof(uint2), s, S, (uint32_t)n,
                       (uint32_t *)base, src, dir, tmin_in, found, tri, t, u, v, (uint32_t)tune.slots - 1u);
    hipLaunchKernelGGL(k_tri_to_mesh_local, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S, (uint32_t)n, tri, mesh);
    FQ(hipGetLastError());
    FQ(hipStreamSynchronize(s));
    return PT_OK;
}

int fast_trace(const DevScene &S, size_t n, const double *src, const double *dir, const double *tmin_in,
               int32_t *found, int32_t *tri, double *t, double *u, double *v, hipStream_t s) {
    uint32_t *head = nullptr;
    if (n >= (1ull << 31)) return PT_ERR_LIMIT;
    if (hipMalloc((void **)&head, PT_HEAD_WORDS * sizeof(uint32_t)) != hipSuccess) return PT_ERR_HIP;
    hipError_t e = hipMemsetAsync(head, 0, PT_HEAD_WORDS * sizeof(uint32_t), s);
    const TraceTune tune = trace_tune();
    unsigned blocks = (unsigned)((n + 63) / 64);
    if (blocks > (unsigned)tune.waves) blocks = (unsigned)tune.waves;
    if (e == hipSuccess) {
        if (S.has_quads && use_quads())
            hipLaunchKernelGGL(k_trace_fast_f64io<LaneStateQ>, dim3(blocks), dim3(64), (size_t)tune.slots * 64 * sizeof(uint2), s, S, (uint32_t)n, head,
                               src, dir, tmin_in, found, tri, t, u, v, (uint32_t)tune.slots - 1u);
        else
            hipLaunchKernelGGL(k_trace_fast_f64io<LaneState>, dim3(blocks), dim3(64), (size_t)tune.slots * 64 * sizeof(uint2), s, S, (uint32_t)n, head,
                               src, dir, tmin_in, found, tri, t, u, v, (uint32_t)tune.slots - 1u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(head);
    return e == hipSuccess ? PT_OK : PT_ERR_HIP;
}
