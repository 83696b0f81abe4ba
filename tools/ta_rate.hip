// Micro-benchmark: throughput of fully divergent per-lane loads of 4, 8 and 16 bytes that hit in L1, in L2 or
// beyond (what bounds the kd traversal).  Prints lane-loads per clock per CU (clock taken as 2.4 GHz, 256 CUs).
//   make tools/ta_rate && tools/ta_rate [--json]
// --json: one line {"l1": .., "l2": .., "beyond": ..} with the 16-byte figures, which bench.py reads in the same run
// as its timings.  MI355X: 8 KB (L1) 1.86, 1 MB (L2) 0.45, 64 MB 0.10 -- for every load width.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

template <int BYTES>
__global__ void __launch_bounds__(256) k_gather(const char *base, uint32_t mask, int iters, uint32_t *sink) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t off = ((x >> 8) & mask) * 16u;                 // 16-byte slots, random per lane
        if (BYTES == 4) acc += *(const uint32_t *)(base + off);
        else if (BYTES == 8) { uint2 v = *(const uint2 *)(base + off); acc += v.x ^ v.y; }
        else { uint4 v = *(const uint4 *)(base + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES>
double run(const char *d, uint32_t slots, uint32_t *sink) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<BYTES>, dim3(blocks), dim3(256), 0, 0, d, slots - 1, 64, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_gather<BYTES>, dim3(blocks), dim3(256), 0, 0, d, slots - 1, iters, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters / (ms * 1e-3);              // lane-loads per second
}

// A coalesced read of a whole buffer, for the calibration below: every byte is fetched exactly once.
__global__ void __launch_bounds__(256) k_stream(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// --calibrate divergent|stream: ONE launch whose memory-side read traffic is known, to be run under
// `rocprofv3 --pmc FETCH_SIZE` (and, in a run of its own, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`): what does the
// counter report for fully divergent 16-byte loads that miss every cache (4 GB buffer, 16 times the Infinity Cache), and
// what for a coalesced read of the same buffer?  Prints the launch's loads and the bytes they must bring in.
static int calibrate(const char *what) {
    const size_t bytes = (size_t)4 << 30;
    char *d; uint32_t *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    if (std::strcmp(what, "divergent") == 0) {
        const int iters = 1024, blocks = 256 * 8;
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)(bytes / 16 - 1), iters, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        const double loads = (double)blocks * 256 * iters;
        std::printf("{\"case\": \"divergent\", \"kernel\": \"k_gather<16>\", \"lane_loads\": %.0f, \"bytes_if_64B_per_load\": %.0f, \"bytes_if_128B_per_load\": %.0f, "
                    "\"ms\": %.3f, \"tb_per_s_if_64B\": %.2f, \"tb_per_s_if_128B\": %.2f}\n",
                    loads, loads * 64, loads * 128, ms, loads * 64 / (ms * 1e-3) / 1e12, loads * 128 / (ms * 1e-3) / 1e12);
    } else {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        std::printf("{\"case\": \"stream\", \"kernel\": \"k_stream\", \"bytes\": %.0f, \"ms\": %.3f, \"tb_per_s\": %.2f}\n", (double)bytes, ms, bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && std::strcmp(argv[1], "--calibrate") == 0) return calibrate(argv[2]);
    const bool json = argc > 1 && std::strcmp(argv[1], "--json") == 0;
    uint32_t *sink;
    if (hipMalloc(&sink, 4) != hipSuccess) return 1;
    double r16s[3];
    int level = 0;
    for (uint32_t kb : {8u, 1024u, 65536u}) {
        const uint32_t slots = kb * 1024u / 16u;
        char *d;
        if (hipMalloc(&d, (size_t)slots * 16) != hipSuccess) return 1;
        (void)hipMemset(d, 1, (size_t)slots * 16);
        const double r4 = run<4>(d, slots, sink), r8 = run<8>(d, slots, sink), r16 = run<16>(d, slots, sink);
        const double per = 256.0 * 2.4e9;                             // CUs x clock
        if (!json) std::printf("%6u KB: 4 B %.2f  8 B %.2f  16 B %.2f lane-loads/clk/CU\n", kb, r4 / per, r8 / per, r16 / per);
        r16s[level++] = r16 / per;
        (void)hipFree(d);
    }
    if (json) std::printf("{\"l1\": %.4f, \"l2\": %.4f, \"beyond\": %.4f}\n", r16s[0], r16s[1], r16s[2]);
    return 0;
}
