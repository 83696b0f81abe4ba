// fp64 3-vectors for device code.  Every operation is spelled out component by
// component in the order the reference evaluates it (util/utilities.cpp:72-162);
// the translation unit is built with -ffp-contract=off so that no multiply-add
// is fused and results match an x86-64 SSE2 build bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

// Everything in this header is meant to round after every operation.
#pragma clang fp contract(off)

struct D3 { double x, y, z; };

PT_DEV D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV D3 dadd(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV D3 dsub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV D3 dscale(double k, D3 a) { return d3(k * a.x, k * a.y, k * a.z); }
PT_DEV D3 dmul(D3 a, D3 b) { return d3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV D3 ddivs(D3 a, double k) { return d3(a.x / k, a.y / k, a.z / k); }
PT_DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV D3 dcross(D3 a, D3 b) {
    return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_DEV double dlen(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_DEV D3 dunit(D3 a) { double m = dlen(a); return d3(a.x / m, a.y / m, a.z / m); }
PT_DEV D3 dreflect(D3 u, D3 n) { return dsub(u, dscale(2 * ddot(u, n), n)); }   // util/utilities.cpp:160-162
PT_DEV double dget(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
PT_DEV D3 dload(const double *p) { return d3(p[0], p[1], p[2]); }
PT_DEV void dstore(double *p, D3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// Ray, util/polygonMesh.cpp:10-20.  Its reciprocal direction and sign bits are read by the box test only, which
// forms them itself (pt_exact.h).
struct RayD { D3 src, dir; };

PT_DEV RayD make_ray(D3 src, D3 dir) {
    RayD r;
    r.src = src; r.dir = dir;
    return r;
}

// ---- counter-based random numbers -------------------------------------------
// Philox4x32-10; key = (seed, view), counter = (pixel, sample, 0, 0).
PT_DEV void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// 53-bit uniform in [0,1) from two words.
PT_DEV double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// cos and sin of 2*pi*r for r in [0,1): exact octant reduction, then Taylor
// polynomials on [0, pi/4] by Horner in x*x with separate multiplies and adds.
// Written once with a fixed operation order so that host and device agree bit
// for bit (libm's sin/cos differ between platforms in the last place).
PT_DEV void sincos_2pi(double r, double &cs, double &sn) {
    double q = r * 8.0;
    int k = (int)q;
    double f = q - (double)k;
    if (k & 1) f = 1.0 - f;
    double x = f * 0.78539816339744830962;
    double z = x * x;
    double s = 2.81145725434552076320e-15;
    s = s * z + -7.64716373181981647590e-13;
    s = s * z + 1.60590438368216145994e-10;
    s = s * z + -2.50521083854417187751e-08;
    s = s * z + 2.75573192239858906526e-06;
    s = s * z + -1.98412698412698412698e-04;
    s = s * z + 8.33333333333333333333e-03;
    s = s * z + -1.66666666666666666667e-01;
    s = s * z + 1.0;
    s = s * x;
    double c = 4.77947733238738529744e-14;
    c = c * z + -1.14707455977297247139e-11;
    c = c * z + 2.08767569878680989792e-09;
    c = c * z + -2.75573192239858906526e-07;
    c = c * z + 2.48015873015873015873e-05;
    c = c * z + -1.38888888888888888889e-03;
    c = c * z + 4.16666666666666666667e-02;
    c = c * z + -0.5;
    c = c * z + 1.0;
    switch (k & 7) {
        case 0: cs = c;  sn = s;  break;
        case 1: cs = s;  sn = c;  break;
        case 2: cs = -s; sn = c;  break;
        case 3: cs = -c; sn = s;  break;
        case 4: cs = -c; sn = -s; break;
        case 5: cs = -s; sn = -c; break;
        case 6: cs = s;  sn = -c; break;
        default: cs = c; sn = -s; break;
    }
}

// uniformSampleHemisphere of util/lighting.cpp:53-59 on the counter RNG:
// returns (sinTheta*cos(phi), r1, sinTheta*sin(phi)).
PT_DEV D3 sample_hemisphere(uint32_t seed, uint32_t view, uint32_t pixel, uint32_t sample, double &r1) {
    uint32_t c[4] = {pixel, sample, 0u, 0u};
    philox4x32_10(c, seed, view);
    r1 = u53(c[0], c[1]);
    double r2 = u53(c[2], c[3]);
    double cs, sn;
    sincos_2pi(r2, cs, sn);
    double sinTheta = sqrt(1 - r1 * r1);
    return d3(sinTheta * cs, r1, sinTheta * sn);
}
