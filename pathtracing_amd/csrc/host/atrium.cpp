// "atrium": a deterministic procedural hall used where the reference's large
// meshes (sponza2.txt, bedroom.txt) are not available.  It is NOT the Crytek
// Sponza; it only has the same order of size and the same mix of features that
// stress a kd-tree: huge floor / wall polygons next to dense small triangles,
// long thin columns, curved arches, thin draped sheets and a finely displaced
// relief.  One mesh, per-hit textured groups ('?' mode) with procedural
// textures, one point light, camera inside the nave.
//
// make_atrium(T, seed) emits exactly T triangles (T >= 2000): the tessellation
// of the fixed parts scales with sqrt(T / 262267) and the relief plus a cloud of
// small "debris" triangles absorb the remainder.
#include <cmath>
#include <stdexcept>

#include "scene.hpp"

namespace pth {

namespace {

struct Rng {                         // xorshift64*, seeded
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { if (!s) s = 1; }
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Builder {
    Mesh &m;
    int32_t tm = -99999, reft = 0;
    explicit Builder(Mesh &mesh) : m(mesh) {}

    int v(double x, double y, double z) { m.vert.push_back({x, y, z}); return (int)m.vert.size() - 1; }
    int n(double x, double y, double z) {
        double l = std::sqrt(x * x + y * y + z * z);
        if (!(l > 0) || !std::isfinite(l)) { x = 0; y = 1; z = 0; l = 1; }
        m.norm.push_back({x / l, y / l, z / l});
        return (int)m.norm.size() - 1;
    }
    int t(double a, double b) {
        auto c = [](double q) { return std::max(0.0, std::min(0.99, q)); };   // the reader's clamp, util/reader.cpp:89
        m.uv.push_back({c(a), c(b)});
        return (int)m.uv.size() - 1;
    }
    void tri(int a, int b, int c, int na, int nb, int nc, int ta, int tb, int tc) {
        Triangle T;
        T.vi[0] = a; T.vi[1] = b; T.vi[2] = c;
        T.ni[0] = na; T.ni[1] = nb; T.ni[2] = nc;
        T.ti[0] = ta; T.ti[1] = tb; T.ti[2] = tc;
        T.tm = tm; T.reft = reft;
        m.tris.push_back(T);
    }

    // Parametric sheet P(a, b), a, b in [0, 1], na x nb cells, smooth normals from central differences.
    template <typename F>
    void sheet(int na, int nb, F P, double urep = 1.0, double vrep = 1.0) {
        std::vector<int> vi((size_t)(na + 1) * (nb + 1)), ni(vi.size()), ti(vi.size());
        const double e = 1e-4;
        for (int i = 0; i <= na; i++)
            for (int j = 0; j <= nb; j++) {
                double a = (double)i / na, b = (double)j / nb;
                // differences stay inside [0, 1]: profile functions need not be defined outside
                D3 p = P(a, b), pa = P(std::min(a + e, 1.0), b), pb = P(a, std::min(b + e, 1.0)), qa = P(std::max(a - e, 0.0), b),
                   qb = P(a, std::max(b - e, 0.0));
                D3 da{pa[0] - qa[0], pa[1] - qa[1], pa[2] - qa[2]}, db{pb[0] - qb[0], pb[1] - qb[1], pb[2] - qb[2]};
                size_t k = (size_t)i * (nb + 1) + j;
                vi[k] = v(p[0], p[1], p[2]);
                ni[k] = n(da[1] * db[2] - da[2] * db[1], da[2] * db[0] - da[0] * db[2], da[0] * db[1] - da[1] * db[0]);
                double uu = a * urep, vv = b * vrep;
                ti[k] = t(0.99 * (uu - std::floor(uu)), 0.99 * (vv - std::floor(vv)));
            }
        for (int i = 0; i < na; i++)
            for (int j = 0; j < nb; j++) {
                size_t k00 = (size_t)i * (nb + 1) + j, k10 = k00 + (nb + 1), k01 = k00 + 1, k11 = k10 + 1;
                tri(vi[k00], vi[k10], vi[k11], ni[k00], ni[k10], ni[k11], ti[k00], ti[k10], ti[k11]);
                tri(vi[k00], vi[k11], vi[k01], ni[k00], ni[k11], ni[k01], ti[k00], ti[k11], ti[k01]);
            }
    }
    static size_t sheet_tris(int na, int nb) { return (size_t)na * nb * 2; }

    // Axis-aligned rectangle as a grid of flat cells; (o + a*U + b*V), normal N.
    void rect(D3 o, D3 U, D3 V, int na, int nb, double urep, double vrep) {
        sheet(na, nb, [&](double a, double b) { return D3{o[0] + a * U[0] + b * V[0], o[1] + a * U[1] + b * V[1], o[2] + a * U[2] + b * V[2]}; },
              urep, vrep);
    }
};

// Procedural textures, BGR 8-bit.
Texture tex_bricks(int n, uint64_t seed) {
    Texture t; t.rows = t.cols = n; t.bgr.resize((size_t)n * n * 3);
    Rng r(seed);
    std::vector<double> tone(64 * 64);
    for (double &x : tone) x = 0.75 + 0.25 * r.uni();
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int row = y * 16 / n, col = ((x + (row & 1) * n / 16) * 8 / n) & 63;
            bool mortar = (y * 16 % n) < n / 20 || ((x + (row & 1) * n / 16) * 8 % n) < n / 20;
            double k = mortar ? 0.55 : tone[(row * 64 + col) & 4095];
            uint8_t *p = &t.bgr[((size_t)y * n + x) * 3];
            p[0] = (uint8_t)(120 * k); p[1] = (uint8_t)(150 * k); p[2] = (uint8_t)(215 * k);
        }
    return t;
}
Texture tex_cloth(int n, int b, int g, int rr) {
    Texture t; t.rows = t.cols = n; t.bgr.resize((size_t)n * n * 3);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            double k = 0.7 + 0.3 * std::sin(x * 0.35) * std::sin(y * 0.35) * 0.5 + 0.15 * ((x / 8 + y / 8) & 1);
            uint8_t *p = &t.bgr[((size_t)y * n + x) * 3];
            p[0] = (uint8_t)std::min(255.0, b * k); p[1] = (uint8_t)std::min(255.0, g * k); p[2] = (uint8_t)std::min(255.0, rr * k);
        }
    return t;
}

}  // namespace

void make_atrium(uint32_t triangles, uint32_t seed, Scene &sc) {
    if (triangles < 2000) throw std::runtime_error("atrium: at least 2000 triangles");
    sc = Scene();
    sc.eye = {-100, 300, -50};
    sc.step = 50; sc.astep = 0.01745329251 * 90; sc.angle = 0.0; sc.zoom = 1.0;
    Light L; L.type = 'p'; L.mag = 35000; L.pos = {0, 400, 0}; L.color = {1, 1, 1};
    sc.lights.push_back(L);
    sc.meshes.emplace_back();
    Mesh &m = sc.meshes.back();
    m.tmaps.push_back(tex_bricks(256, seed + 1));           // 0: walls, floor
    m.tmaps.push_back(tex_cloth(256, 60, 70, 200));         // 1: red drapes
    m.tmaps.push_back(tex_cloth(256, 190, 120, 60));        // 2: blue drapes
    Builder B(m);
    Rng rng(seed);
    const double PI = 3.14159265358979323846;
    const double s = std::sqrt((double)triangles / 262267.0);
    auto seg = [&](int full, int least) { return std::max(least, (int)std::lround(full * s)); };
    const double X0 = -1900, X1 = 1800, Y1 = 1500, Z0 = -1100, Z1 = 1200;

    // ---- shell: floor, ceiling, four walls (large cells) ----
    B.tm = -1;                                                // texture 0, looked up per hit
    B.rect({X0, 0, Z0}, {0, 0, Z1 - Z0}, {X1 - X0, 0, 0}, 4, 8, 6, 10);          // floor (normal +y)
    B.rect({X0, Y1, Z0}, {X1 - X0, 0, 0}, {0, 0, Z1 - Z0}, 8, 4, 10, 6);         // ceiling (normal -y)
    B.rect({X0, 0, Z0}, {X1 - X0, 0, 0}, {0, Y1, 0}, 6, 3, 10, 4);               // wall z = Z0
    B.rect({X0, 0, Z1}, {0, Y1, 0}, {X1 - X0, 0, 0}, 3, 6, 4, 10);               // wall z = Z1
    B.rect({X0, 0, Z0}, {0, Y1, 0}, {0, 0, Z1 - Z0}, 3, 4, 4, 6);                // wall x = X0
    B.rect({X1, 0, Z0}, {0, 0, Z1 - Z0}, {0, Y1, 0}, 4, 3, 6, 4);                // wall x = X1
    // gallery slabs of the upper storey along both colonnades
    for (int side = 0; side < 2; side++) {
        double za = side ? 400 : Z0, zb = side ? Z1 : -400;
        B.rect({X0, 640, za}, {0, 0, zb - za}, {X1 - X0, 0, 0}, 2, 12, 3, 10);   // top
        B.rect({X0, 600, za}, {X1 - X0, 0, 0}, {0, 0, zb - za}, 12, 2, 10, 3);   // underside
        double zf = side ? 400 : -400;
        B.rect({X0, 600, zf}, {X1 - X0, 0, 0}, {0, 40, 0}, 12, 1, 10, 1);        // front face
    }

    // ---- colonnades: 2 rows x 12 columns x 2 storeys, arches between neighbours ----
    B.tm = -99999;
    const int ncol = 12;
    const int cs = seg(32, 6), cr = seg(24, 2);
    const int as = seg(24, 4), at = seg(12, 3);
    for (int row = 0; row < 2; row++)
        for (int st = 0; st < 2; st++) {
            double z = row ? 400.0 : -400.0, y0 = st ? 640.0 : 0.0, h = st ? 560.0 : 600.0;
            for (int c = 0; c < ncol; c++) {
                double x = X0 + 200 + (X1 - X0 - 400) * c / (ncol - 1);
                double rad = 55 + 6 * rng.uni();
                B.sheet(cs, cr, [&](double a, double b) {
                    double ang = 2 * PI * a, bulge = 1.0 + 0.08 * std::sin(PI * b);
                    return D3{x + rad * bulge * std::cos(ang), y0 + h * b, z - rad * bulge * std::sin(ang)};
                });
                if (c + 1 < ncol) {                                            // half-torus arch to the next column
                    double x2 = X0 + 200 + (X1 - X0 - 400) * (c + 1) / (ncol - 1);
                    double R = 0.5 * (x2 - x), xm = 0.5 * (x + x2), ytop = y0 + h - R * 0.35;
                    B.sheet(as, at, [&](double a, double b) {
                        double th = PI * a, ph = 2 * PI * b, rr = 28;
                        double cx = xm - R * std::cos(th), cy = ytop + 0.35 * R * std::sin(th);
                        return D3{cx - rr * std::cos(ph) * std::cos(th), cy + rr * std::cos(ph) * std::sin(th) * 0.35 + 0.0, z + rr * std::sin(ph)};
                    });
                }
            }
        }

    // ---- lathe vases on the nave floor ----
    const int vs = seg(48, 8), vr = seg(40, 6);
    for (int k = 0; k < 8; k++) {
        double x = X0 + 500 + 400.0 * k, z = (k & 1) ? 150.0 : -150.0, hgt = 160 + 40 * rng.uni();
        B.sheet(vs, vr, [&](double a, double b) {
            double ang = 2 * PI * a;
            double prof = 18 + 55 * std::sin(PI * std::pow(b, 0.8)) + 10 * std::sin(6 * PI * b);
            return D3{x + prof * std::cos(ang), hgt * b, z - prof * std::sin(ang)};
        });
    }

    // ---- drapes: thin sine sheets hanging from the gallery ----
    const int dn = seg(64, 8);
    for (int k = 0; k < 6; k++) {
        B.tm = (k & 1) ? -3 : -2;                                              // textures 2 / 1, per-hit lookup
        double x = X0 + 450 + 560.0 * k, z = (k < 3) ? -395.0 : 395.0, ph = 2 * PI * rng.uni();
        B.sheet(dn, dn, [&](double a, double b) {
            double fold = 22 * std::sin(10 * PI * a + ph) * (0.3 + 0.7 * b);
            double xa = (k < 3) ? 1.0 - a : a;                                 // normals face the nave
            return D3{x + 300 * xa, 600 - 430 * b, z + fold + 8 * std::sin(3 * PI * b)};
        }, 3, 3);
    }

    // ---- remainder: displaced relief on the far wall + debris ----
    B.tm = -1;
    size_t have = m.tris.size();
    if (have + 200 > triangles) throw std::runtime_error("atrium: triangle budget too small for the fixed parts");
    size_t rem = triangles - have;
    int rn = (int)std::floor(std::sqrt((double)(rem * 95 / 100) / 2.0));
    if (rn < 2) rn = 2;
    B.sheet(rn, rn, [&](double a, double b) {
        double bump = 25 * std::sin(14 * PI * a) * std::sin(10 * PI * b) + 12 * std::sin(40 * PI * a * b) + 6 * std::cos(55 * PI * (a - b));
        return D3{X1 - 40 - bump, 150 + 1200 * b, Z0 + 300 + (Z1 - Z0 - 600) * a};
    }, 8, 6);
    B.tm = -99999;
    int flat = B.n(0, 1, 0);
    while (m.tris.size() < triangles) {                                        // small free-floating triangles ("debris")
        double x = X0 + 100 + (X1 - X0 - 200) * rng.uni(), y = 20 + (Y1 - 300) * rng.uni(), z = Z0 + 100 + (Z1 - Z0 - 200) * rng.uni();
        double r = 4 + 10 * rng.uni();
        int a = B.v(x, y, z), b = B.v(x + r * (rng.uni() - 0.5), y + r * rng.uni(), z + r * (rng.uni() - 0.5)),
            c = B.v(x + r * (rng.uni() - 0.5), y + r * (rng.uni() - 0.5), z + r * rng.uni());
        B.tri(a, b, c, flat, flat, flat, -1, -1, -1);
    }
}

}  // namespace pth
