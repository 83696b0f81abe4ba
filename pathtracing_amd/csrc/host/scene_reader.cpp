// Scene input: the text formats of the reference (master.txt + OBJ-like mesh
// files, util/reader.cpp:10-126) and the packed single-file form "PTSCENE1".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <unordered_map>

#include "scene.hpp"

namespace pth {

namespace {

// Tokens of a line split at runs of `sep`, with the edge behaviour of
// boost::split(..., token_compress_on) that util/reader.cpp:17,54,98 relies on:
// a separator run counts once, a leading or trailing separator yields an empty
// first or last token, and an empty line yields one empty token.
std::vector<std::string> split_runs(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    bool in_sep = false;
    for (char c : s) {
        if (c == sep) {
            if (!in_sep) { out.push_back(cur); cur.clear(); in_sep = true; }
        } else {
            cur.push_back(c);
            in_sep = false;
        }
    }
    out.push_back(cur);
    return out;
}

bool file_exists(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    return f.good();
}

double num(const std::vector<std::string> &t, size_t i) {
    return i < t.size() ? std::strtod(t[i].c_str(), nullptr) : 0.0;
}

// One texture, looked up as the header documents: decoded PPM first, JPEG last.
bool load_texture(const std::string &dir, const std::string &texture_dir, const std::string &name, Texture &out) {
    if (!texture_dir.empty() && load_ppm(texture_dir + "/" + name + ".ppm", out)) return true;
    if (load_ppm(dir + "/" + name + ".ppm", out)) return true;
    std::string jpg = dir + "/" + name + ".jpg";
    if (file_exists(jpg) && decode_image_file(jpg, out)) return true;
    set_error("texture '" + name + "': no decoded " + name + ".ppm found and " + jpg + " could not be decoded");
    return false;
}

}  // namespace

bool load_ppm(const std::string &path, Texture &out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    int w = 0, h = 0, maxv = 0;
    char magic[3] = {0, 0, 0};
    bool ok = std::fscanf(f, "%2s", magic) == 1 && std::strcmp(magic, "P6") == 0;
    // header fields may be separated by whitespace and '#' comments
    auto next_int = [&](int &v) {
        int c;
        for (;;) {
            c = std::fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
            break;
        }
        if (c == EOF) return false;
        std::ungetc(c, f);
        return std::fscanf(f, "%d", &v) == 1;
    };
    ok = ok && next_int(w) && next_int(h) && next_int(maxv) && maxv == 255 && w > 0 && h > 0;
    if (ok) {
        std::fgetc(f);  // the single whitespace byte after maxval
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
        if (ok) {
            out.rows = h; out.cols = w;
            out.bgr.resize(rgb.size());
            for (size_t i = 0; i < (size_t)w * h; i++) {   // cv::imread stores BGR
                out.bgr[3 * i] = rgb[3 * i + 2];
                out.bgr[3 * i + 1] = rgb[3 * i + 1];
                out.bgr[3 * i + 2] = rgb[3 * i];
            }
        }
    }
    std::fclose(f);
    return ok;
}

// Index sanity of one mesh, for both readers: the kd build, the flattening and the device code index these pools
// without checks.
static bool check_mesh_indices(const Mesh &mesh, const std::string &what) {
    for (const Triangle &tr : mesh.tris) {
        for (int k = 0; k < 3; k++) {
            if (tr.vi[k] < 0 || tr.vi[k] >= (int)mesh.vert.size() || tr.ni[k] < 0 || tr.ni[k] >= (int)mesh.norm.size() ||
                tr.ti[k] < -1 || tr.ti[k] >= (int)mesh.uv.size()) {
                set_error(what + ": face index out of range");
                return false;
            }
            if (tr.tm != -99999 && tr.tm < 0 && tr.ti[k] < 0) {
                set_error(what + ": per-hit textured face without texture coordinates");
                return false;
            }
        }
        if (tr.tm != -99999) {
            const long long k = tr.tm >= 0 ? (long long)tr.tm : -((long long)tr.tm + 1);
            if (k >= (long long)mesh.tmaps.size()) {
                set_error(what + ": texture index out of range");
                return false;
            }
        }
    }
    return true;
}

bool read_scene_dir(const std::string &dir, const std::string &texture_dir, Scene &sc) {
    std::ifstream master(dir + "/master.txt");
    if (!master) { set_error("cannot open " + dir + "/master.txt"); return false; }
    std::vector<std::string> files;
    std::unordered_map<std::string, std::string> tarr;   // group name -> texture base name
    std::string line;
    sc = Scene();
    while (std::getline(master, line)) {                  // util/reader.cpp:15-43
        std::vector<std::string> d = split_runs(line, ' ');
        char type = d[0].empty() ? '\0' : d[0][0];
        if (type == '#') continue;
        if (type == '%') break;
        if (type == 'C' && d.size() >= 8) {
            const double ud = 0.01745329251;
            sc.eye = {num(d, 1), num(d, 2), num(d, 3)};
            sc.step = num(d, 4);
            sc.astep = ud * num(d, 5);
            sc.angle = ud * num(d, 6);
            sc.zoom = num(d, 7);
        } else if (type == 'T' && d.size() >= 3) {
            tarr[d[1]] = d[2];
        } else if (type == 'L' && d.size() >= 9) {
            Light L;
            L.type = d[1].empty() ? 0 : d[1][0];
            L.mag = (double)(float)num(d, 8);             // Light's ctor takes mag as float, util/lighting.cpp:14
            L.pos = {num(d, 2), num(d, 3), num(d, 4)};
            L.color = {num(d, 5), num(d, 6), num(d, 7)};
            sc.lights.push_back(L);
        } else if (type == 'O' && d.size() >= 2) {
            files.push_back(d[1]);
        }
    }
    for (const std::string &fn : files) {                 // util/reader.cpp:45-125
        std::ifstream in(dir + "/" + fn);
        if (!in) { set_error("cannot open mesh file " + dir + "/" + fn); return false; }
        Mesh mesh;
        int tcnt = 0, reft = 0;
        int tval = -99999;   // the reference leaves this unset until the first 'g' record
        bool stop = false;
        while (!stop && std::getline(in, line)) {
            std::vector<std::string> t = split_runs(line, ' ');
            if (t[0].empty()) continue;                   // blank line: no record type
            switch (t[0].back()) {                        // record type = LAST char of the first token, :55
                case '%': stop = true; break;
                case 'g': {
                    reft = 0;
                    tval = -99999;
                    if (t.size() == 1) break;
                    auto it = tarr.find(t[1]);
                    if (it != tarr.end()) {               // every textured group loads its image again, :66-69
                        Texture tx;
                        if (!load_texture(dir, texture_dir, it->second, tx)) return false;
                        mesh.tmaps.push_back(std::move(tx));
                        tval = tcnt++;
                        if (t.size() > 2 && !t[2].empty() && t[2][0] == '?') tval = -tval - 1;
                    }
                    if (t.size() > 2 && !t[2].empty() && t[2][0] == '1') reft = 1;
                    break;
                }
                case 'v': mesh.vert.push_back({num(t, 1), num(t, 2), num(t, 3)}); break;
                case 'n': mesh.norm.push_back({num(t, 1), num(t, 2), num(t, 3)}); break;
                case 't': {
                    auto clamp = [](double v) { return std::max(0.0, std::min(0.99, v)); };
                    mesh.uv.push_back({clamp(num(t, 1)), clamp(num(t, 2))});
                    break;
                }
                case 'f': {
                    std::vector<int> data;
                    int nfld = 0;
                    int sz = (int)t.size() - 1;
                    for (int j = 1; j <= sz; j++) {
                        std::vector<std::string> spc = split_runs(t[j], '/');
                        for (const std::string &s : spc) data.push_back(std::atoi(s.c_str()));
                        nfld = (int)spc.size();
                    }
                    auto at = [&](int k) { return (k >= 0 && k < (int)data.size()) ? data[k] - 1 : -1; };
                    for (int j = 1; j < sz - 1; j++) {    // fan around the first vertex, :104-114
                        Triangle tri;
                        tri.vi[0] = at(0); tri.vi[1] = at(j * nfld); tri.vi[2] = at((j + 1) * nfld);
                        for (int k = 0; k < 3; k++) tri.ni[k] = tri.ti[k] = -1;
                        tri.tm = tval; tri.reft = reft;
                        if (nfld > 2) {
                            tri.ti[0] = at(1); tri.ti[1] = at(j * nfld + 1); tri.ti[2] = at((j + 1) * nfld + 1);
                            tri.ni[0] = at(2); tri.ni[1] = at(j * nfld + 2); tri.ni[2] = at((j + 1) * nfld + 2);
                        } else {
                            tri.ni[0] = at(1); tri.ni[1] = at(j * nfld + 1); tri.ni[2] = at((j + 1) * nfld + 1);
                        }
                        mesh.tris.push_back(tri);
                    }
                    break;
                }
                default: break;
            }
        }
        if (!check_mesh_indices(mesh, "mesh file " + fn)) return false;
        sc.meshes.push_back(std::move(mesh));
    }
    return true;
}

// ---- packed single-file form ------------------------------------------------
namespace {
struct Writer {
    FILE *f;
    bool ok = true;
    void put(const void *p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    void u32(uint32_t v) { put(&v, 4); }
};
struct Reader {
    FILE *f;
    bool ok = true;
    void get(void *p, size_t n) { if (ok && n && std::fread(p, 1, n, f) != n) ok = false; }
    uint32_t u32() { uint32_t v = 0; get(&v, 4); return v; }
};
}  // namespace

bool save_packed(const Scene &sc, const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) { set_error("cannot write " + path); return false; }
    Writer w{f};
    w.put("PTSCENE1", 8);
    double cam[7] = {sc.eye[0], sc.eye[1], sc.eye[2], sc.step, sc.astep, sc.angle, sc.zoom};
    w.put(cam, sizeof cam);
    w.put(sc.background.data(), 24);
    w.u32((uint32_t)sc.lights.size());
    for (const Light &L : sc.lights) {
        int32_t ty = L.type;
        double d[7] = {L.mag, L.pos[0], L.pos[1], L.pos[2], L.color[0], L.color[1], L.color[2]};
        w.put(&ty, 4);
        w.put(d, sizeof d);
    }
    w.u32((uint32_t)sc.meshes.size());
    for (const Mesh &m : sc.meshes) {
        w.put(m.albedo.data(), 24);
        w.u32((uint32_t)m.vert.size()); w.u32((uint32_t)m.norm.size()); w.u32((uint32_t)m.uv.size());
        w.u32((uint32_t)m.tris.size()); w.u32((uint32_t)m.tmaps.size());
        w.put(m.vert.data(), m.vert.size() * 24);
        w.put(m.norm.data(), m.norm.size() * 24);
        w.put(m.uv.data(), m.uv.size() * 16);
        w.put(m.tris.data(), m.tris.size() * sizeof(Triangle));
        for (const Texture &t : m.tmaps) {
            w.u32((uint32_t)t.rows); w.u32((uint32_t)t.cols);
            w.put(t.bgr.data(), t.bgr.size());
        }
    }
    bool ok = w.ok;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) set_error("short write to " + path);
    return ok;
}

bool load_packed(const std::string &path, Scene &sc) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { set_error("cannot open " + path); return false; }
    Reader r{f};
    std::fseek(f, 0, SEEK_END);
    const long long file_bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    char magic[8];
    r.get(magic, 8);
    if (!r.ok || std::memcmp(magic, "PTSCENE1", 8) != 0) {
        std::fclose(f);
        set_error(path + ": not a PTSCENE1 file");
        return false;
    }
    sc = Scene();
    double cam[7];
    r.get(cam, sizeof cam);
    sc.eye = {cam[0], cam[1], cam[2]};
    sc.step = cam[3]; sc.astep = cam[4]; sc.angle = cam[5]; sc.zoom = cam[6];
    r.get(sc.background.data(), 24);
    uint32_t nl = r.u32();
    for (uint32_t i = 0; r.ok && i < nl; i++) {
        Light L;
        double d[7];
        r.get(&L.type, 4);
        r.get(d, sizeof d);
        L.mag = d[0]; L.pos = {d[1], d[2], d[3]}; L.color = {d[4], d[5], d[6]};
        sc.lights.push_back(L);
    }
    uint32_t nm = r.u32();
    for (uint32_t i = 0; r.ok && i < nm; i++) {
        Mesh m;
        r.get(m.albedo.data(), 24);
        uint32_t nv = r.u32(), nn = r.u32(), nt = r.u32(), ntri = r.u32(), ntex = r.u32();
        if (!r.ok) break;
        // counts that the file cannot hold are damage, not sizes to allocate
        if ((long long)nv * 24 + (long long)nn * 24 + (long long)nt * 16 + (long long)ntri * (long long)sizeof(Triangle) +
                (long long)ntex * 8 > file_bytes) { r.ok = false; break; }
        m.vert.resize(nv); m.norm.resize(nn); m.uv.resize(nt); m.tris.resize(ntri);
        r.get(m.vert.data(), (size_t)nv * 24);
        r.get(m.norm.data(), (size_t)nn * 24);
        r.get(m.uv.data(), (size_t)nt * 16);
        r.get(m.tris.data(), (size_t)ntri * sizeof(Triangle));
        for (uint32_t k = 0; r.ok && k < ntex; k++) {
            Texture t;
            t.rows = (int32_t)r.u32(); t.cols = (int32_t)r.u32();
            if (!r.ok || t.rows <= 0 || t.cols <= 0 || (long long)t.rows * t.cols * 3 > file_bytes) { r.ok = false; break; }
            t.bgr.resize((size_t)t.rows * t.cols * 3);
            r.get(t.bgr.data(), t.bgr.size());
            m.tmaps.push_back(std::move(t));
        }
        sc.meshes.push_back(std::move(m));
    }
    std::fclose(f);
    if (!r.ok) { set_error(path + ": truncated or corrupt PTSCENE1 file"); return false; }
    for (size_t i = 0; i < sc.meshes.size(); i++)
        if (!check_mesh_indices(sc.meshes[i], path + ": mesh " + std::to_string(i))) return false;
    return true;
}

}  // namespace pth
