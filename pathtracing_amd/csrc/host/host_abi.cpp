// C ABI of the host library (see include/pathtracing_host.h).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <string>

#include "scene.hpp"

namespace pth {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace pth

using namespace pth;

extern "C" {

const char *pth_last_error(void) { return g_error.c_str(); }

int pth_scene_load_dir(const char *dir, const char *texture_dir, PthScene **out) {
    if (!dir || !out) { set_error("pth_scene_load_dir: null argument"); return PT_ERR_INVALID; }
    PthScene *s = new (std::nothrow) PthScene();
    if (!s) return PT_ERR_OOM;
    try {
        if (!read_scene_dir(dir, texture_dir ? texture_dir : "", s->scene)) { delete s; return PT_ERR_INVALID; }
    } catch (const std::exception &e) {
        set_error(std::string("pth_scene_load_dir: ") + e.what());
        delete s;
        return PT_ERR_INVALID;
    }
    *out = s;
    return PT_OK;
}

int pth_scene_load_packed(const char *path, PthScene **out) {
    if (!path || !out) { set_error("pth_scene_load_packed: null argument"); return PT_ERR_INVALID; }
    PthScene *s = new (std::nothrow) PthScene();
    if (!s) return PT_ERR_OOM;
    try {
        if (!load_packed(path, s->scene)) { delete s; return PT_ERR_INVALID; }
    } catch (const std::exception &e) {
        set_error(std::string("pth_scene_load_packed: ") + e.what());
        delete s;
        return PT_ERR_INVALID;
    }
    *out = s;
    return PT_OK;
}

int pth_scene_save_packed(const PthScene *scene, const char *path) {
    if (!scene || !path) { set_error("pth_scene_save_packed: null argument"); return PT_ERR_INVALID; }
    return save_packed(scene->scene, path) ? PT_OK : PT_ERR_INVALID;
}

int pth_scene_build_tree(PthScene *scene, PthTreeStats *stats) {
    if (!scene) { set_error("pth_scene_build_tree: null scene"); return PT_ERR_INVALID; }
    try {
        PthTreeStats st{};
        build_tree(scene->scene, scene->flat, st);
        if (stats) *stats = st;
    } catch (const std::exception &e) {
        set_error(std::string("pth_scene_build_tree: ") + e.what());
        return PT_ERR_OOM;
    }
    return PT_OK;
}

int pth_scene_desc(const PthScene *scene, PtSceneDesc *d) {
    if (!scene || !d) { set_error("pth_scene_desc: null argument"); return PT_ERR_INVALID; }
    const Flat &f = scene->flat;
    if (!f.built) { set_error("pth_scene_desc: call pth_scene_build_tree first"); return PT_ERR_INVALID; }
    std::memset(d, 0, sizeof *d);
    d->struct_size = (uint32_t)sizeof *d;
    d->n_mesh = (uint32_t)scene->scene.meshes.size();
    d->n_vert = (uint32_t)(f.vert.size() / 3);
    d->n_norm = (uint32_t)(f.norm.size() / 3);
    d->n_uv = (uint32_t)(f.uv.size() / 2);
    d->n_tri = (uint32_t)f.tri_mesh.size();
    d->n_tex = (uint32_t)f.tex.size();
    d->n_light = (uint32_t)f.lights.size();
    d->n_node = (uint32_t)f.nodes.size();
    d->n_ref = (uint32_t)f.refs.size();
    d->vert = f.vert.data(); d->norm = f.norm.data(); d->uv = f.uv.data();
    d->tri_vi = f.tri_vi.data(); d->tri_ni = f.tri_ni.data(); d->tri_ti = f.tri_ti.data();
    d->tri_tm = f.tri_tm.data(); d->tri_reft = f.tri_reft.data();
    d->tri_mesh = f.tri_mesh.data(); d->tri_local = f.tri_local.data();
    d->mesh_albedo = f.mesh_albedo.data();
    d->tex = f.tex.data(); d->lights = f.lights.data();
    for (int k = 0; k < 3; k++) d->background[k] = scene->scene.background[k];
    d->nodes = f.nodes.data(); d->refs = f.refs.data();
    return PT_OK;
}

int pth_scene_camera(const PthScene *scene, double out7[7]) {
    if (!scene || !out7) { set_error("pth_scene_camera: null argument"); return PT_ERR_INVALID; }
    const Scene &s = scene->scene;
    out7[0] = s.eye[0]; out7[1] = s.eye[1]; out7[2] = s.eye[2];
    out7[3] = s.step; out7[4] = s.astep; out7[5] = s.angle; out7[6] = s.zoom;
    return PT_OK;
}

void pth_scene_free(PthScene *scene) { delete scene; }

int pth_views_from_path(const char *path_file, double zoom, PtCamera *out, size_t cap, size_t *n) {
    if (!path_file || !n) { set_error("pth_views_from_path: null argument"); return PT_ERR_INVALID; }
    std::ifstream fin(path_file);
    if (!fin) { set_error(std::string("cannot open ") + path_file); return PT_ERR_INVALID; }
    // NUM_ANGLES / angles[], mainOMP.cpp:146-147
    static const double kAngles[16] = {0, 22.5, 45, 67.5, 90, 112.5, 135, 157.5,
                                       180, 202.5, 225, 247.5, 270, 292.5, 315, 337.5};
    long count = 0;
    if (!(fin >> count) || count < 0) { set_error(std::string(path_file) + ": missing position count"); return PT_ERR_INVALID; }
    size_t k = 0;
    for (long i = 0; i < count; i++) {
        double a, b, c;
        if (!(fin >> a >> b >> c)) { set_error(std::string(path_file) + ": fewer positions than announced"); return PT_ERR_INVALID; }
        for (int j = 0; j < 16; j++, k++) {
            if (!out || k >= cap) continue;
            double m = std::sqrt(a * a + 0.0 * 0.0 + c * c);   // unit(Vec3d(a, 0, c)).x, mainOMP.cpp:164
            double ang = -std::acos(-(a / m));
            ang += M_PI / 180.0 * kAngles[j];                   // mainOMP.cpp:165
            out[k].eye[0] = a; out[k].eye[1] = b; out[k].eye[2] = c;
            out[k].angle = ang;
            out[k].zoom = zoom;
        }
    }
    *n = k;
    return PT_OK;
}

int pth_camera_step(double eye[3], double *angle, double step, double astep, int key) {
    if (!eye || !angle) { set_error("pth_camera_step: null argument"); return PT_ERR_INVALID; }
    double a = *angle;
    switch (key) {                                              // main.cpp:162-187
        case 'w': eye[0] = eye[0] + std::cos(a) * step; eye[2] = eye[2] + std::sin(a) * step; break;
        case 'a': eye[0] = eye[0] - (-std::sin(a) * step); eye[2] = eye[2] - std::cos(a) * step; break;
        case 's': eye[0] = eye[0] - std::cos(a) * step; eye[2] = eye[2] - std::sin(a) * step; break;
        case 'd': eye[0] = eye[0] + (-std::sin(a) * step); eye[2] = eye[2] + std::cos(a) * step; break;
        case 'l': *angle = a - astep; break;
        case 'r': *angle = a + astep; break;
        default: set_error("pth_camera_step: unknown key"); return PT_ERR_INVALID;
    }
    return PT_OK;
}

int pth_decode_image(const char *path, int *rows, int *cols, uint8_t *bgr, size_t capacity) {
    if (!path || !rows || !cols) { set_error("pth_decode_image: bad argument"); return PT_ERR_INVALID; }
    pth::Texture t;
    if (!pth::decode_image_file(path, t)) {
        set_error(std::string("pth_decode_image: ") + path + " is missing or neither a baseline JPEG nor an 8-bit non-interlaced PNG");
        return PT_ERR_INVALID;
    }
    *rows = t.rows; *cols = t.cols;
    if (bgr) {
        if (capacity < t.bgr.size()) { set_error("pth_decode_image: buffer too small"); return PT_ERR_INVALID; }
        std::memcpy(bgr, t.bgr.data(), t.bgr.size());
    }
    return PT_OK;
}

int pth_write_ppm(const char *path, const uint8_t *bgr, int rows, int cols) {
    if (!path || !bgr || rows <= 0 || cols <= 0) { set_error("pth_write_ppm: bad argument"); return PT_ERR_INVALID; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return PT_ERR_INVALID; }
    std::fprintf(f, "P6\n%d %d\n255\n", cols, rows);
    std::string row((size_t)cols * 3, '\0');
    for (int i = 0; i < rows; i++) {
        for (int j = 0; j < cols; j++) {
            const uint8_t *p = bgr + ((size_t)i * cols + j) * 3;
            row[3 * j] = (char)p[2]; row[3 * j + 1] = (char)p[1]; row[3 * j + 2] = (char)p[0];
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    return std::fclose(f) == 0 ? PT_OK : PT_ERR_INVALID;
}

int pth_write_bmp(const char *path, const uint8_t *bgr, int rows, int cols) {
    if (!path || !bgr || rows <= 0 || cols <= 0) { set_error("pth_write_bmp: bad argument"); return PT_ERR_INVALID; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return PT_ERR_INVALID; }
    const uint32_t stride = ((uint32_t)cols * 3 + 3) & ~3u, data = stride * (uint32_t)rows, off = 54;
    uint8_t h[54] = {'B', 'M'};
    auto put32 = [&](int at, uint32_t v) { for (int k = 0; k < 4; k++) h[at + k] = (uint8_t)(v >> (8 * k)); };
    put32(2, off + data); put32(10, off); put32(14, 40); put32(18, (uint32_t)cols); put32(22, (uint32_t)rows);
    h[26] = 1; h[28] = 24; put32(34, data);
    std::fwrite(h, 1, sizeof h, f);
    std::string row(stride, '\0');
    for (int i = rows - 1; i >= 0; i--) {                        // BMP rows run bottom-up, pixels are BGR
        std::memcpy(&row[0], bgr + (size_t)i * cols * 3, (size_t)cols * 3);
        std::fwrite(row.data(), 1, row.size(), f);
    }
    return std::fclose(f) == 0 ? PT_OK : PT_ERR_INVALID;
}

int pth_write_jpeg(const char *path, const uint8_t *bgr, int rows, int cols, int quality) {
    if (!path || !bgr || rows <= 0 || cols <= 0 || rows > 65535 || cols > 65535) { set_error("pth_write_jpeg: bad argument"); return PT_ERR_INVALID; }
    try {
        if (!pth::write_jpeg_file(path, bgr, rows, cols, quality)) { set_error(std::string("cannot write ") + path); return PT_ERR_INVALID; }
    } catch (const std::bad_alloc &) {
        return PT_ERR_OOM;
    }
    return PT_OK;
}

int pth_scene_make_atrium(uint32_t triangles, uint32_t seed, PthScene **out) {
    if (!out) { set_error("pth_scene_make_atrium: null argument"); return PT_ERR_INVALID; }
    PthScene *s = new (std::nothrow) PthScene();
    if (!s) return PT_ERR_OOM;
    try {
        make_atrium(triangles, seed, s->scene);
    } catch (const std::exception &e) {
        set_error(std::string("pth_scene_make_atrium: ") + e.what());
        delete s;
        return PT_ERR_INVALID;
    }
    *out = s;
    return PT_OK;
}

}  // extern "C"
