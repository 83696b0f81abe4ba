// Host-side scene model shared by the reader, the kd-tree build and the
// flattening step.  Plain data; no GPU types.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "pathtracing_host.h"

namespace pth {

using D3 = std::array<double, 3>;

struct Triangle {            // util/polygonMesh.cpp:25-31
    int32_t vi[3], ni[3], ti[3];
    int32_t tm, reft;
};
static_assert(sizeof(Triangle) == 44, "packed-file layout");

struct Texture {             // cv::Mat of type 8UC3
    int32_t rows = 0, cols = 0;
    std::vector<uint8_t> bgr;
};

struct Mesh {                // util/polygonMesh.cpp:37-46
    D3 albedo{1, 1, 1};
    std::vector<D3> vert, norm;
    std::vector<std::array<double, 2>> uv;
    std::vector<Triangle> tris;
    std::vector<Texture> tmaps;
};

struct Light { int32_t type; double mag; D3 pos, color; };

struct Scene {
    D3 eye{0, 0, 0};
    double step = 0, astep = 0, angle = 0, zoom = 0;
    D3 background{1, 1, 1};
    std::vector<Light> lights;
    std::vector<Mesh> meshes;
};

// Flattened scene + tree, the backing store of a PtSceneDesc.
struct Flat {
    std::vector<double> vert, norm, uv, mesh_albedo;
    std::vector<int32_t> tri_vi, tri_ni, tri_ti, tri_tm, tri_reft, tri_mesh, tri_local;
    std::vector<PtTexture> tex;
    std::vector<PtLight> lights;
    std::vector<PtKdNode> nodes;
    std::vector<int32_t> refs;
    std::vector<int32_t> mesh_tri_base;   // first global triangle of each mesh
    bool built = false;
};

void set_error(const std::string &msg);

// scene_reader.cpp
bool read_scene_dir(const std::string &dir, const std::string &texture_dir, Scene &out);
bool load_packed(const std::string &path, Scene &out);
bool save_packed(const Scene &scene, const std::string &path);
bool load_ppm(const std::string &path, Texture &out);
bool decode_jpeg_file(const std::string &path, Texture &out);   // jpeg_decode.cpp
bool decode_png_file(const std::string &path, Texture &out);    // png_decode.cpp
bool decode_image_file(const std::string &path, Texture &out);  // by content: JPEG or PNG, whatever the file is called
bool write_jpeg_file(const std::string &path, const uint8_t *bgr, int rows, int cols, int quality);   // jpeg_encode.cpp

// kd_build.cpp
void flatten_geometry(const Scene &scene, Flat &flat);
void build_tree(const Scene &scene, Flat &flat, PthTreeStats &stats);

// atrium.cpp
void make_atrium(uint32_t triangles, uint32_t seed, Scene &out);

}  // namespace pth

struct PthScene {
    pth::Scene scene;
    pth::Flat flat;
};
