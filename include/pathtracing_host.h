/*
 * pathtracing_host.h -- C ABI of the host-side scene library (libpthost.so).
 *
 * Plain C++ on the CPU, no GPU dependency: the scene reader, the SAH/pull-up
 * kd-tree build, the multi-view camera list and image output that the
 * reference keeps on the host.  It produces the PtSceneDesc that
 * pt_scene_upload() of pathtracing_amd.h consumes.
 */
#ifndef PATHTRACING_HOST_H
#define PATHTRACING_HOST_H

#include "pathtracing_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PthScene PthScene;

/* Numbers printed by the reference after the build (main.cpp:204) plus shape. */
typedef struct PthTreeStats {
    int64_t triangles;        /* "Real Triangles"                       */
    int64_t naive;            /* `sum`,  util/kdtree.cpp:64             */
    int64_t optimized;        /* `sum2`, util/kdtree.cpp:65             */
    int64_t nodes, leaves, max_depth, max_leaf, root_entries;
    double build_seconds;
} PthTreeStats;

const char *pth_last_error(void);

/* init(scene) of util/reader.cpp:10-126: reads <dir>/master.txt and the mesh
   files it names.  Textures: for a texture T the reader loads the decoded
   binary PPM (P6) "<texture_dir>/T.ppm" when texture_dir is given, else
   "<dir>/T.ppm", else decodes "<dir>/T.jpg" when it holds a baseline JPEG or an
   8-bit PNG (the content decides, as with cv::imread). */
int pth_scene_load_dir(const char *dir, const char *texture_dir, PthScene **out);

/* The same scene as one self-contained little-endian file ("PTSCENE1"). */
int pth_scene_load_packed(const char *path, PthScene **out);
int pth_scene_save_packed(const PthScene *scene, const char *path);

/* buildTree(tree) of util/kdtree.cpp:262-281 (SAH sweep over triangle
   centroids + pull-up of triangles common to both children). */
int pth_scene_build_tree(PthScene *scene, PthTreeStats *stats);

/* Borrowed view of the flattened scene; valid until pth_scene_free().
   Requires pth_scene_build_tree(). */
int pth_scene_desc(const PthScene *scene, PtSceneDesc *out);

/* eye[3], step, astep, angle (rad), zoom as read from the C record. */
int pth_scene_camera(const PthScene *scene, double out7[7]);
void pth_scene_free(PthScene *scene);

/* View list of renderMultiple()/renderParallel() (mainOMP.cpp:225-235,164-165):
   N positions x 16 yaw angles, angle = -acos(-unit(a,0,c).x) + k*22.5 deg.
   Writes at most cap cameras, returns the total in *n. */
int pth_views_from_path(const char *path_file, double zoom, PtCamera *out, size_t cap, size_t *n);

/* Camera step of simulate() (main.cpp:160-187): key is one of "wasd" or
   'l' / 'r' for the LEFT / RIGHT arrows. */
int pth_camera_step(double eye[3], double *angle, double step, double astep, int key);

/* Texture decoding as the reader does it (cv::imread(file, IMREAD_COLOR) of util/reader.cpp:67-69): the file's
   content decides -- baseline JPEG, or 8-bit non-interlaced PNG with its alpha channel dropped -- and the result is
   BGR8.  Call with bgr == NULL to get the size; `capacity` is the size of `bgr` in bytes. */
int pth_decode_image(const char *path, int *rows, int *cols, uint8_t *bgr, size_t capacity);

/* Image output.  The reference shows frames with cv::imshow and notes "change to imwrite ... for en masse renderings"
   (mainOMP.cpp:242); its data sets under ml/data/ are such JPEG files.  pth_write_jpeg writes a baseline JPEG
   (YCbCr 4:2:0, standard tables; quality 1..100, 95 is what cv::imwrite uses by default). */
int pth_write_ppm(const char *path, const uint8_t *bgr, int rows, int cols);
int pth_write_bmp(const char *path, const uint8_t *bgr, int rows, int cols);
int pth_write_jpeg(const char *path, const uint8_t *bgr, int rows, int cols, int quality);

/* Deterministic procedural scene of a given triangle budget, standing in for
   meshes that are not available (writes meshes/lights/camera into a scene). */
int pth_scene_make_atrium(uint32_t triangles, uint32_t seed, PthScene **out);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACING_HOST_H */
