// ptmain -- command-line front end with the reference's surface:
//
//     ptmain [scene] [-gi] [-rb] [-rm]
//
// (main.cpp:191-206, mainOMP.cpp:328-354).  Scene input is scenes/<scene>/ in the
// reference's text format, or scenes/<scene>.ptscene when only the packed file
// exists.  The host side (reader, kd build, view list) runs here; rendering goes
// through the C ABI of libptamd.so.  Frames are written as image files instead
// of being shown in a window.
//
// Extensions, all optional: --model main|omp (default: main, or omp with -rm as
// that flag belongs to mainOMP.cpp), --spp N, --exact-gi (fp64 GI), --out DIR,
// --size N, --views N (limit of -rm), --device K, --gpus all (with -rm: every
// visible GPU, the views dealt round-robin to them by pt_render_views -- the
// `mpirun -n` of mainOMP.cpp:344-353 in one process), --keys "wasdlr..." (camera
// steps of simulate(), one frame per key), --stdin (the same steps read from
// standard input, one command per line: a key w a s d l r, "C x y z angle_deg"
// to place the camera, q to quit; one frame per command -- simulate() of
// main.cpp:152-189 without the window), --dataset DIR (with -rm: per view the
// three images the reference's denoiser is trained on, DIR/<SCENE>/<SCENE>4ssp,
// <SCENE>4096ssp and <SCENE>dif, named <scene><view>.ppm: 4 spp, 4096 spp and the
// pass without -gi of mainOMP.cpp:263-270; --noisy-spp / --converged-spp change
// the two counts), --format ppm|jpg (frame files; jpg is what the reference's data
// sets are, written with cv::imwrite's default quality 95), --resume (with -rm: frame
// files that are already there, whole, are kept and their views not rendered again;
// the reference restarts at view 0 and only shows its frames, mainOMP.cpp:242).
#include <cctype>
#include <iostream>
#include <sstream>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pathtracing_amd.h"
#include "pathtracing_host.h"

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

static int die(const char *what, const char *msg) {
    std::fprintf(stderr, "ptmain: %s: %s\n", what, msg);
    return 1;
}

int main(int argc, char **argv) {
    bool gi = false, rb = false, rm = false, exact_gi = false, from_stdin = false, all_gpus = false, resume = false;
    std::string scene, model_s, out = ".", keys, dataset, format = "ppm";
    int spp = 0, size = 0, max_views = -1, dev = 0, noisy_spp = 4, converged_spp = 4096;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "ptmain: %s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (a == "-gi") gi = true;
        else if (a == "-rb") rb = true;
        else if (a == "-rm") rm = true;
        else if (a == "--resume") resume = true;
        else if (a == "-pp") { std::fprintf(stderr, "ptmain: -pp (Keras post-processing) is not part of this build\n"); return 2; }
        else if (a == "--model") model_s = val("--model");
        else if (a == "--spp") spp = std::atoi(val("--spp"));
        else if (a == "--size") size = std::atoi(val("--size"));
        else if (a == "--views") max_views = std::atoi(val("--views"));
        else if (a == "--device") dev = std::atoi(val("--device"));
        else if (a == "--gpus") {
            const std::string g = val("--gpus");
            if (g == "all") all_gpus = true;
            else if (g != "1") { std::fprintf(stderr, "ptmain: --gpus takes 1 or all (choose among GPUs with HIP_VISIBLE_DEVICES)\n"); return 2; }
        }
        else if (a == "--out") out = val("--out");
        else if (a == "--keys") keys = val("--keys");
        else if (a == "--exact-gi") exact_gi = true;
        else if (a == "--stdin") from_stdin = true;
        else if (a == "--dataset") dataset = val("--dataset");
        else if (a == "--format") {
            format = val("--format");
            if (format != "ppm" && format != "jpg") { std::fprintf(stderr, "ptmain: --format must be ppm or jpg\n"); return 2; }
        }
        else if (a == "--noisy-spp") noisy_spp = std::atoi(val("--noisy-spp"));
        else if (a == "--converged-spp") converged_spp = std::atoi(val("--converged-spp"));
        else if (!a.empty() && a[0] != '-') scene = a;
        else { std::fprintf(stderr, "ptmain: unknown option %s\n", a.c_str()); return 2; }
    }
    if (scene.empty()) { std::fprintf(stderr, "usage: ptmain [scene] [-gi] [-rb] [-rm]\n"); return 2; }
    int model = rm ? PT_MODEL_OMP : PT_MODEL_MAIN;
    if (model_s == "main") model = PT_MODEL_MAIN;
    else if (model_s == "omp") model = PT_MODEL_OMP;
    else if (!model_s.empty()) { std::fprintf(stderr, "ptmain: --model must be main or omp\n"); return 2; }

    PthScene *hs = nullptr;
    std::string dir = "scenes/" + scene, packed = dir + ".ptscene";
    int rc = exists(dir + "/master.txt") ? pth_scene_load_dir(dir.c_str(), nullptr, &hs) : pth_scene_load_packed(packed.c_str(), &hs);
    if (rc) return die("scene", pth_last_error());
    PthTreeStats ts;
    if (pth_scene_build_tree(hs, &ts)) return die("kd build", pth_last_error());
    std::printf("Real Triangles: %lld; Naive: %lld; Optimized: %lld\n", (long long)ts.triangles, (long long)ts.naive,
                (long long)ts.optimized);
    PtSceneDesc desc;
    if (pth_scene_desc(hs, &desc)) return die("scene", pth_last_error());
    if (pt_init(all_gpus ? -1 : dev)) return die("pt_init", pt_last_error());
    if (rm) std::printf("World Size: %d\n", pt_device_count());              // mainOMP.cpp:349
    PtScene *ds = nullptr;
    if (pt_scene_upload(&desc, &ds)) return die("pt_scene_upload", pt_last_error());

    double cam7[7];
    pth_scene_camera(hs, cam7);
    PtOptions opt{};
    opt.struct_size = sizeof opt;
    opt.gi = gi; opt.rb = rb; opt.model = model; opt.spp = spp; opt.size = size;
    opt.gi_precision = exact_gi ? PT_GI_F64_EXACT : PT_GI_F32_FAST;
    opt.seed = 0x5EED;
    const int n = size ? size : 512;
    std::vector<uint8_t> frame((size_t)n * n * 3);

    const std::string ext = "." + format;
    auto write_frame = [&](const std::string &file, const uint8_t *bgr) -> int {
        return format == "jpg" ? pth_write_jpeg(file.c_str(), bgr, n, n, 95) : pth_write_ppm(file.c_str(), bgr, n, n);
    };
    // --resume: is this frame file already there, whole?  A PPM of this size has one length; a JPEG ends with its end marker.
    auto frame_file_complete = [&](const std::string &file) -> bool {
        FILE *f = std::fopen(file.c_str(), "rb");
        if (!f) return false;
        bool ok = false;
        if (std::fseek(f, 0, SEEK_END) == 0) {
            const long len = std::ftell(f);
            if (format == "jpg") {
                unsigned char tail[2] = {0, 0};
                ok = len > 4 && std::fseek(f, -2, SEEK_END) == 0 && std::fread(tail, 1, 2, f) == 2 && tail[0] == 0xFF && tail[1] == 0xD9;
            } else {
                char hdr[64];
                ok = len == (long)std::snprintf(hdr, sizeof hdr, "P6\n%d %d\n255\n", n, n) + (long)frame.size();
            }
        }
        std::fclose(f);
        return ok;
    };
    auto render_one = [&](const PtCamera &cam, uint32_t view, const std::string &file) -> int {
        if (resume && frame_file_complete(file)) { std::printf("  kept %s\n", file.c_str()); return 0; }
        opt.view = view;
        PtStats st{};
        auto t0 = std::chrono::high_resolution_clock::now();
        if (pt_render_view(ds, &cam, &opt, frame.data(), nullptr, &st)) return die("pt_render_view", pt_last_error());
        std::chrono::duration<double> el = std::chrono::high_resolution_clock::now() - t0;
        std::printf("Elapsed time: %g s\n", el.count());      // main.cpp:158, mainOMP.cpp:241
        unsigned long long rays = st.primary + st.shadow + st.mirror + st.gi;
        std::printf("  rays: primary %llu shadow %llu mirror %llu gi %llu  (%.1f Mrays/s) -> %s\n",
                    (unsigned long long)st.primary, (unsigned long long)st.shadow, (unsigned long long)st.mirror,
                    (unsigned long long)st.gi, rays / el.count() * 1e-6, file.c_str());
        if (write_frame(file, frame.data())) return die("write", pth_last_error());
        return 0;
    };

    if (rm) {                                                   // renderMultiple, mainOMP.cpp:225-247
        std::string pf = "scenes/paths/" + scene + "path.txt";
        size_t total = 0;
        if (pth_views_from_path(pf.c_str(), cam7[6], nullptr, 0, &total)) return die("path file", pth_last_error());
        std::vector<PtCamera> cams(total);
        if (pth_views_from_path(pf.c_str(), cam7[6], cams.data(), total, &total)) return die("path file", pth_last_error());
        if (max_views >= 0 && (size_t)max_views < total) total = (size_t)max_views;
        std::string upper = scene;
        for (char &c : upper) c = (char)std::toupper((unsigned char)c);
        if (!dataset.empty()) {
            if (noisy_spp < 1 || converged_spp < 1) { std::fprintf(stderr, "ptmain: sample counts must be positive\n"); return 2; }
            mkdir(dataset.c_str(), 0777);
            mkdir((dataset + "/" + upper).c_str(), 0777);
            for (const char *sub : {"4ssp", "4096ssp", "dif"}) mkdir((dataset + "/" + upper + "/" + upper + sub).c_str(), 0777);
        }
        if (dataset.empty() && all_gpus) {
            // the view loop inside the library: batches of views, sharded over the GPUs, frames written between batches
            const size_t batch = 32 * (size_t)pt_device_count();
            std::vector<uint8_t> frames[2] = {std::vector<uint8_t>(batch * frame.size()), std::vector<uint8_t>(batch * frame.size())};
            std::thread writer;                                 // writes the files of one batch while the next one renders
            int write_failed = 0;
            PtStats st{};
            auto t0 = std::chrono::high_resolution_clock::now();
            size_t turn = 0, start = 0;
            if (resume) {                                       // the job goes on at the first view whose file is not whole
                auto file_of = [&](size_t v) { char name[64]; std::snprintf(name, sizeof name, "/%s%zu", scene.c_str(), v); return out + name + ext; };
                while (start < total && frame_file_complete(file_of(start))) start++;
                std::printf("Resuming at view %zu of %zu\n", start, total);
            }
            for (size_t v0 = start; v0 < total; v0 += batch, turn ^= 1) {
                const size_t nb = total - v0 < batch ? total - v0 : batch;
                auto tb = std::chrono::high_resolution_clock::now();
                uint8_t *buf = frames[turn].data();
                if (pt_render_views(ds, cams.data() + v0, nb, (uint32_t)v0, 1, &opt, buf, &st)) {
                    if (writer.joinable()) writer.join();
                    return die("pt_render_views", pt_last_error());
                }
                std::chrono::duration<double> el = std::chrono::high_resolution_clock::now() - tb;
                for (size_t k = 0; k < nb; k++) std::printf("Elapsed time: %g s\n", el.count() / (double)nb);
                if (writer.joinable()) writer.join();          // the other buffer is free again after this
                writer = std::thread([&, buf, v0, nb] {
                    for (size_t k = 0; k < nb; k++) {
                        char name[64];
                        std::snprintf(name, sizeof name, "/%s%zu", scene.c_str(), v0 + k);
                        if (write_frame(out + name + ext, buf + k * frame.size())) write_failed = 1;
                    }
                });
            }
            if (writer.joinable()) writer.join();
            if (write_failed) return die("write", "a frame file could not be written");
            std::chrono::duration<double> el = std::chrono::high_resolution_clock::now() - t0;
            const unsigned long long rays = st.primary + st.shadow + st.mirror + st.gi;
            std::printf("%zu views on %d GPU(s) in %.3f s: %.2f views/s, %.1f Mrays/s\n", total - start, pt_device_count(), el.count(),
                        (double)(total - start) / el.count(), (double)rays / el.count() * 1e-6);
            total = 0;                                          // nothing left for the loop below
        }
        for (size_t v = 0; v < total; v++) {
            char name[64];
            std::snprintf(name, sizeof name, "/%s%zu%s", scene.c_str(), v, ext.c_str());
            if (dataset.empty()) {
                if (render_one(cams[v], (uint32_t)v, out + name)) return 1;
                continue;
            }
            const std::string base = dataset + "/" + upper + "/" + upper;
            opt.gi = 1; opt.spp = noisy_spp;
            if (render_one(cams[v], (uint32_t)v, base + "4ssp" + name)) return 1;
            opt.spp = converged_spp;
            if (render_one(cams[v], (uint32_t)v, base + "4096ssp" + name)) return 1;
            opt.gi = 0; opt.spp = 0;
            if (render_one(cams[v], (uint32_t)v, base + "dif" + name)) return 1;
        }
    } else {                                                    // simulate(), main.cpp:152-189
        PtCamera cam{{cam7[0], cam7[1], cam7[2]}, cam7[5], cam7[6]};
        if (render_one(cam, 0, out + "/" + scene + ext)) return 1;
        for (size_t k = 0; k < keys.size(); k++) {
            if (pth_camera_step(cam.eye, &cam.angle, cam7[3], cam7[4], keys[k])) return die("--keys", pth_last_error());
            char name[64];
            std::snprintf(name, sizeof name, "/%s_step%zu%s", scene.c_str(), k + 1, ext.c_str());
            if (render_one(cam, 0, out + name)) return 1;
        }
        if (from_stdin) {
            std::string line;
            for (size_t k = 1; std::getline(std::cin, line); k++) {
                std::istringstream in(line);
                std::string cmd;
                if (!(in >> cmd)) { k--; continue; }
                if (cmd == "q") break;
                if (cmd == "C") {
                    double x, y, z, deg;
                    if (!(in >> x >> y >> z >> deg)) return die("--stdin", "C needs x y z angle_deg");
                    cam.eye[0] = x; cam.eye[1] = y; cam.eye[2] = z; cam.angle = deg * 0.01745329251;   // util/reader.cpp:24-29
                } else if (cmd.size() == 1 && pth_camera_step(cam.eye, &cam.angle, cam7[3], cam7[4], cmd[0]) == 0) {
                } else {
                    return die("--stdin", ("unknown command '" + line + "'").c_str());
                }
                char name[64];
                std::snprintf(name, sizeof name, "/%s_frame%zu%s", scene.c_str(), k, ext.c_str());
                if (render_one(cam, 0, out + name)) return 1;
                std::fflush(stdout);
            }
        }
    }
    pt_scene_free(ds);
    pth_scene_free(hs);
    pt_shutdown();
    return 0;
}
