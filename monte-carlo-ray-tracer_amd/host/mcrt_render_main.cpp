// mcrt_render - headless driver: what source/main.cpp does without the window.
//
//   main.cpp:234-242   build the default scene, camera, PrepareForRayTracing
//   main.cpp:100-184   render loop            -> Scene::Render (GPU), progressive passes
//   main.cpp:142-164   tonemap + 8-bit store  -> Scene::Resolve (GPU)
//   screenshot.cpp     PNG of the framebuffer -> binary PPM (P6) of the BGRA8 buffer
//
// Usage: mcrt_render [--scene cornell|hexagon] [--width W] [--height H] [--spp N] [--pass P]
//                    [--depth D] [--seed S] [--tonemap reinhard|exposure|none] [--gamma G]
//                    [--exposure E] [--point-light] [--obj FILE x y z] [--unlit] [--device I]
//                    [--out image.ppm] [--raw accum.f64] [--checkpoint FILE] [--resume FILE]
//
// --checkpoint FILE rewrites FILE after every pass (header: magic, width, height, samples done, seed; then the
// W*H*3 accumulator doubles); --resume FILE continues such a render: samples [done, spp) are rendered on top of
// the stored sums.  The reference has nothing comparable (its accumulators live and die with the process).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcrt_scene.hpp"

namespace {
struct CheckpointHeader { char magic[8]; uint32_t width, height; uint64_t samples, seed; };

bool WriteCheckpoint(const std::string& path, Scene& scene, Camera& camera, uint64_t samples, uint64_t seed)
{
    if (!scene.ReadBack(camera)) return false;
    std::vector<double> sums(size_t(camera.pixels.numPixels()) * 3);
    for (int y = 0; y < camera.pixels.height(); ++y)
        for (int x = 0; x < camera.pixels.width(); ++x) {
            ColorDbl c = camera.pixels.GetPixelColor(unsigned(x), unsigned(y));
            double* p = &sums[(size_t(y) * size_t(camera.pixels.width()) + size_t(x)) * 3];
            p[0] = c.r; p[1] = c.g; p[2] = c.b;
        }
    CheckpointHeader h = { { 'M', 'C', 'R', 'T', 'C', 'K', 'P', '1' }, uint32_t(camera.pixels.width()), uint32_t(camera.pixels.height()), samples, seed };
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(sums.data(), sizeof(double), sums.size(), f) == sums.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;      // never leaves a half-written checkpoint behind
}

bool ReadCheckpoint(const std::string& path, unsigned width, unsigned height, uint64_t seed, std::vector<double>& sums, uint64_t& samples)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    CheckpointHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "MCRTCKP1", 8) == 0 && h.width == width && h.height == height && h.seed == seed;
    if (ok) {
        sums.resize(size_t(width) * height * 3);
        ok = std::fread(sums.data(), sizeof(double), sums.size(), f) == sums.size();
        samples = h.samples;
    }
    std::fclose(f);
    return ok;
}

bool WritePPM(const std::string& path, const std::vector<uint8_t>& bgra, int w, int h)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<uint8_t> row(size_t(w) * 3);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = &bgra[(size_t(y) * w + x) * 4];
            row[size_t(x) * 3] = p[2]; row[size_t(x) * 3 + 1] = p[1]; row[size_t(x) * 3 + 2] = p[0];
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}
} // namespace

int main(int argc, char** argv)
{
    std::string sceneName = "cornell", out = "render.ppm", raw, tonemapName = "reinhard", objPath, checkpoint, resume;
    unsigned width = 640, height = 480, spp = 64, pass = 64, depth = 100;   // main.cpp:33-41 defaults
    uint64_t seed = 0x5EED0001ull;
    double gamma = 2.2, exposure = 1.0;                                     // main.cpp:47-48
    float objPos[3] = { 0, 0, 0 };
    bool pointLight = false, unlit = false;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        std::string a = argv[i];
        if (a == "--scene") sceneName = next("--scene");
        else if (a == "--width") width = unsigned(std::atoi(next("--width")));
        else if (a == "--height") height = unsigned(std::atoi(next("--height")));
        else if (a == "--spp") spp = unsigned(std::atoi(next("--spp")));
        else if (a == "--pass") pass = unsigned(std::atoi(next("--pass")));
        else if (a == "--depth") depth = unsigned(std::atoi(next("--depth")));
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 0);
        else if (a == "--tonemap") tonemapName = next("--tonemap");
        else if (a == "--gamma") gamma = std::atof(next("--gamma"));
        else if (a == "--exposure") exposure = std::atof(next("--exposure"));
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--out") out = next("--out");
        else if (a == "--raw") raw = next("--raw");
        else if (a == "--checkpoint") checkpoint = next("--checkpoint");
        else if (a == "--resume") resume = next("--resume");
        else if (a == "--point-light") pointLight = true;
        else if (a == "--unlit") unlit = true;
        else if (a == "--obj") { objPath = next("--obj"); for (float& v : objPos) v = float(std::atof(next("--obj"))); }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (pass == 0 || pass > spp) pass = spp;
    const int tonemap = tonemapName == "none" ? MCRT_TONEMAP_NONE : tonemapName == "exposure" ? MCRT_TONEMAP_EXPOSURE : MCRT_TONEMAP_REINHARD;

    Camera camera{ width, height, 90.0f };                                  // CAMERA_FOV, main.cpp:37
    Scene* scene = nullptr;
    if (sceneName == "hexagon") {
        HexagonScene* s = new HexagonScene();
        s->MoveCameraToRecommendedPosition(camera);
        s->AddExampleObjects();
        s->AddExampleLight(ColorDbl{ 10.0 }, pointLight);
        scene = s;
    } else {
        CornellBoxScene* s = new CornellBoxScene{ 10.0f, 10.0f, 10.0f };
        s->backgroundColor = { 0.0, 0.0, 0.0 };
        s->MoveCameraToRecommendedPosition(camera);
        s->AddExampleObjects();
        s->AddExampleLight(ColorDbl{ 10.0 }, pointLight);                   // LIGHT_STRENGTH, main.cpp:43
        scene = s;
    }
    if (!objPath.empty()) {
        TriangleMesh* mesh = scene->CreateObject<TriangleMesh>();
        mesh->position = vec3(objPos[0], objPos[1], objPos[2]);
        mesh->material.color = ColorDbl(0.5);
        if (!mesh->LoadMesh(objPath)) return 1;
    }
    scene->UseDevice(device);
    scene->PrepareForRayTracing();
    if (!scene->Context()) return 1;
    scene->octree.PrintDebug();

    std::vector<uint8_t> image;
    const auto t0 = std::chrono::steady_clock::now();
    if (unlit) {
        // RAY_TRACE_UNLIT, main.cpp:39,124-135
        std::vector<double> colour(size_t(width) * height * 3);
        if (mcrt_set_camera(scene->Context(), &camera.Description()) != MCRT_OK || mcrt_render_unlit(scene->Context(), colour.data()) != MCRT_OK) {
            std::fprintf(stderr, "unlit render failed: %s\n", mcrt_last_error(scene->Context()));
            return 1;
        }
        image.resize(size_t(width) * height * 4);
        for (size_t p = 0; p < size_t(width) * height; ++p) {
            for (int k = 0; k < 3; ++k) image[p * 4 + size_t(2 - k)] = uint8_t(std::max(std::min(1.0, colour[p * 3 + size_t(k)]), 0.0) * 255.0);
            image[p * 4 + 3] = 255;
        }
    } else {
        unsigned first = 0;
        if (!resume.empty()) {
            std::vector<double> sums;
            uint64_t samples = 0;
            if (!ReadCheckpoint(resume, width, height, seed, sums, samples) || samples > spp) {
                std::fprintf(stderr, "cannot resume from %s (missing, or other image size / seed, or more samples than --spp)\n", resume.c_str());
                return 1;
            }
            if (!scene->Restore(camera, sums, samples)) return 1;
            first = unsigned(samples);
            std::printf("Resumed at %u of %u spp\r\n", first, spp);
        }
        for (unsigned done = first; done < spp; done += pass) {
            const unsigned n = std::min(pass, spp - done);
            if (!scene->Render(camera, done, n, depth, seed)) return 1;
            if (!scene->Resolve(camera, tonemap, gamma, exposure, image)) return 1;      // what the window would show after this pass
            if (!checkpoint.empty() && !WriteCheckpoint(checkpoint, *scene, camera, done + n, seed)) { std::fprintf(stderr, "cannot write %s\n", checkpoint.c_str()); return 1; }
        }
        if (first >= spp && !scene->Resolve(camera, tonemap, gamma, exposure, image)) return 1;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Render finished at %.3f s: %ux%u, %u spp, %.1f Msamples/s\r\n", secs, width, height, spp,
                unlit ? 0.0 : double(width) * height * spp / secs / 1e6);
    if (!WritePPM(out, image, int(width), int(height))) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    if (!raw.empty() && !unlit) {
        if (!scene->ReadBack(camera)) return 1;
        FILE* f = std::fopen(raw.c_str(), "wb");
        if (!f) return 1;
        for (unsigned y = 0; y < height; ++y)
            for (unsigned x = 0; x < width; ++x) {
                ColorDbl c = camera.pixels.GetPixelColor(x, y);
                std::fwrite(&c.r, sizeof(double), 3, f);
            }
        std::fclose(f);
    }
    delete scene;
    return 0;
}
