// mcrt_render - headless driver: what source/main.cpp does without the window.
//
//   main.cpp:234-242   build the default scene, camera, PrepareForRayTracing
//   main.cpp:100-184   render loop            -> Scene::Render (GPU), progressive passes
//   main.cpp:142-164   tonemap + 8-bit store  -> Scene::Resolve (GPU)
//   screenshot.cpp     PNG of the framebuffer -> --screenshot NAME (TakeScreenshot: NAME00001.png, ...), --out FILE.png|.ppm
//   main.cpp:59-62,250-262   one worker thread per hardware thread -> --devices: one thread + context per GPU
//   main.cpp:70-86     RAY_TRACE_RANDOM pixel picks -> --random-pixels
//
// Usage: mcrt_render [--scene cornell|hexagon] [--width W] [--height H] [--spp N] [--pass P]
//                    [--depth D] [--seed S] [--tonemap reinhard|exposure|none] [--gamma G]
//                    [--exposure E] [--point-light] [--obj FILE x y z] [--unlit] [--device I]
//                    [--devices I,J,...] [--random-pixels N [--pick-seed S]] [--device-build]
//                    [--out image.ppm|image.png] [--screenshot NAME] [--raw accum.f64] [--checkpoint FILE] [--resume FILE]
//
// --devices I,J,...  one context and one host thread per listed GPU.  Every pass of P samples per pixel is split
// into contiguous ranges of global sample indices, one per device (the first P % n devices take one sample more);
// the streams are keyed by the global sample index, so the image does not depend on the device list beyond
// summation order.  After a pass the accumulators are summed into the first device's (ONE ncclReduce of W*H*3
// doubles over RCCL, mcrt_reduce), which resolves; the others start their next pass from zero.  A device listed
// more than once (--devices 0,0: several contexts sharing one GPU) is served by a plain device add instead
// (mcrt_add_accumulator); the list is either all distinct or all the same device.
//
// --random-pixels N  the reference's default scheduler (RAY_TRACE_RANDOM, main.cpp:40,70-86): N times, pick a
// pixel with two draws of the driver's generator (seeded by --pick-seed: the reference seeds it from
// std::random_device), trace ONE sample of it and accumulate; a pixel's k-th pick is its sample index k.  Every
// pixel is normalised by its own ray count (main.cpp:142).  --spp / --pass are not used in this mode.
//
// --checkpoint FILE rewrites FILE after every pass (header: magic, width, height, samples done, seed; then the
// W*H*3 accumulator doubles); --resume FILE continues such a render: samples [done, spp) are rendered on top of
// the stored sums.  The reference has nothing comparable (its accumulators live and die with the process).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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

// all threads of a multi-device render meet here (C++17 has no std::barrier)
class Barrier {
    std::mutex m;
    std::condition_variable cv;
    unsigned n, waiting = 0, generation = 0;
public:
    explicit Barrier(unsigned count) : n(count) {}
    void Wait()
    {
        std::unique_lock<std::mutex> lock(m);
        const unsigned gen = generation;
        if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lock, [&] { return gen != generation; });
    }
};

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
    std::string sceneName = "cornell", out = "render.ppm", screenshot, raw, tonemapName = "reinhard", objPath, checkpoint, resume;
    unsigned width = 640, height = 480, spp = 64, pass = 64, depth = 100;   // main.cpp:33-41 defaults
    uint64_t seed = 0x5EED0001ull, pickSeed = 1, randomPixels = 0;
    double gamma = 2.2, exposure = 1.0;                                     // main.cpp:47-48
    float objPos[3] = { 0, 0, 0 };
    bool pointLight = false, unlit = false, deviceBuild = false;
    std::vector<int> devices;
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
        else if (a == "--pick-seed") pickSeed = std::strtoull(next("--pick-seed"), nullptr, 0);
        else if (a == "--random-pixels") randomPixels = std::strtoull(next("--random-pixels"), nullptr, 0);
        else if (a == "--tonemap") tonemapName = next("--tonemap");
        else if (a == "--gamma") gamma = std::atof(next("--gamma"));
        else if (a == "--exposure") exposure = std::atof(next("--exposure"));
        else if (a == "--device") devices.assign(1, std::atoi(next("--device")));
        else if (a == "--devices") {
            devices.clear();
            for (const char* p = next("--devices"); *p;) {
                char* end = nullptr;
                devices.push_back(int(std::strtol(p, &end, 10)));
                if (end == p) { std::fprintf(stderr, "--devices wants a comma-separated list of device indices\n"); return 2; }
                p = (*end == ',') ? end + 1 : end;
            }
        }
        else if (a == "--out") out = next("--out");
        else if (a == "--screenshot") screenshot = next("--screenshot");
        else if (a == "--raw") raw = next("--raw");
        else if (a == "--checkpoint") checkpoint = next("--checkpoint");
        else if (a == "--resume") resume = next("--resume");
        else if (a == "--point-light") pointLight = true;
        else if (a == "--device-build") deviceBuild = true;
        else if (a == "--unlit") unlit = true;
        else if (a == "--obj") { objPath = next("--obj"); for (float& v : objPos) v = float(std::atof(next("--obj"))); }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (devices.empty()) devices.push_back(0);
    if (pass == 0 || pass > spp) pass = spp;
    const int tonemap = tonemapName == "none" ? MCRT_TONEMAP_NONE : tonemapName == "exposure" ? MCRT_TONEMAP_EXPOSURE : MCRT_TONEMAP_REINHARD;
    const unsigned nDev = unsigned(devices.size());
    bool distinct = true, same = true;
    for (unsigned i = 0; i < nDev; ++i)
        for (unsigned j = i + 1; j < nDev; ++j) { if (devices[i] == devices[j]) distinct = false; else same = false; }
    if (nDev > 1 && !distinct && !same) { std::fprintf(stderr, "--devices: list every GPU once, or one GPU several times\n"); return 2; }
    if (nDev > 1 && (unlit || randomPixels)) { std::fprintf(stderr, "--unlit and --random-pixels run on one device\n"); return 2; }

    // one scene + camera per device (main.cpp:234-242): the description is replicated, every GPU holds its own image of it
    std::vector<std::unique_ptr<Scene>> scenes;
    std::vector<std::unique_ptr<Camera>> cameras;
    for (unsigned r = 0; r < nDev; ++r) {
        cameras.emplace_back(new Camera{ width, height, 90.0f });               // CAMERA_FOV, main.cpp:37
        Camera& camera = *cameras.back();
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
        scenes.emplace_back(scene);
        if (!objPath.empty()) {
            TriangleMesh* mesh = scene->CreateObject<TriangleMesh>();
            mesh->position = vec3(objPos[0], objPos[1], objPos[2]);
            mesh->material.color = ColorDbl(0.5);
            if (!mesh->LoadMesh(objPath)) return 1;
        }
        scene->UseDevice(devices[r]);
        scene->octree.buildOnDevice = deviceBuild;
        scene->PrepareForRayTracing();
        if (!scene->Context()) return 1;
    }
    Scene* scene = scenes[0].get();
    Camera& camera = *cameras[0];
    scene->octree.PrintDebug();

    std::vector<void*> comms(nDev, nullptr);
    if (nDev > 1 && distinct) {
        if (mcrt_comm_init_all(int(nDev), devices.data(), comms.data()) != MCRT_OK) {
            std::fprintf(stderr, "mcrt_comm_init_all: %s\n", mcrt_last_error(nullptr));
            return 1;
        }
    }

    std::vector<uint8_t> image;
    uint64_t samplesTraced = 0, droppedTotal = 0;
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
    } else if (randomPixels) {
        // GetNextPixelToRender with RAY_TRACE_RANDOM (main.cpp:70-86) + the one-sample loop of RayTraceNextPixel (:122-139).
        // The picks of a batch are drawn first (they depend on the driver's generator alone), traced together on the
        // GPU, and accumulated in pick order - the sum a single reference thread would build.
        UniformRandomGenerator picker(pickSeed, pickSeed ^ 0x9E3779B97F4A7C15ull);
        std::vector<uint32_t> nextSample(size_t(width) * height, 0), pixels, samples;
        std::vector<double> radiance;
        const uint64_t batch = 1u << 20;
        for (uint64_t done = 0; done < randomPixels; done += batch) {
            const uint64_t n = std::min<uint64_t>(batch, randomPixels - done);
            pixels.resize(n); samples.resize(n);
            for (uint64_t i = 0; i < n; ++i) {
                unsigned x = (unsigned int)(picker.RandomFloat(0.0f, float(width)));
                unsigned y = (unsigned int)(picker.RandomFloat(0.0f, float(height)));
                x = std::min(x, width - 1);
                y = std::min(y, height - 1);
                pixels[i] = y * width + x;
                samples[i] = nextSample[pixels[i]]++;
            }
            if (!scene->TraceSamples(camera, pixels, samples, seed, depth, radiance)) return 1;
            for (uint64_t i = 0; i < n; ++i) {
                const ColorDbl c(radiance[3 * i], radiance[3 * i + 1], radiance[3 * i + 2]);
                if (!(std::isfinite(c.r) && std::isfinite(c.g) && std::isfinite(c.b))) continue;    // dropped, as in Scene::Render (stated deviation)
                camera.pixels.Accumulate(pixels[i] % width, pixels[i] / width, c);
            }
            if (!scene->ResolvePerPixelCounts(camera, tonemap, gamma, exposure, image)) return 1;  // what the window would show now
        }
        samplesTraced = randomPixels;
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
        // One worker per device.  Rank 0 (the first device) holds the job's sums, resolves and checkpoints.
        Barrier barrier(nDev);
        std::vector<int> failed(nDev, 0);
        std::atomic<bool> anyFailed{ false };        // a rank whose pass failed: EVERY rank leaves before the collective, or all ranks would wait for it forever
        std::vector<uint64_t> droppedCleared(nDev, 0);   // non-finite samples a rank counted before its per-pass Clear zeroed the counter
        auto worker = [&](unsigned r) {
            Scene& sc = *scenes[r];
            Camera& cam = *cameras[r];
            bool ok = true;
            for (unsigned done = first; done < spp; done += pass) {
                const unsigned n = std::min(pass, spp - done);
                // this rank's share of the window [done, done + n): contiguous, sizes differ by at most one
                const unsigned base = n / nDev, extra = n % nDev;
                const unsigned mine = base + (r < extra ? 1u : 0u), begin = done + r * base + std::min(r, extra);
                // a rank with no sample in this window still binds its camera (the reduce needs its zeroed accumulator)
                ok = ok && sc.Render(cam, begin, mine, depth, seed);
                if (nDev > 1) {
                    // every rank issues the reduction of this pass, or none does
                    if (!distinct) ok = ok && mcrt_synchronize(sc.Context()) == MCRT_OK;
                    if (!ok) anyFailed.store(true);
                    barrier.Wait();                                              // every rank's pass is enqueued (same GPU: complete) and its verdict known
                    if (anyFailed.load()) { ok = false; break; }
                    if (distinct) ok = sc.Reduce(cam, comms[r], 0);
                    else {
                        if (r == 0) for (unsigned j = 1; j < nDev && ok; ++j) ok = sc.AddAccumulatorOf(cam, *scenes[j]);
                        if (r == 0) ok = ok && mcrt_synchronize(sc.Context()) == MCRT_OK;
                        barrier.Wait();                                          // rank 0 has read the others' sums
                    }
                    if (r != 0 && ok) {
                        uint64_t d = 0;                                          // Clear zeroes the dropped-sample counter with the sums
                        if (mcrt_dropped_samples(sc.Context(), &d) == MCRT_OK) droppedCleared[r] += d;
                        ok = sc.Clear(cam);                                      // their sums now live on rank 0
                    }
                }
                if (r == 0) {
                    sc.SetSampleCount(cam, done + n);
                    ok = ok && sc.Resolve(cam, tonemap, gamma, exposure, image);      // what the window would show after this pass
                    if (ok && !checkpoint.empty() && !WriteCheckpoint(checkpoint, sc, cam, done + n, seed)) { std::fprintf(stderr, "cannot write %s\n", checkpoint.c_str()); ok = false; }
                }
                if (!ok && nDev == 1) break;
                // (nDev > 1: a failure after the reduction is reported at the barrier of the next pass, where all ranks leave together)
            }
            failed[r] = ok ? 0 : 1;
        };
        std::vector<std::thread> threads;
        for (unsigned r = 1; r < nDev; ++r) threads.emplace_back(worker, r);
        worker(0);
        for (std::thread& t : threads) t.join();
        for (unsigned r = 0; r < nDev; ++r) if (failed[r]) { std::fprintf(stderr, "render failed on device %d: %s\n", devices[r], mcrt_last_error(scenes[r]->Context())); return 1; }
        for (unsigned r = 0; r < nDev; ++r) droppedTotal += droppedCleared[r];
        if (first >= spp && !scene->Resolve(camera, tonemap, gamma, exposure, image)) return 1;
        samplesTraced = uint64_t(width) * height * (spp - std::min(first, spp));
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Render finished at %.3f s: %ux%u, %u spp, %u device(s), %.1f Msamples/s\r\n", secs, width, height, randomPixels ? 0u : spp, nDev,
                double(samplesTraced) / secs / 1e6);
    for (auto& sc : scenes) { uint64_t d = 0; if (mcrt_dropped_samples(sc->Context(), &d) == MCRT_OK) droppedTotal += d; }
    if (droppedTotal) std::printf("%llu non-finite samples dropped\r\n", (unsigned long long)droppedTotal);
    const bool png = out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (!(png ? WritePNG(out, image, width, height) : WritePPM(out, image, int(width), int(height)))) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    if (!screenshot.empty() && TakeScreenshot(screenshot, image, width, height).empty()) return 1;
    if (!raw.empty() && !unlit) {
        if (!randomPixels && !scene->ReadBack(camera)) return 1;
        FILE* f = std::fopen(raw.c_str(), "wb");
        if (!f) return 1;
        for (unsigned y = 0; y < height; ++y)
            for (unsigned x = 0; x < width; ++x) {
                ColorDbl c = camera.pixels.GetPixelColor(x, y);
                std::fwrite(&c.r, sizeof(double), 3, f);
            }
        std::fclose(f);
    }
    for (void* c : comms) if (c) mcrt_comm_destroy(c);
    return 0;
}
