// cli.cpp — headless command-line driver: what the reference's GUI main loop does
// (src/main.cpp:549-623: scene set-up, camera, direct lighting or progressive path tracing with
// pathsPerPass / pathsPerPixel), without a window. Writes PFM (float RGB, bottom-up like our rows) and/or
// an 8-bit PPM (what the reference's default framebuffer would show: clamped to [0,1]), prints one JSON line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <unistd.h>

#include "renderer.h"
#include "scenes.h"

using gpuart::Vec3f;

static void usage() {
    std::cerr << "usage: gpuart_cli [--scene box|ply:<file>|cluster|tree] [--width W] [--height H] [--mode direct|pt]\n"
                 "                  [--spp N] [--per-pass K] [--max-segments M] [--seed S] [--tile x0,y0,w,h]\n"
                 "                  [--camera px,py,pz] [--sun az,alt[,off]] [--user-sphere x,y,z,r,em[,specular[,fuzzy]]]\n"
                 "                  [--device D] [--gpus N] [--resume ck] [--checkpoint ck] [--pfm out.pfm] [--ppm out.ppm] [--nearest-first]\n"
                 "  --nearest-first: opt in to the nearer-child-first BVH walk (~10 % faster; soak-verified, not proven to be the reference's image)\n"
                 "  --gpus N: path tracing of ONE frame on devices D..D+N-1 (8-row bands dealt round-robin, gathered over RCCL)\n";
}

static bool parse_floats(const char *s, float *out, int minN, int maxN, int &n) {
    n = 0;
    while (*s && n < maxN) {
        char *end;
        out[n++] = strtof(s, &end);
        if (end == s) return false;
        s = *end == ',' ? end + 1 : end;
    }
    return n >= minN;
}

int main(int argc, char **argv) {
    std::string scene = "box", mode = "pt", pfm, ppm, resume, checkpoint;
    unsigned W = 640, H = 480, spp = 16, perPass = 1, maxSeg = 5, device = 0, gpus = 1;
    long seed = -1;
    bool nearestFirst = false;
    float tile[4] = {0, 0, 0, 0}, campos[3] = {0.1f, -3.05f, 1.0f}, sun[3] = {0, 0, 0}, us[7] = {-0.4f, 0, 0.2f, 0, 0, 0, 0};
    int nTile = 0, nSun = 0, nUs = 0, n;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { std::cerr << what << " needs a value\n"; usage(); exit(2); }
            return argv[++i];
        };
        if (a == "--scene") scene = need("--scene");
        else if (a == "--width") W = (unsigned)atoi(need("--width"));
        else if (a == "--height") H = (unsigned)atoi(need("--height"));
        else if (a == "--mode") mode = need("--mode");
        else if (a == "--spp") spp = (unsigned)atoi(need("--spp"));
        else if (a == "--per-pass") perPass = (unsigned)atoi(need("--per-pass"));
        else if (a == "--max-segments") maxSeg = (unsigned)atoi(need("--max-segments"));
        else if (a == "--seed") seed = atol(need("--seed"));
        else if (a == "--device") device = (unsigned)atoi(need("--device"));
        else if (a == "--gpus") gpus = (unsigned)atoi(need("--gpus"));
        else if (a == "--tile") { if (!parse_floats(need("--tile"), tile, 4, 4, nTile)) { usage(); return 2; } }
        else if (a == "--camera") { if (!parse_floats(need("--camera"), campos, 3, 3, n)) { usage(); return 2; } }
        else if (a == "--sun") { if (!parse_floats(need("--sun"), sun, 2, 3, nSun)) { usage(); return 2; } }
        else if (a == "--user-sphere") { if (!parse_floats(need("--user-sphere"), us, 5, 7, nUs)) { usage(); return 2; } }
        else if (a == "--pfm") pfm = need("--pfm");
        else if (a == "--ppm") ppm = need("--ppm");
        else if (a == "--resume") resume = need("--resume");
        else if (a == "--checkpoint") checkpoint = need("--checkpoint");
        else if (a == "--nearest-first") nearestFirst = true;
        else { usage(); return 2; }
    }
    if (W == 0 || H == 0 || (mode != "direct" && mode != "pt")) { usage(); return 2; }

    // the reference's start-up camera (src/main.cpp:609-613), looking at (0,0,0.95)
    gpuart::Camera cam;
    cam.Pos = Vec3f(campos[0], campos[1], campos[2]);
    cam.Up = Vec3f(0, 0, 1);
    cam.Dir = Vec3f(0, 0, 0.95f) - cam.Pos;
    cam.FovY = 60;
    cam.ScreenDist = 0.2f;

    // One Renderer per GPU; with --gpus N every one is set up identically (same scene, camera, lighting, seed: all draw the
    // same RandSeed sequence) and renders its share of the frame.
    if (gpus < 1 || gpus > 64 || (gpus > 1 && (mode != "pt" || nTile == 4 || !resume.empty() || !checkpoint.empty()))) { usage(); return 2; }
    std::vector<std::unique_ptr<gpuart::Renderer>> rs;
    bool ok = true;
    for (unsigned g = 0; g < gpus && ok; g++) {
        // (GPUART_CLI_SHARED_DEVICE: every rank on --device — only an in-process RCCL stand-in accepts that: tests/test_gather_inprocess.py)
        const unsigned dev = getenv("GPUART_CLI_SHARED_DEVICE") ? device : device + g;
        rs.emplace_back(new gpuart::Renderer(W, H, cam, (int)dev));
        gpuart::Renderer &r = *rs.back();
        if (!r.GetIsOK()) { std::cerr << "Renderer initialization failed on device " << dev << "\n"; return 1; }
        r.SetUserSphere(Vec3f(us[0], us[1], us[2]), us[3], us[4]);
        if (nUs >= 6) r.SetUserSphereSpecular(us[5] != 0);
        if (nUs >= 7) r.SetUserSphereFuzzy(us[6] != 0);
        if (nSun >= 2) { r.SetSunAzimuth(sun[0]); r.SetSunAltitude(sun[1]); if (nSun == 3) r.SetSunDirectLighting(sun[2] == 0); }
        r.SetMaxPathSegments(maxSeg);
        if (seed >= 0) r.SetSeed((uint32_t)seed);
        if (nearestFirst && !r.SetNearestFirst(1024)) return 1;
        if (scene == "box") InitBox(r);
        else if (scene.compare(0, 4, "ply:") == 0) ok = InitDragon(r, scene.c_str() + 4);
        else if (scene.compare(0, 8, "cluster:") == 0) ok = InitCluster(r, scene.c_str() + 8);
        else if (scene.compare(0, 5, "tree:") == 0) ok = InitTree(r, scene.c_str() + 5);
        else if (scene == "cluster") ok = InitCluster(r);
        else if (scene == "tree") ok = InitTree(r);
        else { usage(); return 2; }
        if (!ok || !r.GetIsOK()) { std::cerr << "scene set-up failed\n"; return 1; }
        if (gpus > 1 && !r.SetShare((int)g, (int)gpus)) return 1;
    }
    gpuart::Renderer &r = *rs[0];
    if (nTile == 4 && !r.SetTile((unsigned)tile[0], (unsigned)tile[1], (unsigned)tile[2], (unsigned)tile[3])) return 1;
    const unsigned tw = gpus > 1 ? W : r.GetTileWidth(), th = gpus > 1 ? H : r.GetTileHeight();

    std::vector<float> img((size_t)tw * th * 4);
    const auto t0 = std::chrono::high_resolution_clock::now();
    unsigned done = 0, passes = 0;
    if (mode == "direct") {
        r.RenderDirectLighting();
        ok = r.ReadDirectLighting(img.data());
    } else {
        for (auto &q : rs) q->RestartPathTracing(perPass, spp);
        if (!resume.empty()) {
            if (!r.LoadCheckpoint(resume.c_str())) return 1;
            // the checkpoint restores the run's own target (normally already reached); the command line's --spp /
            // --per-pass say how far to go on from there
            r.ExtendPathTracing(perPass, spp);
        }
        // the reference's draw loop: one pass per frame until pathsPerPixel is reached (src/main.cpp:554-582)
        // (every GPU's pass is only enqueued: the devices work at the same time)
        for (;;) {
            for (auto &q : rs) done = q->RenderPathTracingPass();
            passes++;
            if (done >= r.GetPathsPerPixel()) break;
        }
        for (auto &q : rs) q->Finish();
        if (!checkpoint.empty() && !r.SaveCheckpoint(checkpoint.c_str())) return 1;
        if (gpus > 1 || getenv("GPUART_CLI_FORCE_GATHER")) {  // (the variable: the RCCL path with a single rank, for tests)
            std::vector<gpuart::Renderer *> ranks;
            for (auto &q : rs) ranks.push_back(q.get());
            ok = gpuart::Renderer::GatherRadiance(ranks.data(), (int)gpus, 0, true, img.data());
            // The communicator goes in a phase of its own, not in the destructors at exit. A read-out that gave up (a bounded wait
            // of the library ran out: a message above says which) leaves streams and possibly a parked RCCL thread behind:
            // nothing of that is waited for again — no destructors, no atexit handlers.
            if (ok) ok = gpuart::Renderer::ReleaseCommunicator(ranks.data(), (int)gpus);
            if (!ok) {
                std::cerr << "gpuart_cli: the multi-GPU read-out failed" << (gpuart_hip_comm_stuck() ? " (an RCCL call never returned)" : "")
                          << "; ending without unwinding." << std::endl;
                fflush(nullptr);
                _exit(1);
            }
        } else
            ok = r.ReadRadiance(img.data(), true);
    }
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (!ok) return 1;

    if (!pfm.empty()) {  // PFM rows run bottom-to-top, as ours do
        FILE *f = fopen(pfm.c_str(), "wb");
        if (!f) return 1;
        fprintf(f, "PF\n%u %u\n-1.0\n", tw, th);
        for (size_t i = 0; i < (size_t)tw * th; i++) fwrite(&img[4 * i], sizeof(float), 3, f);
        fclose(f);
    }
    if (!ppm.empty()) {  // top-down, clamped to [0,1] like the GL default framebuffer
        FILE *f = fopen(ppm.c_str(), "wb");
        if (!f) return 1;
        fprintf(f, "P6\n%u %u\n255\n", tw, th);
        for (unsigned y = th; y-- > 0;)
            for (unsigned x = 0; x < tw; x++)
                for (int c = 0; c < 3; c++) {
                    float v = img[4 * ((size_t)y * tw + x) + c];
                    v = v != v ? 0.0f : (v < 0 ? 0.0f : (v > 1 ? 1.0f : v));
                    fputc((int)std::lround(v * 255.0f), f);
                }
        fclose(f);
    }
    printf("{\"scene\": \"%s\", \"mode\": \"%s\", \"frame\": [%u, %u], \"tile\": [%u, %u], \"gpus\": %u, \"paths_per_pixel\": %u, "
           "\"passes\": %u, \"seconds\": %.6f, \"mpaths_per_s\": %.3f}\n",
           scene.c_str(), mode.c_str(), W, H, tw, th, gpus, done, passes, secs,
           mode == "pt" ? (double)tw * th * done / secs / 1e6 : (double)tw * th / secs / 1e6);
    return 0;
}
