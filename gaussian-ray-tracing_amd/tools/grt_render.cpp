// grt_render — headless render/bench CLI that replaces the reference's windowed main loop
// (src/main.cpp:9-131) on a display-less MI355X: same flags -p/--ply, --width, --height and the same
// defaults (src/main.cpp:62-66: ../data/train.ply, 1280x720), the same call sequence
// (tracer.setSize -> initializeOptix -> camera init (src/gui.cpp:50-67) -> updateCamera -> render),
// plus --fisheye --type --sh-degree --plane --sphere --obj --bounces --out frame.ppm --bench N.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../host/GaussianTracer.h"

static void usage()
{
    std::puts("usage: grt_render [-p|--ply scene.ply] [--width W] [--height H] [--fisheye] [--type mirror|normal|glass]\n"
              "                  [--sh-degree 0..3] [--plane] [--sphere] [--obj mesh.obj] [--bounces N]\n"
              "                  [--eye x y z] [--fov deg] [--out frame.ppm] [--bench N]");
}

int main(int argc, char** argv)
{
    std::string ply = "../data/train.ply", out, obj;
    unsigned int width = 1280, height = 720, sh_degree = 0, bounces = 32;
    bool fisheye = false, plane = false, sphere = false;
    int type = MIRROR, bench = 0;
    float eye[3] = {0.0f, 0.0f, 3.0f}, fov = 60.0f;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int k) { if (i + k >= argc) { usage(); std::exit(2); } };
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-p" || a == "--ply") { need(1); ply = argv[++i]; }
        else if (a == "--width") { need(1); width = (unsigned)std::atoi(argv[++i]); }
        else if (a == "--height") { need(1); height = (unsigned)std::atoi(argv[++i]); }
        else if (a == "--fisheye") fisheye = true;
        else if (a == "--plane") plane = true;
        else if (a == "--sphere") sphere = true;
        else if (a == "--obj") { need(1); obj = argv[++i]; }
        else if (a == "--sh-degree") { need(1); sh_degree = (unsigned)std::atoi(argv[++i]); }
        else if (a == "--bounces") { need(1); bounces = (unsigned)std::atoi(argv[++i]); }
        else if (a == "--fov") { need(1); fov = (float)std::atof(argv[++i]); }
        else if (a == "--eye") { need(3); for (int k = 0; k < 3; k++) eye[k] = (float)std::atof(argv[++i]); }
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--bench") { need(1); bench = std::atoi(argv[++i]); }
        else if (a == "--type") {
            need(1);
            const std::string t = argv[++i];
            type = t == "mirror" ? MIRROR : t == "normal" ? NORMAL : t == "glass" ? GLASS : -1;
            if (type < 0) { usage(); return 2; }
        } else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
    }
    try {
        GaussianTracer tracer(ply);
        tracer.setSize(width, height);
        tracer.initializeOptix();
        tracer.params.sh_degree_max = sh_degree;
        tracer.params.mode_fisheye = fisheye;   // gui.cpp:433
        tracer.params.max_bounces = bounces;
        tracer.setRenderType((unsigned)type);   // gui.cpp:178-184

        Camera camera;                          // GUI::initCamera, src/gui.cpp:50-67
        camera.setEye(make_float3(eye[0], eye[1], eye[2]));
        camera.setLookat(tracer.getGaussianCenter());
        camera.setUp(make_float3(0.0f, 1.0f, 0.0f));
        camera.setFovY(fov);
        bool camera_changed = true;
        tracer.updateCamera(camera, camera_changed);
        if (plane) tracer.createPlane();        // gui.cpp:170
        if (sphere) tracer.createSphere();      // gui.cpp:173
        if (!obj.empty()) tracer.createLoadMesh(obj);

        HIPOutputBuffer output_buffer(width, height);
        output_buffer.setStream(tracer.stream);
        tracer.render(output_buffer);
        if (bench > 0) {
            for (int i = 0; i < 3; i++) tracer.render(output_buffer);
            std::vector<double> ms;
            double kms = 0.0;
            for (int i = 0; i < bench; i++) {
                const auto t0 = std::chrono::steady_clock::now();
                tracer.render(output_buffer);   // includes the device sync, like the reference's render timer
                ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                kms += tracer.lastKernelMs();
            }
            std::sort(ms.begin(), ms.end());
            const double med = ms[ms.size() / 2];
            std::printf("frames %d  median %.3f ms/frame  kernel %.3f ms  %.1f Mrays/s (primary)\n", bench, med, kms / bench,
                        (double)width * height / med / 1e3);
        }
        if (!out.empty()) {
            const std::vector<unsigned char>& rgb = output_buffer.download();
            std::ofstream f(out, std::ios::binary);
            f << "P6\n" << width << " " << height << "\n255\n";
            for (unsigned int y = 0; y < height; y++) // row 0 is the bottom of the window (src/Display.cpp:13,184)
                f.write(reinterpret_cast<const char*>(rgb.data() + (size_t)(height - 1 - y) * width * 3), (std::streamsize)width * 3);
            std::cout << "wrote " << out << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
