// grt_render — headless render/bench CLI that replaces the reference's windowed main loop
// (src/main.cpp:9-131) on a display-less MI355X: same flags -p/--ply, --width, --height and the same
// defaults (src/main.cpp:62-66: ../data/train.ply, 1280x720), the same call sequence
// (tracer.setSize -> initializeOptix -> camera init (src/gui.cpp:50-67) -> updateCamera -> render),
// plus --fisheye --type --sh-degree --plane --sphere --obj --bounces --out frame.ppm|frame.png|frame.npy --bench N
// --gpus N (tile-sharded over N GPUs of the node: one tracer and one host thread per GPU; the ranks' tile buffers reach the first
// GPU by ONE RCCL gather per frame over xGMI — grouped ncclSend / ncclRecv, --gather rccl, the default on distinct devices —
// or by peer copies, --gather peer, the fallback when ranks share a device).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../host/Display.h"
#include "../host/GaussianTracer.h"
#include "../host/HipGlue.h"

// Minimal PNG writer (8-bit RGB, stored deflate blocks: no compression library needed; CRC-32 and Adler-32 per the
// PNG / zlib specifications)
static void write_png(const std::string& path, const unsigned char* rgb_top_down, unsigned int w, unsigned int h)
{
    static unsigned int crc_table[256];
    if (!crc_table[1])
        for (unsigned int n = 0; n < 256; n++) {
            unsigned int c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
    auto be32 = [](std::vector<unsigned char>& v, unsigned int x) {
        v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
    };
    std::ofstream f(path, std::ios::binary);
    auto chunk = [&](const char* type, const std::vector<unsigned char>& data) {
        std::vector<unsigned char> buf;
        be32(buf, (unsigned int)data.size());
        buf.insert(buf.end(), type, type + 4);
        buf.insert(buf.end(), data.begin(), data.end());
        unsigned int c = 0xFFFFFFFFu;
        for (size_t i = 4; i < buf.size(); i++) c = crc_table[(c ^ buf[i]) & 0xFFu] ^ (c >> 8);
        be32(buf, c ^ 0xFFFFFFFFu);
        f.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)buf.size());
    };
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<unsigned char> ihdr;
    be32(ihdr, w); be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8-bit, RGB
    chunk("IHDR", ihdr);
    std::vector<unsigned char> raw; // filter byte 0 + row
    raw.reserve((size_t)h * (w * 3 + 1));
    for (unsigned int y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb_top_down + (size_t)y * w * 3, rgb_top_down + (size_t)(y + 1) * w * 3);
    }
    std::vector<unsigned char> z = {0x78, 0x01};
    unsigned int a1 = 1, a2 = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) {
            a1 = (a1 + raw[pos + i]) % 65521u;
            a2 = (a2 + a1) % 65521u;
        }
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    }
    be32(z, (a2 << 16) | a1);
    chunk("IDAT", z);
    chunk("IEND", {});
}

// NumPy .npy v1.0: uint8 [h][w][3], C order — the buffer as the renderer wrote it (row 0 first), what numpy.load returns
static void write_npy(const std::string& path, const unsigned char* rgb, unsigned int w, unsigned int h)
{
    std::string hdr = "{'descr': '|u1', 'fortran_order': False, 'shape': (" + std::to_string(h) + ", " + std::to_string(w) + ", 3), }";
    while ((10 + hdr.size() + 1) % 64) hdr.push_back(' ');
    hdr.push_back('\n');
    std::ofstream f(path, std::ios::binary);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(reinterpret_cast<const char*>(magic), 8);
    const unsigned char len[2] = {(unsigned char)(hdr.size() & 0xFF), (unsigned char)(hdr.size() >> 8)};
    f.write(reinterpret_cast<const char*>(len), 2);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    f.write(reinterpret_cast<const char*>(rgb), (std::streamsize)((size_t)w * h * 3));
}

static void usage()
{
    std::puts("usage: grt_render [-p|--ply scene.ply] [--width W] [--height H] [--fisheye] [--type mirror|normal|glass]\n"
              "                  [--sh-degree 0..3] [--plane] [--sphere] [--obj mesh.obj] [--bounces N]\n"
              "                  [--eye x y z] [--fov deg] [--out frame.ppm|frame.png|frame.npy] [--raw frame.rgb]\n"
              "                  [--move dx dy dz] [--bench N] [--gpus N] [--devices d0,d1,...] [--gather rccl|peer]\n"
              "  --gpus N: the frame's 32x32 tiles are dealt round-robin to N GPUs of this node (one GaussianTracer and one host\n"
              "            thread per GPU, scene replicated), the tile buffers are copied to the first GPU over xGMI and un-permuted\n"
              "            there.  --devices names the GPUs (default 0..N-1; a device may repeat: ranks then share it).\n"
              "  --gather rccl|peer: how the tile buffers reach the first GPU: one RCCL gather per frame (grouped ncclSend / ncclRecv;\n"
              "            default when the devices are distinct; given explicitly it also sends a --gpus 1 frame through the tile\n"
              "            path) or hipMemcpyPeerAsync (default when ranks share a device, which RCCL does not allow).");
}

int main(int argc, char** argv)
{
    std::string ply = "../data/train.ply", out, obj, raw_out;
    float move[3] = {0.0f, 0.0f, 0.0f};
    bool have_move = false;
    unsigned int width = 1280, height = 720, sh_degree = 0, bounces = 32;
    bool fisheye = false, plane = false, sphere = false;
    int type = MIRROR, bench = 0, gpus = 1;
    std::vector<int> devices;
    std::string gather; // "" = auto
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
        else if (a == "--raw") { need(1); raw_out = argv[++i]; }
        else if (a == "--move") { need(3); for (int k = 0; k < 3; k++) move[k] = (float)std::atof(argv[++i]); have_move = true; }
        else if (a == "--bench") { need(1); bench = std::atoi(argv[++i]); }
        else if (a == "--gpus") { need(1); gpus = std::max(1, std::atoi(argv[++i])); }
        else if (a == "--gather") { need(1); gather = argv[++i]; if (gather != "rccl" && gather != "peer") { usage(); return 2; } }
        else if (a == "--devices") {
            need(1);
            for (const char* q = argv[++i]; *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') q++; if (*q) q++; }
        }
        else if (a == "--type") {
            need(1);
            const std::string t = argv[++i];
            type = t == "mirror" ? MIRROR : t == "normal" ? NORMAL : t == "glass" ? GLASS : -1;
            if (type < 0) { usage(); return 2; }
        } else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
    }
    try {
        if (devices.empty()) for (int k = 0; k < gpus; k++) devices.push_back(k);
        gpus = (int)devices.size();
        // one tracer per rank (rank 0 is the one the frame ends up on); each goes through the reference's call sequence
        std::vector<std::unique_ptr<GaussianTracer>> tracers;
        for (int k = 0; k < gpus; k++) {
            tracers.emplace_back(new GaussianTracer(ply));
            GaussianTracer& t = *tracers.back();
            t.setDevice(devices[(size_t)k]);
            t.setSize(width, height);
        }
        auto per_rank = [&](auto&& fn) { // ranks in parallel, one host thread each; the first exception is re-thrown
            std::vector<std::thread> th;
            std::vector<std::string> errs((size_t)gpus);
            for (int k = 0; k < gpus; k++)
                th.emplace_back([&, k] { try { fn(k, *tracers[(size_t)k]); } catch (const std::exception& e) { errs[(size_t)k] = e.what(); if (errs[(size_t)k].empty()) errs[(size_t)k] = "error"; } });
            for (auto& t : th) t.join();
            for (int k = 0; k < gpus; k++) if (!errs[(size_t)k].empty()) throw std::runtime_error("rank " + std::to_string(k) + ": " + errs[(size_t)k]);
        };
        per_rank([&](int, GaussianTracer& t) {
            t.initializeOptix();
            t.params.sh_degree_max = sh_degree;
            t.params.mode_fisheye = fisheye;   // gui.cpp:433
            t.params.max_bounces = bounces;
            t.setRenderType((unsigned)type);   // gui.cpp:178-184
            Camera camera;                     // GUI::initCamera, src/gui.cpp:50-67
            camera.setEye(make_float3(eye[0], eye[1], eye[2]));
            camera.setLookat(t.getGaussianCenter());
            camera.setUp(make_float3(0.0f, 1.0f, 0.0f));
            camera.setFovY(fov);
            bool camera_changed = true;
            t.updateCamera(camera, camera_changed);
            if (plane) t.createPlane();        // gui.cpp:170
            if (sphere) t.createSphere();      // gui.cpp:173
            if (!obj.empty()) t.createLoadMesh(obj);
            if (have_move && !t.getPrimitives().empty()) { // a gizmo drag of the last primitive (src/gui.cpp:430-433)
                Primitive& pr = t.getPrimitives().back();
                pr.transform.m[3][0] += move[0]; pr.transform.m[3][1] += move[1]; pr.transform.m[3][2] += move[2];
                t.updateInstanceTransforms(pr);
            }
        });
        GaussianTracer& tracer = *tracers[0];
        hipglue::setDevice(tracer.device());
        CUDAOutputBuffer output_buffer(width, height); // = HIPOutputBuffer (src/main.cpp:77)
        output_buffer.setStream(tracer.stream);

        // ---- N > 1: tile t of the row-major 32x32 grid belongs to rank t % N (its tile t / N), as in bench.py ----
        const unsigned int TILE = 32, tiles_x = (width + TILE - 1) / TILE, tiles_y = (height + TILE - 1) / TILE;
        const unsigned int n_tiles = tiles_x * tiles_y, max_cnt = (n_tiles + (unsigned)gpus - 1) / (unsigned)gpus;
        const size_t tile_bytes = (size_t)TILE * TILE * 3;
        std::vector<unsigned char*> mine((size_t)gpus, nullptr);
        unsigned char* gathered = nullptr;
        bool distinct = true;
        for (size_t i = 0; i < devices.size(); i++) for (size_t j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
        if (gather == "rccl" && !distinct) throw std::runtime_error("--gather rccl needs distinct devices (RCCL allows one rank per GPU)");
        const bool use_rccl = gather == "rccl" || (gather.empty() && gpus > 1 && distinct);
        const bool tiled = gpus > 1 || gather == "rccl"; // the N-rank path (at one rank too, when the collective is asked for)
        hipglue::Rccl* rccl = use_rccl ? hipglue::rcclInitAll(devices.data(), gpus) : nullptr;
        std::vector<size_t> bytes_of_rank((size_t)gpus, 0);
        for (int k = 0; k < gpus; k++)
            bytes_of_rank[(size_t)k] = (size_t)((n_tiles > (unsigned)k) ? (n_tiles - (unsigned)k + (unsigned)gpus - 1) / (unsigned)gpus : 0u) * tile_bytes;
        if (tiled) {
            for (int k = 0; k < gpus; k++) {
                hipglue::setDevice(devices[(size_t)k]);
                mine[(size_t)k] = static_cast<unsigned char*>(hipglue::deviceAlloc(max_cnt * tile_bytes));
            }
            hipglue::setDevice(tracer.device());
            gathered = static_cast<unsigned char*>(hipglue::deviceAlloc((size_t)gpus * max_cnt * tile_bytes));
        }
        auto frame = [&] {
            if (!tiled) { tracer.render(output_buffer); return; }
            per_rank([&](int k, GaussianTracer& t) {
                const unsigned int cnt = (unsigned int)(bytes_of_rank[(size_t)k] / tile_bytes);
                t.renderTiles(mine[(size_t)k], TILE, TILE, (unsigned)k, (unsigned)gpus, cnt);
                // the rank's compact buffer -> its slice of rank 0's gather buffer, behind the render on the rank's stream:
                // one RCCL gather over xGMI (every rank's thread joins the group), or a peer copy
                if (rccl)
                    hipglue::rcclGatherToRoot(rccl, k, mine[(size_t)k], bytes_of_rank[(size_t)k], gathered, bytes_of_rank.data(), max_cnt * tile_bytes, t.stream);
                else
                    hipglue::copyPeerAsync(gathered + (size_t)k * max_cnt * tile_bytes, tracer.device(), mine[(size_t)k], t.device(),
                                           (size_t)cnt * tile_bytes, t.stream);
                t.sync();
            });
            tracer.assembleTiles(gathered, (unsigned)gpus, max_cnt, TILE, TILE, output_buffer);
            tracer.sync();
        };
        frame();
        if (bench > 0) {
            for (int i = 0; i < 3; i++) frame();
            std::vector<double> ms;
            double kms = 0.0;
            for (int i = 0; i < bench; i++) {
                const auto t0 = std::chrono::steady_clock::now();
                frame();   // includes the device sync, like the reference's render timer
                ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                kms += tracer.lastKernelMs();
            }
            std::sort(ms.begin(), ms.end());
            const double med = ms[ms.size() / 2];
            std::printf("frames %d  median %.3f ms/frame  kernel %.3f ms%s  %.1f Mrays/s (primary)  gpus %d  gather %s\n", bench, med, kms / bench,
                        gpus > 1 ? " (rank 0)" : "", (double)width * height / med / 1e3, gpus, !tiled ? "none" : (rccl ? "rccl" : "peer"));
        }
        for (int k = 0; k < gpus; k++) { if (mine[(size_t)k]) { hipglue::setDevice(devices[(size_t)k]); hipglue::deviceFree(mine[(size_t)k]); } }
        hipglue::setDevice(tracer.device());
        hipglue::deviceFree(gathered);
        hipglue::rcclDestroy(rccl);
        if (!raw_out.empty()) { // the buffer as the renderer wrote it (row 0 first), from the pinned mirror render() filled
            std::ofstream f(raw_out, std::ios::binary);
            f.write(reinterpret_cast<const char*>(output_buffer.getHostPointer()), (std::streamsize)((size_t)width * height * 3));
        }
        if (!out.empty()) {
            // what the viewer's window shows: buffer row 0 at the bottom (src/Display.cpp:13,184)
            const std::vector<unsigned char> top_down = GLDisplay::windowImage(output_buffer);
            if (out.size() > 4 && out.compare(out.size() - 4, 4, ".npy") == 0) {
                // an array for parity checks, not a picture: the renderer's own row order (shaders/tracer.cuh:487)
                write_npy(out, reinterpret_cast<const unsigned char*>(output_buffer.getHostPointer()), width, height);
            } else if (out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0) {
                write_png(out, top_down.data(), width, height);
            } else {
                std::ofstream f(out, std::ios::binary);
                f << "P6\n" << width << " " << height << "\n255\n";
                f.write(reinterpret_cast<const char*>(top_down.data()), (std::streamsize)top_down.size());
            }
            std::cout << "wrote " << out << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
