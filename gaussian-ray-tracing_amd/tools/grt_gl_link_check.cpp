// grt_gl_link_check.cpp — the viewer glue with GL interop, linked for real (-lGL -lamdhip64) and run as far as a machine
// without a display allows.  With a current GL context (a viewer embeds this sequence, src/main.cpp:71-115) it renders one
// frame into the PBO and shows it; without one — every box this repository has been built on — it says so and exits 0.
// tests/test_gl_glue.py builds and runs it: what it proves is that HIPOutputBuffer's GL form and GLDisplay compile against
// the installed GL headers and resolve every GL / HIP-interop symbol they use.
#include <GL/glx.h>

#include <cstdio>
#include <exception>

#include "../host/Display.h"
#include "../host/GaussianTracer.h"

int main(int argc, char** argv)
{
    if (!glXGetCurrentContext()) {
        std::puts("grt_gl_link_check: no current GL context (no display on this machine): HIPOutputBuffer [GL interop] and GLDisplay are linked, not run");
        return 0;
    }
    try { // (reached only when a host application made a context current before calling in)
        const int w = 1280, h = 720;
        GaussianTracer tracer(argc > 1 ? argv[1] : "../data/train.ply");
        tracer.setSize(w, h);
        tracer.initializeOptix();
        CUDAOutputBuffer output_buffer(w, h);   // GL interop: glGenBuffers + hipGraphicsGLRegisterBuffer
        output_buffer.setStream(tracer.stream);
        GLDisplay gldisplay;
        tracer.render(output_buffer);            // map -> kernel -> unmap, on the tracer's stream
        gldisplay.display(w, h, w, h, output_buffer.getPBO());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "grt_gl_link_check: %s\n", e.what());
        return 1;
    }
    return 0;
}
