// Display.h — the reference's GLDisplay (src/Display.h:6-29, Display.cpp:160-208): the frame in the output buffer goes to a
// GL_RGB8 texture and a full-screen quad shows it; the quad's UV = (position + 1) / 2 (Display.cpp:13) puts texture row 0 —
// buffer row 0 — at the BOTTOM of the window.
//   * builds with -DGRT_WITH_GL (DisplayGL.cpp, `make gl`): display() as in the reference — with a pixel-buffer object
//     (HIPOutputBuffer in its GL-interop form) the texture is filled from it, GPU to GPU; with pbo == 0 (the readback form)
//     from the pinned host mirror passed as `host_pixels`.  Needs a current GL 3.3 context.
//   * every build: windowImage(), the same mapping in software — the pixels the window shows, top row first — used by the
//     headless CLI to write image files and by the tests.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "HIPOutputBuffer.h"

class GLDisplay
{
public:
#ifdef GRT_WITH_GL
    GLDisplay();  // compiles the two shaders, makes the texture and the (attribute-less) quad; needs a current context
    ~GLDisplay();
    GLDisplay(const GLDisplay&) = delete;
    GLDisplay& operator=(const GLDisplay&) = delete;
    // src/Display.cpp:160-208 (same arguments; host_pixels: the frame when there is no PBO)
    void display(int32_t screen_res_x, int32_t screen_res_y, int32_t framebuf_res_x, int32_t framebuf_res_y, uint32_t pbo,
                 const void* host_pixels = nullptr) const;
#endif
    // top-down RGB8 image of what display() would put on screen from this buffer (call after render())
    static std::vector<unsigned char> windowImage(const HIPOutputBuffer& buf)
    {
        const size_t w = (size_t)buf.width(), h = (size_t)buf.height();
        std::vector<unsigned char> img(w * h * 3);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(buf.getHostPointer());
        for (size_t y = 0; y < h; y++) std::memcpy(img.data() + y * w * 3, src + (h - 1 - y) * w * 3, w * 3);
        return img;
    }

private:
#ifdef GRT_WITH_GL
    unsigned int m_program = 0u, m_texture = 0u, m_vao = 0u;
    int m_sampler_loc = -1;
#endif
};
