// Display.h — what is left of the reference's GLDisplay (src/Display.h, Display.cpp:160-208) without a GL context.
// GLDisplay::display() binds the output buffer's PBO and uploads it with glTexImage2D(..., nullptr); the textured
// quad's UV = (position + 1) / 2 (Display.cpp:13) puts texture row 0 — buffer row 0 — at the BOTTOM of the window.
// With HIPOutputBuffer there is no PBO; the upload takes the pinned host mirror instead:
//
//     glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
//     glPixelStorei(GL_UNPACK_ALIGNMENT, 1);   // rows of 3-byte pixels are not 4-byte aligned for odd widths
//     glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB8, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE, output_buffer.getHostPointer());
//
// (everything else of Display.cpp stays).  windowImage() is the same mapping in software: the pixels the window
// shows, top row first — used by the headless CLI to write image files and by the tests.
#pragma once
#include <cstring>
#include <vector>

#include "HIPOutputBuffer.h"

struct GLDisplay
{
    // top-down RGB8 image of what display() would put on screen from this buffer (call after render())
    static std::vector<unsigned char> windowImage(const HIPOutputBuffer& buf)
    {
        const size_t w = (size_t)buf.width(), h = (size_t)buf.height();
        std::vector<unsigned char> img(w * h * 3);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(buf.getHostPointer());
        for (size_t y = 0; y < h; y++) std::memcpy(img.data() + y * w * 3, src + (h - 1 - y) * w * 3, w * 3);
        return img;
    }
};
