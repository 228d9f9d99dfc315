// DisplayGL.cpp — GLDisplay::display() (src/Display.cpp:160-208) for builds with -DGRT_WITH_GL.  Written fresh: the quad needs
// no vertex buffer (its corners come from gl_VertexID), the texture is filled from the bound pixel-buffer object — or from
// the pinned host mirror when the output buffer has none — and rows are unpacked with alignment 1 (3-byte pixels: the
// reference's alignment 4 is only right for widths that are multiples of 4).
#ifndef GRT_WITH_GL
#error "DisplayGL.cpp is a -DGRT_WITH_GL translation unit"
#endif
#define GL_GLEXT_PROTOTYPES 1
#include <GL/gl.h>
#include <GL/glext.h>

#include <stdexcept>
#include <string>

#include "Display.h"

namespace {
const char* kVertex = R"(#version 330 core
out vec2 uv;
void main()
{
    vec2 p = vec2(float((gl_VertexID & 1) << 2) - 1.0, float((gl_VertexID & 2) << 1) - 1.0); // one triangle that covers the viewport
    uv = (p + 1.0) * 0.5;                                                                      // buffer row 0 = bottom of the window
    gl_Position = vec4(p, 0.0, 1.0);
}
)";
const char* kFragment = R"(#version 330 core
in vec2 uv;
out vec3 color;
uniform sampler2D frame;
void main() { color = texture(frame, uv).rgb; }
)";

unsigned int compile(GLenum type, const char* src)
{
    const unsigned int s = glCreateShader(type);
    glShaderSource(s, 1, &src, nullptr);
    glCompileShader(s);
    GLint ok = 0;
    glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[1024] = {0};
        glGetShaderInfoLog(s, sizeof(log) - 1, nullptr, log);
        glDeleteShader(s);
        throw std::runtime_error(std::string("GLDisplay: shader: ") + log);
    }
    return s;
}
void glOk(const char* what)
{
    const GLenum e = glGetError();
    if (e != GL_NO_ERROR) throw std::runtime_error(std::string("GLDisplay: ") + what + ": GL error " + std::to_string((unsigned)e));
}
}

GLDisplay::GLDisplay()
{
    const unsigned int vs = compile(GL_VERTEX_SHADER, kVertex), fs = compile(GL_FRAGMENT_SHADER, kFragment);
    m_program = glCreateProgram();
    glAttachShader(m_program, vs);
    glAttachShader(m_program, fs);
    glLinkProgram(m_program);
    GLint ok = 0;
    glGetProgramiv(m_program, GL_LINK_STATUS, &ok);
    glDeleteShader(vs);
    glDeleteShader(fs);
    if (!ok) throw std::runtime_error("GLDisplay: the display program does not link");
    m_sampler_loc = glGetUniformLocation(m_program, "frame");
    glGenVertexArrays(1, &m_vao);
    glGenTextures(1, &m_texture);
    glBindTexture(GL_TEXTURE_2D, m_texture);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glOk("set-up");
}

GLDisplay::~GLDisplay()
{
    if (m_texture) glDeleteTextures(1, &m_texture);
    if (m_vao) glDeleteVertexArrays(1, &m_vao);
    if (m_program) glDeleteProgram(m_program);
}

void GLDisplay::display(int32_t screen_res_x, int32_t screen_res_y, int32_t framebuf_res_x, int32_t framebuf_res_y, uint32_t pbo,
                        const void* host_pixels) const
{
    if (!pbo && !host_pixels) throw std::runtime_error("GLDisplay::display: neither a pixel-buffer object nor host pixels");
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    glViewport(0, 0, framebuf_res_x, framebuf_res_y);
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    glUseProgram(m_program);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, m_texture);
    glBindBuffer(GL_PIXEL_UNPACK_BUFFER, pbo); // 0: the pointer below is host memory; else an offset into the object
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB8, screen_res_x, screen_res_y, 0, GL_RGB, GL_UNSIGNED_BYTE, pbo ? nullptr : host_pixels);
    glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
    glUniform1i(m_sampler_loc, 0);
    glBindVertexArray(m_vao);
    glDrawArrays(GL_TRIANGLES, 0, 3);
    glBindVertexArray(0);
    glDisable(GL_FRAMEBUFFER_SRGB);
    glOk("display");
}
