// HIPOutputBufferGL.cpp — the GL-interop form of HIPOutputBuffer (compiled with -DGRT_WITH_GL only; `make gl`).
// Replaces src/CUDAOutputBuffer.cpp:24-64: the frame lives in a GL pixel-buffer object that HIP maps for the renderer, so a
// frame never crosses PCIe on its way to the window.  HIP's interop entry points are the CUDA ones renamed
// (hip/hip_gl_interop.h); they need a current GL context and a GPU that drives it (hipGLGetDevices).  On a display-less
// box this file is compiled and linked (tests/test_gl_glue.py) and never run.
#ifndef GRT_WITH_GL
#error "HIPOutputBufferGL.cpp is the -DGRT_WITH_GL translation unit"
#endif
#define GL_GLEXT_PROTOTYPES 1
#include <GL/gl.h>
#include <GL/glext.h>
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include <stdexcept>
#include <string>

#include "HIPOutputBuffer.h"

namespace {
void hipOk(hipError_t e, const char* what) // (src/Exception.h:31-80: every failure throws)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("HIPOutputBuffer (GL): ") + what + ": " + hipGetErrorString(e));
}
void glOk(const char* what)
{
    const GLenum e = glGetError();
    if (e != GL_NO_ERROR) throw std::runtime_error(std::string("HIPOutputBuffer (GL): ") + what + ": GL error " + std::to_string((unsigned)e));
}
}

void HIPOutputBuffer::releaseGL()
{
    if (m_gfx) (void)hipGraphicsUnregisterResource(static_cast<hipGraphicsResource_t>(m_gfx));
    m_gfx = nullptr;
    if (m_pbo) glDeleteBuffers(1, &m_pbo);
    m_pbo = 0u;
}

void HIPOutputBuffer::resizeGL(int32_t width, int32_t height)
{
    releaseGL(); // (the reference generates a new buffer per resize and leaves the old one registered: not replicated)
    m_width = width; m_height = height;
    glGenBuffers(1, &m_pbo);
    glBindBuffer(GL_ARRAY_BUFFER, m_pbo);
    glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)((size_t)width * height * 3), nullptr, GL_STREAM_DRAW);
    glBindBuffer(GL_ARRAY_BUFFER, 0u);
    glOk("pixel-buffer object");
    hipGraphicsResource_t res = nullptr;
    // (flags None, not WriteDiscard as src/CUDAOutputBuffer.cpp:38-43 registers its PBO: download() maps the buffer to READ the frame
    //  back, and what a write-discard resource holds after a map is undefined)
    hipOk(hipGraphicsGLRegisterBuffer(&res, m_pbo, hipGraphicsRegisterFlagsNone), "hipGraphicsGLRegisterBuffer");
    m_gfx = res;
}

uchar3* HIPOutputBuffer::mapGL()
{
    hipGraphicsResource_t res = static_cast<hipGraphicsResource_t>(m_gfx);
    hipOk(hipGraphicsMapResources(1, &res, static_cast<hipStream_t>(m_stream)), "hipGraphicsMapResources");
    void* p = nullptr;
    size_t bytes = 0;
    hipOk(hipGraphicsResourceGetMappedPointer(&p, &bytes, res), "hipGraphicsResourceGetMappedPointer");
    if (bytes < (size_t)m_width * m_height * 3) throw std::runtime_error("HIPOutputBuffer (GL): the mapped pixel-buffer object is smaller than the frame");
    return static_cast<uchar3*>(p);
}

void HIPOutputBuffer::unmapGL()
{
    hipGraphicsResource_t res = static_cast<hipGraphicsResource_t>(m_gfx);
    hipOk(hipGraphicsUnmapResources(1, &res, static_cast<hipStream_t>(m_stream)), "hipGraphicsUnmapResources");
}
