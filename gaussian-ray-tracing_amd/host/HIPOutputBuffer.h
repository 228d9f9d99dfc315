// HIPOutputBuffer.h — stands where the reference's CUDAOutputBuffer (a GL pixel-buffer object registered with CUDA,
// src/CUDAOutputBuffer.h:12-39, .cpp:24-64) stands: map() hands the renderer a device uchar3 frame on the tracer's
// stream.  Two forms, chosen at construction:
//   * READBACK (the default of the plain build, and what a display-less MI355X box runs): a device allocation plus a PINNED
//     host mirror; unmap() enqueues the device -> host copy on the same stream, and after render()'s stream synchronisation
//     (src/GaussianTracer.cpp:537) getHostPointer() is the frame the display uploads (Display.h).  getPBO() = 0.
//   * GL INTEROP (builds with -DGRT_WITH_GL, HIPOutputBufferGL.cpp; the reference's only form): a GL pixel-buffer object
//     registered with HIP — glGenBuffers / glBufferData, hipGraphicsGLRegisterBuffer (readable: download() maps it too); map() =
//     hipGraphicsMapResources + hipGraphicsResourceGetMappedPointer on the stream, unmap() = hipGraphicsUnmapResources;
//     getPBO() is the object GLDisplay::display() binds as GL_PIXEL_UNPACK_BUFFER.  Needs a current GL context on a GPU that
//     drives it (hipGLGetDevices); unlike the reference, resize() and the destructor unregister and delete the old object.
// `CUDAOutputBuffer` is an alias, so main.cpp:77-78,100 and GaussianTracer::render(CUDAOutputBuffer&) read as before.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "VecMath.h"

class HIPOutputBuffer
{
public:
#ifdef GRT_WITH_GL
    HIPOutputBuffer(int32_t width, int32_t height, bool gl_interop = true); // (needs a current GL context when gl_interop)
#else
    HIPOutputBuffer(int32_t width, int32_t height);
#endif
    ~HIPOutputBuffer();
    HIPOutputBuffer(const HIPOutputBuffer&) = delete;
    HIPOutputBuffer& operator=(const HIPOutputBuffer&) = delete;

    void setStream(void* stream) { m_stream = stream; } // hipStream_t (CUstream in the reference)
    void resize(int32_t width, int32_t height);

    uchar3* map();  // device pointer, row-major y*width+x, row 0 = bottom of the window
    void unmap();   // READBACK: enqueues the copy into the pinned mirror on the stream; GL: hands the PBO back to GL

    int32_t width() const { return m_width; }
    int32_t height() const { return m_height; }

    unsigned int getPBO() const { return m_pbo; } // 0 = no GL object: GLDisplay::display uploads from getHostPointer()
    // the pinned host mirror (valid once the stream has been synchronised after unmap(), as render() does)
    const uchar3* getHostPointer() const { return m_host; }
    const std::vector<unsigned char>& download(); // synchronous copy of the device frame, RGB8, same layout

private:
    int32_t m_width = 0, m_height = 0;
    uchar3* m_device = nullptr;
    uchar3* m_host = nullptr; // hipHostMalloc
    void* m_stream = nullptr;
    std::vector<unsigned char> m_copy;
    unsigned int m_pbo = 0u;  // GL INTEROP: the pixel-buffer object ...
    void* m_gfx = nullptr;    // ... and its hipGraphicsResource_t
    bool m_gl = false;
    void resizeGL(int32_t width, int32_t height); // HIPOutputBufferGL.cpp
    void releaseGL();
    uchar3* mapGL();
    void unmapGL();
};

using CUDAOutputBuffer = HIPOutputBuffer;
