// HIPOutputBuffer.h — stands where the reference's CUDAOutputBuffer (a GL pixel-buffer object registered with CUDA,
// src/CUDAOutputBuffer.h:12-39, .cpp:24-64) stands: map() hands the renderer a device uchar3 frame on the tracer's
// stream.  There is no GL interop on a display-less MI355X box, so the buffer is a plain device allocation plus a
// PINNED host mirror: unmap() enqueues the device -> host copy on the same stream, and after render()'s stream
// synchronisation (src/GaussianTracer.cpp:537) getHostPointer() is the frame the display uploads (Display.h).
// `CUDAOutputBuffer` is an alias, so main.cpp:77-78,100 and GaussianTracer::render(CUDAOutputBuffer&) read as before.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "VecMath.h"

class HIPOutputBuffer
{
public:
    HIPOutputBuffer(int32_t width, int32_t height);
    ~HIPOutputBuffer();
    HIPOutputBuffer(const HIPOutputBuffer&) = delete;
    HIPOutputBuffer& operator=(const HIPOutputBuffer&) = delete;

    void setStream(void* stream) { m_stream = stream; } // hipStream_t (CUstream in the reference)
    void resize(int32_t width, int32_t height);

    uchar3* map() { return m_device; } // device pointer, row-major y*width+x, row 0 = bottom of the window
    void unmap();                      // enqueues the readback into the pinned mirror on the stream

    int32_t width() const { return m_width; }
    int32_t height() const { return m_height; }

    unsigned int getPBO() const { return 0u; } // no GL object: GLDisplay::display uploads from getHostPointer()
    // the pinned host mirror (valid once the stream has been synchronised after unmap(), as render() does)
    const uchar3* getHostPointer() const { return m_host; }
    const std::vector<unsigned char>& download(); // synchronous copy of the device frame, RGB8, same layout

private:
    int32_t m_width = 0, m_height = 0;
    uchar3* m_device = nullptr;
    uchar3* m_host = nullptr; // hipHostMalloc
    void* m_stream = nullptr;
    std::vector<unsigned char> m_copy;
};

using CUDAOutputBuffer = HIPOutputBuffer;
