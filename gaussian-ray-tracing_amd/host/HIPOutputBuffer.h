// HIPOutputBuffer.h — stands where the reference's CUDAOutputBuffer (GL PBO registered with CUDA,
// src/CUDAOutputBuffer.h:12-39) stands: map() hands the renderer a device uchar3 frame.  On a display-less
// MI355X it is a plain device allocation plus a pinned-host readback (viewer glue).
#pragma once
#include <cstddef>
#include <vector>

#include "VecMath.h"

class HIPOutputBuffer
{
public:
    HIPOutputBuffer(unsigned int width, unsigned int height);
    ~HIPOutputBuffer();
    HIPOutputBuffer(const HIPOutputBuffer&) = delete;
    HIPOutputBuffer& operator=(const HIPOutputBuffer&) = delete;

    void resize(unsigned int width, unsigned int height);
    uchar3* map() { return m_device; }  // device pointer, row-major y*width+x, row 0 = bottom of the window
    void unmap() {}
    void setStream(void* stream) { m_stream = stream; }
    unsigned int width() const { return m_width; }
    unsigned int height() const { return m_height; }
    const std::vector<unsigned char>& download(); // RGB8, same layout

private:
    unsigned int m_width = 0, m_height = 0;
    uchar3* m_device = nullptr;
    void* m_stream = nullptr;
    std::vector<unsigned char> m_host;
};
