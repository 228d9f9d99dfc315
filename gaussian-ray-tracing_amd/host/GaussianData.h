// GaussianData.h — PLY loader + activations (reference src/GaussianData.h:12-41, .cpp:3-151).
// happly is replaced by the library's own reader (grt_host_ply_*); properties are still looked up by name.
#pragma once
#include <string>
#include <vector>

#include "VecMath.h"

struct GaussianParticle
{
    float3 position;
    float3 scale;
    float  rotation[4]; // w, x, y, z (glm::quat(w,x,y,z) in the reference)
    float  opacity;
    float3 sh[16];
};

class GaussianData
{
public:
    explicit GaussianData(const std::string& filename);
    ~GaussianData() = default;

    size_t getVertexCount() const { return particles.size(); }
    float3 getCenter();

    std::vector<GaussianParticle> particles;

private:
    std::string m_filename;
};
