#include "GaussianData.h"

#include <iostream>
#include <stdexcept>

#include "../../include/grt.h"

GaussianData::GaussianData(const std::string& filename) : m_filename(filename)
{
    std::cout << "Loading Gaussian data from " << filename << "\n"; // src/GaussianData.cpp:5
    uint64_t n = 0;
    if (grt_host_ply_count(filename.c_str(), &n) != GRT_OK) throw std::runtime_error(grt_host_last_error());
    std::vector<float> pos(n * 3), dc(n * 3), rest(n * 45), op(n), sc(n * 3), rot(n * 4);
    if (grt_host_ply_read(filename.c_str(), n, pos.data(), dc.data(), rest.data(), op.data(), sc.data(), rot.data()) != GRT_OK)
        throw std::runtime_error(grt_host_last_error());
    std::cout << "Number of vertices: " << n << std::endl; // src/GaussianData.cpp:95
    std::vector<float> apos(n * 3), ascale(n * 3), aquat(n * 4), aop(n), ash(n * 48);
    if (grt_host_activate(n, pos.data(), dc.data(), rest.data(), op.data(), sc.data(), rot.data(), apos.data(),
                          ascale.data(), aquat.data(), aop.data(), ash.data()) != GRT_OK)
        throw std::runtime_error(grt_host_last_error());
    particles.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        GaussianParticle& p = particles[i];
        p.position = make_float3(apos[i * 3], apos[i * 3 + 1], apos[i * 3 + 2]);
        p.scale = make_float3(ascale[i * 3], ascale[i * 3 + 1], ascale[i * 3 + 2]);
        for (int k = 0; k < 4; k++) p.rotation[k] = aquat[i * 4 + k];
        p.opacity = aop[i];
        for (int k = 0; k < 16; k++) p.sh[k] = make_float3(ash[i * 48 + k * 3], ash[i * 48 + k * 3 + 1], ash[i * 48 + k * 3 + 2]);
    }
}

// src/GaussianData.cpp:139-151
float3 GaussianData::getCenter()
{
    float3 center = make_float3(0.0f, 0.0f, 0.0f);
    for (auto& p : particles) {
        center.x += p.position.x;
        center.y += p.position.y;
        center.z += p.position.z;
    }
    center.x /= static_cast<float>(particles.size());
    center.y /= static_cast<float>(particles.size());
    center.z /= static_cast<float>(particles.size());
    return center;
}
