#include "Primitives.h"

#include <fstream>
#include <sstream>
#include <stdexcept>

// src/geometry/Primitives.cpp:6-61 — 0.3 x 0.5 quad in the z = 0 plane, normal +z, 2 triangles
Primitive Primitives::createPlane(float3 position)
{
    Primitive p;
    const unsigned int tessU = 1, tessV = 1;
    const float width = 0.3f, height = 0.5f;
    const float uTile = width / float(tessU), vTile = height / float(tessV);
    const float3 corner = make_float3(-width * 0.5f, -height * 0.5f, 0.0f);
    const float3 normal = make_float3(0.0f, 0.0f, 1.0f);
    for (unsigned int j = 0; j <= tessV; ++j) {
        const float v = float(j) * vTile;
        for (unsigned int i = 0; i <= tessU; ++i) {
            const float u = float(i) * uTile;
            p.vertices.push_back(corner + make_float3(u, v, 0.0f));
            p.normals.push_back(normal);
        }
    }
    const unsigned int stride = tessU + 1;
    for (unsigned int j = 0; j < tessV; ++j)
        for (unsigned int i = 0; i < tessU; ++i) {
            p.indices.push_back(j * stride + i);
            p.indices.push_back(j * stride + i + 1);
            p.indices.push_back((j + 1) * stride + i + 1);
            p.indices.push_back((j + 1) * stride + i + 1);
            p.indices.push_back((j + 1) * stride + i);
            p.indices.push_back(j * stride + i);
        }
    p.index = numberOfPlane++;
    p.type = "Plane";
    p.instanceIndex = numberOfMesh++;
    p.vertex_count = p.vertices.size();
    p.transform = getInitialTransform(position);
    m_primitives.push_back(p);
    return p;
}

// src/geometry/Primitives.cpp:63-140 — UV sphere, radius 0.3, 180 x 90, south pole first
Primitive Primitives::createSphere(float3 position)
{
    Primitive p;
    const unsigned int tessU = 180, tessV = 90;
    const float radius = 0.3f, maxTheta = M_PIf;
    p.vertices.reserve((tessU + 1) * tessV);
    p.indices.reserve(6 * tessU * (tessV - 1));
    const float phi_step = 2.0f * M_PIf / (float)tessU;
    const float theta_step = maxTheta / (float)(tessV - 1);
    for (unsigned int latitude = 0; latitude < tessV; ++latitude) {
        const float theta = (float)latitude * theta_step;
        const float sinTheta = sinf(theta), cosTheta = cosf(theta);
        for (unsigned int longitude = 0; longitude <= tessU; ++longitude) {
            const float phi = (float)longitude * phi_step;
            const float sinPhi = sinf(phi), cosPhi = cosf(phi);
            const float3 normal = make_float3(cosPhi * sinTheta, cosTheta, sinPhi * sinTheta);
            p.vertices.push_back(normal * radius);
            p.normals.push_back(normal);
        }
    }
    const unsigned int columns = tessU + 1;
    for (unsigned int latitude = 0; latitude < tessV - 1; ++latitude)
        for (unsigned int longitude = 0; longitude < tessU; ++longitude) {
            p.indices.push_back(latitude * columns + longitude);
            p.indices.push_back(latitude * columns + longitude + 1);
            p.indices.push_back((latitude + 1) * columns + longitude + 1);
            p.indices.push_back((latitude + 1) * columns + longitude + 1);
            p.indices.push_back((latitude + 1) * columns + longitude);
            p.indices.push_back(latitude * columns + longitude);
        }
    p.index = numberOfSphere++;
    p.type = "Sphere";
    p.instanceIndex = numberOfMesh++;
    p.vertex_count = p.vertices.size();
    p.transform = getInitialTransform(position);
    m_primitives.push_back(p);
    return p;
}

// src/geometry/Primitives.cpp:142-202.  One (position, normal) pair is emitted per face corner, in file
// order, exactly as the reference un-indexes tinyobj's output; positions and normals get the Y flip
// (:175,:179).  Polygons are fan-triangulated (tinyobj's default).  A face corner without a normal index —
// which the reference dereferences at index -1 — is an error here; a parse failure throws instead of exit(1).
Primitive Primitives::createLoadMesh(std::string filename, float3 position)
{
    Primitive p;
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("OBJ: cannot open " + filename);
    std::vector<float3> vs, ns;
    std::string line;
    size_t lineno = 0;
    auto resolve = [&](long idx, size_t n, const char* what) -> size_t {
        if (idx > 0 && (size_t)idx <= n) return (size_t)idx - 1;
        if (idx < 0 && (size_t)(-idx) <= n) return n - (size_t)(-idx);
        throw std::runtime_error("OBJ: bad " + std::string(what) + " index at line " + std::to_string(lineno) + " of " + filename);
    };
    while (std::getline(f, line)) {
        lineno++;
        std::istringstream ss(line);
        std::string tok;
        if (!(ss >> tok)) continue;
        if (tok == "v") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) throw std::runtime_error("OBJ: bad vertex at line " + std::to_string(lineno));
            vs.push_back(make_float3(x, y, z));
        } else if (tok == "vn") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) throw std::runtime_error("OBJ: bad normal at line " + std::to_string(lineno));
            ns.push_back(make_float3(x, y, z));
        } else if (tok == "f") {
            std::vector<std::pair<size_t, size_t>> corners;
            std::string c;
            while (ss >> c) {
                long vi = 0, ni = 0;
                const size_t s1 = c.find('/');
                vi = std::stol(c.substr(0, s1));
                if (s1 == std::string::npos) throw std::runtime_error("OBJ: face corner without a normal at line " + std::to_string(lineno));
                const size_t s2 = c.find('/', s1 + 1);
                if (s2 == std::string::npos || s2 + 1 >= c.size())
                    throw std::runtime_error("OBJ: face corner without a normal at line " + std::to_string(lineno));
                ni = std::stol(c.substr(s2 + 1));
                corners.push_back({resolve(vi, vs.size(), "vertex"), resolve(ni, ns.size(), "normal")});
            }
            if (corners.size() < 3) throw std::runtime_error("OBJ: face with fewer than 3 corners at line " + std::to_string(lineno));
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const size_t tri[3] = {0, k, k + 1};
                for (size_t t : tri) {
                    const float3 v = vs[corners[t].first], n = ns[corners[t].second];
                    p.indices.push_back((unsigned int)p.vertices.size());
                    p.vertices.push_back(make_float3(v.x, -v.y, v.z));
                    p.normals.push_back(make_float3(n.x, -n.y, n.z));
                }
            }
        }
    }
    if (p.indices.empty()) throw std::runtime_error("OBJ: no faces in " + filename);
    p.index = numberOfLoaded++;
    p.type = "LoadedMesh";
    p.instanceIndex = numberOfMesh++;
    p.vertex_count = p.vertices.size();
    p.transform = getInitialTransform(position);
    m_primitives.push_back(p);
    return p;
}
