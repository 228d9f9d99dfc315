/*
 * ref_glue.cpp — TEST INFRASTRUCTURE ONLY (oracle/_ref).
 *
 * Thin extern "C" glue over the parts of the reference that compile in this container from their
 * own sources, in place under /root/reference (nothing is copied, no stand-in headers are written):
 *   src/Camera.cpp + src/Camera.h        -> Camera::UVWFrame
 *   src/geometry/Icosahedron.h           -> proxy vertices / indices
 *   src/vector_math.h                    -> normalize / reflect / cross / length / clamp
 *   third_party/glm                      -> quat ctor, mat3_cast, mat4_cast, scale, translate, mat*vec
 * The CUDA vector-type headers these include (vector_types.h, vector_functions.h, cuda_runtime.h) are
 * the ones that ship inside the Triton wheel of this image.  The OptiX-dependent files (Parameters.h,
 * tracer.cu/.cuh, GaussianTracer.cpp) and the happly/tinyobj-dependent ones are NOT buildable here.
 * Built by oracle/Makefile into oracle/_ref/libgrt_ref.so (git-ignored); used by tests to pin
 * oracle/grt_oracle.c bit-for-bit on these functions.
 */
#include "Camera.h"
#include "geometry/Icosahedron.h"

#include <glm.hpp>
#include <gtc/matrix_transform.hpp>
#include <gtc/quaternion.hpp>

#include <cstring>

extern "C" {

void ref_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovy, float aspect,
                   float U[3], float V[3], float W[3])
{
    Camera cam;
    cam.setEye(make_float3(eye[0], eye[1], eye[2]));
    cam.setLookat(make_float3(lookat[0], lookat[1], lookat[2]));
    cam.setUp(make_float3(up[0], up[1], up[2]));
    cam.setFovY(fovy);
    cam.setAspectRatio(aspect);
    float3 u, v, w;
    cam.UVWFrame(u, v, w);
    U[0] = u.x; U[1] = u.y; U[2] = u.z;
    V[0] = v.x; V[1] = v.y; V[2] = v.z;
    W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

void ref_icosahedron(float verts[36], unsigned int idx[60])
{
    Icosahedron ico;
    std::vector<float3> v = ico.getVertices();
    std::vector<unsigned int> i = ico.getIndices();
    for (int k = 0; k < 12; k++) { verts[k * 3] = v[k].x; verts[k * 3 + 1] = v[k].y; verts[k * 3 + 2] = v[k].z; }
    for (int k = 0; k < 60; k++) idx[k] = i[k];
}

/* glm::mat3_cast of glm::quat(w,x,y,z); output column-major m[c*3+r] */
void ref_mat3_cast(const float q[4], float out[9])
{
    glm::quat g(q[0], q[1], q[2], q[3]);
    glm::mat3 R = glm::mat3_cast(g);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out[c * 3 + r] = R[c][r];
}

/* diag(1/scale) * transpose(mat3_cast(q)) through glm's own mat3 product; output row-major math A[r*3+c] */
void ref_inv_cov(const float scale[3], const float q[4], float out[9])
{
    glm::mat3 Rt = glm::transpose(glm::mat3_cast(glm::quat(q[0], q[1], q[2], q[3])));
    glm::mat3 D(1.0f);
    for (int k = 0; k < 3; k++) D[k][k] = 1.0f / scale[k];
    glm::mat3 M = D * Rt;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out[r * 3 + c] = M[c][r];
}

/* glm mat3 * vec3 with a row-major math matrix as input */
void ref_mat3_vec(const float A[9], const float v[3], float out[3])
{
    glm::mat3 M;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) M[c][r] = A[r * 3 + c];
    glm::vec3 o = M * glm::vec3(v[0], v[1], v[2]);
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
}

float ref_glm_dot(const float a[3], const float b[3])
{
    return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

/* instance matrix translate * (mat4_cast(q) * scale(scale*s)) applied to a proxy vertex */
void ref_instance_vertex(const float pos[3], const float scale[3], const float q[4], float s, const float v[3],
                         float out[3])
{
    glm::vec3 sc(scale[0], scale[1], scale[2]);
    glm::mat4 S = glm::scale(glm::mat4(1.0f), sc * s);
    glm::mat4 R = glm::mat4_cast(glm::quat(q[0], q[1], q[2], q[3]));
    glm::mat4 T = glm::translate(glm::mat4(1.0f), glm::vec3(pos[0], pos[1], pos[2]));
    glm::mat4 M = T * (R * S);
    glm::vec4 w = M * glm::vec4(v[0], v[1], v[2], 1.0f);
    out[0] = w.x; out[1] = w.y; out[2] = w.z;
}

void ref_normalize(const float v[3], float out[3])
{
    float3 r = normalize(make_float3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_reflect(const float i[3], const float n[3], float out[3])
{
    float3 r = reflect(make_float3(i[0], i[1], i[2]), make_float3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_cross(const float a[3], const float b[3], float out[3])
{
    float3 r = cross(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_length(const float v[3]) { return length(make_float3(v[0], v[1], v[2])); }
float ref_dot(const float a[3], const float b[3]) { return dot(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2])); }
float ref_clamp(float f, float a, float b) { return clamp(f, a, b); }

/* mesh placement: translate(position) with identity rotation/scale, applied to a vertex and, through
 * mat3(transform), to a normal */
void ref_mesh_vertex(const float position[3], const float v[3], const float n[3], float vout[3], float nout[3])
{
    glm::mat4 t = glm::translate(glm::mat4(1.0f), glm::vec3(position[0], position[1], position[2]));
    glm::vec4 w = t * glm::vec4(v[0], v[1], v[2], 1.0f);
    glm::vec3 nn = glm::mat3(t) * glm::vec3(n[0], n[1], n[2]);
    vout[0] = w.x; vout[1] = w.y; vout[2] = w.z;
    nout[0] = nn.x; nout[1] = nn.y; nout[2] = nn.z;
}

} /* extern "C" */
