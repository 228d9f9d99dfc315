// grt_host.cpp — host-only helpers exported through the C ABI (include/grt.h, grt_host_*):
// the parts of the reference's host layer that ctypes/C users need without the C++ facade.
//   grt_host_activate     GaussianData::parse activations       src/GaussianData.cpp:97-131
//   grt_host_uvw_frame    Camera::UVWFrame                      src/Camera.cpp:3-13
//   grt_host_ply_*        happly-based PLY load (by property name) src/GaussianData.cpp:20-92
//   grt_host_synth_scene  deterministic synthetic 3DGS scene    SURVEY.md §8(d)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/grt.h"

static thread_local std::string g_host_err;

extern "C" {

const char* grt_host_last_error(void) { return g_host_err.c_str(); }

int grt_host_activate(uint64_t n, const float* pos, const float* f_dc, const float* f_rest, const float* opacity_logit,
                      const float* log_scale, const float* rot, float* out_pos, float* out_scale, float* out_quat,
                      float* out_opacity, float* out_sh)
{
    if (n && (!pos || !f_dc || !f_rest || !opacity_logit || !log_scale || !rot || !out_pos || !out_scale || !out_quat ||
              !out_opacity || !out_sh)) {
        g_host_err = "grt_host_activate: null argument";
        return GRT_ERR_INVALID;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        out_pos[i * 3] = pos[i * 3]; out_pos[i * 3 + 1] = pos[i * 3 + 1]; out_pos[i * 3 + 2] = pos[i * 3 + 2];
        for (int k = 0; k < 3; k++) out_scale[i * 3 + k] = expf(log_scale[i * 3 + k]);          // :101-103
        const float* r = rot + i * 4;
        const float norm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);          // :104-107
        for (int k = 0; k < 4; k++) out_quat[i * 4 + k] = r[k] / norm;                            // (w,x,y,z) :108-111
        out_opacity[i] = 1.0f / (1.0f + expf(-opacity_logit[i]));                                 // :112
        float* sh = out_sh + i * 48;
        sh[0] = f_dc[i * 3]; sh[1] = f_dc[i * 3 + 1]; sh[2] = f_dc[i * 3 + 2];                    // :113
        const float* fr = f_rest + i * 45;
        for (int k = 1; k < 16; k++) {                                                            // :114-128
            sh[k * 3] = fr[k - 1];
            sh[k * 3 + 1] = fr[14 + k];
            sh[k * 3 + 2] = fr[29 + k];
        }
    }
    return GRT_OK;
}

void grt_host_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovy_deg, float aspect,
                        float U[3], float V[3], float W[3])
{
    float w[3] = {lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]};
    const float wlen = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float u[3] = {w[1] * up[2] - w[2] * up[1], w[2] * up[0] - w[0] * up[2], w[0] * up[1] - w[1] * up[0]};
    float inv = 1.0f / sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    u[0] *= inv; u[1] *= inv; u[2] *= inv;
    float v[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
    inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
    const float vlen = wlen * tanf(0.5f * fovy_deg * 3.14159265358979323846f / 180.0f);
    const float ulen = vlen * aspect;
    for (int k = 0; k < 3; k++) { V[k] = v[k] * vlen; U[k] = u[k] * ulen; W[k] = w[k]; }
}

// ---- synthetic scene: SplitMix64 -> U(0,1) -> Box-Muller (fp64 generation, rounded to fp32) ----
static inline uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t& s) { return ((double)(splitmix64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static inline double gauss(uint64_t& s)
{
    const double a = u01(s), b = u01(s);
    return sqrt(-2.0 * log(a)) * cos(6.283185307179586476925 * b);
}

int grt_host_synth_scene(uint64_t seed, uint64_t n, float* pos, float* f_dc, float* f_rest, float* opacity_logit,
                         float* log_scale, float* rot)
{
    if (n && (!pos || !f_dc || !f_rest || !opacity_logit || !log_scale || !rot)) {
        g_host_err = "grt_host_synth_scene: null argument";
        return GRT_ERR_INVALID;
    }
    double centres[64][3];
    {
        uint64_t s = seed * 0xD1342543DE82EF95ull + 0xC1057E25ull;
        for (int c = 0; c < 64; c++)
            for (int k = 0; k < 3; k++) centres[c][k] = 2.0 * u01(s) - 1.0;
    }
    const double mean_ls = log(0.7 * pow((double)(n ? n : 1), -1.0 / 3.0));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        uint64_t s = (seed + 1) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)i * 0xD6E8FEB86659FD93ull);
        splitmix64(s);
        if (u01(s) < 0.7) {
            const int c = (int)(u01(s) * 64.0) & 63;
            for (int k = 0; k < 3; k++) pos[i * 3 + k] = (float)(centres[c][k] + 0.08 * gauss(s));
        } else {
            for (int k = 0; k < 3; k++) pos[i * 3 + k] = (float)(3.0 * u01(s) - 1.5);
        }
        for (int k = 0; k < 3; k++) log_scale[i * 3 + k] = (float)(mean_ls + 0.5 * gauss(s));
        for (int k = 0; k < 4; k++) rot[i * 4 + k] = (float)gauss(s);
        opacity_logit[i] = (float)(1.0 + 2.0 * gauss(s));
        for (int k = 0; k < 3; k++) f_dc[i * 3 + k] = (float)(3.0 * u01(s) - 1.5);
        for (int k = 0; k < 45; k++) f_rest[i * 45 + k] = (float)(0.1 * gauss(s));
    }
    return GRT_OK;
}

// ---- PLY ----
namespace {
struct PlyProp { std::string name; std::string type; size_t size; size_t offset; };
struct PlyHeader {
    bool ascii = false, binary_le = false;
    uint64_t count = 0;
    std::vector<PlyProp> props;
    size_t stride = 0;
    std::streampos data_pos = 0;
    size_t elems_before_bytes = 0; // only vertex-first files are supported
};

size_t ply_type_size(const std::string& t)
{
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}

int ply_parse_header(std::ifstream& f, PlyHeader& h)
{
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") { g_host_err = "not a PLY file"; return GRT_ERR_IO; }
    bool in_vertex = false, seen_vertex = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") {
            std::string fmt;
            ss >> fmt;
            h.ascii = fmt == "ascii";
            h.binary_le = fmt == "binary_little_endian";
            if (!h.ascii && !h.binary_le) { g_host_err = "unsupported PLY format " + fmt; return GRT_ERR_IO; }
        } else if (tok == "element") {
            std::string name;
            uint64_t cnt;
            ss >> name >> cnt;
            in_vertex = (name == "vertex");
            if (in_vertex) { h.count = cnt; seen_vertex = true; }
            else if (!seen_vertex && cnt) { g_host_err = "PLY: element '" + name + "' precedes 'vertex'"; return GRT_ERR_IO; }
        } else if (tok == "property" && in_vertex) {
            std::string type, name;
            ss >> type;
            if (type == "list") { g_host_err = "PLY: list property in vertex element"; return GRT_ERR_IO; }
            ss >> name;
            const size_t sz = ply_type_size(type);
            if (!sz) { g_host_err = "PLY: unknown property type " + type; return GRT_ERR_IO; }
            h.props.push_back({name, type, sz, h.stride});
            h.stride += sz;
        } else if (tok == "end_header") {
            h.data_pos = f.tellg();
            if (!seen_vertex) { g_host_err = "PLY: no vertex element"; return GRT_ERR_IO; }
            return GRT_OK;
        }
    }
    g_host_err = "PLY: truncated header";
    return GRT_ERR_IO;
}

void init_names(std::vector<std::string>& names)
{
    names = {"x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (int k = 0; k < 45; k++) names.push_back("f_rest_" + std::to_string(k));
    names.push_back("opacity");
    for (int k = 0; k < 3; k++) names.push_back("scale_" + std::to_string(k));
    for (int k = 0; k < 4; k++) names.push_back("rot_" + std::to_string(k));
}

inline float prop_as_float(const unsigned char* p, const PlyProp& pr)
{
    if (pr.size == 4 && (pr.type == "float" || pr.type == "float32")) { float v; memcpy(&v, p, 4); return v; }
    if (pr.size == 8) { double v; memcpy(&v, p, 8); return (float)v; }
    if (pr.size == 4 && (pr.type == "int" || pr.type == "int32")) { int32_t v; memcpy(&v, p, 4); return (float)v; }
    if (pr.size == 4) { uint32_t v; memcpy(&v, p, 4); return (float)v; }
    if (pr.size == 2 && (pr.type == "short" || pr.type == "int16")) { int16_t v; memcpy(&v, p, 2); return (float)v; }
    if (pr.size == 2) { uint16_t v; memcpy(&v, p, 2); return (float)v; }
    if (pr.type == "char" || pr.type == "int8") return (float)*(const int8_t*)p;
    return (float)*p;
}
} // namespace

int grt_host_ply_count(const char* path, uint64_t* n_out)
{
    if (!path || !n_out) return GRT_ERR_INVALID;
    std::ifstream f(path, std::ios::binary);
    if (!f) { g_host_err = std::string("cannot open ") + path; return GRT_ERR_IO; }
    PlyHeader h;
    int rc = ply_parse_header(f, h);
    if (rc != GRT_OK) return rc;
    *n_out = h.count;
    return GRT_OK;
}

int grt_host_ply_read(const char* path, uint64_t n, float* pos, float* f_dc, float* f_rest, float* opacity_logit,
                      float* log_scale, float* rot)
{
    if (!path) return GRT_ERR_INVALID;
    std::ifstream f(path, std::ios::binary);
    if (!f) { g_host_err = std::string("cannot open ") + path; return GRT_ERR_IO; }
    PlyHeader h;
    int rc = ply_parse_header(f, h);
    if (rc != GRT_OK) return rc;
    if (h.count != n) { g_host_err = "grt_host_ply_read: vertex count mismatch"; return GRT_ERR_INVALID; }
    // all 59 properties are required, looked up by name (the reference's getProperty throws otherwise)
    std::vector<std::string> names;
    init_names(names);
    std::vector<int> idx(names.size(), -1);
    for (size_t k = 0; k < names.size(); k++) {
        for (size_t j = 0; j < h.props.size(); j++)
            if (h.props[j].name == names[k]) idx[k] = (int)j;
        // Every property but f_rest_* is required, as the reference's getProperty() is.
        const bool is_rest = names[k].compare(0, 7, "f_rest_") == 0;
        if (idx[k] < 0 && !is_rest) { g_host_err = "PLY: missing property '" + names[k] + "'"; return GRT_ERR_IO; }
    }
    // Exports trained at SH degree L < 3 carry 3*K f_rest_* columns, K = (L+1)^2 - 1, CHANNEL-MAJOR with stride K
    // (f_rest_0..K-1 = red, K..2K-1 = green, 2K..3K-1 = blue).  The activation reads the
    // degree-3 layout (stride 15, src/GaussianData.cpp:113-128), so column c*K+j goes to slot c*15+j and the bands the
    // file does not have stay zero.  Any other column count is not a 3DGS export: rejected.
    size_t n_rest = 0;
    while (n_rest < 45 && idx[6 + n_rest] >= 0) n_rest++;
    for (size_t q = n_rest; q < 45; q++)
        if (idx[6 + q] >= 0) { g_host_err = "PLY: f_rest_* columns are not contiguous from f_rest_0"; return GRT_ERR_IO; }
    if (n_rest != 0 && n_rest != 9 && n_rest != 24 && n_rest != 45) {
        g_host_err = "PLY: " + std::to_string(n_rest) + " f_rest_* columns (expected 0, 9, 24 or 45)";
        return GRT_ERR_IO;
    }
    std::vector<int> rest_slot(45, -1); // destination slot of source column q
    {
        const size_t K = n_rest / 3;
        for (size_t q = 0; q < n_rest; q++) rest_slot[q] = (int)((q / K) * 15 + (q % K));
        // columns the file lacks still have to be zeroed: give each missing name one of the unused slots
        std::vector<bool> used(45, false);
        for (size_t q = 0; q < n_rest; q++) used[rest_slot[q]] = true;
        size_t next = 0;
        for (size_t q = n_rest; q < 45; q++) {
            while (used[next]) next++;
            rest_slot[q] = (int)next;
            used[next] = true;
        }
    }
    auto store = [&](uint64_t i, size_t k, float v) {
        if (k < 3) pos[i * 3 + k] = v;
        else if (k < 6) f_dc[i * 3 + (k - 3)] = v;
        else if (k < 51) f_rest[i * 45 + rest_slot[k - 6]] = v;
        else if (k == 51) opacity_logit[i] = v;
        else if (k < 55) log_scale[i * 3 + (k - 52)] = v;
        else rot[i * 4 + (k - 55)] = v;
    };
    f.seekg(h.data_pos);
    if (h.binary_le) {
        const size_t chunk = 4096;
        std::vector<unsigned char> buf(chunk * h.stride);
        for (uint64_t base = 0; base < n; base += chunk) {
            const uint64_t cnt = std::min<uint64_t>(chunk, n - base);
            f.read((char*)buf.data(), (std::streamsize)(cnt * h.stride));
            if ((uint64_t)f.gcount() != cnt * h.stride) { g_host_err = "PLY: truncated vertex data"; return GRT_ERR_IO; }
            for (uint64_t i = 0; i < cnt; i++)
                for (size_t k = 0; k < names.size(); k++) {
                    if (idx[k] < 0) { store(base + i, k, 0.0f); continue; }
                    const PlyProp& pr = h.props[idx[k]];
                    store(base + i, k, prop_as_float(buf.data() + i * h.stride + pr.offset, pr));
                }
        }
    } else {
        std::vector<double> row(h.props.size());
        for (uint64_t i = 0; i < n; i++) {
            for (size_t j = 0; j < h.props.size(); j++)
                if (!(f >> row[j])) { g_host_err = "PLY: truncated ascii vertex data"; return GRT_ERR_IO; }
            for (size_t k = 0; k < names.size(); k++) store(i, k, idx[k] < 0 ? 0.0f : (float)row[idx[k]]);
        }
    }
    return GRT_OK;
}

int grt_host_ply_write(const char* path, uint64_t n, const float* pos, const float* f_dc, const float* f_rest,
                       const float* opacity_logit, const float* log_scale, const float* rot)
{
    if (!path) return GRT_ERR_INVALID;
    FILE* fp = fopen(path, "wb");
    if (!fp) { g_host_err = std::string("cannot create ") + path; return GRT_ERR_IO; }
    // standard 3DGS property order: x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*
    fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %llu\n", (unsigned long long)n);
    const char* head[] = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (const char* s : head) fprintf(fp, "property float %s\n", s);
    for (int k = 0; k < 45; k++) fprintf(fp, "property float f_rest_%d\n", k);
    fprintf(fp, "property float opacity\n");
    for (int k = 0; k < 3; k++) fprintf(fp, "property float scale_%d\n", k);
    for (int k = 0; k < 4; k++) fprintf(fp, "property float rot_%d\n", k);
    fprintf(fp, "end_header\n");
    float row[62];
    for (uint64_t i = 0; i < n; i++) {
        memcpy(row, pos + i * 3, 12);
        row[3] = row[4] = row[5] = 0.0f;
        memcpy(row + 6, f_dc + i * 3, 12);
        memcpy(row + 9, f_rest + i * 45, 180);
        row[54] = opacity_logit[i];
        memcpy(row + 55, log_scale + i * 3, 12);
        memcpy(row + 58, rot + i * 4, 16);
        if (fwrite(row, sizeof(row), 1, fp) != 1) { fclose(fp); g_host_err = "PLY: short write"; return GRT_ERR_IO; }
    }
    fclose(fp);
    return GRT_OK;
}

// ---- procedural primitives and OBJ meshes (reference src/geometry/Primitives.cpp:6-216) ----
// Both procedural primitives are vertex LATTICES (rows x cols points) whose cells are cut into two triangles the same
// way, so one table describes them: lattice size, whether a column is repeated to close a seam, and a per-point
// evaluator.  The numbers a point evaluates to are the reference's (same float operations, glibc sinf/cosf):
//   plane   0.3 x 0.5 quad in z = 0, normal +z, 2 x 2 points                      Primitives.cpp:6-61
//   sphere  radius 0.3, 90 rings (south pole first) x 181 points (seam doubled)   Primitives.cpp:63-140
namespace {
struct LatticeSpec { uint32_t rows, cols; }; // points per column / per row
const LatticeSpec kLattice[2] = {{2u, 2u}, {90u, 181u}};

inline void lattice_point(int kind, uint32_t r, uint32_t c, float pos[3], float nrm[3])
{
    if (kind == GRT_PRIM_PLANE) {
        const float width = 0.3f, height = 0.5f;
        const float u = float(c) * (width / 1.0f), v = float(r) * (height / 1.0f);
        pos[0] = -width * 0.5f + u; pos[1] = -height * 0.5f + v; pos[2] = 0.0f + 0.0f;
        nrm[0] = 0.0f; nrm[1] = 0.0f; nrm[2] = 1.0f;
    } else {
        const float radius = 0.3f, pi = 3.14159265358979323846f;
        const float phi_step = 2.0f * pi / 180.0f, theta_step = pi / 89.0f;
        const float theta = (float)r * theta_step, phi = (float)c * phi_step;
        const float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
        nrm[0] = cp * st; nrm[1] = ct; nrm[2] = sp * st;
        for (int k = 0; k < 3; k++) pos[k] = nrm[k] * radius;
    }
}
} // namespace

int grt_host_primitive_counts(int kind, uint32_t* nv, uint32_t* nf)
{
    if ((kind != GRT_PRIM_PLANE && kind != GRT_PRIM_SPHERE) || !nv || !nf) { g_host_err = "grt_host_primitive_counts: bad argument"; return GRT_ERR_INVALID; }
    const LatticeSpec& L = kLattice[kind];
    *nv = L.rows * L.cols;
    *nf = 2u * (L.rows - 1u) * (L.cols - 1u);
    return GRT_OK;
}

int grt_host_primitive_fill(int kind, float* verts, float* normals, uint32_t* faces)
{
    if ((kind != GRT_PRIM_PLANE && kind != GRT_PRIM_SPHERE) || !verts || !normals || !faces) { g_host_err = "grt_host_primitive_fill: bad argument"; return GRT_ERR_INVALID; }
    const LatticeSpec& L = kLattice[kind];
    for (uint32_t r = 0; r < L.rows; r++)
        for (uint32_t c = 0; c < L.cols; c++) lattice_point(kind, r, c, verts + 3 * (size_t)(r * L.cols + c), normals + 3 * (size_t)(r * L.cols + c));
    // cell (r, c) -> (ll, lr, ur) (ur, ul, ll): the winding the reference emits
    uint32_t* f = faces;
    for (uint32_t r = 0; r + 1 < L.rows; r++)
        for (uint32_t c = 0; c + 1 < L.cols; c++) {
            const uint32_t ll = r * L.cols + c, lr = ll + 1u, ul = ll + L.cols, ur = ul + 1u;
            const uint32_t six[6] = {ll, lr, ur, ur, ul, ll};
            memcpy(f, six, sizeof(six));
            f += 6;
        }
    return GRT_OK;
}

// OBJ reader (tinyobjloader is not vendored).  One (position, normal) pair is emitted per face corner, in file order,
// exactly as the reference un-indexes tinyobj's output (Primitives.cpp:157-191); positions and normals get the Y flip
// (:175,:179).  Polygons are fan-triangulated (tinyobj's default).  A face corner without a normal index — which the
// reference dereferences at index -1 — is an error here; so is any parse failure (the reference calls exit(1)).
namespace {
struct ObjSoup { std::vector<float> v, n; };
int obj_parse(const char* path, ObjSoup& out)
{
    std::ifstream f(path);
    if (!f) { g_host_err = std::string("OBJ: cannot open ") + path; return GRT_ERR_IO; }
    std::vector<float> vs, ns;
    std::string line;
    size_t lineno = 0;
    auto fail = [&](const std::string& what) { g_host_err = "OBJ: " + what + " at line " + std::to_string(lineno) + " of " + path; return GRT_ERR_IO; };
    auto resolve = [](long idx, size_t n, size_t* o) {
        if (idx > 0 && (size_t)idx <= n) { *o = (size_t)idx - 1; return true; }
        if (idx < 0 && (size_t)(-idx) <= n) { *o = n - (size_t)(-idx); return true; }
        return false;
    };
    while (std::getline(f, line)) {
        lineno++;
        std::istringstream ss(line);
        std::string tok;
        if (!(ss >> tok)) continue;
        if (tok == "v" || tok == "vn") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) return fail(tok == "v" ? "bad vertex" : "bad normal");
            std::vector<float>& dst = tok == "v" ? vs : ns;
            dst.push_back(x); dst.push_back(y); dst.push_back(z);
        } else if (tok == "f") {
            std::vector<std::pair<size_t, size_t>> corners;
            std::string c;
            while (ss >> c) {
                const size_t s1 = c.find('/');
                const size_t s2 = s1 == std::string::npos ? std::string::npos : c.find('/', s1 + 1);
                if (s2 == std::string::npos || s2 + 1 >= c.size()) return fail("face corner without a normal");
                long vi = 0, ni = 0;
                try { vi = std::stol(c.substr(0, s1)); ni = std::stol(c.substr(s2 + 1)); } catch (...) { return fail("bad face corner"); }
                size_t a, b;
                if (!resolve(vi, vs.size() / 3, &a)) return fail("bad vertex index");
                if (!resolve(ni, ns.size() / 3, &b)) return fail("bad normal index");
                corners.push_back({a, b});
            }
            if (corners.size() < 3) return fail("face with fewer than 3 corners");
            for (size_t k = 1; k + 1 < corners.size(); k++)
                for (size_t t : {(size_t)0, k, k + 1}) {
                    const float* pv = &vs[corners[t].first * 3];
                    const float* pn = &ns[corners[t].second * 3];
                    out.v.push_back(pv[0]); out.v.push_back(-pv[1]); out.v.push_back(pv[2]);
                    out.n.push_back(pn[0]); out.n.push_back(-pn[1]); out.n.push_back(pn[2]);
                }
        }
    }
    if (out.v.empty()) { g_host_err = std::string("OBJ: no faces in ") + path; return GRT_ERR_IO; }
    return GRT_OK;
}
} // namespace

int grt_host_obj_count(const char* path, uint32_t* nv, uint32_t* nf)
{
    if (!path || !nv || !nf) return GRT_ERR_INVALID;
    ObjSoup s;
    const int rc = obj_parse(path, s);
    if (rc != GRT_OK) return rc;
    *nv = (uint32_t)(s.v.size() / 3);
    *nf = *nv / 3;
    return GRT_OK;
}

int grt_host_obj_read(const char* path, uint32_t nv, float* verts, float* normals, uint32_t* faces)
{
    if (!path || !verts || !normals || !faces) return GRT_ERR_INVALID;
    ObjSoup s;
    const int rc = obj_parse(path, s);
    if (rc != GRT_OK) return rc;
    if (s.v.size() / 3 != nv) { g_host_err = "grt_host_obj_read: vertex count mismatch"; return GRT_ERR_INVALID; }
    memcpy(verts, s.v.data(), s.v.size() * sizeof(float));
    memcpy(normals, s.n.data(), s.n.size() * sizeof(float));
    for (uint32_t i = 0; i < nv; i++) faces[i] = i; // un-indexed soup: face k = corners 3k, 3k+1, 3k+2
    return GRT_OK;
}

// Writes (verts, normals, faces) as OBJ with "%.9g" (round-trips fp32 exactly), v//vn corners sharing one index.
// The Y flip of the reader is NOT undone: write_obj(flip_y(mesh)) followed by a read gives mesh back.
int grt_host_obj_write(const char* path, uint32_t nv, const float* verts, const float* normals, uint32_t nf, const uint32_t* faces)
{
    if (!path || !verts || !normals || !faces) return GRT_ERR_INVALID;
    FILE* fp = fopen(path, "w");
    if (!fp) { g_host_err = std::string("cannot create ") + path; return GRT_ERR_IO; }
    for (uint32_t i = 0; i < nv; i++) fprintf(fp, "v %.9g %.9g %.9g\n", verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    for (uint32_t i = 0; i < nv; i++) fprintf(fp, "vn %.9g %.9g %.9g\n", normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    for (uint32_t k = 0; k < nf; k++) {
        const uint32_t a = faces[3 * k] + 1, b = faces[3 * k + 1] + 1, c = faces[3 * k + 2] + 1;
        if (a > nv || b > nv || c > nv) { fclose(fp); g_host_err = "grt_host_obj_write: face index out of range"; return GRT_ERR_INVALID; }
        fprintf(fp, "f %u//%u %u//%u %u//%u\n", a, a, b, b, c, c);
    }
    fclose(fp);
    return GRT_OK;
}

} // extern "C"
