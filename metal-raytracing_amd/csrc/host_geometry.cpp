// host_geometry.cpp — host-side geometry ingest for the C ABI (no GPU code).
//
//  * OBJ/MTL reader standing in for ModelIO, which the reference uses through
//    MDLAsset(url:vertexDescriptor:bufferAllocator:) (Model.swift:16-21) and
//    Material(material: MDLMaterial?) (SubMesh.swift:37-54).  ModelIO is closed source; the
//    choices made here are: fan triangulation (v0,vi,vi+1); one vertex per distinct
//    (v,vn) pair in first-use order; one submesh per `usemtl` statement in file order; Kd →
//    baseColor, Ks → specular, Ke → emission, Ns → specularExponent, Ni → refractionIndex,
//    d → dissolve; a missing material gives grey 0.8 (the reference would give black,
//    SubMesh.swift:38); missing `vn` gives area-weighted smooth normals (the reference would feed
//    zero normals → NaN).
//  * T*R*S matrices (Mesh.swift:21-24, Utilities.swift:104-166) and the default camera
//    (Scene.swift:40-57).
//  * Deterministic procedural stand-ins for the two meshes the reference checkout lacks
//    (.MISSING_LARGE_BLOBS): dragon (871 414 triangles) and bunny (69 451 triangles).
#include "host_geometry.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <fstream>
#include <sstream>
#include <unordered_map>
#include <algorithm>

namespace mrt {

void set_error(const std::string &msg);   // api.cpp

// ------------------------------------------------------------------------ matrices
static void m4_identity(float m[16]) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }
// column-major: m[col*4+row]; r = a*b with the un-fused order ((a0*b0 + a1*b1) + a2*b2) + a3*b3
static void m4_mul(const float a[16], const float b[16], float r[16]) {
    float t[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            t[j * 4 + i] = ((a[0 * 4 + i] * b[j * 4 + 0] + a[1 * 4 + i] * b[j * 4 + 1]) + a[2 * 4 + i] * b[j * 4 + 2]) + a[3 * 4 + i] * b[j * 4 + 3];
    memcpy(r, t, 64);
}
static void m4_rotate_axis(float radians, float ax, float ay, float az, float m[16]) {   // Utilities.swift:113-126
    float inv = 1.0f / sqrtf((ax * ax + ay * ay) + az * az);
    float x = ax * inv, y = ay * inv, z = az * inv;
    float ct = cosf(radians), st = sinf(radians), ci = 1 - ct;
    m4_identity(m);
    m[0] = ct + x * x * ci;     m[1] = y * x * ci + z * st; m[2] = z * x * ci - y * st;
    m[4] = x * y * ci - z * st; m[5] = ct + y * y * ci;     m[6] = z * y * ci + x * st;
    m[8] = x * z * ci + y * st; m[9] = y * z * ci - x * st; m[10] = ct + z * z * ci;
}
void make_transform(const float p[3], const float r[3], float s, float out[16]) {
    float rx[16], ry[16], rz[16], rot[16], T[16], S[16], tr[16];
    m4_rotate_axis(r[0], 1, 0, 0, rx); m4_rotate_axis(r[1], 0, 1, 0, ry); m4_rotate_axis(r[2], 0, 0, 1, rz);
    m4_mul(rx, ry, rot); m4_mul(rot, rz, rot);                 // Utilities.swift:140-142
    m4_identity(T); T[12] = p[0]; T[13] = p[1]; T[14] = p[2];  // Utilities.swift:104-111
    m4_identity(S); S[0] = S[5] = S[10] = s;                   // Utilities.swift:144-155
    m4_mul(T, rot, tr); m4_mul(tr, S, out);                    // Mesh.swift:24
}
void default_camera(int w, int h, MRTCamera *c) {              // Scene.swift:40-57
    memset(c, 0, sizeof *c);
    c->position = MRTFloat3{0.0f, 1.0f, 5.38f, 0};
    float fov = 45.0f * (3.14159274f / 180.0f);
    float aspect = (float)w / (float)h;
    float ih = tanf(fov / 2.0f);
    float iw = aspect * ih;
    c->right = MRTFloat3{1.0f * iw, 0.0f * iw, 0.0f * iw, 0};
    c->up = MRTFloat3{0.0f * ih, 1.0f * ih, 0.0f * ih, 0};
    c->forward = MRTFloat3{0.0f, 0.0f, -1.0f, 0};
}

// ------------------------------------------------------------------------ normals
// area-weighted smooth normals for the vertices flagged in `need`
static void smooth_normals(MeshData &m, const std::vector<char> &need) {
    size_t nv = m.positions.size() / 3;
    std::vector<double> acc(nv * 3, 0.0);
    for (auto &s : m.submeshes)
        for (size_t t = 0; t + 2 < s.indices.size(); t += 3) {
            uint32_t a = s.indices[t], b = s.indices[t + 1], c = s.indices[t + 2];
            double ax = m.positions[a * 3], ay = m.positions[a * 3 + 1], az = m.positions[a * 3 + 2];
            double e1x = m.positions[b * 3] - ax, e1y = m.positions[b * 3 + 1] - ay, e1z = m.positions[b * 3 + 2] - az;
            double e2x = m.positions[c * 3] - ax, e2y = m.positions[c * 3 + 1] - ay, e2z = m.positions[c * 3 + 2] - az;
            double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
            for (uint32_t v : {a, b, c}) { acc[v * 3] += nx; acc[v * 3 + 1] += ny; acc[v * 3 + 2] += nz; }
        }
    for (size_t v = 0; v < nv; v++) {
        if (!need[v]) continue;
        double l = sqrt(acc[v * 3] * acc[v * 3] + acc[v * 3 + 1] * acc[v * 3 + 1] + acc[v * 3 + 2] * acc[v * 3 + 2]);
        if (l > 0) { m.normals[v * 3] = (float)(acc[v * 3] / l); m.normals[v * 3 + 1] = (float)(acc[v * 3 + 1] / l); m.normals[v * 3 + 2] = (float)(acc[v * 3 + 2] / l); }
        else { m.normals[v * 3] = 0; m.normals[v * 3 + 1] = 1; m.normals[v * 3 + 2] = 0; }
    }
}

// ------------------------------------------------------------------------ MTL / OBJ
static MRTMaterial default_material() {
    MRTMaterial m; memset(&m, 0, sizeof m);
    m.baseColor = MRTFloat3{0.8f, 0.8f, 0.8f, 0};
    m.refractionIndex = 1.0f; m.dissolve = 1.0f;
    return m;
}

static bool load_mtl(const std::string &path, std::unordered_map<std::string, MRTMaterial> &out) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line, cur;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k; ss >> k;
        if (k == "newmtl") { ss >> cur; MRTMaterial m; memset(&m, 0, sizeof m); m.dissolve = 1.0f; m.refractionIndex = 1.0f; out[cur] = m; }
        else if (cur.empty()) continue;
        else if (k == "Kd") { MRTFloat3 &c = out[cur].baseColor; ss >> c.x >> c.y >> c.z; }
        else if (k == "Ks") { MRTFloat3 &c = out[cur].specular; ss >> c.x >> c.y >> c.z; }
        else if (k == "Ke") { MRTFloat3 &c = out[cur].emission; ss >> c.x >> c.y >> c.z; }
        else if (k == "Ns") ss >> out[cur].specularExponent;
        else if (k == "Ni") ss >> out[cur].refractionIndex;
        else if (k == "d") ss >> out[cur].dissolve;
    }
    return true;
}

static inline const char *skip_ws(const char *p) { while (*p == ' ' || *p == '\t') p++; return p; }

bool load_obj(const std::string &path, MeshData &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error("cannot open OBJ file: " + path); return false; }
    std::string dir;
    { size_t s = path.find_last_of('/'); dir = s == std::string::npos ? std::string() : path.substr(0, s + 1); }
    std::vector<float> V, N;
    std::unordered_map<std::string, MRTMaterial> mtl;
    std::unordered_map<uint64_t, uint32_t> dedup;
    std::vector<char> need_normal;
    out = MeshData();
    Submesh *cur = nullptr;
    auto begin_submesh = [&](const std::string &name) {
        if (cur && cur->indices.empty()) out.submeshes.pop_back();
        out.submeshes.emplace_back();
        cur = &out.submeshes.back();
        cur->name = name;
        auto it = mtl.find(name);
        cur->material = it == mtl.end() ? default_material() : it->second;
    };
    std::string line;
    std::vector<uint32_t> poly;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char *p = skip_ws(line.c_str());
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float x = 0, y = 0, z = 0; sscanf(p + 2, "%f %f %f", &x, &y, &z); V.push_back(x); V.push_back(y); V.push_back(z);
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            float x = 0, y = 0, z = 0; sscanf(p + 3, "%f %f %f", &x, &y, &z); N.push_back(x); N.push_back(y); N.push_back(z);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            if (!cur) begin_submesh("");
            poly.clear();
            const char *q = p + 2;
            for (;;) {
                q = skip_ws(q);
                if (!*q) break;
                char *end;
                long vi = strtol(q, &end, 10);
                if (end == q) break;
                long ni = 0; bool has_n = false;
                q = end;
                if (*q == '/') {
                    q++;
                    if (*q != '/') { strtol(q, &end, 10); q = end; }          // vt, ignored
                    if (*q == '/') { q++; ni = strtol(q, &end, 10); has_n = end != q; q = end; }
                }
                long nv = (long)(V.size() / 3), nn = (long)(N.size() / 3);
                long v0 = vi > 0 ? vi - 1 : nv + vi;
                long n0 = has_n ? (ni > 0 ? ni - 1 : nn + ni) : -1;
                if (v0 < 0 || v0 >= nv || n0 >= nn) { set_error("OBJ index out of range in " + path); return false; }
                uint64_t key = ((uint64_t)(uint32_t)v0 << 32) | (uint32_t)(n0 + 1);
                auto it = dedup.find(key);
                uint32_t id;
                if (it == dedup.end()) {
                    id = (uint32_t)(out.positions.size() / 3);
                    dedup.emplace(key, id);
                    for (int k = 0; k < 3; k++) out.positions.push_back(V[v0 * 3 + k]);
                    if (n0 >= 0) { for (int k = 0; k < 3; k++) out.normals.push_back(N[n0 * 3 + k]); need_normal.push_back(0); }
                    else { for (int k = 0; k < 3; k++) out.normals.push_back(0.0f); need_normal.push_back(1); }
                } else id = it->second;
                poly.push_back(id);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) { cur->indices.push_back(poly[0]); cur->indices.push_back(poly[k]); cur->indices.push_back(poly[k + 1]); }
        } else if (!strncmp(p, "usemtl", 6)) {
            std::istringstream ss(p + 6); std::string name; ss >> name; begin_submesh(name);
        } else if (!strncmp(p, "mtllib", 6)) {
            std::istringstream ss(p + 6); std::string name; ss >> name; load_mtl(dir + name, mtl);   // a missing file is tolerated (teapot.obj → default.mtl)
        }
    }
    if (cur && cur->indices.empty()) out.submeshes.pop_back();
    if (std::find(need_normal.begin(), need_normal.end(), (char)1) != need_normal.end()) { smooth_normals(out, need_normal); out.generated_normals = true; }
    if (out.submeshes.empty()) { set_error("OBJ file has no faces: " + path); return false; }
    return true;
}

// ------------------------------------------------------------------------ procedural meshes
namespace {
struct P3 { double x, y, z; };
static P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static P3 operator*(P3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static P3 crossd(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static P3 normd(P3 a) { double l = sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return l > 0 ? a * (1.0 / l) : P3{0, 1, 0}; }

struct ProcMesh {
    std::vector<P3> v; std::vector<uint32_t> idx;
    uint32_t add(P3 p) { v.push_back(p); return (uint32_t)v.size() - 1; }
    void tri(uint32_t a, uint32_t b, uint32_t c) { idx.push_back(a); idx.push_back(b); idx.push_back(c); }
    // closed tube along a centreline: (V+1) rings of U vertices, caps with a centre vertex: 2UV + 2U triangles
    template <class C, class R> void tube(int U, int Vn, C centre, R radius) {
        const double PI = 3.14159265358979323846;
        std::vector<uint32_t> ring((size_t)(Vn + 1) * U);
        P3 prevN{0, 0, 1};
        for (int j = 0; j <= Vn; j++) {
            double s = (double)j / Vn;
            P3 c = centre(s);
            double h = 1e-4;
            P3 t = normd(centre(std::min(1.0, s + h)) - centre(std::max(0.0, s - h)));
            P3 n = normd(prevN - t * (prevN.x * t.x + prevN.y * t.y + prevN.z * t.z));   // parallel-transport frame
            P3 b = crossd(t, n);
            prevN = n;
            for (int i = 0; i < U; i++) {
                double a = 2 * PI * i / U;
                double r = radius(s, a);
                ring[(size_t)j * U + i] = add(c + n * (r * cos(a)) + b * (r * sin(a)));
            }
        }
        for (int j = 0; j < Vn; j++)
            for (int i = 0; i < U; i++) {
                uint32_t a = ring[(size_t)j * U + i], b = ring[(size_t)j * U + (i + 1) % U];
                uint32_t c = ring[(size_t)(j + 1) * U + i], d = ring[(size_t)(j + 1) * U + (i + 1) % U];
                tri(a, c, b); tri(b, c, d);
            }
        uint32_t c0 = add(centre(0.0)), c1 = add(centre(1.0));
        for (int i = 0; i < U; i++) {
            tri(c0, ring[i], ring[(i + 1) % U]);
            tri(c1, ring[(size_t)Vn * U + (i + 1) % U], ring[(size_t)Vn * U + i]);
        }
    }
    // cone: n side triangles (+ n base-fan triangles when closed)
    void cone(int n, P3 base, P3 axis, double radius, bool closed) {
        const double PI = 3.14159265358979323846;
        P3 t = normd(axis);
        P3 ref = fabs(t.y) < 0.9 ? P3{0, 1, 0} : P3{1, 0, 0};
        P3 u = normd(crossd(t, ref)), w = crossd(t, u);
        uint32_t apex = add(base + axis);
        std::vector<uint32_t> r(n);
        for (int i = 0; i < n; i++) { double a = 2 * PI * i / n; r[i] = add(base + u * (radius * cos(a)) + w * (radius * sin(a))); }
        for (int i = 0; i < n; i++) tri(apex, r[i], r[(i + 1) % n]);
        if (closed) { uint32_t c = add(base); for (int i = 0; i < n; i++) tri(c, r[(i + 1) % n], r[i]); }
    }
};

static void finish(ProcMesh &pm, const double half[3], const char *name, const MRTMaterial &mat, MeshData &out) {
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    for (auto &p : pm.v) { lo[0] = std::min(lo[0], p.x); hi[0] = std::max(hi[0], p.x); lo[1] = std::min(lo[1], p.y); hi[1] = std::max(hi[1], p.y); lo[2] = std::min(lo[2], p.z); hi[2] = std::max(hi[2], p.z); }
    out = MeshData();
    out.positions.resize(pm.v.size() * 3); out.normals.assign(pm.v.size() * 3, 0.0f);
    for (size_t i = 0; i < pm.v.size(); i++) {
        double q[3] = {pm.v[i].x, pm.v[i].y, pm.v[i].z};
        for (int k = 0; k < 3; k++) out.positions[i * 3 + k] = (float)(((q[k] - lo[k]) / (hi[k] - lo[k]) * 2.0 - 1.0) * half[k]);
    }
    out.submeshes.emplace_back();
    out.submeshes[0].name = name; out.submeshes[0].material = mat; out.submeshes[0].indices = pm.idx;
    smooth_normals(out, std::vector<char>(pm.v.size(), 1));
    out.generated_normals = true;
}
}  // namespace

// Dragon stand-in: a serpentine body with scale-like displacement, four legs, two horns, a row of
// 25 dorsal spikes and a tongue, all tessellated as closed tubes with near-square quads (no sliver
// fans: the Stanford scan it stands in for is uniformly tessellated).
// 740 352 + 102 912 + 12 928 + 15 000 + 222 = 871 414 triangles (Stanford dragon count).
// Centred at the origin, half-extents (0.45, 0.317, 0.20).
void make_dragon_proxy(MeshData &out) {
    const double PI = 3.14159265358979323846;
    ProcMesh pm;
    pm.v.reserve(450000); pm.idx.reserve(871414 * 3);
    auto spine = [&](double s) -> P3 {
        double x = -1.0 + 2.0 * s;
        double y = 0.18 * sin(2.2 * PI * s + 0.4) + 0.55 * s * s * s - 0.05;
        double z = 0.22 * sin(3.0 * PI * s) * (1.0 - 0.5 * s);
        return P3{x, y, z};
    };
    auto body_base = [&](double s) -> double {
        double base = 0.23 * pow(sin(PI * std::min(1.0, 0.02 + s * 1.16)), 1.5) * (1.0 - 0.55 * s);
        if (s > 0.86) base += 0.10 * exp(-pow((s - 0.94) / 0.045, 2.0));            // head bulge
        double taper = std::min(1.0, std::min(s, 1.0 - s) / 0.004);                  // ends close to a point: tiny caps
        return (0.02 + base) * (0.03 + 0.97 * taper);
    };
    auto body_r = [&](double s, double a) -> double {
        double base = body_base(s);
        double scales = 0.010 * sin(140.0 * PI * s) * sin(24.0 * a) + 0.005 * sin(61.0 * a + 300.0 * s);
        double belly = 1.0 - 0.18 * cos(a);
        return std::max(0.0005, base * belly + scales * std::min(1.0, base / 0.2));
    };
    pm.tube(512, 722, spine, body_r);                                               // 2*512*723 = 740 352
    const double leg_s[4] = {0.30, 0.30, 0.62, 0.62};
    const double leg_side[4] = {1, -1, 1, -1};
    for (int l = 0; l < 4; l++) {
        P3 root = spine(leg_s[l]);
        double sd = leg_side[l];
        auto leg_c = [&](double s) -> P3 {
            double bend = sin(PI * s);
            return P3{root.x + 0.10 * bend * (l < 2 ? 1 : -1) + 0.05 * s, root.y - 0.05 - (root.y + 0.62) * s, root.z + sd * (0.14 + 0.10 * bend)};
        };
        auto leg_r = [&](double s, double a) -> double {
            double taper = std::min(1.0, std::min(s, 1.0 - s) / 0.01);
            return (0.055 * (1.0 - 0.6 * s) + 0.02 * exp(-pow((s - 0.97) / 0.05, 2.0)) + 0.003 * sin(8.0 * a) * sin(40.0 * s)) * (0.05 + 0.95 * taper);
        };
        pm.tube(64, 200, leg_c, leg_r);                                             // 4 x 25 728
    }
    for (int h = 0; h < 2; h++) {
        P3 root = spine(0.95);
        double sd = h ? 1 : -1;
        auto horn_c = [&](double s) -> P3 { return P3{root.x - 0.22 * s + 0.05 * s * s, root.y + 0.10 + 0.28 * s, root.z + sd * (0.06 + 0.10 * s * s)}; };
        auto horn_r = [&](double s, double a) -> double { return 0.03 * (1.0 - 0.97 * s) * (0.05 + 0.95 * std::min(1.0, s / 0.02)) + 0.001 * sin(6.0 * a + 30.0 * s) * (1.0 - s); };
        pm.tube(32, 100, horn_c, horn_r);                                           // 2 x 6 464
    }
    for (int k = 0; k < 25; k++) {
        double s = 0.10 + 0.78 * k / 24.0;
        P3 c = spine(s);
        double r0 = body_base(s);
        P3 base{c.x, c.y + 0.70 * r0, c.z};
        double hgt = 0.10 + 0.05 * sin(7.0 * s);
        auto spike_c = [&](double q) -> P3 { return P3{base.x - 0.03 * q, base.y + hgt * q, base.z}; };
        auto spike_r = [&](double q, double a) -> double { return 0.022 * (1.0 - 0.97 * q) * (0.1 + 0.9 * std::min(1.0, q / 0.05)); };
        pm.tube(12, 24, spike_c, spike_r);                                          // 25 x 600
    }
    {
        P3 mouth = spine(0.995);
        auto tongue_c = [&](double q) -> P3 { return P3{mouth.x + 0.09 * q, mouth.y - 0.03 - 0.03 * q * q, mouth.z + 0.01 * sin(6.0 * q)}; };
        auto tongue_r = [&](double q, double a) -> double { return 0.006 * (1.0 - 0.9 * q); };
        pm.tube(3, 36, tongue_c, tongue_r);                                         // 222
    }
    MRTMaterial mat; memset(&mat, 0, sizeof mat);                 // Resources/dragon.mtl: Kd 1 0 0, Ks .2, Ns 37.25, Ni 1, d 1
    mat.baseColor = MRTFloat3{1.0f, 0.0f, 0.0f, 0}; mat.specular = MRTFloat3{0.2f, 0.2f, 0.2f, 0};
    mat.specularExponent = 37.254902f; mat.refractionIndex = 1.0f; mat.dissolve = 1.0f;
    const double half[3] = {0.45, 0.317, 0.20};
    finish(pm, half, "Dragon", mat, out);
}

// Second dragon stand-in with IRREGULAR connectivity (sensitivity check for the benchmark geometry: the tube proxy above is a
// regular grid of near-square quads).  An assembly of subdivided, noise-displaced, tangentially jittered icospheres (plus a few
// small closed solids that make the count exact) laid along the same spine, inside the same extents, with the same material:
//   2 x 327 680 + 2 x 81 920 + 2 x 20 480 + 2 x 5 120 + 3 x 320 + 2 x 20 + 8 + 6 = 871 414 triangles.
// Triangle sizes vary by more than 10x across a blob (a nonlinear warp of the sphere before projection), vertices are jittered
// along the surface by up to 0.3 edge lengths, and both the vertex numbering and the triangle order are shuffled.
namespace {
struct Rng { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } double uni() { return next() / 2147483648.0; } };

static void icosphere(int level, std::vector<P3> &v, std::vector<uint32_t> &f) {
    const double t = (1.0 + sqrt(5.0)) / 2.0;
    const double base[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t}, {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
    const int faces[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                              {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    v.clear(); f.clear();
    for (auto &b : base) v.push_back(normd(P3{b[0], b[1], b[2]}));
    for (auto &t3 : faces) { f.push_back(t3[0]); f.push_back(t3[1]); f.push_back(t3[2]); }
    for (int l = 0; l < level; l++) {
        std::unordered_map<uint64_t, uint32_t> mid;
        mid.reserve(f.size());
        auto midpoint = [&](uint32_t a, uint32_t b) -> uint32_t {
            const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
            auto it = mid.find(key);
            if (it != mid.end()) return it->second;
            v.push_back(normd((v[a] + v[b]) * 0.5));
            return mid[key] = (uint32_t)v.size() - 1;
        };
        std::vector<uint32_t> g; g.reserve(f.size() * 4);
        for (size_t k = 0; k < f.size(); k += 3) {
            const uint32_t a = f[k], b = f[k + 1], c = f[k + 2], ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            const uint32_t q[12] = {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca};
            g.insert(g.end(), q, q + 12);
        }
        f.swap(g);
    }
}
}  // namespace

// hostile = the stress variant of the same assembly (same triangle count, extents and material): a 100 : 1 range of triangle sizes (the belly
// is a coarse level-4 blob, one hind leg a level-7 blob of half the size) and every 100th triangle pulled out into a sliver up to 50 times
// its edge (a duplicated vertex: the surface gets a fin and a hole, as scanned meshes have) — long thin triangles whose boxes overlap
// hundreds of others, the case the builder's triangle pre-splitting exists for.
static void dragon_irregular_impl(MeshData &out, bool hostile) {
    const double PI = 3.14159265358979323846;
    ProcMesh pm;
    pm.v.reserve(440000); pm.idx.reserve(871414 * 3);
    Rng rng{0x9E3779B97F4A7C15ull};
    auto spine = [&](double s) -> P3 {
        double x = -1.0 + 2.0 * s;
        return P3{x, 0.18 * sin(2.2 * PI * s + 0.4) + 0.55 * s * s * s - 0.05, 0.22 * sin(3.0 * PI * s) * (1.0 - 0.5 * s)};
    };
    std::vector<P3> sv; std::vector<uint32_t> sf;
    // one blob: unit directions -> nonlinear warp (uneven triangle sizes) -> ellipsoid radii * (1 + noise) -> tangential jitter
    auto blob = [&](int level, P3 c, P3 radii, double warp, double bump, double freq) {
        icosphere(level, sv, sf);
        const double edge = 1.2 / (double)(1 << level);                       // ~ edge length on the unit sphere
        const uint32_t base = (uint32_t)pm.v.size();
        for (auto &d0 : sv) {
            P3 d = normd(P3{d0.x + warp * d0.x * d0.x * d0.x + 0.35 * warp * d0.y * d0.z, d0.y + warp * d0.y * fabs(d0.y) * 0.8, d0.z + 0.6 * warp * d0.z * d0.x});
            const P3 j = P3{rng.uni() - 0.5, rng.uni() - 0.5, rng.uni() - 0.5} * (0.35 * edge);
            d = normd(d + j);                                                  // jitter along the surface (renormalised)
            const double n = sin(freq * d.x + 1.3) * sin(freq * 1.31 * d.y + 0.2) * sin(freq * 0.83 * d.z + 2.1) + 0.5 * sin(2.7 * freq * d.x * d.y + 3.0 * d.z);
            const double r = 1.0 + bump * n;
            pm.add(P3{c.x + radii.x * r * d.x, c.y + radii.y * r * d.y, c.z + radii.z * r * d.z});
        }
        for (size_t k = 0; k < sf.size(); k += 3) pm.tri(base + sf[k], base + sf[k + 1], base + sf[k + 2]);
    };
    blob(7, spine(0.30), P3{0.42, 0.30, 0.26}, 0.55, 0.10, 9.0);                // chest
    blob(hostile ? 4 : 7, spine(0.62), P3{0.40, 0.26, 0.24}, 0.45, 0.12, 11.0); // belly / hind
    blob(6, spine(0.93), P3{0.16, 0.14, 0.12}, 0.50, 0.08, 7.0);                // head
    blob(6, spine(0.08), P3{0.20, 0.08, 0.07}, 0.60, 0.10, 6.0);                // tail
    for (int l = 0; l < 2; l++) { P3 r = spine(0.30); blob(5, P3{r.x + 0.05, r.y - 0.42, r.z + (l ? 0.22 : -0.22)}, P3{0.07, 0.26, 0.07}, 0.40, 0.06, 5.0); }   // fore legs
    for (int l = 0; l < 2; l++) {                                                // hind legs
        P3 r = spine(0.62);
        const bool fine = hostile && l == 0;
        blob(fine ? 7 : 4, P3{r.x - 0.03, r.y - 0.40, r.z + (l ? 0.20 : -0.20)}, fine ? P3{0.028, 0.10, 0.028} : P3{0.07, 0.24, 0.07}, 0.40, 0.06, 5.0);
    }
    for (int k = 0; k < 3; k++) { P3 r = spine(0.35 + 0.2 * k); blob(2, P3{r.x, r.y + 0.34, r.z}, P3{0.03, 0.09, 0.03}, 0.2, 0.0, 1.0); }                        // dorsal spikes
    for (int k = 0; k < 2; k++) { P3 r = spine(0.96); blob(0, P3{r.x - 0.02, r.y + 0.17, r.z + (k ? 0.05 : -0.05)}, P3{0.02, 0.08, 0.02}, 0.0, 0.0, 1.0); }     // horns
    {   // an octahedron (8) and a triangular bipyramid (6) make the count exact
        P3 c = spine(0.995); c.x += 0.05;
        uint32_t o[6]; const double e = 0.02;
        const P3 ov[6] = {{e, 0, 0}, {-e, 0, 0}, {0, e, 0}, {0, -e, 0}, {0, 0, e}, {0, 0, -e}};
        for (int k = 0; k < 6; k++) o[k] = pm.add(c + ov[k]);
        const int of[8][3] = {{0, 2, 4}, {2, 1, 4}, {1, 3, 4}, {3, 0, 4}, {2, 0, 5}, {1, 2, 5}, {3, 1, 5}, {0, 3, 5}};
        for (auto &t3 : of) pm.tri(o[t3[0]], o[t3[1]], o[t3[2]]);
        P3 c2 = spine(0.0); c2.x -= 0.04;
        uint32_t b[5];
        for (int k = 0; k < 3; k++) b[k] = pm.add(P3{c2.x, c2.y + e * cos(2 * PI * k / 3), c2.z + e * sin(2 * PI * k / 3)});
        b[3] = pm.add(P3{c2.x - 2 * e, c2.y, c2.z}); b[4] = pm.add(P3{c2.x + 2 * e, c2.y, c2.z});
        for (int k = 0; k < 3; k++) { pm.tri(b[k], b[(k + 1) % 3], b[4]); pm.tri(b[(k + 1) % 3], b[k], b[3]); }
    }
    if (hostile) {          // slivers: every 100th triangle keeps a and c and gets b' = a + f (b - a), f up to 50, kept inside the mesh's box
        P3 lo{1e30, 1e30, 1e30}, hi{-1e30, -1e30, -1e30};
        for (auto &q : pm.v) { lo.x = std::min(lo.x, q.x); lo.y = std::min(lo.y, q.y); lo.z = std::min(lo.z, q.z); hi.x = std::max(hi.x, q.x); hi.y = std::max(hi.y, q.y); hi.z = std::max(hi.z, q.z); }
        const size_t nt = pm.idx.size() / 3;
        for (size_t t = 37; t < nt; t += 100) {
            const P3 a = pm.v[pm.idx[3 * t]], b = pm.v[pm.idx[3 * t + 1]];
            const P3 e = b - a;
            const double len = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
            if (!(len > 0.0)) continue;
            const double f = std::min(50.0, 0.25 / len);
            P3 q = a + e * f;
            q.x = std::min(hi.x, std::max(lo.x, q.x)); q.y = std::min(hi.y, std::max(lo.y, q.y)); q.z = std::min(hi.z, std::max(lo.z, q.z));
            pm.idx[3 * t + 1] = pm.add(q);
        }
    }
    // shuffle the vertex numbering and the triangle order (Fisher-Yates, fixed seed)
    {
        const uint32_t nv = (uint32_t)pm.v.size(); const size_t nt = pm.idx.size() / 3;
        std::vector<uint32_t> perm(nv);
        for (uint32_t i = 0; i < nv; i++) perm[i] = i;
        for (uint32_t i = nv - 1; i > 0; i--) std::swap(perm[i], perm[rng.next() % (i + 1)]);
        std::vector<P3> nvv(nv);
        for (uint32_t i = 0; i < nv; i++) nvv[perm[i]] = pm.v[i];
        pm.v.swap(nvv);
        for (auto &ix : pm.idx) ix = perm[ix];
        for (size_t i = nt - 1; i > 0; i--) {
            const size_t j = rng.next() % (i + 1);
            for (int k = 0; k < 3; k++) std::swap(pm.idx[3 * i + k], pm.idx[3 * j + k]);
        }
    }
    MRTMaterial mat; memset(&mat, 0, sizeof mat);                 // Resources/dragon.mtl, as make_dragon_proxy
    mat.baseColor = MRTFloat3{1.0f, 0.0f, 0.0f, 0}; mat.specular = MRTFloat3{0.2f, 0.2f, 0.2f, 0};
    mat.specularExponent = 37.254902f; mat.refractionIndex = 1.0f; mat.dissolve = 1.0f;
    const double half[3] = {0.45, 0.317, 0.20};
    finish(pm, half, "Dragon", mat, out);
}
void make_dragon_proxy_irregular(MeshData &out) { dragon_irregular_impl(out, false); }
void make_dragon_proxy_hostile(MeshData &out) { dragon_irregular_impl(out, true); }

// Bunny stand-in: 65 792 + 3 264 + 395 = 69 451 triangles (Stanford bunny count).
void make_bunny_proxy(MeshData &out) {
    const double PI = 3.14159265358979323846;
    ProcMesh pm;
    auto body_c = [&](double s) -> P3 { return P3{-0.5 + s, 0.25 * sin(PI * s) * s, 0.0}; };
    auto body_r = [&](double s, double a) -> double {
        double r = 0.42 * sqrt(std::max(0.0, sin(PI * std::min(1.0, s * 1.35)))) * (1.0 - 0.35 * s) + (s > 0.72 ? 0.16 * sin(PI * (s - 0.72) / 0.28) : 0.0);
        return std::max(0.003, r * (1.0 + 0.04 * sin(5.0 * a) * sin(9.0 * PI * s)));
    };
    pm.tube(128, 256, body_c, body_r);
    for (int e = 0; e < 2; e++) {
        double sd = e ? 1 : -1;
        auto ear_c = [&](double s) -> P3 { return P3{0.36 - 0.12 * s, 0.30 + 0.45 * s, sd * (0.07 + 0.06 * s)}; };
        auto ear_r = [&](double s, double a) -> double { return (0.055 * sin(PI * std::min(1.0, 0.15 + 0.85 * s)) + 0.004) * (1.0 - 0.5 * fabs(sin(a))); };
        pm.tube(16, 50, ear_c, ear_r);
    }
    pm.cone(395, P3{-0.52, 0.02, 0.0}, P3{-0.10, 0.06, 0.0}, 0.06, false);
    MRTMaterial mat = default_material();
    const double half[3] = {0.40, 0.40, 0.30};
    finish(pm, half, "Bunny", mat, out);
}

}  // namespace mrt
