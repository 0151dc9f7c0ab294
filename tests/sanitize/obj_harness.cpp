// ASan/UBSan harness for the OBJ / MTL reader (host_geometry.cpp): loads every file named on the command line, checks the invariants the scene upload relies on
#include "host_geometry.h"
#include <cstdio>
#include <cmath>
namespace mrt { void set_error(const std::string &) {} }
int main(int argc, char **argv) {
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        mrt::MeshData m;
        const bool ok = mrt::load_obj(argv[i], m);
        if (!ok) continue;
        const size_t nv = m.positions.size() / 3;
        if (m.positions.size() % 3 || m.normals.size() != m.positions.size()) { printf("%s: array sizes %zu %zu\n", argv[i], m.positions.size(), m.normals.size()); bad++; }
        for (auto &s : m.submeshes) {
            if (s.indices.size() % 3) { printf("%s: index count %zu\n", argv[i], s.indices.size()); bad++; }
            for (uint32_t ix : s.indices) if (ix >= nv) { printf("%s: index %u of %zu\n", argv[i], ix, nv); bad++; break; }
        }
    }
    return bad ? 1 : 0;
}
