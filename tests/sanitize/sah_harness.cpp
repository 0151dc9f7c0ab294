// ASan/UBSan harness for the host binned-SAH builder (bvh_host_sah.cpp): random and degenerate box sets, topology invariants
#include "scene_device.h"
#include <cstdio>
#include <random>
#include <vector>
#include <cmath>
namespace mrt { void host_sah_topology(const float4 *lo, const float4 *hi, uint32_t n, std::vector<uint32_t> &order, std::vector<uint32_t> &left, std::vector<uint32_t> &right, std::vector<uint32_t> &parent); }
int main() {
    std::mt19937 rng(7); int bad = 0;
    const float specials[] = {0.0f, -0.0f, 1.0f, 1e-30f, 1e30f, 3.0e38f, -3.0e38f, INFINITY, -INFINITY, NAN};
    for (int rep = 0; rep < 600; rep++) {
        const uint32_t n = rep < 40 ? 1 + rep : (rng() % 5 == 0 ? 1 + rng() % 20000 : 1 + rng() % 300);
        const int kind = rng() % 6;
        std::vector<float4> lo(n), hi(n);
        std::uniform_real_distribution<float> U(-1.0f, 1.0f);
        for (uint32_t i = 0; i < n; i++) {
            float c[3], e[3];
            for (int k = 0; k < 3; k++) { c[k] = kind == 1 ? 0.5f : kind == 2 ? (k == 0 ? U(rng) : 0.0f) : U(rng) * (kind == 3 ? 1e30f : 1.0f); e[k] = kind == 4 ? 0.0f : std::fabs(U(rng)) * 0.01f; }
            lo[i] = make_float4(c[0] - e[0], c[1] - e[1], c[2] - e[2], 0); hi[i] = make_float4(c[0] + e[0], c[1] + e[1], c[2] + e[2], 0);
            if (kind == 5 && rng() % 7 == 0) { float *p = rng() % 2 ? &lo[i].x : &hi[i].x; p[rng() % 3] = specials[rng() % 10]; }
        }
        std::vector<uint32_t> o, l, r, p;
        mrt::host_sah_topology(lo.data(), hi.data(), n, o, l, r, p);
        if (o.size() != n || p.size() != 2 * (size_t)n - 1 || (n > 1 && (l.size() != n - 1 || r.size() != n - 1))) { printf("rep %d n %u: sizes\n", rep, n); bad++; continue; }
        std::vector<uint8_t> seen(n, 0); for (uint32_t x : o) { if (x >= n || seen[x]) { printf("rep %d n %u: order\n", rep, n); bad++; break; } seen[x] = 1; }
        std::vector<uint8_t> ref(2 * (size_t)n - 1, 0);
        for (uint32_t i = 0; i + 1 < n; i++) for (uint32_t c : {l[i], r[i]}) { if (c >= 2 * n - 1 || ref[c] || p[c] != i) { printf("rep %d n %u kind %d: child %u of %u\n", rep, n, kind, c, i); bad++; i = n; break; } ref[c] = 1; }
    }
    printf("bad %d\n", bad); return bad ? 1 : 0;
}
