// dragon_scene.cpp — what GameViewController.viewDidLoad + Renderer.init + N x draw(in:) do in the
// reference (GameViewController.swift:17-43, Renderer.swift:45-71, :284-351), on the C++ host mirror.
//   c++ -std=c++17 -Iinclude examples/dragon_scene.cpp -Lmetal-raytracing_amd -lmrt_hip -o dragon_scene
//   ./dragon_scene [width height frames out.ppm [instancing [devices]]]     instancing = 1: the two spheres share one BLAS under a TLAS;
//   devices = "0,1,2,3": the same frames on a device group (mrt::GroupRenderer: replicated scene, tile shards, one reduce per image)
#include <cstdio>
#include <cstdlib>
#include "mrt.hpp"

int main(int argc, char **argv) {
    int w = argc > 1 ? atoi(argv[1]) : 800, h = argc > 2 ? atoi(argv[2]) : 600;      // the storyboard's 800x600 MTKView
    int frames = argc > 3 ? atoi(argv[3]) : 16;
    const char *out = argc > 4 && argv[4][0] != '-' ? argv[4] : nullptr;
    const bool instancing = argc > 5 && atoi(argv[5]) != 0;
    if (const char *res = getenv("MRT_RESOURCES")) mrt::resourceDirectory() = res;
    std::vector<int> devices;
    if (argc > 6) for (const char *p = argv[6]; *p;) { devices.push_back((int)strtol(p, const_cast<char **>(&p), 10)); if (*p == ',') p++; }
    try {
        mrt::DragonScene scene(w, h);
        if (!devices.empty()) {
            mrt::GroupRenderer group(w, h, scene, devices, 1, 3, instancing);
            group.draw(frames);
            std::vector<float> acc = group.gather();                 // blocks until the image is assembled on rank 0
            MRTRenderStats rs = group.stats();
            double sum = 0; for (size_t i = 0; i < acc.size(); i += 4) sum += acc[i] + acc[i + 1] + acc[i + 2];
            printf("group=%d frames=%llu completed=%llu closest=%llu shadow=%llu checksum=%.9g reduce=\"%s\"\n", group.size(), (unsigned long long)rs.frames,
                   (unsigned long long)group.framesCompleted(), (unsigned long long)rs.closest_rays, (unsigned long long)rs.shadow_rays, sum, group.reduceMode().c_str());
            return 0;
        }
        mrt::Renderer renderer(w, h, scene, 0, 1, 3, instancing);
        MRTSceneStats ss = renderer.sceneStats();
        renderer.draw(frames);
        renderer.wait();
        MRTRenderStats rs = renderer.stats();
        std::vector<float> acc = renderer.accumulation();
        double sum = 0; for (size_t i = 0; i < acc.size(); i += 4) sum += acc[i] + acc[i + 1] + acc[i + 2];
        printf("triangles=%llu frames=%llu frameIndex=%u closest=%llu shadow=%llu ms=%.3f checksum=%.9g\n", (unsigned long long)ss.triangles,
               (unsigned long long)rs.frames, renderer.frameIndex(), (unsigned long long)rs.closest_rays, (unsigned long long)rs.shadow_rays, rs.ms_gpu_last, sum);
        if (out) {
            std::vector<uint8_t> img = renderer.tonemapped();
            FILE *f = fopen(out, "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", out); return 2; }
            fprintf(f, "P6\n%d %d\n255\n", w, h);
            for (size_t i = 0; i < img.size(); i += 4) fwrite(&img[i], 1, 3, f);
            fclose(f);
        }
    } catch (const mrt::Error &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
