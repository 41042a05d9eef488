// example_render.cpp — the reference's RenderScene() flow (RayTracer.cpp:11-20,
// SceneRenderer.cpp:13-29) on the GPU backend, for a 64^3 test grid.  Arguments: the HIP device ordinals to render
// on (default: 0); with several the frame's rows are spread over them (`example_render 0 1 2 3 4 5 6 7` on a node;
// `example_render 0 0` rehearses two on a one-GPU box).  Prints the mean and a hash of the frame's bits: the frame
// is the same for every device count.  Build:
//   g++ -std=c++17 -O2 example_render.cpp -L../csrc -lvr_backend -Wl,-rpath,'$ORIGIN/../csrc' -o example_render
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "GpuIntegrator.hpp"

int main(int argc, char** argv)
{
    using namespace vr_host;
    std::vector<int> devices;
    for (int a = 1; a < argc; ++a) devices.push_back(std::atoi(argv[a]));
    if (devices.empty()) devices.push_back(0);
    const int n = 64;
    std::vector<float> vox((size_t)n * n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const float r = std::sqrt((i - 32.f) * (i - 32.f) + (j - 32.f) * (j - 32.f) + (k - 32.f) * (k - 32.f));
                vox[((size_t)i * n + j) * n + k] = r < 24.f ? 1.f - r / 24.f : 0.f;
            }
    vr_grid_desc g{};
    g.voxels = vox.data();
    for (int a = 0; a < 3; ++a) { g.bbox_min[a] = -n / 2; g.bbox_max[a] = n / 2 - 1; }
    g.inv_mat[0] = g.inv_mat[4] = g.inv_mat[8] = 1.f;
    g.mat[0] = g.mat[4] = g.mat[8] = 1.f;
    g.max_value = -1.f;

    try {
        SceneInfo scene(devices);
        scene.setDensity(g);
        Options o = defaultOptions();
        o.width = o.widthReference = 256; o.height = o.heightReference = 256;
        o.rayPerPixelCount = 16; o.maxDepth = 16;
        o.sigma_a = 0.01f; o.sigma_s = 0.09f; o.heyneyGreensteinG = 0.924f;
        o.stepSizeMin = 0.005f; o.stepSizeMax = 9999.9f; o.stepSizeMultiplier = 0.001f;
        o.cameraPosition[2] = 70.f; o.fov = 45.f;
        o.backgroundColor[0] = 1.44f; o.backgroundColor[1] = 2.808f; o.backgroundColor[2] = 3.92f;
        o.lightPosition[1] = 1.f; o.lightColor[0] = o.lightColor[1] = o.lightColor[2] = 1024.f;
        auto renderer = IntegratorFactory::GetIntegrator("ratioTracking");
        std::vector<Vec3f> fb((size_t)o.width * o.height);
        renderer->render(o, &scene, fb.data());
        double sum = 0;
        for (const Vec3f& p : fb) sum += p.x + p.y + p.z;
        uint64_t hash = 1469598103934665603ull;                       // FNV-1a over the frame's bit patterns
        for (const Vec3f& p : fb) {
            uint32_t w[3];
            std::memcpy(w, &p, sizeof w);
            for (uint32_t v : w) { hash ^= v; hash *= 1099511628211ull; }
        }
        std::printf("devices %u\n", scene.deviceCount());
        std::printf("mean channel value %.6f\n", sum / (3.0 * fb.size()));
        std::printf("frame hash %016llx\n", (unsigned long long)hash);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
