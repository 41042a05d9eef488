// GpuIntegrator.hpp — host-side mirror of the reference's integrator interface over the C-ABI.
//
// The reference selects an integrator by name (IntegratorFactory::GetIntegrator,
// integrators/IntegratorFactory.cpp:12-24) and drives it through
//   BaseIntegrator::render(Options&, SceneInfo*)                                 (BaseIntegrator.h:69)
//   BaseIntegrator::renderPartial(Vec3f* orig, Vec3f* pix, ThreadInfo*, const Options&, SceneInfo*)  (:67)
// This header keeps those names and argument meanings for the integrators on the GPU path
// and forwards to include/vr_backend.h.  Options / SceneInfo here are the POD subsets the path
// reads (vr_options_pod, vr_grid_desc, vr_mesh_desc); INTEGRATION.md shows how the reference's own
// structs map onto them.  Header-only, C++17, links libvr_backend.so only.
//
// Error behaviour: the reference exits or throws on fatal errors (SceneLoader.cpp:90,213) and
// dereferences nullptr for an unknown integrator name (SceneRenderer.cpp:22).  Here every failure
// becomes a std::runtime_error carrying vr_backend_last_error(); GetIntegrator returns nullptr
// for names outside the path, as the reference's factory does for unknown names.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vr_backend.h"

namespace vr_host {

using Options = vr_options_pod;

struct Vec3f {
    float x, y, z;
};

// ThreadInfo (Utils/Types.h:148-152): absolute rows [fromHeight, toHeight)
struct ThreadInfo {
    uint32_t fromHeight;
    uint32_t toHeight;
};

// The SceneInfo members the path reads (Utils/Types.h:129-146), resident on every GPU of the list (the grid is
// replicated: 288 GB of HBM each).  One GPU by default; with several, render() spreads the frame's rows over them as
// BaseIntegrator::render spreads them over its worker threads (BaseIntegrator.cpp:143-181).
class SceneInfo {
public:
    explicit SceneInfo(int hip_device = 0) : SceneInfo(std::vector<int>{ hip_device }) {}
    explicit SceneInfo(const std::vector<int>& hip_devices)
    {
        if (vr_group_create(hip_devices.data(), (uint32_t)hip_devices.size(), &group_) != VR_OK)
            throw std::runtime_error(std::string("vr_group_create: ") + vr_group_last_error(nullptr));
    }
    ~SceneInfo() { vr_group_destroy(group_); }
    SceneInfo(const SceneInfo&) = delete;
    SceneInfo& operator=(const SceneInfo&) = delete;

    // densityGrid (+ temperatureGrid): SceneLoader.cpp:207-230
    void setDensity(const vr_grid_desc& density, const vr_grid_desc* temperature = nullptr)
    {
        checkGroup(vr_group_set_density(group_, &density, temperature));
    }
    // scene + primitives + materials: SceneLoader.cpp:69-199
    void setMesh(const vr_mesh_desc& mesh) { checkGroup(vr_group_set_mesh(group_, &mesh)); }

    vr_backend_group* group() const { return group_; }
    uint32_t deviceCount() const { return vr_group_size(group_); }
    // the first GPU's backend: renderPartial, the reference's one-worker call, runs there
    vr_backend* backend() const { return vr_group_backend(group_, 0); }
    void check(int rc) const
    {
        if (rc != VR_OK) throw std::runtime_error(vr_backend_last_error(backend()));
    }
    void checkGroup(int rc) const
    {
        if (rc != VR_OK) throw std::runtime_error(vr_group_last_error(group_));
    }

private:
    vr_backend_group* group_ = nullptr;
};

class BaseIntegrator {
public:
    explicit BaseIntegrator(vr_integrator_id id) : id_(id) {}
    virtual ~BaseIntegrator() = default;

    // BaseIntegrator::renderPartial: writes width * (toHeight - fromHeight) pixels, first row first.
    // `orig` is the camera position the reference passes separately (&options.cameraPosition).
    void renderPartial(const Vec3f* orig, Vec3f* pix, const ThreadInfo* threadInfo, const Options& options, SceneInfo* scene)
    {
        Options o = options;
        if (orig) { o.cameraPosition[0] = orig->x; o.cameraPosition[1] = orig->y; o.cameraPosition[2] = orig->z; }
        scene->check(vr_backend_render_rows(scene->backend(), &o, id_, threadInfo->fromHeight, threadInfo->toHeight,
                                            reinterpret_cast<float*>(pix)));
    }

    // BaseIntegrator::render up to the end of its timed span (BaseIntegrator.cpp:131-193): the
    // float framebuffer before tonemap and save, rows i, i+N, ... on GPU i of the scene's N.  The caller owns
    // `framebuffer` (options.width * options.height pixels), as render() owns its `new Vec3f[]`.
    void render(const Options& options, SceneInfo* scene, Vec3f* framebuffer)
    {
        scene->checkGroup(vr_group_render(scene->group(), &options, id_, reinterpret_cast<float*>(framebuffer)));
    }

    vr_integrator_id id() const { return id_; }

private:
    vr_integrator_id id_;
};

struct IntegratorFactory {
    // integrators/IntegratorFactory.cpp:12-24, restricted to the names on the GPU path
    static std::unique_ptr<BaseIntegrator> GetIntegrator(const std::string& name)
    {
        static const struct { const char* name; vr_integrator_id id; } table[] = {
            { "homogeneousRayMarcherNEE", VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_NEE },
            { "homogeneousRayMarcherImproved", VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_IMPROVED },
            { "ratioTracking", VR_INTEGRATOR_RATIO_TRACKING },
            { "deltaTracking", VR_INTEGRATOR_DELTA_TRACKING },
            { "nanoVDBSimple", VR_INTEGRATOR_NANOVDB_SIMPLE },
            { "nanoVDBEmission", VR_INTEGRATOR_NANOVDB_EMISSION },
            { "heterogeneousPerlinNoise", VR_INTEGRATOR_HETEROGENEOUS_PERLIN_NOISE },
            { "densitySampling", VR_INTEGRATOR_DENSITY_SAMPLING },
            // no counterpart in the reference's factory: the trackers bounded by the mesh
            { "meshRatioTracking", VR_INTEGRATOR_MESH_RATIO_TRACKING },
            { "meshDeltaTracking", VR_INTEGRATOR_MESH_DELTA_TRACKING },
        };
        for (const auto& e : table)
            if (name == e.name) return std::make_unique<BaseIntegrator>(e.id);
        return nullptr;
    }
};

// Options with the defaults of Utils/XMLManager.cpp:39-107
inline Options defaultOptions()
{
    Options o{};
    o.sigma_s = 0.5f; o.sigma_a = 0.5f;
    o.rayPerPixelCount = 1; o.maxDepth = 4;
    o.width = 400; o.widthReference = 400; o.height = 400; o.heightReference = 400;
    o.multiThreadedChunkSize = 20;
    o.fov = 90.0f;
    o.stepSizeMin = 0.01f; o.stepSizeMax = 1.0f; o.stepSizeMultiplier = 1.0f;
    o.lightRayDensityMultiplier = 1.0f; o.shadowRayDensityMultiplier = 1.0f;
    return o;
}

} // namespace vr_host
