/*
 * ref_kat.cpp — known-answer driver over the REFERENCE's own code (TEST INFRASTRUCTURE).
 *
 * This file contains no restatement: every function below calls straight into the
 * reference's translation units (Utils.cpp, ONB.cpp, PhaseFunction.cpp, DirectionSampler.cpp,
 * RandomGenerator.cpp, PerlinNoiseSampler.cpp, MultithreadingHelper.cpp) and its vendored
 * header-only NanoVDB (Ray.h, NanoVDB.h, GridBuilder.h, GridStats.h), compiled where they
 * lie under /root/reference by oracle/Makefile into oracle/_ref/libvr_refkat.so.
 * tests/golden/make_golden.py calls it to produce the committed golden vectors that pin
 * oracle/vr_oracle.c; nothing on the GPU box needs it.
 *
 * Not covered (unbuildable here): the integrators' castRay and
 * BaseIntegrator (they include <Windows.h> / Process.h and need Embree + FreeImage to link).
 */
#include "Utils.h"
#include "ONB.h"
#include "PhaseFunction.h"
#include "DirectionSampler.h"
#include "RandomGenerator.h"
#include "PerlinNoiseSampler.h"
#include "MultithreadingHelper.h"
#include "nanovdb/util/GridBuilder.h"
#include "nanovdb/util/IO.h"
#include <fstream>
#include "tinyobjloader/tiny_obj_loader.h"   // declarations; the implementation TU is compiled in place (Makefile)
#include "pugixml-1.9/src/pugixml.hpp"       // the reference's XML library; pugixml.cpp is compiled in place too

#include <cstring>

#include "../volume_renderer_amd/host/NanoVdbBridge.hpp"   // OURS, instantiated on the reference's types

float noise(float x, float y, float z); // defined (non-static) in Utils/PerlinNoiseSampler.cpp:46

namespace {
struct RefGrid {
    nanovdb::GridHandle<nanovdb::HostBuffer> handle;
    nanovdb::FloatGrid*                      grid = nullptr;
    nanovdb::CoordBBox                       bbox;
    nanovdb::Extrema<float>                  extrema;
};
inline void ray_export(const nanovdb::Ray<float>& r, float* out)
{
    for (int i = 0; i < 3; ++i) { out[i] = r.eye()[i]; out[3 + i] = r.dir()[i]; }
    out[6] = r.t0(); out[7] = r.t1();
}
} // namespace

extern "C" {

void refk_normalize(const float* v, float* out)
{
    Vec3f r = Utils::normalize(Vec3f(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float refk_clamp(float lo, float hi, float v) { return Utils::clamp(lo, hi, v); }
float refk_deg2rad(float d) { return Utils::deg2rad(d); }
float refk_dot(const float* a, const float* b) { return Utils::dotProduct(Vec3f(a[0], a[1], a[2]), Vec3f(b[0], b[1], b[2])); }
void refk_cross(const float* a, const float* b, float* out)
{
    Vec3f r = Utils::crossProduct(Vec3f(a[0], a[1], a[2]), Vec3f(b[0], b[1], b[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void refk_rotate(float pitch, float roll, float yaw, float* p)
{
    Vec3f v(p[0], p[1], p[2]);
    Utils::rotate(pitch, roll, yaw, &v);
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
}
float refk_henyey_greenstein(float g, float c) { return PhaseFunction::henyey_greenstein(g, c); }

void refk_get_floats(unsigned seed, float lo, float hi, int n, float* out)
{
    RandomGenerator gen(seed);
    for (int i = 0; i < n; ++i) out[i] = gen.getFloat(lo, hi);
}
/* sampleHenyeyGreenstein with the reference's generator; out_xi receives the two uniforms
 * that generator hands out for this seed (same calls, same order: DirectionSampler.cpp:43,58) */
void refk_sample_hg(float g, const float* dir, unsigned seed, float* out_dir, float* out_xi)
{
    RandomGenerator gen(seed), replay(seed);
    DirectionSampler sampler;
    Vec3f r = sampler.sampleHenyeyGreenstein(g, Vec3f(dir[0], dir[1], dir[2]), &gen);
    out_dir[0] = r.x; out_dir[1] = r.y; out_dir[2] = r.z;
    out_xi[0] = replay.getFloat(0.0f, 0.9999f);
    out_xi[1] = replay.getFloat(0.0f, 0.9999f);
}
float refk_perlin_noise(float x, float y, float z)
{
    PerlinNoiseSampler::getInstance(); // fills the doubled permutation table
    return noise(x, y, z);
}
float refk_perlin_eval_density(const float* p)
{
    return PerlinNoiseSampler::getInstance()->eval_density(Vec3f(p[0], p[1], p[2]));
}
int refk_chunks(uint32_t chunkSize, uint32_t chunkCount, uint32_t* out_offsets, int cap)
{
    MultithreadingHelper h(chunkSize, chunkCount);
    int n = 0;
    uint32_t off;
    while (h.tryReservingChunk(off)) { if (n < cap) out_offsets[n] = off; ++n; }
    return n;
}

/* ---- NanoVDB: build a FloatGrid from a dense box, the way SceneLoader.cpp:207-230 reads one ---- */
void* refk_grid_create_named(const float* dense, const int* bmin, const int* bmax, double voxel_size,
                             const double* origin, float background, const char* name);
void* refk_grid_create(const float* dense, const int* bmin, const int* bmax, double voxel_size,
                       const double* origin, float background)
{
    return refk_grid_create_named(dense, bmin, bmax, voxel_size, origin, background, "density");
}
/* the same with a grid name ("density" / "temperature": the names SceneLoader.cpp:208-209 asks readGrid for) */
void* refk_grid_create_named(const float* dense, const int* bmin, const int* bmax, double voxel_size,
                             const double* origin, float background, const char* name)
{
    auto* g = new RefGrid;
    const int ny = bmax[1] - bmin[1] + 1, nz = bmax[2] - bmin[2] + 1;
    nanovdb::GridBuilder<float> builder(background, nanovdb::GridClass::FogVolume);
    // voxels equal to the background stay inactive, as GridBuilder::operator() leaves them
    // (GridBuilder.h:224-226); that operator's static_assert does not compile under C++20.
    auto acc = builder.getAccessor();
    for (int i = bmin[0]; i <= bmax[0]; ++i)
        for (int j = bmin[1]; j <= bmax[1]; ++j)
            for (int k = bmin[2]; k <= bmax[2]; ++k) {
                const float v = dense[((size_t)(i - bmin[0]) * ny + (j - bmin[1])) * nz + (k - bmin[2])];
                if (v != background) acc.setValue(nanovdb::Coord(i, j, k), v);
            }
    g->handle = builder.getHandle<>(voxel_size, nanovdb::Vec3d(origin[0], origin[1], origin[2]), name);
    g->grid = g->handle.grid<float>();
    g->bbox = g->grid->tree().bbox();                          // SceneLoader.cpp:227-229
    g->extrema = nanovdb::getExtrema(*g->grid, g->grid->indexBBox()); // SceneLoader.cpp:219
    return g;
}
void refk_grid_info(void* h, int* bmin, int* bmax, float* minmax, float* inv_mat9, float* vec3, float* background)
{
    auto* g = static_cast<RefGrid*>(h);
    for (int i = 0; i < 3; ++i) { bmin[i] = g->bbox.min()[i]; bmax[i] = g->bbox.max()[i]; }
    minmax[0] = g->extrema.min(); minmax[1] = g->extrema.max();
    const nanovdb::Map& m = g->grid->map();
    for (int i = 0; i < 9; ++i) inv_mat9[i] = m.mInvMatF[i];
    for (int i = 0; i < 3; ++i) vec3[i] = m.mVecF[i];
    *background = g->grid->tree().background();
}
void refk_grid_mat(void* h, float* mat9)
{
    const nanovdb::Map& m = static_cast<RefGrid*>(h)->grid->map();
    for (int i = 0; i < 9; ++i) mat9[i] = m.mMatF[i];
}
/* indexToWorldF as IntegratorNanoVDBSimple.cpp:41 calls it */
void refk_index_to_world(void* h, const float* p, int n, float* out)
{
    auto* g = static_cast<RefGrid*>(h);
    for (int i = 0; i < n; ++i) {
        auto w = g->grid->indexToWorldF(nanovdb::Vec3<float>(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
        out[3 * i] = w[0]; out[3 * i + 1] = w[1]; out[3 * i + 2] = w[2];
    }
}
float refk_grid_value(void* h, const float* p)
{
    auto* g = static_cast<RefGrid*>(h);
    auto acc = g->grid->tree().getAccessor();                  // IntegratorDeltaTracking.cpp:25
    return acc.getValue(nanovdb::Coord::Floor(nanovdb::Vec3<float>(p[0], p[1], p[2]))); // :55
}
void refk_grid_values(void* h, const float* p, int n, float* out)
{
    auto* g = static_cast<RefGrid*>(h);
    auto acc = g->grid->tree().getAccessor();
    for (int i = 0; i < n; ++i)
        out[i] = acc.getValue(nanovdb::Coord::Floor(nanovdb::Vec3<float>(p[3 * i], p[3 * i + 1], p[3 * i + 2])));
}
int refk_world_ray_clip(void* h, const float* eye, const float* dir, float* out8)
{
    auto* g = static_cast<RefGrid*>(h);
    nanovdb::Vec3<float> e(eye[0], eye[1], eye[2]), d(dir[0], dir[1], dir[2]);
    nanovdb::Ray<float> wRay(e, d);                            // IntegratorDeltaTracking.cpp:33
    nanovdb::Ray<float> iRay = wRay.worldToIndexF(*g->grid);   // :35
    bool hit = iRay.clip(g->bbox);                             // :37
    ray_export(iRay, out8);
    return hit ? 1 : 0;
}
int refk_index_ray_clip(void* h, const float* eye, const float* dir, float* out8)
{
    auto* g = static_cast<RefGrid*>(h);
    nanovdb::Vec3<float> e(eye[0], eye[1], eye[2]), d(dir[0], dir[1], dir[2]);
    nanovdb::Ray<float> iRay(e, d);                            // IntegratorDeltaTracking.cpp:147
    bool hit = iRay.clip(g->bbox);                             // :150
    ray_export(iRay, out8);
    return hit ? 1 : 0;
}
void refk_ray_at(const float* eye, const float* dir, float t, float* out)
{
    nanovdb::Ray<float> r(nanovdb::Vec3<float>(eye[0], eye[1], eye[2]), nanovdb::Vec3<float>(dir[0], dir[1], dir[2]));
    auto p = r(t);
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}
void refk_grid_destroy(void* h) { delete static_cast<RefGrid*>(h); }
/* nanovdb::io::writeGrid of one or two grids into one .nvdb (uncompressed), the file format SceneLoader.cpp:208 reads */
int refk_grid_write(const char* path, void* h0, void* h1)
{
    try {
        std::ofstream os(path, std::ios::out | std::ios::binary | std::ios::trunc);
        if (!os.is_open()) return 1;
        nanovdb::io::writeGrid(os, static_cast<RefGrid*>(h0)->handle);
        if (h1) nanovdb::io::writeGrid(os, static_cast<RefGrid*>(h1)->handle);
        return os.good() ? 0 : 1;
    } catch (const std::exception&) { return 2; }
}

/* ---- our reference-side bridge (volume_renderer_amd/host/NanoVdbBridge.hpp) on the reference's types ---- */
/* describeGrid on the grid above; voxels may be NULL to query the size first */
long long refk_bridge_grid(void* h, int* bmin, int* bmax, float* inv_mat9, float* vec3, float* background_max2,
                           float* voxels, long long capacity, float* mat9)
{
    auto* g = static_cast<RefGrid*>(h);
    vr_host::DenseGrid dg;
    vr_host::describeGrid(*g->grid, g->bbox, g->extrema.max(), dg);
    for (int i = 0; i < 3; ++i) { bmin[i] = dg.desc.bbox_min[i]; bmax[i] = dg.desc.bbox_max[i]; vec3[i] = dg.desc.vec[i]; }
    for (int i = 0; i < 9; ++i) { inv_mat9[i] = dg.desc.inv_mat[i]; mat9[i] = dg.desc.mat[i]; }
    background_max2[0] = dg.desc.background; background_max2[1] = dg.desc.max_value;
    const long long n = (long long)dg.voxels.size();
    if (voxels && capacity >= n) memcpy(voxels, dg.voxels.data(), (size_t)n * sizeof(float));
    return n;
}
/* tinyobj::LoadObj as SceneLoader.cpp:79 calls it, then describeMesh; counts[3] = nverts, ntris, nshapes;
 * arrays may be NULL to query the counts first.  Returns 0 on success. */
int refk_bridge_obj(const char* obj_path, const char* base_dir, unsigned* counts, float* verts, unsigned* tri_idx,
                    unsigned* prim_to_shape, float* dissolve, float* diffuse)
{
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, obj_path, base_dir)) return 1;
    vr_host::TriangleSoup soup;
    vr_host::describeMesh(attrib, shapes, materials, soup);
    counts[0] = soup.desc.nverts; counts[1] = soup.desc.ntris; counts[2] = soup.desc.nshapes;
    if (verts) memcpy(verts, soup.verts.data(), soup.verts.size() * sizeof(float));
    if (tri_idx) memcpy(tri_idx, soup.tri_idx.data(), soup.tri_idx.size() * sizeof(unsigned));
    if (prim_to_shape) memcpy(prim_to_shape, soup.prim_to_shape.data(), soup.prim_to_shape.size() * sizeof(unsigned));
    if (dissolve) memcpy(dissolve, soup.shape_dissolve.data(), soup.shape_dissolve.size() * sizeof(float));
    if (diffuse) memcpy(diffuse, soup.shape_diffuse.data(), soup.shape_diffuse.size() * sizeof(float));
    return 0;
}

/* ---- pugixml conversions as Utils/XMLManager.cpp:16-107 uses them: root = doc.child("options"), then
 *      root.child(child).text() or .attribute(attr), converted with as_float/as_int/as_uint/as_bool(default).
 *      kind: 0 float, 1 int, 2 uint, 3 bool.  *present = 0 when the document does not parse. ---- */
double refk_xml_number(const char* xml, const char* child, const char* attr, int kind, double def, int* present)
{
    pugi::xml_document doc;
    *present = doc.load_string(xml) ? 1 : 0;
    pugi::xml_node n = doc.child("options").child(child);
    if (attr) {
        pugi::xml_attribute a = n.attribute(attr);
        switch (kind) {
        case 0: { float f = a.as_float((float)def); return f; }
        case 1: return a.as_int((int)def);
        case 2: return a.as_uint((unsigned)def);
        default: return a.as_bool(def != 0) ? 1 : 0;
        }
    }
    pugi::xml_text t = n.text();
    switch (kind) {
    case 0: { float f = t.as_float((float)def); return f; }
    case 1: return t.as_int((int)def);
    case 2: return t.as_uint((unsigned)def);
    default: return t.as_bool(def != 0) ? 1 : 0;
    }
}
/* as_string of the same node/attribute; index selects the index-th <model> under <models> when child == "model" */
int refk_xml_string(const char* xml, const char* child, const char* attr, int index, char* out, int cap)
{
    pugi::xml_document doc;
    if (!doc.load_string(xml)) return -1;
    pugi::xml_node root = doc.child("options"), n;
    if (strcmp(child, "model") == 0) {
        int i = 0;
        for (auto m : root.child("models").children("model")) { if (i++ == index) n = m; }
        if (!n && index >= 0) return -2;
    } else n = root.child(child);
    const char* s = attr ? n.attribute(attr).as_string() : n.text().as_string();
    snprintf(out, (size_t)cap, "%s", s);
    return (int)strlen(s);
}

} // extern "C"
