// NanoVdbBridge.hpp — reference-side scene hand-over: NanoVDB grid / tinyobj mesh -> the C-ABI's dense
// descriptions (".nvdb -> bricked grid converter on the host").
//
// The reference keeps its fields as nanovdb::FloatGrid (SceneLoader.cpp:207-230: io::readGrid,
// getExtrema, tree().bbox()) and its meshes as tinyobj shapes (SceneLoader.cpp:69-175).  This header
// turns them into vr_grid_desc / vr_mesh_desc; the backend then re-bricks the voxels on the device.
// It is header-only and includes NO NanoVDB or tinyobj header itself: include the reference's own
// <nanovdb/NanoVDB.h> (and tiny_obj_loader.h) first, as SceneLoader.cpp does, then this file.  Nothing
// here is compiled into libvr_backend.so; oracle/ref_kat.cpp instantiates it against the reference's
// vendored NanoVDB to test it (tests/test_nanovdb_bridge.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <vector>

#include "../../include/vr_types.h"

namespace vr_host {

// Owns the dense copy a vr_grid_desc points into.
struct DenseGrid {
    std::vector<float> voxels;
    vr_grid_desc       desc{};
};

// grid     : the FloatGrid of SceneInfo::densityGrid / temperatureGrid            (Types.h:141-142)
// bbox     : SceneInfo::gridBoundingBox = grid->tree().bbox()                     (SceneLoader.cpp:227-229)
// max_value: SceneInfo::densityExtrema.max() from nanovdb::getExtrema             (SceneLoader.cpp:219,223);
//            pass a negative number to let the backend compute it from the voxels.
// Every voxel of the bbox is read through a ReadAccessor, so inactive voxels and tiles come out as
// the values getValue() returns for them (NanoVDB.h:5248-5263) — what the integrators see.
template <typename GridT, typename BBoxT>
inline void describeGrid(const GridT& grid, const BBoxT& bbox, float max_value, DenseGrid& out)
{
    using CoordT = typename std::decay<decltype(bbox.min())>::type;
    vr_grid_desc& d = out.desc;
    int64_t n[3];
    for (int a = 0; a < 3; ++a) {
        d.bbox_min[a] = bbox.min()[a];
        d.bbox_max[a] = bbox.max()[a];
        n[a] = (int64_t)bbox.max()[a] - bbox.min()[a] + 1;
        if (n[a] < 0) n[a] = 0;
    }
    out.voxels.assign((size_t)(n[0] * n[1] * n[2]), 0.0f);
    auto   acc = grid.tree().getAccessor();
    size_t o = 0;
    for (int i = d.bbox_min[0]; i <= d.bbox_max[0]; ++i)              // k (z) fastest, as vr_grid_desc wants
        for (int j = d.bbox_min[1]; j <= d.bbox_max[1]; ++j)
            for (int k = d.bbox_min[2]; k <= d.bbox_max[2]; ++k) out.voxels[o++] = acc.getValue(CoordT(i, j, k));
    d.voxels = out.voxels.data();
    // index = mInvMatF * (world - mVecF), world = mMatF * index + mVecF: the members Map::applyInverseMapF /
    // applyMapF read (NanoVDB.h:2262-2270, :2249)
    const auto& m = grid.map();
    for (int a = 0; a < 9; ++a) { d.inv_mat[a] = m.mInvMatF[a]; d.mat[a] = m.mMatF[a]; }
    for (int a = 0; a < 3; ++a) d.vec[a] = m.mVecF[a];
    d.background = grid.tree().background();
    d.max_value = max_value;
}

// Owns the arrays a vr_mesh_desc points into.
struct TriangleSoup {
    std::vector<float>    verts;
    std::vector<uint32_t> tri_idx, prim_to_shape;
    std::vector<float>    shape_dissolve, shape_diffuse;
    vr_mesh_desc          desc{};
};

// attrib / shapes / materials: what tinyobj::LoadObj filled for one model (SceneLoader.cpp:74-79).
// Same buffers as the reference hands Embree (SceneLoader.cpp:150-175): the vertex array as loaded,
// one index triple per face in shape order, so the hit's primitive id is the global face index;
// prim_to_shape is the `primitives` table the integrators index with it (SceneLoader.cpp:146,
// IntegratorHomogeneousRayMarcherNEE.cpp:33), shape_dissolve the
// `materials[shapes[s].mesh.material_ids[0]].dissolve` they read next (:49-52).  Faces must be
// triangles (LoadObj triangulates by default).
template <typename AttribT, typename ShapesT, typename MaterialsT>
inline void describeMesh(const AttribT& attrib, const ShapesT& shapes, const MaterialsT& materials, TriangleSoup& out)
{
    out.verts.assign(attrib.vertices.begin(), attrib.vertices.end());
    out.tri_idx.clear(); out.prim_to_shape.clear(); out.shape_dissolve.clear(); out.shape_diffuse.clear();
    uint32_t s = 0;
    for (const auto& shape : shapes) {
        const auto& mesh = shape.mesh;
        for (size_t f = 0; f + 2 < mesh.indices.size(); f += 3) {
            for (int c = 0; c < 3; ++c) out.tri_idx.push_back((uint32_t)mesh.indices[f + c].vertex_index);
            out.prim_to_shape.push_back(s);
        }
        const int mat = mesh.material_ids.empty() ? -1 : mesh.material_ids[0];
        const bool has_mat = mat >= 0 && (size_t)mat < materials.size();
        out.shape_dissolve.push_back(has_mat ? (float)materials[(size_t)mat].dissolve : 1.0f);
        for (int c = 0; c < 3; ++c) out.shape_diffuse.push_back(has_mat ? (float)materials[(size_t)mat].diffuse[c] : 0.0f);
        ++s;
    }
    out.desc.verts_xyz = out.verts.data();
    out.desc.nverts = (uint32_t)(out.verts.size() / 3);
    out.desc.tri_idx = out.tri_idx.data();
    out.desc.ntris = (uint32_t)(out.tri_idx.size() / 3);
    out.desc.prim_to_shape = out.prim_to_shape.data();
    out.desc.shape_dissolve = out.shape_dissolve.data();
    out.desc.shape_diffuse = out.shape_diffuse.data();
    out.desc.nshapes = (uint32_t)out.shape_dissolve.size();
}

} // namespace vr_host
