// DenseGridFile.hpp — the ".vrgrid" file: one vr_grid_desc with its voxels, as plain little-endian bytes.
//
// It is the hand-over format between the reference side, which can read .nvdb (nvdb_to_dense.cpp, built
// against the reference's vendored NanoVDB), and hosts that cannot (volume_renderer_amd/densefile.py).
//
//   offset  size  field
//        0     8  magic "VRGRID01"
//        8    12  int32  bbox_min[3]          inclusive index bbox = tree().bbox()           (SceneLoader.cpp:227-229)
//       20    12  int32  bbox_max[3]
//       32    36  float  inv_mat[9]           Map::mInvMatF                                   (NanoVDB.h:2262-2270)
//       68    12  float  vec[3]               Map::mVecF
//       80    36  float  mat[9]               Map::mMatF
//      116     4  float  background           tree().background()
//      120     4  float  max_value            getExtrema(...).max(), negative = not recorded  (SceneLoader.cpp:219,223)
//      124     4  uint32 reserved (0)
//      128     8  uint64 nvoxels              = product of the bbox dimensions
//      136   4*n  float  voxels, k (z) fastest, then j, then i
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vr_types.h"

namespace vr_host {

inline bool writeDenseGridFile(const std::string& path, const vr_grid_desc& d, std::string* err = nullptr)
{
    uint64_t n = 1;
    for (int a = 0; a < 3; ++a) {
        const int64_t e = (int64_t)d.bbox_max[a] - d.bbox_min[a] + 1;
        n *= (uint64_t)(e > 0 ? e : 0);
    }
    unsigned char h[136];
    std::memset(h, 0, sizeof h);
    std::memcpy(h, "VRGRID01", 8);
    std::memcpy(h + 8, d.bbox_min, 12);
    std::memcpy(h + 20, d.bbox_max, 12);
    std::memcpy(h + 32, d.inv_mat, 36);
    std::memcpy(h + 68, d.vec, 12);
    std::memcpy(h + 80, d.mat, 36);
    std::memcpy(h + 116, &d.background, 4);
    std::memcpy(h + 120, &d.max_value, 4);
    std::memcpy(h + 128, &n, 8);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { if (err) *err = "cannot open " + path + " for writing"; return false; }
    bool ok = std::fwrite(h, 1, sizeof h, f) == sizeof h;
    ok = ok && (n == 0 || std::fwrite(d.voxels, sizeof(float), (size_t)n, f) == (size_t)n);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok && err) *err = "short write to " + path;
    return ok;
}

// Reads a .vrgrid; `voxels` owns the data desc.voxels points into.
inline bool readDenseGridFile(const std::string& path, std::vector<float>& voxels, vr_grid_desc& d, std::string* err = nullptr)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    unsigned char h[136];
    uint64_t n = 0, want = 1;
    bool ok = std::fread(h, 1, sizeof h, f) == sizeof h && std::memcmp(h, "VRGRID01", 8) == 0;
    if (ok) {
        std::memcpy(d.bbox_min, h + 8, 12);
        std::memcpy(d.bbox_max, h + 20, 12);
        std::memcpy(d.inv_mat, h + 32, 36);
        std::memcpy(d.vec, h + 68, 12);
        std::memcpy(d.mat, h + 80, 36);
        std::memcpy(&d.background, h + 116, 4);
        std::memcpy(&d.max_value, h + 120, 4);
        std::memcpy(&n, h + 128, 8);
        for (int a = 0; a < 3; ++a) {
            const int64_t e = (int64_t)d.bbox_max[a] - d.bbox_min[a] + 1;
            want *= (uint64_t)(e > 0 ? e : 0);
        }
        ok = n == want;
    }
    if (ok) {
        voxels.resize((size_t)n);
        ok = n == 0 || std::fread(voxels.data(), sizeof(float), (size_t)n, f) == (size_t)n;
        d.voxels = voxels.data();
    }
    std::fclose(f);
    if (!ok && err) *err = path + ": not a .vrgrid file, or truncated";
    return ok;
}

} // namespace vr_host
