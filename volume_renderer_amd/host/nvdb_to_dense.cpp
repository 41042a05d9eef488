// nvdb_to_dense — .nvdb -> .vrgrid converter for the reference side.
//
// Does what SceneLoader.cpp:207-230 does with options.densityField — io::readGrid of the grids named "density"
// and "temperature", getExtrema over indexBBox(), tree().bbox() — and writes the result through
// NanoVdbBridge.hpp's describeGrid as dense .vrgrid files (DenseGridFile.hpp) that any host of libvr_backend.so
// can load (Python: volume_renderer_amd/densefile.py) without NanoVDB.
//
// It needs the NanoVDB headers the reference vendors, so it is built where those are, against them:
//
//   g++ -std=c++17 -O2 -I <reference>/VolumeRenderer/RayTracer nvdb_to_dense.cpp -o nvdb_to_dense
//   nvdb_to_dense bunny_cloud.nvdb out_prefix        -> out_prefix.density.vrgrid [+ out_prefix.temperature.vrgrid]
//
// Nothing of it is part of libvr_backend.so.  oracle/Makefile builds it in place for tests/test_nvdb_converter.py.
#include <nanovdb/NanoVDB.h>
#include <nanovdb/util/GridHandle.h>
#include <nanovdb/util/GridStats.h>
#include <nanovdb/util/IO.h>

#include <cstdio>
#include <exception>
#include <string>

#include "DenseGridFile.hpp"
#include "NanoVdbBridge.hpp"

static int convert(const std::string& file, const char* gridName, const std::string& out, bool required)
{
    auto handle = nanovdb::io::readGrid<nanovdb::HostBuffer>(file, gridName);      // SceneLoader.cpp:208-209
    const nanovdb::FloatGrid* grid = handle.grid<float>();                         // :211, :215
    if (!grid) {
        if (required) std::fprintf(stderr, "%s: no float grid named \"%s\"\n", file.c_str(), gridName);
        return required ? 1 : 0;
    }
    const auto extrema = nanovdb::getExtrema(*grid, grid->indexBBox());            // :219, :223
    vr_host::DenseGrid dense;
    vr_host::describeGrid(*grid, grid->tree().bbox(), extrema.max(), dense);       // :227-229
    std::string err;
    if (!vr_host::writeDenseGridFile(out, dense.desc, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::printf("%s: \"%s\" bbox [%d %d %d]..[%d %d %d], max %g, %zu voxels -> %s\n", file.c_str(), gridName,
                dense.desc.bbox_min[0], dense.desc.bbox_min[1], dense.desc.bbox_min[2], dense.desc.bbox_max[0],
                dense.desc.bbox_max[1], dense.desc.bbox_max[2], (double)dense.desc.max_value, dense.voxels.size(), out.c_str());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s <file.nvdb> <output prefix>\n", argv[0]); return 2; }
    try {
        if (int rc = convert(argv[1], "density", std::string(argv[2]) + ".density.vrgrid", true)) return rc;
        return convert(argv[1], "temperature", std::string(argv[2]) + ".temperature.vrgrid", false);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s: %s\n", argv[1], e.what());
        return 1;
    }
}
