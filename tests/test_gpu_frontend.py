"""A reference-schema config file end to end on the GPU: XML -> OBJ / .vrgrid -> vr_backend_render -> tonemap -> PNG,
against the oracle rendering the same scene through its own structures (exact mode: the PNG bytes' pixels must match)."""
import ctypes as C
import os

import numpy as np
import pytest

import frontend_cases as fc
import oracle_lib as ol
import scenes
from volume_renderer_amd import config, densefile, objloader, render_config
from volume_renderer_amd.abi import INTEGRATORS, Grid

pytestmark = pytest.mark.gpu

MARCHER_XML = """<?xml version="1.0"?>
<options>
  <models><model baseDir="models\\">cube_two_materials.obj</model></models>
  <sigma_a>0.5</sigma_a>
  <rayPerPixelCount>4</rayPerPixelCount>
  <maxDepth>16</maxDepth>
  <integrator>{integrator}</integrator>
  <width reference="96">96</width>
  <height startOffset="8" reference="80">64</height>
  <camera x="0" y="0.825" z="6.75" pitch="0" yaw="0" roll="0" fov="60"/>
  <backgroundColor r="0.572" g="0.772" b="0.921"/>
  <light x="0" y="1" z="0" r="19.65" g="4.5" b="13.5" multiplier="0.1"/>
  <heyneygreenstein_g>0.4</heyneygreenstein_g>
</options>
"""

TRACKER_XML = """<?xml version="1.0"?>
<options>
  <models></models>
  <densityField baseDir="fields\\">puff.nvdb</densityField>
  <sigma_a>0.02</sigma_a>
  <sigma_s>0.2</sigma_s>
  <heyneygreenstein_g>0.4</heyneygreenstein_g>
  <rayPerPixelCount>4</rayPerPixelCount>
  <stepSize min="0.01" max="999999.9" multiplier="0.5"/>
  <maxDepth>16</maxDepth>
  <integrator>ratioTracking</integrator>
  <width startOffset="10" reference="80">60</width>
  <height reference="48">48</height>
  <camera x="0" y="0" z="70" fov="45"/>
  <backgroundColor r="1.0" g="1.0" b="1.0" multiplier="0.25"/>
  <light x="0" y="-1" z="0" r="0.933" g="0.34" b="0.16" multiplier="40.0"/>
  <emission r="0.015" g="0.247" b="0.596" multiplier="6.0"/>
  <shadowRayDensityMultiplier>4.0</shadowRayDensityMultiplier>
</options>
"""


def oracle_png_pixels(oracle, scene, cfg):
    o = cfg.options
    oracle.vro_set_math_mode(ol.MATH_SHARED)
    try:
        img = ol.render(scene, o, INTEGRATORS[cfg.integrator], rng_mode=ol.RNG_PHILOX, seed=scenes.SEED)
    finally:
        oracle.vro_set_math_mode(ol.MATH_LIBM)
    img = np.ascontiguousarray(img)
    oracle.vro_tonemap_reinhard_extended(ol.fptr(img), o.width * o.height)
    out = np.zeros((o.height, o.width, 3), dtype=np.uint8)
    oracle.vro_to_rgb8(ol.fptr(img), o.width, o.height, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


@pytest.mark.parametrize("integrator", ["homogeneousRayMarcherNEE", "homogeneousRayMarcherImproved", "densitySampling"])
def test_marcher_config_end_to_end(oracle, tmp_path, integrator):
    os.makedirs(tmp_path / "models")
    fc.write_case(str(tmp_path / "models"), "cube_two_materials", fc.OBJ_CASES["cube_two_materials"])
    xml = tmp_path / "configRenderer.xml"
    xml.write_text(MARCHER_XML.format(integrator=integrator))
    out = render_config.main([str(xml), "--mode", "exact", "--out", str(tmp_path / "out.png"), "--quiet"])
    got = render_config.read_png_rgb8(out)

    cfg = config.load_renderer_options(str(xml))
    mesh = objloader.load_obj(cfg.resolve(cfg.models[0]), str(tmp_path / "models") + "/").to_mesh()
    sc, _keep = ol.make_scene(mesh=mesh)
    want = oracle_png_pixels(oracle, sc, cfg)
    assert got.shape == (64, 96, 3) and np.array_equal(got, want)
    assert len(np.unique(got.reshape(-1, 3), axis=0)) > 8             # a picture, not a flat background


def test_tracker_config_with_converted_grids(oracle, tmp_path):
    """<densityField> names an .nvdb; the .density/.temperature .vrgrid files beside it (what nvdb_to_dense writes) are used."""
    n = 40
    vox = np.zeros((n, n, n), dtype=np.float32)
    oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
    dens = Grid(vox, voxel_size=1.0)
    temp = Grid((vox * 2.0).astype(np.float32)[4:30, 2:38, 5:33].copy(), bbox_min=(-16, -18, -15), voxel_size=1.0)
    os.makedirs(tmp_path / "fields")
    densefile.save_dense_grid(str(tmp_path / "fields" / "puff.density.vrgrid"), dens)
    densefile.save_dense_grid(str(tmp_path / "fields" / "puff.temperature.vrgrid"), temp)
    xml = tmp_path / "explosion-configRenderer.xml"
    xml.write_text(TRACKER_XML)
    out = render_config.main([str(xml), "--mode", "exact_skip", "--out-dir", str(tmp_path), "--quiet"])
    assert os.path.basename(out).endswith("_ratioTracking_area10_0_ray4_boun16.png")
    got = render_config.read_png_rgb8(out)

    cfg = config.load_renderer_options(str(xml))
    sc, _keep = ol.make_scene(density=dens, temperature=temp)
    want = oracle_png_pixels(oracle, sc, cfg)
    assert got.shape == (48, 60, 3) and np.array_equal(got, want)
    assert len(np.unique(got.reshape(-1, 3), axis=0)) > 8


def test_missing_converted_grid_is_reported(tmp_path):
    xml = tmp_path / "c.xml"
    xml.write_text(TRACKER_XML)
    with pytest.raises(FileNotFoundError, match="nvdb_to_dense"):
        render_config.main([str(xml), "--quiet", "--out", str(tmp_path / "o.png")])
