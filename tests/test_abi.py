"""CPU: the product library exists, loads, and exports exactly the entry points include/pt_api.h declares.
No compute calls are made here (no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)) - {"pt_status"})


def test_header_declares_the_boundary():
    names = declared_functions()
    for required in ("pt_scene_create", "pt_scene_destroy", "pt_render", "pt_render_device", "pt_render_multi", "pt_intersect", "pt_bsdf_sample",
                     "pt_bsdf_eval", "pt_emission", "pt_curve_eval", "pt_camera_samples", "pt_last_error", "pt_device_info"):
        assert required in names


def test_library_exports_every_declared_symbol(pkg):
    if not os.path.exists(pkg.LIBRARY_PATH):
        pytest.fail("HIP engine not built: run python -c 'import __graft_entry__ as g; g.build()'")
    lib = C.CDLL(pkg.LIBRARY_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted("pt_" + f for f in pkg.api.API_FUNCTIONS) == declared_functions()


def test_oracle_exports_the_same_boundary(pkg, oracle):
    for name in declared_functions():
        if name in ("pt_render_device", "pt_render_multi", "pt_device_count", "pt_device_info", "pt_write_png", "pt_write_exr", "pt_scene_create_tuned", "pt_tuning_default"):
            continue
        assert hasattr(oracle.lib, "ptref_" + name[3:]), name


def test_struct_layouts_match_the_header(pkg):
    """ctypes mirrors vs the C compiler's view of include/pt_api.h."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include "pt_api.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pt_tuning), sizeof(pt_curve), sizeof(pt_texture_layer), sizeof(pt_texstack),
         sizeof(pt_material), sizeof(pt_mesh), sizeof(pt_instance), sizeof(pt_environment), sizeof(pt_camera), sizeof(pt_scene_desc),
         sizeof(pt_render_desc), sizeof(pt_profile), sizeof(pt_hit), sizeof(pt_output_desc), sizeof(pt_compare_stats));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    a = pkg.api
    mine = [C.sizeof(t) for t in (a.Tuning, a.Curve, a.TextureLayer, a.TexStack, a.Material, a.Mesh, a.Instance, a.Environment, a.Camera,
                                  a.SceneDesc, a.RenderDesc, a.Profile, a.Hit, a.OutputDesc, a.CompareStats)]
    assert mine == sizes


def test_scene_file_library_exports_every_declared_symbol(pkg):
    """include/pt_scene_file.h (the TOML front end, libptscene.so)."""
    text = open(os.path.join(ROOT, "include", "pt_scene_file.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))
    assert len(names) >= 18 and "pt_scene_file_load" in names and "pt_config_render_desc" in names
    lib = C.CDLL(pkg.scene_file.LIBRARY_PATH)
    for name in names:
        assert hasattr(lib, name), name
    # the ctypes mirror of pt_render_settings against the C compiler's view
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write('#include <stdio.h>\n#include "pt_scene_file.h"\nint main(void) { printf("%zu\\n", sizeof(pt_render_settings)); return 0; }')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        assert int(subprocess.check_output([os.path.join(d, "t")])) == C.sizeof(pkg.scene_file.RenderSettings)


def test_command_line_renderer_dry_run(pkg):
    """ptcli (src/bin/main.rs): parses config + scene and stops before rendering with --dry-run; fails loudly without a GPU
    otherwise (no CPU fallback)."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.LIBRARY_PATH), "ptcli")
    if not os.path.exists(exe):
        pytest.fail("ptcli not built: run python -c 'import __graft_entry__ as g; g.build()'")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([exe, "--root", pkg.PACKAGE_DIR, "--config", "data/config_two_passes.toml", "--dry-run", "--stdout-log-level", "info", "--output-dir", os.path.join(d, "out")],
                           capture_output=True, text=True, cwd=d)
        assert r.returncode == 0, r.stderr
        assert "7 instances" in r.stdout and "constructing renderer" in r.stdout and "render done" not in r.stdout
        r = subprocess.run([exe, "--root", pkg.PACKAGE_DIR, "--config", "data/no_such_config.toml"], capture_output=True, text=True, cwd=d)
        assert r.returncode == 1 and "couldn't read config.toml" in r.stderr
        r = subprocess.run([exe, "--bogus"], capture_output=True, text=True, cwd=d)
        assert r.returncode == 2


def test_tuning_default_reads_the_environment_once(pkg, monkeypatch):
    """pt_tuning_default (no device needed): the PT_AMD_* variables map onto the struct's fields; unset variables leave the defaults (0 / -1)."""
    lib = pkg.api.Library(pkg.LIBRARY_PATH, "pt_")
    for name in ("PT_AMD_BATCH", "PT_AMD_BLOCKS_PER_CU", "PT_AMD_NO_FUSE", "PT_AMD_PARK_DYNAMIC", "PT_AMD_MULTI_VIRTUAL", "PT_AMD_STAGE_TIMING", "PT_AMD_GENERAL_FORMS"):
        monkeypatch.delenv(name, raising=False)
    t = lib.tuning_default()
    assert (t.flags, t.batch_slots, t.blocks_per_cu, t.park_blocks_per_cu, t.park_dynamic, t.shade_form, t.lds_all_limit, t.multi_virtual) == (0, 0, 0, 0, -1, 0, 0, 0)
    assert list(t.reserved) == [0] * 2 and t.group_evict_below == 0 and t.top_evict_below == 0 and t.light_prepass_max == 0 and t.park_block == 0 and t.walk_evict_below == 0 and t.walk_search_below == 0
    monkeypatch.setenv("PT_AMD_BATCH", "4096"); monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "8"); monkeypatch.setenv("PT_AMD_NO_FUSE", "1")
    monkeypatch.setenv("PT_AMD_PARK_DYNAMIC", "0"); monkeypatch.setenv("PT_AMD_MULTI_VIRTUAL", "4"); monkeypatch.setenv("PT_AMD_STAGE_TIMING", "0")
    monkeypatch.setenv("PT_AMD_GENERAL_FORMS", "1")
    monkeypatch.setenv("PT_AMD_LIGHT_PREPASS_MAX", "5"); monkeypatch.setenv("PT_AMD_TOP_EVICT_BELOW", "24"); monkeypatch.setenv("PT_AMD_GROUP_EVICT_BELOW", "40")
    monkeypatch.setenv("PT_AMD_NO_ONE_LIGHT", "1")
    t = lib.tuning_default()
    assert (t.batch_slots, t.blocks_per_cu, t.park_dynamic, t.multi_virtual) == (4096, 8, 0, 4)
    assert (t.light_prepass_max, t.top_evict_below, t.group_evict_below) == (5, 24, 40)     # (round 5's fields)
    assert t.flags & pkg.api.TUNE_NO_ONE_LIGHT
    t.flags &= ~pkg.api.TUNE_NO_ONE_LIGHT
    assert t.flags == pkg.api.TUNE_NO_FUSE | pkg.api.TUNE_NO_STAGE_TIMING | pkg.api.TUNE_GENERAL_FORMS
