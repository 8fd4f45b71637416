"""
The host side of the library under the address and undefined-behaviour sanitizers (host code only;
the GPU side cannot be sanitized on this hardware pool).  tests/native/check_host.hip includes the
library's source, feeds the package's flat scenes (bunny, cloth + textured bunny, cow + teapot +
sphere field, the 484-sphere cover field) and synthetic ones (empty, every remainder of the loops'
unrolling, 33 large groups) through the host halves of wrt_add_* and wrt_scene_commit, and replays
the record fetches of the group loops index by index: every fetch, the one past a group's last record
included, must lie inside its array.
"""
import contextlib
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import helpers

SANITIZE = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
            "-Xarch_host", "-fno-sanitize-recover=undefined"]


def build_with_sanitizers(source, exe):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.fail("hipcc not found: the host side cannot be checked")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-ffp-contract=off", "-std=c++17"] + SANITIZE +
                           [os.path.join(helpers.REPO, "tests", "native", source), "-o", str(exe)],
                           capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stderr[-3000:]


def run_sanitized(exe, args=()):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)


def write_scene(path, flat):
    """The flat scene's groups in the layout check_host.hip reads."""
    kinds = {"spheres": 0, "triangles": 1, "disk": 2}
    with open(path, "wb") as fh:
        np.array([0x53545257, len(flat["groups"])], dtype=np.int32).tofile(fh)
        for g in flat["groups"]:
            if g["type"] == "spheres":
                n = len(g["radii"])
                np.array([kinds["spheres"], n, 0], dtype=np.int32).tofile(fh)
                np.ascontiguousarray(g["centres"], dtype=np.single).tofile(fh)
                np.ascontiguousarray(g["radii"], dtype=np.single).tofile(fh)
                np.ascontiguousarray(g["materials"], dtype=np.ubyte).tofile(fh)
            elif g["type"] == "triangles":
                n = len(g["pt0"])
                np.array([kinds["triangles"], n, int(g["material_id"])], dtype=np.int32).tofile(fh)
                for k in ("pt0", "A", "B", "n0", "n1", "n2", "uv0", "uv1", "uv2"):
                    np.ascontiguousarray(g[k], dtype=np.single).tofile(fh)
                np.ascontiguousarray(g["bounds_centre"], dtype=np.single).tofile(fh)
                np.array([g["bounds_radius"]], dtype=np.single).tofile(fh)
            else:
                np.array([kinds["disk"], 1, int(g["material_index"])], dtype=np.int32).tofile(fh)
                for k in ("centre", "normal", "U", "V"):
                    np.ascontiguousarray(g[k], dtype=np.single).tofile(fh)
                np.array([g["radius"]], dtype=np.double).tofile(fh)


def test_scene_staging_and_packing_under_sanitizers(tmp_path):
    from weekend_raytracer import _flatten, main, scenes
    files = []
    with contextlib.redirect_stdout(io.StringIO()):
        built = {"bunny": scenes.bunny_diffuse_scene(16 / 9), "cloth_bunny": scenes.cloth_bunny_scene(16 / 9),
                 "cow_teapot_field": scenes.cow_teapot_field_scene(16 / 9),
                 "cover484": scenes.cover_scene(16 / 9, scenes.sphere_field_records(480)),
                 "disks": scenes.disk_test_scene(16 / 9)}
    for name, objs in built.items():
        files.append(tmp_path / (name + ".bin"))
        write_scene(files[-1], _flatten.flatten(*objs, main.HORIZON_COLOUR, main.SKY_COLOUR))
    exe = tmp_path / "check_host"
    build_with_sanitizers("check_host.hip", exe)
    run = run_sanitized(exe, files)
    assert run.returncode == 0, (run.stdout[-3000:], run.stderr[-3000:])
    assert "host check passed" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr
    for name in built:
        assert name + ".bin" in run.stdout


def test_hierarchy_builder_under_sanitizers(tmp_path):
    """tests/native/check_tree.hip (the structure of every hierarchy the builder makes) with the same sanitizers."""
    exe = tmp_path / "check_tree"
    build_with_sanitizers("check_tree.hip", exe)
    run = run_sanitized(exe)
    assert run.returncode == 0, (run.stdout[-3000:], run.stderr[-3000:])
    assert "tree check passed" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr
