"""The box hierarchy wrt_scene_commit builds, checked on the host alone (tests/native/check_tree.hip
includes the library's source and calls its builder): every index a walk can follow stays inside the
array, every triangle is in exactly one leaf, boxes hold their triangles, and the height fits the
walk's stack, for clouds, degenerate layouts and the size limits."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tree_structure(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.fail("hipcc not found: the hierarchy builder cannot be checked")
    exe = tmp_path / "check_tree"
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-ffp-contract=off", "-std=c++17",
                            os.path.join(ROOT, "tests", "native", "check_tree.hip"), "-o", str(exe)],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-3000:]
    assert "tree check passed" in run.stdout
