"""The reference's dynamic-geometry caller drives the C++ mirror.

tests/golden/transfer_scenes.json is written by tools/refapp/transfer_probe.cpp (build container only):
src/common/PTRTtransfer.cuh compiled IN PLACE from the reference tree with -DUNIFIED_SCENE_ENABLE_PT against
host/ptrt/scene.hpp, running buildPTScene on a UnifiedScene (a `Triangles` sheet, a dynamic cube, a floor, a baked sphere,
two lights) and then twice { the application moves the sheet's triangleVerts, the cube and the camera;
updatePTScene(scene, unified) -- which rewrites mesh->vertices / faces, sets bvhDirty / vertsDirty and ends in
scene.commitObjectChanges() --; updatePTCamera }.  It holds the canonical byte stream (host/ptrt/serialize.hpp) of the
flattened scene after the build and after each step, under the default policy = the reference's host rebuild.  Here the
Python recipe (ptrt_amd.scenes.transfer_demo / transfer_step) must give the same bytes; tests/test_transfer_gpu.py runs the
probe itself on the GPU under every dynamic-geometry policy."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "transfer_scenes.json")))


def same_bytes(got, want, what):
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        pytest.fail(f"{what}: streams differ at byte {first} of {len(want)} (got {len(got)} bytes)")


def test_python_recipe_equals_the_reference_caller_step_by_step(P):
    w, h = GOLD["width"], GOLD["height"]
    assert (w, h) == P.scenes.TRANSFER_SIZE and GOLD["cells"] == P.scenes.TRANSFER_CELLS and len(GOLD["steps"]) == 3
    s = P.Scene(w, h, device=P.HOST_ONLY)
    water, cube = P.scenes.transfer_demo(s)
    for g in GOLD["steps"]:
        if g["step"] > 0:
            P.scenes.transfer_step(s, water, cube, g["step"])  # (a host-only scene commits its host half)
        want = bytes.fromhex(g["hex"])
        assert len(want) == g["bytes"] and g["meshes"] == 4
        same_bytes(s.serialize(), want, f"step {g['step']}")
    assert s.commitCounts() == (0, 0)
    s.close()


def test_the_steps_change_what_they_should():
    """vertices of the sheet, the cube's matrices, the camera: each step's stream differs from the previous one; the face
    count does not (the `Triangles` rewrite keeps 288 faces of three fresh vertices each)."""
    a, b, c = (bytes.fromhex(g["hex"]) for g in GOLD["steps"])
    assert len(a) == len(b) == len(c) and a != b and b != c
    n_mesh = int.from_bytes(a[:4], "little")
    n_vert = int.from_bytes(a[4:8], "little")
    assert n_mesh == 4 and n_vert == 288 * 3
