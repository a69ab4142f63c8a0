"""The reference's own application layer builds the same scenes over the C++ mirror as the Python recipes do.

tests/golden/refapp_scenes.json is written by tools/refapp (build container only): src/pathtracer/app_utils.cuh
compiled IN PLACE from the reference tree against host/ptrt/{scene,view}.hpp -- RenderConfig, the Materials
library, the camera / visualisation controllers, buildSceneById -- run for the scenes that need no OBJ asset.  It
holds the canonical byte stream (host/ptrt/serialize.hpp) of each flattened scene: vertices, faces, BLAS, TLAS,
matrices, the 17 material arrays, lights, camera, sky.  Here the Python recipes (ptrt_amd.scenes.lit_test /
material_matrix) must give the same bytes, and on the GPU those scenes render bit-identically to the oracle."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "refapp_scenes.json")))


def _build(P, key, w=800, h=600, device=None):
    s = P.Scene(w, h, device=P.HOST_ONLY if device is None else device)
    (P.scenes.material_matrix if key == "scene10" else P.scenes.lit_test)(s)
    s.setBVHLeafTarget(12, 5)  # buildSceneById's last call (app_utils.cuh:803)
    return s


@pytest.mark.parametrize("key", ["scene0", "scene10", "scene_default"])
def test_python_recipe_equals_the_reference_builder(P, key):
    g = GOLD[key]
    s = _build(P, key)
    b = s.serialize()
    want = bytes.fromhex(g["hex"])
    assert len(want) == g["bytes"]
    if b != want:  # say where, not just that
        n = min(len(b), len(want))
        first = next((i for i in range(n) if b[i] != want[i]), n)
        pytest.fail(f"{key}: streams differ at byte {first} of {len(want)} (got {len(b)} bytes)")
    s.close()


def test_golden_describes_the_expected_scenes():
    assert GOLD["scene0"]["name"] == "Lit Test Scene" and GOLD["scene0"]["meshes"] == 2 and GOLD["scene0"]["triangles"] == 14
    assert GOLD["scene10"]["name"] == "Material Matrix (Cubes)" and GOLD["scene10"]["meshes"] == 17
    assert GOLD["scene10"]["triangles"] == 2 + 16 * 12 and GOLD["scene10"]["lights"] == 3
    assert GOLD["scene_default"]["hex"] == GOLD["scene0"]["hex"]  # invalid id -> createLitTestScene


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["scene0", "scene10"])
def test_reference_app_scenes_render_like_the_oracle(P, O, key):
    """Scene 10 has one cube per material class of the path (metal, clearcoat, glass, thin film, sheen, emitter...)."""
    from common import assert_frames_equal, render_both
    s = _build(P, key, 160, 120, device=0)
    assert s.serialize()[:4] == bytes.fromhex(GOLD[key]["hex"])[:4]
    gpu, cpu = render_both(P, O, s, P.blue_noise_table(), spp=4, depth=6, frames=2)
    assert_frames_equal(gpu, cpu)
    assert np.asarray(gpu[1]["accum"]).any()
    s.close()
