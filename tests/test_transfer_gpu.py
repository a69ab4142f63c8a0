"""configs[4]'s path reached from the reference's call sites only.

oracle/_ref/transfer_probe is the reference's own src/common/PTRTtransfer.cuh compiled in place over the C++ mirror
(tools/refapp/transfer_probe.cpp).  On the GPU it builds a scene with buildPTScene, renders, and twice lets the
"application" move a `Triangles` mesh's vertices, a dynamic cube and the camera and call the UNCHANGED sequence
    updatePTScene(scene, unified)   ->  ... mesh->bvhDirty = mesh->vertsDirty = true ... scene.commitObjectChanges()
    updatePTCamera(scene, unified)
then renders again -- under each dynamic-geometry policy (Scene::setDynamicGeometryPolicy: the one line an application
adds).  Checked here: under GpuRefit / GpuRebuild the commits reach ptrt_update_vertices + ptrt_refit / ptrt_build_bvh (the
geometry is uploaded ONCE), and under every policy each frame -- RGB8 and the HDR image -- equals the oracle's rendering of
the scene the probe reports, which in turn equals, byte for byte, the scene the Python recipe builds the same way."""
import os
import struct
import subprocess

import numpy as np
import pytest

from common import bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "transfer_probe")


def read_probe(path):
    raw = open(path, "rb").read()
    magic, w, h, frames, spp, depth = struct.unpack_from("<6I", raw, 0)
    assert magic == 0x54525046
    off, out = 24, []

    def blob():
        nonlocal off
        (n,) = struct.unpack_from("<Q", raw, off)
        off += 8
        b = raw[off:off + n]
        off += n
        return b

    for _ in range(frames):
        gpu_commits, uploads = struct.unpack_from("<2Q", raw, off)
        off += 16
        scene, rgb, accum = blob(), blob(), blob()
        out.append(dict(gpu_commits=gpu_commits, uploads=uploads, scene=scene,
                        rgb8=np.frombuffer(rgb, np.uint8).reshape(-1, 3), accum=np.frombuffer(accum, np.float32).reshape(-1, 3)))
    assert off == len(raw)
    return w, h, spp, depth, out


@pytest.mark.parametrize("policy", ["HostRebuild", "GpuRefit", "GpuRebuild"])
def test_update_pt_scene_then_commit_reaches_the_gpu_path(P, O, blue_noise, tmp_path, policy):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/transfer_probe not built (needs /root/reference: make -C tools/refapp)")
    pid = P.Scene.POLICIES[policy]
    dump = str(tmp_path / f"transfer_{pid}.bin")
    r = subprocess.run([EXE, "gpu", str(pid), dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    w, h, spp, depth, frames = read_probe(dump)
    assert (w, h) == P.scenes.TRANSFER_SIZE and len(frames) == 3

    # the reference's caller reached the refit: two commits on the GPU path, the geometry uploaded once
    if policy == "HostRebuild":
        assert [(f["gpu_commits"], f["uploads"]) for f in frames] == [(0, 1), (0, 2), (0, 3)]
    else:
        assert [(f["gpu_commits"], f["uploads"]) for f in frames] == [(0, 1), (1, 1), (2, 1)]

    # the same sequence through the Python recipe, same policy: same scene bytes at every step, same frames, == oracle
    s = P.Scene(w, h, device=0)
    water, cube = P.scenes.transfer_demo(s)
    s.setDynamicGeometryPolicy(policy)
    s.setPerfSamplesPerPixel(spp)
    s.setMaxBounceDepth(depth)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, w * h)
    for k, f in enumerate(frames):
        if k > 0:
            P.scenes.transfer_step(s, water, cube, k)
        fc = s.getFrameCount()  # (a commit restarts the accumulation: the frame index, and with it the jitter, begins again)
        assert fc == 0
        rgb = s.render_to_host()
        accum = s.read(P.BUF_ACCUM)
        assert s.serialize() == f["scene"], f"step {k}: the probe's scene and the recipe's differ"
        assert np.array_equal(rgb.reshape(-1, 3), f["rgb8"]) and np.array_equal(bits(accum), bits(f["accum"])), f"step {k}"
        c = O.render(s.flatten(), w, h, spp, depth, fc, blue_noise, rng, threads=8)
        assert np.array_equal(bits(c["accum"]), bits(f["accum"])), f"step {k}: the probe's frame differs from the oracle's"
        assert np.array_equal(O.tonemap(c["accum"], w, h, threads=4).reshape(-1, 3), f["rgb8"])
        assert np.array_equal(s.read(P.BUF_RNG), rng)
    assert s.commitCounts() == ((0, 3) if policy == "HostRebuild" else (2, 1))
    assert frames[2]["accum"].any() and not np.array_equal(frames[1]["accum"], frames[2]["accum"])
    s.close()


def test_policy_falls_back_when_the_topology_changes(P, O, blue_noise):
    """A commit whose mesh has a different face count (or a new mesh) takes the reference's route whatever the policy, and the
    policy resumes on the next commit with the new topology."""
    w, h = 96, 64
    s = P.Scene(w, h, device=0)
    water, cube = P.scenes.transfer_demo(s)
    s.setPerfSamplesPerPixel(1)
    s.setMaxBounceDepth(3)
    s.setDenoiserEnabled(False)
    s.setBloomEnabled(False)
    s.initBlueNoise()
    s.uploadToGPU()
    s.setDynamicGeometryPolicy("GpuRefit")  # chosen AFTER the upload: the untouched meshes' face lists are what the device holds
    rng = O.xorwow_init(P.DEFAULT_SEED, 0, w * h)

    def frame(k):
        fc = s.getFrameCount()
        rgb = s.render_to_host()
        c = O.render(s.flatten(), w, h, 1, 3, fc, blue_noise, rng, threads=4)
        assert np.array_equal(bits(s.read(P.BUF_ACCUM)), bits(c["accum"])), k
        assert np.array_equal(O.tonemap(c["accum"], w, h).reshape(-1), rgb.reshape(-1))

    frame(0)
    P.scenes.transfer_step(s, water, cube, 1)
    assert s.commitCounts() == (1, 1)
    frame(1)
    s.setTriangleSoup(water, P.scenes.transfer_sheet(2)[:200])  # fewer triangles: host rebuild + upload
    s.commitObjectChanges()
    assert s.commitCounts() == (1, 2)
    frame(2)
    s.setTriangleSoup(water, P.scenes.transfer_sheet(1)[:200])  # same count again: back on the GPU path
    s.commitObjectChanges()
    assert s.commitCounts() == (2, 2)
    frame(3)
    s.scale(s.addCube(P.Material((0.5, 0.5, 0.5))), (0.3, 0.3, 0.3))  # a new mesh: upload
    s.setTriangleSoup(water, P.scenes.transfer_sheet(2)[:200])
    s.commitObjectChanges()
    assert s.commitCounts() == (2, 3)
    frame(4)
    s.close()
