#!/usr/bin/env python3
"""Writes tests/golden/ref_probe.json from oracle/_ref/ref_probe -- the curand-free part of the
reference (blue-noise generator, TAA jitter, Light/vec3/Ray layout) compiled from the reference's
own sources in this container by `make -C oracle ref` (see oracle/ref_probe.cpp).  Data only: the
table is stored as its SHA-256 plus the first 64 values, everything else as float bit patterns."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def run_probe():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    doc = json.loads(subprocess.check_output([exe]))
    bn = np.array(doc["blue_noise"].pop("bits"), dtype=np.uint32)
    doc["blue_noise"]["sha256"] = hashlib.sha256(bn.tobytes()).hexdigest()
    doc["blue_noise"]["first64_bits"] = [int(v) for v in bn[:64]]
    doc["built_from"] = ("common/bluenoise.cuh, pathtracer/rendering/taa.cuh, pathtracer/scene/lights.cuh, "
                         "common/vec3.cuh, common/ray.cuh, common/matrix.cuh, common/vec4.cuh, common/triangle.cuh of the reference; g++ -O2 -ffp-contract=off, libstdc++")
    return doc


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    json.dump(run_probe(), open(os.path.join(HERE, "ref_probe.json"), "w"), indent=1)
    print("wrote ref_probe.json", file=sys.stderr)
