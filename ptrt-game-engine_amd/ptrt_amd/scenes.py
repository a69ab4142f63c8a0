"""Scene recipes for the measured configurations, written against the
`Scene` host API exactly as an application of the reference would write them.

  cornell(scene)   "CORNELL-REF": DemoScenes::createCornellBox layout
                   (src/raytracer/RTapp_utils.cuh:251-312) expressed through the PT Scene API:
                   8 cube meshes, 96 triangles, 1 point light, sky disabled.
  showcase(scene)  stand-in for reference scene 8 (src/pathtracer/app_utils.cuh:585-679), whose
                   OBJ assets are not in the repository: 10 x addSphere(71) (100,820 triangles)
                   at the scene-8 positions with the scene-8 materials, lights and camera,
                   plus an addPlaneXZ floor.
  fluid(scene, t)  dynamic height-field "water" over a static sphere "ship" (config 5).
  many(scene, n)   Cornell box + n small cubes and spheres, every third an instance with its own transform:
                   more meshes than one TLAS leaf holds (17), i.e. a TLAS with inner nodes.
  coincident(scene) duplicated, coplanar and degenerate triangles: the cases in which first-found-wins decides.
"""
import math

import numpy as np

from . import Material

WHITE = dict(albedo=(0.73, 0.73, 0.73), roughness=0.6)
RED = dict(albedo=(0.65, 0.05, 0.05), roughness=0.6)
GREEN = dict(albedo=(0.12, 0.45, 0.15), roughness=0.6)


def cornell(scene, quads=False):
    """Canonical Cornell box; ``quads=True`` swaps the five wall slabs for 2-triangle quads."""
    white, red, green = Material(**WHITE), Material(**RED), Material(**GREEN)
    light = Material(albedo=(0.0, 0.0, 0.0), roughness=0.0, emission=(15.0, 15.0, 15.0))
    box = Material(albedo=(0.9, 0.9, 0.9), roughness=0.2)

    def slab(mat, scale, pos, rot=None):
        m = scene.addCube(mat)
        scene.scale(m, scale)
        scene.moveTo(m, pos)
        if rot is not None:
            scene.rotateSelfEulerXYZ(m, rot)
        return m

    if not quads:
        slab(white, (10.0, 10.0, 0.1), (0, 0, -10))   # back
        slab(red, (0.1, 10.0, 10.0), (-5, 0, -5))     # left
        slab(green, (0.1, 10.0, 10.0), (5, 0, -5))    # right
        slab(white, (10.0, 0.1, 10.0), (0, -5, -5))   # floor
        slab(white, (10.0, 0.1, 10.0), (0, 5, -5))    # ceiling
    else:
        def quad(mat, a, b, c, d):
            scene.addTriangles([a + b + c, a + c + d], mat)
        quad(white, (-5, -5, -10), (5, -5, -10), (5, 5, -10), (-5, 5, -10))
        quad(red, (-5, -5, 0), (-5, -5, -10), (-5, 5, -10), (-5, 5, 0))
        quad(green, (5, -5, -10), (5, -5, 0), (5, 5, 0), (5, 5, -10))
        quad(white, (-5, -5, 0), (5, -5, 0), (5, -5, -10), (-5, -5, -10))
        quad(white, (-5, 5, -10), (5, 5, -10), (5, 5, 0), (-5, 5, 0))
    slab(light, (2.0, 0.1, 2.0), (0, 4.9, -5))
    slab(box, (1.5, 3.0, 1.5), (-1.5, -3.5, -6), (0, 0.3, 0))
    slab(box, (1.5, 1.5, 1.5), (1.5, -4.25, -4), (0, -0.4, 0))
    scene.addPointLight((0, 4.5, -5), (1.0, 0.9, 0.8), 3.0, 20.0)
    scene.setCamera((0, 0, 5), (0, 0, -5), (0, 1, 0), 40.0)
    scene.disableSky()
    return scene


def _f0(ior):
    f = np.float32((np.float32(ior) - np.float32(1.0)) / (np.float32(ior) + np.float32(1.0)))
    return float(f * f)  # iorToF0 (material_lib.cuh:142-145) in fp32


class Materials:
    """The application's material library (src/pathtracer/app_utils.cuh:60-191), parameter for parameter, through the
    same constructor + field assignments.  tests/test_refapp_scenes.py holds the scenes built from it to the scenes
    the reference's own buildSceneById builds over the C++ mirror."""
    Silver = staticmethod(lambda: Material((0.97, 0.96, 0.91), 0.05, 1.0))
    BrushedAluminum = staticmethod(lambda: Material((0.91, 0.92, 0.92), 0.3, 1.0))
    Gold = staticmethod(lambda: Material((1.00, 0.78, 0.34), 0.1, 1.0))
    Copper = staticmethod(lambda: Material((0.95, 0.64, 0.54), 0.2, 1.0))
    Titanium = staticmethod(lambda: Material((0.542, 0.497, 0.449), 0.15, 1.0))
    Glass = staticmethod(lambda: Material((1.0, 1.0, 1.0), 0.0, transmission=1.0, ior=1.5, specular=(_f0(1.5),) * 3))
    FrostedGlass = staticmethod(lambda: Materials.Glass().set("roughness", 0.2))
    Water = staticmethod(lambda: Materials.Glass().set("ior", 1.33))
    Diamond = staticmethod(lambda: Materials.Glass().set("ior", 2.417).set("specular", (_f0(2.417),) * 3))
    SoapBubble = staticmethod(lambda: Material((1.0, 1.0, 1.0), 0.0, transmission=0.95, ior=1.01, iridescence=1.0,
                                               iridescenceThickness=400.0))
    OilSlick = staticmethod(lambda: Material((0.1, 0.1, 0.1), 0.4, 0.8, iridescence=1.0, iridescenceThickness=600.0))
    VelvetRed = staticmethod(lambda: Material((0.4, 0.01, 0.05), 0.8, sheen=1.0, sheenTint=(1.0, 0.5, 0.5)))
    SatinBlue = staticmethod(lambda: Material((0.1, 0.1, 0.6), 0.3, sheen=0.8, anisotropy=0.6))
    # metallic is assigned AFTER construction, so specular stays 0.04 (app_utils.cuh:133-139)
    CarPaintMidnight = staticmethod(lambda: Material((0.02, 0.02, 0.15), 0.5, clearcoat=1.0,
                                                     clearcoatRoughness=0.01).set("metallic", 0.4))
    LacqueredWood = staticmethod(lambda: Material((0.2, 0.1, 0.02), 0.6, clearcoat=1.0, clearcoatRoughness=0.05))
    PlasticRed = staticmethod(lambda: Material((0.8, 0.1, 0.1), 0.3))
    RubberBlack = staticmethod(lambda: Material((0.05, 0.05, 0.05), 0.8))
    Wax = staticmethod(lambda: Material((0.9, 0.8, 0.5), 0.3, transmission=0.2))
    Jade = staticmethod(lambda: Material((0.1, 0.6, 0.3), 0.4, subsurfaceRadius=1.0, subsurfaceColor=(0.1, 0.8, 0.4)))
    MarbleCarrara = staticmethod(lambda: Material((0.95, 0.95, 0.95), 0.1, 0.5))

    @staticmethod
    def GlowingNeon(color):
        e = np.asarray(color, dtype=np.float32) * np.float32(10.0)
        return Material((0.0, 0.0, 0.0)).set("emission", e)


def lit_test(scene):
    """Scenes::createLitTestScene (app_utils.cuh:196-206), scene id 0 and the fallback of buildSceneById."""
    scene.addPlaneXZ(-1.0, 50.0, Material((0.8, 0.8, 0.8), 0.5))
    cube = scene.addCube(Materials.Silver())
    scene.moveTo(cube, (0, 0.5, 3))
    scene.addSpotLight((-3, 5, 2), (1, -1, 1), (1.0, 1.0, 1.0), 5.0)
    scene.addPointLight((2, 3, 1), (0.8, 0.8, 1.0), 2.0)
    scene.setCamera((0, 1.5, -2), (0, 0.5, 3), (0, 1, 0), 60.0)
    return scene


def material_matrix(scene):
    """buildSceneById case 10, "Material Matrix (Cubes)" (app_utils.cuh:729-795): a floor and a 4 x 4 grid of cubes,
    one per material class of the path (metals, clearcoat, plastics, glass, thin film, sheen, subsurface, emitter)."""
    scene.addPlaneXZ(-1.0, 50.0, Material((0.2, 0.2, 0.2), 0.8))
    rows = cols = 4
    spacing = np.float32(2.0)
    start_x = -(np.float32(cols - 1) * spacing) / np.float32(2.0)
    start_z = -(np.float32(rows - 1) * spacing) / np.float32(2.0) - np.float32(5.0)
    M = Materials
    palette = [M.Silver(), M.Gold(), M.Copper(), M.Titanium(), M.CarPaintMidnight(), M.PlasticRed(), M.RubberBlack(),
               M.LacqueredWood(), M.Glass(), M.FrostedGlass(), M.SoapBubble(), M.OilSlick(), M.VelvetRed(), M.SatinBlue(),
               M.Jade(), M.GlowingNeon((0.2, 1.0, 0.2))]
    for r in range(rows):
        for c in range(cols):
            cube = scene.addCube(palette[r * cols + c])
            x = float(start_x + np.float32(c) * spacing)
            z = float(start_z + np.float32(r) * spacing)
            scene.scale(cube, 0.7)
            scene.moveTo(cube, (x, 0.0, z))
            scene.moveTo(cube, (x, float(np.float32(-1.0) + np.float32(0.7)), z))
            scene.rotateSelfEulerXYZ(cube, (0, 0.7, 0))
    scene.addSpotLight((0, 8, -5), (0, -1, 0), (1.0, 1.0, 1.0), 10.0, 0.1, 0.5, 2.0, 0.1)
    scene.addPointLight((-5, 2, -2), (1.0, 0.8, 0.8), 2.0, 10.0, 0.2)
    scene.addPointLight((5, 2, -2), (0.8, 0.8, 1.0), 2.0, 10.0, 0.2)
    scene.setCamera((0, 6, 4), (0, 0, -5), (0, 1, 0), 50.0)
    scene.setSkyGradient((0.1, 0.1, 0.1), (0.02, 0.02, 0.02))
    return scene


def _showcase_materials():
    m = []
    m.append(Material((0.95, 0.64, 0.54), 0.2, 1.0))                                   # Copper
    m.append(Material((0.95, 0.95, 0.95), 0.1, 0.5))                                   # MarbleCarrara
    m.append(Material((0.1, 0.6, 0.3), 0.4, subsurfaceRadius=1.0, subsurfaceColor=(0.1, 0.8, 0.4)))  # Jade
    m.append(Material((1.0, 1.0, 1.0), 0.0, transmission=0.95, ior=1.01, iridescence=1.0,
                      iridescenceThickness=400.0))                                     # SoapBubble
    m.append(Material((0.542, 0.497, 0.449), 0.15, 1.0))                               # Titanium
    # CarPaintMidnight: metallic is assigned AFTER construction, so specular stays 0.04 (app_utils.cuh:133-139)
    m.append(Material((0.02, 0.02, 0.15), 0.5, clearcoat=1.0, clearcoatRoughness=0.01).set("metallic", 0.4))
    m.append(Material((1.00, 0.78, 0.34), 0.1, 1.0))                                   # Gold
    m.append(Material((0.9, 0.8, 0.5), 0.3, transmission=0.2))                         # Wax
    m.append(Material((0.4, 0.01, 0.05), 0.8, sheen=1.0, sheenTint=(1.0, 0.5, 0.5)))   # VelvetRed
    f0 = np.float32((np.float32(1.5) - np.float32(1.0)) / (np.float32(1.5) + np.float32(1.0)))
    f0 = float(f0 * f0)                                                                # iorToF0(1.5)
    m.append(Material((1.0, 1.0, 1.0), 0.0, transmission=1.0, ior=1.5, specular=(f0, f0, f0)))  # Glass
    return m


def showcase(scene, segments=71):
    """~100k-triangle BVH scene (config 3/4). `segments` scales the triangle count: 2*segments^2 per sphere."""
    floor_y = 2.0 - 10.0 / 2.0
    xs = (-8, -4, 0, 4, 8)
    mats = _showcase_materials()
    heights = [3.0, 2.0, 3.0, 3.0, 3.0, 3.0, 2.0, 3.0, 3.0, 3.0]
    rot = (0, 0.3, 0)
    for i in range(10):
        z = -12 if i < 5 else -8
        mesh = scene.addSphere(segments, mats[i])
        scene.scale(mesh, 3.0)
        scene.moveTo(mesh, (xs[i % 5], floor_y + heights[i], z))
        scene.rotateSelfEulerXYZ(mesh, rot)
    scene.addPlaneXZ(floor_y, 50.0, Material((0.4, 0.4, 0.4), 0.9, specular=(0.0, 0.0, 0.0)))
    scene.setSkyGradient((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    one = (1.0, 1.0, 1.0)
    scene.addSpotLight((0, 6.5, -10), (0, -1, 0), one, 15.0, 0.1, 0.8, 2.0, 0.1)
    scene.addSpotLight((-6, 6.5, -10), (0, -1, 0), one, 12.0, 0.1, 0.8, 2.0, 0.1)
    scene.addSpotLight((6, 6.5, -10), (0, -1, 0), one, 12.0, 0.1, 0.8, 2.0, 0.1)
    scene.addPointLight((0, 2, 4), (0.8, 0.8, 0.8), 5.0, 20.0, 0.1)
    scene.addPointLight((-8, 1, 4), (0.5, 0.5, 0.5), 3.0, 20.0, 0.1)
    scene.addPointLight((8, 1, 4), (0.5, 0.5, 0.5), 3.0, 20.0, 0.1)
    cam_pos, cam_at = (0.0, 2.0, 5.0), (0.0, 0.0, -10.0)
    focus = math.sqrt(sum((a - b) ** 2 for a, b in zip(cam_at, cam_pos)))
    scene.setCamera(cam_pos, cam_at, (0, 1, 0), 50.0, 0.0, focus)
    return scene


def million(scene, segments=250):
    """The workload of the reference's only published numbers ("a scene with about 1 million triangles, and 8 separate
    models", Test game screenshots/readme.txt:16-19; the models themselves are not in the repository): 8 x
    addSphere(250) = 8 x 125,000 triangles in the showcase arrangement (materials, lights, camera, floor of scene 8,
    app_utils.cuh:585-679) -- nine meshes, a single-leaf TLAS."""
    floor_y = 2.0 - 10.0 / 2.0
    mats = _showcase_materials()
    xs = (-6, -2, 2, 6)
    for i in range(8):
        mesh = scene.addSphere(segments, mats[(i * 5) % 10 if i % 2 else i])
        scene.scale(mesh, 3.0)
        scene.moveTo(mesh, (xs[i % 4], floor_y + 3.0, -12 if i < 4 else -8))
        scene.rotateSelfEulerXYZ(mesh, (0, 0.3, 0))
    scene.addPlaneXZ(floor_y, 50.0, Material((0.4, 0.4, 0.4), 0.9, specular=(0.0, 0.0, 0.0)))
    scene.setSkyGradient((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    one = (1.0, 1.0, 1.0)
    scene.addSpotLight((0, 6.5, -10), (0, -1, 0), one, 15.0, 0.1, 0.8, 2.0, 0.1)
    scene.addSpotLight((-6, 6.5, -10), (0, -1, 0), one, 12.0, 0.1, 0.8, 2.0, 0.1)
    scene.addSpotLight((6, 6.5, -10), (0, -1, 0), one, 12.0, 0.1, 0.8, 2.0, 0.1)
    scene.addPointLight((0, 2, 4), (0.8, 0.8, 0.8), 5.0, 20.0, 0.1)
    scene.addPointLight((-8, 1, 4), (0.5, 0.5, 0.5), 3.0, 20.0, 0.1)
    scene.addPointLight((8, 1, 4), (0.5, 0.5, 0.5), 3.0, 20.0, 0.1)
    cam_pos, cam_at = (0.0, 2.0, 5.0), (0.0, 0.0, -10.0)
    focus = math.sqrt(sum((a - b) ** 2 for a, b in zip(cam_at, cam_pos)))
    scene.setCamera(cam_pos, cam_at, (0, 1, 0), 50.0, 0.0, focus)
    return scene


_WAVES = ((0.35, (0.45, 0.20), 1.3), (0.20, (-0.30, 0.55), 2.1), (0.12, (0.80, -0.65), 3.4))


def water_vertices(cells, t, half=20.0):
    """Unshared-vertex height field (what addTriangles produces): cells*cells*2 triangles * 3 vertices."""
    g = np.linspace(-half, half, cells + 1, dtype=np.float32)
    x, z = np.meshgrid(g, g, indexing="xy")
    y = np.zeros_like(x)
    for a, (kx, kz), w in _WAVES:
        y += np.float32(a) * np.sin(np.float32(kx) * x + np.float32(kz) * z - np.float32(w * t)).astype(np.float32)
    p = np.stack([x, y, z], axis=-1)
    a, b = p[:-1, :-1], p[:-1, 1:]
    c, d = p[1:, 1:], p[1:, :-1]
    tris = np.stack([np.stack([a, c, b], axis=2), np.stack([a, d, c], axis=2)], axis=2)  # CCW from +Y
    return np.ascontiguousarray(tris.reshape(-1, 3), dtype=np.float32)


def fluid(scene, cells=256, t=0.0, ship_segments=100):
    """Config 5: water surface (2*cells^2 triangles, re-posed per frame) over a static 'ship'."""
    water = Material((1.0, 1.0, 1.0), 0.0, transmission=1.0, ior=1.33,
                     specular=(0.04, 0.04, 0.04))
    v = water_vertices(cells, t)
    w = scene.addTriangles(v.reshape(-1, 9), water)
    ship = scene.addSphere(ship_segments, Material((0.6, 0.35, 0.2), 0.5))
    scene.scale(ship, (6.0, 2.0, 3.0))
    scene.moveTo(ship, (0.0, 0.4, -2.0))
    scene.setSkyGradient((0.35, 0.55, 0.95), (0.9, 0.95, 1.0))
    scene.addDirectionalLight((-0.4, -1.0, -0.3), (1.0, 0.96, 0.9), 3.0)
    scene.setCamera((0.0, 6.0, 18.0), (0.0, 0.0, 0.0), (0, 1, 0), 45.0)
    return w, ship


def many(scene, n=40, instanced=True, sphere_segments=5):
    """Cornell box + n small cubes and spheres (every third an instance with its own transform, every seventh
    transmissive): more meshes than a TLAS leaf holds, so the TLAS is a real tree."""
    cornell(scene)
    rs = np.random.RandomState(3)
    for k in range(n):
        mat = Material(tuple(rs.uniform(0.2, 0.9, 3)), float(rs.uniform(0.05, 0.8)), float(k % 4 == 0),
                       transmission=1.0 if k % 7 == 3 else 0.0, ior=1.4)
        m = scene.addSphere(sphere_segments, mat) if k % 2 else scene.addCube(mat)
        pos = (float(rs.uniform(-4, 4)), float(rs.uniform(-4.5, 3.5)), float(rs.uniform(-9, -2)))
        if instanced and k % 3 == 0:
            scene.setPosition(m, pos)
            scene.setRotation(m, tuple(rs.uniform(-1, 1, 3)))
            scene.setInstanceScale(m, tuple(rs.uniform(0.2, 0.5, 3)))
        else:
            scene.scale(m, tuple(rs.uniform(0.2, 0.5, 3)))
            scene.moveTo(m, pos)


def _wall(n, z, half=3.0, dup=True):
    """n x n axis-aligned quads at depth z; with `dup` every triangle twice."""
    tris = []
    xs = np.linspace(-half, half, n + 1, dtype=np.float32)
    for j in range(n):
        for i in range(n):
            a, b = (xs[i], xs[j], z), (xs[i + 1], xs[j], z)
            c, d = (xs[i + 1], xs[j + 1], z), (xs[i], xs[j + 1], z)
            for t in ((a, b, c), (a, c, d)):
                tris.extend([t, t] if dup else [t])
    return tris


def coincident(scene, n=5, leaf=8):
    """Geometry in which the ORDER of the reference's traversal decides what is seen (intersection.cuh:247, :561: `t < best`,
    strict -- of triangles at exactly the same distance the one found first wins) and its arithmetic leaves the ordinary
    range (zero-area and collinear triangles: the |det| < 1e-6 reject, intersection.cuh:231).  Eight meshes; `leaf` is the
    builder's leaf target for BLAS and TLAS alike (scene.cuh:556): 8 keeps the meshes in one TLAS leaf, 2 makes a real TLAS."""
    red, blue = Material((0.8, 0.1, 0.1), 0.5), Material((0.1, 0.2, 0.9), 0.05, 1.0)
    white, green = Material((0.73, 0.73, 0.73), 0.6), Material((0.1, 0.7, 0.2), 0.3)
    glass = Material((0.95, 0.95, 0.95), 0.02, 0.0, transmission=1.0, ior=1.5)
    p, q, r = (0.5, 0.5, -5.0), (1.5, 0.5, -5.0), (2.5, 0.5, -5.0)
    junk = [(p, p, q), (p, q, r), (q, q, q),                                    # two equal vertices, collinear, a point
            ((0.0, 0.0, -5.5), (1e-4, 0.0, -5.5), (0.0, 1e-4, -5.5)),           # |det| below the reference's 1e-6
            ((-2.9, -2.9, -5.9), (2.9, -2.9, -5.9), (2.9, -2.9000001, -5.9))]   # a needle
    scene.addTriangles(_wall(n, -6.0) + junk, red)  # every wall triangle twice: exact ties inside one BLAS
    # the same triangles again as a second mesh with another material: exact ties ACROSS meshes -- the first mesh of the TLAS
    # leaf wins in the reference, whatever order the pairs are walked in here
    scene.addTriangles(_wall(n, -6.0, dup=False), blue)
    # ... and once more with the other winding: the same plane, distances that differ in the last bits or not at all
    scene.addTriangles([(c, b, a) for a, b, c in _wall(n, -6.0, dup=False)], green)
    scene.addPlaneXZ(-3.0, 6.0, white)
    scene.addPlaneXZ(-3.0, 6.0, green)  # coplanar floors: every floor hit is a tie, every shadow ray starts on both
    box = scene.addCube(glass)          # its back face lies IN the wall's plane, its bottom face in the floors'
    scene.scale(box, (2.0, 2.0, 2.0))
    scene.moveTo(box, (-1.5, -2.0, -5.0))
    inst = scene.addCube(green)         # an instance whose transform maps a face onto the wall's plane as well
    scene.setPosition(inst, (1.5, 0.0, -5.5))
    scene.setRotation(inst, (0.0, math.pi / 2, 0.0))
    scene.setInstanceScale(inst, (1.0, 2.0, 1.0))
    scene.addPointLight((0.0, 2.0, -3.0), (1.0, 0.9, 0.8), 40.0, 30.0, 0.0)
    scene.addPointLight((2.0, -1.0, -6.0), (0.3, 0.5, 1.0), 30.0, 30.0, 0.25)  # IN the wall's plane: its shadow rays graze the triangles
    scene.setSkyGradient((0.5, 0.7, 1.0), (1.0, 1.0, 1.0))
    scene.setCamera((0.0, 0.0, 4.0), (0.0, 0.0, -6.0), (0.0, 1.0, 0.0), 45.0)
    scene.setBVHLeafTarget(leaf, 0)


# ---- the reference's dynamic-geometry caller (src/common/PTRTtransfer.cuh) ------------------------------------------------
# tools/refapp/transfer_probe.cpp compiles buildPTScene / updatePTScene / updatePTCamera of the reference in place over the
# C++ mirror and runs them on a UnifiedScene; the functions below are the same scene and the same per-step changes through
# this module's Scene API, call by call (tests/test_transfer_scenes.py holds the two to the same bytes).
TRANSFER_CELLS, TRANSFER_SIZE = 12, (320, 180)


def unified_material(albedo=(0.8, 0.8, 0.8), roughness=0.5, metallic=0.0, **fields):
    """toPTMaterial(UnifiedMaterial(albedo, roughness, metallic)) (PTRTtransfer.cuh:242-275, 2038-2060): every field is
    assigned, so Material(alb, rough, met)'s transmission-roughness floor does not apply."""
    one, t = np.float32(1.0), np.float32(metallic)
    alb = np.asarray(albedo, dtype=np.float32)
    m = Material()  # the defaults are UnifiedMaterial's
    m.set("albedo", alb).set("roughness", roughness).set("metallic", metallic)
    m.set("specular", (one - t) * np.float32(0.04) + t * alb)  # ::lerp(vec3(0.04f), albedo, metallic)
    m.set("transmissionRoughness", 0.0)
    for k, v in fields.items():
        m.set(k, v)
    return m


def transfer_sheet(k, cells=TRANSFER_CELLS):
    """transfer_probe.cpp's sheet(k): (cells^2 * 2, 9) float32, the same fp32 products and sums in the same order."""
    f = np.float32
    t = f(k)
    i = np.arange(cells + 1, dtype=np.float32)
    x, z = np.meshgrid(f(-3.0) + f(0.5) * i, f(-3.0) + f(0.5) * i, indexing="xy")  # x along i, z along j
    a, b = x * z, x - z
    y = (f(0.05) * a + (f(0.04) * t) * b) + (f(0.02) * t) * (x * x)
    p = np.stack([x, y.astype(np.float32), z], axis=-1).astype(np.float32)
    A, B = p[:-1, :-1], p[:-1, 1:]
    C, D = p[1:, 1:], p[1:, :-1]
    tris = np.stack([np.stack([A, C, B], axis=2), np.stack([A, D, C], axis=2)], axis=2)
    return np.ascontiguousarray(tris.reshape(-1, 9), dtype=np.float32)


def transfer_demo(scene):
    """make_unified() + buildPTScene (PTRTtransfer.cuh:2120-2200): camera, leaf target, meshes in order (a `Triangles`
    sheet, a DYNAMIC cube = an instance, a floor, a static sphere whose transform is baked: scale, rotate, moveTo), lights,
    sky.  Returns the mesh indices (water, cube)."""
    scene.setCamera((0.0, 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 42.0, 0.0, 1.0)
    scene.setBVHLeafTarget(4, 2)
    water = scene.addTriangles(transfer_sheet(0), unified_material((0.2, 0.45, 0.8), 0.15, 0.0))
    cube = scene.addCube(unified_material((1.0, 0.766, 0.336), 0.1, 1.0, specular=(1.0, 0.782, 0.344)))  # UnifiedMaterial::Gold
    scene.setPosition(cube, (-1.5, 0.9, 0.5))
    scene.setRotation(cube, (0.2, 0.4, 0.0))
    scene.setInstanceScale(cube, (0.8, 0.8, 0.8))
    scene.addPlaneXZ(-1.0, 6.0, unified_material((0.7, 0.7, 0.65), 0.9, 0.0))
    ball = scene.addSphere(12, unified_material((0.8, 0.3, 0.25), 0.4, 0.0))
    scene.scale(ball, (1.2, 0.9, 1.2))
    scene.rotateSelfEulerXYZ(ball, (0.0, 0.3, 0.1))
    scene.moveTo(ball, (1.6, 0.8, -0.5))
    scene.addPointLight((0.0, 5.0, 2.0), (1.0, 0.95, 0.9), 40.0, 100.0, 0.3)
    scene.addSpotLight((-3.0, 4.0, 3.0), (0.6, -0.8, -0.6), (0.6, 0.7, 1.0), 60.0, 0.3, 0.5, 50.0, 0.0)
    scene.setSkyGradient((0.3, 0.5, 0.9), (0.9, 0.9, 1.0))
    return water, cube


def transfer_step(scene, water, cube, k):
    """step(u, k) + updatePTScene + updatePTCamera (PTRTtransfer.cuh:2204-2393): the `Triangles` mesh's vertices and faces
    rewritten with both dirty flags set, the dynamic cube's transform, commitObjectChanges(), then the camera."""
    f = np.float32
    scene.setTriangleSoup(water, transfer_sheet(k))
    scene.setPosition(cube, (f(-1.5) + f(0.4) * f(k), 0.9, 0.5))
    scene.setRotation(cube, (0.2, f(0.4) + f(0.3) * f(k), 0.0))
    scene.setInstanceScale(cube, (0.8, 0.8, 0.8))
    scene.commitObjectChanges()
    scene.setCamera((f(0.5) * f(k), 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 42.0, 0.0, 1.0)
