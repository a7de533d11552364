"""Scene authoring on the host, and the scenes the benchmark configurations name.

SceneBuilder mirrors the reference's scene-definition surface (scene.h:20-58): materials are
created first and referred to by the objects that use them, objects are kept in list order, and
the camera takes the reference constructor's arguments (camera.h:38-46).  desc() flattens it to
the arrays pt_scene_upload takes.
"""
import numpy as np

from . import api

f32 = np.float32


class SceneBuilder:
    def __init__(self, background=(0.0, 0.0, 0.0)):
        self.prim_type, self.prim_param, self.prim_material = [], [], []
        self.mat_type, self.mat_param = [], []
        self.background = tuple(float(f32(x)) for x in background)
        self.camera_args = None

    # ---- materials (material.h:87,150,220,292); each returns the material's index
    def _mat(self, tag, p):
        self.mat_type.append(tag)
        self.mat_param.append([f32(x) for x in p])
        return len(self.mat_type) - 1

    def lambertian(self, albedo):
        return self._mat(api.LAMBERTIAN, [albedo[0], albedo[1], albedo[2], 0])

    def metal(self, albedo, fuzz):
        fuzz = f32(fuzz)
        return self._mat(api.METAL, [albedo[0], albedo[1], albedo[2], f32(1.0) if fuzz > f32(1.0) else fuzz])  # material.h:150

    def dielectric(self, ir):
        return self._mat(api.DIELECTRIC, [ir, 0, 0, 0])

    def diffuse_light(self, emit):
        return self._mat(api.DIFFUSE_LIGHT, [emit[0], emit[1], emit[2], 0])

    # ---- objects, in list order (hittable_list.h:76)
    def _prim(self, tag, p, mat):
        if not (0 <= mat < len(self.mat_type)):
            raise ValueError("material %r does not exist" % (mat,))
        self.prim_type.append(tag)
        self.prim_param.append([f32(x) for x in p])
        self.prim_material.append(mat)
        return len(self.prim_type) - 1

    def sphere(self, center, radius, mat):  # sphere.h:26
        return self._prim(api.SPHERE, [center[0], center[1], center[2], radius, 0, 0], mat)

    def xy_rect(self, x0, x1, y0, y1, k, mat):  # aarect.h:35
        return self._prim(api.XY_RECT, [x0, x1, y0, y1, k, 0], mat)

    def xz_rect(self, x0, x1, z0, z1, k, mat):  # aarect.h:130
        return self._prim(api.XZ_RECT, [x0, x1, z0, z1, k, 0], mat)

    def yz_rect(self, y0, y1, z0, z1, k, mat):  # aarect.h:215
        return self._prim(api.YZ_RECT, [y0, y1, z0, z1, k, 0], mat)

    def cone(self, center, height, radius, mat):  # cone.h:28-31
        height, radius = f32(height), f32(radius)
        tangens = (radius / height) * (radius / height)
        return self._prim(api.CONE, [center[0], center[1], center[2], height, radius, tangens], mat)

    def cylinder2(self, center, height, radius, mat):  # cylinder.h:28 -- (center, HEIGHT, RADIUS)
        return self._prim(api.CYLINDER, [center[0], center[1], center[2], height, radius, 0], mat)

    def camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):  # camera.h:38-46
        self.camera_args = (tuple(lookfrom), tuple(lookat), tuple(vup), float(f32(vfov)), float(f32(aspect)),
                            float(f32(aperture)), float(f32(focus_dist)))

    def arrays(self):
        return dict(
            prim_type=np.array(self.prim_type, dtype=np.uint8),
            prim_param=np.array(self.prim_param, dtype=np.float32).reshape(-1, 6),
            prim_material=np.array(self.prim_material, dtype=np.uint32),
            mat_type=np.array(self.mat_type, dtype=np.uint8),
            mat_param=np.array(self.mat_param, dtype=np.float32).reshape(-1, 4),
        )

    def desc(self):
        if self.camera_args is None:
            raise ValueError("scene has no camera")
        cam = api.camera_look_at(*self.camera_args)
        return api.SceneDesc(camera=cam, background=self.background, **self.arrays())


class ArrayScene(SceneBuilder):
    """A scene given directly as arrays (the large generated scenes)."""

    def __init__(self, arrays, background, camera_args):
        super().__init__(background)
        self._arrays = arrays
        self.camera_args = camera_args

    def arrays(self):
        return self._arrays


def default_scene():
    """scene::scene() (scene.h:20-58): ground sphere, one sphere, one emissive xy_rect."""
    s = SceneBuilder(background=(0, 0, 0))
    ground = s.lambertian((0.5, 0.5, 0.5))
    s.sphere((0, -1000, 0), 1000, ground)
    m2 = s.lambertian((0.4, 0.2, 0.1))
    s.sphere((0, 2, 0), 2.0, m2)
    light = s.diffuse_light((4, 4, 4))
    s.xy_rect(3, 5, 1, 3, -2, light)
    s.camera((26, 3, 6), (0, 2, 0), (0, 1, 0), 20.0, f32(16.0) / f32(9.0), 0.1, 10.0)
    return s


def cornell_scene():
    """scene::scene(int) (scene.h:64-119): five rect walls, a cylinder, a rect light."""
    s = SceneBuilder(background=(0, 0, 0))
    red = s.lambertian((0.65, 0.05, 0.05))
    s.yz_rect(0, 555, 0, 555, 0, red)
    white = s.lambertian((0.73, 0.73, 0.73))
    s.xz_rect(0, 555, 0, 555, 0, white)
    white2 = s.lambertian((0.73, 0.73, 0.73))
    s.xz_rect(0, 555, 0, 555, 555, white2)
    white3 = s.lambertian((0.73, 0.73, 0.73))
    s.xy_rect(0, 555, 0, 555, 555, white3)
    green2 = s.lambertian((0.12, 0.45, 0.15))
    s.cylinder2((348, 0, 200), 200, 64, green2)  # scene.h:85: height 200, radius 64
    green = s.lambertian((0.12, 0.45, 0.15))
    s.yz_rect(0, 555, 0, 555, 555, green)
    light = s.diffuse_light((15, 15, 15))
    s.xz_rect(213, 343, 227, 332, 554, light)
    s.camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.1, 10.0)
    return s


class _Lcg:
    """64-bit LCG (Knuth MMIX constants); uniform floats in [0,1) from the top 24 bits."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return f32((self.s >> 40) * (1.0 / 16777216.0))


def random_spheres_scene(seed=12345):
    """The 'Ray Tracing in One Weekend' final scene: ground, a 22x22 grid of small spheres
    (80% lambertian, 15% metal, 5% glass) and three large ones; constant sky background."""
    g = _Lcg(seed)
    s = SceneBuilder(background=(0.7, 0.8, 1.0))
    s.sphere((0, -1000, 0), 1000, s.lambertian((0.5, 0.5, 0.5)))
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = g.next()
            c = (f32(a) + f32(0.9) * g.next(), f32(0.2), f32(b) + f32(0.9) * g.next())
            dx, dy, dz = c[0] - f32(4), c[1] - f32(0.2), c[2]
            if np.sqrt(dx * dx + dy * dy + dz * dz) > f32(0.9):
                if choose < f32(0.8):
                    alb = tuple(g.next() * g.next() for _ in range(3))
                    m = s.lambertian(alb)
                elif choose < f32(0.95):
                    alb = tuple(f32(0.5) + f32(0.5) * g.next() for _ in range(3))
                    m = s.metal(alb, f32(0.5) * g.next())
                else:
                    m = s.dielectric(1.5)
                s.sphere(c, 0.2, m)
    s.sphere((0, 1, 0), 1.0, s.dielectric(1.5))
    s.sphere((-4, 1, 0), 1.0, s.lambertian((0.4, 0.2, 0.1)))
    s.sphere((4, 1, 0), 1.0, s.metal((0.7, 0.6, 0.5), 0.0))
    s.camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, f32(16.0) / f32(9.0), 0.1, 10.0)
    return s


def _palette(rng, n):
    """n materials: 60% lambertian, 20% metal, 15% dielectric, 5% light."""
    u = rng.random(n, dtype=np.float32)
    mt = np.where(u < 0.6, api.LAMBERTIAN, np.where(u < 0.8, api.METAL, np.where(u < 0.95, api.DIELECTRIC, api.DIFFUSE_LIGHT))).astype(np.uint8)
    mp = np.zeros((n, 4), dtype=np.float32)
    col = rng.random((n, 3), dtype=np.float32)
    mp[:, :3] = col
    mp[mt == api.METAL, 3] = (0.5 * rng.random(n, dtype=np.float32))[mt == api.METAL]
    mp[mt == api.DIELECTRIC] = np.array([1.5, 0, 0, 0], dtype=np.float32)
    mp[mt == api.DIFFUSE_LIGHT, :3] = 2.0 + 6.0 * col[mt == api.DIFFUSE_LIGHT]
    return mt, mp


def mixed_scene(n=100_000, seed=7, extent=200.0, n_materials=1024, aspect=16.0 / 9.0):
    """n objects, a quarter each of spheres, cones, cylinders and axis-aligned rects, uniform in
    an extent^3 box, sizes in [0.2, 1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mt, mp = _palette(rng, n_materials)
    kind = rng.integers(0, 4, n)
    rect_axis = rng.integers(0, 3, n)
    ptype = np.where(kind == 0, api.SPHERE, np.where(kind == 1, api.CONE, np.where(kind == 2, api.CYLINDER, api.XY_RECT + rect_axis))).astype(np.uint8)
    c = (rng.random((n, 3), dtype=np.float32) * f32(extent)).astype(np.float32)
    sz = (f32(0.2) + f32(0.8) * rng.random((n, 2), dtype=np.float32)).astype(np.float32)
    pp = np.zeros((n, 6), dtype=np.float32)
    sph = ptype == api.SPHERE
    pp[sph, :3] = c[sph]
    pp[sph, 3] = sz[sph, 0]
    cc = (ptype == api.CONE) | (ptype == api.CYLINDER)
    pp[cc, :3] = c[cc]
    pp[cc, 3] = sz[cc, 0] + f32(0.5)  # height
    pp[cc, 4] = sz[cc, 1] * f32(0.5)  # radius
    cone = ptype == api.CONE
    pp[cone, 5] = (pp[cone, 4] / pp[cone, 3]) * (pp[cone, 4] / pp[cone, 3])  # cone.h:30
    for tag, (a0, a1, ak) in {api.XY_RECT: (0, 1, 2), api.XZ_RECT: (0, 2, 1), api.YZ_RECT: (1, 2, 0)}.items():
        m = ptype == tag
        pp[m, 0] = c[m, a0]
        pp[m, 1] = c[m, a0] + sz[m, 0]
        pp[m, 2] = c[m, a1]
        pp[m, 3] = c[m, a1] + sz[m, 1]
        pp[m, 4] = c[m, ak]
    pm = rng.integers(0, n_materials, n).astype(np.uint32)
    e = float(extent)
    cam = ((-0.35 * e, 0.8 * e, -0.6 * e), (0.5 * e, 0.4 * e, 0.5 * e), (0, 1, 0), 40.0, float(f32(aspect)), 0.1, 1.2 * e)
    arrays = dict(prim_type=ptype, prim_param=pp, prim_material=pm, mat_type=mt, mat_param=mp)
    return ArrayScene(arrays, (0.7, 0.8, 1.0), cam)


def sphere_slab_scene(n=1_000_000, seed=11, aspect=16.0 / 9.0):
    """n spheres of radius 0.05..0.2 uniform in a 400 x 50 x 400 slab, over a ground sphere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mt, mp = _palette(rng, 1024)
    mt[0], mp[0] = api.LAMBERTIAN, np.array([0.5, 0.5, 0.5, 0], dtype=np.float32)
    pp = np.zeros((n + 1, 6), dtype=np.float32)
    pp[0, :4] = [0, -1000, 0, 1000]
    pp[1:, 0] = rng.random(n, dtype=np.float32) * f32(400) - f32(200)
    pp[1:, 1] = rng.random(n, dtype=np.float32) * f32(50)
    pp[1:, 2] = rng.random(n, dtype=np.float32) * f32(400) - f32(200)
    pp[1:, 3] = f32(0.05) + f32(0.15) * rng.random(n, dtype=np.float32)
    pm = rng.integers(1, 1024, n + 1).astype(np.uint32)
    pm[0] = 0
    cam = ((-260, 90, -260), (0, 20, 0), (0, 1, 0), 35.0, float(f32(aspect)), 0.1, 380.0)
    arrays = dict(prim_type=np.zeros(n + 1, dtype=np.uint8), prim_param=pp, prim_material=pm, mat_type=mt, mat_param=mp)
    return ArrayScene(arrays, (0.7, 0.8, 1.0), cam)
