"""Host-callback render path (source_amd/optical/hybrid.py) without a GPU: the wave scheduler, the per-node Philox streams and the
host forms of the materials, with the CPU oracle standing in for the device's batched hit / contains queries. The frames must
equal the oracle's own path tracer bit for bit (same Philox counters, same arithmetic), which is the same statement the GPU tests
make with the device in the loop (tests/test_gpu_parity.py::test_host_callback_path_*)."""
import numpy as np
import pytest

from source_amd import scenes
from source_amd.device import DeviceScene


class OracleScene:
    """Duck type of source_amd.device.DeviceScene for the host-callback path: batched queries answered by the oracle."""

    def __init__(self, orc, flat):
        self.orc, self.flat = orc, flat

    def hit_batch(self, origin, direction, max_distance=None, geometry=False):
        return self.orc.hit_batch(self.flat, origin, direction, max_distance, geometry=geometry, threads=self.orc.max_threads())

    def contains_batch(self, points):
        return self.orc.contains_batch(self.flat, points)

    _intersection = DeviceScene._intersection


def render_both(orc, ns, world, cam, pipe, seed, monkeypatch):
    flat = world.flatten()
    fake = OracleScene(orc, flat)
    monkeypatch.setattr(world, "build_accelerator", lambda force=False: fake)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=seed, host_materials=True)
    cam.observe()
    nx, ny = cam.pixels
    ref_m, ref_v, ref_rays = np.zeros(pipe.frame.shape), np.zeros(pipe.frame.shape), 0
    for sl in cam._slice_spectrum():
        keep = []
        desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, nx, ny))
        om, ov, n_rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        ref_rays += n_rays
        ref_m[:, :, sl.offset:sl.offset + sl.bins] = om.reshape(ny, nx, sl.bins).transpose(1, 0, 2)
        ref_v[:, :, sl.offset:sl.offset + sl.bins] = ov.reshape(ny, nx, sl.bins).transpose(1, 0, 2)
    return pipe.frame, ref_m, ref_v, ref_rays


def test_lambert_room_equals_oracle(orc, ns, monkeypatch):
    """Diffuse room with a CSG solid, a smooth mesh, a glowing volume and a null shell: Lambert scattering, roulette, null
    surfaces and volume emission, every material evaluated by its host method."""
    world, prims = scenes.build_lambert(ns)
    cam, pipe = scenes.lambert_camera(ns, world, (20, 16), spp=3, bins=4, extinction=(0.2, 2, 9))
    frame, m, v, rays = render_both(orc, ns, world, cam, pipe, 5, monkeypatch)
    assert np.array_equal(frame.mean, m) and np.array_equal(frame.variance, v) and (frame.samples == 3).all()
    assert cam.stats["rays"] == rays and (m > 0).mean() > 0.3


def test_cornell_importance_sampling_and_glass_equals_oracle(orc, ns, monkeypatch):
    """Cornell box: multiple importance sampling towards the light and the glass objects (ContinuousBSDF's mixture), clear
    dielectrics (Fresnel choice, total internal reflection), an RGB pipeline next to the spectral one, two accumulating passes."""
    world, prims = scenes.build_cornell(ns)
    rgb, spectral = ns.RGBPipeline2D(), ns.SpectralRadiancePipeline2D()
    cam, _ = scenes.cornell_camera(ns, world, (14, 12), 2, 5, pipelines=[rgb, spectral])
    frame, m, v, rays = render_both(orc, ns, world, cam, spectral, 3, monkeypatch)
    assert np.array_equal(frame.mean, m) and np.array_equal(frame.variance, v)
    assert cam.stats["rays"] == rays and np.isfinite(rgb.xyz_frame.mean).all() and rgb.xyz_frame.mean[:, :, 1].max() > 0
    cam.observe()                                           # second pass: fresh counters, merged by combine_samples
    assert (spectral.frame.samples == 4).all() and not np.array_equal(spectral.frame.mean, m)


def test_user_material_renders_and_matches_a_lowered_twin(orc, ns, monkeypatch):
    """A material the library has never seen — written against the reference's plugin API (evaluate_surface tracing TWO daughters,
    evaluate_volume) — renders through observe(); a user subclass that merely re-implements Lambert's shading gives the frame of
    the built-in Lambert bit for bit."""
    from source_amd.optical.material import Material, has_device_lowering

    class MyLambert(ns.Lambert):                            # same physics, user code: not recognised as lowered
        def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
            from source_amd.optical.material import hemisphere_cosine_pdf
            pdf = hemisphere_cosine_pdf(s_out)
            if pdf == 0.0:
                return ray.new_spectrum()
            spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
            spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
            spectrum.mul_scalar(pdf)
            return spectrum

    class HalfMirrorHalfGlow(Material):                     # two daughters per hit + a volume contribution
        def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal, w2p, p2w, intersection):
            n = normal.transform_with_inverse(w2p).normalise()          # world-space normal
            d = ray.direction
            k = 2 * (d.x * n.x + d.y * n.y + d.z * n.z)
            mirror = ns.Vector3D(d.x - k * n.x, d.y - k * n.y, d.z - k * n.z)
            origin = (inside_point if exiting else outside_point).transform(p2w)
            a = ray.spawn_daughter(origin, mirror).trace(world)
            through = ray.spawn_daughter((outside_point if exiting else inside_point).transform(p2w), d).trace(world)
            a.mul_scalar(0.5)
            a.mad_scalar(0.5, through.samples)
            return a

        def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, w2p, p2w):
            spectrum.samples[:] = spectrum.samples + 0.05 * start_point.vector_to(end_point).length
            return spectrum

    assert not has_device_lowering(MyLambert()) and not has_device_lowering(HalfMirrorHalfGlow())
    frames = []
    for lambert in (ns.Lambert, MyLambert):
        world, prims = scenes.build_lambert(ns, with_volume=False, csg=False)
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = lambert(p.material.reflectivity)
        ns.Sphere(0.25, world, ns.translate(0.35, 0.1, 0.9), HalfMirrorHalfGlow())
        cam, pipe = scenes.lambert_camera(ns, world, (16, 12), spp=2, bins=3, extinction=(0.2, 2, 8))
        flat = world.flatten()
        fake = OracleScene(orc, flat)
        monkeypatch.setattr(world, "build_accelerator", lambda force=False, fake=fake: fake)
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=9)          # no host_materials flag: the user material forces the path
        cam.observe()
        frames.append((pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"]))
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1]) and frames[0][2] == frames[1][2]
    assert np.isfinite(frames[0][0]).all() and (frames[0][0] > 0).mean() > 0.15 and frames[0][2] > 16 * 12 * 2 * 2


def _render(orc, ns, monkeypatch, build, camera, seed, **engine):
    world, prims = build()
    cam, pipe = camera(world)
    fake = OracleScene(orc, world.flatten())
    monkeypatch.setattr(world, "build_accelerator", lambda force=False, fake=fake: fake)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=seed, host_materials=True, **engine)
    cam.observe()
    return pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"]


def test_array_forms_equal_the_per_node_plugin_calls(orc, ns, monkeypatch):
    """The library's materials evaluated for whole waves in numpy (the default) against the same materials called node by node
    through evaluate_surface / evaluate_volume (per_node_materials=True: deferred daughters, and the classic re-evaluation
    wherever a volume surrounds the node): frames and ray counts equal, on the Lambert room (CSG, mesh, glowing volume, null
    shell) and on the Cornell box (importance sampling, glass)."""
    scenes_ = ((lambda: scenes.build_lambert(ns), lambda w: scenes.lambert_camera(ns, w, (14, 10), spp=2, bins=3, extinction=(0.2, 2, 9))),
               (lambda: scenes.build_cornell(ns), lambda w: scenes.cornell_camera(ns, w, (12, 10), 2, 4)))
    for build, camera in scenes_:
        a = _render(orc, ns, monkeypatch, build, camera, 7)
        b = _render(orc, ns, monkeypatch, build, camera, 7, per_node_materials=True, host_workers=1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and (a[0] > 0).mean() > 0.2


def test_worker_processes_do_not_change_the_frame(orc, ns, monkeypatch):
    """Python materials evaluated by three forked worker processes (their rays traced by the parent) give the frame of one
    process bit for bit: every node draws from its own counter-based stream."""
    from source_amd.optical import hybrid
    from source_amd.optical.material import hemisphere_cosine_pdf

    class MyLambert(ns.Lambert):
        def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
            pdf = hemisphere_cosine_pdf(s_out)
            if pdf == 0.0:
                return ray.new_spectrum()
            spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
            spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
            spectrum.mul_scalar(pdf)
            return spectrum

    def build():
        world, prims = scenes.build_cornell(ns)
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = MyLambert(p.material.reflectivity)
        return world, prims

    assert hybrid.python_materials(build()[0]) and not hybrid.python_materials(scenes.build_cornell(ns)[0])
    monkeypatch.setattr(hybrid, "MIN_RAYS_PER_WORKER", 40)
    camera = lambda w: scenes.cornell_camera(ns, w, (12, 10), 2, 4)     # noqa: E731
    one = _render(orc, ns, monkeypatch, build, camera, 11, host_workers=1)
    three = _render(orc, ns, monkeypatch, build, camera, 11, host_workers=3)
    lowered = _render(orc, ns, monkeypatch, lambda: scenes.build_cornell(ns), camera, 11)
    assert np.array_equal(one[0], three[0]) and np.array_equal(one[1], three[1]) and one[2] == three[2]
    assert np.array_equal(one[0], lowered[0]) and one[2] == lowered[2]


def test_a_failing_material_in_a_worker_reaches_the_caller(orc, ns, monkeypatch):
    from source_amd.optical import hybrid

    class Broken(ns.Lambert):
        def evaluate_shading(self, *args):
            raise ZeroDivisionError("user bug")

    def build():
        world, prims = scenes.build_cornell(ns)
        prims[0].material = Broken()
        return world, prims

    monkeypatch.setattr(hybrid, "MIN_RAYS_PER_WORKER", 40)
    with pytest.raises(RuntimeError, match="user bug"):
        _render(orc, ns, monkeypatch, build, lambda w: scenes.cornell_camera(ns, w, (12, 10), 2, 4), 1, host_workers=2)
