import sys; sys.path.insert(0, '/root/repo')
from source_amd import api as rs
world = rs.World()
rs.Sphere(0.5, world, rs.translate(0, 0, 2), rs.UniformSurfaceEmitter(rs.ConstantSF(1.0)))
hit = world.hit(rs.Ray(rs.Point3D(0, 0, 0), rs.Vector3D(0, 0, 1)))
print(hit.ray_distance, hit.primitive)
pipe = rs.SpectralRadiancePipeline2D()
cam = rs.PinholeCamera((512, 512), parent=world, pipelines=[pipe])
cam.render_engine = rs.SerialEngine()
cam.quiet = True
cam.observe()
print(pipe.frame.mean.shape, pipe.frame.mean[256, 256, :3], pipe.frame.samples.max())
