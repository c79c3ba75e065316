import sys, time, os
sys.path.insert(0, os.getcwd())
import splat_renderer_amd as sr
n, w, h = sr.scene.CONFIGS["C2"]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera(); cam.setAspect(w / h); u = cam.uniforms(w, h)
dev = sr.Device(0)
pm = sr.SplatPropertyManager(dev, n); pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals); pbuf = pm.getPropertyBuffer()
r = sr.Renderer(dev, None, "rgba8unorm", n)
def timed(k):
    dev.sync(); t0 = time.perf_counter()
    for _ in range(k): r.render(u, pbuf, nbuf, None, w, h)
    dev.sync(); return (time.perf_counter() - t0) / k * 1e3
for _ in range(5): r.render(u, pbuf, nbuf, None, w, h)
print("after 5 warm-up frames:  50 frames ->", round(timed(50), 4))
print("next 50:", round(timed(50), 4), " next 50:", round(timed(50), 4))
for _ in range(1000): r.render(u, pbuf, nbuf, None, w, h)
print("after 1000 more: 50 frames ->", round(timed(50), 4), " 50 ->", round(timed(50), 4))
time.sleep(2.0)
print("after 2 s idle: 50 frames ->", round(timed(50), 4), " 50 ->", round(timed(50), 4), " 50 ->", round(timed(50), 4))
