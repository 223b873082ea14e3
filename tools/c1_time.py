import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
rt = importlib.import_module("raytracer-rust_amd")
G = os.path.join(os.getcwd(), "tests", "golden")
for name in ("spheres", "cube"):
    t0 = time.perf_counter()
    sc = rt.Scene.load(os.path.join(G, "scenes", name, name + ".yaml"), base_dir=G)
    t1 = time.perf_counter()
    ctx = rt.Context(0); ctx.upload(sc)
    t2 = time.perf_counter()
    ctx.render_image(rt.RenderParams("uniform:2", depth=5, seed=1))
    t3 = time.perf_counter()
    ts = []
    for k in range(5):
        a = time.perf_counter(); ctx.render_image(rt.RenderParams("uniform:2", depth=5, seed=k)); ts.append(time.perf_counter() - a)
    ctx.stats_reset(); ctx.render_image(rt.RenderParams("uniform:2", depth=5, seed=9)); st = ctx.stats()
    print(name, "load %.1f ms, context+upload %.1f ms, first render %.1f ms, steady render %.2f ms (%d rays, %d iterations)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, min(ts)*1e3, st["rays_total"], len(st["rays_per_iteration"])))
    ctx.close()
