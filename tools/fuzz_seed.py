import sys, importlib, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import fuzz_scenes, oracle_loader, parity_suite as ps
from util import film_metrics
pkg=importlib.import_module('rust-pathtracer_amd')
engine, oracle = pkg.load(), oracle_loader.load(pkg)
seed=int(sys.argv[1])
b=fuzz_scenes.random_scene(seed)
o, d = fuzz_scenes.random_rays(seed, 1 << 13)
se, so = engine.create_scene(b), oracle.create_scene(b)
he, ho = se.intersect(o, d), so.intersect(o, d)
print("valid equal", np.array_equal(he["valid"], ho["valid"]), int((he["valid"] != ho["valid"]).sum()))
try:
    ps.assert_hits_equal(he, ho); print("hits equal")
except AssertionError as e: print("hits differ", str(e)[:300])
idx=np.nonzero(he["valid"] != ho["valid"])[0][:5]
for i in idx: print(i, o[i], d[i], he["valid"][i], ho["valid"][i], he["t"][i], ho["t"][i], he["instance"][i], ho["instance"][i])
rd = pkg.api.render_desc(40, 32, 3, 6, light_samples=int(1 + seed % 3), seed=seed, hero_wavelengths=4 if seed % 5 == 0 else 1)
film,prof=se.render(rd); ref,rprof=so.render(rd)
bad=~np.isfinite(ref); print('nonfinite ref', bad.sum(), 'film', (~np.isfinite(film)).sum())
print(film_metrics(np.nan_to_num(film),np.nan_to_num(ref)))
print((prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits), (rprof.camera_rays, rprof.bounce_rays, rprof.shadow_rays, rprof.env_hits))
print("sweep", se.uses_leaf_sweep(), "instances", len(b.instances))
