import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("rust-pathtracer_amd")
engine = pkg.load()
def run(builder, label, w=1920, h=1080, spp=30, mb=12, L=2):
    sc = engine.create_scene(builder)
    rd = pkg.api.render_desc(w, h, spp, mb, min_bounces=1, light_samples=L, seed=1)
    sc.render(rd)
    film, prof = sc.render(rd)
    ks = list(prof.kernel_seconds)[:5]; kl = list(prof.kernel_launches)[:5]; it = list(prof.stage_items)[:5]
    print(label, "Ms/s %.1f" % (w*h*spp/prof.seconds/1e6), {n: "%.0f us x%d" % (1e6*s/max(1,l), l) for n, s, l, i in zip(("gen","ext","shade","shadow","acc"), ks, kl, it)}, flush=True)
    return film
mode = os.environ.get("PT_AMD_NO_PARK", "0")
f = run(pkg.scene.cornell_gem(), "C3 gem   no_park=" + mode)
np.save("/tmp/c3_%s.npy" % mode, f)
f = run(pkg.scene.hdri_test(), "C4 hdri  no_park=" + mode, 1024, 1024, 30, 4, 6)
np.save("/tmp/c4_%s.npy" % mode, f)
