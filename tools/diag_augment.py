import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import AUG_CASE, synthetic_slide
from mclstexp_amd import input_pipeline as ip
from oracle import ref_input as R
img = synthetic_slide()
dimg = ip.to_device_image(img)
r = AUG_CASE["r"]
base = {k: np.asarray(AUG_CASE[k]) for k in ("order", "brightness", "contrast", "saturation", "hflip", "angle")}
def run(d, tag):
    out = ip.her2st_train_patches(dimg, AUG_CASE["centers_xy"], r=r, draws=d).cpu().numpy()
    for i, (x, y) in enumerate(AUG_CASE["centers_xy"]):
        ref = R.her2st_train_transform(R.crop(img, y, x, r), d["order"][i], float(d["brightness"][i]), float(d["contrast"][i]),
                                       float(d["saturation"][i]), bool(d["hflip"][i]), float(d["angle"][i]))
        bad = (out[i] != ref)
        if bad.any():
            idx = np.argwhere(bad)[0]
            print(tag, "patch", i, "mismatches", int(bad.sum()), "first", idx, out[i][tuple(idx)] * 255, ref[tuple(idx)] * 255,
                  "order", d["order"][i], d["brightness"][i], d["contrast"][i], d["saturation"][i], d["hflip"][i], d["angle"][i])
run(base, "full")
d = dict(base); d["angle"] = np.zeros(8); d["hflip"] = np.zeros(8, dtype=int); run(d, "jitter-only")
for name in ("brightness", "contrast", "saturation"):
    d = dict(base); d["angle"] = np.zeros(8); d["hflip"] = np.zeros(8, dtype=int)
    for other in ("brightness", "contrast", "saturation"):
        if other != name: d[other] = np.ones(8)
    run(d, name + "-only")
d = dict(base); d["brightness"] = d["contrast"] = d["saturation"] = np.ones(8); run(d, "geom-only")
print("done")
