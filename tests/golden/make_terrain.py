"""Regenerates the config-4 terrain fixtures (run in the build container, where /root/reference exists).

  terrain_heights.npz : the 1024 x 1024 height field of SURVEY.md 8(d) config 4, produced by the REFERENCE's own vendored
                        include/FastNoiseLite.h compiled in place (oracle/ref_noise_driver.cpp -> oracle/_ref/ref_noise):
                        FastNoiseLite(1337), Perlin, frequency 0.01, h = (int)((n + 1.0) * 33.0 * 4) + 120. Data only.
  terrain.json        : the frozen scene definition -- which columns of the field are filled (the whole field exceeds the
                        format's 2^23-texel pointer limit, see "texel_survey"), band, floor, camera pose -- and what the
                        oracle derives from it: voxels, texels, tex_dim, FNV-1a-64 of the flattened stream.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

SIZE, AMP, SEED = 1024, 4, 1337
WINDOW = {"x0": 224, "z0": 224, "nx": 576, "nz": 576}   # centred; 576^2 columns x 22.9 texels = 7.6 M < 2^23
BAND, FLOOR = 8, 20
POSE = [512.5, 420.5, 1000.5, -90.0, -20.0]


def heights():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_noise")
    tmp = "/tmp/terrain_heights.u16"
    info = json.loads(subprocess.check_output([exe, tmp, str(SIZE), str(AMP), str(SEED)]))
    h = np.fromfile(tmp, dtype="<u2").reshape(SIZE, SIZE)
    os.remove(tmp)
    return h, info


def texels_of(h, x0, z0, nx, nz, band):
    t = O.new_tree()
    O.fill_heights(t, h, x0, z0, nx, nz, band, FLOOR)
    n = O.lib().o_octree_texel_size(t)
    O.lib().o_octree_delete(t)
    return int(n)


def main():
    h, info = heights()
    np.savez_compressed(os.path.join(HERE, "terrain_heights.npz"), heights=h)
    survey = {}
    if "--survey" in sys.argv:  # why the whole field cannot be the scene: texels of the flattened tree, limit 2^23 = 8388608
        for band in (8, 4, 3):
            survey[f"1024x1024 band {band}"] = texels_of(h, 0, 0, SIZE, SIZE, band)
    t = O.new_tree()
    O.fill_heights(t, h, WINDOW["x0"], WINDOW["z0"], WINDOW["nx"], WINDOW["nz"], BAND, FLOOR)
    tex, dim = O.flatten(t)
    n_vox = int(np.minimum(BAND, h[WINDOW["z0"]:WINDOW["z0"] + WINDOW["nz"], WINDOW["x0"]:WINDOW["x0"] + WINDOW["nx"]].astype(int) - FLOOR).sum())
    out = {
        "source": "heights: the reference's include/FastNoiseLite.h (1.1.1) compiled in place by oracle/ref_noise_driver.cpp; "
                  "tree/flatten: oracle/octree_oracle.c",
        "generator": {"size": SIZE, "amp": AMP, "seed": SEED, "noise": "Perlin", "frequency": 0.01,
                      "formula": "h = (int)((GetNoise((float)x, (float)z) + 1.0) * 33.0 * 4) + 120", **info},
        "window": WINDOW, "band": BAND, "floor": FLOOR, "pose": POSE,
        "voxels_inserted": n_vox, "texels": int(tex.size // 4), "tex_dim": dim, "fnv1a64": "%016x" % O.fnv1a64(tex),
        "texel_limit": 1 << 23,
    }
    old = os.path.join(HERE, "terrain.json")
    if not survey and os.path.exists(old):
        survey = json.load(open(old)).get("texel_survey", {})
    out["texel_survey"] = survey
    json.dump(out, open(old, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
