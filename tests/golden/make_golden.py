"""Regenerates the committed golden fixtures (run in the build container).

  flatten.json : texel counts / tex_dim / FNV-1a-64 of octree_texture() for the three shipped maps.
                 The hashes are the ones SURVEY.md 8(c) recorded from the reference's own host code;
                 this script re-derives them with the oracle and refuses to write on a mismatch.
  camera.json  : written by oracle/_ref/ref_camera (reference Camera.hpp + glm), see oracle/Makefile.
  frames.json  : per-configuration outputs of the CPU restatement of raytracing.comp: FNV-1a-64 of the
                 RGBA8 and RG32I images, hit counts and the exact texel-fetch count F that defines the
                 algorithmic byte count B_algo = 4*F + 12*W*H.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

SURVEY_HASHES = {"dragon": "2de7c1f93b4b006c", "monu9": "dcbfa0413ceb4e8f", "nature": "41d4046e9b028993"}

# (name, map, W, H, pose(x,y,z,yaw,pitch), modes)
FRAMES = [
    ("dragon_1080p", "dragon", 1920, 1080, (63.5, 60.5, 140.5, -90.0, -10.0), (0, 1)),
    ("dragon_default_720p", "dragon", 1280, 720, (34.0, 60.0, 34.0, -90.0, 0.0), (0, 1, 2)),
    ("monu9_720p", "monu9", 1280, 720, (48.5, 60.5, 170.5, -90.0, -12.0), (0, 1)),
    ("nature_4k", "nature", 3840, 2160, (60.5, 80.5, 200.5, -90.0, -20.0), (0, 1)),
    ("dragon_256x144", "dragon", 256, 144, (63.5, 60.5, 140.5, -90.0, -10.0), (0, 1, 2)),
    ("monu9_192x108", "monu9", 192, 108, (48.5, 60.5, 170.5, -90.0, -12.0), (0, 1, 2)),
    ("nature_200x112", "nature", 200, 112, (60.5, 80.5, 200.5, -90.0, -20.0), (0, 1, 2)),
    ("dragon_inside_101x67", "dragon", 101, 67, (60.3, 30.7, 25.2, 37.0, 12.0), (0, 1, 2)),
    ("dragon_720p_full", "dragon", 1280, 720, (63.5, 60.5, 140.5, -90.0, -10.0), (2,)),
    ("dragon_1080p_full", "dragon", 1920, 1080, (63.5, 60.5, 140.5, -90.0, -10.0), (2,)),
    # BASELINE config 4: the height-field terrain of tests/golden/terrain.json (make_terrain.py), its frozen pose
    ("terrain_1080p", "terrain", 1920, 1080, None, (0, 1)),
    ("terrain_240x136", "terrain", 240, 136, None, (0, 1, 2)),
    # the full path tracer (+ the displayed frame) at the other configurations' sizes
    ("monu9_720p_full", "monu9", 1280, 720, (48.5, 60.5, 170.5, -90.0, -12.0), (2,)),
    ("terrain_1080p_full", "terrain", 1920, 1080, None, (2,)),
    ("nature_4k_full", "nature", 3840, 2160, (60.5, 80.5, 200.5, -90.0, -20.0), (2,)),
    # BASELINE config 4 at its named extent: the whole 1024 x 1024 height field, 24.07 M texels -- beyond the 2^23 a
    # reference pointer texel addresses, so the oracle reads it through its wide-pointer stream (a NON-REFERENCE extension,
    # oracle.h o_scene.wide_pointers) and the product takes it as records (vrth_world_records -> vrt_upload_records)
    ("terrain_full_1080p", "terrain_full", 1920, 1080, None, (0, 1)),
    ("terrain_full_1080p_full", "terrain_full", 1920, 1080, None, (2,)),
    ("terrain_full_240x136", "terrain_full", 240, 136, None, (0, 1, 2)),
    # the reference's own translucent scene (src/main.cpp:505-633 as data: make_room.py), from inside and outside the room
    ("room_inside_1080p", "room", 1920, 1080, "inside", (0, 1)),
    ("room_inside_1080p_full", "room", 1920, 1080, "inside", (2,)),
    ("room_inside_720p_full", "room", 1280, 720, "inside", (2,)),
    ("room_outside_1080p_full", "room", 1920, 1080, "outside", (2,)),
    ("room_outside_720p_full", "room", 1280, 720, "outside", (2,)),
    ("room_inside_256x144", "room", 256, 144, "inside", (0, 1, 2)),
    ("room_outside_256x144", "room", 256, 144, "outside", (0, 1, 2)),
]


def room_tree():
    """the ordered octree_insert calls of tests/golden/room.npz into an oracle tree"""
    import numpy as np
    d = np.load(os.path.join(HERE, "room.npz"))
    mats = json.load(open(os.path.join(HERE, "room.json")))["materials"]
    t = O.new_tree()
    L = O.lib()
    for (x, y, z), c, m in zip(d["xyz"], d["color"], d["material"]):
        mm = mats[int(m)]
        L.o_octree_insert(t, O.VoxelObj(O.IVec3(int(x), int(y), int(z)), int(c), O.Voxel(mm["refraction"], mm["illumination"], mm["k"])))
    return t


def only_has(prefix):
    if "--only" not in sys.argv:
        return True
    return any(n.startswith(prefix) for n in sys.argv[sys.argv.index("--only") + 1].split(","))


def main():
    flat = {}
    scenes = {}
    for name, want in SURVEY_HASHES.items():
        tree, ok, n = O.load_vox(os.path.join(HERE, "maps", name + ".vox"))
        tex, dim = O.flatten(tree)
        h = "%016x" % O.fnv1a64(tex)
        if h != want:
            raise SystemExit(f"{name}: oracle flatten hash {h} != recorded {want}")
        flat[name] = {"voxels_inserted": n, "texels": tex.size // 4, "bytes": int(tex.size), "tex_dim": dim,
                      "fnv1a64": h}
        scenes[name] = (tex, dim)
    json.dump({"source": "SURVEY.md 8(c) (reference host code); re-derived by oracle/", "maps": flat},
              open(os.path.join(HERE, "flatten.json"), "w"), indent=1)

    import numpy as np
    terr = json.load(open(os.path.join(HERE, "terrain.json")))
    t = O.new_tree()
    wd = terr["window"]
    O.fill_heights(t, np.load(os.path.join(HERE, "terrain_heights.npz"))["heights"], wd["x0"], wd["z0"], wd["nx"], wd["nz"],
                   terr["band"], terr["floor"])
    scenes["terrain"] = O.flatten(t)
    if "%016x" % O.fnv1a64(scenes["terrain"][0]) != terr["fnv1a64"]:
        raise SystemExit("terrain: flatten hash differs from terrain.json (run make_terrain.py)")

    wide = set()
    if only_has("terrain_full"):
        t = O.new_tree()
        O.fill_heights(t, np.load(os.path.join(HERE, "terrain_heights.npz"))["heights"], 0, 0, 1024, 1024, terr["band"], terr["floor"])
        scenes["terrain_full"] = O.flatten(t, wide=True)
        wide.add("terrain_full")
        O.lib().o_octree_delete(t)
    room = json.load(open(os.path.join(HERE, "room.json")))
    scenes["room"] = O.flatten(room_tree())

    # `make_golden.py --only name[,name...]`: compute only these entries and merge them into the committed frames.json
    only = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else None
    frames = json.load(open(os.path.join(HERE, "frames.json")))["frames"] if only else {}
    for name, m, W, H, pose, modes in FRAMES:
        if only is not None and name not in only:
            continue
        if pose is None:
            pose = tuple(terr["pose"])
        elif isinstance(pose, str):
            pose = tuple(room["poses"][pose])
        tex, dim = scenes[m]
        (ip, iv, cp), _ = O.camera_ubo(pose[:3], pose[3], pose[4], W, H)
        s = O.make_scene(tex, dim, ip, iv, cp, wide=m in wide)
        for mode in modes:
            rgba, idd, fm, st = O.render(s, W, H, mode, want_fetch_map=name.endswith('1080p') or name.endswith('1080p_full'))
            frames[f"{name}/mode{mode}"] = {
                "map": m, "width": W, "height": H, "pose": list(pose), "mode": mode,
                "rgba_fnv1a64": "%016x" % O.fnv1a64(rgba), "id_dist_fnv1a64": "%016x" % O.fnv1a64(idd),
                "hits": st["hits"], "fetches": st["fetches"], "finds": st["finds"], "steps": st["steps"],
                "root_restarts": st["root_restarts"], "shadow_rays": st["shadow_rays"],
                "b_algo_bytes": 4 * st["fetches"] + 12 * W * H,
            }
            if m in wide:
                frames[f"{name}/mode{mode}"]["oracle_stream"] = "wide-pointer extension (non-reference): %d texels" % (tex.size // 4)
            if mode == 2 and name.endswith("_full"):  # what the reference puts on screen: quad.frag over the two images
                frames[f"{name}/mode{mode}"]["shown_fnv1a64"] = "%016x" % O.fnv1a64(O.denoise(rgba, idd))
            if fm is not None:  # per-row fetch totals: exact B_algo of any row shard of the bench frame
                frames[f"{name}/mode{mode}"]["row_fetches"] = [int(v) for v in fm.sum(axis=1)]
            print(name, mode, frames[f"{name}/mode{mode}"]["rgba_fnv1a64"], st["fetches"] / (W * H))
    json.dump({"source": "oracle/rt_oracle.c (CPU restatement of raytracing.comp; shader parity unpinned)",
               "frames": frames}, open(os.path.join(HERE, "frames.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
