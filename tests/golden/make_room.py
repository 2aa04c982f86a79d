"""Writes tests/golden/room.npz + room.json: the reference's only translucent scene, as DATA.

The scene is the room `src/main.cpp:505-633` builds (commented out in the shipped file, next to the terrain generator
config 4 comes from): a grass floor, three wood walls, a glass wall (alpha 40, refraction 1.5), a jelly sphere
(alpha 100, refraction 1.38) inside and a two-coloured "soccer ball" outside; materials and colours from the tables
`main.cpp:220-259`. It is the one scene the reference defines that drives `raytracing.comp:546-572` (reflect / refract,
the 8-deep ray stack) and the Beer-Lambert absorption `:482-486,512-516`.

The fixture is the ORDERED list of `octree_insert` calls those loops make (x, y, z, ColorRGBA, material row): the ball's
latitude/longitude pattern (`atan2f`, `acosf`, `floorf`, main.cpp:611-619) is evaluated here once, in float32, and stored;
neither the oracle nor the product evaluates it again. Insertion order is kept because later inserts overwrite earlier ones
(the glass wall replaces the wood of the two corners it shares with the north and south walls).

Poses (room.json) are ours: the reference has no camera preset for the room. `inside` stands in the room and looks at the
jelly sphere with the glass wall and the ball behind it; `outside` looks back through the glass wall from beyond the ball.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# main.cpp:220-232 (refraction, illumination, k) and the rows used by the room
MATERIALS = {"GRASS": (3.0, 0.0, 0.0), "WOOD": (3.0, 0.0, 0.0), "STONE": (3.0, 0.0, 0.0), "GLASS": (1.5, 0.0, 0.0),
             "JELLY": (1.38, 0.0, 0.0)}
MAT_ROWS = ["GRASS", "WOOD", "STONE", "GLASS", "JELLY"]


def rgba(r, g, b, a):   # make_color_rgba, src/color.c:9-12
    return (r << 24) | (g << 16) | (b << 8) | a


def main():
    f32 = np.float32
    out = []   # (x, y, z, color, material row)

    def put(x, y, z, color, mat):
        out.append((x, y, z, color, MAT_ROWS.index(mat)))

    x0, x1, z0, z1, floor_y, wall_h = 12, 51, 12, 51, 20, 20          # main.cpp:506-511
    for x in range(x0, x1 + 1):                                        # :514-520
        for z in range(z0, z1 + 1):
            put(x, floor_y, z, rgba(100, 200, 80, 255), "GRASS")
    for y in range(floor_y + 1, floor_y + wall_h + 1):                 # :523-551
        for x in range(x0, x1 + 1):
            put(x, y, z0, rgba(140, 90, 50, 255), "WOOD")
        for x in range(x0, x1 + 1):
            put(x, y, z1, rgba(140, 90, 50, 255), "WOOD")
        for z in range(z0, z1 + 1):
            put(x0, y, z, rgba(140, 90, 50, 255), "WOOD")
        for z in range(z0, z1 + 1):
            put(x1, y, z, rgba(100, 100, 230, 40), "GLASS")
    cx, cz, cy, radius = (x0 + x1) // 2, (z0 + z1) // 2, floor_y + 6, 5   # :554-557
    margin = f32(0.87)                                                  # :562
    for x in range(cx - radius - 1, cx + radius + 2):                   # :564-583
        for y in range(cy - radius - 1, cy + radius + 2):
            for z in range(cz - radius - 1, cz + radius + 2):
                dx, dy, dz = f32(x - cx), f32(y - cy), f32(z - cz)
                dist = np.sqrt(f32(f32(dx * dx + dy * dy) + dz * dz))
                if dist <= f32(radius) + margin:
                    put(x, y, z, rgba(240, 100, 100, 100), "JELLY")
    bx, bz, by, br = x1 + 15, (z0 + z1) // 2 + 8, floor_y + 8, 12       # :587-590
    pi = np.arccos(f32(-1.0)).astype(f32)                               # :601
    patch = f32(pi / f32(3.0))                                          # :602
    for x in range(bx - br, bx + br + 1):                               # :604-631
        for y in range(by - br, by + br + 1):
            for z in range(bz - br, bz + br + 1):
                dx, dy, dz = x - bx, y - by, z - bz
                d2 = f32(dx * dx + dy * dy + dz * dz)
                if d2 > f32(br * br):
                    continue
                rlen = np.sqrt(d2)
                if rlen < f32(1e-6):
                    rlen = f32(1e-6)
                theta = f32(np.arctan2(f32(dz), f32(dx)) + pi)
                phi = np.arccos(f32(f32(dy) / rlen))
                a = int(np.floor(f32(theta / patch)))
                b = int(np.floor(f32(phi / patch)))
                if ((a + b) & 1) == 0:
                    put(x, y, z, rgba(240, 240, 240, 255), "STONE")
                else:
                    put(x, y, z, rgba(20, 20, 20, 255), "WOOD")
    arr = np.array(out, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "room.npz"), xyz=arr[:, :3].astype(np.int16), color=arr[:, 3].astype(np.uint32),
                        material=arr[:, 4].astype(np.uint8))
    json.dump({
        "source": "src/main.cpp:505-633 (the commented-out room), materials main.cpp:220-259; written by tests/golden/make_room.py",
        "inserts": int(arr.shape[0]),
        "materials": [{"name": n, "refraction": MATERIALS[n][0], "illumination": MATERIALS[n][1], "k": MATERIALS[n][2]} for n in MAT_ROWS],
        "poses": {"inside": [14.5, 30.5, 16.5, 32.0, -10.0], "outside": [98.5, 34.5, 52.5, 197.0, -8.0]},
    }, open(os.path.join(HERE, "room.json"), "w"), indent=1)
    print(arr.shape[0], "inserts")


if __name__ == "__main__":
    main()
