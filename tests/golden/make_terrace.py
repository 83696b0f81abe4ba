#!/usr/bin/env python3
"""Write tests/golden/terrace/: a small open scene authored for the tests (not a reference scene).

It holds what cbox, car and pine do not: a directional light next to a point light (util/lighting.cpp:24-36), sky
behind the geometry (background pixels, G-buffer miss marker), a mirror that sees both, smooth normals that lean
away from their faces, a non-square texture in per-hit uv mode (util/polygonMesh.cpp:98-103 scales both
coordinates by the row count), a PNG behind a .jpg name in per-vertex mode, faces given as v//vn, and a bumpy
floor of a few hundred triangles so that the kd-tree has some depth.  The two textures are copies of decoder
fixtures from tests/golden/jpeg/.
"""
import math
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "terrace")

MASTER = """# A small open scene kept with the tests (tests/golden/make_terrace.py).
# x, y, z, step, astep, angle, zoom
C 0 60 260 10 5 -90 1.0
O terrace.txt
# type, light, pos, RGB values
L d 0.5 1 0.3 1 0.95 0.9 1.2
L p 60 140 80 0.9 0.9 1 4000
T wall tiles
T side paper
"""


def main():
    os.makedirs(OUT, exist_ok=True)
    shutil.copyfile(os.path.join(HERE, "jpeg", "c420_mcu.jpg"), os.path.join(OUT, "tiles.jpg"))      # 64 x 48
    shutil.copyfile(os.path.join(HERE, "jpeg", "png_rgb.png"), os.path.join(OUT, "paper.jpg"))        # PNG data, 45 x 37
    with open(os.path.join(OUT, "master.txt"), "w") as f:
        f.write(MASTER)
    v, vt, vn, faces = [], [], [], []

    def vert(x, y, z):
        v.append((x, y, z))
        return len(v)

    def norm(x, y, z):
        vn.append((x, y, z))
        return len(vn)

    for t in ((0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5)):
        vt.append(t)
    up, front, right = norm(0, 1, 0), norm(0, 0, 1), norm(1, 0, 0)

    # floor: a 12 x 12 grid over [-150, 150]^2 with gentle bumps, smooth normals from the height field
    n = 12
    def height(x, z):
        return 4.0 * math.sin(x * 0.043) * math.cos(z * 0.037)
    grid = [[0] * (n + 1) for _ in range(n + 1)]
    gnorm = [[0] * (n + 1) for _ in range(n + 1)]
    for i in range(n + 1):
        for j in range(n + 1):
            x, z = -150 + 300 * j / n, -150 + 300 * i / n
            grid[i][j] = vert(x, round(height(x, z), 4), z)
            dx = 4.0 * 0.043 * math.cos(x * 0.043) * math.cos(z * 0.037)
            dz = -4.0 * 0.037 * math.sin(x * 0.043) * math.sin(z * 0.037)
            gnorm[i][j] = norm(round(-dx, 5), 1, round(-dz, 5))
    faces.append("g floor")
    for i in range(n):
        for j in range(n):
            a, b, c, d = (i, j), (i, j + 1), (i + 1, j + 1), (i + 1, j)
            faces.append("f " + " ".join(f"{grid[p][q]}/{k + 1}/{gnorm[p][q]}" for k, (p, q) in enumerate((a, b, c, d))))

    # back wall, per-hit uv mode ("?"), non-square texture
    w = [vert(-150, -5, -150), vert(150, -5, -150), vert(150, 160, -150), vert(-150, 160, -150)]
    faces += ["g wall ?", "f " + " ".join(f"{p}/{k + 1}/{front}" for k, p in enumerate(w))]
    # side panel, per-vertex colour mode
    s = [vert(-150, -5, 150), vert(-150, -5, -150), vert(-150, 120, -150), vert(-150, 120, 150)]
    faces += ["g side", "f " + " ".join(f"{p}/{k + 1}/{right}" for k, p in enumerate(s))]
    # mirror, faces without texture coordinates
    m = [vert(-130, 5, -40), vert(-50, 5, -120), vert(-50, 115, -120), vert(-130, 115, -40)]
    mn = norm(0.70710678, 0, 0.70710678)
    faces += ["g glass 1", "f " + " ".join(f"{p}//{mn}" for p in m)]
    # pyramid with normals that lean away from the faces
    b = [vert(-10, 3, -20), vert(70, 3, -20), vert(70, 3, 60), vert(-10, 3, 60)]
    apex = vert(30, 75, 20)
    bn = [norm(-0.6, 0.5, -0.6), norm(0.6, 0.5, -0.6), norm(0.6, 0.5, 0.6), norm(-0.6, 0.5, 0.6)]
    an = norm(0.2, 1, 0.1)
    faces.append("g pyramid")
    for k in range(4):
        k2 = (k + 1) % 4
        faces.append(f"f {b[k]}/1/{bn[k]} {b[k2]}/2/{bn[k2]} {apex}/5/{an}")
    # a sliver hovering just above the floor
    sl = [vert(80, 6, 90), vert(140, 6.5, 90), vert(110, 6, 140)]
    faces += ["g sliver", "f " + " ".join(f"{p}//{an}" for p in sl)]

    with open(os.path.join(OUT, "terrace.txt"), "w") as f:
        f.write("# written by tests/golden/make_terrace.py\n")
        for p in v:
            f.write("v %g %g %g\n" % p)
        for t in vt:
            f.write("vt %g %g\n" % t)
        for q in vn:
            f.write("vn %g %g %g\n" % q)
        f.write("\n".join(faces) + "\n")


if __name__ == "__main__":
    main()
