#!/usr/bin/env python3
"""Generates the fixtures of tests/golden/extra/ (SURVEY 8f-2 / 8f-4 surface): small scenes in the reference's own
YAML + OBJ + PNG formats that exercise what scenes/cube and scenes/spheres do not:

  textured_quad.yaml  a mesh with `v/vt` faces (mesh.rs:147-157,186-215) and a Texture material that is not the cube,
                      UVs that reach u = 1 and v = 0 (materials.rs:197-230 indexes one past the edge there), seen from
                      its BACK, where the texture emits its texel (materials.rs:236-238): a deterministic image
  two_objects.yaml    an OBJ with two `o` objects and `all_objects: true` (additive key; the reference keeps the first
                      `o` only, scene.rs:169-170) -- two_objects_first.yaml is the same scene without the key
  mesh_light.yaml     an emissive OBJ mesh acting as a light (scene.rs:129-143: sample point = centroid of its first
                      emissive triangle) over a Lambertian floor

Run from the repository root:  python tests/golden/extra/make_extra.py
Paths inside the YAML files are relative to tests/golden (the tests pass it as base_dir; the reference resolves
against the working directory)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def write(name, text):
    with open(os.path.join(HERE, name), "w") as f:
        f.write(text)


def main():
    # ---- 8x4 texture with 32 distinct colours (rows differ, so a wrap to the next row is visible)
    tex = np.zeros((4, 8, 3), np.uint8)
    for y in range(4):
        for x in range(8):
            tex[y, x] = (30 * x + 20, 60 * y + 40, (37 * (x + 3 * y)) % 256)
    Image.fromarray(tex, "RGB").save(os.path.join(HERE, "tex8x4.png"))

    # ---- textured quad, `v/vt` faces, UVs on the full [0,1]^2; wound clockwise as seen from +z, so its normal
    # points to -z and a camera on the +z side sees the back face
    write("quad_uv.obj", """# v/vt faces, no normals
o Quad
v -1.6 -1.0 0.10
v 1.5 -1.1 -0.20
v 1.7 1.0 0.00
v -1.4 1.2 0.30
vt 0.0 0.0
vt 1.0 0.0
vt 1.0 1.0
vt 0.0 1.0
usemtl Tex
f 1/1 3/3 2/2
f 1/1 4/4 3/3
""")
    write("textured_quad.yaml", """width: 96
height: 64
background: [0.1, 0.2, 0.3]
camera:
  eye: [0.3, 0.2, 5.0]
  look_at: [0.0, 0.0, 0.0]
  up: [0.0, 1.0, 0.0]
  fovy: 40.0
objects:
  - type: Mesh
    path: ./extra/quad_uv.obj
    materials:
      Tex:
        type: Texture
        texture_path: ./extra/tex8x4.png
""")

    # ---- two `o` objects in one file; indices are global to the file (mesh.rs:269-289)
    write("two_o.obj", """o Left
v -2.0 -1.0 0.0
v -0.2 -1.0 0.0
v -0.2 1.0 0.0
v -2.0 1.0 0.0
usemtl Red
f 1 2 3
f 1 3 4
o Right
v 0.2 -1.0 -0.5
v 2.0 -1.0 -0.5
v 1.1 1.2 -0.5
usemtl Green
f 5 6 7
""")
    body = """width: 80
height: 48
background: [0.0, 0.0, 0.25]
camera:
  eye: [0.0, 0.0, 6.0]
  look_at: [0.0, 0.0, 0.0]
  up: [0.0, 1.0, 0.0]
  fovy: 40.0
objects:
  - type: Mesh
    path: ./extra/two_o.obj
%s    materials:
      Red:
        type: Emissive
        color: [1.0, 0.25, 0.0625]
      Green:
        type: Emissive
        color: [0.0625, 1.0, 0.25]
"""
    write("two_objects.yaml", body % "    all_objects: true\n")
    write("two_objects_first.yaml", body % "")

    # ---- an emissive mesh as the light: lamp = two triangles (the first one's centroid is the light's sample
    # point), plus a non-emissive part of the same mesh listed first
    write("lamp.obj", """o Lamp
v -0.5 2.4 -0.5
v 0.5 2.4 -0.5
v 0.5 2.4 0.5
v -0.5 2.4 0.5
v -1.5 0.0 -1.0
v -0.5 0.0 -1.0
v -1.0 1.0 -1.0
usemtl Grey
f 5 6 7
usemtl Glow
f 1 2 3
f 1 3 4
""")
    write("mesh_light.yaml", """width: 96
height: 72
background: [0.05, 0.05, 0.1]
camera:
  eye: [0.0, 1.2, 5.0]
  look_at: [0.0, 0.8, 0.0]
  up: [0.0, 1.0, 0.0]
  fovy: 45.0
objects:
  - type: Plane
    center: [0.0, 0.0, 0.0]
    normal: [0.0, 1.0, 0.0]
    material:
      type: Lambertian
      albedo: [0.6, 0.6, 0.6]
  - type: Mesh
    path: ./extra/lamp.obj
    materials:
      Grey:
        type: Lambertian
        albedo: [0.5, 0.4, 0.3]
      Glow:
        type: Emissive
        color: [4.0, 3.5, 3.0]
  - type: Sphere
    center: [0.9, 0.5, 0.3]
    radius: 0.5
    material:
      type: Metal
      albedo: [0.8, 0.8, 0.9]
      fuzziness: 0.1
""")


if __name__ == "__main__":
    main()
