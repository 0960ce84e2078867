"""Synthetic scenes in the reference's scene.xml / OBJ formats.

The reference ships a single scene (``assets/scene.xml``) and no mesh asset, so
the benchmark scenes are generated: deterministic, parameterised only by the
sizes below.  Vertex coordinates are quantised to 6 decimals *before* use so
that the in-memory arrays and an OBJ text file written from them parse to the
same binary32 values.
"""
import numpy as np

PHASES = ((5.58423974, 6.03632667), (4.13504349, 0.95037259),
          (4.35124413, 0.95581374), (2.04381011, 1.07836715))


def _quant6(x):
    """Round to 6 decimals as integers of micro-units (what '%.6f' text would hold)."""
    return np.rint(np.asarray(x, np.float64) * 1e6).astype(np.int64)


def _from_q(q):
    return (q.astype(np.float64) / 1e6).astype(np.float32)


def heightfield(nx=1000, ny=500, x_half=20.0, y_half=10.0):
    """Height-field grid of nx*ny quads = 2*nx*ny triangles.

    Returns dict(v [nv,3] f32, vt [nv,3] f32, f [nf,3] u32, qv, qvt) where qv/qvt
    are the micro-unit integers the floats were derived from.
    z = sum_o 1.5*2^-o * sin(2^o*0.7x+phi_o) * cos(2^o*0.9y+psi_o)
    """
    xs = np.linspace(-x_half, x_half, nx + 1)
    ys = np.linspace(-y_half, y_half, ny + 1)
    X, Y = np.meshgrid(xs, ys)  # row-major, x fastest
    Z = np.zeros_like(X)
    for o, (phi, psi) in enumerate(PHASES):
        Z += 1.5 * 2.0 ** (-o) * np.sin(2.0 ** o * 0.7 * X + phi) * np.cos(2.0 ** o * 0.9 * Y + psi)
    qv = _quant6(np.stack([X, Y, Z], -1).reshape(-1, 3))
    qvt = _quant6(np.stack([X / (2 * x_half) + 0.5, Y / (2 * y_half) + 0.5, np.zeros_like(X)], -1).reshape(-1, 3))
    idx = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1).astype(np.uint32)
    a, b, c, d = idx, idx + 1, idx + nx + 2, idx + nx + 1
    f = np.empty((idx.size * 2, 3), np.uint32)
    f[0::2] = np.stack([a, b, c], -1)
    f[1::2] = np.stack([a, c, d], -1)
    return dict(v=_from_q(qv), vt=_from_q(qvt), f=f, qv=qv, qvt=qvt)


def _fmt6(q):
    s = "-" if q < 0 else ""
    q = abs(int(q))
    return "%s%d.%06d" % (s, q // 1000000, q % 1000000)


def write_obj(path, mesh):
    """OBJ text with `v`, `vt` and `f a/a b/b c/c` records (no normals: the reader computes them)."""
    with open(path, "w") as fp:
        for r in mesh["qv"]:
            fp.write("v %s %s %s\n" % (_fmt6(r[0]), _fmt6(r[1]), _fmt6(r[2])))
        for r in mesh["qvt"]:
            fp.write("vt %s %s %s\n" % (_fmt6(r[0]), _fmt6(r[1]), _fmt6(r[2])))
        for a, b, c in mesh["f"] + 1:
            fp.write("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c))


def mesh_scene_xml(obj_name="mesh_1m.obj", width=1920, height=1080):
    """1M-triangle configuration: mesh + glass sphere + mirror sphere + floor, one spherical light."""
    return f"""<xml>
  <scene>
    <object type="obj" name="{obj_name}" material="smooth"/>
    <object type="sphere" name="glass" material="glass">
      <scale value="3"/>
      <translate x="6" y="-2" z="6"/>
    </object>
    <object type="sphere" name="mirror" material="mirror">
      <scale value="3"/>
      <translate x="-6" y="2" z="6"/>
    </object>
    <object type="plane" name="floor" material="floor">
      <scale value="60"/>
      <translate z="-4"/>
    </object>
    <material type="blinn" name="smooth">
      <diffuse value="0.7"/>
      <specular value="0.3"/>
      <glossiness value="64"/>
    </material>
    <material type="blinn" name="glass">
      <diffuse value="0"/>
      <specular value="0"/>
      <glossiness value="1024"/>
      <refraction value="0.9" index="1.52"/>
    </material>
    <material type="blinn" name="mirror">
      <diffuse value="0"/>
      <specular value="0.9"/>
      <glossiness value="128"/>
    </material>
    <material type="blinn" name="floor">
      <diffuse value="0.7"/>
      <specular value="0"/>
    </material>
    <light type="point" name="pointLight">
      <intensity value="2500"/>
      <position x="0" y="0" z="30"/>
      <size value="1"/>
    </light>
  </scene>
  <camera gamma="sRGB">
    <position x="0" y="-45" z="22"/>
    <target x="0" y="0" z="0"/>
    <up x="0" y="0" z="1"/>
    <fov value="40"/>
    <width value="{width}"/>
    <height value="{height}"/>
  </camera>
</xml>
"""


def cornell_scene_xml(width=1280, height=720):
    """The scene the reference ships with (its only one): a closed box of five walls around a mirror, a glass and a
    glossy sphere, one spherical light under the ceiling.  Written out from the tables below; the host-side tests
    check that the library's reader turns this text into exactly the arrays the reference's loader produced for its
    own file."""
    walls = (("WallBottom", "wall", None, "z", -12), ("WallTop", "wall", (180, "x"), "z", 12),
             ("WallBack", "wall", (90, "x"), "y", 20), ("WallLeft", "wallRed", (90, "y"), "x", -15),
             ("WallRight", "wallGreen", (-90, "y"), "x", 15))
    spheres = (("sphere1", "reflective", 6, (0, 4, 6)), ("sphere2", "refractive", 5, (9, -3, 5)), ("sphere3", "smooth", 5, (-9, -3, 5)))
    out = ["<xml>", "  <scene>", '    <object name="box">', '      <translate x="0" y="0" z="12"/>']
    for name, mtl, rot, axis, off in walls:
        out.append(f'      <object type="plane" name="{name}" material="{mtl}">')
        out.append('        <scale value="32"/>')
        if rot:
            out.append(f'        <rotate angle="{rot[0]}" {rot[1]}="1"/>')
        out.append(f'        <translate {axis}="{off}"/>')
        out.append("      </object>")
    out.append("    </object>")
    for name, mtl, r, (x, y, z) in spheres:
        out += [f'    <object type="sphere" name="{name}" material="{mtl}">', f'      <scale value="{r}"/>',
                f'      <translate x="{x}" y="{y}" z="{z}"/>', "    </object>"]
    for name, rgb in (("wall", (1, 1, 1)), ("wallRed", (1.0, 0.2, 0.2)), ("wallGreen", (0.2, 1.0, 0.2))):
        out += [f'    <material type="blinn" name="{name}">',
                f'      <diffuse r="{rgb[0]}" g="{rgb[1]}" b="{rgb[2]}" value="0.7"/>', '      <specular value="0"/>', "    </material>"]
    out += ['    <material type="blinn" name="reflective"><diffuse value="0"/><specular value="0.9"/><glossiness value="128"/>'
            '<reflection value="0.9"/></material>',
            '    <material type="blinn" name="refractive"><diffuse value="0"/><specular value="0"/><glossiness value="1024"/>'
            '<refraction value="0.9" index="1.52"/></material>',
            '    <material type="blinn" name="smooth"><diffuse value="0.7"/><specular value="0.3"/><glossiness value="64"/></material>',
            '    <light type="point" name="pointLight"><intensity value="500"/><position x="0" y="0" z="18"/>'
            '<size value="1"/><attenuation value="1"/></light>',
            "  </scene>", '  <camera gamma="sRGB">', '    <position x="0" y="-60" z="12"/>', '    <target x="0" y="0" z="12"/>',
            '    <up x="0" y="0" z="1"/>', '    <fov value="30"/>', f'    <width value="{width}"/>', f'    <height value="{height}"/>',
            "  </camera>", "</xml>", ""]
    return "\n".join(out)


class _PCG32:
    """PCG32 with the reference's one-argument seeding (src/rng.h:34,38,90-98); scene generation only."""
    M = 0x5851F42D4C957F2D
    MASK = (1 << 64) - 1

    def __init__(self, seq):
        v = seq
        v ^= v >> 31; v = (v * 0x7FB5D329728EA185) & self.MASK
        v ^= v >> 27; v = (v * 0x81DADEF4BC2DD44D) & self.MASK
        v ^= v >> 33
        self.state, self.inc = 0, ((seq << 1) | 1) & self.MASK
        self.u32(); self.state = (self.state + v) & self.MASK; self.u32()

    def u32(self):
        old = self.state
        self.state = (old * self.M + self.inc) & self.MASK
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def f32(self):
        r = np.float32(self.u32()) * np.float32(2.0 ** -32)
        return float(min(r, np.float32(1.0 - 2.0 ** -24)))


_PRESETS = (
    dict(kd=(.7, .7, .7), ks=0.0, g=20),
    dict(kd=(.7, .14, .14), ks=0.0, g=20),
    dict(kd=(.14, .7, .14), ks=0.0, g=20),
    dict(kd=(.7, .7, .7), ks=0.3, g=64),
    dict(kd=(.5, .5, .5), ks=0.3, g=20),
    dict(kd=(0, 0, 0), ks=0.9, g=128, kr=0.9),
    dict(kd=(0, 0, 0), ks=0.0, g=1024, kt=0.9, ior=1.52),
    dict(kd=(.35, .35, .35), ks=0.6, g=128),
)


def spheres_scene_xml(seed=1001, grid=8, width=1920, height=1080):
    """Sphere/plane-only configuration: grid*grid spheres over a floor and a back wall."""
    rng = _PCG32(seed)
    out = ["<xml>", "  <scene>",
           '    <object type="plane" name="floor" material="wall"><scale value="40"/></object>',
           '    <object type="plane" name="back" material="wall"><scale value="40"/>'
           '<rotate angle="90" x="1"/><translate y="20"/></object>']
    half = (grid - 1) / 2.0
    for iy in range(grid):
        for ix in range(grid):
            r = 0.8 + 0.8 * rng.f32()
            m = int(8 * rng.f32())
            out.append(f'    <object type="sphere" name="s{iy}_{ix}" material="m{m}"><scale value="{r:.9g}"/>'
                       f'<translate x="{(ix - half) * 4.5:.9g}" y="{(iy - half) * 4.5:.9g}" z="{r:.9g}"/></object>')
    out.append('    <material type="blinn" name="wall"><diffuse value="0.7"/><specular value="0"/></material>')
    for i, p in enumerate(_PRESETS):
        kd = p["kd"]
        s = [f'    <material type="blinn" name="m{i}">',
             f'<diffuse r="{kd[0]:.9g}" g="{kd[1]:.9g}" b="{kd[2]:.9g}"/>',
             f'<specular value="{p["ks"]:.9g}"/><glossiness value="{p["g"]}"/>']
        if "kr" in p:
            s.append(f'<reflection value="{p["kr"]:.9g}"/>')
        if "kt" in p:
            s.append(f'<refraction value="{p["kt"]:.9g}" index="{p["ior"]:.9g}"/>')
        s.append("</material>")
        out.append("".join(s))
    out += ['    <light type="point" name="pointLight"><intensity value="1500"/>'
            '<position x="0" y="0" z="30"/><size value="1"/></light>',
            "  </scene>",
            '  <camera gamma="sRGB">', '    <position x="0" y="-45" z="18"/>', '    <target x="0" y="0" z="2"/>',
            '    <up x="0" y="0" z="1"/>', '    <fov value="40"/>',
            f'    <width value="{width}"/>', f'    <height value="{height}"/>', "  </camera>", "</xml>", ""]
    return "\n".join(out)


def lights_scene_xml(width=96, height=54, grid=3):
    """The sphere grid lit by all three kinds of light the scene format has: the spherical light (first, as the
    path integrator wants it), an ambient light and a directional one whose shadows fall across the floor."""
    xml = spheres_scene_xml(seed=1001, grid=grid, width=width, height=height)
    more = ('    <light type="ambient" name="ambientLight"><intensity r="0.12" g="0.15" b="0.2"/></light>\n'
            '    <light type="direct" name="directLight"><intensity r="0.8" g="0.7" b="0.5"/>'
            '<direction x="0.5" y="0.35" z="-1"/></light>\n')
    return xml.replace("  </scene>", more + "  </scene>", 1)


def mesh_and_spheres_scene_xml(obj_names=("mesh_a.obj", "mesh_b.obj"), grid=8, width=160, height=90, seed=77):
    """Triangle meshes among many objects: the height field twice (the second one smaller, turned and lifted, so that
    rays meet two meshes), grid*grid small spheres hovering over them, a floor.  For the scene walks that look at
    objects through the hierarchy over their boxes."""
    rng = _PCG32(seed)
    out = ["<xml>", "  <scene>",
           f'    <object type="obj" name="{obj_names[0]}" material="smooth"/>',
           f'    <object type="obj" name="{obj_names[1]}" material="m3"><scale value="0.5"/><rotate angle="35" z="1"/>'
           '<translate x="4" y="-6" z="7"/></object>',
           '    <object type="plane" name="floor" material="wall"><scale value="60"/><translate z="-4"/></object>']
    half = (grid - 1) / 2.0
    for iy in range(grid):
        for ix in range(grid):
            r = 0.3 + 0.5 * rng.f32()
            m = int(8 * rng.f32())
            z = 4.0 + 3.0 * rng.f32()
            out.append(f'    <object type="sphere" name="s{iy}_{ix}" material="m{m}"><scale value="{r:.9g}"/>'
                       f'<translate x="{(ix - half) * 4.2:.9g}" y="{(iy - half) * 2.2:.9g}" z="{z:.9g}"/></object>')
    out.append('    <material type="blinn" name="wall"><diffuse value="0.7"/><specular value="0"/></material>')
    out.append('    <material type="blinn" name="smooth"><diffuse value="0.7"/><specular value="0.3"/><glossiness value="64"/></material>')
    for i, p in enumerate(_PRESETS):
        kd = p["kd"]
        t = [f'    <material type="blinn" name="m{i}">', f'<diffuse r="{kd[0]:.9g}" g="{kd[1]:.9g}" b="{kd[2]:.9g}"/>',
             f'<specular value="{p["ks"]:.9g}"/><glossiness value="{p["g"]}"/>']
        if "kr" in p:
            t.append(f'<reflection value="{p["kr"]:.9g}"/>')
        if "kt" in p:
            t.append(f'<refraction value="{p["kt"]:.9g}" index="{p["ior"]:.9g}"/>')
        t.append("</material>")
        out.append("".join(t))
    out += ['    <light type="point" name="pointLight"><intensity value="2500"/><position x="0" y="0" z="30"/><size value="1"/></light>',
            "  </scene>", '  <camera gamma="sRGB">', '    <position x="0" y="-45" z="22"/>', '    <target x="0" y="0" z="0"/>',
            '    <up x="0" y="0" z="1"/>', '    <fov value="40"/>', f'    <width value="{width}"/>', f'    <height value="{height}"/>',
            "  </camera>", "</xml>", ""]
    return "\n".join(out)


def nested_scene_xml(width=320, height=180):
    """Small closed room built from a two-level node hierarchy with rotations,
    non-uniform scales and a thin-lens camera; exercises every transform path."""
    return f"""<xml>
  <scene>
    <object name="room">
      <rotate angle="12" z="1"/>
      <translate x="1" y="2" z="10"/>
      <object type="plane" name="floor" material="white">
        <scale x="18" y="22" z="1"/>
        <translate z="-10"/>
      </object>
      <object type="plane" name="ceil" material="white">
        <scale value="22"/>
        <rotate angle="180" y="1"/>
        <translate z="10"/>
      </object>
      <object type="plane" name="back" material="blue">
        <scale value="22"/>
        <rotate angle="90" x="1"/>
        <translate y="16"/>
      </object>
      <object type="plane" name="left" material="red">
        <scale value="22"/>
        <rotate angle="90" y="1"/>
        <translate x="-13"/>
      </object>
      <object type="plane" name="right" material="white">
        <scale value="22"/>
        <rotate angle="-90" y="1"/>
        <translate x="13"/>
      </object>
      <object name="group">
        <scale x="1.5" y="1" z="0.75"/>
        <rotate angle="30" x="1" y="1" z="0.5"/>
        <translate x="-2" y="1" z="-4"/>
        <object type="sphere" name="ellipsoid" material="glossy">
          <scale value="3"/>
        </object>
        <object type="sphere" name="moon" material="glass">
          <scale value="1.25"/>
          <translate x="4" y="-3" z="2"/>
        </object>
      </object>
    </object>
    <object type="sphere" name="mirror" material="mirror">
      <scale value="3.5"/>
      <translate x="6" y="4" z="4"/>
    </object>
    <material type="blinn" name="white"><diffuse value="0.75"/><specular value="0"/></material>
    <material type="blinn" name="red"><diffuse r="0.8" g="0.15" b="0.1"/><specular value="0"/></material>
    <material type="blinn" name="blue"><diffuse r="0.15" g="0.2" b="0.8"/><specular value="0.1"/><glossiness value="30"/></material>
    <material type="blinn" name="glossy"><diffuse r="0.5" g="0.4" b="0.1"/><specular value="0.4"/><glossiness value="48"/></material>
    <material type="blinn" name="glass"><diffuse value="0"/><specular value="0.05"/><glossiness value="512"/><refraction value="0.9" index="1.45"/></material>
    <material type="blinn" name="mirror"><diffuse value="0"/><specular value="0.9"/><glossiness value="200"/></material>
    <light type="point" name="lamp">
      <intensity r="420" g="400" b="380"/>
      <position x="1" y="1" z="17"/>
      <size value="1.25"/>
    </light>
  </scene>
  <camera gamma="sRGB">
    <position x="2" y="-50" z="11"/>
    <target x="0" y="0" z="9"/>
    <up x="0" y="0" z="1"/>
    <fov value="32"/>
    <focaldist value="48"/>
    <dof value="0.6"/>
    <width value="{width}"/>
    <height value="{height}"/>
  </camera>
</xml>
"""


def texture_image(width=24, height=16, seed=5):
    """A small RGB image (smooth gradients plus a little noise) as [h, w, 3] uint8."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:height, 0:width]
    img = np.stack([128 + 100 * np.sin(2 * np.pi * x / width), 128 + 100 * np.cos(2 * np.pi * y / height),
                    255.0 * ((x // 4 + y // 4) % 2)], -1) + rng.randint(-20, 21, (height, width, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def write_ppm(path, img):
    """Binary PPM (P6) as the reference's LoadPPM reads it."""
    h, w, _ = img.shape
    with open(path, "wb") as fp:
        fp.write(b"P6\n# generated\n%d %d\n255\n" % (w, h))
        fp.write(np.ascontiguousarray(img, np.uint8).tobytes())


def textured_scene_xml(image="tex.ppm", width=96, height=54):
    """Open scene that exercises every texture path: a checkerboard floor with a texture transformation, an image-mapped
    sphere, mirror and glass spheres (secondary rays leave the scene: environment map), a textured background."""
    return f"""<xml>
  <scene>
    <object type="plane" name="floor" material="checker"><scale value="30"/></object>
    <object type="sphere" name="globe" material="mapped"><scale value="4"/><rotate angle="25" z="1"/><translate x="-7" y="2" z="4"/></object>
    <object type="sphere" name="mirror" material="mirror"><scale value="3.5"/><translate x="1" y="5" z="3.5"/></object>
    <object type="sphere" name="glass" material="glass"><scale value="3"/><translate x="7" y="-2" z="3"/></object>
    <material type="blinn" name="checker">
      <diffuse value="0.8" texture="checkerboard">
        <color1 r="0.9" g="0.2" b="0.1"/><color2 value="0.95"/>
        <scale x="0.25" y="0.125" z="1"/><rotate angle="30" z="1"/><translate x="0.1" y="0.2"/>
      </diffuse>
      <specular value="0.1"/><glossiness value="40"/>
    </material>
    <material type="blinn" name="mapped">
      <diffuse value="0.9" texture="{image}"><scale x="0.5" y="1" z="1"/></diffuse>
      <specular value="0.2"/><glossiness value="30"/>
    </material>
    <material type="blinn" name="mirror"><diffuse value="0"/><specular value="0.8"/><glossiness value="300"/><reflection value="0.8"/></material>
    <material type="blinn" name="glass"><diffuse value="0"/><specular value="0.1"/><glossiness value="800"/><refraction value="0.9" index="1.5"/></material>
    <light type="point" name="lamp"><intensity value="900"/><position x="-4" y="-8" z="24"/><size value="1.5"/></light>
    <background r="0.6" g="0.7" b="1.0" texture="checkerboard">
      <color1 value="0.3"/><color2 r="0.9" g="0.95" b="1"/><scale value="0.2"/>
    </background>
    <environment value="0.9" texture="{image}"><rotate angle="45" z="1"/></environment>
  </scene>
  <camera gamma="sRGB">
    <position x="0" y="-38" z="14"/>
    <target x="0" y="0" z="3"/>
    <up x="0" y="0" z="1"/>
    <fov value="38"/>
    <width value="{width}"/>
    <height value="{height}"/>
  </camera>
</xml>
"""


# ---- PNG files (texture assets for the tests: every colour type, bit depth, filter type, interlaced or not) ----
def _png_pack_rows(samples, depth):
    """[h, w, channels] integer samples -> list of packed scanlines (bytes)."""
    h, w, ch = samples.shape
    if depth == 8:
        return [samples[y].astype(np.uint8).tobytes() for y in range(h)]
    if depth == 16:
        return [samples[y].astype(">u2").tobytes() for y in range(h)]
    per = 8 // depth
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1).astype(np.uint8)
        flat = np.concatenate([flat, np.zeros((-len(flat)) % per, np.uint8)]).reshape(-1, per)
        b = np.zeros(len(flat), np.uint8)
        for k in range(per):
            b |= flat[:, k] << (8 - depth * (k + 1))
        rows.append(b.tobytes())
    return rows


def _png_filter(rows, bpp):
    """Scanlines -> filtered stream; row y uses filter type y % 5 (all five types appear)."""
    out, prev = bytearray(), None
    for y, row in enumerate(rows):
        t = y % 5
        cur = np.frombuffer(row, np.uint8).astype(np.int32)
        up = np.frombuffer(prev, np.uint8).astype(np.int32) if prev is not None else np.zeros_like(cur)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        if t == 0:
            f = cur
        elif t == 1:
            f = cur - left
        elif t == 2:
            f = cur - up
        elif t == 3:
            f = cur - (left + up) // 2
        else:
            p = left + up - upleft
            pa, pb, pc = abs(p - left), abs(p - up), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            f = cur - pred
        out.append(t)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = row
    return bytes(out)


def png_bytes(samples, color_type, depth, interlace=False, palette=None):
    """A PNG file of the given colour type (0 grey, 2 RGB, 3 palette, 4 grey+alpha, 6 RGBA) and bit depth from integer
    samples [h, w, channels]; rows cycle through the five filter types; interlace=True writes the seven Adam7 passes."""
    import struct
    import zlib
    samples = np.asarray(samples)
    h, w, ch = samples.shape
    bits = ch * depth
    bpp = max(1, bits // 8)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    stream = b""
    if interlace:
        for sx, sy, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[sy::dy, sx::dx]
            if sub.shape[0] and sub.shape[1]:
                stream += _png_filter(_png_pack_rows(sub, depth), bpp)
    else:
        stream = _png_filter(_png_pack_rows(samples, depth), bpp)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    out += chunk(b"tEXt", b"Comment\x00generated test asset")
    comp = zlib.compress(stream, 9)
    half = len(comp) // 2
    out += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:])  # image data may be split over chunks
    return out + chunk(b"IEND", b"")


def png_cases(seed=11, width=13, height=11):
    """{name: PNG file bytes}: every legal (colour type, bit depth) pair, plain and interlaced."""
    rng = np.random.RandomState(seed)
    cases = {}
    for ctype, depths, ch in ((0, (1, 2, 4, 8, 16), 1), (2, (8, 16), 3), (3, (1, 2, 4, 8), 1), (4, (8, 16), 2), (6, (8, 16), 4)):
        for depth in depths:
            for il in (False, True):
                pal = None
                hi = 1 << depth
                if ctype == 3:
                    n = min(hi, 200)  # some indices of the 8-bit case point past the palette
                    pal = rng.randint(0, 256, (n, 3))
                s = rng.randint(0, hi, (height, width, ch))
                cases[f"c{ctype}_d{depth}_{'adam7' if il else 'plain'}"] = png_bytes(s, ctype, depth, il, pal)
    cases["rgb8_1x1"] = png_bytes(rng.randint(0, 256, (1, 1, 3)), 2, 8)
    cases["rgb8_3x2_adam7"] = png_bytes(rng.randint(0, 256, (2, 3, 3)), 2, 8, True)
    return cases


# ---- an OBJ object that brings its own materials (usemtl groups + a material library) ----
def multi_material_mesh(nx=12, ny=8):
    """A small height field whose quads name one of three materials in turn - groups interleaved in the file, so the
    loader has to regroup the faces.  Returns (obj text, mtl text)."""
    m = heightfield(nx, ny, 8.0, 5.0)
    v, vt, f = m["v"], m["vt"], m["f"]
    lines = ["mtllib multi.mtl"]
    lines += ["v %.6f %.6f %.6f" % tuple(p) for p in v.astype(np.float64)]
    lines += ["vt %.6f %.6f %.6f" % tuple(p) for p in vt.astype(np.float64)]
    names = ("red_matte", "green_gloss", "mirror_blue")
    for q in range(nx * ny):
        lines.append("usemtl " + names[(q + q // nx) % 3])
        for t in (2 * q, 2 * q + 1):
            a, b, c = (int(i) + 1 for i in f[t])
            lines.append(f"f {a}/{a} {b}/{b} {c}/{c}")
    mtl = """# three materials; a fourth that no face names
newmtl red_matte
Kd 0.8 0.15 0.1
Ks 0.05
Ns 12
illum 2
newmtl unused
Kd 0 0 1
newmtl green_gloss
Kd 0.1 0.7 0.2
Ks 0.4 0.4 0.4
Ns 90
Ni 1.3
illum 2
newmtl mirror_blue
Kd 0.05 0.05 0.3
Ks 0.7 0.7 0.8
Ns 300
illum 3
"""
    return "\n".join(lines) + "\n", mtl


def multi_material_scene_xml(obj_name="multi.obj", width=96, height=54):
    """The object above (no material attribute: its node gets the OBJ file's materials) over a floor, one sphere."""
    return f"""<xml>
  <scene>
    <object type="obj" name="{obj_name}"><translate z="2"/></object>
    <object type="plane" name="floor" material="floor"><scale value="40"/><translate z="-3"/></object>
    <object type="sphere" name="ball" material="ball"><scale value="2"/><translate x="5" y="-3" z="5"/></object>
    <material type="blinn" name="floor"><diffuse value="0.7"/><specular value="0"/></material>
    <material type="blinn" name="ball"><diffuse r="0.7" g="0.6" b="0.2"/><specular value="0.3"/><glossiness value="50"/></material>
    <light type="point" name="lamp"><intensity value="1200"/><position x="2" y="-6" z="25"/><size value="1"/></light>
    <background value="0.1"/>
  </scene>
  <camera gamma="sRGB">
    <position x="0" y="-26" z="14"/>
    <target x="0" y="0" z="1"/>
    <up x="0" y="0" z="1"/>
    <fov value="40"/>
    <width value="{width}"/>
    <height value="{height}"/>
  </camera>
</xml>
"""
