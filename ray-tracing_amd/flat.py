"""Build a flattened scene (include/rt_hip.h RtSceneDesc) directly from Python: the pointer-free form the device
boundary consumes.  Used by the JSON scene loader (the reference's codegen/data/*.json vocabulary,
codegen/src/scenes_file.rs:19-152) and by the randomized parity tests.  Pure host code: ctypes + math."""
import ctypes as C
import json
import math

from . import _ffi

SPHERE, MOVING_SPHERE, RECT_XY, RECT_XZ, RECT_YZ, CUBOID = range(6)
LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, ISOTROPIC, FROSTED = range(6)
SOLID, CHECKER_3D, CHECKER_UV, NOISE, IMAGE = range(5)
TRANSLATE, ROTATE_Y = 0, 1


def _unit(v):
    n = math.sqrt(sum(c * c for c in v))
    return [c / n for c in v]


def _cross(a, b):
    return [a[1] * b[2] - b[1] * a[2], b[0] * a[2] - a[0] * b[2], a[0] * b[1] - b[0] * a[1]]


class FlatScene:
    def __init__(self):
        self.prims, self.materials, self.textures, self.xforms = [], [], [], []
        self.world, self.lights = [], []
        self.nodes = []  # (box6, left, right)
        self.perlins, self.images = [], []
        self.background = [0.0, 0.0, 0.0]
        self.cam = None
        self._keep = None

    # ---- records -----------------------------------------------------------------------------
    def texture(self, kind, color=(0, 0, 0), odd=0, even=0, table=0, scale=0.0):
        self.textures.append((kind, odd, even, table, tuple(float(c) for c in color), float(scale)))
        return len(self.textures) - 1

    def solid(self, r, g, b):
        return self.texture(SOLID, (r, g, b))

    def material(self, kind, texture=0, albedo=(0, 0, 0), fuzz=0.0, ref_idx=0.0):
        self.materials.append((kind, texture, tuple(float(c) for c in albedo), float(fuzz), float(ref_idx)))
        return len(self.materials) - 1

    def prim(self, kind, material, d):
        d = [float(x) for x in d] + [0.0] * (10 - len(d))
        self.prims.append((kind, material, d))
        return len(self.prims) - 1

    def sphere(self, c, r, m):
        return self.prim(SPHERE, m, [*c, r])

    def moving_sphere(self, c1, c2, t1, t2, r, m):
        return self.prim(MOVING_SPHERE, m, [*c1, *c2, t1, t2, r])

    def rect(self, axis, a1, a2, b1, b2, k, face, m):
        return self.prim(RECT_XY + axis, m, [a1, a2, b1, b2, k, face])

    def cuboid(self, mn, mx, m):
        return self.prim(CUBOID, m, [*mn, *mx])

    def xform_chain(self, chain):
        """chain: outermost first, items ('t', (x,y,z)) or ('r', degrees)."""
        begin = len(self.xforms)
        for kind, v in chain:
            if kind == "t":
                self.xforms.append((TRANSLATE, [float(x) for x in v]))
            else:
                rad = float(v) * math.pi / 180.0
                self.xforms.append((ROTATE_Y, [math.sin(rad), math.cos(rad), 0.0]))
        return begin, len(chain)

    def instance(self, geom_index, bvh=False, chain=(), medium=None):
        """medium: (density, isotropic material index)."""
        xb, xc = self.xform_chain(chain)
        return (1 if bvh else 0, geom_index, xb, xc, 1 if medium else 0, medium[1] if medium else 0, float(medium[0]) if medium else 0.0)

    def add_world(self, *a, **k):
        self.world.append(self.instance(*a, **k))

    def add_light(self, *a, **k):
        self.lights.append(self.instance(*a, **k))

    def node(self, box6, left, right):
        """children: >= 0 node index (must be created AFTER the parent: pre-order), < 0 = ~prim."""
        self.nodes.append([[float(x) for x in box6], left, right])
        return len(self.nodes) - 1

    # ---- helpers -----------------------------------------------------------------------------
    def prim_box(self, i):
        kind, _, d = self.prims[i]
        if kind == SPHERE:
            r = abs(d[3])
            return [d[0] - r, d[1] - r, d[2] - r, d[0] + r, d[1] + r, d[2] + r]
        if kind == MOVING_SPHERE:  # objects.rs:194-205 for (0, 1)
            r = abs(d[8])
            cs = [[d[k] + (d[3 + k] - d[k]) * ((t - d[6]) / (d[7] - d[6])) for k in range(3)] for t in (0.0, 1.0)]
            return [min(cs[0][k], cs[1][k]) - r for k in range(3)] + [max(cs[0][k], cs[1][k]) + r for k in range(3)]
        if kind == CUBOID:
            return list(d[:6])
        axis = kind - RECT_XY
        ia, ib, ik = (1 if axis == 2 else 0), (1 if axis == 0 else 2), 2 - axis
        mn, mx = [0.0] * 3, [0.0] * 3
        mn[ia], mx[ia], mn[ib], mx[ib], mn[ik], mx[ik] = d[0], d[1], d[2], d[3], d[4] - 0.0001, d[4] + 0.0001
        return mn + mx

    def build_bvh(self, prim_ids, rng):
        """Median-split tree in the reference's shape (bvh.rs:85-111: random axis, sort by box min, leaf = the object);
        nodes are appended in pre-order.  Returns the root node index (needs >= 2 primitives)."""
        assert len(prim_ids) >= 2

        def rec(ids):
            axis = rng.randrange(3)
            ids = sorted(ids, key=lambda i: self.prim_box(i)[axis])
            me = self.node([0] * 6, 0, 0)
            half = len(ids) // 2
            lo, hi = ids[:half], ids[half:]
            kids = []
            for part in (lo, hi):
                if len(part) == 1:
                    kids.append((~part[0], self.prim_box(part[0])))
                else:
                    k = rec(part)
                    kids.append((k, self.nodes[k][0]))
            (l, lb), (r, rb) = kids
            self.nodes[me] = [[min(lb[k], rb[k]) for k in range(3)] + [max(lb[3 + k], rb[3 + k]) for k in range(3)], l, r]
            return me

        return rec(list(prim_ids))

    def camera(self, lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, t1=0.0, t2=1.0):
        """camera.rs:21-56"""
        theta = vfov * math.pi / 180.0
        vh = 2.0 * math.tan(theta / 2.0)
        vw = vh * aspect_ratio
        w = _unit([lookfrom[k] - lookat[k] for k in range(3)])
        u = _unit(_cross(vup, w))
        v = _unit(_cross(w, u))
        hor = [u[k] * vw * focus_dist for k in range(3)]
        ver = [v[k] * vh * focus_dist for k in range(3)]
        llc = [lookfrom[k] - hor[k] / 2.0 - ver[k] / 2.0 - w[k] * focus_dist for k in range(3)]
        self.cam = (list(map(float, lookfrom)), llc, hor, ver, u, v, w, aperture / 2.0, float(t1), float(t2))

    # ---- (de)serialisation: plain lists, e.g. for fixtures ---------------------------------------
    def to_dict(self):
        return {"prims": self.prims, "materials": self.materials, "textures": self.textures, "xforms": self.xforms,
                "world": self.world, "lights": self.lights, "nodes": self.nodes, "background": self.background, "cam": self.cam,
                "perlins": self.perlins, "images": [(w, h, list(bytes(d))) for (w, h, d) in self.images]}

    @classmethod
    def from_dict(cls, d):
        fs = cls()
        fs.prims = [(k, m, list(q)) for k, m, q in d["prims"]]
        fs.materials = [(k, t, tuple(a), f, r) for k, t, a, f, r in d["materials"]]
        fs.textures = [(k, o, e, tb, tuple(c), sc) for k, o, e, tb, c, sc in d["textures"]]
        fs.xforms = [(k, list(v)) for k, v in d["xforms"]]
        fs.world = [tuple(t) for t in d["world"]]
        fs.lights = [tuple(t) for t in d["lights"]]
        fs.nodes = [[list(b), l, r] for b, l, r in d["nodes"]]
        fs.background = list(d["background"])
        fs.cam = tuple(d["cam"])
        fs.perlins = d.get("perlins", [])
        fs.images = [(w, h, bytes(px)) for (w, h, px) in d.get("images", [])]
        return fs

    # ---- ctypes --------------------------------------------------------------------------------
    def desc(self):
        """-> RtSceneDesc (the arrays it points into are kept alive by this object)."""
        F = _ffi
        prims = (F.RtPrim * max(1, len(self.prims)))()
        for i, (k, m, d) in enumerate(self.prims):
            prims[i].kind, prims[i].material = k, m
            for j in range(10):
                prims[i].d[j] = d[j]
        mats = (F.RtMaterial * max(1, len(self.materials)))()
        for i, (k, t, al, fz, ri) in enumerate(self.materials):
            mats[i].kind, mats[i].texture, mats[i].fuzz, mats[i].ref_idx = k, t, fz, ri
            for j in range(3):
                mats[i].albedo[j] = al[j]
        texs = (F.RtTexture * max(1, len(self.textures)))()
        for i, (k, o, e, tb, col, sc) in enumerate(self.textures):
            texs[i].kind, texs[i].odd, texs[i].even, texs[i].table, texs[i].scale = k, o, e, tb, sc
            for j in range(3):
                texs[i].color[j] = col[j]
        xf = (F.RtXform * max(1, len(self.xforms)))()
        for i, (k, v) in enumerate(self.xforms):
            xf[i].kind = k
            for j in range(3):
                xf[i].v[j] = v[j]

        def insts(lst):
            arr = (F.RtInstance * max(1, len(lst)))()
            for i, t in enumerate(lst):
                (arr[i].geom_kind, arr[i].geom_index, arr[i].xform_begin, arr[i].xform_count, arr[i].is_medium,
                 arr[i].phase_material, arr[i].density) = t
            return arr

        world, lights = insts(self.world), insts(self.lights)
        n = len(self.nodes)
        cols = [(C.c_double * max(1, n))(*[nd[0][k] for nd in self.nodes]) for k in range(6)]
        left = (C.c_int32 * max(1, n))(*[nd[1] for nd in self.nodes])
        right = (C.c_int32 * max(1, n))(*[nd[2] for nd in self.nodes])
        perl = (F.RtPerlin * max(1, len(self.perlins)))()
        for i, (px, py, pz, rv) in enumerate(self.perlins):
            for k in range(256):
                perl[i].perm_x[k], perl[i].perm_y[k], perl[i].perm_z[k] = px[k], py[k], pz[k]
                for j in range(3):
                    perl[i].ranvec[k][j] = rv[k][j]
        imgs = (F.RtImage * max(1, len(self.images)))()
        bufs = []
        for i, (w, h, data) in enumerate(self.images):
            buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
            bufs.append(buf)
            imgs[i].width, imgs[i].height, imgs[i].rgb = w, h, C.cast(buf, C.POINTER(C.c_uint8))
        d = F.RtSceneDesc()
        d.abi_version = 1
        d.n_prims, d.prims = len(self.prims), C.cast(prims, C.POINTER(F.RtPrim))
        d.n_materials, d.materials = len(self.materials), C.cast(mats, C.POINTER(F.RtMaterial))
        d.n_textures, d.textures = len(self.textures), C.cast(texs, C.POINTER(F.RtTexture))
        d.n_perlins, d.perlins = len(self.perlins), C.cast(perl, C.POINTER(F.RtPerlin))
        d.n_images, d.images = len(self.images), C.cast(imgs, C.POINTER(F.RtImage))
        d.n_xforms, d.xforms = len(self.xforms), C.cast(xf, C.POINTER(F.RtXform))
        d.n_world, d.world = len(self.world), C.cast(world, C.POINTER(F.RtInstance))
        d.n_lights, d.lights = len(self.lights), C.cast(lights, C.POINTER(F.RtInstance))
        d.bvh.n_nodes = n
        for name, col in zip(("min_x", "min_y", "min_z", "max_x", "max_y", "max_z"), cols):
            setattr(d.bvh, name, C.cast(col, C.POINTER(C.c_double)))
        d.bvh.left, d.bvh.right = C.cast(left, C.POINTER(C.c_int32)), C.cast(right, C.POINTER(C.c_int32))
        for j in range(3):
            d.background[j] = float(self.background[j])
        o, llc, hor, ver, u, v, w, lr, t1, t2 = self.cam
        for j in range(3):
            d.camera.origin[j], d.camera.lower_left_corner[j], d.camera.horizontal[j], d.camera.vertical[j] = o[j], llc[j], hor[j], ver[j]
            d.camera.u[j], d.camera.v[j], d.camera.w[j] = u[j], v[j], w[j]
        d.camera.lens_radius, d.camera.t1, d.camera.t2 = lr, t1, t2
        self._keep = (prims, mats, texs, xf, world, lights, cols, left, right, perl, imgs, bufs, d)
        return d

    @property
    def desc_ptr(self):
        return C.cast(C.pointer(self.desc() if self._keep is None else self._keep[-1]), C.c_void_p)


def scene_from_json(text, aspect_ratio=None, background=(0.0, 0.0, 0.0)):
    """The reference's JSON scene vocabulary (codegen/src/scenes_file.rs:19-152; data in codegen/data/*.json):
    objects: ObjectList{items[]} (nested lists allowed at list level) | Sphere{center,radius,material} | BVHNode{left,right[,bounding_box]};
    materials Lambertian{albedo: texture} | Metal{albedo,fuzz} | Dielectric{ref_idx} | DiffuseLight{emit: texture};
    textures SolidColor{color} | CheckerTexture{odd,even}; camera{lookfrom,lookat,vup,vfov,aspect_ratio,aperture,focus_dist}.
    As in scenes_file.rs: a BVHNode's box is the surrounding box of its children's boxes over (0,1)
    (`BvhNodeStatic::new(left, right, 0.0, 1.0)`, :50-56 — the JSON's own bounding_box field is not used for the scene;
    it is kept beside it as `fs.json_boxes[node] = [min xyz, max xyz]`: the one place where the reference holds numbers
    produced by its own Sphere::bounding_box / Aabb::surrounding_box, which tests compare the recomputed boxes with),
    background = Vec3::zero(), lights = None, shutter 0..1 (:161-176).  The proc-macro bakes this into static
    types (and is disabled as shipped, SWITCH = false); here it becomes a flat scene at run time."""
    j = json.loads(text) if isinstance(text, str) else text
    fs = FlatScene()
    fs.background = list(background)
    fs.json_boxes = {}

    def vec(v):
        return [float(v[0]), float(v[1]), float(v[2])] if isinstance(v, (list, tuple)) else [float(v["x"]), float(v["y"]), float(v["z"])]

    def tex(t):
        if t["type"] == "SolidColor":
            return fs.solid(*vec(t["color"]))
        if t["type"] == "CheckerTexture":
            return fs.texture(CHECKER_3D, odd=tex(t["odd"]), even=tex(t["even"]))
        raise ValueError("texture type " + t["type"])

    def mat(m):
        ty = m["type"]
        if ty == "Lambertian":
            return fs.material(LAMBERTIAN, texture=tex(m["albedo"]))
        if ty == "Metal":
            return fs.material(METAL, albedo=vec(m["albedo"]), fuzz=float(m["fuzz"]))
        if ty == "Dielectric":
            return fs.material(DIELECTRIC, ref_idx=float(m["ref_idx"]))
        if ty == "DiffuseLight":
            return fs.material(DIFFUSE_LIGHT, texture=tex(m["emit"]))
        raise ValueError("material type " + ty)

    def geom(o):
        """-> ('prim', index) | ('node', index), created in pre-order"""
        if o["type"] == "Sphere":
            return ("prim", fs.sphere(vec(o["center"]), float(o["radius"]), mat(o["material"])))
        if o["type"] == "BVHNode":
            me = fs.node([0.0] * 6, 0, 0)
            if "bounding_box" in o:
                fs.json_boxes[me] = vec(o["bounding_box"]["min"]) + vec(o["bounding_box"]["max"])
            kids, boxes = [], []
            for side in ("left", "right"):
                k, i = geom(o[side])
                kids.append(i if k == "node" else ~i)
                boxes.append(fs.nodes[i][0] if k == "node" else fs.prim_box(i))
            (lb, rb) = boxes
            fs.nodes[me] = [[min(lb[k], rb[k]) for k in range(3)] + [max(lb[3 + k], rb[3 + k]) for k in range(3)], kids[0], kids[1]]
            return ("node", me)
        raise ValueError("object type " + o["type"])

    def add(o):
        # An ObjectList inside the world list scans its items in order with the caller's closest-so-far as t_max
        # (src/objects.rs:38-48), which is the scan of the items spliced into the outer list: flattened in place.
        if o["type"] == "ObjectList":
            for x in o["items"]:
                add(x)
            return
        k, i = geom(o)
        fs.add_world(i, bvh=(k == "node"))

    add(j["objects"])
    c = j["camera"]
    fs.camera(vec(c["lookfrom"]), vec(c["lookat"]), vec(c["vup"]), float(c["vfov"]),
              float(aspect_ratio if aspect_ratio is not None else c["aspect_ratio"]), float(c["aperture"]), float(c["focus_dist"]), 0.0, 1.0)
    return fs
