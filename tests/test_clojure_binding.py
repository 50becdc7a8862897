"""The Clojure side of the boundary (clojure/raytracing/{flatten,gpu,core_gpu}.clj), as far as it can be checked
without a JVM (this image has none, so the files have never been loaded by Clojure):

  * they read as Clojure: every delimiter closes, strings and character literals are well formed;
  * every symbol a form uses is a local of that form, a var the namespace defines, a var of a namespace it
    requires (for this repository's own namespaces: one that namespace really defines), a clojure.core var, or
    Java interop on an imported (or java.lang) class — so no helper is left undefined;
  * the numbers they hard-code — struct sizes and field offsets of rt_camera, rt_params and rt_scene_desc, the
    enum values and strides of include/rtmi.h, the scene-file header — equal the C side's (ctypes, abi.py);
  * the record fields they read are the fields the reference's records have (listed here from the defrecord
    forms, with the lines they stand on).
"""
import ctypes as C
import os
import re

import pytest

from raytracing_amd import abi, scenefile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLJ = os.path.join(ROOT, "clojure", "raytracing")
FILES = {"raytracing.flatten": "flatten.clj", "raytracing.gpu": "gpu.clj", "raytracing.core-gpu": "core_gpu.clj"}


# ------------------------------------------------------------------------------- a reader
class Sym(str):
    pass


class Kw(str):
    pass


class Str(str):
    pass


class Node(list):
    def __init__(self, kind, items=()):
        super().__init__(items)
        self.kind = kind  # "(" "[" "{" "#{" "#("


CLOSE = {"(": ")", "[": "]", "{": "}", "#{": "}", "#(": ")"}
TOKEN_END = set(" \t\r\n,()[]{}\";")


def read_all(text):
    pos, n = 0, len(text)

    def skip():
        nonlocal pos
        while pos < n:
            c = text[pos]
            if c in " \t\r\n,":
                pos += 1
            elif c == ";":
                while pos < n and text[pos] != "\n":
                    pos += 1
            else:
                return

    def read():
        nonlocal pos
        skip()
        if pos >= n:
            raise EOFError
        c = text[pos]
        if c in ")]}":
            raise SyntaxError(f"unexpected {c!r} at offset {pos}")
        for opener in ("#{", "#(", "(", "[", "{"):
            if text.startswith(opener, pos):
                pos += len(opener)
                node = Node(opener)
                while True:
                    skip()
                    if pos >= n:
                        raise SyntaxError(f"{opener!r} never closed")
                    if text[pos] in ")]}":
                        if text[pos] != CLOSE[opener]:
                            raise SyntaxError(f"{opener!r} closed by {text[pos]!r} at offset {pos}")
                        pos += 1
                        return node
                    node.append(read())
        if c == '"' or text.startswith('#"', pos):
            pos += 1 if c == '"' else 2
            start = pos
            while pos < n and text[pos] != '"':
                pos += 2 if text[pos] == "\\" else 1
            if pos >= n:
                raise SyntaxError("string never closed")
            pos += 1
            return Str(text[start:pos - 1])
        if c == "\\":
            m = re.compile(r"\\(newline|space|tab|return|u[0-9a-fA-F]{4}|.)", re.S).match(text, pos)
            pos = m.end()
            return Str(m.group(1))
        if c == "'" or c == "`" or c == "@" or c == "~":
            pos += 1
            return Node("(", [Sym("quote" if c == "'" else "deref"), read()])
        if c == "^":
            pos += 1
            read()  # the metadata: a type hint, a keyword or a map
            return read()
        if text.startswith("#_", pos):
            pos += 2
            read()
            return read()
        if text.startswith("#'", pos):
            pos += 2
            return read()
        start = pos
        while pos < n and text[pos] not in TOKEN_END:
            pos += 1
        tok = text[start:pos]
        assert tok, f"empty token at offset {start}"
        if tok.startswith(":"):
            return Kw(tok)
        if re.fullmatch(r"[-+]?(0[xX][0-9a-fA-F]+|\d+(\.\d*)?([eE][-+]?\d+)?[MN]?)", tok):
            return int(tok, 16) if tok.lower().startswith(("0x", "-0x", "+0x")) else float(tok) if re.search(r"[.eE]", tok) else int(tok.rstrip("MN"))
        return Sym(tok)

    forms = []
    while True:
        try:
            forms.append(read())
        except EOFError:
            return forms


# --------------------------------------------------------------------- what the names may be
CORE = set("""
* + - / < <= = > >= aget alength and apply aset aset-byte assoc bit-and bit-not bit-or byte-array concat cond condp conj
cons count dec def defn defn- do doseq dotimes double double-array doto ex-info first flatten fn for get if if-let
if-not inc instance? int int-array into into-array let long loop make-array map map-indexed mapv max min neg? new nil? not
ns or pos? quot quote range recur reduce repeat rest second seq some str throw try catch finally unchecked-byte vec vector
when when-let when-not with-open zero? true false nil deref proxy keys vals merge not= mod rem identity partial comp
""".split())
BINDERS = {"let", "loop", "doseq", "for", "with-open", "when-let", "if-let", "dotimes"}
JAVA_LANG = {"System", "Math", "String", "Class", "Integer", "Long", "Double", "Thread", "Object", "Runnable"}
# the namespaces of the reference that the files require, with the vars they take from them (file:line)
REFERENCE_VARS = {
    "raytracing.camera": {"camera"},                 # camera.clj:7
    "raytracing.core": {"display-image"},            # core.clj:32
    "raytracing.scenes": {"random-scene", "two-spheres", "two-perlin-spheres", "earth", "simple-light", "cornell-box",
                          "cornell-smoke", "final-scene"},   # scenes.clj:51,28,33,38,44,71,89,109
}
REFERENCE_NAMESPACES = {"raytracing.base", "raytracing.rectangles", "raytracing.box", "raytracing.rotate", "raytracing.translate",
                        "raytracing.constant-medium", "raytracing.materials", "raytracing.texture", "raytracing.noise",
                        "raytracing.camera", "raytracing.core", "raytracing.scenes"}
# the record classes they import, with the fields they may read (the defrecord forms of the reference)
RECORD_FIELDS = {
    "Sphere": {"center", "radius", "mat-ptr"},                                    # base.clj:229
    "MovingSphere": {"center0", "center1", "time0", "time1", "radius", "mat-ptr"},   # base.clj:263-267
    "HittableList": {"objects"},                                                  # base.clj:90
    "BvhNode": {"left", "right", "box"},                                          # base.clj:116
    "Aabb": {"minimum", "maximum"},                                               # base.clj:59
    "XYRect": {"mp", "x0", "x1", "y0", "y1", "k"},                                # rectangles.clj:8-13
    "XZRect": {"mp", "x0", "x1", "z0", "z1", "k"},                                # rectangles.clj:45-50
    "YZRect": {"mp", "y0", "y1", "z0", "z1", "k"},                                # rectangles.clj:82-87
    "Box": {"box-min", "box-max", "sides"},                                       # box.clj:8
    "RotateY": {"ptr", "sin-theta", "cos-theta", "hasbox", "bbox"},               # rotate.clj:9
    "Translate": {"ptr", "offset"},                                               # translate.clj:8
    "ConstantMedium": {"boundary", "phase-function", "neg-inv-density"},          # constant_medium.clj:9
    "Lambertian": {"albedo"}, "Metal": {"albedo", "fuzz"}, "Dielectric": {"ir"},   # materials.clj:14,31,45
    "DiffuseLight": {"emit"}, "Isotropic": {"albedo"},                            # materials.clj:67,85
    "SolidColor": {"color-value"}, "CheckerTexture": {"odd", "even"},              # texture.clj:15,23
    "NoiseTexture": {"perlin-noise", "scale"},                                    # texture.clj:38-39
    "ImageTexture": {"data", "width", "height", "bytes-per-scanline"},            # texture.clj:50-53
    "Perlin": {"point-count", "ranvec", "perm-x", "perm-y", "perm-z"},            # noise.clj:35-37
}


def parse_ns(form):
    assert form[0] == "ns"
    info = {"name": str(form[1]), "aliases": {}, "refers": {}, "classes": set(), "requires": set()}
    for clause in form[2:]:
        if isinstance(clause, Node) and clause and clause[0] == ":require":
            for spec in clause[1:]:
                ns = str(spec[0])
                info["requires"].add(ns)
                opts = dict(zip(spec[1::2], spec[2::2]))
                if ":as" in opts:
                    info["aliases"][str(opts[":as"])] = ns
                for name in opts.get(":refer", []):
                    info["refers"][str(name)] = ns
        if isinstance(clause, Node) and clause and clause[0] == ":import":
            for spec in clause[1:]:
                for cls in spec[1:]:
                    info["classes"].add(str(cls))
    return info


def defined_vars(forms):
    out = set()
    for f in forms:
        if isinstance(f, Node) and f.kind == "(" and f and f[0] in ("def", "defn", "defn-", "defmacro", "declare"):
            out.update(str(x) for x in (f[1:] if f[0] == "declare" else f[1:2]))
    return out


def binding_names(b, out):
    """Every symbol a binding form introduces (vectors and maps destructure)."""
    if isinstance(b, Sym):
        if b != "&":
            out.add(str(b))
    elif isinstance(b, Node) and b.kind == "[":
        for x in b:
            binding_names(x, out)
    elif isinstance(b, Node) and b.kind == "{":
        items = list(b)
        for k, v in zip(items[0::2], items[1::2]):
            if k == ":keys":
                out.update(str(s) for s in v)
            elif k == ":as":
                out.add(str(v))
            elif k == ":or":
                pass
            else:
                binding_names(k, out)


def check_symbols(ns_name, forms, all_defs):
    info = parse_ns(forms[0])
    own = all_defs[ns_name]
    problems, keywords = [], set()

    def resolve(sym, local):
        s = str(sym)
        if s in local or s in own or s in CORE or s in ("%", "%1", "%2", "%3", "%&", "&"):
            return
        if s in info["refers"]:
            target = info["refers"][s]
            if s not in REFERENCE_VARS.get(target, all_defs.get(target, set())):
                problems.append(f"{ns_name}: {s} is referred from {target}, which does not define it")
            return
        if s.startswith(".") or s.endswith(".") or s.startswith("->"):
            base = s.strip(".")
            if s.endswith(".") and base.split("$")[0].split(".")[-1] not in info["classes"] | JAVA_LANG and "." not in base:
                problems.append(f"{ns_name}: constructor of a class that is not imported: {s}")
            return
        if "/" in s and s != "/":
            head, name = s.split("/", 1)
            if head in info["aliases"]:
                target = info["aliases"][head]
                known = all_defs.get(target) if target in all_defs else REFERENCE_VARS.get(target)
                if known is None or name not in known:
                    problems.append(f"{ns_name}: {s} - {target} has no var {name}")
                return
            if head in info["classes"] | JAVA_LANG:
                return
            problems.append(f"{ns_name}: {s} names neither an alias nor an imported class")
            return
        if s in info["classes"] | JAVA_LANG or re.fullmatch(r"(java|javax)\.[\w.$]+", s):
            return
        problems.append(f"{ns_name}: symbol {s} is not defined, referred, imported or a clojure.core var")

    def walk(x, local):
        if isinstance(x, Kw):
            keywords.add(str(x))
        elif isinstance(x, Sym):
            resolve(x, local)
        elif isinstance(x, Node):
            if x.kind == "#(":
                return walk(Node("(", x), local)
            if x.kind != "(" or not x:
                for y in x:
                    walk(y, local)
                return
            head = x[0]
            if head in ("quote",):
                return
            if head == "ns":
                return
            if head in ("def",):
                for y in x[2:]:
                    walk(y, local)
                return
            if head in ("defn", "defn-", "fn"):
                rest = list(x[1:])
                inner = set(local)
                if rest and isinstance(rest[0], Sym):
                    inner.add(str(rest.pop(0)))
                while rest and isinstance(rest[0], (Str,)) or (rest and isinstance(rest[0], Node) and rest[0].kind == "{" and head != "fn"):
                    rest.pop(0)
                arities = rest if rest and isinstance(rest[0], Node) and rest[0].kind == "(" else [Node("(", rest)]
                for ar in arities:
                    scope = set(inner)
                    binding_names(ar[0], scope)
                    for kdef in [b for b in ([ar[0]] if True else [])]:
                        walk_defaults(kdef, scope)
                    for y in ar[1:]:
                        walk(y, scope)
                return
            if head in BINDERS:
                scope = set(local)
                bv = x[1]
                items = list(bv)
                i = 0
                while i < len(items):
                    if isinstance(items[i], Kw):   # :let [..] :when test inside doseq / for
                        if items[i] == ":let":
                            sub = list(items[i + 1])
                            for k in range(0, len(sub), 2):
                                walk(sub[k + 1], scope)
                                binding_names(sub[k], scope)
                        else:
                            walk(items[i + 1], scope)
                        i += 2
                        continue
                    walk(items[i + 1], scope)
                    binding_names(items[i], scope)
                    walk_defaults(items[i], scope)
                    i += 2
                for y in x[2:]:
                    walk(y, scope)
                return
            if head == "catch":
                scope = set(local) | {str(x[2])}
                for y in x[3:]:
                    walk(y, scope)
                return
            if head == "proxy":
                for y in x[2:3]:
                    walk(y, local)
                for method in x[3:]:
                    scope = set(local) | {"this"}
                    binding_names(method[1], scope)
                    for y in method[2:]:
                        walk(y, scope)
                return
            if head == "condp":
                for y in x[1:]:
                    walk(y, local)
                return
            if head == "new" or head == ".":
                for y in x[2:]:
                    walk(y, local)
                return
            for y in x:
                walk(y, local)

    def walk_defaults(b, scope):
        """The default expressions of an :or map are evaluated too."""
        if isinstance(b, Node) and b.kind == "{":
            items = list(b)
            for k, v in zip(items[0::2], items[1::2]):
                if k == ":or":
                    for dv in list(v)[1::2]:
                        walk(dv, scope)
        elif isinstance(b, Node) and b.kind == "[":
            for y in b:
                walk_defaults(y, scope)

    for f in forms[1:]:
        walk(f, set())
    return problems, keywords, info


@pytest.fixture(scope="module")
def sources():
    out = {}
    for ns, fname in FILES.items():
        text = open(os.path.join(CLJ, fname)).read()
        out[ns] = (text, read_all(text))
    return out


def test_files_read_as_clojure(sources):
    for ns, (text, forms) in sources.items():
        assert forms and forms[0][0] == "ns" and str(forms[0][1]) == ns, ns
        assert len(forms) > 3
    with pytest.raises(SyntaxError):
        read_all("(defn f [x] (+ x 1)")        # the reader does notice an open form
    with pytest.raises(SyntaxError):
        read_all("(let [a 1) a]")


def test_every_symbol_is_defined_or_imported(sources):
    all_defs = {ns: defined_vars(forms) for ns, (_, forms) in sources.items()}
    problems = []
    for ns, (_, forms) in sources.items():
        p, _, info = check_symbols(ns, forms, all_defs)
        problems += p
        for req in info["requires"]:
            assert req in all_defs or req in REFERENCE_NAMESPACES, f"{ns} requires {req}, which neither side has"
    assert not problems, "\n".join(problems)
    # the helpers the binding is made of, by name
    need = {"raytracing.flatten": {"new-state", "prim-node!", "list-node!", "bvh-node!", "wrap-node!", "medium-node!", "mat!", "tex!",
                                   "perlin!", "image!", "freeze", "node!", "flatten-world", "scene-bytes"},
            "raytracing.gpu": {"write-desc!", "write-camera!", "write-params!", "check!", "render-gpu", "render-multi-gpu", "decode-jpeg"},
            "raytracing.core-gpu": {"-main", "render-into!", "worlds"}}
    for ns, names in need.items():
        assert names <= all_defs[ns], f"{ns} lacks {sorted(names - all_defs[ns])}"
    # and the checker itself does find a hole
    broken = read_all("(ns x.y) (defn f [a] (undefined-helper a))")
    assert check_symbols("x.y", broken, {"x.y": defined_vars(broken)})[0]


def test_record_fields_are_the_references(sources):
    _, forms = sources["raytracing.flatten"]
    _, keywords, _ = check_symbols("raytracing.flatten", forms, {"raytracing.flatten": defined_vars(forms)})
    fields = set().union(*RECORD_FIELDS.values())
    own_keys = {":" + k for k in ("prim-type prim-d prim-mat node-kind node-i node-d children mat-type mat-d mat-tex tex-type tex-d tex-i "
                                  "ranvec perm img-w img-h images n-media node-of mat-of tex-of perlin-of image-of abi-version root "
                                  "sphere moving-sphere xy-rect xz-rect yz-rect prim list bvh translate rotate-y medium lambertian metal "
                                  "dielectric diffuse-light isotropic solid checker noise image value values stride require import").split()}
    unknown = {k for k in keywords if k not in own_keys and k[1:] not in fields}
    assert not unknown, f"keywords that are neither the flattener's own nor fields of the reference's records: {sorted(unknown)}"
    for must in ("center", "radius", "mat-ptr", "mp", "k", "sides", "objects", "left", "right", "box", "ptr", "offset", "sin-theta",
                 "cos-theta", "boundary", "phase-function", "neg-inv-density", "albedo", "fuzz", "ir", "emit", "color-value", "odd",
                 "even", "perlin-noise", "scale", "data", "ranvec", "perm-x", "perm-y", "perm-z", "minimum", "maximum"):
        assert ":" + must in keywords, f"the flattener never reads {must}"


def _map_of(forms, name):
    for f in forms:
        if isinstance(f, Node) and f and f[0] == "def" and f[1] == name:
            body = f[-1]
            if isinstance(body, Node) and body.kind == "{":
                items = list(body)
                return {str(k)[1:].replace("-", "_"): v for k, v in zip(items[0::2], items[1::2])}
            return body
    raise KeyError(name)


def test_hard_coded_layouts_equal_the_c_structs(sources):
    _, g = sources["raytracing.gpu"]
    assert _map_of(g, "camera-bytes") == C.sizeof(abi.rt_camera) == 192
    assert _map_of(g, "params-bytes") == C.sizeof(abi.rt_params) == 64
    assert _map_of(g, "desc-bytes") == C.sizeof(abi.rt_scene_desc) == 208
    cam = _map_of(g, "camera-offsets")
    assert cam == {n: getattr(abi.rt_camera, n).offset for n, _ in abi.rt_camera._fields_}
    prm = _map_of(g, "params-offsets")
    assert prm == {n: getattr(abi.rt_params, n).offset for n, _ in abi.rt_params._fields_}
    desc = _map_of(g, "desc-offsets")
    rename = {"node_children": "node_children", "img_rgb8": "img_rgb8"}
    assert desc == {rename.get(n, n): getattr(abi.rt_scene_desc, n).offset for n, _ in abi.rt_scene_desc._fields_}
    assert _map_of(g, "split-tiles") == abi.RT_SPLIT_TILES and _map_of(g, "split-samples") == abi.RT_SPLIT_SAMPLES
    assert _map_of(g, "default-seed") == 0x5EED0001
    _, f = sources["raytracing.flatten"]
    assert _map_of(f, "abi-version") == abi.RTMI_ABI_VERSION
    for clj, py in (("prim-stride", abi.RT_PRIM_STRIDE), ("node-istride", abi.RT_NODE_ISTRIDE), ("node-dstride", abi.RT_NODE_DSTRIDE),
                    ("mat-dstride", abi.RT_MAT_DSTRIDE), ("tex-dstride", abi.RT_TEX_DSTRIDE), ("tex-istride", abi.RT_TEX_ISTRIDE),
                    ("scene-header-bytes", scenefile.HEADER_BYTES)):
        assert _map_of(f, clj) == py, clj
    assert _map_of(f, "scene-magic").encode() == scenefile.MAGIC
    assert _map_of(f, "prim-kinds") == {"sphere": abi.RT_PRIM_SPHERE, "moving_sphere": abi.RT_PRIM_MOVING_SPHERE, "xy_rect": abi.RT_PRIM_XY_RECT,
                                        "xz_rect": abi.RT_PRIM_XZ_RECT, "yz_rect": abi.RT_PRIM_YZ_RECT}
    assert _map_of(f, "node-kinds") == {"prim": abi.RT_NODE_PRIM, "list": abi.RT_NODE_LIST, "bvh": abi.RT_NODE_BVH,
                                        "translate": abi.RT_NODE_TRANSLATE, "rotate_y": abi.RT_NODE_ROTATE_Y, "medium": abi.RT_NODE_MEDIUM}
    assert _map_of(f, "mat-kinds") == {"lambertian": abi.RT_MAT_LAMBERTIAN, "metal": abi.RT_MAT_METAL, "dielectric": abi.RT_MAT_DIELECTRIC,
                                       "diffuse_light": abi.RT_MAT_DIFFUSE_LIGHT, "isotropic": abi.RT_MAT_ISOTROPIC}
    assert _map_of(f, "tex-kinds") == {"solid": abi.RT_TEX_SOLID, "checker": abi.RT_TEX_CHECKER, "noise": abi.RT_TEX_NOISE, "image": abi.RT_TEX_IMAGE}


def test_downcall_signatures_match_the_header(sources):
    """Every (down "rt_..." (fd ...)) names a function include/rtmi.h declares, with as many arguments."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtmi.h")).read(), flags=re.S)
    protos = {m.group(2): (m.group(1).strip(), [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"])
              for m in re.finditer(r"^([\w \*]+?)\b(rt_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S)}
    _, g = sources["raytracing.gpu"]
    seen = 0
    for f in g:
        if isinstance(f, Node) and f and f[0] == "def" and isinstance(f[-1], Node) and f[-1] and f[-1][0] == "down":
            name, fdesc = str(f[-1][1]), f[-1][2]
            assert name in protos, f"{name} is not declared in include/rtmi.h"
            ret, args = protos[name]
            layouts = [str(a) for a in fdesc[1:]]
            if fdesc[0] == "fd":
                ret_layout, layouts = layouts[0], layouts[1:]
                assert (ret_layout == "ADDR") == ("*" in ret), name
                assert ret_layout == "I32" or "*" in ret, name
            else:
                assert ret == "void", name
            assert len(layouts) == len(args), f"{name}: {len(layouts)} layouts for {len(args)} parameters"
            for lay, arg in zip(layouts, args):
                want = "ADDR" if "*" in arg else "I64" if re.search(r"\b(u?int64_t|size_t)\b", arg) else "I32" if re.search(r"\bu?int32_t\b|\bint\b", arg) else "?"
                assert lay == want, f"{name}: parameter `{arg}` bound as {lay}"
            seen += 1
    assert seen >= 12
