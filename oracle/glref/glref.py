"""ctypes front-end of oracle/_ref/libglref.so — container-only TEST INFRASTRUCTURE.

Runs the reference's unmodified GLSL (loaded by path from /root/reference/shaders) on Mesa
llvmpipe; used only by tests/golden/make_golden.py (golden-vector generation) and by the
llvmpipe timing script. Not importable on the GPU box (no /root/reference there) and never
imported by the product package.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "_ref", "libglref.so")
REF_SHADERS = os.environ.get("GPUART_REF_SHADERS", "/root/reference/shaders")

GL_FRAGMENT_SHADER = 0x8B30
GL_VERTEX_SHADER = 0x8B31
GL_RENDERER = 0x1F01
GL_VERSION = 0x1F02


def available():
    return os.path.exists(LIB) and os.path.isdir(REF_SHADERS)


class GLRef:
    def __init__(self):
        L = C.CDLL(LIB)
        self.L = L
        L.glref_last_error.restype = C.c_char_p
        L.glref_string.restype = C.c_char_p
        L.glref_string.argtypes = [C.c_uint]
        L.glref_shader_src.argtypes = [C.c_uint, C.c_char_p]
        L.glref_shader_file.argtypes = [C.c_uint, C.c_char_p]
        L.glref_program.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.glref_tex2d_rgba32f.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.glref_tbo_rgba32f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.glref_fbo.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.glref_clear_fbo.argtypes = [C.c_int] + [C.c_float] * 4
        L.glref_uniform_loc.argtypes = [C.c_int, C.c_char_p]
        L.glref_uniform1i.argtypes = [C.c_int, C.c_int]
        L.glref_uniform1ui.argtypes = [C.c_int, C.c_uint]
        L.glref_uniform1f.argtypes = [C.c_int, C.c_float]
        L.glref_uniform3f.argtypes = [C.c_int] + [C.c_float] * 3
        L.glref_uniform4f.argtypes = [C.c_int] + [C.c_float] * 4
        L.glref_read_tex.argtypes = [C.c_int, C.c_void_p]
        if L.glref_init() != 0:
            raise RuntimeError("glref_init: " + self.err())
        self._file_shaders = {}

    def err(self):
        return self.L.glref_last_error().decode()

    def renderer(self):
        return self.L.glref_string(GL_RENDERER).decode()

    # -- shaders ---------------------------------------------------------------------------
    def shader_src(self, typ, src):
        s = self.L.glref_shader_src(typ, src.encode())
        if s < 0:
            raise RuntimeError(self.err())
        return s

    def ref_shader(self, name, shader_dir=None, edit=None):
        """Compiles reference shader `name` (e.g. 'sphere.glsl') from its path.

        `edit` = optional (old, new) textual substitution applied in memory to ONE constant
        (used only for MAX_PATH_SEGMENTS, which the reference hard-codes as a shader const).
        """
        d = shader_dir or REF_SHADERS
        key = (d, name, edit)
        if key in self._file_shaders:
            return self._file_shaders[key]
        typ = GL_VERTEX_SHADER if name == "vertex.glsl" else GL_FRAGMENT_SHADER
        path = os.path.join(d, name)
        if edit is None:
            s = self.L.glref_shader_file(typ, path.encode())
            if s < 0:
                raise RuntimeError(name + ": " + self.err())
        else:
            src = open(path).read()
            assert src.count(edit[0]) == 1, (name, edit)
            s = self.shader_src(typ, src.replace(edit[0], edit[1]))
        self._file_shaders[key] = s
        return s

    def program(self, shaders):
        arr = (C.c_int * len(shaders))(*shaders)
        p = self.L.glref_program(arr, len(shaders))
        if p < 0:
            raise RuntimeError(self.err())
        return p

    # -- resources -------------------------------------------------------------------------
    def tex(self, w, h, data=None):
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float32)
            assert data.size == w * h * 4
            return self.L.glref_tex2d_rgba32f(w, h, data.ctypes.data)
        return self.L.glref_tex2d_rgba32f(w, h, None)

    def tbo(self, quads):
        quads = np.ascontiguousarray(quads, dtype=np.float32)
        buf = C.c_int(0)
        t = self.L.glref_tbo_rgba32f(quads.ctypes.data, quads.size // 4, C.byref(buf))
        return t, buf.value

    def fbo(self, texs):
        arr = (C.c_int * len(texs))(*texs)
        f = self.L.glref_fbo(arr, len(texs))
        if f < 0:
            raise RuntimeError(self.err())
        return f

    def read(self, tex, w, h):
        out = np.empty((h, w, 4), dtype=np.float32)
        self.L.glref_read_tex(tex, out.ctypes.data)
        return out

    # -- uniforms / draw -------------------------------------------------------------------
    def set_uniforms(self, prog, **u):
        """Values: int -> 1i, ('u', int) -> 1ui, float -> 1f, 3/4-sequence -> 3f/4f."""
        self.L.glref_use(prog)
        for name, v in u.items():
            loc = self.L.glref_uniform_loc(prog, name.encode())
            if loc < 0:
                raise RuntimeError("uniform %s not active" % name)
            if isinstance(v, tuple) and len(v) == 2 and v[0] == "u":
                self.L.glref_uniform1ui(loc, int(v[1]))
            elif isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                self.L.glref_uniform1i(loc, int(v))
            elif isinstance(v, (float, np.floating)):
                self.L.glref_uniform1f(loc, float(v))
            else:
                v = [float(x) for x in v]
                if len(v) == 3:
                    self.L.glref_uniform3f(loc, *v)
                elif len(v) == 4:
                    self.L.glref_uniform4f(loc, *v)
                else:
                    raise ValueError(name)

    def bind(self, unit, tex, buffer_tex=False):
        self.L.glref_bind_tex(unit, 1 if buffer_tex else 0, tex)

    def draw(self, prog, fbo, w, h):
        self.L.glref_use(prog)
        if self.L.glref_draw_quad(fbo, w, h) != 0:
            raise RuntimeError(self.err())

    def finish(self):
        self.L.glref_finish()


# ---------------------------------------------------------------------------------------------
# Per-function probes: a tiny fragment main() of OUR OWN, linked against unmodified reference
# shader objects (the reference links separately compiled objects the same way,
# src/renderer.cpp:259-359).  Inputs come from N x 1 RGBA32F textures In0..In{k-1}, outputs go
# to up to 8 RGBA32F render targets.
# ---------------------------------------------------------------------------------------------
PROBE_HEAD = """#version 330 core
in vec2 UV;
uniform sampler2D In0; uniform sampler2D In1; uniform sampler2D In2; uniform sampler2D In3;
uniform sampler2D In4; uniform sampler2D In5;
uniform samplerBuffer BVH;
layout(location=0) out vec4 O0; layout(location=1) out vec4 O1; layout(location=2) out vec4 O2;
layout(location=3) out vec4 O3; layout(location=4) out vec4 O4; layout(location=5) out vec4 O5;
"""


def run_probe(gl, body, ref_objs, inputs, nout, bvh=None, decls=""):
    """Runs `body` (GLSL statements of main) over n samples.

    inputs: list of (n,4) float32 arrays, visible as `vec4 i0, i1, ...`;
    returns list of `nout` (n,4) float32 arrays from O0..O{nout-1}.
    """
    n = inputs[0].shape[0]
    # one row of n pixels; keep n <= 16384 (max texture size)
    W = n
    src = PROBE_HEAD + decls + "\nvoid main(){\n"
    for k in range(len(inputs)):
        src += "  vec4 i%d = texelFetch(In%d, ivec2(int(gl_FragCoord.x), 0), 0);\n" % (k, k)
    for k in range(6):
        src += "  O%d = vec4(0);\n" % k
    src += body + "\n}\n"
    fs = gl.shader_src(GL_FRAGMENT_SHADER, src)
    objs = [gl.ref_shader("vertex.glsl"), fs] + [gl.ref_shader(o) for o in ref_objs]
    prog = gl.program(objs)
    texs_in = [gl.tex(W, 1, a) for a in inputs]
    texs_out = [gl.tex(W, 1) for _ in range(nout)]
    fbo = gl.fbo(texs_out)
    gl.L.glref_use(prog)
    unit = 0
    for k, t in enumerate(texs_in):
        loc = gl.L.glref_uniform_loc(prog, ("In%d" % k).encode())
        if loc >= 0:
            gl.bind(unit, t)
            gl.L.glref_uniform1i(loc, unit)
            unit += 1
    tb = None
    if bvh is not None:
        tb = gl.tbo(bvh)
        loc = gl.L.glref_uniform_loc(prog, b"BVH")
        if loc >= 0:
            gl.bind(unit, tb[0], buffer_tex=True)
            gl.L.glref_uniform1i(loc, unit)
    gl.draw(prog, fbo, W, 1)
    gl.finish()
    outs = [gl.read(t, W, 1).reshape(n, 4).copy() for t in texs_out]
    for t in texs_in + texs_out:
        gl.L.glref_delete_tex(t)
    gl.L.glref_delete_fbo(fbo)
    if tb:
        gl.L.glref_delete_tex(tb[0])
        gl.L.glref_delete_buffer(tb[1])
    gl.L.glref_delete_program(prog)
    gl.L.glref_delete_shader(fs)
    return outs


def run_probe_big(gl, body, ref_objs, inputs, nout, bvh=None, decls="", chunk=8192):
    n = inputs[0].shape[0]
    outs = [[] for _ in range(nout)]
    for s in range(0, n, chunk):
        o = run_probe(gl, body, ref_objs, [a[s:s + chunk] for a in inputs], nout, bvh, decls)
        for k in range(nout):
            outs[k].append(o[k])
    return [np.concatenate(x) for x in outs]
