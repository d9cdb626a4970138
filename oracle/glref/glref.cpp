// glref — container-only TEST INFRASTRUCTURE (never shipped, never on the product path).
//
// Runs the reference's *unmodified* GLSL shaders (read at run time, by path, from
// /root/reference/shaders — nothing is copied) on the system's Mesa llvmpipe software
// rasteriser, without an X server, so that golden vectors for the hot path can be
// produced by "the reference itself, run here" (task §③).
//
// Everything here is our own code:
//   * an X-less OpenGL 3.3 core context made by driving Mesa's DRI_SWRast driver
//     extension directly (swrast_dri.so + GL/internal/dri_interface.h, both system files);
//   * GL entry points resolved through libglapi's _glapi_get_proc_address;
//   * a tiny C ABI (ctypes-friendly) to compile shader objects from a file path or a source
//     string, link programs, feed RGBA32F textures / buffer textures / uniforms, draw a
//     full-screen quad into an RGBA32F FBO and read the result back.
//
// The reference's C++ host files are NOT built here: they include <nanogui/nanogui.h>, an
// un-vendored submodule that the image lacks, and writing a stand-in header for it is not
// allowed — see DESIGN.md "Oracle". Only the GLSL (which needs nothing but a GL 3.3 driver,
// which the image has) is executed.
//
// Build: see oracle/Makefile (output: oracle/_ref/libglref.so, git-ignored).

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

std::string g_err;
void set_err(const std::string &s) { g_err = s; }

// ---- DRI swrast loader callbacks -------------------------------------------------------
int g_draw_w = 16, g_draw_h = 16;

void cb_getDrawableInfo(__DRIdrawable *, int *x, int *y, int *w, int *h, void *) {
    *x = 0; *y = 0; *w = g_draw_w; *h = g_draw_h;
}
void cb_putImage(__DRIdrawable *, int, int, int, int, int, char *, void *) {}
void cb_getImage(__DRIdrawable *, int, int, int w, int h, char *data, void *) {
    memset(data, 0, (size_t)w * h * 4);
}
void cb_putImage2(__DRIdrawable *, int, int, int, int, int, int, char *, void *) {}
void cb_getImage2(__DRIdrawable *, int, int, int, int h, int stride, char *data, void *) {
    memset(data, 0, (size_t)stride * h);
}

__DRIswrastLoaderExtension g_loader_ext;
const __DRIextension *g_loader_exts[2];

const __DRIcoreExtension *g_core = nullptr;
const __DRIswrastExtension *g_swrast = nullptr;
__DRIscreen *g_screen = nullptr;
__DRIcontext *g_ctx = nullptr;
__DRIdrawable *g_draw = nullptr;

typedef void (*proc_t)(void);
proc_t (*glapi_get_proc)(const char *) = nullptr;

// ---- GL entry points ------------------------------------------------------------------
#define GLFN(ret, name, ...) ret (*p_##name)(__VA_ARGS__) = nullptr;
#include "glfuncs.inc"
#undef GLFN

bool load_gl() {
    bool ok = true;
#define GLFN(ret, name, ...)                                                         \
    p_##name = (ret(*)(__VA_ARGS__))glapi_get_proc(#name);                           \
    if (!p_##name) { set_err(std::string("missing GL entry point ") + #name); ok = false; }
#include "glfuncs.inc"
#undef GLFN
    return ok;
}

GLuint g_quad_vao = 0, g_quad_vbo = 0, g_quad_ebo = 0;

}  // namespace

extern "C" {

const char *glref_last_error() { return g_err.c_str(); }

/// Creates the X-less llvmpipe context. Returns 0 on success.
int glref_init(void) {
    if (g_ctx) return 0;
    const char *drv = getenv("GLREF_SWRAST_DRI");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *hapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!hapi) { set_err(std::string("dlopen libglapi.so.0: ") + dlerror()); return -1; }
    glapi_get_proc = (proc_t(*)(const char *))dlsym(hapi, "_glapi_get_proc_address");
    if (!glapi_get_proc) { set_err("no _glapi_get_proc_address"); return -1; }

    void *hdrv = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!hdrv) { set_err(std::string("dlopen swrast_dri: ") + dlerror()); return -1; }
    auto get_exts = (const __DRIextension **(*)(void))dlsym(hdrv, "__driDriverGetExtensions_swrast");
    if (!get_exts) { set_err("no __driDriverGetExtensions_swrast"); return -1; }
    const __DRIextension **drv_exts = get_exts();
    for (int i = 0; drv_exts[i]; i++) {
        if (!strcmp(drv_exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)drv_exts[i];
        if (!strcmp(drv_exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)drv_exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) { set_err("DRI_Core/DRI_SWRast(v4) not found"); return -1; }

    memset(&g_loader_ext, 0, sizeof g_loader_ext);
    g_loader_ext.base.name = __DRI_SWRAST_LOADER;
    g_loader_ext.base.version = 3;
    g_loader_ext.getDrawableInfo = cb_getDrawableInfo;
    g_loader_ext.putImage = cb_putImage;
    g_loader_ext.getImage = cb_getImage;
    g_loader_ext.putImage2 = cb_putImage2;
    g_loader_ext.getImage2 = cb_getImage2;
    g_loader_exts[0] = &g_loader_ext.base;
    g_loader_exts[1] = nullptr;

    const __DRIconfig **configs = nullptr;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, drv_exts, &configs, nullptr);
    if (!g_screen || !configs || !configs[0]) { set_err("createNewScreen2 failed"); return -1; }

    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], nullptr, 2, attribs, &err, nullptr);
    if (!g_ctx) { set_err("createContextAttribs failed, err=" + std::to_string(err)); return -1; }
    g_draw = g_swrast->createNewDrawable(g_screen, configs[0], nullptr);
    if (!g_draw) { set_err("createNewDrawable failed"); return -1; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { set_err("bindContext failed"); return -1; }

    if (!load_gl()) return -1;

    // Full-screen quad, same geometry as the reference's (4 corners of [-1,1]^2 as a triangle
    // fan: src/gl_utils.cpp:133-155,225-254 — restated from its description, not copied).
    const GLfloat verts[] = {-1, -1, 1, -1, 1, 1, -1, 1};
    const GLuint idx[] = {0, 1, 2, 3};
    p_glGenVertexArrays(1, &g_quad_vao);
    p_glBindVertexArray(g_quad_vao);
    p_glGenBuffers(1, &g_quad_vbo);
    p_glBindBuffer(GL_ARRAY_BUFFER, g_quad_vbo);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof verts, verts, GL_STATIC_DRAW);
    p_glGenBuffers(1, &g_quad_ebo);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, g_quad_ebo);
    p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof idx, idx, GL_STATIC_DRAW);
    p_glDisable(GL_BLEND);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_STENCIL_TEST);
    p_glDisable(GL_CULL_FACE);
    return 0;
}

const char *glref_string(unsigned name) {
    const GLubyte *s = p_glGetString(name);
    return s ? (const char *)s : "";
}

/// Compiles a shader object from a source string. Returns shader id (>0) or -1.
int glref_shader_src(unsigned type, const char *src) {
    GLuint sh = p_glCreateShader(type);
    p_glShaderSource(sh, 1, &src, nullptr);
    p_glCompileShader(sh);
    GLint ok = 0;
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[8192]; GLsizei n = 0;
        p_glGetShaderInfoLog(sh, sizeof log, &n, log);
        set_err(std::string("compile failed: ") + log);
        p_glDeleteShader(sh);
        return -1;
    }
    return (int)sh;
}

/// Compiles a shader object from a file (the reference's shaders are passed by path).
int glref_shader_file(unsigned type, const char *path) {
    std::ifstream f(path);
    if (!f) { set_err(std::string("cannot open ") + path); return -1; }
    std::stringstream ss; ss << f.rdbuf();
    return glref_shader_src(type, ss.str().c_str());
}

/// Links a program from shader objects; binds fragment outputs 'out0..' by declared location.
int glref_program(const int *shaders, int n) {
    GLuint prog = p_glCreateProgram();
    for (int i = 0; i < n; i++) p_glAttachShader(prog, (GLuint)shaders[i]);
    p_glBindAttribLocation(prog, 0, "Position");
    p_glLinkProgram(prog);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[8192]; GLsizei len = 0;
        p_glGetProgramInfoLog(prog, sizeof log, &len, log);
        set_err(std::string("link failed: ") + log);
        p_glDeleteProgram(prog);
        return -1;
    }
    return (int)prog;
}

void glref_delete_program(int prog) { p_glDeleteProgram((GLuint)prog); }
void glref_delete_shader(int sh) { p_glDeleteShader((GLuint)sh); }

int glref_tex2d_rgba32f(int w, int h, const float *data) {
    GLuint t; p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, data);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    return (int)t;
}
void glref_delete_tex(int t) { GLuint u = (GLuint)t; p_glDeleteTextures(1, &u); }

/// Buffer texture of RGBA32F quads (the reference's BVH container, src/renderer.cpp:472-475).
int glref_tbo_rgba32f(const float *data, size_t nquads, int *out_buf) {
    GLuint b; p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_TEXTURE_BUFFER, b);
    p_glBufferData(GL_TEXTURE_BUFFER, (GLsizeiptr)(nquads * 16), data, GL_STATIC_DRAW);
    GLuint t; p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_BUFFER, t);
    p_glTexBuffer(GL_TEXTURE_BUFFER, GL_RGBA32F, b);
    if (out_buf) *out_buf = (int)b;
    return (int)t;
}
void glref_delete_buffer(int b) { GLuint u = (GLuint)b; p_glDeleteBuffers(1, &u); }

/// FBO with n RGBA32F colour attachments (draw buffers 0..n-1).
int glref_fbo(const int *texs, int n) {
    GLuint f; p_glGenFramebuffers(1, &f);
    p_glBindFramebuffer(GL_FRAMEBUFFER, f);
    GLenum bufs[8];
    for (int i = 0; i < n; i++) {
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, (GLuint)texs[i], 0);
        bufs[i] = GL_COLOR_ATTACHMENT0 + i;
    }
    p_glDrawBuffers(n, bufs);
    GLenum st = p_glCheckFramebufferStatus(GL_FRAMEBUFFER);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    if (st != GL_FRAMEBUFFER_COMPLETE) { set_err("FBO incomplete"); return -1; }
    return (int)f;
}
void glref_delete_fbo(int f) { GLuint u = (GLuint)f; p_glDeleteFramebuffers(1, &u); }

void glref_clear_fbo(int fbo, float r, float g, float b, float a) {
    p_glBindFramebuffer(GL_FRAMEBUFFER, (GLuint)fbo);
    p_glClearColor(r, g, b, a);
    p_glClear(GL_COLOR_BUFFER_BIT);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
}

void glref_use(int prog) { p_glUseProgram((GLuint)prog); }
int glref_uniform_loc(int prog, const char *name) { return p_glGetUniformLocation((GLuint)prog, name); }
void glref_uniform1i(int loc, int v) { p_glUniform1i(loc, v); }
void glref_uniform1ui(int loc, unsigned v) { p_glUniform1ui(loc, v); }
void glref_uniform1f(int loc, float v) { p_glUniform1f(loc, v); }
void glref_uniform3f(int loc, float a, float b, float c) { p_glUniform3f(loc, a, b, c); }
void glref_uniform4f(int loc, float a, float b, float c, float d) { p_glUniform4f(loc, a, b, c, d); }

void glref_bind_tex(int unit, int is_buffer_tex, int tex) {
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(is_buffer_tex ? GL_TEXTURE_BUFFER : GL_TEXTURE_2D, (GLuint)tex);
}

/// Draws the full-screen quad with the current program into 'fbo' (viewport w x h).
int glref_draw_quad(int fbo, int w, int h) {
    p_glBindFramebuffer(GL_FRAMEBUFFER, (GLuint)fbo);
    p_glViewport(0, 0, w, h);
    p_glBindVertexArray(g_quad_vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, g_quad_vbo);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, g_quad_ebo);
    p_glEnableVertexAttribArray(0);
    p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 0, nullptr);
    p_glDrawElements(GL_TRIANGLE_FAN, 4, GL_UNSIGNED_INT, nullptr);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { set_err("GL error " + std::to_string(e)); return -1; }
    return 0;
}

void glref_finish(void) { p_glFinish(); }

void glref_read_tex(int tex, float *out_rgba) {
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, out_rgba);
}

}  // extern "C"
