#!/usr/bin/env python3
"""Builds oracle/_ref/libgs_ref.so: the REFERENCE'S OWN compute shaders, compiled for the CPU.

TEST INFRASTRUCTURE ONLY -- the checker the restated oracle (gs_oracle.c) is pinned against.

The reference cannot be built here (no Vulkan loader/ICD, no glslang, no glm), but its arithmetic is all in
GLSL text under /root/reference/src/shaders/.  This script reads that text where it lies (nothing is copied into
the repository; generated sources live in a temporary directory and only the .so lands in oracle/_ref/, which
is git-ignored), rewrites what C++ cannot parse, and compiles each shader's main() against
oracle/glsl_cpu/glsl_compat.hpp with g++ -ffp-contract=off.  The rewrites are purely syntactic:

  1. `#include "./common.glsl"` is replaced by that file's text; `#version` / `#extension` lines are dropped.
  2. resource declarations become globals of the same names:
       layout(...) [readonly|writeonly] buffer B { T name[]; };   ->  static buffer<T> name;
       layout(...) uniform B { members };  (UBO and push constants) ->  static <member>;  for each member
       layout(...) uniform writeonly image2D name;                 ->  static image2D name;
       layout(local_size_x = X, local_size_y = Y, local_size_z = Z) in;  ->  static const uint local_size[3] = {X, Y, Z};
  3. floating-point literals without a suffix get an `f` (in GLSL `2.0` is a float; in C++ it would be a double
     and silently widen the arithmetic).
  4. the text is wrapped in `namespace glsl { namespace cs_<name> { ... } }` followed by glsl_cpu/harness.hpp.

The radix sort (sort/hist.comp + sort/sort.comp x 8, Renderer.cpp:598-629) is compiled and RUN like the rest
(gsr_radix_sort_pairs): its workgroups -- barrier(), `shared` arrays, subgroup operations, shared-memory atomics -- execute
on glsl_cpu/workgroup.hpp (one fiber per invocation).  Two more syntactic rewrites apply to those two files:

  5. comments are stripped first (the buffer blocks carry comments inside their braces);
  6. `shared T[N] name;` -> `static thread_local T name[(N) + 64];` (one copy per host thread = per running workgroup; the 64
     elements of slack because sort.comp:128 reads `sums[lsID]` for every lane of a 32-wide subgroup from an 8-element array --
     harmless on a GPU, where only lanes < 8 reach the result, and kept harmless here);
     `layout (local_size_x = X) in;` -> `static const uint local_size[3] = {X, 1, 1};`.

The per-frame checks use gsr_sort_pairs (std::stable_sort: what eight stable LSD passes over all 64 key bits amount to);
tests/test_oracle_vs_ref.py pins the shader text's result to it.

The reference's HOST arithmetic runs too (gsr_load_records, gsr_update_uniforms, gsr_camera_translate): the text of
GSScene.cpp:17-24 (struct VertexStorage), the body of GSScene::load's conversion loop (GSScene.cpp:37-58), GSScene.h's struct Vertex,
Renderer.h's struct UniformBuffer and struct Camera, and Renderer::updateUniforms (Renderer.cpp:719-754) whole are cut out of
the files where they lie and compiled verbatim against glsl_cpu/glm_stub.hpp (glm is absent here: the stub restates the inside
of the glm calls that text makes, citing glm's files; the order of calls, constants, sign flips and the double / float mix are
the reference's own text).  The objects that text touches -- plyFile, verteces, swapchain, camera, uniformBuffer -- are supplied
by the generated harness with the members it uses.

Usage: python oracle/build_ref.py [--reference /root/reference] [--keep-generated DIR]
"""
import argparse
import hashlib
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
SHADERS = ["precomp_cov3d", "preprocess", "prefix_sum", "preprocess_sort", "tile_boundary", "render"]
WORKGROUP_SHADERS = {"sort_hist": "sort/hist.comp", "sort_sort": "sort/sort.comp"}  # need barrier() / shared / subgroup operations
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-w"]

FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")
BUFFER_BLOCK = re.compile(r"layout\s*\([^)]*\)\s*(?:readonly\s+|writeonly\s+)?buffer\s+\w+\s*\{\s*(\w+)\s+(\w+)\s*\[\s*\]\s*;\s*\}\s*;")
UNIFORM_BLOCK = re.compile(r"layout\s*\([^)]*\)\s*uniform\s+\w+\s*\{(.*?)\}\s*;", re.S)
IMAGE_DECL = re.compile(r"layout\s*\([^)]*\)\s*uniform\s+(?:writeonly\s+|readonly\s+)?image2D\s+(\w+)\s*;")
LOCAL_SIZE = re.compile(r"layout\s*\(\s*local_size_x\s*=\s*([^,]+),\s*local_size_y\s*=\s*([^,]+),\s*local_size_z\s*=\s*([^)]+)\)\s*in\s*;")
LOCAL_SIZE_X = re.compile(r"layout\s*\(\s*local_size_x\s*=\s*([^,)]+)\)\s*in\s*;")
SHARED_ARRAY = re.compile(r"\bshared\s+(\w+)\s*\[([^\]]+)\]\s*(\w+)\s*;")


def glsl_to_cpp(text, shader_dir, workgroup=False):
    if workgroup:  # rewrites 5 and 6
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        text = SHARED_ARRAY.sub(lambda m: f"static thread_local {m.group(1)} {m.group(3)}[({m.group(2)}) + 64];", text)
        text = LOCAL_SIZE_X.sub(lambda m: f"static const uint local_size[3] = {{{m.group(1)}, 1, 1}};", text)
        if re.search(r"\bshared\b", text):
            raise RuntimeError("unhandled shared declaration")

    def include(m):
        with open(os.path.join(shader_dir, m.group(1))) as f:
            return f.read() + "\n"
    text = re.sub(r'^\s*#include\s+"([^"]+)"\s*$', include, text, flags=re.M)
    text = re.sub(r"^\s*#(version|extension)\b.*$", "", text, flags=re.M)
    text = BUFFER_BLOCK.sub(lambda m: f"static buffer<{m.group(1)}> {m.group(2)};", text)
    text = IMAGE_DECL.sub(lambda m: f"static image2D {m.group(1)};", text)

    def members(m):
        decls = [d.strip() for d in m.group(1).split(";") if d.strip()]
        return "\n".join(f"static {d};" for d in decls)
    text = UNIFORM_BLOCK.sub(members, text)
    text = LOCAL_SIZE.sub(lambda m: f"static const uint local_size[3] = {{{m.group(1)}, {m.group(2)}, {m.group(3)}}};", text)
    if re.search(r"\blayout\s*\(", text):
        raise RuntimeError("unhandled layout declaration:\n" + "\n".join(l for l in text.splitlines() if "layout" in l))
    out = []
    for line in text.splitlines():  # rewrite 3, never inside a preprocessor line or a // comment
        if line.lstrip().startswith("#"):
            out.append(line)
            continue
        code, sep, comment = line.partition("//")
        out.append(FLOAT_LIT.sub(r"\1f", code) + sep + comment)
    return "\n".join(out)


def braces(text, start):
    """Index just past the brace that closes the one at text[start]."""
    assert text[start] == "{"
    depth = 0
    for k in range(start, len(text)):
        if text[k] == "{":
            depth += 1
        elif text[k] == "}":
            depth -= 1
            if depth == 0:
                return k + 1
    raise RuntimeError("unbalanced braces")


def cut(text, pattern, what):
    """The text from the match of `pattern` (which ends at an opening brace) through the matching closing brace (+ a following ';')."""
    m = re.search(pattern, text)
    if not m:
        raise RuntimeError(f"build_ref: cannot find {what} in the reference")
    end = braces(text, m.end() - 1)
    if text[end:end + 1] == ";":
        end += 1
    return text[m.start():end], text[m.end():end - 1 - (1 if text[end - 1] == ";" else 0)]


def host_text_to_cpp(reference):
    """cs_host.cpp: the reference's loader loop and camera uniforms, verbatim, against glm_stub.hpp."""
    def read(rel):
        with open(os.path.join(reference, rel)) as f:
            return f.read()
    scene_cpp, scene_h = read("src/GSScene.cpp"), read("src/GSScene.h")
    rend_cpp, rend_h = read("src/Renderer.cpp"), read("src/Renderer.h")
    vertex_storage, _ = cut(scene_cpp, r"struct VertexStorage\s*\{", "struct VertexStorage")
    vertex, _ = cut(scene_h, r"struct Vertex\s*\{", "struct Vertex")
    uniform_buffer, _ = cut(rend_h, r"struct alignas\(16\) UniformBuffer\s*\{", "struct UniformBuffer")
    camera, _ = cut(rend_h, r"struct Camera\s*\{", "struct Camera")
    _, load_loop = cut(scene_cpp, r"for \(auto i = 0; i < header\.numVertices; i\+\+\)\s*\{", "GSScene::load's loop")
    _, update_uniforms = cut(rend_cpp, r"void Renderer::updateUniforms\(\)\s*\{", "Renderer::updateUniforms")
    return f"""// GENERATED by oracle/build_ref.py from the reference's sources where they lie -- never committed.
#define NDEBUG 1  // (the loop asserts that the PLY's normals are zero: GSScene.cpp:55-57)
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "glm_stub.hpp"
namespace ref_host {{
// ---- GSScene.cpp:17-24
{vertex_storage}
// ---- GSScene.h:41-46
{vertex}
// ---- Renderer.h:21-29
{uniform_buffer}
// ---- Renderer.h:40-50
{camera}
static_assert(sizeof(VertexStorage) == 62 * 4 && sizeof(Vertex) == 60 * 4 && sizeof(UniformBuffer) == 160, "record layouts");
// what the text below touches, with the members it uses
struct FakeIfstream {{
    const char* p;
    void read(char* dst, size_t n) {{ std::memcpy(dst, p, n); p += n; }}
    bool is_open() const {{ return true; }}
    bool eof() const {{ return false; }}
}};
struct FakeSwapchain {{ struct {{ uint32_t width, height; }} swapchainExtent; }};
struct FakeUniformBuffer {{
    void* dst;
    void upload(const void* src, size_t n, size_t) {{ std::memcpy(dst, src, n); }}
}};
static Camera camera;
static FakeSwapchain swapchain_object, *swapchain = &swapchain_object;
static FakeUniformBuffer uniform_object, *uniformBuffer = &uniform_object;

// ---- Renderer.cpp:719-754, verbatim
void updateUniforms() {{{update_uniforms}}}

extern "C" void gsr_load_records(const float* records, uint64_t n, float* vertices) {{
    FakeIfstream plyFile{{reinterpret_cast<const char*>(records)}};
    Vertex* verteces = reinterpret_cast<Vertex*>(vertices);
    struct {{ uint64_t numVertices; }} header{{n}};
    for (uint64_t i = 0; i < header.numVertices; i++) {{
        // ---- GSScene.cpp:37-58, verbatim
{load_loop}
    }}
}}
// cam: position[3], rotation (w, x, y, z), fov, near, far  (the gs_camera of include/gs3d_hip.h)
static void set_camera(const float* cam) {{
    camera.position = glm::vec3(cam[0], cam[1], cam[2]);
    camera.rotation = glm::quat(cam[3], cam[4], cam[5], cam[6]);
    camera.fov = cam[7];
    camera.nearPlane = cam[8];
    camera.farPlane = cam[9];
}}
extern "C" void gsr_update_uniforms(const float* cam, uint32_t width, uint32_t height, void* out160) {{
    set_camera(cam);
    swapchain_object.swapchainExtent.width = width;
    swapchain_object.swapchainExtent.height = height;
    uniform_object.dst = out160;
    updateUniforms();
}}
extern "C" void gsr_camera_translate(const float* cam, const float* t, float* position_out) {{
    set_camera(cam);
    camera.translate(glm::vec3(t[0], t[1], t[2]));  // Renderer.h:47-49
    position_out[0] = camera.position.x;
    position_out[1] = camera.position.y;
    position_out[2] = camera.position.z;
}}
}}  // namespace ref_host
"""


def sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("GS_REFERENCE", "/root/reference"))
    ap.add_argument("--keep-generated", default=None, help="also leave the generated C++ here (debugging; never commit)")
    args = ap.parse_args()
    shader_dir = os.path.join(args.reference, "src", "shaders")
    if not os.path.isfile(os.path.join(shader_dir, "render.comp")):
        print(f"build_ref: no reference shaders under {shader_dir}; keeping any prebuilt oracle/_ref", file=sys.stderr)
        return 0 if os.path.exists(os.path.join(OUT_DIR, "libgs_ref.so")) else 1
    os.makedirs(OUT_DIR, exist_ok=True)
    compat = os.path.join(HERE, "glsl_cpu")
    provenance = []
    with tempfile.TemporaryDirectory() as tmp:
        gen_dir = args.keep_generated or tmp
        os.makedirs(gen_dir, exist_ok=True)
        objs = []
        files = {name: name + ".comp" for name in SHADERS}
        files.update(WORKGROUP_SHADERS)
        for name in list(files) + ["common"]:
            path = os.path.join(shader_dir, "common.glsl" if name == "common" else files[name])
            provenance.append(f"{os.path.relpath(path, args.reference)} sha256={sha256(path)}")
        for name in files:
            with open(os.path.join(shader_dir, files[name])) as f:
                body = glsl_to_cpp(f.read(), shader_dir, workgroup=name in WORKGROUP_SHADERS)
            src = os.path.join(gen_dir, f"cs_{name}.cpp")
            header = "workgroup.hpp" if name in WORKGROUP_SHADERS else "glsl_compat.hpp"
            with open(src, "w") as f:
                f.write(f'#include "{header}"\n#define CS_{name.upper()} 1\n'
                        f"namespace glsl {{ namespace cs_{name} {{\n{body}\n#include \"harness.hpp\"\n}} }}\n")
            obj = os.path.join(tmp, f"cs_{name}.o")
            subprocess.check_call(["g++", *CXXFLAGS, "-I", compat, "-c", src, "-o", obj])
            objs.append(obj)
        for rel in ("src/GSScene.cpp", "src/GSScene.h", "src/Renderer.cpp", "src/Renderer.h"):
            provenance.append(f"{rel} sha256={sha256(os.path.join(args.reference, rel))}")
        hsrc = os.path.join(gen_dir, "cs_host.cpp")
        with open(hsrc, "w") as f:
            f.write(host_text_to_cpp(args.reference))
        hobj = os.path.join(tmp, "cs_host.o")
        subprocess.check_call(["g++", *[x for x in CXXFLAGS if x != "-std=c++17"], "-std=c++20", "-I", compat, "-c", hsrc, "-o", hobj])
        objs.append(hobj)
        prov = os.path.join(gen_dir, "provenance.cpp")
        with open(prov, "w") as f:
            lines = "\\n".join(provenance)
            f.write(f'extern "C" const char* gsr_sources() {{ return "{lines}"; }}\n')
        pobj = os.path.join(tmp, "provenance.o")
        subprocess.check_call(["g++", *CXXFLAGS, "-c", prov, "-o", pobj])
        sobj = os.path.join(tmp, "sort_pairs.o")
        subprocess.check_call(["g++", *CXXFLAGS, "-c", os.path.join(compat, "sort_pairs.cpp"), "-o", sobj])
        out = os.path.join(OUT_DIR, "libgs_ref.so")
        subprocess.check_call(["g++", "-shared", "-fopenmp", *objs, pobj, sobj, "-o", out, "-lm"])
    print(f"build_ref: {out}\n  " + "\n  ".join(provenance))
    return 0


if __name__ == "__main__":
    sys.exit(main())
