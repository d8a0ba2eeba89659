#!/usr/bin/env python3
"""ORACLE tooling -- test infrastructure only.

hlsl2cpp.py: build-time source-to-source pass that lets g++ compile the REFERENCE's own HLSL-2021 shader headers
(/root/reference/Source/ZetaRenderPass/**/*.hlsl{,i} and the C++/HLSL shared headers they include) as C++ against
oracle/ref_hlsl/hlsl_shim.h.  Nothing is copied into the repository: the rewritten files are written under oracle/_ref/gen/
(git-ignored, rebuilt by `make -C oracle -f _ref.mk`) and only exist where /root/reference exists.

The pass is purely lexical and keeps every arithmetic statement of the reference as it is written; what it changes is the
HLSL surface syntax C++ does not have:

  * `out T x` / `inout T x` / `in T x` parameters   ->  `T& x` / `T& x` / `T x`
  * `this.member`                                   ->  `this->member`
  * `[unroll]`, `[loop]`, `[branch]`, `[flatten]`, `[numthreads(..)]`, `[WaveSize(..)]`  ->  removed
  * `: register(..)`, `: SV_Xxx` semantics                                               ->  removed
  * unsuffixed floating literals (HLSL: float)      ->  `f`-suffixed (C++ would make them double)
  * `M._11` / `M._m00` matrix element access        ->  `M.m(0, 0)`
  * `#ifdef __cplusplus` in the shared headers      ->  a never-defined macro (the HLSL branch is the one that is compiled)
  * `ConstantBuffer<T> name`                        ->  `T name`
  * `static const` data members with initialisers   ->  `static inline const`
  * `row_major`, `precise`, `globallycoherent`, `nointerpolation` -> removed
  * `#include "X"`                                  ->  the rewritten copy of X (same relative layout under gen/)

usage: hlsl2cpp.py <reference root> <out dir> <file relative to reference root> [...]   (includes are followed recursively)
"""
import os
import re
import sys

FLOAT_LIT = re.compile(r"""
    (?<![\w.])                       # not glued to an identifier, a number or a swizzle
    (
        (?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?    # 1.0  1.  .5  1.0e-3
      | \d+[eE][+-]?\d+                        # 1e-4
    )
    (?![\w.])                        # no suffix (f, h, l) and not part of something longer
    """, re.X)

ATTR = re.compile(r"\[\s*(unroll|loop|branch|flatten|fastopt|allow_uav_condition|numthreads|WaveSize|earlydepthstencil|noinline|call)\s*(\([^\]]*\))?\s*\]")
SEMANTIC = re.compile(r"\)\s*:\s*SV_\w+")
PARAM_SEM = re.compile(r"(\w)\s*:\s*SV_\w+")
REGISTER = re.compile(r"\s*:\s*register\s*\([^)]*\)")
MAT_ELEM = re.compile(r"\._(?:m)?([0-4])([0-4])\b")
INCLUDE = re.compile(r'^(\s*#\s*include\s*)"([^"]+)"', re.M)
QUAL = re.compile(r"\b(row_major|precise|globallycoherent|nointerpolation|uniform)\b\s*")
CBUFFER = re.compile(r"\bConstantBuffer\s*<\s*([\w:]+)\s*>")
# parameter qualifiers: only inside parameter lists, i.e. after `(` or `,` (never `in`side a for-each or an identifier)
OUTPARAM = re.compile(r"(?<=[(,])(\s*)(?:inout|out)\s+((?:const\s+)?[\w:]+(?:\s*<[^<>()]*(?:<[^<>()]*>[^<>()]*)*>)?)\s+(\w+)")
INPARAM = re.compile(r"(?<=[(,])(\s*)in\s+(?=[\w:])")


def strip_comments_keep_layout(src):
    """returns src with comments blanked (same length, newlines kept) so that regexes never fire inside comments"""
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j]))
            i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def suffix_float_literals(code):
    def rep(m):
        return m.group(1) + "f"
    return FLOAT_LIT.sub(rep, code)


def fix_static_const_members(code):
    """`static const T name = init;` inside a struct body needs `inline` in C++ (non-integral types).  Applied to every
    `static const` declaration at brace depth >= 1 whose enclosing brace was opened by struct/class/namespace-less scope is
    hard to know lexically; `static inline const` is legal at namespace and class scope but not at block scope, so the
    pass tracks what opened each brace."""
    out = []
    stack = []          # kind of every open brace: 's' struct/class/union/enum, 'n' namespace, 'b' anything else
    i, n = 0, len(code)
    last_kw = None
    token = re.compile(r"\b(struct|class|union|namespace|enum)\b|[{};()]|static\s+const\b")
    pos = 0
    paren = 0
    pending = None
    for m in token.finditer(code):
        t = m.group(0)
        if t in ("struct", "class", "union", "namespace", "enum"):
            if paren == 0:
                pending = "n" if t == "namespace" else "s"
        elif t == "(":
            paren += 1
        elif t == ")":
            paren = max(0, paren - 1)
            pending = None if pending != "s" else pending
        elif t == "{":
            stack.append(pending or "b")
            pending = None
        elif t == "}":
            if stack:
                stack.pop()
            pending = None
        elif t == ";":
            pending = None
        else:   # static const
            scope = stack[-1] if stack else "n"
            if scope in ("s", "n") and paren == 0:
                out.append(code[pos:m.start()])
                out.append("static inline const")
                pos = m.end()
    out.append(code[pos:])
    return "".join(out)


MAIN_RE = re.compile(r"\[\s*numthreads\s*\(([^\]]*)\)\s*\]\s*void\s+main\s*\(([^)]*)\)", re.S)
SEM_TO_ARG = {"SV_DispatchThreadID": "zr_DTid", "SV_GroupID": "zr_Gid", "SV_GroupThreadID": "zr_GTid", "SV_GroupIndex": "zr_Gidx"}


def entry_point_wrapper(code):
    """for a compute shader: `zr_numthreads` (the [numthreads] attribute) and `zr_main_dispatch(DTid, Gid, GTid, Gidx)`, which forwards the
    system values main() asks for through its parameter semantics (they are stripped from the signature further down)"""
    m = MAIN_RE.search(code)
    if not m:
        return ""
    dims = [d.strip() for d in m.group(1).split(",")]
    args = []
    for prm in m.group(2).split(","):
        sem = prm.split(":")[-1].strip()
        args.append(SEM_TO_ARG[sem])
    return ("\nstatic const uint zr_numthreads[3] = {%s, %s, %s};\n"
            "static inline void zr_main_dispatch(uint3 zr_DTid, uint3 zr_Gid, uint3 zr_GTid, uint zr_Gidx) { main(%s); }\n"
            % (dims[0], dims[1], dims[2], ", ".join(args)))


def translate(src, relpath, special):
    bare = strip_comments_keep_layout(src)
    code = bare
    wrapper = entry_point_wrapper(code)
    code = code.replace("__cplusplus", "__ZR_REF_NEVER_DEFINED__")
    code = ATTR.sub("", code)
    code = REGISTER.sub("", code)
    code = SEMANTIC.sub(")", code)
    code = PARAM_SEM.sub(r"\1", code)
    code = QUAL.sub("", code)
    code = CBUFFER.sub(r"\1", code)
    code = re.sub(r"\bthis\s*\.", "this->", code)
    code = MAT_ELEM.sub(lambda m: ".m(%d, %d)" % ((int(m.group(1)) - (0 if "_m" in m.group(0) else 1)), (int(m.group(2)) - (0 if "_m" in m.group(0) else 1))), code)
    prev = None
    while prev != code:          # several qualified parameters in one list
        prev = code
        code = OUTPARAM.sub(r"\1\2& \3", code)
    code = INPARAM.sub(r"\1", code)
    # macro-constant swizzle: FLT_MAX.xxx -> float3(FLT_MAX)
    code = re.sub(r"\b([A-Z][A-Z0-9_]+)\.(x{2,4})\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)
    # half literals: 1.0h -> half(1.0f)
    code = re.sub(r"(?<![\w.])(\d+\.\d*|\d+)h\b", r"half(\1f)", code)
    # integer-literal swizzle: 0.xx -> int2(0)
    code = re.sub(r"(?<![\w.])(\d+)\.(x{2,4})\b", lambda m: "int%d(%s)" % (len(m.group(2)), m.group(1)), code)
    # scalar-literal swizzle, unsuffixed: 1.0.xxx -> float3(1.0f)
    code = re.sub(r"(?<![\w.])(\d+\.\d*)\.(x{2,4}|r{2,4})\b", lambda m: "float%d(%sf)" % (len(m.group(2)), m.group(1)), code)
    code = suffix_float_literals(code)
    # scalar-literal swizzle: 1.0f.xxx -> float3(1.0f)
    code = re.sub(r"(?<![\w.])(\d[\w.]*f)\.(x{2,4}|r{2,4})\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)
    # HLSL 2021 logical functions are C++ alternative tokens
    code = re.sub(r"\band\s*\(", "hlsl_and(", code)
    code = re.sub(r"\bor\s*\(", "hlsl_or(", code)
    code = fix_static_const_members(code)
    for pat, rep in special.get(os.path.basename(relpath), []) + special.get("*", []):
        code, k = re.subn(pat, rep, code)
    return code + wrapper


# File-specific lexical fixes (pattern, replacement): places where HLSL and C++ disagree on something the generic rules cannot see.
# Every entry keeps the arithmetic as written; it only resolves overloads / declarations the way DXC does.
SPECIAL = {
    "ReSTIR_PT_Sort.hlsl": [
        # 16-bit group dimensions multiplied with 32-bit IDs (HLSL promotes; the C++ vector templates do not mix element types)
        (r"const uint16_t2 GroupDim = uint16_t2\(", "const uint2 GroupDim = uint2("),
        # cast of the enum array to a vector
        # float2 -> int2 (HLSL converts implicitly, truncating)
        (r"neighborPixel = prevUV \* renderDim;", "neighborPixel = int2(prevUV * renderDim);"),
        (r"uint4 result = \(uint4\)error;", "uint4 result = uint4((uint)error[0], (uint)error[1], (uint)error[2], (uint)error[3]);"),
    ],
    "Display.hlsl": [
        # vertex-to-pixel interpolant semantics of the VSOut struct members
        (r"float4 PosSS : SV_Position;", "float4 PosSS;"), (r"float2 TexCoord : TEXCOORD;", "float2 TexCoord;"),
        # Texture2D::operator[] with the float2 SV_Position (HLSL converts float2 -> uint2 implicitly; only the debug views read it)
        (r"g_coat\[psin\.PosSS\.xy\]", "g_coat[uint2(psin.PosSS.xy)]"),
    ],
    "StaticTextureSamplers.hlsli": [
        # static samplers are identified by name (RendererCore.cpp:450-545 defines filter / address mode per name)
        (r"SamplerState\s+(g_sam\w+)\s*;", r'static const SamplerState \1 = SamplerState::Named("\1");'),
    ],
    "RT.hlsli": [
        # OffsetRayRTG: `int3 of_i = int_scale * geometricNormal;` -- HLSL converts float3 -> int3 implicitly (truncation toward zero)
        (r"int3 of_i = int_scale \* geometricNormal;", "int3 of_i = int3(int_scale * geometricNormal);"),
    ],
    "GBufferRT_Inline.hlsl": [
        # uint2 swizzle passed to an int2 parameter (HLSL converts silently; a C++ proxy cannot convert to two vector types)
        (r"GBufferRT::UVDifferentials\(DTid\.xy,", "GBufferRT::UVDifferentials(int2(DTid.xy),"),
    ],
    "HLSLCompat.h": [
        # member functions marked CONST in the shared headers are const on the C++ side; the HLSL branch drops the keyword, C++ needs it
        (r"#define CONST\s*\n", "#define CONST const\n"),
    ],
    "Material.h": [
        # the one getter the reference forgot to mark CONST (it is called on `const Material` values in RayQuery.hlsli)
        (r"half_ GetTransmissionDepth\(\)", "half_ GetTransmissionDepth() const"),
    ],
    "RayQuery.hlsli": [
        # Visibility_Segment, APPROXIMATE_EMISSIVE_SHADOW_RAY branch: hand the light's ID to the query, which implements the ABI's
        # order-independent definition of the approximate segment (hlsl_rt.h: triangles carrying that ID are not occluders)
        (r"(ray\.TMax = Math::PrevFloat32\(rayT \* 0\.999f - Math::NextFloat32\(ray\.TMin\)\);\s*ray\.Direction = wi;)",
         r"\1 g_rqIgnoreID = triID; g_rqHasIgnoreID = true;"),
        # `cond ? half : 0`: HLSL converts the literal to half; C++ cannot pick between half and int
        (r"\? mat\.GetTransmissionDepth\(\) : 0;", "? mat.GetTransmissionDepth() : half(0);"),
        (r"\? \(half\)mat\.GetSubsurface\(\) : 0;", "? (half)mat.GetSubsurface() : half(0);"),
    ],
    "Params.hlsli": [
        # DirectLighting/Emissive/Params.hlsli: the reference's compile-time switch of the emissive ReSTIR DI half-vector copy shift, 0 in the file.  Guarded so that the
        # `e1h` permutation of _ref.mk can compile the shaders with the switch at 1 (-DUSE_HALF_VECTOR_COPY_SHIFT=1) -- what a maintainer does by editing the 0
        (r"#define USE_HALF_VECTOR_COPY_SHIFT 0", "#ifndef USE_HALF_VECTOR_COPY_SHIFT\n#define USE_HALF_VECTOR_COPY_SHIFT 0\n#endif"),
    ],
    "Reservoir.hlsli": [
        # `.x` on a scalar (legal HLSL): give the scalar a 1-component vector type
        (r"float inF = g_inF\[DTid\]\.x;", "float1 inF = g_inF[DTid].x;"),
        (r"void UnpackMetadataX\(uint metadata\)", "void UnpackMetadataX(uint1 metadata)"),
    ],
    "ReSTIR_PT_SpatialSearch.hlsl": [
        (r"const int2 samplePosSS = round\(float2\(DTid\) \+ rotated\);", "const int2 samplePosSS = int2(round(float2(DTid) + rotated));"),
    ],
    "FireflyFilter.hlsl": [
        # int2 compared with uint2: HLSL converts the signed side to unsigned (a negative address is "beyond" the image)
        (r"any\(addr >= renderDim\)", "any(uint2(addr) >= renderDim)"),
        # uint2 swizzles passed to float2 / int2 parameters
        (r"Math::WorldPosFromScreenSpace\(DTid\.xy,", "Math::WorldPosFromScreenSpace(float2(DTid.xy),"),
        (r"FilterFirefly\(g_composited, color, DTid\.xy, GTid\.xy,", "FilterFirefly(g_composited, color, int2(DTid.xy), int2(GTid.xy),"),
    ],
    "TAA.hlsl": [
        (r"int2 closestDepthAddress = [^;]*;", "int2 closestDepthAddress = int2(0);"),
        # uint2 + int2 (HLSL: unsigned arithmetic, then back to int2 -- the same bits)
        (r"DTid\.xy \+ int2\(i, j\)", "int2(DTid.xy) + int2(i, j)"),
        (r"DTid\.xy \+ closestDepthAddress", "int2(DTid.xy) + closestDepthAddress"),
    ],
    "LightVoxelGrid.hlsli": [
        # SignNotZero<T> on an int3 argument: HLSL converts the float3 result to int3 and back; the values are +-1 either way
        (r"Math::SignNotZero\(voxelIdxCamSpace\)", "Math::SignNotZero(float3(voxelIdxCamSpace))"),
    ],
    "*": [
        # implicit float2 -> int2 (truncation) and int swizzle -> uint3 argument in the temporal reprojection code
        (r"int2 prevPixel = prevUV \* renderDim;", "int2 prevPixel = int2(prevUV * renderDim);"),
        (r"RNG::PCG3d\(prevPixel\.xyx\)", "RNG::PCG3d(uint3(prevPixel.xyx))"),
        (r"int2 samplePosSS = prevPixel \+ \(i > 0\) \* offset;", "int2 samplePosSS = int2(float2(prevPixel) + (float)(i > 0) * offset);"),
        (r"RNG::PCG3d\(samplePosSS\.xyx\)", "RNG::PCG3d(uint3(samplePosSS.xyx))"),
        (r"RNG::PCG3d\(candidate\.posSS\.xyx\)", "RNG::PCG3d(uint3(candidate.posSS.xyx))"),
        (r"const int2 posSS_i = round\(float2\(DTid\) \+ rotated\);", "const int2 posSS_i = int2(round(float2(DTid) + rotated));"),
        (r"RNG::PCG3d\(posSS_i\.xyx\)", "RNG::PCG3d(uint3(posSS_i.xyx))"),
        # `out uint2 swizzledGid` receives a uint16_t2 variable: HLSL converts on the way out, a C++ reference cannot
        (r"\buint16_t2 swizzledGid;", "uint2 swizzledGid;"),
        # HLSL `groupshared T x[N];` at file scope: one copy per thread group -> per-thread storage owned by the group runner
        (r"\bgroupshared\b", "ZR_GROUPSHARED"),
    ],
}


def process(ref_root, out_dir, rel, done, special):
    rel = os.path.normpath(rel)
    if rel in done:
        return
    done.add(rel)
    src_path = os.path.join(ref_root, rel)
    with open(src_path, encoding="utf-8", errors="replace") as f:
        src = f.read()
    code = translate(src, rel, special)

    def inc(m):
        target = os.path.normpath(os.path.join(os.path.dirname(rel), m.group(2)))
        if os.path.exists(os.path.join(ref_root, target)):
            process(ref_root, out_dir, target, done, special)
            # keep the include relative: the gen tree mirrors the reference tree
            return m.group(0)
        return m.group(0)
    code = INCLUDE.sub(inc, code)
    dst = os.path.join(out_dir, rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    header = "// GENERATED at build time by oracle/ref_hlsl/hlsl2cpp.py from %s -- never committed\n" % src_path
    with open(dst, "w") as f:
        f.write(header + code)


def write_swizzles(out_dir):
    """hlsl_swizzles_{2,3,4}.inc: one proxy member per swizzle of 2..4 components, in xyzw and rgba spelling"""
    os.makedirs(out_dir, exist_ok=True)
    import itertools
    for n in (2, 3, 4):
        lines = []
        for names in ("xyzw"[:n], "rgba"[:n]):
            for k in (2, 3, 4):
                for combo in itertools.product(range(n), repeat=k):
                    lines.append("Swz<T, %d, %s> %s;" % (n, ", ".join(map(str, combo)), "".join(names[i] for i in combo)))
        with open(os.path.join(out_dir, "hlsl_swizzles_%d.inc" % n), "w") as f:
            f.write("\n".join(lines) + "\n")


def main():
    ref_root, out_dir = sys.argv[1], sys.argv[2]
    write_swizzles(out_dir)
    done = set()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    special = dict(SPECIAL)
    try:
        import hlsl2cpp_special
        for k, v in hlsl2cpp_special.SPECIAL.items():
            special.setdefault(k, [])
            special[k] = special[k] + v
    except ImportError:
        pass
    for rel in sys.argv[3:]:
        process(ref_root, out_dir, rel, done, special)
    print("hlsl2cpp: %d files -> %s" % (len(done), out_dir))


if __name__ == "__main__":
    main()
