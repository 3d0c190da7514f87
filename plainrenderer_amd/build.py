"""Builds the HIP backend for gfx950 with hipcc, in-tree: libplr.so (the C ABI, the C++ frame host, the benchmarked PLR_MATH_FAST kernel set and the kernels that
serve both math modes) and libplr_exact.so (csrc/kernels_exact/: the PLR_MATH_EXACT launch paths of the GI trace, the GI filters, the deferred shade, TAA and bloom -
the reference's operation order, bit-exact against the oracle; loaded by libplr.so on demand: plr_set_math_mode(PLR_MATH_EXACT), or an execution the fast set declines).

Every translation unit is compiled with -ffp-contract=off: the kernels' arithmetic is specified operation
by operation (no FMA contraction) so results are reproducible against an IEEE scalar evaluation.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PLR_BUILD_TAG=<tag> (experiments: an A/B pair of libraries in one tree): objects under csrc/_obj_<tag>, the library as libplr_<tag>.so; pick it at run time with
# PLR_LIB=<path> (plainrenderer_amd/backend.py). The default build has no tag.
_TAG = os.environ.get("PLR_BUILD_TAG", "")
OBJ_DIR = os.path.join(CSRC, "_obj" + ("_" + _TAG if _TAG else ""))
LIB_PATH = os.path.join(HERE, "libplr" + ("_" + _TAG if _TAG else "") + ".so")
EXACT_LIB_PATH = os.path.join(HERE, "libplr_exact" + ("_" + _TAG if _TAG else "") + ".so")  # found by libplr.so next to itself (same tag)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA_FLAGS = os.environ.get("PLR_EXTRA_FLAGS", "").split()  # experiment hook, e.g. -DPLR_SHADE_WAVES=5

# kernels_fast/*.hip: restructured kernels for PLR_MATH_FAST. FMA contraction on, divide/sqrt may use the v_rcp/v_rsq based
# sequences; still no -ffast-math (NaN guards and comparisons keep IEEE semantics).
FAST_FLAGS_REPLACE = {"-ffp-contract=off": "-ffp-contract=fast-honor-pragmas", "-fhip-fp32-correctly-rounded-divide-sqrt": "-fno-hip-fp32-correctly-rounded-divide-sqrt"}
FAST_FLAGS_EXTRA = ["-DPLR_FAST_SET=1"] + os.environ.get("PLR_FAST_FLAGS", "").split()  # PLR_FAST_SET: detmath.h min/max as single instructions; + experiment hook

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    # no SLP vectorisation: pairing scalar fp32 operations into v_pk_fma/mul/add_f32 costs register-pair shuffles (v_mov) and measured
    # slower in every VALU-bound kernel (4K frame 1.096 -> 1.051 ms: shade 218 -> 200 us, TAA 180 -> 163, spatial filter 113 -> 107)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result",
]


def _rocm_lib_dir():
    """where librccl lives: $ROCM_PATH/lib, else the lib directory next to the hipcc in use, else /opt/rocm/lib"""
    for root in (os.environ.get("ROCM_PATH"), os.path.dirname(os.path.dirname(os.path.realpath(HIPCC)))):
        if root and os.path.exists(os.path.join(root, "lib", "librccl.so")):
            return os.path.join(root, "lib")
    return "/opt/rocm/lib"


def _sources(subdirs=("kernels", "kernels_fast", "frontend")):
    out = [os.path.join(CSRC, "backend.cpp")] if "frontend" in subdirs else []
    for sub in subdirs:
        d = os.path.join(CSRC, sub)
        if os.path.isdir(d):
            for f in sorted(os.listdir(d)):
                if f.endswith((".hip", ".cpp")):
                    out.append(os.path.join(d, f))
    return out


def _headers_digest():
    h = hashlib.sha1()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for dp, _, files in os.walk(root):
            if "_obj" in os.path.basename(dp):
                continue
            for f in sorted(files):
                if f.endswith((".h", ".hpp")):
                    with open(os.path.join(dp, f), "rb") as fh:
                        h.update(fh.read())
    h.update(" ".join(FLAGS + sorted(FAST_FLAGS_REPLACE.values()) + EXTRA_FLAGS + FAST_FLAGS_EXTRA).encode())
    return h.hexdigest()


def _compile(src, digest, verbose):
    obj = os.path.join(OBJ_DIR, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
    stamp = obj + ".stamp"
    with open(src, "rb") as fh:
        key = hashlib.sha1(fh.read() + digest.encode()).hexdigest()
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj, False
    flags = FLAGS
    if os.sep + "kernels_fast" + os.sep in src:
        flags = [FAST_FLAGS_REPLACE.get(f, f) for f in FLAGS] + FAST_FLAGS_EXTRA
    # a source may ask for extra flags of its own with a line "// PLR_BUILD_FLAGS: <flags>" (they go last, so they win over the set's defaults)
    with open(src) as fh:
        for line in fh:
            if line.startswith("// PLR_BUILD_FLAGS:"):
                flags = flags + line.split(":", 1)[1].split()
    cmd = [HIPCC, "-x", "hip"] + flags + EXTRA_FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip() and verbose:
        print(r.stderr)
    with open(stamp, "w") as fh:
        fh.write(key)
    return obj, True


def build(verbose=False, jobs=None):
    """Compile every HIP source for gfx950 and link libplr.so. Cross-compiles without a GPU."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    digest = _headers_digest()
    srcs, exact_srcs = _sources(), _sources(("kernels_exact",))
    jobs = jobs or min(6, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, digest, verbose), srcs + exact_srcs))
    objs = [o for o, _ in results[:len(srcs)]]
    exact_objs = [o for o, _ in results[len(srcs):]]

    def link(path, objects, extra, relink):
        if not relink and os.path.exists(path):
            return
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", path] + objects + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # librccl: the band exchange of the C++ host (csrc/frontend/band_exchange.cpp) calls ncclSend / ncclRecv / ncclAllReduce directly
    link(LIB_PATH, objs, ["-Wl,-soname," + os.path.basename(LIB_PATH), "-L" + _rocm_lib_dir(), "-lrccl"], any(c for _, c in results[:len(srcs)]))
    # the exact set resolves the backend's symbols (PassCtx, the shader registry) against the libplr.so that loads it: linked against it, found through $ORIGIN
    link(EXACT_LIB_PATH, exact_objs, ["-Wl,-soname," + os.path.basename(EXACT_LIB_PATH), "-Wl,-rpath,$ORIGIN", "-L" + HERE, "-l:" + os.path.basename(LIB_PATH),
                                      "-Wl,--no-undefined"], any(c for _, c in results))
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
