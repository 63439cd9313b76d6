"""Build sea_amd/csrc/libsea_hip.so (gfx950 only) with hipcc.  `python -m sea_amd.build [--force]`.

The library is built IN-TREE so that it travels with the repository snapshot to the GPU box; it links only the HIP
runtime (no torch types cross the C ABI, include/sea_hip.h).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "build")          # objects + compiler temporaries (git-ignored; only the .so has to travel to the GPU box)
LIB = os.path.join(CSRC, "libsea_hip.so")
SOURCES = ["core.hip", "gemm.hip", "gemm_norm.hip", "mlp_fused.hip", "attention.hip", "rowops.hip", "train.hip", "bwd.hip", "attention_bwd.hip", "kvstep.hip", "gemv.hip", "chain.hip", "gemm_adaln.hip", "mlp_block.hip", "adaln_qkv.hip", "gemm256.hip", "gemm_ws.hip"]
HEADERS = ["sea_common.hpp", "gemm_core.hpp", "norm_epilogue.hpp", "gemm_tile.hpp", os.path.join("..", "..", "include", "sea_hip.h")]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs, so epilogues / softmax read them without v_accvgpr_read moves.
# -ffast-math -fno-finite-math-only: reciprocal / approximate-function / reassociation freedoms for the row kernels and epilogues; infinities
# and NaNs keep their meaning (the attention kernels mask with -inf); f32 denormals flush to zero (no range-fixup code around v_exp / v_rcp).  Parity bars (fp32 <= 1e-4) are checked with these flags on.
FLAGS = (os.environ.get("SEA_EXTRA_FLAGS", "").split()) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-ffast-math", "-fno-finite-math-only", "-fgpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


# Per-file flags: none.  (Round 1 built gemm.hip and attention_bwd.hip with -fno-slp-vectorize to mask wrong values of
# qkv_rope_kernel<bf16, 64, 64>; the cause is now known — see lint_isa below — and is excluded for EVERY file by the lint instead.)
FILE_FLAGS: dict = {}

# gfx950 erratum found in round 2 (tools/isa_variants.py, tools/pk_opsel_bench.hip): a packed fp32 VOP3P instruction
# (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose LOW result half selects the HIGH dword of its src1 pair (op_sel:[x,1] / op_sel:[x,1,y])
# returns a wrong low result in lanes 48-63 — about 2..8 of 10^4 executions — whenever another wave of the CU is issuing MFMAs at that
# moment; never when no MFMA runs beside it; independent of the registers, of wait states around it, of negation modifiers, of the MFMA
# result form.  High-half selects of src0 / src2 and low-half selects for the high result are not affected.  hipcc (ROCm 7.2) forms such
# instructions when its SLP vectoriser packs scalar fp32 code with a lane swap (the RoPE rotation, sums of squares, horizontal adds), so
# every device assembly listing of the build is checked and the build FAILS if one is present: the source is then rewritten with the
# scalar-lane helpers of sea_common.hpp (fma1 / mul1 / add1), which the vectoriser cannot pack.
_PK_SRC1_HI = __import__("re").compile(r"^\s*v_pk_(?:fma|mul|add)_f32\b.*\bop_sel:\[[01],1[,\]]")


def lint_isa(asm_path: str) -> list:
    """[(kernel symbol, line number, instruction)] of every packed fp32 instruction with the forbidden src1 select in a device .s file."""
    bad, kernel = [], "?"
    with open(asm_path) as f:
        for no, line in enumerate(f, 1):
            if line[:1].isalpha() or line[:1] == "_":   # "symbol:   ; @symbol" opens a function
                kernel = line.split(":")[0]
            elif _PK_SRC1_HI.match(line):
                bad.append((kernel, no, line.strip()))
    return bad


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _asm_of(obj: str) -> str:
    return obj[:-2] + "-hip-amdgcn-amd-amdhsa-gfx950.s"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what is stale, lint every device listing, link.  Serialised by a file lock: several rank processes (bench.py --gpus N, mp.spawn tests)
    may call this at once on a box where the objects are missing (csrc/build does not travel), and must not remove each other's temporaries."""
    import fcntl

    os.makedirs(BUILD, exist_ok=True)
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    os.makedirs(BUILD, exist_ok=True)
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs) or not os.path.exists(_asm_of(obj)):
            # -save-temps=obj: the device assembly listing lands next to the object (one compile), for lint_isa
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(s, []) + ["-save-temps=obj", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return r.returncode, r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for rc, out in ex.map(run, jobs):
                if out.strip() and verbose:
                    print(out)
                if rc != 0:
                    raise RuntimeError("hipcc failed:\n" + out)
    for f in os.listdir(BUILD):   # -save-temps leaves preprocessed sources, bitcode and host listings (tens of MB): only objects and device listings are kept
        stem = f.split("-hip-")[0].split(".")[0]
        if (not (f.endswith(".o") and "-hip-" not in f) and not f.endswith("-hip-amdgcn-amd-amdhsa-gfx950.s")) or stem + ".hip" not in srcs:
            try:
                os.remove(os.path.join(BUILD, f))
            except FileNotFoundError:
                pass
    findings = []
    for obj in objs:
        for kernel, no, ins in lint_isa(_asm_of(obj)):
            findings.append(f"{os.path.basename(_asm_of(obj))}:{no}: {kernel}: {ins}")
    if findings:
        for obj in objs:   # a later build must not link these objects
            if lint_isa(_asm_of(obj)):
                os.remove(obj)
        raise RuntimeError("sea_amd.build: packed fp32 instructions with the low half reading the high dword of src1 (gfx950 erratum, see build.py):\n  "
                           + "\n  ".join(findings[:40]) + (f"\n  ... {len(findings) - 40} more" if len(findings) > 40 else ""))
    if force or jobs or _stale(LIB, objs):
        rc, out = run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if rc != 0:
            raise RuntimeError("link failed:\n" + out)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
