"""Diagnostic for the qkv_rope_kernel<bf16, 64, 64> wrong-value defect: build hand-edited ISA variants of that ONE kernel and
replay each on bit-identical inputs.

    python tools/isa_variants.py build      (build container: hipcc -S, text edits, clang -x assembler, ld.lld -> tools/isa_out/*.hsaco)
    python tools/isa_variants.py run        (GPU box: load every .hsaco with hipModuleLoad, replay the self.qkv_rope launch of the cfg2 plan RUNS times)

The kernel takes its whole launch description (QkvLaunch, gemm.hip) by value, so the replay builds that struct from the plan's own record and
launches the variant with hipModuleLaunchKernel; workspaces are restored before every launch and compared bit for bit with the first replay.
"""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "isa_out")
KERNEL = "_Z15qkv_rope_kernelIDF16bLi64ELi64ELb0EEv9QkvLaunch"
LLVM = "/opt/rocm/lib/llvm/bin"
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-ffast-math", "-fno-finite-math-only", "-fgpu-flush-denormals-to-zero"]
VGPR = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
NOP = "\ts_nop 7\n"


def _func_span(lines):
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end") or lines[i].strip().startswith(".end_amdhsa_kernel"))
    return a, b


def _edit(text, pattern, before=False, what=NOP, only_after_line=None):
    """Insert `what` after (or before) every instruction of the kernel whose text matches `pattern`."""
    lines = text.splitlines(keepends=True)
    a, b = _func_span(lines)
    rx = re.compile(pattern)
    out, n = [], 0
    for i, l in enumerate(lines):
        hit = a <= i < b and rx.search(l) and not l.lstrip().startswith(";")
        if hit and before:
            out.append(what)
            n += 1
        out.append(l)
        if hit and not before:
            out.append(what)
            n += 1
    return "".join(out), n


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "sea_amd", "csrc", "gemm.hip")

    def asm(name, flags):
        s = os.path.join(OUT, name + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", src, "-o", s], check=True, stderr=subprocess.DEVNULL)
        return open(s).read()

    slp = asm("slp", BASE + VGPR)
    variants = {
        "v0_slp": slp,
        "v1_noslp": asm("noslp", BASE + VGPR + ["-fno-slp-vectorize"]),
        "v2_slp_agprform": asm("slp_agpr", BASE),
    }
    edits = {
        "v3_nop_after_every_pk": r"\bv_pk_(mul|fma|add)_f32\b",
        "v4_nop_after_pk_mul": r"\bv_pk_mul_f32\b",
        "v5_nop_after_pk_fma": r"\bv_pk_fma_f32\b",
        "v6_nop_after_vmcnt0": r"\bs_waitcnt vmcnt\(0\)",
        "v7_nop_after_exec_write": r"\b(s_and_saveexec_b64|s_andn2_saveexec_b64|s_or_b64 exec|s_xor_b64 exec)",
        "v8_nop_after_cvt_pk": r"\bv_cvt_pk_bf16_f32\b",
        "v9_nop_after_barrier": r"\bs_barrier\b",
        "v10_nop_after_pk_add": r"\bv_pk_add_f32\b",
    }
    for name, pat in edits.items():
        variants[name], n = _edit(slp, pat)
        print(name, "edits:", n)
    variants["v11_nop_before_every_pk"], n = _edit(slp, r"\bv_pk_(mul|fma|add)_f32\b", before=True)
    # literal rewrites of the rotation of the second row block (i = 1) of the first column block (j = 0) — where every failure sits:
    #   pk_mul#2: (v22, v23) = (s1 * v3, c1 * v3) ; pk_fma#1: o2 = c1 * v2 - v22 (lo; reads the HIGH half of v[16:17]) ; pk_fma#2: o3 = s1 * v2 + v23 (hi)
    mul2 = "\tv_pk_mul_f32 v[22:23], v[24:25], v[14:15] op_sel:[1,0] op_sel_hi:[0,0]\n"
    fma1 = "\tv_pk_fma_f32 v[14:15], v[24:25], v[16:17], v[22:23] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
    fma2 = "\tv_pk_fma_f32 v[16:17], v[24:25], v[16:17], v[22:23] op_sel:[0,1,0]\n"
    lines = slp.splitlines(keepends=True)
    a, b = _func_span(lines)
    head, body, tail = "".join(lines[:a]), "".join(lines[a:b]), "".join(lines[b:])
    assert body.count(mul2 + fma1 + fma2) == 1, "the compiled rotation no longer has the analysed form: re-read the ISA"

    def more_vgprs(text):   # the kernel descriptor follows the function body: registers v68..v71 for the rewrites that need fresh ones
        k = text.index(".amdhsa_kernel " + KERNEL)
        e = text.index(".end_amdhsa_kernel", k)
        desc = text[k:e]
        assert ".amdhsa_next_free_vgpr 68" in desc and ".amdhsa_accum_offset 68" in desc
        return text[:k] + desc.replace(".amdhsa_next_free_vgpr 68", ".amdhsa_next_free_vgpr 72").replace(".amdhsa_accum_offset 68", ".amdhsa_accum_offset 72") + text[e:]

    variants["v12_scalar_fma1"] = head + body.replace(fma1, "\tv_fma_f32 v14, v24, v17, -v22\n") + tail
    variants["v13_scalar_mul2"] = head + body.replace(mul2, "\tv_mul_f32 v22, v25, v14\n\tv_mul_f32 v23, v24, v14\n") + tail
    variants["v14_fresh_dst_mul2"] = more_vgprs(head + body.replace(mul2 + fma1 + fma2, (mul2 + fma1 + fma2).replace("v[22:23]", "v[68:69]")) + tail)
    variants["v15_fma1_twice"] = more_vgprs(head + body.replace(fma1 + fma2, fma1.replace("v[14:15], v[24:25]", "v[68:69], v[24:25]") + fma1 + fma2) + tail)
    for name, text in variants.items():
        s, o, h = (os.path.join(OUT, name + ext) for ext in (".s", ".o", ".hsaco"))
        open(s, "w").write(text)
        subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", o, "-o", h], check=True)
        os.remove(o)
        os.remove(s)
        print("built", h)


def run():
    sys.path.insert(0, ROOT)
    import torch
    from sea_amd import _native as N
    from sea_amd.engine import Plan
    from sea_amd.models.temporal import TemporalModel

    class QkvLaunch(C.Structure):
        _fields_ = [("g", N.SeaQkvGroup * 16), ("tile_start", C.c_int32 * 17), ("n_groups", C.c_int32), ("single_buffer", C.c_int32), ("c", N.SeaQkvCommon)]

    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
    m.set_compute_dtype("bf16")
    m = m.to(dev).eval()
    x = torch.randn(1, 2024, 3, 256, generator=torch.Generator().manual_seed(1234)).to(dev)
    ib = torch.rand(1, 2024, 1, generator=torch.Generator().manual_seed(1235)).to(dev)
    eng = m.engine()
    eng.params.sync()
    p = Plan(eng, 1, 2024, "full")
    p.bind(x, ib, torch.empty_like(x))
    p.run()
    torch.cuda.synchronize()
    bufs = [t for t in p._keep if isinstance(t, torch.Tensor)]
    recs = [r for r in p.records if r.fn is not None]
    k = [r.name for r in recs].index("self.qkv_rope")
    stream = torch.cuda.current_stream().cuda_stream
    for r in recs[:k]:
        assert r.fn(*r.args, stream) == 0
    torch.cuda.synchronize()
    pre = [t.clone() for t in bufs]
    arr, n, common = recs[k].args[0], recs[k].args[1], recs[k].keep[1]
    L = QkvLaunch()
    total = 0
    for i in range(n):
        C.memmove(C.byref(L.g[i]), C.byref(arr[i]), C.sizeof(N.SeaQkvGroup))
        L.tile_start[i] = total
        total += ((arr[i].M + 63) // 64) * ((arr[i].N + 63) // 64)
    L.tile_start[n] = total
    L.n_groups, L.single_buffer = n, 1
    C.memmove(C.byref(L.c), C.byref(common), C.sizeof(N.SeaQkvCommon))
    size = C.c_size_t(C.sizeof(L))
    extra = (C.c_void_p * 5)(1, C.cast(C.byref(L), C.c_void_p), 2, C.cast(C.byref(size), C.c_void_p), 3)
    runs = int(os.environ.get("RUNS", "300"))
    names = sorted(f[:-6] for f in os.listdir(OUT) if f.endswith(".hsaco"))
    only = os.environ.get("ONLY")
    ref = None
    for name in names:
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        mod, fn = C.c_void_p(), C.c_void_p()
        assert hip.hipModuleLoad(C.byref(mod), os.path.join(OUT, name + ".hsaco").encode()) == 0, name
        assert hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL.encode()) == 0, name

        def one():
            for t, s in zip(bufs, pre):
                t.copy_(s)
            rc = hip.hipModuleLaunchKernel(fn, total, 1, 1, 256, 1, 1, 16384, C.c_void_p(stream), None, extra)
            assert rc == 0, (name, rc)
            torch.cuda.synchronize()
            return [t.clone() for t in bufs]

        first = one()
        if ref is None:   # the library's own launch, for a value check of the harness
            for t, s in zip(bufs, pre):
                t.copy_(s)
            assert recs[k].fn(*recs[k].args, stream) == 0
            torch.cuda.synchronize()
            ref = [t.clone() for t in bufs]
        vs_lib = sum(int((a.view(torch.uint8) != b.view(torch.uint8)).sum()) for a, b in zip(ref, first))
        bad, where = 0, []
        for it in range(runs):
            cur = one()
            diff = [(i, (a.float() != b.float()).nonzero()) for i, (a, b) in enumerate(zip(first, cur)) if not torch.equal(a.view(torch.uint8), b.view(torch.uint8))]
            if diff:
                bad += 1
                if len(where) < 3:
                    i, idx = diff[0]
                    where.append((i, tuple(first[i].shape), idx.shape[0], idx.min(dim=0).values.tolist(), idx.max(dim=0).values.tolist()))
                    if name.startswith("v0_") and first[i].dim() == 4:
                        _, h_, t_, d_ = idx[0].tolist()
                        d0 = d_ & ~3
                        rope = eng.rope_self   # [max_len, hd/2, 2] (cos, sin)
                        isq = first[i].shape[2] == 2024 and any(arr[g].Qout == bufs[i].data_ptr() for g in range(n))
                        print("   evidence:", "Q" if isq else "K", "buffer", i, "head", h_, "t", t_, "cols", d0, "..", d0 + 3,
                              "good", [round(v, 5) for v in first[i][0, h_, t_, d0:d0 + 4].float().tolist()],
                              "bad", [round(v, 5) for v in cur[i][0, h_, t_, d0:d0 + 4].float().tolist()],
                              "cs", [round(v, 5) for v in rope[t_, d0 // 2:d0 // 2 + 2].flatten().tolist()], flush=True)
        print(f"{name}: {bad} of {runs} replays differ from the first; bytes different from the library's launch: {vs_lib}; {where}", flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
