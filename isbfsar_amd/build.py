"""In-tree build of libisbfsar_hip.so (hipcc, gfx950 only).

    python -m isbfsar_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
working tree to the GPU box. Objects are rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libisbfsar_hip.so")
OBJ = os.path.join(CSRC, "build")

SOURCES = [
    "isb_common.cpp",
    "gemm_f32.hip",
    "ar_kernels.hip",
    "ar_api.cpp",
    "hpe_kernels.hip",
    # the convolution family: one translation unit per kernel family (none above ~90 s), the dispatcher that picks a tile per layer
    "conv_dispatch.hip",
    "conv_igemm.hip",
    "conv_gemm1x1.hip",
    "conv_gemm1x1_gate.hip",
    "conv_3x3.hip",
    "conv_fused_mb.hip",
    "conv_ws.hip",
    "conv_dw_se.hip",
    "hpe_api.cpp",
    "det_kernels.hip",
    "det_api.cpp",
    "dist_api.cpp",
    "rgb_kernels.hip",
    "rgb_api.cpp",
]
# optional units appear as they are written
for _extra in ("conv_mb8.hip", "conv_mb16.hip"):
    if os.path.exists(os.path.join(CSRC, _extra)):
        SOURCES.append(_extra)

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=on"]
# ISB_BUILD_PROBES=1 python -m isbfsar_amd.build --force: also the tile variants / kernel forms that were measured and NOT selected
# (EXPERIMENTS.md, tools/): ~130 more instantiations, a 6-minute build. The product build holds what the three networks launch.
if os.environ.get("ISB_BUILD_PROBES", "0") not in ("", "0"):
    FLAGS.append("-DISB_BUILD_PROBES")
    SOURCES.insert(SOURCES.index("conv_gemm1x1_gate.hip") + 1, "conv_wsk.hip")     # variant 157: measured 2x slower (EXPERIMENTS.md round 5)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libisbfsar_hip.so)")


def _deps_mtime() -> float:
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
GUARD_SRC = "wsreg_guard.cpp"          # defines isb_wsreg_verified(); compiled AFTER the check below with its result


def _disassemble(lib_path: str) -> str:
    """gfx950 disassembly of the library's code object, prefixed with a newline; on failure the reason (no leading newline)."""
    if not os.path.exists(OBJDUMP):
        return f"{OBJDUMP} not found: the built code cannot be verified"
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        r = subprocess.run([OBJDUMP, "--offloading", so], capture_output=True, cwd=tmp)
        if r.returncode != 0:
            return "llvm-objdump --offloading failed: " + r.stderr.decode(errors="replace")[-300:]
        text = "\n"
        for f in sorted(os.listdir(tmp)):
            if f.startswith("lib.so.") and f.endswith("gfx950"):
                d = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True)
                if d.returncode != 0:
                    return "llvm-objdump -d failed: " + d.stderr[-300:]
                text += d.stdout
    return text


def mbfront8_wait_counted(lib_path: str):
    """mbfront8_kernel (csrc/conv_mb8.hip) waits for the next sample's LDS-DMA input tiles with a hand-counted
    `s_waitcnt vmcnt(5)`: exactly five vector-memory operations (four 16-byte D-row stores, one 16-byte pooled-means store) may be
    issued between those requests and the wait. vmcnt(N) waits until all but the N YOUNGEST vector-memory operations are done, so
    the dangerous direction is FEWER operations behind the requests than counted (a store the compiler merged or dropped: the wait
    would then let a tile piece stay in flight at the barrier and the MFMAs would read stale LDS -- silently); MORE
    operations (spill traffic, a split store) only make the wait stricter, but they also mean the code is not the code that was
    measured, so the check is an exact match either way and a future edit must not relax it towards "at most". So the built
    code is checked: every instantiation must be free of scratch instructions and hold exactly five global_store_dwordx4
    between its loop's LDS-DMA requests and the end of the loop body (the stamp stores of the tuning probe follow the loop;
    until the pool moved to dw_mm.h's butterfly the fifth was a global_store_dword). Returns None when the library passes, else
    the reason."""
    text = _disassemble(lib_path)
    if not text.startswith("\n"):
        return text
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1) if "mbfront8_kernel" in m.group(1) else None
            if cur:
                kernels[cur] = []
        elif cur and line.strip():
            kernels[cur].append(line.split("//")[0].split())
    if len(kernels) < 2:
        return f"only {len(kernels)} mbfront8_kernel instantiations found in the code object"
    for name, lines in kernels.items():
        ops = [t[0] for t in lines if t]
        if any(o.startswith("scratch_") for o in ops):
            return f"{name}: scratch (spill) instructions count in vmcnt"
        if "vmcnt(5)" not in " ".join(" ".join(t) for t in lines):
            return f"{name}: the counted wait is gone"
        # the loop body: from the LAST LDS-DMA request (the next sample's tiles) to the backward branch; stamp stores come after it
        last_dma = max((i for i, o in enumerate(ops) if o == "global_load_lds_dwordx4"), default=None)
        if last_dma is None:          # (another lowering of the request, e.g. buffer_load ... lds: fail closed, not with a traceback)
            return f"{name}: no global_load_lds_dwordx4 found -- the input requests are lowered differently than the check knows"
        vm = [o for o in ops[last_dma + 1:] if o.startswith(("global_", "buffer_", "flat_"))]
        body = vm[:5]
        if body != ["global_store_dwordx4"] * 5:
            return f"{name}: expected five global_store_dwordx4 behind the input requests, found {vm[:8]}"
        if any(o not in ("global_store_dwordx2", "flat_store_dwordx2") for o in vm[5:]):      # (behind the loop: only the probe's 8-byte stamp stores, volatile in the source)
            return f"{name}: vector-memory operations besides the five stores behind the input requests: {vm[5:]}"
    return None


# kernels whose f32 -> fp16 conversions need no saturation, with the reason
FP16_NO_OVERFLOW = {"ar_proto_all_kernel": "converts softmax probabilities exp2(s' - lse) <= 1 (the MFMA's P operand)"}


def fp16_conversions_saturate(lib_path: str, text: str = None):
    """T16<true>::pack2 / from_f32 no longer clamp -- they rely on MODE.FP16_OVFL, which a kernel sets by calling
    T16<F16>::enter() (conv_common.h); a kernel that forgets it would store inf and poison every later layer, silently. Enforced on
    the built code: every kernel of the code object that executes a v_cvt_*f16_f32 must EITHER set the mode bit
    (s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1)) OR clamp every converted value explicitly (f2h_: at least one v_med3_f32 per
    conversion; the stem, the weight conversion, the AR tuple images) OR be listed in FP16_NO_OVERFLOW with its reason.
    Returns None when the library passes, else the reason."""
    if text is None:
        text = _disassemble(lib_path)
    if not text.startswith("\n"):
        return text
    kern, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kern[cur] = [0, 0, 0]
        elif cur and line.strip():
            t = line.split("//")[0]
            if re.search(r"\bv_cvt_(pk_|pkrtz_)?f16_f32", t):
                kern[cur][0] += 1
            elif "s_setreg_imm32_b32" in t and "HW_REG_MODE, 23, 1), 1" in t:
                kern[cur][1] += 1
            elif "v_med3_f32" in t:
                kern[cur][2] += 1
    n_mode = 0
    for name, (cvt, mode, med3) in kern.items():
        if not cvt:
            continue
        if mode:
            n_mode += 1
        elif med3 < cvt and not any(k in name for k in FP16_NO_OVERFLOW):
            return f"{name}: {cvt} f32 -> fp16 conversions, MODE.FP16_OVFL never set and only {med3} explicit clamps (T16<F16>::enter() missing?)"
    if n_mode < 30:
        return f"only {n_mode} kernels set MODE.FP16_OVFL: the disassembly was not understood"
    return None


def wspipe_registers_private(lib_path: str):
    """gemm1x1_wspipe_kernel (csrc/conv_ws.hip) keeps global loads in flight in literally named registers -- a[200:255]
    with one wave per SIMD, v[228:255] with two -- that the register allocator sees only as clobbers of the request asm.
    Whether the compiler kept its hands off them depends on the hipcc that built the library, so the built code is
    checked: disassemble every gemm1x1_wspipe_kernel instantiation and require that nothing but the requests
    (global_load_dwordx4) and their LDS writes (ds_write_b128) names a staging register. Returns None when the library
    passes, else the reason (also when llvm-objdump is missing: unverified = not trusted)."""
    text = _disassemble(lib_path)
    if not text.startswith("\n"):
        return text
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1) if "gemm1x1_wspipe_kernel" in m.group(1) else None
            if cur:
                kernels[cur] = []
        elif cur and line.strip():
            kernels[cur].append(line.split("//")[0])
    if len(kernels) < 10:       # variants 184 (NK 3 / 6 / 7), 185, 186 (NK 12) x bf16 / fp16 operands; probe builds have more
        return f"only {len(kernels)} gemm1x1_wspipe_kernel instantiations found in the code object"

    def regs(line, letter):
        out = set()
        for lo, hi in re.findall(rf"\b{letter}\[(\d+):(\d+)\]", line):
            out.update(range(int(lo), int(hi) + 1))
        out.update(int(n) for n in re.findall(rf"\b{letter}(\d+)\b", line))
        return out

    for name, lines in kernels.items():
        m = re.search(r"ILi(\d+)ELb[01]ELb[01]ELi(\d+)ELi(\d+)ELi(\d+)ELb[01]ELb[01]EEE", name)
        if not m:
            return f"cannot parse the template arguments of {name}"
        nk, tmb, wpc, nwm = (int(x) for x in m.groups())
        two_waves = wpc * nwm == 2                  # waves per SIMD
        letter, lo = ("v", 228) if two_waves else ("a", 200)
        n_req = n_wr = 0
        for line in lines:
            if not any(x >= lo for x in regs(line, letter)):
                continue
            op = next((t for t in line.split() if t.startswith(("global_", "ds_", "v_", "s_", "buffer_", "scratch_"))), "")
            if op == "global_load_dwordx4":
                n_req += 1
            elif op == "ds_write_b128":
                n_wr += 1
                if two_waves and not all(x >= lo for x in regs(line.split(",", 1)[1], "v")):
                    return f"{name}: LDS write mixes staging and ordinary registers: {line.strip()}"
            else:
                return f"{name}: staging register in a foreign instruction: {line.strip()}"
        if not (n_req > 0 and n_wr > 0):
            return f"{name}: no staged requests found"
    return None


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".", "_") + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr.strip())

    if jobs:
        with ThreadPoolExecutor(max_workers=min(int(os.environ.get("ISB_BUILD_JOBS", "8")), len(jobs))) as ex:
            list(ex.map(run, jobs))
    guard_o = os.path.join(OBJ, "wsreg_guard.o")
    if jobs or force or not os.path.exists(LIB) or not os.path.exists(guard_o):
        # fail closed: link once without the guard's verdict, check the code the compiler actually produced,
        # then compile the verdict in. isb_wsreg_verified() == 0 makes conv_dispatch.hip fall back to the tile kernels.
        def guard(ok: int, ok8: int):
            run([hipcc, "-O2", "-std=c++17", "-fPIC", f"-DISB_WSREG_VERIFIED={ok}", f"-DISB_MBF8_VERIFIED={ok8}", "-c",
                 os.path.join(CSRC, GUARD_SRC), "-o", guard_o])
            run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, guard_o, "-ldl", "-o", LIB])
        guard(0, 0)
        why = wspipe_registers_private(LIB)
        why8 = mbfront8_wait_counted(LIB)
        if why is not None:
            print(f"[build] WARNING: weights-stationary expand kernels DISABLED (tile kernels are used instead): {why}", flush=True)
        if why8 is not None:
            print(f"[build] WARNING: fused front of the 8x8 MBConv blocks DISABLED (expand + depthwise launches run instead): {why8}", flush=True)
        if why is None or why8 is None:
            guard(int(why is None), int(why8 is None))
        why16 = fp16_conversions_saturate(LIB)
        if why16 is not None:       # no fallback exists for a kernel that could store inf: refuse the library
            os.remove(LIB)
            raise RuntimeError(f"fp16 storage kernels must saturate: {why16}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
