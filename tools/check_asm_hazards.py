#!/usr/bin/env python3
"""Static check of the one hazard hipcc cannot see: a register that an inline-asm load (`global_load_dword vN, ...` / `buffer_load_dwordx4 v[a:b], ...` between
;;#ASMSTART / ;;#ASMEND) is still filling must not be read before the inline-asm `s_waitcnt vmcnt(..)` that retires it.  The
compiler believes the value exists as soon as the asm statement has "executed", so any copy it schedules in between (PHI
copies of a switch over "+v" operands did exactly that in dw_bwd_slab: non-finite gradients now and then) reads stale data.

    python tools/check_asm_hazards.py [file.hip ...]      # exit 1 on a finding
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "titanet_amd", "csrc")
sys.path.insert(0, ROOT)
from titanet_amd.csrc.build import FLAGS, SOURCES  # noqa: E402


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4..7}"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_asm(text):
    findings = []
    func = None
    pending = {}          # register -> line of the asm load filling it
    lds_ops = []          # inline-asm LDS operations in program order: set of destination registers (empty for writes), or
                          # None once retired — they retire IN ORDER, so an asm `s_waitcnt lgkmcnt(N)` retires all but the last N
    in_asm = False
    for n, line in enumerate(text.split("\n"), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, pending, lds_ops = m.group(1), {}, []
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0]
        if in_asm:
            if re.match(r"s_waitcnt\s+vmcnt", code):
                pending = {}
            m = re.match(r"(?:global|buffer)_load_dword(?:x\d)?\s+(v\d+|v\[\d+:\d+\])\s*,", code)
            if m and "lds" not in code:
                for r in regs_of(m.group(1)):
                    pending[r] = n
            # hand-counted LDS pipelines (tn_mfma_sched_lds, rwgemm_k512_v2_kernel)
            m = re.match(r"s_waitcnt\s+lgkmcnt\((\d+)\)", code)
            if m:
                keep = int(m.group(1))
                live = [i for i, o in enumerate(lds_ops) if o is not None]
                for i in (live[:-keep] if keep else live):
                    lds_ops[i] = None
            m = re.match(r"ds_read\w*\s+(v\d+|v\[\d+:\d+\])\s*,", code)
            if m:
                lds_ops.append((regs_of(m.group(1)), n))
            elif re.match(r"ds_write", code):
                lds_ops.append((set(), n))
            continue
        if code.startswith("s_endpgm"):
            pending, lds_ops = {}, []
            continue
        if any(o is not None and o[0] for o in lds_ops):
            if re.match(r"s_waitcnt\s+.*lgkmcnt\(0\)", code):
                lds_ops = []
            else:
                ops = code.split(None, 1)
                used = regs_of(ops[1]) if len(ops) > 1 else set()
                inflight = set().union(*[o[0] for o in lds_ops if o is not None])
                hit = used & inflight
                if hit:
                    findings.append((func, n, code + "   [in-flight inline-asm ds_read]", sorted(hit)))
        if pending:
            if re.match(r"s_waitcnt\s+vmcnt\(0\)", code):
                pending = {}
                continue
            ops = code.split(None, 1)
            used = regs_of(ops[1]) if len(ops) > 1 else set()
            hit = used & set(pending)
            if hit:
                findings.append((func, n, code, sorted(hit)))
    return findings


STORE_WINDOW = 12      # instructions after the store in which a write of its data registers is flagged


def check_store_data(text):
    """Second hazard (round 5, found in se_combine_fwd_v3_kernel on gfx950): a buffer store of more than 8 bytes whose SCALAR offset
    is an SGPR, followed within a few instructions by a VALU write (or a load) into one of its data registers — hipcc's hazard
    recogniser only covers the immediate-offset form, and the last lanes of the store went out with the NEW register contents.
    Flags any write of a >64-bit store's data registers within STORE_WINDOW instructions when the store's soffset is a register."""
    findings = []
    func = None
    recent = []           # (line, data registers, text) of the flagged-form stores still inside the window
    for n, line in enumerate(text.split("\n"), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, recent = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0].strip()
        if not code or code.endswith(":"):
            continue
        ops = code.split(None, 1)
        if len(ops) > 1 and recent:
            # destination = first operand of everything but stores / compares into SGPRs / s_* instructions
            if not re.match(r"(buffer_store|global_store|flat_store|scratch_store|ds_write|s_|v_cmp|v_readlane|v_readfirstlane)", ops[0]):
                dst = regs_of(ops[1].split(",")[0])
                for ln, regs, txt in recent:
                    hit = dst & regs
                    if hit:
                        findings.append((func, n, f"{code}   [data of `{txt}` at line {ln}]", sorted(hit)))
        recent = [(ln, regs, txt) for ln, regs, txt in recent if n - ln < STORE_WINDOW]
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\])\s*,\s*(\S+)\s*,\s*(s\[\d+:\d+\])\s*,\s*(\S+)", code)
        if m and re.match(r"s\d+$", m.group(4)):
            recent.append((n, regs_of(m.group(1)), code))
    return findings


def main():
    if len(sys.argv) > 1 and sys.argv[1].endswith(".s"):      # an ISA listing made elsewhere (tuning harnesses)
        f = check_asm(open(sys.argv[1]).read()) + check_store_data(open(sys.argv[1]).read())
        print(f"{sys.argv[1]}: {len(f)} reads of in-flight asm-load registers")
        for func, n, code, regs in f[:40]:
            print(f"   {func[:60]} line {n}: {code}   (v{regs})")
        sys.exit(1 if f else 0)
    srcs = [os.path.basename(a) for a in sys.argv[1:]] or SOURCES
    from concurrent.futures import ThreadPoolExecutor

    def one(src):
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + [f for f in FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", "-"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        return src, check_asm(r.stdout) + check_store_data(r.stdout)
    bad = 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        for src, f in ex.map(one, srcs):
            print(f"{src}: {len(f)} reads of in-flight asm-load registers / early writes of store data")
            for func, n, code, regs in f[:20]:
                print(f"   {func[:60]} line {n}: {code}   (v{regs})")
            bad += len(f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
