"""Static check of the hand-counted memory waits of `fmpc_cold_affine` (csrc/fmpc_kernel_affine.hip).

The kernel requests the next tile's operand image with `global_load_dwordx2` through inline assembly the compiler does not
see as a load, and waits for it later with a hand-counted `s_waitcnt vmcnt(N)` (`fa_request_a` / `fa_await_a<N>`): between the
two the compiler believes the destination registers are valid.  A register-allocation change (compiler bump, other flags)
could insert a copy or a spill of such a register in that window and z would silently be wrong.  This test compiles the file
to gfx950 assembly (no GPU needed) and walks every kernel's instruction stream with the vmcnt model of the ISA -- one
in-order counter for loads and stores -- asserting that NO instruction reads or writes a register with a load still in
flight, and that the kernels use no scratch at all.  The bitwise affine-vs-three-kernel GPU test stays the numerical gate."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpc-sensorlessao_amd", "csrc", "fmpc_kernel_affine.hip")
HIPCC = "/opt/rocm/bin/hipcc"

VM_OP = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _parse(asm):
    """{kernel name: [(label or None, mnemonic, operand text)]} for every .amdhsa kernel function."""
    kernels, cur, name = {}, None, None
    knames = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", asm))
    for line in asm.splitlines():
        s = line.split(";")[0].rstrip()
        m = re.match(r"^(\S+):\s*$", s)
        if m:
            if m.group(1) in knames:
                name, cur = m.group(1), []
                kernels[name] = cur
            elif cur is not None:
                cur.append((m.group(1), None, None))
            continue
        if cur is None or not s.startswith("\t") or s.strip().startswith("."):
            continue
        parts = s.strip().split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else ""))
        if parts[0] == "s_endpgm":
            pass
    return kernels


def _scan(ins, start, stop, queue, errors, kname):
    """Walk ins[start:stop]; queue = in-flight memory operations, oldest first: set of destination registers (empty for stores)."""
    for i in range(start, stop):
        label, op, args = ins[i]
        if op is None:
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", args)
            if m:
                keep = int(m.group(1))
                del queue[:max(0, len(queue) - keep)]
            elif re.fullmatch(r"\s*\d+\s*|0x[0-9a-fA-F]+", args.strip()):
                queue.clear()                                   # a raw immediate: treat as a full wait
            continue
        is_load = bool(VM_OP.match(op)) and "_load" in op
        # (a load INTO a register with an older load in flight is fine: loads return in order; its address operands are checked)
        used = _regs(args.split(",", 1)[1] if is_load and "," in args else args)
        for dest, where in queue:
            hit = used & dest
            if hit:
                errors.append("%s: `%s %s` touches %s while the load issued at instruction %d is still in flight"
                              % (kname, op, args, sorted(hit), where))
        if VM_OP.match(op):
            dest = set()
            if "_load" in op or ("atomic" in op and "glc" in args):
                dest = _regs(args.split(",")[0])
            queue.append((dest, i))


def check_kernel(kname, ins):
    errors = []
    labels = {lab: i for i, (lab, op, _) in enumerate(ins) if op is None}
    queue = []
    i = 0
    done_backedges = set()
    # linear order, and once more around every loop with the state at its back-edge (loads requested at the bottom of an
    # iteration are awaited at the top of the next one)
    _scan(ins, 0, len(ins), queue, errors, kname)
    for i, (lab, op, args) in enumerate(ins):
        if op and op.startswith("s_cbranch") or op == "s_branch":
            tgt = args.strip()
            if tgt in labels and labels[tgt] < i and i not in done_backedges:
                done_backedges.add(i)
                q = []
                _scan(ins, labels[tgt], i, q, [], kname)        # state at the bottom of one iteration ...
                _scan(ins, labels[tgt], i, q, errors, kname)    # ... carried around the back-edge
    return errors


@pytest.fixture(scope="module")
def affine_asm(tmp_path_factory):
    if not os.path.exists(HIPCC) or shutil.which("make") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "affine.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    return open(out).read()


def test_checker_finds_a_planted_hazard():
    ins = [(None, "global_load_dwordx2", "v[4:5], v[0:1], off"), (None, "v_add_f64", "v[6:7], v[4:5], v[2:3]"),
           (None, "s_waitcnt", "vmcnt(0)"), (None, "v_add_f64", "v[6:7], v[4:5], v[2:3]")]
    errs = check_kernel("k", ins)
    assert len(errs) == 1 and "instruction 0" in errs[0]
    ok = [(None, "global_load_dwordx2", "v[4:5], v[0:1], off"), (None, "global_store_dwordx2", "v[0:1], v[8:9], off"),
          (None, "s_waitcnt", "vmcnt(1)"), (None, "v_add_f64", "v[6:7], v[4:5], v[2:3]")]
    assert check_kernel("k", ok) == []
    # a loop whose request at the bottom is awaited at the top of the next iteration: clean; without the wait: flagged
    loop = [("L", None, None), (None, "s_waitcnt", "vmcnt(0)"), (None, "v_mov_b32", "v9, v4"),
            (None, "global_load_dwordx2", "v[4:5], v[0:1], off"), (None, "s_cbranch_scc1", "L")]
    assert check_kernel("k", loop) == []
    bad = [("L", None, None), (None, "v_mov_b32", "v9, v4"), (None, "global_load_dwordx2", "v[4:5], v[0:1], off"),
           (None, "s_cbranch_scc1", "L")]
    assert len(check_kernel("k", bad)) == 1


def test_affine_kernels_have_no_scratch_and_no_register_touched_under_a_load_in_flight(affine_asm):
    kernels = _parse(affine_asm)
    names = [k for k in kernels if "fmpc_cold_affine" in k]
    assert len(names) >= 2, list(kernels)                       # <true> and <false>
    for blk in re.findall(r"\.amdhsa_kernel\s+\S*fmpc_cold_affine.*?\.end_amdhsa_kernel", affine_asm, flags=re.S):
        assert re.search(r"\.amdhsa_private_segment_fixed_size\s+0\b", blk), "the affine kernel uses scratch"
    assert not re.search(r"\bscratch_(load|store)", "\n".join("%s %s" % (op, a) for k in names for _, op, a in kernels[k] if op))
    for k in names:
        ins = kernels[k]
        assert sum(1 for _, op, _ in ins if op == "global_load_dwordx2") >= 28          # two request blocks of 14
        assert any(op == "s_waitcnt" and "vmcnt(16)" in a for _, op, a in ins)          # the hand-counted wait is there
        errs = check_kernel(k, ins)
        assert not errs, "\n".join(errs[:10])
