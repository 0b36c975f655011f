"""Static check of the hand-ordered LDS stores that fold the strict lower triangle of L into the factor stream
(csrc/fmpc_kernel_wave.hip, `fw_phase_factor`: "every lane 1 .. N-1 writes ALL its N - 1 entries at base + j, j DESCENDING").

The fold relies on the ORDER of a wavefront's LDS writes across lanes: what a lane writes beyond its own entries is overwritten by
the owner of the slot at a smaller j, i.e. LATER.  The compiler would pair neighbouring stores into `ds_write2_b64` (which writes
its LOWER slot first) and is free to reorder stores that cannot alias within a lane -- so the stores are issued by hand, one
`ds_write_b64` per entry.  This test compiles the file to gfx950 assembly (no GPU needed) and asserts, in both instances of
`fw_iteration<27, .>`'s non-export code path, that there is a run of N - 1 = 26 consecutive `ds_write_b64` on ONE address register
with offsets 8 (N - 2), ..., 8, 0 in exactly that order and nothing but those stores in between.  The GPU parity tests stay the
numerical gate."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpc-sensorlessao_amd", "csrc", "fmpc_kernel_wave.hip")
HIPCC = "/opt/rocm/bin/hipcc"
N = 27


@pytest.fixture(scope="module")
def wave_asm(tmp_path_factory):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "wave.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), SRC],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _functions(asm):
    """{symbol: [instruction text]} for every function of the file."""
    funcs, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = []
            funcs[m.group(1)] = cur
            continue
        if cur is None:
            continue
        s = line.split(";")[0].strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        cur.append(s)
    return funcs


def test_fold_of_L_is_issued_as_single_stores_in_descending_order(wave_asm):
    funcs = _functions(wave_asm)
    # the per-problem-factor instance (EX = false) streams the folded L; the export instance (EX = true) does not
    name = next(k for k in funcs if "fw_iteration" in k and "Lb0" in k)
    ins = funcs[name]
    want = [8 * j for j in range(N - 2, -1, -1)]
    pat = re.compile(r"^ds_write_b64\s+(v\d+),\s*v\[\d+:\d+\](?:\s+offset:(0x[0-9a-fA-F]+|\d+))?$")
    off = lambda m_: int(m_.group(2) or "0", 0)
    found = 0
    i = 0
    while i < len(ins):
        m = pat.match(ins[i])
        if m and off(m) == want[0]:
            reg = m.group(1)
            offs = []
            j = i
            while j < len(ins) and len(offs) < len(want):
                mj = pat.match(ins[j])
                if not mj or mj.group(1) != reg:
                    break
                offs.append(off(mj))
                j += 1
            if offs == want:
                found += 1
                i = j
                continue
        i += 1
    assert found >= 1, "no run of %d single ds_write_b64 with descending offsets on one address register in %s" % (N - 1, name)
    # and no paired store on the way: between the first and the last store of such a run there is nothing else (checked above by
    # the consecutive match); the export instance must not contain the run at all (it streams row-major tiles)
    ex = next(k for k in funcs if "fw_iteration" in k and "Lb1" in k)
    run = 0
    for t in funcs[ex]:
        m = pat.match(t)
        run = run + 1 if (m and off(m) == want[run if run < len(want) else 0]) else (1 if (m and off(m) == want[0]) else 0)
        assert run < len(want), "the export instance carries the fold"
