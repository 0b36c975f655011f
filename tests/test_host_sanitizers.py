"""The library's host-only builders under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU
build; GPU ASan is not available on the pool).  mpc-sensorlessao_amd/csrc/fmpc_host.cpp -- the iteration-invariant Y blocks and
their de-duplication, the twisted block factorisation in long double with its operator images and sweep schedules, the
least-recently-used cache of fmpc_solve_once -- is plain C++: tests/host_san/host_build_test.cpp links it with g++ and
checks numerically that the two sweeps, executed step by step from the schedules, reproduce a dense solve of Y."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc-sensorlessao_amd", "csrc")


@pytest.fixture(scope="module")
def san_binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("host_san") / "host_build_test")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-Wall", "-Wextra", os.path.join(ROOT, "tests", "host_san", "host_build_test.cpp"), os.path.join(CSRC, "fmpc_host.cpp"),
           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("T", [1, 2, 3, 7, 8, 9, 30])
@pytest.mark.parametrize("has_xf", [0, 1])
def test_host_builders_under_asan_ubsan(san_binary, T, has_xf):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for var_order in ((2, 1) if T in (3, 30) else (2,)):
        r = subprocess.run([san_binary, "27", "144", str(T), str(has_xf), str(var_order), str(100 + T)], capture_output=True, text=True,
                           env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr


@pytest.fixture(scope="module")
def ramp_cold_binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("host_san_ramp") / "ramp_cold_test")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-Wall", "-Wextra", os.path.join(ROOT, "tests", "host_san", "ramp_cold_test.cpp"), os.path.join(CSRC, "fmpc_host.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("cfg", ["8 5 6 0 1 1", "8 5 10 1 1 2", "5 8 4 0 1 3", "8 5 6 0 2 4", "6 4 1 0 1 5", "27 144 2 0 1 6", "27 144 3 1 2 7"])
def test_ramp_cold_form_builder_under_asan_ubsan(ramp_cold_binary, cfg):
    """fmpc_host_build_ramp_cold (the cold-start step with the ramp-rate rows as a constant KKT matrix plus a rank-m diagonal term,
    VAR_1/fast_mpc_ineq_const.m:58-76): the constants it builds, used exactly as the device kernel uses them, reproduce an
    independent dense solve of the full KKT system of each problem (long double) to 1e-11 -- VAR(1) and VAR(2), with and without
    terminal rows and disturbance, T = 1, and the AO sizes."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([ramp_cold_binary] + cfg.split(), capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
