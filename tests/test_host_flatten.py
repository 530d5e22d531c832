"""Host half of the library on a CPU: rpt_amd/csrc/rpt_capi.cpp compiled with malloc-backed HIP stubs
(tests/host/) commits a room, a mesh scene, a scene of groups with a shared mesh, a scene in the reference-epsilon mode (groups as
records under frames, per-triangle constants with the reference's own operation order) and a few invalid shapes.

  * under AddressSanitizer + UBSan (g++): no report, exit code 0;
  * at -O0, -O3 and -O3 -fno-unroll-loops: byte-identical flattened scenes (arena checksums).  The last
    variant once produced different trees: `(&v.x)[a]` is undefined behaviour and clang dropped the y / z
    iterations of a loop it did not unroll.

No GPU, no oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "flatten_harness.cpp")
COMMON = ["-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")]


def _build_and_run(tmp_path, name, compiler, flags):
    exe = str(tmp_path / name)
    subprocess.check_call([compiler] + flags + COMMON + [SRC, "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-2000:]
    return p.stdout


@pytest.mark.timeout(600)
def test_flatten_is_clean_under_sanitizers_and_independent_of_the_optimiser(tmp_path):
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("HIP headers not installed")
    outs = {"g++ -O1 asan+ubsan": _build_and_run(tmp_path, "san", "g++", ["-O1", "-g", "-fsanitize=address,undefined",
                                                                           "-fno-omit-frame-pointer"])}
    outs["g++ -O0"] = _build_and_run(tmp_path, "o0", "g++", ["-O0"])
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if os.path.exists(clang) or shutil.which("clang++"):
        clang = clang if os.path.exists(clang) else shutil.which("clang++")
        outs["clang++ -O3"] = _build_and_run(tmp_path, "c3", clang, ["-O3"])
        outs["clang++ -O3 -fno-unroll-loops"] = _build_and_run(tmp_path, "c3nu", clang, ["-O3", "-fno-unroll-loops"])
    first = outs["g++ -O1 asan+ubsan"]
    for k, v in outs.items():
        assert v == first, f"{k} flattened the scenes differently:\n{v}\nvs\n{first}"
    lines = {l.split()[0]: l.split() for l in first.splitlines() if l.split()[0] in ("room", "mesh", "groups", "empty")}
    # name rc spheres cubes planes tris aabbs rects bvh_tris bvh_nodes scan_bytes scene_bytes scene_bvh prims instances shared shell
    assert lines["room"][1] == "rc=0" and lines["room"][3:8] == ["1", "0", "0", "1", "6"] and lines["room"][16] == "5"
    assert lines["mesh"][2:6] == ["1", "0", "1", "18"] and lines["mesh"][8] == "2048" and lines["mesh"][12] == "0"
    assert lines["groups"][2:4] == ["30", "40"] and lines["groups"][12:16] == ["1", "110", "40", "1"] and lines["groups"][8] == "192"
    eps = [l for l in first.splitlines() if l.startswith("epsilon ")]
    assert len(eps) == 1 and "rc=0" in eps[0] and "records=9 " in eps[0]     # 4 plain objects + the 5 shapes inside the (nested) groups
    assert "bad kind rc=-1" in first and "empty mesh rc=-1" in first and "singular rc=-1" in first and "second commit rc=-2" in first
