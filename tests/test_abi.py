"""CPU: the C-ABI library loads without a GPU and exports exactly what include/aoenv.h declares; the
ctypes mirror of AoCfg and of the enums matches the header (checked with gcc).  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "aoenv.h")


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, REPO)
    import __graft_entry__ as g
    g.build()
    from rlao_amd import _lib
    return _lib


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aoenv_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    lib = built_lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in aoenv.h but not exported by libaoenv.so"
        assert n in built_lib.EXPORTS, f"{n} has no ctypes prototype in rlao_amd/_lib.py"
    for n in built_lib.EXPORTS:
        assert n in names, f"{n} is bound in _lib.py but not declared in aoenv.h"
    assert lib.aoenv_abi_version() == built_lib.ABI_VERSION


def _header_enumerators():
    """Every AOENV_* enumerator of the header's enum blocks (not the #defines)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = []
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", src, flags=re.S):
        names += re.findall(r"\b(AOENV_[A-Z0-9_]+)\b", body)
    return sorted(set(names))


def test_cfg_layout_and_enums_match_header(built_lib, tmp_path):
    L = built_lib
    structs = {"AoCfg": L.AoCfg, "AoDetector": L.AoDetector}
    enums = _header_enumerators()
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for sname, st in structs.items():
        prog.append(f'printf("{sname}.size %zu\\n", sizeof({sname}));')
        prog += [f'printf("{sname}.{f[0]} %zu\\n", offsetof({sname}, {f[0]}));' for f in st._fields_]
    prog += [f'printf("{e} %d\\n", (int){e});' for e in enums]
    prog += ['printf("AOENV_ABI_VERSION %d\\n", (int)AOENV_ABI_VERSION);', "return 0;}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for sname, st in structs.items():
        assert int(out[f"{sname}.size"]) == C.sizeof(st), sname
        for f in st._fields_:
            assert int(out[f"{sname}.{f[0]}"]) == getattr(st, f[0]).offset, (sname, f[0])
    assert int(out["AOENV_ABI_VERSION"]) == L.ABI_VERSION
    # the Python mirror of every enumerator: AOENV_C_X -> _lib.C_X, AOENV_B_X -> B_X, AOENV_OPT_X -> OPT_X, AOENV_F32/F64/WFS_*
    checked = 0
    for e in enums:
        short = e[len("AOENV_"):]
        if short.startswith("K_"):
            continue                                            # AoKernel: mirrored by the order of KERNEL_NAMES
        if short.endswith("_COUNT"):
            continue
        assert hasattr(L, short), f"{e} has no mirror {short} in rlao_amd/_lib.py"
        assert int(out[e]) == getattr(L, short), e
        checked += 1
    assert checked >= 40
    assert int(out["AOENV_K_COUNT"]) == len(L.KERNEL_NAMES)
    assert int(out["AOENV_B_COUNT"]) == 1 + max(getattr(L, n) for n in dir(L) if n.startswith("B_"))
    assert int(out["AOENV_C_COUNT"]) == 1 + max(getattr(L, n) for n in dir(L) if n.startswith("C_") and isinstance(getattr(L, n), int))


def test_no_gpu_means_loud_failure(built_lib):
    """Without a GPU the env refuses to exist (there is no CPU fallback) and create() reports an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rlao_amd.env import BatchedAOEnv
    with pytest.raises(built_lib.AoEnvError):
        BatchedAOEnv(n_envs=1)
    lib = built_lib.load()
    cfg = built_lib.AoCfg(abi_version=built_lib.ABI_VERSION, dtype=0, n_env=1, resolution=24, n_layer=0, layer_res=28,
                          n_inner=208, n_outer=116, n_act=5, n_valid_act=21, dm_separable=1, wfs_type=0, n_subap=4,
                          n_valid_subap=12, n_signal=24, cam_res=24, n_loop=8, max_group=1)
    h = C.c_void_p()
    assert lib.aoenv_create(C.byref(cfg), 0, C.byref(h)) != 0
    assert len(lib.aoenv_last_error()) > 0
    bad = built_lib.AoCfg(abi_version=999)
    assert lib.aoenv_create(C.byref(bad), 0, C.byref(h)) != 0
    assert b"ABI" in lib.aoenv_last_error()


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    """the host side of libaoenv built with -fsanitize=address,undefined (`make asan`; device code as
    usual), driven through the ABI's validation and error paths by a sanitized C driver.  No sanitizer report, exit code 0."""
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang")
    subprocess.run(["make", "-C", os.path.join(REPO, "rlao_amd", "csrc"), "asan", "-j4"], check=True, capture_output=True)
    lib_dir = os.path.join(REPO, "build", "asan")
    exe = tmp_path / "asan_driver"
    subprocess.run([clang, "-std=c11", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", f"-I{os.path.join(REPO, 'include')}",
                    os.path.join(REPO, "tests", "native", "asan_driver.c"), "-o", str(exe), f"-L{lib_dir}", "-laoenv_asan",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "asan driver ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr
