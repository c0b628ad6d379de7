"""CPU: the C-ABI library loads without a GPU and exports exactly the symbols include/modegpt_hip.h declares."""
import os
import re
import subprocess

import pytest

from modegpt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "modegpt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(mdg_[a-z0-9_]+)\s*\(", src))


def test_header_matches_binding_table():
    assert header_symbols() == set(_lib.SIGNATURES)


def test_library_loads_and_exports_everything():
    lib = _lib.load()  # no GPU needed to dlopen and bind
    assert lib.mdg_abi_version() == _lib.ABI_VERSION == 9
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mdg_[a-z0-9_]+)", out))
    assert header_symbols() <= exported
    assert exported == header_symbols(), f"undeclared exports: {exported - header_symbols()}"


def test_pure_size_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.mdg_cov_accum_ws_bytes(32768, 14336, 1) == 0          # enough tiles: no split-K
    assert lib.mdg_cov_accum_ws_bytes(32768, 128, 8) > 0             # head Grams split over tokens
    n = 14336
    assert lib.mdg_ridge_scores_ws_bytes(n) >= (2 * n * n + n * n // 4) * 8
    assert lib.mdg_potrf_inv_diag_elems(300) == 3 * 128 * 128 + 16


def test_bad_arguments_return_status_not_abort():
    lib = _lib.load()
    rc = lib.mdg_cov_accum(None, 99, 10, 10, 1, 10, 0, None, 10, 100, None, 0, None)
    assert rc == _lib.MDG_ERR_BAD_ARG and b"dtype" in lib.mdg_last_error()
    rc = lib.mdg_qk_select(None, None, 4, 2, 16, 0.0, 0.0, 8, 0, None, None, None, None)
    assert rc == _lib.MDG_ERR_BAD_ARG


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/modegpt_hip.h compiles as C99 with gcc (no C++, no HIP, no torch types), and a
    C program linked against libmodegpt_hip.so can call the entry points that need no device (version, size queries,
    argument checks -> status + thread-local message)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "modegpt_amd")
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "modegpt_hip.h"
int main(void) {
  if (mdg_abi_version() < 1) return 1;
  if (mdg_potrf_inv_diag_elems(1000) <= 0) return 2;
  if (mdg_ridge_scores_ws_bytes(14336) == 0) return 3;
  /* r = 3 is odd: refused before any device work */
  int rc = mdg_rope_gather((const void*)0, MDG_BF16, 12, 1, 1, 4, 2, 3, 16, (const void*)0, (const void*)0, 0,
                           (const int64_t*)0, (const void*)0, 1e-6, (void*)0, (void*)0);
  if (rc != MDG_ERR_BAD_ARG) return 4;
  if (strstr(mdg_last_error(), "even") == NULL) return 5;
  printf("abi %d ok\n", mdg_abi_version());
  return 0;
}
''')
    exe = tmp_path / "abi"
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{os.path.join(root, 'include')}", str(src), "-o", str(exe),
           f"-L{lib_dir}", "-lmodegpt_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "ok" in run.stdout
