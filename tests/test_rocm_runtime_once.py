"""Root cause of the `double free or corruption (!prev)` abort at the exit of
round 2's GPU test process (gpurun_out/r2c4_1.log) and of round 4's first run
with an in-process `import torch` (gpurun_out/r4c2_1.log) -- reproduced and
pinned WITHOUT a GPU: the children below only import.

The binding loaded libspmv_scpa_amd.so with RTLD_GLOBAL; the symbols of its
dependency closure (librccl -> librocm_smi64) became process-global, a later
`import torch` bound against them, and glibc aborted inside exit().  The
binding now loads with RTLD_LOCAL and shares ONE copy of the ROCm runtime with
torch (its bundled copies are loaded first, by path, when torch is installed):
either import order ends with one HIP / RCCL / HSA runtime and a clean exit."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, re, sys
sys.path.insert(0, %(root)r)
%(imports)s
seen = {}
for line in open("/proc/self/maps"):
    m = re.search(r"(/\S+\.so\S*)$", line.strip())
    if m:
        base = re.sub(r"\.so.*", ".so", os.path.basename(m.group(1)))
        seen.setdefault(base, set()).add(os.path.realpath(m.group(1)))
for lib in ("libamdhip64.so", "librccl.so", "libhsa-runtime64.so"):
    print("COPIES", lib, len(seen.get(lib, ())), sorted(seen.get(lib, ())))
print("child-ok")
"""

needs_torch = pytest.mark.skipif(importlib.util.find_spec("torch") is None,
                                 reason="torch is not installed")


def run_child(imports, **env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c",
                        CHILD % {"root": ROOT, "imports": imports}],
                       capture_output=True, text=True, timeout=600, env=env)
    copies = {l.split()[1]: int(l.split()[2])
              for l in r.stdout.splitlines() if l.startswith("COPIES")}
    return r, copies


@needs_torch
@pytest.mark.parametrize("imports", [
    "import spmv_scpa_amd\nimport torch",          # the order that aborted
    "import torch\nimport spmv_scpa_amd",          # bench.py's order
    "import spmv_scpa_amd",                        # never imports torch
], ids=["ours_then_torch", "torch_then_ours", "ours_only"])
def test_one_rocm_runtime_whatever_the_import_order(imports):
    r, copies = run_child(imports)
    tail = r.stdout[-1500:] + "\n---- stderr ----\n" + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert "child-ok" in r.stdout, tail
    for word in ("double free", "corruption", "Aborted", "core dumped"):
        assert word not in r.stderr, tail
    assert copies["libamdhip64.so"] == 1 and copies["librccl.so"] == 1, tail
    assert copies["libhsa-runtime64.so"] <= 1, tail


@needs_torch
def test_the_abort_was_the_global_load_and_nothing_else():
    """the old way of loading, restated with bare ctypes: the library with
    RTLD_GLOBAL, then torch -- an abort at exit on a box without a GPU, with
    no handle ever created.  (Documents the cause; skipped should a future
    torch / ROCm pairing no longer collide.)"""
    lib = os.path.join(ROOT, "spmv_scpa_amd", "lib", "libspmv_scpa_amd.so")
    code = ("import ctypes\n"
            "ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)\n"
            "import torch\nprint('imports done')\n" % lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True,
                       text=True, timeout=600)
    assert "imports done" in r.stdout
    if r.returncode == 0:
        pytest.skip("RTLD_GLOBAL + torch no longer collide on this image")
    assert "double free or corruption" in r.stderr
    # ... and the same two loads with RTLD_LOCAL leave cleanly
    r = subprocess.run([sys.executable, "-c",
                        code.replace("RTLD_GLOBAL", "RTLD_LOCAL")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "double free" not in r.stderr, r.stderr


def test_elf_reader_and_runtime_report():
    """the binding's ELF reader names what the loader will match on, and the
    import-time check sees one runtime copy of each library"""
    sys.path.insert(0, ROOT)
    import spmv_scpa_amd as S
    soname, needed = S._elf_dynamic(S.LIB_PATH)
    assert any(n.startswith("libamdhip64.so.") for n in needed), needed
    assert any(n.startswith("librccl.so.") for n in needed), needed
    assert S._elf_dynamic(__file__) == (None, [])
    assert S.ROCM_RUNTIME_ONCE is True
    assert all(len(v) == 1 for v in S.mapped_rocm_runtimes().values())
    if S.ROCM_RUNTIME_SHARED_WITH_TORCH:
        for path in S.ROCM_RUNTIME_SHARED_WITH_TORCH:
            name = os.path.basename(path)
            assert S._elf_dynamic(path)[0] in [
                n for n in needed if n.startswith(name)], path


@needs_torch
def test_a_bundled_runtime_with_another_soname_is_not_preloaded(tmp_path):
    """a torch wheel whose libamdhip64 carries another SONAME
    (another ROCm major) must not be preloaded -- the loader would not take it
    for the library's DT_NEEDED name and two runtimes would be mapped.  Staged
    with a fake `torch` package whose lib/ holds a copy of /opt/rocm's HIP
    runtime patched to SONAME libamdhip64.so.6."""
    src = "/opt/rocm/lib/libamdhip64.so"
    if not os.path.exists(src):
        pytest.skip("no /opt/rocm runtime to stage the fake wheel from")
    pkg = tmp_path / "torch"
    (pkg / "lib").mkdir(parents=True)
    (pkg / "__init__.py").write_text("raise ImportError('fake torch')\n")
    data = open(os.path.realpath(src), "rb").read()
    assert data.count(b"libamdhip64.so.7\0") >= 1
    (pkg / "lib" / "libamdhip64.so").write_bytes(
        data.replace(b"libamdhip64.so.7\0", b"libamdhip64.so.6\0"))
    code = ("import sys, warnings\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    import spmv_scpa_amd as S\n"
            "print('SHARED', S.ROCM_RUNTIME_SHARED_WITH_TORCH)\n"
            "print('WARNED', [str(x.message)[:60] for x in w])\n"
            "print('MAPS', S.mapped_rocm_runtimes())\n" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SHARED None" in r.stdout, r.stdout
    assert "SONAME" in r.stdout and "libamdhip64.so.6" not in r.stdout.split(
        "MAPS")[1], r.stdout
